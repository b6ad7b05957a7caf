import sqlite3, sys
db = sys.argv[1]
con = sqlite3.connect(db); cur = con.cursor()
rows = list(cur.execute("select name,duration,grid_x,grid_y,grid_z,lds_size,vgpr_count,start from kernels order by start"))
names=[r[0] for r in rows]
idx=[i for i,n in enumerate(names) if 'bilinear_up2' in n]
a,b=idx[-2]+1, idx[-1]+1
tot=0; agg={}
for r in rows[a:b]:
    if len(sys.argv)>2: print(f"{r[0].replace('void ','')[:44]:44s} {r[1]/1000:9.1f} us grid {r[2]//256 if r[2] else 0}x{r[3]}x{r[4]} lds {r[5]} vgpr {r[6]}")
    tot+=r[1]; k=r[0].split('<')[0].replace('void ','').split('(')[0]; agg[k]=agg.get(k,0)+r[1]
print('total us', round(tot/1000,1), 'wall', round((rows[b-1][7]+rows[b-1][1]-rows[a][7])/1000,1), {k:round(v/1000,1) for k,v in agg.items()})
