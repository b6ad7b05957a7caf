import csv, collections, sys
d = sys.argv[1]
def table(name):
    rows = list(csv.DictReader(open(f'{d}/{name}/{name}_counter_collection.csv')))
    disp = collections.OrderedDict()
    for r in rows:
        e = disp.setdefault(r['Dispatch_Id'], dict(name=r['Kernel_Name'], grid=r['Grid_Size'], lds=r.get('LDS_Block_Size'), t0=int(r['Start_Timestamp']), t1=int(r['End_Timestamp'])))
        e[r['Counter_Name']] = float(r['Counter_Value'])
    return list(disp.values())
def lastfwd(t):
    idx=[i for i,e in enumerate(t) if 'bilinear_up2' in e['name']]
    return t[idx[-2]+1: idx[-1]+1]
A,B,C,F,W = [lastfwd(table(n)) for n in 'a b c fetch write'.split()]
filt = sys.argv[2] if len(sys.argv)>2 else ''
print(f"{'kernel':22s} {'us':>7s} {'lds':>6s} {'waves':>6s} {'valu':>8s} {'salu':>8s} {'smem':>7s} {'ldsI':>8s} {'mfma':>8s} | {'wait%':>6s} {'winst%':>6s} {'act%':>6s} {'vmrd':>8s} {'vmwr':>7s} | {'mfmabusy%':>9s} {'ldsconf%':>8s} {'lvlvmem':>8s} {'lvlwav':>7s} | {'rdMB':>7s} {'wrMB':>7s}")
for a,b,c,f,w in zip(A,B,C,F,W):
    if filt not in a['name']: continue
    wc=a['SQ_WAVE_CYCLES']; busy=a['SQ_BUSY_CYCLES']
    us=(a['t1']-a['t0'])/1000
    print(f"{a['name'].replace('void ','')[:22]:22s} {us:7.1f} {a['lds']:>6s} {a['SQ_WAVES']:6.0f} {a['SQ_INSTS_VALU']:8.0f} {a['SQ_INSTS_SALU']:8.0f} {a['SQ_INSTS_SMEM']:7.0f} {a['SQ_INSTS_LDS']:8.0f} {a['SQ_INSTS_MFMA']:8.0f} | {100*b['SQ_WAIT_ANY']/wc:6.1f} {100*b['SQ_WAIT_INST_ANY']/wc:6.1f} {100*b['SQ_ACTIVE_INST_ANY']/wc:6.1f} {b['SQ_INSTS_VMEM_RD']:8.0f} {b['SQ_INSTS_VMEM_WR']:7.0f} | {100*c['SQ_VALU_MFMA_BUSY_CYCLES']/max(1,c.get('SQ_BUSY_CYCLES',busy))/4:9.1f} {100*c['SQ_LDS_BANK_CONFLICT']/max(1,c['SQ_LDS_IDX_ACTIVE']):8.1f} {c['SQ_INST_LEVEL_VMEM']/max(1,wc)*1:8.3f} {c['SQ_LEVEL_WAVES']/max(1,busy):7.2f} | {f['FETCH_SIZE']*2/1024:7.1f} {w['WRITE_SIZE']/1024:7.1f}")
