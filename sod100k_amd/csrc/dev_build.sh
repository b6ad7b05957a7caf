#!/bin/bash
# Developer build: libcsnet_hip.so for gfx950 with per-kernel resource usage and ISA kept in $OUT (default /tmp/csn_build).
# The product build recipe is sod100k_amd/_native.py:build() (same flags without the diagnostics).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${OUT:-/tmp/csn_build}"
mkdir -p "$OUT"
cd "$OUT"
# the build id the loader checks (csn_build_sources_sha16, include/csnet_hip.h): the unit sod100k_amd/_native.py generates
python3 -c "import sys; sys.path.insert(0, '$HERE/../..'); from sod100k_amd import _native as N; N.write_build_id_object(N.hipcc_path(), N.sources_sha16(), '$OUT')"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage -save-temps -I"$HERE" \
  -o "$HERE/libcsnet_hip.so" "$HERE/csn_plan.hip" "$HERE/k_misc.hip" "$HERE/k_goct_pw.hip" "$HERE/k_ms.hip" "$HERE/k_train.hip" "$HERE/k_wgrad.hip" "$HERE/k_goct_c3.hip" "$HERE/k_csf.hip" "$HERE/k_wgrad_c3.hip" "$HERE/k_wgrad_bf.hip" "$HERE/k_pw4.hip" "$HERE/k_c3q.hip" "$HERE/k_pwq.hip" "$HERE/k_ilb.hip" "$HERE/k_head.hip" "$OUT/csn_build_id.o" \
  > "$OUT/build.log" 2>&1 || { cat "$OUT/build.log" | grep -E "error" -A3 | head -40; exit 1; }
python3 - "$OUT/build.log" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
rows = []
cur = {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = m.group(1)
print(f"{'kernel':60s} {'sgpr':>5s} {'vgpr':>5s} {'scr':>5s} {'occ':>4s}")
for r in rows:
    print(f"{r['name'][:60]:60s} {r.get('sgpr','?'):>5s} {r.get('vgpr','?'):>5s} {r.get('scratch','?'):>5s} {r.get('occ','?'):>4s}")
PY
