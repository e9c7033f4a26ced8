#!/bin/bash
# per-dispatch trace of ONE VAE decode + backward (B = 8): GroupNorm / elementwise kernels with their achieved GB/s
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/vt; rocprofv3 --kernel-trace -d /tmp/vt -o v --output-format csv -- python scripts/dev/vae_only.py > gpurun_out/vt.log 2>&1
f=$(find /tmp/vt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows); sel = rows[int(n * 0.8):]          # the last two iterations
agg = collections.OrderedDict()
for r in sel:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
    if name.startswith('gemm_') or name.startswith('conv_'): continue
    grid = f"{r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}/{r.get('Workgroup_Size_X','')}"
    a = agg.setdefault((name, grid), [0, 0.0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
with open('gpurun_out/vae_trace.txt', 'w') as out:
    for (name, grid), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        line = f"{name:44s} {grid:22s} n={c:3d} tot={t:8.1f} us avg={t/c:7.1f} us"
        out.write(line + "\n"); print(line)
PY
