#!/bin/bash
# per-dispatch trace of ONE U-Net forward (B = 16): name, grid, duration, gap to the previous kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/ut; rocprofv3 --kernel-trace -d /tmp/ut -o u --output-format csv -- python scripts/dev/unet_only.py > gpurun_out/ut.log 2>&1
f=$(find /tmp/ut -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last forward: find the last timestep_embed_kernel
idx = [i for i, r in enumerate(rows) if 'timestep_embed' in r['Kernel_Name']]
lo = idx[-1]
sel = rows[lo:]
out = open('gpurun_out/unet_trace.txt', 'w')
prev_end = None; tot = 0; gaps = 0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) if prev_end else 0
    prev_end = e
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    grid = f"{r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}x{r.get('Grid_Size_Z','')}/{r.get('Workgroup_Size_X','')}"
    out.write(f"{name:60s} {grid:22s} {(e-s)/1e3:8.1f} us  gap {gap/1e3:6.1f}\n")
    tot += e - s; gaps += gap
out.write(f"kernels {len(sel)} busy {tot/1e6:.3f} ms gaps {gaps/1e6:.3f} ms span {(int(sel[-1]['End_Timestamp'])-int(sel[0]['Start_Timestamp']))/1e6:.3f} ms\n")
print(f"kernels {len(sel)} busy {tot/1e6:.3f} ms gaps {gaps/1e6:.3f} ms")
PY
