#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/pu; rocprofv3 --kernel-trace --stats -d /tmp/pu -o u --output-format csv -- python scripts/dev/unet_only.py > gpurun_out/pu.log 2>&1
f=$(find /tmp/pu -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/unet_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/unet_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms per fwd', tot/1e6/10)
for r in rows[:25]:
    print(f"{r['Name'][:80].replace('(anonymous namespace)::',''):80s} {int(r['Calls'])/10:6.1f} {float(r['TotalDurationNs'])/1e7:8.1f} us/fwd  {float(r['AverageNs'])/1e3:7.1f} us avg")
PY
DMX_PROF_CSV=gpurun_out/unet_shapes.csv python - <<'PY'
import sys, torch, ctypes as C
sys.path.insert(0, '.')
from diffmusic_amd.engine import UNetEngine
from diffmusic_amd import _lib as L
eng = UNetEngine(); eng.load_state_dict(eng.synth_state_dict(0))
B = 16
x = torch.randn(B, 8, 250, 16, device="cuda"); t = torch.full((B,), 501.0, device="cuda"); cls = torch.randn(B, 512, device="cuda")
for _ in range(3): out = eng.forward(x, t, cls)
torch.cuda.synchronize()
L.lib().dmx_prof_begin()
out = eng.forward(x, t, cls)
ms, fl = C.c_double(), C.c_double()
n = L.lib().dmx_prof_end(C.byref(ms), C.byref(fl))
print("gemm launches", n, "ms", ms.value, "TF", fl.value/1e12)
PY
python scripts/dev/shape_summary.py gpurun_out/unet_shapes.csv 40
