#!/bin/bash
# A/B two builds on the SAME box: usage ab.sh "<flagsA>" "<flagsB>"
for round in 1 2; do
for v in "$1" "$2"; do
  DMX_EXTRA_FLAGS="$v" python -m diffmusic_amd.build --force > gpurun_out/build.log 2>&1 || { echo build failed; grep -m3 error -A5 gpurun_out/build.log; continue; }
  echo -n "[$v] "; python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done
done
python -m diffmusic_amd.build --force > /dev/null 2>&1
