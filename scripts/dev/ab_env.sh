#!/bin/bash
# A/B two environment settings on the SAME box and build: usage ab_env.sh "VAR=1" "VAR2=1"  ("" = default)
for round in 1 2; do
for v in "$1" "$2"; do
  echo -n "[$v] "; env $v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['all_gemm_kernels'])"
done
done
