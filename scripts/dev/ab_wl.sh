#!/bin/bash
# A/B env settings on another workload: ab_wl.sh <workload> "ENV=1" "ENV2=1"
w=$1; shift
for round in 1 2; do
for v in "$@"; do
  echo -n "[$w $v] "; env $v python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
done
