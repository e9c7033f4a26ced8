#!/bin/bash
# bench several environment settings on the same box, two rounds
for round in 1 2; do
for v in "$@"; do
  echo -n "[$v] "; env $v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
done
