#!/bin/bash
# GPU box: A/B of environment switches on the same device: default bench with stage times, twice interleaved
# usage: r03_abenv.sh <tag> "VAR=val" "VAR=val VAR2=val" ...   (first variant: no extra env)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03_${1:-env}; mkdir -p $out
shift
for r in 1 2; do
  i=0
  for v in "" "$@"; do
    echo "=== env: ${v:-default} (round $r)"
    env $v timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_${i}_$r.json 2> $out/bench.err || tail -3 $out/bench.err
    python -c "
import json; d=json.load(open('$out/bench_${i}_$r.json')); print('bench', d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms'].items()})"
    i=$((i+1))
  done
done
