#!/bin/bash
# GPU box: A/B of library variants (DMX_LIB_PATH: both arms through the ctypes binding) on the same device: default bench, interleaved twice
# usage: r03_ablib.sh <tag> libA.so libB.so ...   (files under diffmusic_amd/lib/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03_${1:-ablib}; mkdir -p $out
for r in 1 2; do
for lib in ${@:2}; do
  export DMX_LIB_PATH=$GRAFT_REPO_ROOT/diffmusic_amd/lib/$lib
  echo "=== lib: $lib (round $r)"
  timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_${lib}_$r.json 2> $out/bench.err || tail -3 $out/bench.err
  python -c "
import json; d=json.load(open('$out/bench_${lib}_$r.json')); print('bench', d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms'].items()}, d['roofline']['achieved'])"
done
done
