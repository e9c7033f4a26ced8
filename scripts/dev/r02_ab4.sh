#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r02_${1:-l}; mkdir -p $out
for lib in "" ${@:2} ""; do
  if [ -n "$lib" ]; then export DMX_LIB_PATH=$GRAFT_REPO_ROOT/diffmusic_amd/lib/$lib; else unset DMX_LIB_PATH; fi
  echo "=== lib: ${lib:-default}"
  timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_${lib:-default}.json 2> $out/bench.err || tail -3 $out/bench.err
  python -c "
import json; d=json.load(open('$out/bench_${lib:-default}.json')); print('bench', d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'], d['roofline']['all_gemm_kernels']['achieved'])"
done
