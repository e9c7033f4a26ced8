#!/bin/bash
# GPU box: A/B of library variants (DMX_LIB_PATH) on the same device: GEMM micro-bench + default bench, interleaved twice
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r02_${1:-ab}; mkdir -p $out
for r in 1 2; do
for lib in "" ${@:2}; do
  if [ -n "$lib" ]; then export DMX_LIB_PATH=$GRAFT_REPO_ROOT/diffmusic_amd/lib/$lib; else unset DMX_LIB_PATH; fi
  echo "=== lib: ${lib:-default} (round $r)"
  if [ $r = 1 ]; then timeout -k 10 200 python scripts/dev/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/gemm_bench_${lib:-default}.log; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-stage-times > $out/bench_${lib:-default}_$r.json 2> $out/bench.err || tail -3 $out/bench.err
  python -c "
import json; d=json.load(open('$out/bench_${lib:-default}_$r.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['all_gemm_kernels']['achieved'])"
done
done
