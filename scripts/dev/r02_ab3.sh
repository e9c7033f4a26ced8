#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r02_${1:-k}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_hifigan.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
for lib in "" ${@:2} ""; do
  if [ -n "$lib" ]; then export DMX_LIB_PATH=$GRAFT_REPO_ROOT/diffmusic_amd/lib/$lib; else unset DMX_LIB_PATH; fi
  echo "=== lib: ${lib:-default}"
  timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_${lib:-default}.json 2> $out/bench.err || tail -3 $out/bench.err
  python -c "
import json; d=json.load(open('$out/bench_${lib:-default}.json')); print('bench', d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'], d['roofline']['all_gemm_kernels']['achieved'])"
done
for lib in "" ${@:2}; do
  if [ -n "$lib" ]; then export DMX_LIB_PATH=$GRAFT_REPO_ROOT/diffmusic_amd/lib/$lib; else unset DMX_LIB_PATH; fi
  rm -rf /tmp/pb; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stage-times > $out/prof_run.log 2>&1 || tail -5 $out/prof_run.log
  cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $out/kernel_stats_${lib:-default}.csv
done
