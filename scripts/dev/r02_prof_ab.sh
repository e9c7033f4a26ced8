#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r02_${1:-j}; mkdir -p $out
for lib in "" ${@:2}; do
  if [ -n "$lib" ]; then export DMX_LIB_PATH=$GRAFT_REPO_ROOT/diffmusic_amd/lib/$lib; else unset DMX_LIB_PATH; fi
  rm -rf /tmp/pb; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stage-times > $out/prof_run.log 2>&1 || tail -5 $out/prof_run.log
  cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $out/kernel_stats_${lib:-default}.csv
done
