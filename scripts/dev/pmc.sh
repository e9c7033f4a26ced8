#!/bin/bash
# two separate counter passes over the default bench (1 warm-up + 2 timed + 1 profiled step = 4 steps)
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp DMX_SINGLE_STREAM=1
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/bench_$c.log 2>&1
  python scripts/dev/pmc_summary.py /tmp/pmc_$c $c gpurun_out/pmc/r01_pmc_${c}_per_kernel.csv
done
