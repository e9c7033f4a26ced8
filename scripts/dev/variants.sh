#!/bin/bash
for v in "-DDMX_ISSUE_MID" "-DDMX_ISSUE_MID -DDMX_NOEPI"; do
  echo "=== variant: [$v]"
  DMX_EXTRA_FLAGS="$v" python -m diffmusic_amd.build --force > gpurun_out/build.log 2>&1 || { echo build failed; tail -5 gpurun_out/build.log; continue; }
  python scripts/dev/gemm_bench.py 2>&1 | grep -E "TF/s"
done
python -m diffmusic_amd.build --force > /dev/null 2>&1
