#!/bin/bash
for v in "$@"; do
  DMX_EXTRA_FLAGS="$v" python -m diffmusic_amd.build --force > gpurun_out/build.log 2>&1 || { echo build failed; grep -m3 error -A5 gpurun_out/build.log; continue; }
  echo "== [$v]"; timeout -k 10 200 python scripts/dev/pair_bench.py
done
