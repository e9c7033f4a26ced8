#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_hifigan.py -x -q 2>&1 | tail -2
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export DMX_LIB_PATH=$GRAFT_REPO_ROOT/diffmusic_amd/lib/$lib; else unset DMX_LIB_PATH; fi
  echo "=== lib: ${lib:-default}"
  timeout -k 10 200 python scripts/dev/pair_bench.py 2>&1 | grep -v amdgpu.ids
done
