#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export DMX_LIB_PATH=$GRAFT_REPO_ROOT/diffmusic_amd/lib/$lib; else unset DMX_LIB_PATH; fi
  echo "=== lib: ${lib:-default}"
  timeout -k 10 200 python scripts/dev/gemm_bench.py 2>&1 | grep -v amdgpu.ids | grep "k11\|k3 \|N128\|plain"
done
