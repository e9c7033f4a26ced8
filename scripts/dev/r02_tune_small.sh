#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp

timeout -k 10 900 python scripts/dev/tune_small.py profiles/r02_gemm_shapes.csv > gpurun_out/tune_small.log 2>&1; echo "tune rc=$?"; tail -4 gpurun_out/tune_small.log
