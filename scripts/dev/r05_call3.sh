#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_call3_tests.log 2>&1; rc=$?; tail -8 gpurun_out/r05_call3_tests.log; echo "tests rc=$rc"
