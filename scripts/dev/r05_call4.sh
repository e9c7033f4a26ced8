#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_htsat.py -x -q -s > gpurun_out/r05_htsat_tests.log 2>&1; rc=$?; grep -E "stage|final|rel-L2|passed|failed|Error|error|assert" gpurun_out/r05_htsat_tests.log | tail -60; echo "htsat tests rc=$rc"
timeout -k 10 600 python -m pytest tests/test_gpu_style.py tests/test_gpu_pipeline.py -x -q -s > gpurun_out/r05_style_tests.log 2>&1; rc=$?; grep -E "rel|passed|failed|Error|assert" gpurun_out/r05_style_tests.log | tail -20; echo "style tests rc=$rc"
