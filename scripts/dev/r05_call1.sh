#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_lanes.py -x -q > gpurun_out/r05_lanes_tests.log 2>&1; echo "lanes tests rc=$?" | tee -a gpurun_out/r05_lanes_tests.log
tail -15 gpurun_out/r05_lanes_tests.log
timeout -k 10 300 python bench.py --gpus 1 --force-dist --backend nccl --steps 3 --warmup 1 --settle 2 --no-cpu-baseline --no-stage-times --no-full-trajectory > gpurun_out/r05_rccl_world1.json 2> gpurun_out/r05_rccl_world1.err; echo "rccl world-1 rc=$?"
tail -3 gpurun_out/r05_rccl_world1.err; cat gpurun_out/r05_rccl_world1.json | cut -c1-1500
timeout -k 10 900 bash scripts/dev/r05_lanes_ab.sh
