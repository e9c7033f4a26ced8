#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_groupnorm.py tests/test_gpu_vae.py tests/test_gpu_gemm.py tests/test_gpu_multirank.py -x -q > gpurun_out/r05_call2_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r05_call2_tests.log; echo "tests rc=$rc"
[ $rc = 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_batch_parity.py -x -q -k "bench_batch" > gpurun_out/r05_call2_batch.log 2>&1; rc=$?; tail -5 gpurun_out/r05_call2_batch.log; echo "batch parity rc=$rc"
cp gpurun_out/batch_parity_*.json gpurun_out/ 2>/dev/null
rm -rf /tmp/pb; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --steps 8 --warmup 1 --settle 3 --no-cpu-baseline --no-stage-times --no-full-trajectory > gpurun_out/r05_prof_plain.log 2>&1 || tail -5 gpurun_out/r05_prof_plain.log
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) gpurun_out/r05a_bench_kernel_stats.csv
rm -rf /tmp/pl; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pl -o l --output-format csv -- python bench.py --lanes 2 --steps 8 --warmup 1 --settle 3 --no-cpu-baseline --no-stage-times --no-full-trajectory > gpurun_out/r05_prof_lanes.log 2>&1 || tail -5 gpurun_out/r05_prof_lanes.log
cp $(find /tmp/pl -name "*kernel_stats.csv" | head -1) gpurun_out/r05_lanes2_kernel_stats.csv
python scripts/dev/r05_lanes_trace.py $(find /tmp/pl -name "*kernel_trace.csv" | head -1) $(find /tmp/pb -name "*kernel_trace.csv" | head -1) > gpurun_out/r05_lanes_trace_summary.txt 2>&1; cat gpurun_out/r05_lanes_trace_summary.txt
head -1 $(find /tmp/pl -name "*kernel_trace.csv" | head -1)
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-full-trajectory --steps 20 --warmup 3 > gpurun_out/r05a_bench_$i.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05a_bench_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms'])"; done
