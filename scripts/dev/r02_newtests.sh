#!/bin/bash
# GPU box: only the tests added since the last full run + the style workload bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r02_${1:-b}; mkdir -p $out
timeout -k 10 800 python -m pytest tests/test_gpu_style.py tests/test_gpu_torch_ops.py tests/test_gpu_parity_rows.py "tests/test_gpu_fullsize_parity.py::test_fullsize_teacher_forced_step[diffmusic_style_audioldm2]" tests/test_gpu_step.py tests/test_gpu_pipeline.py -x -q -s > $out/pytest_new.log 2>&1; echo "pytest rc=$?"
grep -n "err\|rel\|passed\|failed\|Error" $out/pytest_new.log | tail -30
timeout -k 10 300 python bench.py --workload diffmusic_style_audioldm2 --steps 5 --warmup 2 > $out/bench_style.json 2> $out/bench_style.err; echo "bench style rc=$?"; tail -3 $out/bench_style.err; cut -c1-400 $out/bench_style.json
