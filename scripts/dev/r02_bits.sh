#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r02_${1:-f}; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_hifigan.py tests/test_gpu_step.py tests/test_gpu_fullsize.py tests/test_gpu_torch_ops.py tests/test_gpu_parity_rows.py "tests/test_gpu_fullsize_parity.py::test_fullsize_teacher_forced_step[dps_inpainting]" -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
for r in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench_$r.json 2> $out/bench.err || tail -3 $out/bench.err
python -c "
import json; d=json.load(open('$out/bench_$r.json')); print('bench', d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['achieved'], d['roofline']['all_gemm_kernels'])"
done
