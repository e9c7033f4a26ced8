#!/bin/bash
# GPU box: GEMM correctness tests, micro-bench on the hot shapes, the default bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r02_${1:-c}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_hifigan.py tests/test_gpu_vae.py tests/test_gpu_unet.py tests/test_gpu_attention.py tests/test_gpu_fullsize.py "tests/test_gpu_fullsize_parity.py::test_fullsize_teacher_forced_step[dps_inpainting]" -x -q > $out/pytest_gemm.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gemm.log
timeout -k 10 200 python scripts/dev/gemm_bench.py > $out/gemm_bench.log 2>&1; cat $out/gemm_bench.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -2 $out/bench.err
python - <<PY
import json
d=json.load(open('$out/bench.json'))
print(d['value'], d['ms_per_step'], d['stage_ms'])
print(d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['all_gemm_kernels'])
PY
