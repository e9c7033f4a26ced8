#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_htsat.py tests/test_gpu_style.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r05_call6_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r05_call6_tests.log; echo "tests rc=$rc"
timeout -k 10 400 python bench.py --workload diffmusic_style_audioldm2 --steps 6 --warmup 2 > gpurun_out/r05_bench_diffmusic_style_audioldm2.json 2> gpurun_out/r05_style_bench.err; echo "rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r05_bench_diffmusic_style_audioldm2.json')); print(d['value'], d['ms_per_step'], d['stage_ms'])"
echo "== U-Net GroupNorm statistics from the producers (image-aligned slots): off / on, interleaved" > gpurun_out/r05_unet_gn_parts.log
for i in 1 2; do
  for kind in musicldm audioldm2; do
    python scripts/dev/unet_only.py $kind 2>&1 | tail -1 | sed 's/^/off: /' >> gpurun_out/r05_unet_gn_parts.log
    DMX_UNET_GN_PARTS=1 python scripts/dev/unet_only.py $kind 2>&1 | tail -1 | sed 's/^/on:  /' >> gpurun_out/r05_unet_gn_parts.log
  done
done
cat gpurun_out/r05_unet_gn_parts.log
