#!/bin/bash
for v in "-DDMX_SWP -DDMX_NOISSUE -DDMX_NOBARRIER" "-DDMX_SWP -DDMX_NOISSUE" "-DDMX_SWP -DDMX_NOEPI" "-DDMX_SWP"; do
  echo "=== variant: [$v]"
  DMX_EXTRA_FLAGS="$v" python -m diffmusic_amd.build --force > gpurun_out/build.log 2>&1 || { echo build failed; grep -m3 error -A5 gpurun_out/build.log; continue; }
  python scripts/dev/gemm_one.py vae 2>&1 | grep TF/s
  python scripts/dev/gemm_one.py s2 2>&1 | grep TF/s
done
python -m diffmusic_amd.build --force > /dev/null 2>&1
