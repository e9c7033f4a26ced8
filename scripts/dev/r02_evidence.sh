#!/bin/bash
# Round-2 evidence, all from one source tree on one GPU box: bench JSON lines (headline + the other GPU workloads), rocprofv3 kernel
# stats, per-launch GEMM shapes, roctx marker trace, U-Net-only profile, and two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE).
# Everything lands in gpurun_out/evidence/ and is copied to profiles/r02_* by hand afterwards.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-r02}; out=gpurun_out/evidence; mkdir -p $out
sha=$(cat $(ls diffmusic_amd/csrc/*.hip diffmusic_amd/csrc/*.h bench.py | sort) | sha256sum | cut -c1-16)
echo "{\"source_sha16\": \"$sha\", \"note\": \"sha256 of diffmusic_amd/csrc/*.hip, *.h and bench.py concatenated in sorted order\"}" > $out/${tag}_pmc_meta.json
timeout -k 10 400 python bench.py > $out/${tag}_bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/${tag}_bench.json
for wl in dsg_phase_audioldm2 mpgd_sr4 diffmusic_style_audioldm2; do
  timeout -k 10 300 python bench.py --workload $wl --steps 6 --warmup 2 > $out/${tag}_bench_$wl.json 2>> $out/bench.err; echo "$wl rc=$?"
done
rm -rf /tmp/pb; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stage-times > $out/prof_run.log 2>&1 || tail -5 $out/prof_run.log
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
DMX_PROF_CSV=$out/${tag}_gemm_shapes.csv timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-times > /dev/null 2>&1
rm -rf /tmp/pm; DMX_ROCTX=1 timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats -d /tmp/pm -o m --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stage-times > $out/marker_run.log 2>&1 || tail -5 $out/marker_run.log
for f in $(find /tmp/pm -name "*marker*stats*.csv" -o -name "*marker_api_stats.csv" | head -3); do cp $f $out/${tag}_roctx_$(basename $f); done
ls /tmp/pm/* | head
rm -rf /tmp/pu; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pu -o u --output-format csv -- python scripts/dev/unet_only.py > $out/unet_run.log 2>&1 || tail -3 $out/unet_run.log
cp $(find /tmp/pu -name "*kernel_stats.csv" | head -1) $out/${tag}_unet_only_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-times > $out/pmc_$c.log 2>&1 || tail -3 $out/pmc_$c.log
  python scripts/dev/pmc_summary.py /tmp/pmc_$c $c $out/${tag}_pmc_${c}_per_kernel.csv | head -8
done
# MFMA-pipe utilisation per kernel (north_star: "MFMA utilisation on the attention / conv path"): cycles the matrix pipe of a SIMD is
# busy / cycles its CU is busy, one more separate --pmc pass
rm -rf /tmp/pmc_mfma
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d /tmp/pmc_mfma -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-times > $out/pmc_mfma.log 2>&1 || tail -3 $out/pmc_mfma.log
python scripts/dev/pmc_mfma_summary.py /tmp/pmc_mfma $out/${tag}_pmc_mfma_util_per_kernel.csv | head -12
ls -la $out
