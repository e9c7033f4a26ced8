#!/bin/bash
# Round-5 evidence, all from one source tree on one GPU box: bench JSON lines (headline, the other GPU workloads, the strong-scaling
# single-GPU references), rocprofv3 kernel stats, per-launch GEMM shapes, U-Net-only profile, and SEPARATE --pmc passes
# (FETCH_SIZE, WRITE_SIZE, MFMA-pipe utilisation).  Everything lands in gpurun_out/evidence/ and is copied to profiles/r05_* afterwards.
# usage: r05_evidence.sh [tag] [part]   part = all | bench | prof | pmc
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-r05}; part=${2:-all}; out=gpurun_out/evidence; mkdir -p $out
sha=$(cat $(ls diffmusic_amd/csrc/*.hip diffmusic_amd/csrc/*.h diffmusic_amd/csrc/*.inc bench.py | sort) | sha256sum | cut -c1-16)
echo "{\"source_sha16\": \"$sha\", \"note\": \"sha256 of diffmusic_amd/csrc/*.hip, *.h, *.inc (tile table) and bench.py concatenated in sorted order\"}" > $out/${tag}_pmc_meta.json
if [ $part = all ] || [ $part = bench ]; then
  timeout -k 10 500 python bench.py > $out/${tag}_bench.json 2> >(tee $out/bench.err >&2); echo "bench rc=$?"; cut -c1-200 $out/${tag}_bench.json
  for wl in dsg_phase_audioldm2 mpgd_sr4 diffmusic_style_audioldm2; do
    timeout -k 10 300 python bench.py --workload $wl --steps 6 --warmup 2 > $out/${tag}_bench_$wl.json 2>> $out/bench.err; echo "$wl rc=$?"
  done
  # strong-scaling references on ONE GPU: the whole global batch of configs[2] (32 clips) and configs[3] (16 clips)
  timeout -k 10 300 python bench.py --workload dsg_phase_audioldm2 --global-batch 32 --steps 4 --warmup 1 > $out/${tag}_bench_strong_dsg_g32_n1.json 2>> $out/bench.err; echo "strong dsg rc=$?"
  timeout -k 10 300 python bench.py --workload mpgd_sr4 --global-batch 16 --steps 6 --warmup 2 > $out/${tag}_bench_strong_mpgd_g16_n1.json 2>> $out/bench.err; echo "strong mpgd rc=$?"
  python - <<PY
import json
o = "$out/${tag}"
def ms(p): return json.load(open(p))["ms_per_step"]
pred = {}
try:
    pred["configs[2] dsg_phase_audioldm2: 32 clips on 1 GPU / 4 clips on 1 GPU (= predicted 8-GPU speed-up before the gather)"] = round(ms(o + "_bench_strong_dsg_g32_n1.json") / ms(o + "_bench_dsg_phase_audioldm2.json"), 3)
    pred["configs[3] mpgd_sr4: 16 clips on 1 GPU / 4 clips on 1 GPU (= predicted 4-GPU speed-up)"] = round(ms(o + "_bench_strong_mpgd_g16_n1.json") / ms(o + "_bench_mpgd_sr4.json"), 3)
except Exception as e:
    pred["error"] = repr(e)
json.dump(pred, open(o + "_strong_scaling_prediction.json", "w"), indent=1); print(pred)
PY
  # 2-rank rehearsal of the strong-scaling launch on this one GPU (gloo, both ranks on cuda:0): exercises spawn, sharding, gather
  timeout -k 10 300 python bench.py --gpus 2 --backend gloo --share-gpu --global-batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-stage-times > $out/${tag}_bench_rehearsal_strong2.json 2> $out/rehearsal.err; echo "rehearsal rc=$?"; tail -2 $out/rehearsal.err
fi
if [ $part = all ] || [ $part = prof ]; then
  rm -rf /tmp/pb; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stage-times > $out/prof_run.log 2>&1 || tail -5 $out/prof_run.log
  cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
  DMX_PROF_CSV=$out/${tag}_gemm_shapes.csv timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-times > /dev/null 2>&1
  for kind in musicldm audioldm2; do
    python scripts/dev/unet_only.py $kind > $out/${tag}_unet_time_$kind.log 2>&1; cat $out/${tag}_unet_time_$kind.log
    rm -rf /tmp/pu; timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/pu -o u --output-format csv -- python scripts/dev/unet_only.py $kind > $out/unet_run.log 2>&1 || tail -3 $out/unet_run.log
    sfx=""; [ $kind = audioldm2 ] && sfx="_audioldm2"
    cp $(find /tmp/pu -name "*kernel_stats.csv" | head -1) $out/${tag}_unet_only${sfx}_kernel_stats.csv
    python scripts/dev/trace_summary.py $(find /tmp/pu -name "*kernel_trace.csv" | head -1) 13 > $out/${tag}_unet_forward_sequence_$kind.txt; head -1 $out/${tag}_unet_forward_sequence_$kind.txt
  done
  for wl in dsg_phase_audioldm2 mpgd_sr4 diffmusic_style_audioldm2; do
    rm -rf /tmp/pw; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pw -o w --output-format csv -- python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-stage-times > $out/prof_$wl.log 2>&1 || tail -3 $out/prof_$wl.log
    cp $(find /tmp/pw -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_${wl}_kernel_stats.csv
  done
fi
if [ $part = all ] || [ $part = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-times > $out/pmc_$c.log 2>&1 || tail -3 $out/pmc_$c.log
    python scripts/dev/pmc_summary.py /tmp/pmc_$c $c $out/${tag}_pmc_${c}_per_kernel.csv | head -8
  done
  rm -rf /tmp/pmc_mfma
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d /tmp/pmc_mfma -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-times > $out/pmc_mfma.log 2>&1 || tail -3 $out/pmc_mfma.log
  python scripts/dev/pmc_mfma_summary.py /tmp/pmc_mfma $out/${tag}_pmc_mfma_util_per_kernel.csv | head -12
fi
ls -la $out
