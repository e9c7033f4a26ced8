#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --workload diffmusic_style_audioldm2 --steps 6 --warmup 2 > gpurun_out/r05_bench_diffmusic_style_audioldm2.json 2> gpurun_out/r05_style_bench.err; echo "rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r05_bench_diffmusic_style_audioldm2.json')); print(d['value'], d['ms_per_step'], d['stage_ms'])"
rm -rf /tmp/pw; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pw -o w --output-format csv -- python bench.py --workload diffmusic_style_audioldm2 --steps 4 --warmup 1 --no-cpu-baseline --no-stage-times --no-full-trajectory > gpurun_out/prof_style.log 2>&1 || tail -3 gpurun_out/prof_style.log
cp $(find /tmp/pw -name "*kernel_stats.csv" | head -1) gpurun_out/r05_bench_diffmusic_style_audioldm2_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05_bench_diffmusic_style_audioldm2_kernel_stats.csv')))
keys=("win_attn","ln_rows","gelu_","embed_","interp_bwd","gram_","absmax","l2_loss")
tot=0
for r in rows:
    if any(k in r['Name'] for k in keys):
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms"); tot+=float(r['TotalDurationNs'])
print("tower-specific kernels total ms over the run:", tot/1e6)
PY
