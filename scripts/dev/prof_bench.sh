#!/bin/bash
# committed evidence: rocprofv3 kernel stats of the default bench command + the bench JSON line + per-launch GEMM shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-v4}
mkdir -p gpurun_out/prof
python bench.py > gpurun_out/prof/r01_bench_${tag}.json.log 2> gpurun_out/prof/bench_err.log || tail -5 gpurun_out/prof/bench_err.log
tail -1 gpurun_out/prof/r01_bench_${tag}.json.log | cut -c1-400
rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof/prof_run.log 2>&1 || tail -5 gpurun_out/prof/prof_run.log
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) gpurun_out/prof/r01_bench_${tag}_kernel_stats.csv
DMX_PROF_CSV=gpurun_out/prof/r01_gemm_shapes_${tag}.csv python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python scripts/dev/shape_summary.py gpurun_out/prof/r01_gemm_shapes_${tag}.csv 12
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof/r01_bench_${tag}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel ms per step (7 steps incl. warm-up/profiled)', tot/1e6/7)
for r in rows[:16]:
    print(f"{r['Name'][:70].replace('(anonymous namespace)::',''):70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}%")
PY
