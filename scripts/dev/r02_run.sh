#!/bin/bash
# one GPU-box call: GPU tests, default bench (JSON line), 2-rank rehearsal of the multi-GPU path, rocprofv3 kernel stats
# usage: scripts/dev/r02_run.sh <tag> [tests|notests]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-a}; mode=${2:-tests}
out=gpurun_out/r02_$tag; mkdir -p $out
if [ "$mode" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
  tail -5 $out/pytest.log
fi
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err; cut -c1-600 $out/bench.json
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --share-gpu --steps 3 --warmup 1 --batch 4 --no-cpu-baseline --no-stage-times > $out/bench_rehearsal2.json 2> $out/bench_rehearsal2.err; echo "rehearsal rc=$?"; tail -2 $out/bench_rehearsal2.err; cut -c1-300 $out/bench_rehearsal2.json
rm -rf /tmp/pb; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stage-times > $out/prof_run.log 2>&1 || tail -5 $out/prof_run.log
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $out/r02_bench_${tag}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$out/r02_bench_${tag}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel ms per step (17 steps: 10 conditioning + warm-up + timed + profiled, + final decode)', tot/1e6/17)
for r in rows[:24]:
    print(f"{r['Name'][:70].replace('(anonymous namespace)::',''):70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/17:8.3f} ms/step {r['Percentage']}%")
PY
