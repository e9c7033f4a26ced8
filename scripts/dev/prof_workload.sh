#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
w=${1:-dsg_phase_audioldm2}
rm -rf /tmp/pw; rocprofv3 --kernel-trace --stats -d /tmp/pw -o w --output-format csv -- python bench.py --workload $w --steps 4 --warmup 1 --settle 0 --no-cpu-baseline --no-stage-times > gpurun_out/pw.log 2>&1 || tail -5 gpurun_out/pw.log
tail -1 gpurun_out/pw.log | cut -c1-200
cp $(find /tmp/pw -name "*kernel_stats.csv" | head -1) gpurun_out/${w}_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${w}_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel ms per step (6 steps)', tot/1e6/6)
for r in rows[:36]:
    print(f"{r['Name'][:75].replace('(anonymous namespace)::',''):75s} {int(r['Calls'])/6:7.1f} {float(r['TotalDurationNs'])/6e6:7.2f} ms/step {float(r['AverageNs'])/1e3:7.1f} us")
PY
