#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/pv; rocprofv3 --kernel-trace --stats -d /tmp/pv -o v --output-format csv -- python scripts/dev/vae_only.py > gpurun_out/pv.log 2>&1 || tail -5 gpurun_out/pv.log
f=$(find /tmp/pv -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/vae_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/vae_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms per fwd+bwd', tot/1e6/10)
for r in rows[:28]:
    print(f"{r['Name'][:80].replace('(anonymous namespace)::',''):80s} {int(r['Calls'])/10:6.1f} {float(r['TotalDurationNs'])/1e7:8.1f} us/iter  {float(r['AverageNs'])/1e3:7.1f} us avg")
PY
