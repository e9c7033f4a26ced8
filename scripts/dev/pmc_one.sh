#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in WRITE_SIZE; do
rm -rf /tmp/pp; timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c -d /tmp/pp -o p --output-format csv -- python scripts/dev/pmc_calib.py > /tmp/pp.log 2>&1 || tail -3 /tmp/pp.log
python - <<'PY'
import csv, glob, sys
seen=set()
for f in glob.glob('/tmp/pp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_' in r['Kernel_Name'] or 'copyBuffer' in r['Kernel_Name']:
            k=(r['Kernel_Name'][:60], r['Counter_Name'], float(r['Counter_Value']))
            if k not in seen: print(*k); seen.add(k)
PY
done
