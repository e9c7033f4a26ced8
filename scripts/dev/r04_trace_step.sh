#!/bin/bash
# GPU box: rocprof kernel stats of the default bench (no CPU leg): per-kernel totals -> gpurun_out/<tag>_bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-r04}
rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --no-cpu-baseline --no-full-trajectory > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_bench_prof.err
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${tag}_bench_kernel_stats.csv')))
n=[int(r['Calls']) for r in rows if 'stft_mel_bwd' in r['Name']][0]
print('steps profiled', n)
for r in rows[:45]:
    nm=r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:70]
    print(f"{nm:70s} {int(r['Calls'])/n:7.1f}/step {float(r['TotalDurationNs'])/1e6/n:7.3f} ms/step {float(r['AverageNs'])/1e3:8.1f} us")
PY
