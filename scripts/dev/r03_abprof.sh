#!/bin/bash
# GPU box: per-kernel A/B of two library builds (DMX_LIB_PATH): rocprofv3 kernel stats of the same bench run, joined per kernel name
# usage: r03_abprof.sh <tag> libA.so libB.so
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03_${1:-abprof}; mkdir -p $out
for lib in $2 $3; do
  export DMX_LIB_PATH=$GRAFT_REPO_ROOT/diffmusic_amd/lib/$lib
  rm -rf /tmp/pb; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stage-times > $out/prof_$lib.log 2>&1 || tail -5 $out/prof_$lib.log
  cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $out/stats_$lib.csv
done
python - $out/stats_$2.csv $out/stats_$3.csv <<'PY'
import csv, sys
def rd(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
a, b = rd(sys.argv[1]), rd(sys.argv[2])
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort()
steps = 6.0
print("total A %.3f ms/step   B %.3f ms/step" % (sum(v[1] for v in a.values()) / steps / 1e6, sum(v[1] for v in b.values()) / steps / 1e6))
for d, k, ca, ta, cb, tb in rows[:14] + rows[-14:]:
    print("%+8.3f ms/step  %6d %9.3f -> %6d %9.3f   %s" % (d / steps / 1e6, ca, ta / steps / 1e6, cb, tb / steps / 1e6, k[:150]))
PY
