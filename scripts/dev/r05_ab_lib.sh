#!/bin/bash
# GPU box: same-device A/B of two builds of libdiffmusic_hip.so (both through the ctypes binding), interleaved twice.
# usage: r05_ab_lib.sh <tag> <libA> <libB> [bench args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; A=$2; B=$3; shift 3
out=gpurun_out/r05_ab_$tag.log; : > $out
for r in 1 2; do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    DMX_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-full-trajectory --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v ($lib) round $r:', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms'].items()})" >> $out
  done
done
cat $out
