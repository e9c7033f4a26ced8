#!/bin/bash
# GPU box: clip lanes with the U-Net stream confined to a reserved set of CUs and the sweep stream to the rest (CU masks), vs plain loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05_lanes_mask.log; : > $out
common="--no-cpu-baseline --no-stage-times --no-full-trajectory --steps 20 --warmup 3"
run() {
  echo "== $1" >> $out
  env $2 timeout -k 10 200 python bench.py $common $3 2>>gpurun_out/r05_lanes_mask.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'lanes': d['config']['lanes'], 'finite': d['config']['finite']}))" >> $out
}
run "plain" "A=1" "--lanes 1"
run "lanes 2, priority, no mask" "A=1" "--lanes 2"
run "lanes 2, U-Net on mask 0x01010101 (32 CUs), sweep on the rest" "DMX_LANE_CU_MASK=01010101" "--lanes 2"
run "lanes 2, U-Net on mask 0x11111111 (64 CUs)" "DMX_LANE_CU_MASK=11111111" "--lanes 2"
run "lanes 2, U-Net on mask 0x00010001 (16 CUs)" "DMX_LANE_CU_MASK=00010001" "--lanes 2"
run "plain (again)" "A=1" "--lanes 1"
cat $out
