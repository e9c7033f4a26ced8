#!/bin/bash
# GPU box: clip lanes A/B (same device, interleaved twice): plain loop vs 2 / 3 lanes, with and without U-Net stream priority.
set -o pipefail
out=gpurun_out/r05_lanes_ab.log
mkdir -p gpurun_out
: > $out
common="--no-cpu-baseline --no-stage-times --no-full-trajectory --steps 20 --warmup 3"
run() {  # label, env, args
  echo "== $1" >> $out
  env $2 python bench.py $common $3 2>>gpurun_out/r05_lanes_ab.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({k: d[k] for k in ('value', 'ms_per_step')} | {'lanes': d['config']['lanes'], 'dev_ms': d['config']['device_ms_per_step'], 'finite': d['config']['finite'], 'roofline_frac': d['roofline']['frac'], 'dom_ms': d['roofline']['kernel_ms_per_step']}))" >> $out
}
for rnd in 0 1; do
  run "cfg2 plain (round $rnd)" "A=1" "--lanes 1"
  run "cfg2 lanes 2 priority (round $rnd)" "A=1" "--lanes 2"
  run "cfg2 lanes 2 no priority (round $rnd)" "DMX_LANE_UNET_PRIORITY=0" "--lanes 2"
  run "cfg2 lanes 3 priority (round $rnd)" "A=1" "--lanes 3"
done
for rnd in 0 1; do
  run "cfg3 (4 clips) plain (round $rnd)" "A=1" "--workload dsg_phase_audioldm2 --lanes 1"
  run "cfg3 (4 clips) lanes 2 (round $rnd)" "A=1" "--workload dsg_phase_audioldm2 --lanes 2"
done
echo "== cfg2 lanes 2 / lanes 1 with the 200-step trajectory" >> $out
python bench.py --no-cpu-baseline --no-stage-times --steps 20 --warmup 3 --lanes 2 2>>gpurun_out/r05_lanes_ab.err | tee gpurun_out/r05_bench_lanes2.json | python -c "import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['full_trajectory'])" >> $out
python bench.py --no-cpu-baseline --no-stage-times --steps 20 --warmup 3 --lanes 1 2>>gpurun_out/r05_lanes_ab.err | tee gpurun_out/r05_bench_lanes1.json | python -c "import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['full_trajectory'])" >> $out
cat $out
