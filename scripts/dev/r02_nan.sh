#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for r in 1 2; do for v in "" "--no-nan-check"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-stage-times $v > gpurun_out/nan_$r.json 2> gpurun_out/nan.err || tail -3 gpurun_out/nan.err
  python -c "
import json; d=json.load(open('gpurun_out/nan_$r.json')); print('bench [$v]', d['value'], d['ms_per_step'], d['config']['device_ms_per_step'], d['config'].get('settle_last3_ms'))"
done; done
