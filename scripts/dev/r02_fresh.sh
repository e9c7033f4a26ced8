#!/bin/bash
# first process on a fresh box: the default bench with / without the conditioning phase
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 300 python bench.py --no-cpu-baseline --no-stage-times $1 > gpurun_out/fresh_$2.json 2> gpurun_out/fresh.err || tail -3 gpurun_out/fresh.err
python -c "
import json; d=json.load(open('gpurun_out/fresh_$2.json')); print('bench', d['value'], d['ms_per_step'], d['config'].get('settle_steps'), d['config'].get('settle_first3_ms'), d['config'].get('settle_last3_ms'))"
