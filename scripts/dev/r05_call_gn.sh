#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_groupnorm.py tests/test_gpu_vae.py tests/test_gpu_batch_parity.py -x -q -k "not trajectory and not config1" 2>&1 | tail -3
rm -rf /tmp/pb; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o b --output-format csv -- python bench.py --steps 8 --warmup 1 --settle 3 --no-cpu-baseline --no-stage-times --no-full-trajectory > /dev/null 2>&1
grep -E "gn_parts_kernel|gn_bwd_parts" $(find /tmp/pb -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | cut -c1-140
timeout -k 10 200 python bench.py --no-cpu-baseline --no-full-trajectory --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['stage_ms'])"
