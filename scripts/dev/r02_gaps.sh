#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for r in 1 2; do echo "=== process $r"; timeout -k 10 200 python scripts/dev/step_gaps.py 24 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gaps_$r.txt; done
