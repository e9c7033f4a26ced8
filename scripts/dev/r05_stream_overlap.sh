#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r05_stream_overlap.log; : > $out
python scripts/dev/r05_stream_overlap.py -1 2>/dev/null | tail -1 >> $out
python scripts/dev/r05_stream_overlap.py 0 2>/dev/null | tail -1 >> $out
GPU_MAX_HW_QUEUES=8 python scripts/dev/r05_stream_overlap.py -1 2>/dev/null | tail -1 | sed 's/^/GPU_MAX_HW_QUEUES=8: /' >> $out
HIP_FORCE_DEV_KERNARG=1 python scripts/dev/r05_stream_overlap.py -1 2>/dev/null | tail -1 | sed 's/^/HIP_FORCE_DEV_KERNARG=1: /' >> $out
cat $out
