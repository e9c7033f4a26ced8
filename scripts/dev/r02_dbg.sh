#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for np in "" 1; do for c0 in 128 512; do for sl in 1.0 0.1; do
  if [ -n "$np" ]; then export DMX_NO_PAIR=1; else unset DMX_NO_PAIR; fi
  C0=$c0 SLOPE=$sl timeout -k 10 120 python scripts/dev/dbg_bits.py 2>&1 | grep -v amdgpu.ids
done; done; done
