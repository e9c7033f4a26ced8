#!/bin/bash
# GPU box: per-kernel trace of U-Net forwards (both variants): stats CSV + the ordered launch list of ONE forward with durations and gaps
# usage: r04_unet_trace.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-r04}; out=gpurun_out/${tag}_unet; mkdir -p $out
for kind in musicldm audioldm2; do
  python scripts/dev/unet_only.py $kind > $out/time_$kind.log 2>&1; cat $out/time_$kind.log
  rm -rf /tmp/pu; rocprofv3 --kernel-trace --stats -d /tmp/pu -o u --output-format csv -- python scripts/dev/unet_only.py $kind > $out/prof_$kind.log 2>&1
  cp $(find /tmp/pu -name "*kernel_stats.csv" | head -1) $out/${kind}_kernel_stats.csv
  cp $(find /tmp/pu -name "*kernel_trace.csv" | head -1) /tmp/${kind}_trace.csv
  python scripts/dev/trace_summary.py /tmp/${kind}_trace.csv 13 > $out/${kind}_forward_sequence.txt
  head -12 $out/${kind}_forward_sequence.txt
done
