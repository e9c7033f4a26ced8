#!/bin/bash
# shapes of the other GPU workloads (their own batch sizes) -> small-M tile / split-K plans appended to the table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for w in dsg_phase_audioldm2 mpgd_sr4 diffmusic_style_audioldm2; do
  DMX_PROF_CSV=gpurun_out/shapes_$w.csv timeout -k 10 200 python bench.py --workload $w --steps 2 --warmup 1 --settle 0 --no-cpu-baseline --no-stage-times > /dev/null 2> gpurun_out/tune_wl.err || tail -3 gpurun_out/tune_wl.err
  wc -l gpurun_out/shapes_$w.csv
done
timeout -k 10 1000 python scripts/dev/tune_small.py gpurun_out/shapes_dsg_phase_audioldm2.csv gpurun_out/shapes_mpgd_sr4.csv gpurun_out/shapes_diffmusic_style_audioldm2.csv > gpurun_out/tune_wl.log 2>&1; echo "tune rc=$?"; tail -3 gpurun_out/tune_wl.log | cut -c1-200
