#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "-DDMX_X" "-DDMX_NOEPI"; do
  DMX_EXTRA_FLAGS="$v" python -m diffmusic_amd.build --force > gpurun_out/build.log 2>&1 || { echo build failed; continue; }
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pp; timeout -k 10 120 rocprofv3 --kernel-trace --pmc $c -d /tmp/pp -o p --output-format csv -- python scripts/dev/gemm_probe1.py > /tmp/pp.log 2>&1
    python - "$v" "$c" <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob('/tmp/pp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_glds' in r['Kernel_Name']:
            a=agg[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
print(sys.argv[1], {k:(v[0], round(v[1]/max(v[0],1))) for k,v in agg.items()})
PY
  done
done
