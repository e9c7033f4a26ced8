#!/bin/bash
# MFMA-pipe / wait / LDS counters of the GEMM micro-bench (separate passes, kernel-trace only beside --pmc)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/pmcg
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf /tmp/pg$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pg$i -o p --output-format csv -- python ${PMC_SCRIPT:-scripts/dev/gemm_bench.py} > gpurun_out/pmcg/run$i.log 2>&1 || tail -3 gpurun_out/pmcg/run$i.log
  python - $i <<'PY'
import csv, glob, sys, collections
i = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'/tmp/pg{i}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'gemm_' not in n and 'conv_pair' not in n: continue
        key = (n.replace('(anonymous namespace)::', '').replace('void ', '')[:48], r.get('Grid_Size', ''))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
with open(f'gpurun_out/pmcg/set{i}.txt', 'w') as out:
    for key, cs in agg.items():
        line = f"{key[0]:48s} grid {key[1]:>9s} n={len(next(iter(cs.values()))):3d} " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items()))
        out.write(line + "\n"); print(line)
PY
done
