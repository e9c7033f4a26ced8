"""Per-kernel MFMA-pipe utilisation from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES pass:
util = MFMA-busy cycles (summed over the 4 SIMDs of every CU) / (4 x CU-busy cycles)."""
import collections, csv, glob, sys
src, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (name, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[name] += 1
rows = []
for name, c in agg.items():
    mf, cu = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CU_CYCLES", 0.0)
    if cu <= 0: continue
    rows.append((mf, name, cnt[name], cu, mf / (4.0 * cu)))
rows.sort(reverse=True)
with open(out, "w") as fh:
    fh.write("kernel,launches,SQ_VALU_MFMA_BUSY_CYCLES_total,SQ_BUSY_CU_CYCLES_total,mfma_pipe_util\n")
    for mf, name, n, cu, u in rows:
        fh.write(f"\"{name}\",{n},{mf:.0f},{cu:.0f},{u:.4f}\n")
for mf, name, n, cu, u in rows[:12]:
    print(f"{name[:70]:70s} n={n:5d} mfma_util={u:.3f}")
