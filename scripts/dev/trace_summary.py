"""rocprofv3 --kernel-trace CSV of `unet_only.py` (3 warm-up + n timed forwards) -> launches per forward, device time per forward, and the
ordered launch list of the LAST forward (name, duration us, gap to the previous kernel's end us).  usage: trace_summary.py trace.csv n_forwards"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]
# forwards are delimited by the first kernel of a forward: timestep_embed_kernel (one per forward)
starts = [i for i, r in enumerate(rows) if "timestep_embed_kernel" in r["Kernel_Name"]]
# the conversion of the contexts precedes it for AudioLDM2: take f32_to_bf16 directly before as part of the same forward
fw = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    fw.append(rows[a:b])
fw = fw[-(nf - 3):] if len(fw) >= nf else fw
per = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in f) / 1e3 for f in fw]
span = [(int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])) / 1e3 for f in fw]
print(f"forwards {len(fw)}  launches per forward {sorted(len(f) for f in fw)[len(fw)//2]}  kernel-time sum per forward {sorted(per)[len(per)//2]:.1f} us  "
      f"first-start..last-end span {sorted(span)[len(span)//2]:.1f} us")
last = fw[-2] if len(fw) > 1 else fw[-1]
agg = collections.OrderedDict()
for r in last:
    k = short(r["Kernel_Name"]); a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("--- per kernel in one forward (count, total us)")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"{a[0]:4d} {a[1]:9.1f}  {k}")
print("--- sequence: idx dur_us gap_us grid wg name")
prev = None
for i, r in enumerate(last):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    print(f"{i:4d} {(e - s)/1e3:8.2f} {gap:7.2f} {r.get('Grid_Size_X','?'):>8s} {r.get('Workgroup_Size_X','?'):>5s} {short(r['Kernel_Name'])}")
