"""rocprofv3 --kernel-trace CSV of `bench.py --lanes 2` -> why the staggered clip lanes lose.
Kernels are attributed to the U-Net chain or to the guidance sweep by the hardware queue they ran on (the high-priority U-Net stream and
the sweep stream map to different queues).  Printed: per queue the number of launches and the sum of kernel durations; how much of the
U-Net queue's kernel time overlaps kernels of the sweep queue; and for the kernels that only the U-Net launches, the average duration
under a concurrent sweep against the same kernels in the plain loop's trace (second argument).
usage: r05_lanes_trace.py lanes_trace.csv plain_trace.csv"""
import csv, sys, collections


def load(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    return rows


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


lanes, plain = load(sys.argv[1]), load(sys.argv[2])
# the last 40 % of the trace = steady-state timed steps (the settle / warm-up steps come first)
t0 = lanes[0]["s"] + 0.6 * (lanes[-1]["e"] - lanes[0]["s"])
win = [r for r in lanes if r["s"] >= t0]
byq = collections.defaultdict(list)
for r in win:
    byq[r["Queue_Id"]].append(r)
unet_only = ("flash_attn_fwd_kernel", "timestep_embed", "gn_small_kernel", "cfg_combine", "splitk_epilogue_kernel", "gn_finalize_apply_kernel")
uq = max(byq, key=lambda q: sum(any(u in r["Kernel_Name"] for u in unet_only) for r in byq[q]))
print(f"window {1e-6 * (win[-1]['e'] - win[0]['s']):.2f} ms, {len(win)} launches, queues: " +
      ", ".join(f"{q}{' (U-Net)' if q == uq else ''}: {len(v)} launches, {1e-6 * sum(r['e'] - r['s'] for r in v):.2f} ms of kernel time" for q, v in byq.items()))
# overlap of U-Net-queue kernels with kernels of any other queue
others = sorted(((r["s"], r["e"]) for q, v in byq.items() if q != uq for r in v))
merged = []
for s, e in others:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
import bisect
starts = [m[0] for m in merged]
tot = ov = 0
for r in byq[uq]:
    tot += r["e"] - r["s"]
    i = max(0, bisect.bisect_right(starts, r["s"]) - 1)
    while i < len(merged) and merged[i][0] < r["e"]:
        ov += max(0, min(r["e"], merged[i][1]) - max(r["s"], merged[i][0]))
        i += 1
print(f"U-Net queue: {1e-6 * tot:.2f} ms of kernel time in the window, {100.0 * ov / max(tot, 1):.1f} % of it concurrent with a sweep kernel")
# same kernels, plain loop vs under the sweep
pl = collections.defaultdict(list)
for r in plain[int(0.6 * len(plain)):]:
    pl[short(r["Kernel_Name"])].append(r["e"] - r["s"])
ln = collections.defaultdict(list)
for r in byq[uq]:
    ln[short(r["Kernel_Name"])].append(r["e"] - r["s"])
print("kernel (U-Net queue)                                          launches   avg us under sweep   avg us plain loop")
for k, v in sorted(ln.items(), key=lambda kv: -sum(kv[1]))[:18]:
    p = pl.get(k)
    print(f"{k:62s} {len(v):6d} {1e-3 * sum(v) / len(v):18.1f} {(1e-3 * sum(p) / len(p)) if p else float('nan'):18.1f}")
# the chain as a whole: first-start .. last-end of each U-Net forward (delimited by timestep_embed_kernel)
fw = [i for i, r in enumerate(byq[uq]) if "timestep_embed" in r["Kernel_Name"]]
spans = [1e-6 * (byq[uq][b - 1]["e"] - byq[uq][a]["s"]) for a, b in zip(fw, fw[1:])]
if spans:
    print(f"U-Net forward (batch of one lane) first launch .. last launch end under the other lane's sweep: median {sorted(spans)[len(spans) // 2]:.2f} ms over {len(spans)} forwards")
