"""Time the 8-wave tile candidates on the big shapes of a benchmark step (DMX_PROF_CSV dump) -- isolated launches."""
import collections, csv, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from tune_tiles import time_cfg
from diffmusic_amd import _lib as L
shapes = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["cfg"]) in (20, 21, 22, 30) or int(r["cfg"]) >= 40: continue
    key = (int(r["M"]), int(r["N"]), int(r["K"]), int(r["Z"]))
    s = shapes.setdefault(key, dict(taps=int(r["taps"]), flags=int(r["flags"]), ms=0.0, n=0, cfg=int(r["cfg"])))
    s["ms"] += float(r["ms"]); s["n"] += 1
saved = 0.0
for (M, N, K, Z), s in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"]):
    if s["ms"] < float(sys.argv[2] if len(sys.argv) > 2 else 0.15) or M < int(sys.argv[3] if len(sys.argv) > 3 else 30000) or Z > 1: continue
    res = {}
    for c in (1, 2, 7, 8, 9, 10, 11, 19):
        if c in (1, 7, 8) and N % 256: continue
        try: res[c] = time_cfg(M, N, K, Z, s["taps"], s["flags"], c, 4)
        except Exception: pass
    if not res: continue
    best = min(res, key=res.get)
    cur = res.get(s["cfg"], s["ms"] / s["n"])
    saved += (cur - res[best]) * s["n"]
    print(f"M={M:8d} N={N:5d} K={K:6d} n={s['n']:3d} cur cfg {s['cfg']} {cur*1e3:7.1f} us best cfg {best} {res[best]*1e3:7.1f} us  " + " ".join(f"{c}:{t*1e3:.0f}" for c, t in res.items()), flush=True)
print(f"estimated saving: {saved:.2f} ms per step")
