"""Summarise a DMX_PROF_CSV dump: time per (cfg, shape)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    k = (int(r['cfg']), int(r['M']), int(r['N']), int(r['K']), int(r['Z']), int(r['taps']))
    a = agg[k]; a[0] += 1; a[1] += float(r['ms']); a[2] += float(r['tflops']) * float(r['ms'])
tot = sum(a[1] for a in agg.values()); print('total ms', round(tot, 2), 'launches', len(rows))
bycfg = collections.defaultdict(float)
for k, a in agg.items(): bycfg[k[0]] += a[1]
print('by cfg', {k: round(v, 2) for k, v in sorted(bycfg.items())})
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print(k, a[0], round(a[1], 3), 'ms', round(1e3 * a[1] / a[0], 1), 'us/launch', round(a[2] / a[1]), 'TF/s')
