import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from gemm_bench_lib import *
EPI = L.EPI_BIAS | L.EPI_RESID | L.EPI_LRELU2 | L.EPI_NO_C
for name, args in [("s2 k3", (8, 20004, 256, 256, 3, 1)), ("s2 k11", (8, 20004, 256, 256, 11, 5)), ("s1 k3", (8, 5001, 512, 512, 3, 1)), ("s1 k11", (8, 5001, 512, 512, 11, 5))]:
    for cfg in (0, 1, 7, 8, 15, 17, 18):
        bench(f"{name} cfg{cfg}", *args, flags=EPI, reps=6, cfg=cfg)
