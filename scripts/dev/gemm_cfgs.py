import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from gemm_bench_lib import *
EPI = L.EPI_RESID | L.EPI_LRELU2 | L.EPI_NO_C
for name, args in [("s2 k3", (8, 20004, 256, 256, 3, 1)), ("s1 k3", (8, 5001, 512, 512, 3, 1)), ("s2 k7", (8, 20004, 256, 256, 7, 3)), ("s1 k7", (8, 5001, 512, 512, 7, 3)), ("s2 k11", (8, 20004, 256, 256, 11, 5))]:
    for cfg in (0, 1, 7, 8, 2, 9, 11):
        bench(f"{name} cfg{cfg}", *args, flags=EPI, reps=8, cfg=cfg)
