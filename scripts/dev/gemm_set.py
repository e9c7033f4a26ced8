import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from gemm_bench_lib import *
EPI = L.EPI_BIAS | L.EPI_RESID | L.EPI_LRELU2 | L.EPI_NO_C
for ncu in (32, 256):
    bench(f"1 round on {ncu} CUs k3 256", 1, 256 * ncu, 256, 256, 3, 1, flags=EPI, reps=10, cfg=1)
    bench(f"1 round on {ncu} CUs k3 256 C-only", 1, 256 * ncu, 256, 256, 3, 1, flags=0, reps=10, cfg=1)
    bench(f"1 round on {ncu} CUs k3 320", 1, 320 * ncu, 256, 256, 3, 1, flags=EPI, reps=10, cfg=7)
