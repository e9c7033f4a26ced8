import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from gemm_bench_lib import *
EPI = L.EPI_BIAS | L.EPI_RESID | L.EPI_LRELU2 | L.EPI_NO_C
for ncu in (32, 256):
    bench(f"1 round on {ncu} CUs k3 256", 1, 256 * ncu, 256, 256, 3, 1, flags=EPI, reps=10, cfg=1)
    bench(f"1 round on {ncu} CUs k3 256 C-only", 1, 256 * ncu, 256, 256, 3, 1, flags=0, reps=10, cfg=1)
bench("s1 k3", 8, 5001, 512, 512, 3, 1, flags=EPI, reps=8)
bench("s1 k11", 8, 5001, 512, 512, 11, 5, flags=EPI, reps=5)
bench("s2 k3", 8, 20004, 256, 256, 3, 1, flags=EPI, reps=8)
bench("s2 k7", 8, 20004, 256, 256, 7, 3, flags=EPI, reps=5)
bench("s2 k11", 8, 20004, 256, 256, 11, 5, flags=EPI, reps=5)
bench("s3 k3", 8, 40008, 128, 128, 3, 1, flags=EPI, reps=8)
bench("s3 k7", 8, 40008, 128, 128, 7, 3, flags=EPI, reps=5)
bench("s3 k11", 8, 40008, 128, 128, 11, 5, flags=EPI, reps=5)
bench("vae 128", 1, 512000, 128, 128, 9, 1, flags=L.EPI_BIAS, reps=5)
