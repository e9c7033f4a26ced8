import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from gemm_bench_lib import *
EPI = L.EPI_BIAS | L.EPI_RESID | L.EPI_LRELU2 | L.EPI_NO_C
bench("1 round on 256 CUs k3 256", 1, 256 * 256, 256, 256, 3, 1, flags=EPI, reps=2, cfg=1)
