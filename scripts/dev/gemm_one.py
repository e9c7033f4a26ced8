import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from gemm_bench_lib import *
EPI = L.EPI_BIAS | L.EPI_RESID | L.EPI_LRELU2
which = sys.argv[1] if len(sys.argv) > 1 else "s2"
if which == "s2": bench("hifigan s2 k11 (N256)", 8, 20004, 256, 256, 11, 5, flags=EPI, reps=5)
if which == "vae": bench("vae conv-like N512", 1, 128000, 512, 512, 9, 1, flags=0, reps=5)
