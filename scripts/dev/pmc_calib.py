import torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
x = torch.randn(16 * 1024 * 1024, device="cuda").half()      # 32 MiB
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
from gemm_bench_lib import *
EPI = L.EPI_BIAS | L.EPI_RESID | L.EPI_LRELU2 | L.EPI_NO_C
bench("probe", 1, 256 * 256, 256, 256, 3, 1, flags=EPI, reps=1, cfg=1)
bench("probe C only", 1, 256 * 256, 256, 256, 3, 1, flags=0, reps=1, cfg=1)
bench("probe legacy C only", 1, 256 * 256, 256, 256, 3, 1, flags=0, reps=1, cfg=3)
