"""GPU box: do the three independent resblock branches (k = 3, 7, 11) of a HiFi-GAN level run faster side by side on three streams
than back to back on one?  (In-kernel stamps: all 256 CUs reach their epilogue within ~1 us of each other and share HBM for 8-14 us
while the K loops leave it idle; tiles of different K de-phase the bursts.)  Raw launches of the level's layer shapes."""
import sys, ctypes as C, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from diffmusic_amd import _lib as L
import tune_tiles as T

def mk(M, N, taps, flags, cfg):
    K = N * taps
    x = torch.randn(M, N, device="cuda").half(); w = torch.randn(N, K, device="cuda").half() * 0.05
    out = torch.empty(M, N, device="cuda", dtype=torch.float16); aux = torch.randn(M, N, device="cuda").half(); out2 = torch.empty_like(out)
    bias = torch.zeros(N, device="cuda")
    d = T.desc(A=x, W=w, C=out, C2=out2, R=aux, X=aux, bias=bias, rowbias=bias, M=M, N=N, K=K, ldw=K, Hi=1, Wi=M, Ci=N, lda=N, Hq=1, Wq=M,
               ntaps=taps, Ho=1, Wo=M, ldc=N, ldr=N, ldx=N, ldc2=N, flags=flags, act_slope=0.1, mask_slope=0.1, resid_inv_slope=10.0,
               tdy=[0] * taps, tdx=[t - taps // 2 for t in range(taps)], tile_cfg=cfg)
    d._keep = (x, w, out, aux, out2, bias)
    return d

def go(d, stream):
    L.check(L.lib().dmx_gemm_raw(C.byref(d), C.sizeof(d), C.c_void_p(stream.cuda_stream)), "gemm")

for M, N, flags, what in [(160032, 256, 289, "C=256 fwd"), (160032, 256, 805, "C=256 bwd"), (40008, 512, 289, "C=512 fwd"), (40008, 512, 805, "C=512 bwd")]:
    br = [[mk(M, N, k, flags, 7) for _ in range(6)] for k in (3, 7, 11)]
    s0 = torch.cuda.current_stream(); ss = [torch.cuda.Stream() for _ in range(3)]
    def seq():
        for b in br:
            for d in b: go(d, s0)
    def par():
        e = torch.cuda.Event(); e.record(s0)
        for b, s in zip(br[::-1], ss):               # longest branch first
            s.wait_event(e)
            for d in b: go(d, s)
            e2 = torch.cuda.Event(); e2.record(s); s0.wait_event(e2)
    res = {}
    for name, f in (("one stream", seq), ("three streams", par), ("one stream", seq), ("three streams", par)):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s0)
        for _ in range(5): f()
        e1.record(s0); torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 5)
    print(what, {k: [round(x, 3) for x in v] for k, v in res.items()}, "ms per level (18 launches)", flush=True)
