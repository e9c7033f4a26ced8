"""GPU box: does running two independent half-size problems side by side on CU-masked streams (each on half of every XCD's CUs) beat running
them back to back on the whole chip?  The 8-wave GEMM tiles alternate between a compute phase that leaves HBM idle (K loop) and a burst
that leaves the matrix pipe idle (prologue loads, epilogue stores), all CUs in lockstep; two lanes that drift out of phase would share
HBM between half as many CUs.  Raw launches of the step's layer shapes at 4 clips per lane vs 8 clips on one stream."""
import sys, time, ctypes as C, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts/dev')
from diffmusic_amd import _lib as L
import tune_tiles as T
hip = C.CDLL("libamdhip64.so")

def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return s

def mk(M, N, taps, flags, cfg=0):
    K = N * taps
    x = torch.randn(M, N, device="cuda").half(); w = torch.randn(N, K, device="cuda").half() * 0.05
    out = torch.empty(M, N, device="cuda", dtype=torch.float16); aux = torch.randn(M, N, device="cuda").half(); out2 = torch.empty_like(out)
    bias = torch.zeros(N, device="cuda")
    d = T.desc(A=x, W=w, C=out, C2=out2, R=aux, X=aux, bias=bias, rowbias=bias, M=M, N=N, K=K, ldw=K, Hi=1, Wi=M, Ci=N, lda=N, Hq=1, Wq=M,
               ntaps=taps, Ho=1, Wo=M, ldc=N, ldr=N, ldx=N, ldc2=N, flags=flags, act_slope=0.1, mask_slope=0.1, resid_inv_slope=10.0,
               tdy=[0] * taps, tdx=[t - taps // 2 for t in range(taps)], tile_cfg=cfg)
    d._keep = (x, w, out, aux, out2, bias)
    return d

def go(d, s):
    L.check(L.lib().dmx_gemm_raw(C.byref(d), C.sizeof(d), s), "gemm")

def wall(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

s0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
even = masked_stream([0x55555555] * 8); odd = masked_stream([0xAAAAAAAA] * 8)
lo = masked_stream([0x0000FFFF] * 8); hi = masked_stream([0xFFFF0000] * 8)
for Mfull, N, taps, flags, what in [(160032, 256, 3, 289, "hifigan C=256 k=3 fwd"), (160032, 256, 3, 805, "hifigan C=256 k=3 bwd"),
                                    (160032, 256, 11, 805, "hifigan C=256 k=11 bwd"), (40008, 512, 7, 289, "hifigan C=512 k=7 fwd"),
                                    (128000, 256, 9, 5, "vae C=256 3x3"), (512000, 128, 9, 5, "vae C=128 3x3")]:
    n = 8
    full = [mk(Mfull, N, taps, flags) for _ in range(n)]
    halfA = [mk(Mfull // 2, N, taps, flags) for _ in range(n)]; halfB = [mk(Mfull // 2, N, taps, flags) for _ in range(n)]
    def one():
        for d in full: go(d, s0)
    def seq_halves():
        for a, b in zip(halfA, halfB): go(a, s0); go(b, s0)
    def lanes(sa, sb):
        def f():
            for a, b in zip(halfA, halfB): go(a, sa); go(b, sb)
        return f
    r = {"whole batch, one stream": wall(one), "two halves, one stream": wall(seq_halves), "two lanes, even/odd CUs": wall(lanes(even, odd)),
         "two lanes, low/high 16 CUs per 32": wall(lanes(lo, hi)), "whole batch again": wall(one)}
    print(what, {k: round(v, 3) for k, v in r.items()}, "ms per", n, "layers", flush=True)
