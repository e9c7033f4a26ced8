"""Time the fused HiFi-GAN resblock pair kernel at the benchmark shapes (fwd and bwd forms)."""
import sys, ctypes as C, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from diffmusic_amd import _lib as L
import test_gpu_gemm as G

def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

def run(da, db):
    L.check(L.lib().dmx_conv_pair_raw(C.byref(da) if da is not None else None, C.byref(db), C.sizeof(db), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pair")

adt = L.act_dtype()
cases = [(8, 40008, 128, 3, 1), (8, 40008, 128, 7, 3), (8, 40008, 128, 11, 5), (8, 160032, 32, 3, 1), (8, 160032, 32, 7, 3), (8, 160032, 32, 11, 5), (8, 80016, 64, 3, 1), (8, 80016, 64, 7, 3), (8, 80016, 64, 11, 5)]
for B, T, Cc, k, dil in cases:
    g = torch.Generator().manual_seed(0)
    xa = torch.randn(B, T, Cc, generator=g).to(adt).cuda()
    ha = torch.zeros_like(xa); xn = torch.zeros_like(xa); gc = torch.randn(B, T, Cc, generator=g).to(adt).cuda()
    w = (torch.randn(Cc, k * Cc, generator=g) / (Cc * k) ** 0.5).to(adt).cuda()
    b1 = torch.zeros(Cc).cuda()
    hb = torch.zeros(B, T, Cc // 8, dtype=torch.uint8, device="cuda"); xb = torch.zeros_like(hb)
    # the forms the HiFi-GAN executor launches: sign-bit tape out (forward), sign-bit masks in (backward)
    da = G._conv_desc(L, xa, w, k, dil, Cc, B, T, C=ha, C2=None, B2=hb, ldb2=Cc // 8, bias=b1,
                      flags=L.EPI_BIAS | L.EPI_LRELU2 | L.EPI_NO_C | L.EPI_BITS2, act_slope=0.1)
    db = G._conv_desc(L, ha, w, k, 1, Cc, B, T, C=xn, C2=xn, B2=xb, ldb2=Cc // 8, bias=b1, R=xa, resid_inv_slope=10.0, act_slope=0.1,
                      flags=L.EPI_BIAS | L.EPI_RESID | L.EPI_RESID_INV | L.EPI_LRELU2 | L.EPI_NO_C | L.EPI_BITS2)
    tf = bench(lambda: run(da, db))
    ea = G._conv_desc(L, gc, w, k, 1, Cc, B, T, flip=True, C=ha, XB=hb, ldxb=Cc // 8, flags=L.EPI_MASKBITS, mask_slope=0.1)
    eb = G._conv_desc(L, ha, w, k, dil, Cc, B, T, flip=True, C=xn, XB=xb, ldxb=Cc // 8, R=gc, mask_slope=0.1, flags=L.EPI_MASKBITS | L.EPI_RESID)
    tb = bench(lambda: run(ea, eb))
    ts = bench(lambda: run(None, db))
    fl = 2 * 2.0 * B * T * Cc * Cc * k
    by = B * T * Cc * 2
    print(f"C={Cc} k={k} dil={dil}: fwd {tf:7.1f} us ({fl/tf/1e6:6.0f} TF/s, {2.125*by/tf/1e3:6.0f} GB/s min-traffic)  bwd {tb:7.1f} us ({fl/tb/1e6:6.0f} TF/s, {2.125*by/tb/1e3:6.0f} GB/s)  single {ts:7.1f} us", flush=True)
