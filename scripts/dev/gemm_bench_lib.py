"""Micro-benchmark of the implicit-GEMM kernels on the hot shapes of config 2 (through dmx_gemm_raw)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from diffmusic_amd import _lib as L

def desc(**kw):
    d = L.GemmDesc(); d.Z = d.Zi = 1; d.sy = d.sx = d.osy = d.osx = 1; d.alpha = 1.0
    for k, v in kw.items():
        if k in ("tdy", "tdx"):
            for i, t in enumerate(v): getattr(d, k)[i] = t
        elif isinstance(v, torch.Tensor): setattr(d, k, v.data_ptr())
        else: setattr(d, k, v)
    return d

def bench(name, B, T, Ci, Co, k, dil=1, reps=20, flags=0, cfg=0):
    x = torch.randn(B, T, Ci, device="cuda").half(); w = (torch.randn(Co, k * Ci, device="cuda") / (k * Ci) ** 0.5).half()
    out = torch.empty(B, T, Co, device="cuda", dtype=torch.float16); res = torch.randn(B, T, Co, device="cuda").half(); out2 = torch.empty_like(out)
    bias = torch.randn(Co, device="cuda")
    pad = (k * dil - dil) // 2
    d = desc(A=x, W=w, C=out, C2=out2, R=res, bias=bias, M=B * T, N=Co, K=k * Ci, ldw=k * Ci, Hi=1, Wi=T, Ci=Ci, lda=Ci, Hq=1, Wq=T, ntaps=k,
             Ho=1, Wo=T, ldc=Co, ldr=Co, ldx=Co, ldc2=Co, flags=flags, act_slope=0.1, tdy=[0] * k, tdx=[t * dil - pad for t in range(k)], tile_cfg=cfg)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: L.check(L.lib().dmx_gemm_raw(C.byref(d), C.sizeof(d), st), "gemm")
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:34s} M={B*T:8d} N={Co:4d} K={k*Ci:5d}  {ms*1e3:8.1f} us  {2.0*B*T*Co*k*Ci/ms/1e9:7.1f} TF/s", flush=True)

