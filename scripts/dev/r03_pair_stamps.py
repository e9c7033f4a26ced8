"""GPU box, diagnostic library only (conv_pair.hip built with -DDMX_PAIR_STAMPS, loaded through DMX_LIB_PATH): where a workgroup of
the fused resblock-pair kernel spends its life.  Prints mean microseconds per phase over the first 4096 workgroups of one launch:
slab load | stage A K loop | intermediate write-back (+ tape) | stage B K loop | epilogue, and the workgroup's whole lifetime."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from diffmusic_amd import _lib as L
import test_gpu_gemm as G
lib = L.lib()
lib.dmx_pair_stamps_read.argtypes = [C.c_void_p]; lib.dmx_pair_stamps_read.restype = C.c_int
adt = L.act_dtype()
def run(da, db):
    L.check(lib.dmx_conv_pair_raw(C.byref(da) if da is not None else None, C.byref(db), C.sizeof(db), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pair")
for B, T, Cc, k, dil in [(8, 160032, 32, 11, 5), (8, 160032, 32, 3, 1), (8, 80016, 64, 11, 5), (8, 80016, 64, 3, 1), (8, 40008, 128, 11, 5), (8, 40008, 128, 3, 1)]:
    g = torch.Generator().manual_seed(0)
    xa = torch.randn(B, T, Cc, generator=g).to(adt).cuda()
    ha = torch.zeros_like(xa); xn = torch.zeros_like(xa)
    w = (torch.randn(Cc, k * Cc, generator=g) / (Cc * k) ** 0.5).to(adt).cuda()
    b1 = torch.zeros(Cc).cuda()
    hb = torch.zeros(B, T, Cc // 8, dtype=torch.uint8, device="cuda"); xb = torch.zeros_like(hb)
    da = G._conv_desc(L, xa, w, k, dil, Cc, B, T, C=ha, C2=None, B2=hb, ldb2=Cc // 8, bias=b1,
                      flags=L.EPI_BIAS | L.EPI_LRELU2 | L.EPI_NO_C | L.EPI_BITS2, act_slope=0.1)
    db = G._conv_desc(L, ha, w, k, 1, Cc, B, T, C=xn, C2=xn, B2=xb, ldb2=Cc // 8, bias=b1, R=xa, resid_inv_slope=10.0, act_slope=0.1,
                      flags=L.EPI_BIAS | L.EPI_RESID | L.EPI_RESID_INV | L.EPI_LRELU2 | L.EPI_NO_C | L.EPI_BITS2)
    for _ in range(3): run(da, db)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(da, db); e1.record(); torch.cuda.synchronize()
    st = np.zeros(4096 * 8, dtype=np.uint64)
    assert lib.dmx_pair_stamps_read(st.ctypes.data_as(C.c_void_p)) == 0
    st8 = st.reshape(4096, 8).astype(np.float64) / 100.0               # 100 MHz -> us
    mid = (st8[:, 6] - st8[:, 2], st8[:, 7] - st8[:, 6], st8[:, 3] - st8[:, 7])   # residual + barrier | pointwise tail + barrier | tape write
    st = st8[:, :6]
    d = np.diff(st, axis=1)
    life = st[:, 5] - st[:, 0]
    t0 = st[:, 0].min()
    print(f"C={Cc} k={k}: kernel {e0.elapsed_time(e1) * 1e3:7.1f} us | per workgroup: load {d[:,0].mean():5.2f}  stageA {d[:,1].mean():5.2f}  mid {d[:,2].mean():5.2f}  "
          f"stageB {d[:,3].mean():5.2f}  epilogue {d[:,4].mean():5.2f}  life {life.mean():5.2f} us (min {life.min():.2f} max {life.max():.2f}); "
          f"mid = residual {mid[0].mean():.2f} + tail {mid[1].mean():.2f} + tape {mid[2].mean():.2f}", flush=True)
