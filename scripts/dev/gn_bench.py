"""GroupNorm forward on the VAE's big tensors through dmx_groupnorm_raw: time per call (three launches) for the current geometry."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from diffmusic_amd import _lib as L
for (B, P, Cc) in ((8, 64000, 128), (8, 64000, 256), (8, 16000, 256), (8, 16000, 512)):
    x = torch.randn(B, P, Cc, device="cuda").half(); y = torch.empty_like(x)
    G = 32
    gamma = torch.ones(Cc, device="cuda"); beta = torch.zeros(Cc, device="cuda")
    stats = torch.empty(B, G, 2, device="cuda"); scale = torch.empty(B, Cc, device="cuda"); shift = torch.empty(B, Cc, device="cuda")
    part = torch.empty(L.lib().dmx_groupnorm_scratch_floats(B, Cc, G), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: L.lib().dmx_groupnorm_raw(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()),
                                          C.c_void_p(stats.data_ptr()), C.c_void_p(scale.data_ptr()), C.c_void_p(shift.data_ptr()), C.c_void_p(part.data_ptr()),
                                          B, P, Cc, G, 1e-5, 1, st)
    big = torch.empty(600 << 20, dtype=torch.uint8, device="cuda")
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        big.zero_()                                   # push x out of the Infinity Cache, as the producing conv's traffic does in the real step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    by = B * P * Cc * 2
    print(f"B={B} P={P} C={Cc}: {1e3*ts[len(ts)//2]:7.1f} us per norm (3 tensor passes of {by/1e6:.0f} MB -> {3*by/ts[len(ts)//2]/1e9:.2f} TB/s)")
