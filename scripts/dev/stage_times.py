"""Per-stage device times of one guided step (config 2), via torch events."""
import sys, torch, ctypes as C
sys.path.insert(0, '.')
import bench
from diffmusic_amd import _lib as L
pipe, op, meas, lat, pe2, Lw = bench.build_problem(8, 0, torch.device("cuda"))
sch = pipe.scheduler
ts = sch._timesteps_host
def ev(): return torch.cuda.Event(enable_timing=True)
for _ in range(2): lat, _ = bench.one_step(pipe, lat, ts[0], pe2, meas, Lw)
acc = {}
def timed(name, f):
    a, b = ev(), ev(); a.record(); r = f(); b.record(); torch.cuda.synchronize(); acc[name] = acc.get(name, 0) + a.elapsed_time(b); return r
N = 3
for i in range(N):
    t = ts[i + 2]
    eps = timed("unet(2B)+cfg", lambda: pipe._unet_eps(lat, t, pe2, 2.0, True))
    x = lat.float().contiguous(); x0 = torch.empty_like(x)
    _, a_t, a_p, sigma = sch._scalars(t, 0.0)
    L.lib().dmx_sched_pred_x0(C.c_void_p(x.data_ptr()), C.c_void_p(eps.data_ptr()), C.c_void_p(x0.data_ptr()), x.numel(), a_t, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    zs = 1.0 / pipe.vae.config.scaling_factor
    mel = timed("vae fwd", lambda: pipe.vae.decode_hip(x0, z_scale=zs, keep_state=True))
    wav = timed("hifigan fwd", lambda: pipe.vocoder.forward(mel))
    loss, dwav = timed("operator+mel+loss fwd/bwd", lambda: op.guidance(wav, Lw, meas, "mel_spectrogram"))
    inv = torch.empty(8, device="cuda")
    L.lib().dmx_grad_normalize(C.c_void_p(dwav.data_ptr()), C.c_void_p(inv.data_ptr()), 8, dwav.shape[1], 64.0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    dmel = timed("hifigan bwd", lambda: pipe.vocoder.backward(dwav))
    g0 = timed("vae bwd", lambda: pipe.vae.backward(dmel, z_scale=zs))
tot = sum(acc.values())
for k, v in acc.items(): print(f"{k:28s} {v/N:7.2f} ms  {100*v/tot:5.1f}%")
print(f"{'sum':28s} {tot/N:7.2f} ms")
