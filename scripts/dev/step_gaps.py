"""Where does a step's wall time go on the host?  Per step: host ms to enqueue the U-Net, host ms to enqueue the scheduler step, host ms
blocked in the NaN-check sync, GPU ms between the step's first and last event, and the GPU idle gap to the previous step."""
import sys, time, torch
sys.path.insert(0, '.')
import bench as Bm
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
pipe, op, meas, lat, cond, L = Bm.build_problem(8, 0, dev, "dps_inpainting")
ts = pipe.scheduler._timesteps_host
b = pipe._bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
rows = []
torch.cuda.synchronize()
for i in range(N):
    t0 = time.perf_counter()
    evs[i][0].record()
    eps = pipe._unet_eps(lat, ts[i], cond, b["gscale"], True)
    t1 = time.perf_counter()
    out = pipe.scheduler.step(eps, ts[i], lat, eta=b["eta"], generator=b["gens"], measurement=meas, vae=pipe.vae, vocoder=pipe.vocoder,
                              original_waveform_length=L, ip_guidance_rate=b["rate"], supervised_space="mel_spectrogram")
    evs[i][1].record()
    t2 = time.perf_counter()
    bad = bool(torch.isnan(out.loss).any())
    t3 = time.perf_counter()
    lat = out.prev_sample
    rows.append((t0, t1, t2, t3))
torch.cuda.synchronize()
print("step  host_unet  host_sched  host_sync   wall   gpu_ms  gpu_gap_to_prev")
for i in range(N):
    t0, t1, t2, t3 = rows[i]
    gpu = evs[i][0].elapsed_time(evs[i][1])
    gap = evs[i - 1][1].elapsed_time(evs[i][0]) if i else 0.0
    print(f"{i:4d} {1e3*(t1-t0):9.2f} {1e3*(t2-t1):10.2f} {1e3*(t3-t2):9.2f} {1e3*(t3-t0):7.2f} {gpu:8.2f} {gap:8.2f}")
