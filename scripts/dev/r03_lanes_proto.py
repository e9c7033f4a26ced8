"""GPU box: the guidance sweep (VAE decode -> HiFi-GAN -> mel loss -> HiFi-GAN backward -> VAE backward) of the benchmark batch as ONE
batch of 8 on the whole chip against TWO lanes of 4 clips on CU-masked streams (each lane on half of every XCD's CUs, own engines and
operator state).  Prints milliseconds per sweep and whether the lanes' loss / gradient equal the whole-batch result bitwise."""
import sys, time, ctypes as C, torch
sys.path.insert(0, '.')
import bench
from diffmusic_amd import _lib as L
hip = C.CDLL("libamdhip64.so")

def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(len(words)), arr) == 0
    return torch.cuda.ExternalStream(s.value)

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "dps_inpainting"
B = 8
full = bench.build_problem(B, 0, dev, wl)
lanes = [bench.build_problem(B // 2, 0, dev, wl, clip_ids=list(range(h * B // 2, (h + 1) * B // 2))) for h in range(2)]
streams = [masked_stream([0x55555555] * 8), masked_stream([0xAAAAAAAA] * 8)]
g = torch.Generator().manual_seed(3)
x0 = (0.5 * torch.randn(B, 8, 250, 16, generator=g)).to(dev)
Lw = full[5]

def sweep_full():
    pipe, op, meas = full[0], full[1], full[2]
    return pipe.scheduler._guidance(x0, meas, pipe.vae, pipe.vocoder, Lw, "mel_spectrogram")

def sweep_lanes(budget=256):
    main = torch.cuda.current_stream()
    e0 = torch.cuda.Event(); e0.record(main)
    outs = []
    if budget != 256: L.lib().dmx_set_cu_budget(budget)       # (experimental entry point, removed with the experiment: run with budget = 256)
    for h, (pr, s) in enumerate(zip(lanes, streams)):
        pipe, op, meas = pr[0], pr[1], pr[2]
        with torch.cuda.stream(s):
            s.wait_event(e0)
            outs.append(pipe.scheduler._guidance(x0[h * B // 2:(h + 1) * B // 2].contiguous(), meas, pipe.vae, pipe.vocoder, Lw, "mel_spectrogram"))
            e = torch.cuda.Event(); e.record(s)
        main.wait_event(e)
    if budget != 256: L.lib().dmx_set_cu_budget(256)
    return outs

def unet():
    pipe, lat, cond = full[0], full[3], full[4]
    return pipe._unet_eps(lat, int(pipe.scheduler._timesteps_host[60]), cond, pipe._bench["gscale"], True)

def wall(f, reps=5):
    # a step = U-Net on the whole chip, then the sweep; the host enqueues the sweep while the U-Net runs, as in the product
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        unet(); f(); torch.cuda.synchronize()   # (the product syncs once per step for the NaN check)
    return (time.perf_counter() - t0) / reps * 1e3

lf, gf, sf = sweep_full()
lo = sweep_lanes()
torch.cuda.synchronize()
ll = torch.cat([o[0].reshape(-1) for o in lo]); gl = torch.cat([o[1] for o in lo]); sl = torch.cat([o[2].reshape(-1) for o in lo])
print("loss equal", torch.equal(lf.reshape(-1), ll), "grad equal", torch.equal(gf, gl), "scale equal", torch.equal(sf.reshape(-1), sl),
      "max rel grad diff", float((gf - gl).abs().max() / gf.abs().max()), flush=True)
print(f"U-Net alone {wall(lambda: None):.3f} ms", flush=True)
for rnd in range(2):
    print(f"round {rnd}: whole batch {wall(sweep_full):.3f} ms | two lanes on CU-masked streams {wall(sweep_lanes):.3f} ms", flush=True)
# the same two lanes on ordinary (unmasked) streams
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
print(f"two lanes on unmasked streams: {wall(sweep_lanes):.3f} ms", flush=True)
