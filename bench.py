#!/usr/bin/env python
"""Headline benchmark: denoising steps/sec, MusicLDM + DPS music_inpainting, 10 s @16 kHz clips,
200-step schedule, batch 8 per GPU (BASELINE.json configs[1]).  One "step" = one pass of the hot loop
over the batch: U-Net on the 2B CFG batch -> CFG combine -> DPSScheduler.step (x0, VAE decode,
HiFi-GAN, mask, log-mel, L2, hand-written backward sweep, fused update).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0.  Weights are seeded synthetic (no checkpoints offline), clips are
synthetic sums of sinusoids; inputs are resident in HBM before the timed region.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCHED_CFG = dict(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear",
                 trained_betas=None, clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                 timestep_spacing="leading", rescale_betas_zero_snr=False)
SR, SECONDS, N_STEPS, ZETA, GUIDANCE_SCALE = 16000, 10, 200, 5e-4, 2.0
ALGO_TFLOP_PER_CLIP_STEP = 3.52        # BASELINE.md section 2 (U-Net 2x fwd + VAE fwd/dgrad + HiFi-GAN fwd/dgrad)
PEAK_TFLOPS_16BIT = 2500.0             # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)


def synth_clip(k, length):
    """y_k(n) = 0.5*sum_j a_j sin(2 pi f_j n / sr + phi_j) + 0.01 N(0,1), seeded per clip."""
    g = torch.Generator().manual_seed(1000 + k)
    f = torch.exp(torch.empty(4).uniform_(math.log(55.0), math.log(7000.0), generator=g))
    a = torch.empty(4).uniform_(0.2, 1.0, generator=g)
    ph = torch.empty(4).uniform_(0, 2 * math.pi, generator=g)
    n = torch.arange(length, dtype=torch.float32)
    y = 0.5 * (a[:, None] * torch.sin(2 * math.pi * f[:, None] * n[None] / SR + ph[:, None])).sum(0)
    return (y + 0.01 * torch.randn(length, generator=g)).clamp(-1, 1)


WORKLOADS = {
    # name: (pipeline, scheduler, eta, rate, task, clips per GPU, BASELINE.json config)
    "dps_inpainting": ("musicldm", "dps", 0.0, 5e-4, "music_inpainting", 8, "configs[1]"),
    "dsg_phase_audioldm2": ("audioldm2", "dsg", 1.0, 0.08, "phase_retrieval", 4, "configs[2] (4 clips per GPU)"),
    "mpgd_sr4": ("musicldm", "mpgd", 0.0, 5e-3, "super_resolution", 4, "configs[3] (4 clips per GPU)"),
}


def build_problem(B, rank, device, workload="dps_inpainting"):
    from diffmusic_amd.pipelines import get_pipeline
    from diffmusic_amd.schedulers import get_scheduler
    from diffmusic_amd import inverse_problem as P
    from diffmusic_amd.torch_utils import randn_tensor
    pname, sname, eta, rate, task, _, _ = WORKLOADS[workload]
    pipe = get_pipeline(pname).from_pretrained("synthetic", seed=0).to(device)
    noiser = P.get_noiser("gaussian", 0.0)
    if task == "music_inpainting":
        op = P.MusicInpaintingOperator(SECONDS, SR, "box", 2, 3, 0.3, 0.1, 1.0, noiser=noiser)
    elif task == "phase_retrieval":
        op = P.PhaseRetrievalOperator(noiser=noiser)
    else:
        op = P.SuperResolutionOperator(SR, 4, noiser=noiser)
    pipe.scheduler = get_scheduler(sname)(operator=op, **SCHED_CFG)
    pipe.scheduler.set_timesteps(N_STEPS, device=device)
    L = SECONDS * SR
    clips = torch.stack([synth_clip(rank * B + i, L) for i in range(B)]).to(device)
    measurement = op.forward(clips)
    gens = [torch.Generator().manual_seed(rank * B + i) for i in range(B)]
    latents = randn_tensor((B, 8, 250, 16), generator=gens, device=device, dtype=torch.float32)
    g7 = torch.Generator().manual_seed(7)
    if pname == "musicldm":
        pe = torch.nn.functional.normalize(torch.randn(B, 512, generator=g7), dim=-1).to(device)
        cond = dict(class_labels=torch.cat([pe, pe], dim=0))      # prompt="" in the reference: cond == uncond, CFG batch kept at 2B
        gscale = GUIDANCE_SCALE
    else:                                                          # AudioLDM2: GPT-2 states (B,8,768), T5 states (B,16,1024), mask ones
        ge = torch.randn(B, 8, 768, generator=g7).to(device)
        te = torch.randn(B, 16, 1024, generator=g7).to(device)
        cond = dict(class_labels=None, encoder_hidden_states=torch.cat([ge, ge]), encoder_hidden_states_1=torch.cat([te, te]),
                    encoder_attention_mask_1=torch.ones(2 * B, 16, device=device))
        gscale = 3.5
    pipe._bench = dict(eta=eta, rate=rate, gscale=gscale, gens=gens)
    return pipe, op, measurement, latents, cond, L


def one_step(pipe, latents, t, cond, measurement, L):
    b = pipe._bench
    eps = pipe._unet_eps(latents, t, cond, b["gscale"], True)
    out = pipe.scheduler.step(eps, t, latents, eta=b["eta"], generator=b["gens"], measurement=measurement, vae=pipe.vae,
                              vocoder=pipe.vocoder, original_waveform_length=L, ip_guidance_rate=b["rate"],
                              supervised_space="mel_spectrogram")
    return out.prev_sample, out.loss


def cpu_baseline(seed_sd, threads):
    """The CPU restatement (oracle/, fp32 eager torch + autograd) timed on the host cores: 1 clip x 1 DPS step."""
    from oracle import models as OM, operators as OO, schedulers as OS
    torch.set_num_threads(threads)
    unet, vae, voc = OM.UNetMusicLDM().eval(), OM.VaeDecoder().eval(), OM.HifiGan().eval()
    unet.load_state_dict(seed_sd["unet"], strict=True)
    vae.load_state_dict(seed_sd["vae"], strict=True)
    voc.load_state_dict(seed_sd["vocoder"], strict=False)
    op = OO.MusicInpaintingOperator(SECONDS, SR, "box", 2, 3, 0.3, 0.1, 1.0, noiser=OO.get_noiser("gaussian", 0.0))
    sched = OS.DPSScheduler(operator=op, **SCHED_CFG)
    sched.set_timesteps(N_STEPS)
    L = SECONDS * SR
    y = op.forward(synth_clip(0, L)[None])
    x = torch.randn(1, 8, 250, 16, generator=torch.Generator().manual_seed(0))
    pe = torch.nn.functional.normalize(torch.randn(1, 512, generator=torch.Generator().manual_seed(7)), dim=-1)
    ts = [int(t) for t in sched.timesteps]
    n_meas, t0 = 3, None
    for i in range(1 + n_meas):                  # 1 warm-up + 3 measured steps
        if i == 1:
            t0 = time.perf_counter()
        with torch.no_grad():
            e2 = unet(torch.cat([x, x]), ts[i], class_labels=torch.cat([pe, pe]))[0]
        e = e2[:1] + GUIDANCE_SCALE * (e2[1:] - e2[:1])
        x = sched.step(e, ts[i], x, eta=0.0, measurement=y, vae=vae, vocoder=voc, original_waveform_length=L,
                       ip_guidance_rate=ZETA, supervised_space="mel_spectrogram").prev_sample
    return (time.perf_counter() - t0) / n_meas


def pmc_traffic(prefix):
    """HBM-side bytes per launch of one kernel family (its 8-wave tiles, >= 192 rows) from the committed counter passes
    (profiles/r01_pmc_{FETCH,WRITE}_SIZE_per_kernel.csv: separate `rocprofv3 --pmc` runs of this same command,
    KiB -> bytes, FETCH_SIZE doubled per the gfx950 correction).  Counters cannot be read from inside the process."""
    import csv
    tot, n = 0.0, 0
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            with open(os.path.join(here, "profiles", f"r01_pmc_{c}_per_kernel.csv")) as fh:
                rows = [r for r in csv.DictReader(fh) if r["kernel"].startswith(prefix + "<") and int(r["kernel"].split("<")[1].split(",")[0]) >= 192]
            tot += sum(float(r[f"{c}_bytes_total"]) for r in rows)
            n = sum(int(r["launches"]) for r in rows)
    except (OSError, KeyError):
        return None, None
    return (round(tot / n) if n else None), "profiles/r01_pmc_{FETCH,WRITE}_SIZE_per_kernel.csv (rocprofv3 --pmc, separate passes)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nan-check", action="store_true", help="skip the per-step host-side NaN test of the loss (the reference loop has it)")
    ap.add_argument("--workload", default="dps_inpainting", choices=sorted(WORKLOADS),
                    help="default = the headline config (BASELINE.json configs[1]); the others are the remaining GPU configs")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    wl = args.workload
    B = args.batch if wl == "dps_inpainting" else WORKLOADS[wl][5]
    pipe, op, measurement, latents, pe2, L = build_problem(B, rank, device, wl)
    ts = pipe.scheduler._timesteps_host

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    k = 0
    for _ in range(args.warmup):
        latents, _ = one_step(pipe, latents, ts[k % N_STEPS], pe2, measurement, L)
        k += 1
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    loss = None
    nan_steps = 0
    for _ in range(args.steps):
        latents, loss = one_step(pipe, latents, ts[k % N_STEPS], pe2, measurement, L)
        if not args.no_nan_check:            # the loop body's NaN test (pipeline_musicldm.py:741-742): one host sync per step
            nan_steps += int(bool(torch.isnan(loss).any()))
        k += 1
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    tmax = torch.tensor([wall], dtype=torch.float64, device=device)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    finite = bool(torch.isfinite(loss).all()) and bool(torch.isfinite(latents).all())

    # ---- roofline leg: one extra step with HIP events around every implicit-GEMM launch
    import ctypes as C
    from diffmusic_amd import _lib as Lb
    Lb.lib().dmx_prof_begin()
    one_step(pipe, latents, ts[k % N_STEPS], pe2, measurement, L)
    ms, fl = C.c_double(), C.c_double()
    n_launch = Lb.lib().dmx_prof_end(C.byref(ms), C.byref(fl))
    # headline config: the analytic count of BASELINE.md; other workloads: the FLOPs the launches actually issued
    algo_tflop_step = ALGO_TFLOP_PER_CLIP_STEP * B if wl == "dps_inpainting" else fl.value / 1e12
    dms, dfl, dby = C.c_double(), C.c_double(), C.c_double()
    n_dom = Lb.lib().dmx_prof_dominant(C.byref(dms), C.byref(dfl), C.byref(dby))
    traffic, traffic_src = pmc_traffic("gemm_glds_kernel") if args.workload == "dps_inpainting" else (None, None)
    # dominant kernel = gemm_glds_kernel (LDS-DMA tiles): its issued FLOPs scaled by algorithmic/issued of the step
    scale = algo_tflop_step / (fl.value / 1e12) if fl.value > 0 else 1.0
    dom_tflop = dfl.value / 1e12 * scale
    achieved = dom_tflop / (dms.value * 1e-3) if dms.value > 0 else 0.0
    all_rate = algo_tflop_step / (ms.value * 1e-3) if ms.value > 0 else 0.0
    roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_TFLOPS_16BIT, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS_16BIT, 4), "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(dby.value / max(n_dom, 1)),
                "kernel": "gemm_glds_kernel (implicit-GEMM conv / batched NT GEMM on LDS-DMA tiles, fp16 MFMA)",
                "launches_per_step": n_dom, "kernel_ms_per_step": round(dms.value, 3), "avg_launch_us": round(1e3 * dms.value / max(n_dom, 1), 1),
                "algorithmic_tflop_per_step": round(dom_tflop, 2),
                "all_gemm_kernels": {"launches_per_step": n_launch, "kernel_ms_per_step": round(ms.value, 3),
                                     "algorithmic_tflop_per_step": round(algo_tflop_step, 2), "issued_tflop_per_step": round(fl.value / 1e12, 2),
                                     "achieved": round(all_rate, 1), "frac": round(all_rate / PEAK_TFLOPS_16BIT, 4)},
                "step_share": round(ms.value / (1e3 * wall / args.steps), 3)}

    if rank == 0:
        steps_per_s = args.steps / wall * world                 # one step advances B clips on each of `world` GPUs
        res = {"metric": "denoising steps/sec (10 s clip, 200-step DPS)", "value": round(steps_per_s, 4),
               "unit": f"steps/s (each step advances a batch of {B} clips per GPU)", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * wall / args.steps, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": {"dps_inpainting": "MusicLDM + DPS music_inpainting, 10 s @16 kHz, 200-step schedule, batch 8 per GPU "
                                      "(BASELINE.json configs[1])"}.get(wl, f"{wl}: {WORKLOADS[wl][0]} + {WORKLOADS[wl][1]} {WORKLOADS[wl][4]}, 10 s @16 kHz, "
                                                                        f"200-step schedule, {WORKLOADS[wl][6]}"),
                          "global_batch": B * world, "clips_per_gpu": B,
                          "clip_steps_per_sec": round(steps_per_s * B, 3), "parallelism": f"clip-sharded x{world}",
                          "device_ms_per_step": round(dev_ms / args.steps, 3), "finite": finite, "nan_check_per_step": not args.no_nan_check,
                          "final_loss_clip0": float(loss.reshape(-1)[0])},
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline and wl == "dps_inpainting":
            threads = min(16, len(os.sched_getaffinity(0)))      # the GPU box's CPU share, not the host's core count
            sd = {"unet": pipe.unet.synth_state_dict(0), "vae": pipe.vae.synth_state_dict(1), "vocoder": pipe.vocoder.synth_state_dict(2)}
            print(f"[bench] timing the CPU oracle on {threads} threads (1 clip x 4 steps) ...", file=sys.stderr, flush=True)
            sec = cpu_baseline(sd, threads)
            print(f"[bench] CPU oracle: {sec:.2f} s per clip-step", file=sys.stderr, flush=True)
            res["cpu_baseline"] = {"value": round(1.0 / (sec * B), 6), "unit": "steps/s (batch-8 equivalent, extrapolated from 1 clip)",
                                   "cores": threads, "kind": "port",
                                   "sample": f"1 clip x 3 DPS steps after 1 warm-up (U-Net 2x fwd + guided step each), fp32 eager torch + autograd, {sec:.2f} s/step"}
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
