#!/usr/bin/env python
"""Headline benchmark: denoising steps/sec, MusicLDM + DPS music_inpainting, 10 s @16 kHz clips,
200-step schedule, batch 8 per GPU (BASELINE.json configs[1]; --batch / --global-batch / --workload select other cases).  One "step" = one pass of the hot loop
over the batch: U-Net on the 2B CFG batch -> CFG combine -> DPSScheduler.step (x0, VAE decode,
HiFi-GAN, mask, log-mel, L2, hand-written backward sweep, fused update).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process starts N rank processes itself (fresh children, started before
anything here touches the GPU), one per device, RCCL rendezvous on 127.0.0.1, and exits with the worst
child exit code.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
ranks are the launcher's; a world size that differs from --gpus is an error (exit 3).

Prints ONE JSON line on rank 0.  Weights are seeded synthetic (no checkpoints offline), clips are
synthetic sums of sinusoids; inputs are resident in HBM before the timed region.
After the timed loop the final latents are decoded and the (B, 160000) waveforms of all ranks are
all-gathered once (the path's only collective); its time is reported next to the loop's.
"""
import argparse
import json
import math
import os
import subprocess
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
DEFAULT_LANES = 1
ALGO_TFLOP_PER_CLIP_STEP = 3.52        # BASELINE.md section 2 (U-Net 2x fwd + VAE fwd/dgrad + HiFi-GAN fwd/dgrad)
PEAK_TFLOPS_16BIT = 2500.0             # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0                 # HBM3E (MI355X_MICROARCH.md)
MEL_PATH_BYTES_PER_CLIP_STEP = 2.4e6   # fused-ideal fp32 traffic of STFT + mel + loss, forward + backward


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
    "diffmusic_style_audioldm2": ("audioldm2", "diffmusic", 1.0, 0.08, "style_guidance", 8, "configs[4] (500-step schedule)"),
}
WORKLOAD_STEPS = {"diffmusic_style_audioldm2": 500}


def build_problem(B, rank, device, workload="dps_inpainting", clip_ids=None):
    """Pipeline + operator + measurement + latents + conditioning of `B` clips.  `clip_ids` (default rank*B ... rank*B + B - 1) are the
    GLOBAL clip numbers: clip k's audio, latent noise and generator depend on k only, never on how the clips are split over ranks."""
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
    elif task == "style_guidance":
        op = P.StyleGuidanceOperator(SR, noiser=noiser, device=device)
    else:
        op = P.SuperResolutionOperator(SR, 4, noiser=noiser)
    pipe.scheduler = get_scheduler(sname)(operator=op, **SCHED_CFG)
    pipe.scheduler.set_timesteps(WORKLOAD_STEPS.get(workload, N_STEPS), device=device)
    L = SECONDS * SR
    if clip_ids is None:
        clip_ids = [rank * B + i for i in range(B)]
    assert len(clip_ids) == B
    clips = torch.stack([synth_clip(k, L) for k in clip_ids]).to(device)
    measurement = op.forward(clips)
    gens = [torch.Generator().manual_seed(k) for k in clip_ids]
    latents = randn_tensor((B, 8, 250, 16), generator=gens, device=device, dtype=torch.float32)
    # conditioning rows are keyed by the global clip number too (row k of a seeded draw), so a clip's trajectory does not depend on the split
    n_all = max(clip_ids) + 1
    g7, g8 = torch.Generator().manual_seed(7), torch.Generator().manual_seed(8)
    if pname == "musicldm":
        pe = torch.nn.functional.normalize(torch.randn(n_all, 512, generator=g7), dim=-1)[clip_ids].to(device)
        cond = dict(class_labels=torch.cat([pe, pe], dim=0))      # prompt="" in the reference: cond == uncond, CFG batch kept at 2B
        gscale = GUIDANCE_SCALE
    else:                                                          # AudioLDM2: GPT-2 states (B,8,768), T5 states (B,16,1024), mask ones
        ge = torch.randn(n_all, 8, 768, generator=g7)[clip_ids].to(device)
        te = torch.randn(n_all, 16, 1024, generator=g8)[clip_ids].to(device)
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


class Stepper:
    """Advances the B clips of this rank through consecutive steps of the schedule: the plain loop (lanes = 1: U-Net, then the guided
    step, for the whole batch, NaN test of the loss per step) or the product's clip lanes (diffmusic_amd/pipelines/lanes.py: the batch
    cut into `lanes` contiguous groups whose U-Net forwards run under each other's guidance sweeps; same per-step NaN test, per lane)."""

    def __init__(self, pipe, latents, cond, measurement, L, lanes, nan_check):
        self.pipe, self.cond, self.measurement, self.L, self.nan_check = pipe, cond, measurement, L, nan_check
        self.B = latents.shape[0]
        self.n_lanes = max(1, min(int(lanes), self.B))
        self.latents_plain = latents
        self.lanes = None
        if self.n_lanes > 1:
            from diffmusic_amd.pipelines.lanes import Lane, LaneRunner, split_sizes
            self.runner = LaneRunner(latents.device)
            gens = pipe._bench["gens"]
            self.lanes, o = [], 0
            for n in split_sizes(self.B, self.n_lanes):
                ids = list(range(o, o + n))
                o += n
                rows = ids + [self.B + k for k in ids]
                c = {k: (v[rows].contiguous() if v is not None else None) for k, v in cond.items()}
                m = measurement[ids].contiguous() if measurement.shape[0] == self.B and self.B > 1 else measurement
                self.lanes.append(Lane(ids, latents[ids].contiguous(), c, m, [gens[k] for k in ids]))

    @property
    def latents(self):
        return self.latents_plain if self.lanes is None else torch.cat([ln.latents for ln in self.lanes], dim=0)

    def set_latents(self, latents):
        if self.lanes is None:
            self.latents_plain = latents
        else:
            for ln in self.lanes:
                ln.latents = latents[ln.ids].contiguous()

    def advance(self, ts):
        """One step per entry of `ts` (host timesteps).  Returns (loss of the last step over all clips, number of steps with a NaN loss)."""
        pipe, b = self.pipe, self.pipe._bench
        if self.lanes is None:
            nan_steps, loss = 0, None
            for t in ts:
                self.latents_plain, loss = one_step(pipe, self.latents_plain, t, self.cond, self.measurement, self.L)
                if self.nan_check:        # the loop body's NaN test (pipeline_musicldm.py:741-742): one host sync per step
                    nan_steps += int(bool(torch.isnan(loss).any()))
            return loss, nan_steps

        def unet_fn(ln, i):
            return pipe._unet_eps(ln.latents, ts[i], ln.cond, b["gscale"], True)

        def step_fn(ln, i, eps):
            out = pipe.scheduler.step(eps, ts[i], ln.latents, eta=b["eta"], generator=ln.generator, measurement=ln.measurement, vae=pipe.vae,
                                      vocoder=pipe.vocoder, original_waveform_length=self.L, ip_guidance_rate=b["rate"],
                                      supervised_space="mel_spectrogram")
            return out.prev_sample, out.loss
        bad = self.runner.run(self.lanes, len(ts), unet_fn, step_fn, 1 if self.nan_check else 10 ** 9)
        loss = torch.cat([ln.losses[-1].reshape(-1) for ln in self.lanes])
        return loss, int(bad is not None)


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """CPUs this process may actually use: (host logical CPUs, affinity-set size, cgroup CPU quota or None).  On the GPU boxes of this
    pool the affinity set is the whole host (256) while the container's cgroup grants 16 CPUs: threads beyond the quota only add
    scheduling overhead (measured: the oracle did not finish a step in 7 minutes on 256 threads), so the quota is the share."""
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                          # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                                                 # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    return os.cpu_count(), len(os.sched_getaffinity(0)), quota


def cpu_baseline(seed_sd, threads, batch, n_meas):
    """The CPU restatement (oracle/, fp32 eager torch + autograd) timed on the host cores: `batch` clips, DPS steps.
    Returns seconds per (batch-`batch`) step over `n_meas` measured steps after 1 warm-up step."""
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
    y = op.forward(torch.stack([synth_clip(i, L) for i in range(batch)]))
    x = torch.cat([torch.randn(1, 8, 250, 16, generator=torch.Generator().manual_seed(i)) for i in range(batch)])
    pe = torch.nn.functional.normalize(torch.randn(batch, 512, generator=torch.Generator().manual_seed(7)), dim=-1)
    ts = [int(t) for t in sched.timesteps]
    n_warm = 1
    t0 = time.perf_counter()
    for i in range(n_warm + n_meas):
        if i == n_warm:
            t0 = time.perf_counter()
        with torch.no_grad():
            e2 = unet(torch.cat([x, x]), ts[i], class_labels=torch.cat([pe, pe]))[0]
        e = e2[:batch] + GUIDANCE_SCALE * (e2[batch:] - e2[:batch])
        x = sched.step(e, ts[i], x, eta=0.0, measurement=y, vae=vae, vocoder=voc, original_waveform_length=L,
                       ip_guidance_rate=ZETA, supervised_space="mel_spectrogram").prev_sample
        print(f"[bench]   CPU oracle batch {batch}: step {i + 1}/{n_warm + n_meas} done", file=sys.stderr, flush=True)
    return (time.perf_counter() - t0) / n_meas


def pmc_traffic(prefix):
    """HBM-side bytes per launch of one kernel family (its 8-wave tiles, >= 192 rows) from the newest committed counter passes
    (profiles/rNN_pmc_{FETCH,WRITE}_SIZE_per_kernel.csv: separate `rocprofv3 --pmc` runs of this same command,
    KiB -> bytes, FETCH_SIZE doubled per the gfx950 correction).  Counters cannot be read from inside the process, so this
    is the measurement of the profiled run named in `traffic_source`, not of this run."""
    import csv
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    tags = sorted({os.path.basename(p).split("_pmc_")[0] for p in glob.glob(os.path.join(here, "profiles", "r*_pmc_FETCH_SIZE_per_kernel.csv"))})
    if not tags:
        return None, None
    tag = tags[-1]
    tot, n = 0.0, 0
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            with open(os.path.join(here, "profiles", f"{tag}_pmc_{c}_per_kernel.csv")) as fh:
                rows = [r for r in csv.DictReader(fh) if r["kernel"].startswith(prefix + "<") and int(r["kernel"].split("<")[1].split(",")[0]) >= 192]
            tot += sum(float(r[f"{c}_bytes_total"]) for r in rows)
            n = sum(int(r["launches"]) for r in rows)
    except (OSError, KeyError):
        return None, None
    src = f"profiles/{tag}_pmc_{{FETCH,WRITE}}_SIZE_per_kernel.csv (rocprofv3 --pmc, separate passes of `python bench.py`"
    meta = os.path.join(here, "profiles", f"{tag}_pmc_meta.json")
    if os.path.exists(meta):
        with open(meta) as fh:
            src += ", source tree " + json.load(fh).get("source_sha16", "?")
    return (round(tot / n) if n else None), src + ")"


def visible_gpus():
    """Number of GPUs this process may use, WITHOUT bringing up the HIP runtime in the parent (its children must be the first
    processes to touch the GPU): KFD topology nodes that have SIMDs, cut down by HIP_/ROCR_VISIBLE_DEVICES lists."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for d in os.listdir(base):
            with open(os.path.join(base, d, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except OSError:
        return None                                       # unknown here: let the ranks report a shortfall themselves
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n, argv, need_gpus, timeout_s=3300.0):
    """--gpus N without a launcher: N fresh rank processes (this parent never touches the GPU), RCCL rendezvous on 127.0.0.1.
    The children are polled: when one exits non-zero (or the deadline passes) the survivors are terminated, so nobody is left
    waiting in a rendezvous or barrier; a rendezvous port that another process grabbed first is retried on a new port."""
    import socket
    have = visible_gpus()
    if have is not None and have < need_gpus:
        print(f"[bench] --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 3
    for attempt in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), DMX_BENCH_SPAWNED="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
        deadline = time.monotonic() + timeout_s
        rcs = [None] * n
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
            failed = [rc for rc in rcs if rc not in (None, 0)]
            if failed and rcs[0] is None and 98 not in failed:
                # a client rank failed first (e.g. it reached a foreign listener on a stolen port): rank 0's own verdict -- exit code 98
                # for EADDRINUSE -- decides whether this attempt is retried, so give it a few seconds to arrive
                try:
                    rcs[0] = procs[0].wait(timeout=5)
                except subprocess.TimeoutExpired:
                    pass
                failed = [rc for rc in rcs if rc not in (None, 0)]
            if failed or time.monotonic() > deadline:
                for r, p in enumerate(procs):              # exactly the PIDs started above
                    if rcs[r] is None:
                        p.terminate()
                for r, p in enumerate(procs):
                    if rcs[r] is None:
                        try:
                            rcs[r] = p.wait(timeout=15)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            rcs[r] = p.wait()
                if not failed:
                    print(f"[bench] ranks did not finish within {timeout_s:.0f} s", file=sys.stderr)
                    return 5
                break
            time.sleep(0.2)
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
        if not bad:
            return 0
        # only rendezvous failures are retried (exit codes 98 = EADDRINUSE, 97 = init_process_group raised; see main): any other early
        # exit -- an import error, a bad argument -- is deterministic and is reported once
        if any(rc in (97, 98) for _, rc in bad) and attempt < 2:
            print(f"[bench] rendezvous on port {port} failed (exit codes {bad}); retrying on another port", file=sys.stderr)
            continue
        print(f"[bench] ranks failed: {bad}", file=sys.stderr)
        return max(abs(rc) for _, rc in bad)
    return 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--settle", type=int, default=10, help="throw-away conditioning steps (full loop body, on a copy of the latents) before the "
                    "W warm-up steps; 0 = none.  They absorb one-time costs that would otherwise land in a short timed region whatever W is: "
                    "first-call initialisation (~0.2-0.4 s) and the lazy load of the torch kernels of the NaN check (~30 ms); DESIGN.md section 4")
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (weak scaling: per-GPU work fixed as N grows); default = the workload's "
                    "own batch (8 for the headline config)")
    ap.add_argument("--global-batch", type=int, default=0, help="STRONG scaling: G clips in total, split over the --gpus N ranks (clip k -> rank "
                    "k mod N as in Pipeline.__call__(shard=True)); valid at N = 1 too, e.g. --global-batch 32 --gpus 1 is the one-GPU number that "
                    "BASELINE.json configs[2] (32 clips over 8 GPUs) is divided by")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU leg (0 = this process's CPU share: the cgroup CPU quota where one is set, "
                    "else the affinity set, capped at --cpu-share-cap on hosts whose affinity set exceeds 32 CPUs)")
    ap.add_argument("--cpu-share-cap", type=int, default=16, help="assumed per-GPU CPU share on a many-core host without a visible cgroup quota "
                    "(this pool grants 16 CPUs per GPU)")
    ap.add_argument("--no-full-trajectory", action="store_true", help="skip the end-to-end leg (all 200 steps + final decode after the K timed steps)")
    ap.add_argument("--cpu-batch", type=int, default=0, help="batch of the second CPU leg (0 = 8 if host memory allows, else 4 / 2 / none)")
    ap.add_argument("--no-nan-check", action="store_true", help="skip the per-step host-side NaN test of the loss (the reference loop has it)")
    ap.add_argument("--no-stage-times", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (default); gloo only to rehearse the "
                    "multi-rank path on a box with fewer GPUs than ranks (with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (gloo backend only)")
    ap.add_argument("--lanes", type=int, default=DEFAULT_LANES, help="clip lanes per GPU (diffmusic_amd/pipelines/lanes.py): the per-GPU batch is cut "
                    "into this many groups and each group's U-Net forward runs under the other groups' guidance sweeps; 1 = the plain loop "
                    "(U-Net, then the guided step, for the whole batch)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (and time the final gather through it) even at "
                    "--gpus 1: the RCCL path at world size 1, in a fresh child process like every spawned rank")
    ap.add_argument("--workload", default="dps_inpainting", choices=sorted(WORKLOADS),
                    help="default = the headline config (BASELINE.json configs[1]); the others are the remaining GPU configs")
    args = ap.parse_args()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if (args.gpus > 1 or args.force_dist) and env_world == 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], 1 if args.share_gpu else args.gpus))
    world, rank, local = env_world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(3)
    import torch.distributed as dist
    if args.share_gpu:
        if args.backend != "gloo":
            print("[bench] --share-gpu is a rehearsal mode and needs --backend gloo (RCCL wants one device per rank)", file=sys.stderr)
            sys.exit(3)
        local = 0
    use_dist = world > 1 or args.force_dist
    if use_dist:
        torch.cuda.set_device(local)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))   # backend "nccl" is RCCL on ROCm
            else:
                dist.init_process_group("gloo")
        except Exception as e:                                   # noqa: BLE001
            if "address already in use" in str(e).lower() or "EADDRINUSE" in str(e):
                sys.exit(98)                                     # spawn_ranks retries on another port
            import traceback
            traceback.print_exc()
            sys.exit(97)                                         # the rendezvous itself failed (e.g. a foreign listener on a stolen port): retried too
        if dist.get_world_size() != args.gpus:
            print(f"[bench] only {dist.get_world_size()} of {args.gpus} ranks joined", file=sys.stderr)
            sys.exit(3)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    wl = args.workload
    strong = args.global_batch > 0
    if strong:
        if args.batch:
            print("[bench] --batch (weak scaling) and --global-batch (strong scaling) are exclusive", file=sys.stderr)
            sys.exit(3)
        if args.global_batch < world:
            print(f"[bench] --global-batch {args.global_batch} < {world} ranks: every rank needs at least one clip", file=sys.stderr)
            sys.exit(3)
        clip_ids = list(range(rank, args.global_batch, world))    # clip k -> rank k mod N (diffmusic_amd/parallel.py shard_indices)
        G = args.global_batch
    else:
        Bw = args.batch or WORKLOADS[wl][5]
        clip_ids = list(range(rank * Bw, (rank + 1) * Bw))
        G = Bw * world
    B = len(clip_ids)
    n_sched = WORKLOAD_STEPS.get(wl, N_STEPS)
    pipe, op, measurement, latents, pe2, L = build_problem(B, rank, device, wl, clip_ids)
    ts = pipe.scheduler._timesteps_host
    latents0 = latents.clone()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- conditioning, outside the contract's W warm-up + K timed steps: the full loop body on a throw-away copy of the latents, so that
    # one-time costs cannot land in the timed region whatever W is.  Measured on fresh boxes: the first step of a process takes 0.2-0.4 s
    # (library initialisation), and the first torch.isnan(...).any() of a process loads torch's elementwise kernels lazily (~30 ms: it
    # used to fall into the timed loop and read as "the first process on a box is 3-6 ms/step slower", 44.8 -> 41.0 ms over 10 steps).
    # The trajectory that is warmed up and timed below starts from the untouched latents.
    nan_check = not args.no_nan_check
    n_lanes = max(1, min(args.lanes, B))
    stepper = Stepper(pipe, latents, pe2, measurement, L, n_lanes, nan_check)
    settle_ms = []
    if args.settle > 0:
        se0, se1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(args.settle):
            se0.record()
            stepper.advance([ts[i % n_sched]])          # the whole loop body, host NaN check included (its torch kernels load lazily:
            se1.record()                                # measured ~30 ms once per process on a fresh box)
            torch.cuda.synchronize()
            settle_ms.append(se0.elapsed_time(se1))
        stepper.set_latents(latents0)                   # the trajectory that is warmed up and timed below starts from the untouched latents
    k = 0
    if args.warmup > 0:                      # the same body as the timed loop below
        stepper.advance([ts[(k + j) % n_sched] for j in range(args.warmup)])
        k += args.warmup
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    loss, nan_steps = stepper.advance([ts[(k + j) % n_sched] for j in range(args.steps)])
    k += args.steps
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    tmax = torch.tensor([wall], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    latents = stepper.latents
    finite = bool(torch.isfinite(loss).all()) and bool(torch.isfinite(latents).all())

    # ---- after the loop: final decode of this rank's clips and the path's only collective, one all_gather of (B, L) waveforms
    from diffmusic_amd import parallel
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    mel = pipe.vae.decode(latents / pipe.vae.config.scaling_factor).sample
    audio = pipe.vocoder(mel.squeeze(1))[:, :L].float().contiguous()
    e1.record()
    barrier()
    tg = time.perf_counter()
    gathered = parallel.gather_waveforms(audio, G)
    torch.cuda.synchronize()
    gather_ms = 1e3 * (time.perf_counter() - tg)
    assert gathered.shape == (G, L)
    gathered_equals_local = bool(torch.equal(gathered[parallel.shard_indices(G, rank, world)].to(audio.device), audio))   # this rank's rows came back bit-equal
    finite = finite and bool(torch.isfinite(gathered).all())
    final_decode_ms = e0.elapsed_time(e1)

    # ---- per-stage device times (HIP events around each stage, 2 extra steps) + the STFT / mel sub-path's achieved HBM rate
    stages = mel_path = None
    if not args.no_stage_times:
        from diffmusic_amd import profiling
        profiling.enable(events=True)
        for _ in range(2):                  # the plain loop body on the whole batch: stage times without overlap between stages
            one_step(pipe, latents, ts[k % n_sched], pe2, measurement, L)
        stages = {kk: round(v, 3) for kk, v in profiling.stage_ms().items()}
        profiling.enable(events=False)
        mel_ms = stages.get("operator_mel_loss_fwd_bwd")
        if mel_ms:
            by = MEL_PATH_BYTES_PER_CLIP_STEP * B
            mel_path = {"algorithmic_bytes": int(by), "stage_ms": mel_ms, "achieved_GBps": round(by / (mel_ms * 1e-3) / 1e9, 2),
                        "peak_GBps": PEAK_HBM_GBPS, "frac": round(by / (mel_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 5),
                        "note": "operator + transform (STFT, mel, dB) + L2 + hand-written backward + per-clip gradient rescale, HIP events around the "
                                "stage; algorithmic bytes = fused-ideal 2.4 MB per clip-step (SURVEY.md section 8d); per-kernel times in profiles/"}

    # ---- roofline leg: one extra step with HIP events around every implicit-GEMM launch
    import ctypes as C
    from diffmusic_amd import _lib as Lb
    Lb.lib().dmx_prof_begin()
    stepper.advance([ts[k % n_sched]])       # as timed above: with clip lanes the launches have the lanes' shapes and run under overlap
    ms, fl = C.c_double(), C.c_double()
    n_launch = Lb.lib().dmx_prof_end(C.byref(ms), C.byref(fl))
    # headline config: the analytic count of BASELINE.md; other workloads: the FLOPs the launches actually issued
    algo_tflop_step = ALGO_TFLOP_PER_CLIP_STEP * B if wl == "dps_inpainting" else fl.value / 1e12
    dms, dfl, dby = C.c_double(), C.c_double(), C.c_double()
    n_dom = Lb.lib().dmx_prof_dominant(C.byref(dms), C.byref(dfl), C.byref(dby))
    traffic, traffic_src = pmc_traffic("gemm_glds_kernel") if args.workload == "dps_inpainting" else (None, None)
    # dominant kernel = gemm_glds_kernel (LDS-DMA tiles): its issued FLOPs, scaled DOWN by algorithmic/issued of the step where the launches
    # issue more than the analytic count (channel / tile padding) and never up: where the build does the reference's arithmetic in fewer
    # multiply-adds (the x2 upsampling folded into the VAE's 3x3 convolutions: 4/9 of the reference's) the kernel is credited with what it
    # issued, and only `whole_step` (reference work per step / wall time) carries the saving
    scale = min(1.0, algo_tflop_step / (fl.value / 1e12)) if fl.value > 0 else 1.0
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
                "whole_step": {"achieved": round(algo_tflop_step / (wall / args.steps), 1), "frac": round(algo_tflop_step / (wall / args.steps) / PEAK_TFLOPS_16BIT, 4)},
                "step_share": round(ms.value / (1e3 * wall / args.steps), 3)}

    # ---- end-to-end leg: the whole trajectory (all n_sched steps of the loop body, NaN check included) + final decode, from fresh
    # latents -- what run.py:317-332 sees per call, and several seconds of GPU time that an outside sampler can see
    full_traj = None
    if not args.no_full_trajectory:
        stepper.set_latents(latents0.clone())
        barrier()
        tf = time.perf_counter()
        loss_f, _ = stepper.advance(list(ts[:n_sched]))
        torch.cuda.synchronize()
        t_loop = time.perf_counter() - tf
        lat_f = stepper.latents
        mel_f = pipe.vae.decode(lat_f / pipe.vae.config.scaling_factor).sample
        audio_f = pipe.vocoder(mel_f.squeeze(1))[:, :L].float()
        torch.cuda.synchronize()
        t_all = time.perf_counter() - tf
        tm = torch.tensor([t_loop, t_all], dtype=torch.float64, device=device)
        if use_dist:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        full_traj = {"steps": n_sched, "loop_wall_s": round(float(tm[0]), 4), "wall_s_with_final_decode": round(float(tm[1]), 4),
                     "steps_per_sec": round(n_sched / float(tm[0]) * (1 if strong else world), 4),
                     "finite": bool(torch.isfinite(audio_f).all()) and bool(torch.isfinite(loss_f).all()),
                     "final_loss_clip0": float(loss_f.reshape(-1)[0]),
                     "note": "all steps of the schedule from fresh latents + final VAE / vocoder decode, host wall clock, max over ranks; not `value`"}
        del lat_f, mel_f, audio_f

    rc = 0
    if rank == 0:
        # weak scaling (default): every rank advances its own B clips per step, so the job does `world` batch-B steps per loop pass
        # and value = K / wall * world (steps of B clips).  Strong scaling (--global-batch G): one step advances all G clips of the
        # job, value = K / wall (steps of G clips).  clip_steps_per_sec = G * K / wall in both conventions.
        steps_per_s = args.steps / wall * (1 if strong else world)
        clip_steps = G * args.steps / wall
        base = {"dps_inpainting": "MusicLDM + DPS music_inpainting, 10 s @16 kHz, 200-step schedule",
                }.get(wl, f"{wl}: {WORKLOADS[wl][0]} + {WORKLOADS[wl][1]} {WORKLOADS[wl][4]}, 10 s @16 kHz, {n_sched}-step schedule")
        is_headline = wl == "dps_inpainting" and not strong and B == 8
        wl_text = (f"{base}, {G} clips in total split over {world} GPU(s) ({B} on rank 0)" if strong else f"{base}, batch {B} per GPU") + \
                  (" (BASELINE.json configs[1])" if is_headline else f" ({WORKLOADS[wl][6]})" if wl != "dps_inpainting" else " (configs[1] at another batch)")
        unit = (f"steps/s (strong scaling: one step advances all {G} clips of the job; value = K / wall)" if strong else
                f"steps/s (weak scaling: one step advances a batch of {B} clips on one GPU; value = K / wall x {world} GPU(s))")
        if n_lanes > 1:
            unit += f"; the {B} clips of a GPU run as {n_lanes} clip lanes of {'+'.join(str(len(ln.ids)) for ln in stepper.lanes)}, staggered"
        res = {"metric": "denoising steps/sec (10 s clip, 200-step DPS)", "value": round(steps_per_s, 4) if finite else None,
               "unit": unit, "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * wall / args.steps, 3), "higher_is_better": True,
               "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": wl_text, "global_batch": G, "clips_per_gpu": B, "lanes": n_lanes,
                          "clip_steps_per_sec": round(clip_steps, 3), "parallelism": f"clip-sharded x{world} ({'RCCL' if args.backend == 'nccl' else 'gloo REHEARSAL'} world size {world}, no per-step collective)",
                          "device_ms_per_step": round(dev_ms / args.steps, 3), "finite": finite, "nan_steps": nan_steps,
                          "nan_check_per_step": not args.no_nan_check,
                          "settle_steps": len(settle_ms), "settle_first3_ms": [round(v, 2) for v in settle_ms[:3]], "settle_last3_ms": [round(v, 2) for v in settle_ms[-3:]], "final_loss_clip0": float(loss.reshape(-1)[0]),
                          "cached_reference_transform": bool(getattr(op, "cache_reference", False)),
                          "launched_by": "bench.py spawn" if os.environ.get("DMX_BENCH_SPAWNED") else ("torch.distributed.run" if world > 1 else "single process")},
               "after_loop": {"final_decode_ms": round(final_decode_ms, 3), "gather_ms": round(gather_ms, 3) if use_dist else 0.0,
                              "gather_bytes_per_rank": int(audio.numel() * 4), "gather_world_size": world,
                              "collective": (f"all_gather of (clips_per_gpu, 160000) fp32 waveforms, once per call ({'RCCL' if args.backend == 'nccl' else 'gloo'}, world size {world})"
                                             if use_dist else "none (single rank, torch.distributed not initialised)"),
                              "gathered_equals_local": gathered_equals_local},
               "full_trajectory": full_traj, "stage_ms": stages, "mel_path": mel_path, "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline and wl == "dps_inpainting" and not strong:
            host_cpus, affinity, quota = cpu_share()
            # the whole CPU share of this process: the cgroup quota where one is set; else the affinity set -- unless that is a whole
            # many-core host (> 32), where this pool's documented per-GPU share of 16 is assumed (see cpu_share)
            share = min(affinity, max(1, int(quota + 0.5))) if quota else (affinity if affinity <= 32 else max(1, args.cpu_share_cap))
            threads = share if args.cpu_threads <= 0 else min(args.cpu_threads, affinity)
            sd = {"unet": pipe.unet.synth_state_dict(0), "vae": pipe.vae.synth_state_dict(1), "vocoder": pipe.vocoder.synth_state_dict(2)}
            print(f"[bench] timing the CPU oracle on {threads} threads (host has {host_cpus} logical CPUs, affinity {affinity}, cgroup CPU quota "
                  f"{quota}): 1 clip x (1 warm-up + 3 measured) steps ...", file=sys.stderr, flush=True)
            sec1 = cpu_baseline(sd, threads, 1, 3)
            print(f"[bench] CPU oracle: {sec1:.2f} s per clip-step", file=sys.stderr, flush=True)
            cb = args.cpu_batch
            if cb == 0:
                try:
                    import psutil
                    avail = psutil.virtual_memory().available / 2 ** 30
                except Exception:
                    avail = 0.0
                cb = next((b for b in (8, 4, 2) if 12.0 * b + 8.0 < 0.6 * avail), 1)      # ~12 GiB of autograd state per clip
            secb = None
            n_b = 2
            if cb > 1:
                print(f"[bench] timing the CPU oracle at batch {cb}: 1 warm-up + {n_b} measured steps ...", file=sys.stderr, flush=True)
                secb = cpu_baseline(sd, threads, cb, n_b)
                print(f"[bench] CPU oracle: {secb:.2f} s per batch-{cb} step", file=sys.stderr, flush=True)
            # the best the CPU can do: the better of the measured batched leg and 8 x the batch-1 leg (the batched oracle holds 8 autograd
            # tapes and is memory-bound: slower per clip than batch 1 on these hosts); both raw legs are reported next to it
            vb = 1.0 / (secb * (8.0 / cb)) if secb else 0.0
            v1 = 1.0 / (sec1 * 8)
            v8 = max(vb, v1)
            res["cpu_baseline"] = {"value": round(v8, 6),
                                   "unit": "steps/s (batch-8 step; " + (f"measured at batch {cb}" if vb >= v1 else "8 x the batch-1 rate, the faster of the two legs") + ")",
                                   "value_rule": "max(batched leg, 8 x batch-1 leg)", f"batch{cb}_steps_per_sec": round(vb, 6) if secb else None,
                                   "cores": threads, "host_logical_cpus": host_cpus, "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
                                   "threads_used": threads, "threads_rule": "cgroup quota" if quota else ("affinity set" if affinity <= 32 else f"{max(1, args.cpu_share_cap)} (--cpu-share-cap: affinity set is the whole host, no cgroup quota visible)"),
                                   "kind": "port", "cpu_model": cpu_model_string(), "torch": torch.__version__,
                                   "batch1_seconds_per_step": round(sec1, 3), f"batch{cb}_seconds_per_step": round(secb, 3) if secb else None,
                                   "batch1_x8_steps_per_sec": round(1.0 / (sec1 * 8), 6),
                                   "gpu_over_cpu": round(clip_steps / (8 * v8), 1),
                                   "sample": f"oracle/ (fp32 eager torch + autograd, U-Net 2x fwd + guided DPS step), warmed: 1 clip x 3 measured steps after 1 "
                                             f"warm-up ({sec1:.2f} s/step)" + (f"; {cb} clips x {n_b} measured steps after 1 warm-up ({secb:.2f} s/step)" if secb else "")}
        print(json.dumps(res), flush=True)
        if not finite:
            print(f"[bench] non-finite loss / latents / waveforms (nan_steps={nan_steps}): value set to null", file=sys.stderr)
            rc = 4
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
