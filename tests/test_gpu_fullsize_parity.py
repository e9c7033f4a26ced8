"""-m gpu: teacher-forced HIP-vs-oracle parity at PRODUCTION size and width: one 10 s clip (latent 8 x 250 x 16, mel 1000 x 64,
160 032 samples), the benchmark architectures (HIFIGAN_DEFAULT / VAE_DEFAULT / UNET_*_DEFAULT), one test per GPU config of
BASELINE.json: configs[1] MusicLDM + DPS inpainting, configs[2] AudioLDM2 + DSG phase retrieval, configs[3] MusicLDM + MPGD
SR x4, configs[4] AudioLDM2 + DiffMusic style guidance (CLAP Gram loss, build-defined semantics).  The reference step being matched is diffmusic/schedulers/scheduling_dps.py:137-219 (and siblings), restated in
oracle/schedulers.py on fp32 eager torch + autograd (about 2 s per clip-step on the GPU box's 16 host threads).

Every stage is compared on the SAME input as the oracle stage (teacher-forced per stage) and the whole step once more end to
end; the per-stage table is printed and written to gpurun_out/fullsize_parity_<workload>.json.
Tolerances: rel-L2(prev_sample) <= 1e-2 and loss rel err <= 1e-2 on the fp16 MFMA path."""
import json
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT)


def _rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _cos(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))


def _oracle_nets(pipe, wl):
    from oracle import models as OM
    if "audioldm2" in wl:
        ru = OM.UNetMusicLDM(class_embed_dim=0, attn_cross_dims=(None, 768, 1024)).eval()
    else:
        ru = OM.UNetMusicLDM().eval()
    rv, rh = OM.VaeDecoder().eval(), OM.HifiGan().eval()
    ru.load_state_dict(pipe.unet.synth_state_dict(0), strict=True)
    rv.load_state_dict(pipe.vae.synth_state_dict(1), strict=True)
    rh.load_state_dict(pipe.vocoder.synth_state_dict(2), strict=False)
    return ru, rv, rh


def _oracle_op(task, op=None):
    import copy
    from oracle import operators as OO
    n = OO.get_noiser("gaussian", 0.0)
    if task == "style_guidance":                          # same (seeded random) HTS-AT weights as the HIP operator's tower
        return OO.StyleGuidanceOperator(16000, clap_model=copy.deepcopy(op.clap).cpu().float().eval(), noiser=n)
    if task == "music_inpainting":
        return OO.MusicInpaintingOperator(10, 16000, "box", 2, 3, 0.3, 0.1, 1.0, noiser=n)
    if task == "phase_retrieval":
        return OO.PhaseRetrievalOperator(noiser=n)
    return OO.SuperResolutionOperator(16000, 4, noiser=n)


# workload -> index of the mid-trajectory timestep; every workload is also checked at the FIRST step of its schedule (t = 996 /
# 999: alpha_bar = 1.5e-4, x0_hat = (x - sqrt(1 - a) eps) / sqrt(a) amplifies by 81x -- the fp16 range / NaN-retry trigger of
# pipeline_musicldm.py:741-756) and at the LAST one (t = 1: alpha_bar_prev = final_alpha_cumprod)
CASES = {"dps_inpainting": 60, "dsg_phase_audioldm2": 100, "mpgd_sr4": 140, "diffmusic_style_audioldm2": 250}
WHICH = {"first": lambda wl: 0, "mid": lambda wl: CASES[wl], "last": lambda wl: -1}


@pytest.mark.parametrize("which", ["mid", "first", "last"])
@pytest.mark.parametrize("wl", sorted(CASES))
def test_fullsize_teacher_forced_step(wl, which):
    import bench
    from diffmusic_amd import _lib as Lb
    from oracle import schedulers as OS
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    dev = torch.device("cuda")
    pname, sname, eta, rate, task, _, _ = bench.WORKLOADS[wl]
    pipe, op, meas, lat, cond, L = bench.build_problem(1, 0, dev, wl)
    gscale = pipe._bench["gscale"]
    sched = pipe.scheduler
    t = sched._timesteps_host[WHICH[which](wl)]
    ru, rv, rh = _oracle_nets(pipe, wl)
    rop = _oracle_op(task, op)
    rs = OS.get_scheduler(sname)(operator=rop, **bench.SCHED_CFG)
    rs.set_timesteps(bench.WORKLOAD_STEPS.get(wl, bench.N_STEPS))
    rep = {"workload": wl, "which": which, "timestep": t}

    # ---- measurement operator on the same clip
    clip = bench.synth_clip(0, L)[None]
    y_ref = rop.forward(clip)
    rep["operator_forward"] = _rel(meas, y_ref)

    # ---- stage 1: U-Net on the 2B CFG batch + combine (pipeline_musicldm.py:692-708)
    x = lat.cpu().float()
    eps_hip = pipe._unet_eps(lat, t, cond, gscale, True)
    with torch.no_grad():
        kw = {k: (v.cpu() if v is not None else None) for k, v in cond.items()}
        e2 = ru(torch.cat([x, x]), t, **kw)[0]
    eps_ref = e2[:1] + gscale * (e2[1:] - e2[:1])
    rep["unet_eps"] = _rel(eps_hip, eps_ref)

    # ---- oracle chain with every intermediate gradient kept (scheduling_dps.py:195-212)
    a_t = float(rs.alphas_cumprod[t])
    x0 = ((x - (1 - a_t) ** 0.5 * eps_ref) / a_t ** 0.5).detach()
    sf = rv.config.scaling_factor
    x0r = x0.clone().requires_grad_(True)
    mel_ref = rv.decode(x0r / sf).sample                       # (1,1,1000,64)
    mel_ref.retain_grad()
    wav_ref = rop.inverse_transform(mel_ref, rh)               # (1,160032)
    wav_ref.retain_grad()
    yy = rop.forward(wav_ref[:, :L])
    loss_ref = torch.linalg.norm(rop.transform(y_ref) - rop.transform(yy))
    loss_ref.backward()
    dwav_ref, dmel_ref, dx0_ref = wav_ref.grad, mel_ref.grad.squeeze(1), x0r.grad

    # ---- stage 2: VAE decode forward / backward on the oracle's x0 and the oracle's dmel
    mel16, mel32 = pipe.vae.decode_hip(x0.to(dev).contiguous(), z_scale=1.0 / sf, keep_state=True, want_f32=True)
    rep["vae_mel"] = _rel(mel32, mel_ref.squeeze(1))
    s_m = 64.0 / float(dmel_ref.abs().max())
    dx0_hip = pipe.vae.backward((dmel_ref * s_m).to(device=dev, dtype=Lb.act_dtype()).contiguous(), z_scale=1.0 / sf) / s_m
    rep["vae_bwd"] = _rel(dx0_hip, dx0_ref)
    rep["vae_bwd_cos"] = _cos(dx0_hip, dx0_ref)

    # ---- stage 3: HiFi-GAN forward / backward on the oracle's mel and the oracle's dwav
    mel_in = mel_ref.detach().squeeze(1).to(device=dev, dtype=Lb.act_dtype()).contiguous()
    wav_hip = pipe.vocoder.forward(mel_in)
    rep["vocoder_wav"] = _rel(wav_hip, wav_ref)
    s_w = 64.0 / float(dwav_ref.abs().max())
    dmel_hip = pipe.vocoder.backward((dwav_ref * s_w).to(dev).contiguous()).float() / s_w
    rep["vocoder_bwd"] = _rel(dmel_hip, dmel_ref)
    rep["vocoder_bwd_cos"] = _cos(dmel_hip, dmel_ref)

    # ---- stage 4: operator + transform + L2 and its hand-written backward on the oracle's waveform
    loss_op, dwav_hip = op.guidance(wav_ref.detach().to(dev).contiguous(), L, meas, "mel_spectrogram")
    rep["operator_loss"] = abs(float(loss_op.reshape(-1)[0]) - float(loss_ref)) / abs(float(loss_ref))
    rep["operator_bwd"] = _rel(dwav_hip, dwav_ref)

    # ---- the whole guided step end to end, teacher-forced on (x_t, eps_ref, noise)
    z = torch.randn(x.shape, generator=torch.Generator().manual_seed(123))
    kw = dict(eta=eta, ip_guidance_rate=rate, original_waveform_length=L, supervised_space="mel_spectrogram")
    nk = dict(sample_noise=z) if sname in ("dsg", "diffmusic") else dict(variance_noise=z if eta > 0 else None)
    sched.debug_keep_grad = True
    out = sched.step(eps_ref.to(dev), t, lat, measurement=meas, vae=pipe.vae, vocoder=pipe.vocoder,
                     **kw, **{k: (v.to(dev) if v is not None else None) for k, v in nk.items()})
    torch.cuda.synchronize()
    ro = rs.step(eps_ref, t, x, measurement=y_ref, vae=rv, vocoder=rh, **kw, **nk)
    rep["step_loss"] = _rel(out.loss.reshape(-1), ro.loss.reshape(-1))
    rep["step_grad"] = _rel(sched.last_grad, ro.sample)
    rep["step_grad_cos"] = _cos(sched.last_grad, ro.sample)
    rep["step_x0"] = _rel(out.pred_original_sample, ro.pred_original_sample)
    rep["step_prev_sample"] = _rel(out.prev_sample, ro.prev_sample)
    rep["loss_value"] = float(ro.loss.reshape(-1)[0])
    print("\n" + "\n".join(f"  {k:>20s}: {v:.3e}" if isinstance(v, float) else f"  {k:>20s}: {v}" for k, v in rep.items()))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"fullsize_parity_{wl}" + ("" if which == "mid" else "_" + which) + ".json"), "w") as fh:
            json.dump(rep, fh, indent=1)
    except OSError:
        pass
    assert all(math.isfinite(v) for v in rep.values() if isinstance(v, float)), rep
    assert rep["operator_forward"] < 1e-4, rep
    assert rep["unet_eps"] < 1e-2, rep
    assert rep["vae_mel"] < 1e-2, rep
    assert rep["vocoder_wav"] < 1e-2, rep
    assert rep["operator_loss"] < 2e-3 and rep["operator_bwd"] < 2e-2, rep
    assert rep["step_loss"] < 1e-2, rep
    assert rep["step_prev_sample"] < 1e-2, rep
    # input-gradients: measured 1.0-1.6e-3 (VAE), 5.0-5.7e-2 / cos 0.9984-0.9987 (vocoder: leaky-relu' mask flips of the 16-bit
    # activations), 4-8e-2 / cos 0.9968-0.9992 (whole step); asserted at measured + margin
    assert rep["vae_bwd"] < 5e-3 and rep["vae_bwd_cos"] > 0.9999, rep
    assert rep["vocoder_bwd"] < 8e-2 and rep["vocoder_bwd_cos"] > 0.995, rep
    assert rep["step_grad"] < 0.12 and rep["step_grad_cos"] > 0.99, rep
