#!/usr/bin/env python
"""End-to-end driver for one inverse problem on the MI355X engine: the output stage of SURVEY.md section 8f row 1.

Follows the flow of the reference's run.py (:147-220 operator / scheduler / pipeline wiring, :317-370 call and
outputs) on the three registries of this package.  The text encoders are out of scope, so the prompt conditioning is an
embedding file (`--prompt_embeds x.npy`, (B, 512) CLAP text embeddings for MusicLDM) or a seeded unit vector.

    python examples/run_inverse_problem.py -c dps -t music_inpainting --wav a.wav b.wav --weights /ckpt/musicldm
    python examples/run_inverse_problem.py -c mpgd -t super_resolution --num_inference_steps 20      # synthetic clips + weights
"""
import argparse
import math
import os
import sys
from pathlib import Path

import numpy as np
import scipy.io.wavfile
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diffmusic_amd.config import compose                                            # noqa: E402
from diffmusic_amd import inverse_problem as P                                      # noqa: E402
from diffmusic_amd.metrics import LogSpectralDistance, MeanSquaredError             # noqa: E402
from diffmusic_amd.pipelines import get_pipeline                                    # noqa: E402
from diffmusic_amd.schedulers import get_scheduler                                  # noqa: E402

TASKS = ("music_generation", "music_inpainting", "super_resolution", "phase_retrieval", "music_dereverberation")


def build_operator(task, cfg, mask_type):
    """run.py:157-212: one operator per task, constructor arguments from the data / model config."""
    noiser = P.get_noiser(**cfg.inverse_problem.noise)
    d, scale = cfg.data, 1
    if task == "music_generation":
        op = P.IdentityOperator(sample_rate=d.sample_rate)
    elif task == "music_inpainting":
        op = P.MusicInpaintingOperator(audio_length_in_s=cfg.model.pipe.audio_length_in_s, sample_rate=d.sample_rate, mask_type=mask_type,
                                       start_inpainting_s=d.start_inpainting_s - d.start_s, end_inpainting_s=d.end_inpainting_s - d.start_s,
                                       mask_percentage=0.3, interval_s=1, mask_duration_s=0.1, noiser=noiser)
    elif task == "super_resolution":
        scale = 2
        op = P.SuperResolutionOperator(sample_rate=d.sample_rate, scale=scale, noiser=noiser)
    elif task == "phase_retrieval":
        op = P.PhaseRetrievalOperator(n_fft=d.n_fft, hop_length=d.hop_length, win_length=d.win_length, noiser=noiser)
    elif task == "music_dereverberation":
        op = P.MusicDereverberationOperator(ir_length=5000, decay_factor=0.99, noiser=noiser)
    else:
        raise ValueError(f"Unknown task: {task}")
    return op, scale


def load_clips(paths, n, sr, length, seed, start_s=0.0):
    """(B, length) fp32 in [-1, 1]: wav files decoded, mixed down to mono and resampled to `sr` by the dataset loader
    (diffmusic_amd/data/dataloader.py; reference dataloader.py:47-89), cropped from `start_s` / zero-padded to `length`;
    seeded synthetic chords fill up to `n`."""
    from diffmusic_amd.data.dataloader import load_wav
    from diffmusic_amd.pipelines.prompt_audioldm2 import resample_to
    clips = []
    for p in paths:
        x, rate = load_wav(p)
        x = x.mean(dim=0, keepdim=True)
        if rate != sr:
            x = resample_to(x, rate, sr)
        x = x[0, int(start_s * sr):][:length]
        clips.append(torch.nn.functional.pad(x, (0, max(0, length - x.numel()))))
    g = torch.Generator().manual_seed(seed)
    while len(clips) < n:
        f = 110.0 * 2 ** (torch.randint(0, 36, (4,), generator=g).float() / 12)
        t = torch.arange(length, dtype=torch.float32) / sr
        clips.append(0.2 * torch.sin(2 * math.pi * f[:, None] * t[None]).sum(0).clamp(-1, 1))
    return torch.stack(clips[:max(n, len(paths))])


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-c", "--config_name", default="dps", choices=["ddim", "dps", "mpgd", "dsg", "diffmusic"])
    ap.add_argument("-t", "--task", default="music_inpainting", choices=TASKS)
    ap.add_argument("-m", "--model", default="musicldm", choices=["musicldm", "audioldm2"])
    ap.add_argument("--data", default="moises")
    ap.add_argument("--mask_type", default="box", choices=["box", "random", "periodic"])
    ap.add_argument("--supervised_space", default="mel_spectrogram")
    ap.add_argument("--weights", default="synthetic", help="checkpoint directory with {unet,vae,vocoder}/*.safetensors, or 'synthetic'")
    ap.add_argument("--wav", nargs="*", default=[], help="input clips (any PCM / float wav: mixed to mono and resampled to the data sample rate); synthetic clips fill up to --batch")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--prompt_embeds", default=None, help=".npy with (B, 512) text embeddings (MusicLDM)")
    ap.add_argument("--num_inference_steps", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--output_dir", default="outputs")
    ap.add_argument("--show_progress", action="store_true")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    cfg = compose(args.config_name, overrides=[f"data={args.data}", f"model={args.model}"])
    if args.model != "musicldm":
        raise SystemExit("this driver feeds MusicLDM's class-embedding conditioning; AudioLDM2 needs its T5 / GPT-2 states (see bench.py --workload)")
    device = torch.device("cuda")
    op, scale = build_operator(args.task, cfg, args.mask_type)
    pipe = get_pipeline(cfg.model.name).from_pretrained(args.weights, seed=args.seed).to(device)
    pipe.scheduler = get_scheduler(cfg.name)(operator=op, **cfg.model.scheduler)
    pipe_kw = dict(cfg.model.pipe)
    if args.num_inference_steps:
        pipe_kw["num_inference_steps"] = args.num_inference_steps
    sr, length = cfg.data.sample_rate, int(pipe_kw["audio_length_in_s"] * cfg.data.sample_rate)
    gt = load_clips(args.wav, args.batch, sr, length, args.seed).to(device)
    B = gt.shape[0]
    measurement = op.forward(gt)                                                   # run.py:290-300: degrade the ground truth once
    if args.prompt_embeds:
        pe = torch.from_numpy(np.load(args.prompt_embeds)).float()
    else:
        pe = torch.nn.functional.normalize(torch.randn(B, 512, generator=torch.Generator().manual_seed(args.seed)), dim=-1)
    gens = [torch.Generator().manual_seed(args.seed + i) for i in range(B)]
    audio = pipe(prompt_embeds=pe[:B], measurement=measurement, eta=cfg.scheduler.eta, ip_guidance_rate=cfg.scheduler.ip_guidance_rate,
                 generator=gens, show_progress=args.show_progress, supervised_space=args.supervised_space, **pipe_kw).audios
    out = Path(args.output_dir, cfg.model.name, cfg.data.name, args.config_name, args.task)
    for d in ("wav_input", "wav_recon", "wav_label", "mel_recon"):
        os.makedirs(out / d, exist_ok=True)
    to_mel = P.IdentityOperator(sample_rate=sr)                                     # log-mel of the result, like run.py:345-350
    for i in range(B):
        name = Path(args.wav[i]).stem if i < len(args.wav) else f"synthetic_{args.seed + i}"
        scipy.io.wavfile.write(out / "wav_label" / f"{name}.wav", sr, gt[i].cpu().numpy())
        if args.task != "phase_retrieval" and measurement.dim() == 2:
            scipy.io.wavfile.write(out / "wav_input" / f"{name}.wav", sr // scale, measurement[i].float().cpu().numpy())
        scipy.io.wavfile.write(out / "wav_recon" / f"{name}.wav", sr, audio[i])
        mel = to_mel.transform(torch.from_numpy(audio[i:i + 1]).to(device))[0].T      # (frames, 64)
        pipe.save_mel_spectrogram(mel[: int(pipe_kw["audio_length_in_s"] * 100)], out / "mel_recon" / f"{name}.png")
    ref = gt.cpu().numpy()
    print(f"wrote {B} clip(s) to {out}; LSD {LogSpectralDistance().score(ref, audio[:, :length]):.4f}  MSE {MeanSquaredError().score(ref, audio[:, :length]):.6f}")


if __name__ == "__main__":
    main()
