"""Generates tests/golden/*.npz by running the REFERENCE's own scheduler / operator sources
(imported from /root/reference under tests/oracle_shim) on small seeded inputs.
Run in the build container only:   python tests/golden/gen_golden.py
Nothing of the reference is copied: the fixtures are inputs + expected outputs."""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "oracle_shim"))
sys.path.insert(0, "/root/reference")

from diffmusic.schedulers import get_scheduler                       # noqa: E402  (reference)
from diffmusic.inverse_problem import operator as ref_op             # noqa: E402
from diffmusic.inverse_problem import get_noiser                     # noqa: E402
from diffmusic.torch_utils import randn_tensor                       # noqa: E402
from tests.golden.toy import ToyVae, ToyVocoder                      # noqa: E402

from tests.golden.cases import CASES, SCHED_CFG, SR, H, W, L      # noqa: E402


def make_operator(task):
    noiser = get_noiser("gaussian", 0.0)
    if task == "music_inpainting":
        # 1 "second" of L samples: box mask zeros [0.25 L, 0.5 L)
        return ref_op.MusicInpaintingOperator(audio_length_in_s=1, sample_rate=L, mask_type="box",
                                              start_inpainting_s=0.25, end_inpainting_s=0.5,
                                              mask_percentage=0.3, mask_duration_s=0.1, interval_s=0.2,
                                              noiser=noiser)
    if task == "phase_retrieval":
        return ref_op.PhaseRetrievalOperator(noiser=noiser)
    if task == "super_resolution":
        return ref_op.SuperResolutionOperator(sample_rate=SR, scale=2, noiser=noiser)
    raise ValueError(task)


def gen_scheduler_fixtures():
    out = {}
    vae, voc = ToyVae(), ToyVocoder()
    for ci, (name, task, eta, rate, n_steps, space) in enumerate(CASES):
        op = make_operator(task)
        sched = get_scheduler(name)(operator=op, **SCHED_CFG)
        sched.set_timesteps(n_steps)
        g = torch.Generator().manual_seed(100 + ci)
        clean = 0.3 * torch.sin(torch.arange(L) * 0.05)[None] + 0.05 * torch.randn(1, L, generator=g)
        y = op.forward(clean)
        for si in (0, len(sched.timesteps) // 2, len(sched.timesteps) - 1):
            t = int(sched.timesteps[si])
            x = torch.randn(1, 8, H, W, generator=g)
            eps = torch.randn(1, 8, H, W, generator=g)
            gen = torch.Generator().manual_seed(7 + si)
            kw = dict(measurement=y, vae=vae, vocoder=voc, original_waveform_length=L,
                      ip_guidance_rate=rate, supervised_space=space, eta=eta, generator=gen)
            if name == "ddim":
                kw.update(encoder_hidden_states=torch.zeros(1), encoder_hidden_states_1=torch.zeros(1))
            o = sched.step(eps, t, x, **kw)
            key = f"c{ci}_{name}_{task}_{space}_s{si}"
            out[key + "/x"] = x.numpy()
            out[key + "/eps"] = eps.numpy()
            out[key + "/y"] = y.numpy()
            out[key + "/meta"] = np.array([t, n_steps, eta, rate, 7 + si], dtype=np.float64)
            out[key + "/prev_sample"] = o.prev_sample.detach().numpy()
            out[key + "/pred_original_sample"] = o.pred_original_sample.detach().numpy()
            out[key + "/loss"] = o.loss.detach().numpy().astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "scheduler_steps.npz"), **out)
    print("scheduler fixtures:", len(out))


def gen_operator_fixtures():
    out = {}
    noiser = get_noiser("gaussian", 0.0)
    for kind in ("box", "periodic"):                 # masks (operator.py:87-121)
        op = ref_op.MusicInpaintingOperator(audio_length_in_s=10, sample_rate=SR, mask_type=kind,
                                            start_inpainting_s=2, end_inpainting_s=3, mask_percentage=0.3,
                                            mask_duration_s=0.1, interval_s=1.0, noiser=noiser)
        out[f"mask_{kind}/zeros"] = np.nonzero(op.mask[0].numpy() == 0)[0].astype(np.int64)
    torch.manual_seed(1234)
    op = ref_op.MusicInpaintingOperator(audio_length_in_s=10, sample_rate=SR, mask_type="random",
                                        start_inpainting_s=2, end_inpainting_s=3, mask_percentage=0.3,
                                        mask_duration_s=0.5, interval_s=1.0, noiser=noiser)
    out["mask_random_seed1234/zeros"] = np.nonzero(op.mask[0].numpy() == 0)[0].astype(np.int64)
    g = torch.Generator().manual_seed(5)
    wav = 0.2 * torch.randn(2, 4000, generator=g)
    out["wav"] = wav.numpy()
    out["phase_retrieval/forward"] = ref_op.PhaseRetrievalOperator(noiser=noiser).forward(wav).numpy()
    dr = ref_op.MusicDereverberationOperator(ir_length=500, decay_factor=0.99, noiser=noiser)
    torch.manual_seed(77)
    out["dereverb_seed77/forward"] = dr.forward(wav).numpy()
    torch.manual_seed(77)
    out["dereverb_seed77/ir"] = dr.generate_impulse_response(500, 0.99).numpy()
    gens = [torch.Generator().manual_seed(k) for k in range(3)]   # torch_utils.py:65-72
    out["randn_list/out"] = randn_tensor((3, 8, 5, 4), generator=gens, device=torch.device("cpu"),
                                         dtype=torch.float32).numpy()
    np.savez_compressed(os.path.join(HERE, "operators.npz"), **out)
    print("operator fixtures:", len(out))


if __name__ == "__main__":
    gen_scheduler_fixtures()
    gen_operator_fixtures()
