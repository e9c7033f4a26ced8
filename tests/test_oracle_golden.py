"""Pins the CPU oracle against golden vectors produced by the reference's own sources
(tests/golden/gen_golden.py): scheduler step bodies, masks, rect-STFT, reverb, randn_tensor."""
import os
import numpy as np
import pytest
import torch

from oracle import operators as O
from oracle import schedulers as S
from oracle.rng import randn_tensor
from tests.golden.toy import ToyVae, ToyVocoder
from tests.golden.cases import CASES, SCHED_CFG, L, SR


def _op(task):
    n = O.get_noiser("gaussian", 0.0)
    if task == "music_inpainting":
        return O.MusicInpaintingOperator(1, L, "box", 0.25, 0.5, 0.3, 0.1, 0.2, noiser=n)
    if task == "phase_retrieval":
        return O.PhaseRetrievalOperator(noiser=n)
    return O.SuperResolutionOperator(SR, 2, noiser=n)


@pytest.fixture(scope="module")
def steps(golden_dir):
    return np.load(os.path.join(golden_dir, "scheduler_steps.npz"))


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_scheduler_step_matches_reference(steps, ci):
    name, task, eta, rate, n_steps, space = CASES[ci]
    sched = S.get_scheduler(name)(operator=_op(task), per_clip_norm=False, **SCHED_CFG)
    sched.set_timesteps(n_steps)
    keys = sorted({k.rsplit("/", 1)[0] for k in steps.files if k.startswith(f"c{ci}_")})
    assert len(keys) == 3
    for key in keys:
        t, n, eta_, rate_, seed = steps[key + "/meta"]
        assert int(n) == n_steps
        x, eps, y = (torch.from_numpy(steps[key + "/" + s]) for s in ("x", "eps", "y"))
        o = sched.step(eps, int(t), x, eta=eta, generator=torch.Generator().manual_seed(int(seed)),
                       measurement=y, vae=ToyVae(), vocoder=ToyVocoder(), original_waveform_length=L,
                       ip_guidance_rate=rate, supervised_space=space)
        for f, tol in (("prev_sample", 2e-5), ("pred_original_sample", 2e-5)):
            ref = steps[key + "/" + f]
            got = getattr(o, f).numpy()
            assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (key, f)
        ref_loss = steps[key + "/loss"].reshape(-1)[0]
        got_loss = float(o.loss.reshape(-1)[0])
        assert abs(got_loss - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss)), key


def test_per_clip_norm_equals_b1_runs(steps):
    """Batch-B with per-clip norms == B independent B=1 runs."""
    name, task, eta, rate, n_steps, space = CASES[6]           # dsg
    sched = S.get_scheduler(name)(operator=_op(task), per_clip_norm=True, **SCHED_CFG)
    sched.set_timesteps(n_steps)
    g = torch.Generator().manual_seed(3)
    x, eps = torch.randn(2, 8, 10, 4, generator=g), torch.randn(2, 8, 10, 4, generator=g)
    y = 0.1 * torch.randn(2, L, generator=g)
    z = torch.randn(2, 8, 10, 4, generator=g)
    kw = dict(eta=eta, vae=ToyVae(), vocoder=ToyVocoder(), original_waveform_length=L,
              ip_guidance_rate=rate, supervised_space=space)
    both = sched.step(eps, 501, x, measurement=y, sample_noise=z, **kw).prev_sample
    for i in range(2):
        one = sched.step(eps[i:i + 1], 501, x[i:i + 1], measurement=y[i:i + 1], sample_noise=z[i:i + 1], **kw).prev_sample
        assert torch.allclose(both[i:i + 1], one, atol=1e-5)


def test_operator_fixtures(golden_dir):
    fx = np.load(os.path.join(golden_dir, "operators.npz"))
    n = O.get_noiser("gaussian", 0.0)
    for kind in ("box", "periodic"):
        op = O.MusicInpaintingOperator(10, SR, kind, 2, 3, 0.3, 0.1, 1.0, noiser=n)
        assert np.array_equal(np.nonzero(op.mask[0].numpy() == 0)[0], fx[f"mask_{kind}/zeros"])
    assert np.array_equal(fx["mask_box/zeros"], np.arange(32000, 48000))
    torch.manual_seed(1234)
    op = O.MusicInpaintingOperator(10, SR, "random", 2, 3, 0.3, 0.5, 1.0, noiser=n)
    assert np.array_equal(np.nonzero(op.mask[0].numpy() == 0)[0], fx["mask_random_seed1234/zeros"])
    wav = torch.from_numpy(fx["wav"])
    mag = O.PhaseRetrievalOperator(noiser=n).forward(wav).numpy()
    assert mag.shape == fx["phase_retrieval/forward"].shape == (2, 513, 26)
    assert np.abs(mag - fx["phase_retrieval/forward"]).max() < 1e-4
    dr = O.MusicDereverberationOperator(500, 0.99, noiser=n)
    torch.manual_seed(77)
    out = dr.forward(wav).numpy()
    assert out.shape == (2, 4001) and np.abs(out - fx["dereverb_seed77/forward"]).max() < 1e-4
    out2 = dr.forward(wav, ir=torch.from_numpy(fx["dereverb_seed77/ir"])).numpy()
    assert np.abs(out2 - fx["dereverb_seed77/forward"]).max() < 1e-4
    gens = [torch.Generator().manual_seed(k) for k in range(3)]
    r = randn_tensor((3, 8, 5, 4), generator=gens, device=torch.device("cpu"), dtype=torch.float32).numpy()
    assert np.array_equal(r, fx["randn_list/out"])


def test_philox_known_answer_vectors():
    """Random123's philox4x32-10 known-answer vectors pin the numpy restatement that the device RNG is tested against."""
    from oracle.rng import philox4x32_10
    kat = [([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0), [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert [int(v) for v in philox4x32_10([ctr], key)[0]] == want
