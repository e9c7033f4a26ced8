"""-m gpu: the hand-written CLAP HTS-AT tower (csrc/htsat.hip, `HtsatEngine`) against `transformers.ClapAudioModel` in fp32 with the SAME
weights (every parameter perturbed so that biases, LayerNorm affines, BatchNorm statistics and the relative position bias tables all
matter): the hidden state entering every Swin block (tape hook), the final token features, the Gram matrix and its transpose, and the
input-gradient of a scalar through the whole tower (torch autograd as the reference).  Reference intent:
diffmusic/inverse_problem/operator.py:253-271 (style guidance); bars from VERDICT round 4: features <= 5e-3, gradient cos >= 0.995."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _cos(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))


@pytest.fixture(scope="module")
def tower():
    from transformers import ClapAudioConfig, ClapAudioModel
    from diffmusic_amd.engine import HtsatEngine
    torch.manual_seed(11)
    model = ClapAudioModel(ClapAudioConfig()).float().eval()
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("relative_position_bias_table"):
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        bn = model.audio_encoder.batch_norm
        bn.running_mean.copy_(-20.0 + 3.0 * torch.randn(64, generator=g))
        bn.running_var.copy_(150.0 * (1.0 + 0.3 * torch.rand(64, generator=g)))
    model = model.cuda()
    for p in model.parameters():
        p.requires_grad_(False)
    eng = HtsatEngine(model.config).load_state_dict(model.state_dict(), strict=True)
    return model, eng


def _mel(B, frames, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(frames, dtype=torch.float32)[None, :, None]
    f = torch.arange(64, dtype=torch.float32)[None, None, :]
    return (-25.0 + 12.0 * torch.sin(0.05 * t + 0.3 * f) + 6.0 * torch.randn(B, frames, 64, generator=g)).cuda().contiguous()


def _hf_tokens(out):
    """ClapAudioEncoder's last_hidden_state (B, C, 2, 32) back to (B, 64 tokens in grid order, C): [b, c, fb, g * 8 + t] = token (g * 2 + fb) * 8 + t."""
    B, Cc = out.shape[:2]
    return out.reshape(B, Cc, 2, 4, 8).permute(0, 3, 2, 4, 1).reshape(B, 64, Cc)


@pytest.mark.parametrize("B,frames", [(2, 1001), (1, 201), (1, 1024)])
def test_tower_forward_block_by_block_and_features(tower, B, frames):
    from diffmusic_amd import _lib as L
    model, eng = tower
    mel = _mel(B, frames, 5 + frames)
    block_inputs = []
    hooks = []
    for stage in model.audio_encoder.layers:
        for blk in stage.blocks:
            hooks.append(blk.register_forward_pre_hook(lambda m, args: block_inputs.append(args[0].detach().clone())))
    with torch.no_grad():
        ref = model(input_features=mel[:, None], is_longer=None, return_dict=True).last_hidden_state
    for h in hooks:
        h.remove()
    feat = eng.forward(mel, keep_state=True)
    torch.cuda.synchronize()
    depths, k, worst = model.config.depths, 0, 0.0
    for s, d in enumerate(depths):
        for j in range(d):
            want = block_inputs[k]
            got = torch.empty(want.numel(), dtype=L.act_dtype(), device="cuda")
            n = L.lib().dmx_htsat_tape_raw(eng._h, s, j, 0, C.c_void_p(got.data_ptr()), got.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert n == want.numel(), (s, j, n, want.shape)
            torch.cuda.synchronize()
            r = _rel(got.reshape(want.shape), want)
            worst = max(worst, r)
            print(f"  stage {s} block {j}: input rel-L2 {r:.2e}")
            assert r < 6e-3, f"hidden state entering stage {s} block {j} deviates: {r:.2e}"
            k += 1
    r = _rel(feat, _hf_tokens(ref))
    print(f"  final token features rel-L2 {r:.2e} (worst block input {worst:.2e})")
    assert r < 5e-3


def test_gram_forward_and_transpose(tower):
    from diffmusic_amd.engine import gram, gram_backward
    g = torch.Generator().manual_seed(3)
    f = torch.randn(3, 64, 768, generator=g).cuda()
    dg = torch.randn(3, 768, 768, generator=g).cuda()
    G = gram(f)
    fr = f.clone().requires_grad_(True)
    Gr = torch.bmm(fr.transpose(1, 2), fr) / 64
    (dfr,) = torch.autograd.grad((Gr * dg).sum(), fr)
    assert _rel(G, Gr) < 1e-5 and _rel(gram_backward(f, dg), dfr) < 1e-5


@pytest.mark.parametrize("B,frames", [(2, 1001), (1, 201)])
def test_tower_input_gradient(tower, B, frames):
    model, eng = tower
    mel = _mel(B, frames, 9 + frames)
    g = torch.Generator().manual_seed(4)
    cot = torch.randn(B, 64, 768, generator=g).cuda()
    eng.forward(mel, keep_state=True)
    scale = torch.tensor([0.5, 2.0][:B], device="cuda")
    dmel = eng.backward((cot / scale[:, None, None]).contiguous(), scale=scale)
    x = mel.clone().requires_grad_(True)
    out = _hf_tokens(model(input_features=x[:, None], is_longer=None, return_dict=True).last_hidden_state)
    (ref,) = torch.autograd.grad((out * cot).sum(), x)
    r, c = _rel(dmel, ref), _cos(dmel, ref)
    print(f"  d(features . cotangent) / d mel: rel-L2 {r:.2e}, cos {c:.5f}")
    assert bool(torch.isfinite(dmel).all())
    assert c > 0.995 and r < 5e-2
    again = eng.backward((cot / scale[:, None, None]).contiguous(), scale=scale)
    assert torch.equal(again, dmel)                                   # bitwise run to run (no atomics)
