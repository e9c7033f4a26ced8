"""-m gpu: the fused forward attention kernel (C-ABI hook dmx_flash_attn_raw) against torch fp32 softmax attention."""
import ctypes as C
import math
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,heads,dh,Nq,Nk,bias", [(2, 8, 32, 1000, 1000, False), (1, 8, 48, 252, 252, False), (2, 8, 80, 64, 64, False),
                                                   (2, 4, 32, 300, 8, True), (1, 2, 64, 130, 77, True), (1, 1, 96, 129, 200, False),
                                                   (3, 8, 32, 1000, 52, True),
                                                   # grids of more than 256 128-query workgroups: the two-query-tile form (QT = 2) of every
                                                   # head-dim class, with ragged query / key tails and the biased (generic) block path
                                                   (5, 8, 32, 1100, 1000, False), (5, 8, 32, 900, 333, True), (6, 6, 48, 1000, 200, False),
                                                   (8, 5, 64, 1000, 129, True), (9, 4, 80, 1000, 96, False), (9, 4, 96, 1000, 70, True)])
def test_flash_attention_forward(B, heads, dh, Nq, Nk, bias):
    from diffmusic_amd import _lib as L
    adt = L.act_dtype()
    g = torch.Generator().manual_seed(11)
    Cc = heads * dh
    q = torch.randn(B, Nq, Cc, generator=g).to(adt).cuda()
    k = torch.randn(B, Nk, Cc, generator=g).to(adt).cuda()
    v = torch.randn(B, Nk, Cc, generator=g).to(adt).cuda()
    cb = None
    if bias:
        cb = torch.where(torch.rand(B, Nk, generator=g) < 0.3, -10000.0, 0.0)
        cb[:, 0] = 0.0
        cb = cb.cuda().contiguous()
    o = torch.zeros(B, Nq, Cc, dtype=adt, device="cuda")
    scale = 1.0 / math.sqrt(dh)
    L.check(L.lib().dmx_flash_attn_raw(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(o.data_ptr()),
                                       C.c_void_p(cb.data_ptr()) if cb is not None else None, B, Nq, Nk, 0, Cc, heads, scale,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "flash_attn")
    torch.cuda.synchronize()
    qf = q.float().view(B, Nq, heads, dh).transpose(1, 2)
    kf = k.float().view(B, Nk, heads, dh).transpose(1, 2)
    vf = v.float().view(B, Nk, heads, dh).transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) * scale
    if cb is not None:
        s = s + cb[:, None, None, :]
    ref = (torch.softmax(s, dim=-1) @ vf).transpose(1, 2).reshape(B, Nq, Cc)
    err = ((o.float() - ref).norm() / ref.norm()).item()
    assert err < 3e-3, err       # fp16 probabilities and outputs; statistics and accumulation in fp32
