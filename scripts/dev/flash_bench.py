"""GPU box: flash-attention forward alone, hot loop, one or more library builds interleaved on the same device.

usage: flash_bench.py [libA.so libB.so ...]     (files under diffmusic_amd/lib/; default: the product library)
Shapes = the self / cross attentions of the two U-Nets at the bench batch (2B = 16 MusicLDM, 2B = 8 AudioLDM2).
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # name, B, Nq, Nk, C, heads
    ("musicldm L1 self", 16, 1024, 1024, 256, 8),
    ("musicldm L2 self", 16, 256, 256, 384, 12),
    ("musicldm L3 self", 16, 64, 64, 640, 20),
    ("audioldm2 L1 self", 8, 1024, 1024, 256, 4),
    ("audioldm2 L1 cross", 8, 1024, 8, 256, 4),
    ("audioldm2 L2 self", 8, 256, 256, 384, 6),
]


def load(name):
    lib = C.CDLL(os.path.join(ROOT, "diffmusic_amd", "lib", name))
    f = lib.dmx_flash_attn_raw
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_float, C.c_void_p]
    return f


def main():
    names = sys.argv[1:] or ["libdiffmusic_hip.so"]
    fns = [load(n) for n in names]
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (tag, B, Nq, Nk, Cc, heads) in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(1)
        q, k, v = (torch.randn(B, n, Cc, generator=g).to(dev, torch.float16) for n in (Nq, Nk, Nk))
        outs, res = [], []
        for rep in range(2):
            for i, f in enumerate(fns):
                o = torch.empty(B, Nq, Cc, device=dev, dtype=torch.float16)
                scale = (Cc // heads) ** -0.5
                call = lambda: f(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, B, Nq, Nk, Cc, Cc, heads, scale, st)
                for _ in range(20):
                    assert call() == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 300
                e0.record()
                for _ in range(n):
                    call()
                e1.record()
                torch.cuda.synchronize()
                res.append((names[i], rep, e0.elapsed_time(e1) / n * 1e3))
                if rep == 0:
                    outs.append(o.float())
        flop = 4.0 * B * Nq * Nk * Cc
        line = "  ".join(f"{n}#{r}: {us:7.2f} us ({flop / us * 1e-6:6.1f} TF/s)" for n, r, us in res)
        diff = max(((outs[0] - o).abs().max().item() for o in outs[1:]), default=0.0)
        print(f"{tag:20s} {line}   max|diff| vs first {diff:.2e}", flush=True)


if __name__ == "__main__":
    main()
