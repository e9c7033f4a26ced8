"""GPU box: the wrapped CLAP (HTS-AT) tower of the style-guidance operator (config 5): forward + input-gradient of the Gram loss at
B = 8 -- fp32 eager (round 2), fp16 / bf16 autocast, and each of them captured in a HIP graph.  Prints ms per call and the
distance of loss / gradient to the fp32 eager result."""
import sys, time, torch
sys.path.insert(0, '.')
from diffmusic_amd import inverse_problem as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
op = P.StyleGuidanceOperator(16000, noiser=None, device="cuda")
g = torch.Generator().manual_seed(0)
wav = (0.2 * torch.randn(B, 160032, generator=g)).cuda()
meas = (0.2 * torch.randn(B, 160000, generator=g)).cuda()
ref = op.transform(meas)
feats, n48 = op._features(wav, 160000)
feats = feats.contiguous()

def fb(x, ac):
    with torch.enable_grad():
        fg = x.detach().requires_grad_(True)
        with torch.autocast("cuda", dtype=ac, enabled=ac is not None):
            gram = op._gram(fg)
        diff = (ref - gram.float()).flatten(1)
        loss = torch.linalg.vector_norm(diff, dim=1)
        (d,) = torch.autograd.grad(loss.sum(), fg)
    return loss.detach(), d

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): out = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, 1e3 * (time.perf_counter() - t0) / n, out

base = None
for name, ac in (("fp32", None), ("fp16", torch.float16), ("bf16", torch.bfloat16)):
    dev_ms, wall_ms, (loss, d) = timeit(lambda: fb(feats, ac))
    if base is None: base = (loss.clone(), d.clone())
    rl = float(((loss - base[0]).abs() / base[0].abs()).max()); rd = float((d - base[1]).norm() / base[1].norm())
    print(f"eager {name}: device {dev_ms:.2f} ms, wall {wall_ms:.2f} ms; loss rel {rl:.2e}, grad rel-L2 {rd:.2e}, finite {bool(torch.isfinite(d).all())}", flush=True)
    # graph capture
    try:
        static_in = feats.clone()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fb(static_in, ac)
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            sl, sd = fb(static_in, ac)
        def replay():
            static_in.copy_(feats); gr.replay(); return sl, sd
        dev_ms, wall_ms, (loss, d) = timeit(replay)
        rl = float(((loss - base[0]).abs() / base[0].abs()).max()); rd = float((d - base[1]).norm() / base[1].norm())
        print(f"graph {name}: device {dev_ms:.2f} ms, wall {wall_ms:.2f} ms; loss rel {rl:.2e}, grad rel-L2 {rd:.2e}", flush=True)
    except Exception as e:
        print(f"graph {name}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
