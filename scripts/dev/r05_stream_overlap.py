"""GPU box: do a chain of small launches (a U-Net forward) on a high-priority stream and a sequence of chip-filling launches (a vocoder
forward + backward) on a normal stream overlap at all?  No dependencies between the two streams; both are enqueued up front.
Prints: each alone, both together (wall), and the first-launch .. last-launch span of the U-Net under the concurrent sweep."""
import sys, time, torch
sys.path.insert(0, '.')
from diffmusic_amd.engine import UNetEngine, HifiGanEngine
from diffmusic_amd import _lib as L
prio = int(sys.argv[1]) if len(sys.argv) > 1 else -1
unet = UNetEngine(); unet.load_state_dict(unet.synth_state_dict(0))
voc = HifiGanEngine(); voc.load_state_dict(voc.synth_state_dict(1))
B = 4
x = torch.randn(2 * B, 8, 250, 16, device="cuda"); t = torch.full((2 * B,), 501.0, device="cuda"); cls = torch.randn(2 * B, 512, device="cuda")
mel = torch.randn(B, 1000, 64, device="cuda").to(L.act_dtype())
U, S = torch.cuda.Stream(priority=prio), torch.cuda.Stream()
def run_unet(n=1):
    with torch.cuda.stream(U):
        for _ in range(n): unet.forward(x, t, cls)
def run_sweep(n=1):
    with torch.cuda.stream(S):
        for _ in range(n):
            w = voc.forward(mel); voc.backward(torch.ones_like(w))
for _ in range(2): run_unet(); run_sweep()
torch.cuda.synchronize()
def wall(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
tu = min(wall(lambda: run_unet(2)) for _ in range(3)) / 2
ts = min(wall(lambda: run_sweep(2)) for _ in range(3)) / 2
def both():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run_sweep(2)                      # enqueue the chip-filling work first, then the chain
    with torch.cuda.stream(U): e0.record()
    run_unet(2)
    with torch.cuda.stream(U): e1.record()
    return e0, e1
res = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0, e1 = both(); torch.cuda.synchronize()
    res.append(((time.perf_counter() - t0) * 1e3, e0.elapsed_time(e1)))
res.sort()
print(f"priority {prio}: U-Net alone {tu:.2f} ms | vocoder fwd+bwd (B = {B}) alone {ts:.2f} ms | 2 x both together: wall {res[0][0]:.2f} ms "
      f"(serial would be {2 * (tu + ts):.2f}), the two U-Net forwards span {res[0][1]:.2f} ms on their stream", flush=True)
