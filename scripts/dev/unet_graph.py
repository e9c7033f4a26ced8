import sys, torch, time
sys.path.insert(0, '.')
from diffmusic_amd.engine import UNetEngine
eng = UNetEngine()
eng.load_state_dict(eng.synth_state_dict(0))
B = 16
x = torch.randn(B, 8, 250, 16, device="cuda"); t = torch.full((B,), 501.0, device="cuda"); cls = torch.randn(B, 512, device="cuda")
def ev(): return torch.cuda.Event(enable_timing=True)
for _ in range(3): out = eng.forward(x, t, cls)
torch.cuda.synchronize()
a, b = ev(), ev(); t0 = time.perf_counter(); a.record()
for _ in range(5): out = eng.forward(x, t, cls)
b.record(); t_host = (time.perf_counter() - t0) / 5; torch.cuda.synchronize()
print(f"eager: device {a.elapsed_time(b)/5:.2f} ms/iter, host enqueue {t_host*1e3:.2f} ms/iter")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): out = eng.forward(x, t, cls)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out_g = eng.forward(x, t, cls)
torch.cuda.synchronize()
for _ in range(2): g.replay()
torch.cuda.synchronize()
a, b = ev(), ev(); a.record()
for _ in range(5): g.replay()
b.record(); torch.cuda.synchronize()
print(f"graph replay: {a.elapsed_time(b)/5:.2f} ms/iter ; max diff vs eager {float((out_g - out).abs().max()):.3e}")
