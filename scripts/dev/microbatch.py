"""Experiment: one B=8 pipeline vs two concurrent B=4 pipelines on two HIP streams (same total work)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda")
def run(n_pipes, B_each, steps=6, warm=2):
    ps = [bench.build_problem(B_each, i, dev) for i in range(n_pipes)]
    streams = [torch.cuda.Stream() for _ in range(n_pipes)]
    lats = [p[3] for p in ps]
    ts = ps[0][0].scheduler._timesteps_host
    def step(k):
        for i, (pipe, op, meas, _, pe2, L) in enumerate(ps):
            with torch.cuda.stream(streams[i]):
                lats[i], _ = bench.one_step(pipe, lats[i], ts[k], pe2, meas, L)
    for k in range(warm): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warm, warm + steps): step(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
print("1 x B=8 : %.2f ms/step" % run(1, 8))
print("2 x B=4 : %.2f ms/step" % run(2, 4))
print("4 x B=2 : %.2f ms/step" % run(4, 2))
