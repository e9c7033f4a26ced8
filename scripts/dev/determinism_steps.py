"""Two identical short trajectories per workload must agree bit for bit."""
import sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda")
for wl in ("dps_inpainting", "dsg_phase_audioldm2", "mpgd_sr4"):
    outs = []
    for rep in range(2):
        torch.manual_seed(0)
        pipe, op, meas, lat, cond, L = bench.build_problem(int(sys.argv[1]) if len(sys.argv) > 1 else 2, 0, dev, wl)
        ts = pipe.scheduler._timesteps_host
        for k in range(3):
            lat, loss = bench.one_step(pipe, lat, ts[k], cond, meas, L)
        outs.append((lat.clone(), loss.clone()))
        del pipe
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    print(f"{wl:24s} {'IDENTICAL' if same else 'DIFFERENT'}  max|dlat| {(outs[0][0]-outs[1][0]).abs().max().item():.3e}", flush=True)
