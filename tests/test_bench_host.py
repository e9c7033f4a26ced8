"""CPU: host-side helpers of bench.py -- the CPU share of the cpu_baseline leg (cgroup quota over affinity), the GPU count taken from
the KFD topology without touching the HIP runtime, and the command line (weak / strong scaling are exclusive)."""
import builtins
import io
import os
import subprocess
import sys

import pytest

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT)


def _fake_open(files):
    real = builtins.open

    def f(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise OSError(path)
            return io.StringIO(files[path])
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup"):
            raise OSError(path)
        return real(path, *a, **k)
    return f


def test_cpu_share_reads_the_cgroup_quota(monkeypatch):
    import bench
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    host, aff, quota = bench.cpu_share()
    assert quota == 16.0 and host == os.cpu_count() and aff == len(os.sched_getaffinity(0))
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert bench.cpu_share()[2] is None
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n",
                                                      "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert bench.cpu_share()[2] == 8.0
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n",
                                                      "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert bench.cpu_share()[2] is None


def test_visible_gpus_counts_kfd_nodes_and_honours_visibility_lists(monkeypatch, tmp_path):
    import bench
    nodes = tmp_path / "nodes"
    for i, simd in enumerate((0, 1024, 1024, 1024)):                 # node 0: the CPU (no SIMDs), three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 64\nsimd_count {simd}\n")
    real_listdir, real_open = os.listdir, builtins.open
    base = "/sys/class/kfd/kfd/topology/nodes"
    monkeypatch.setattr(os, "listdir", lambda p: real_listdir(str(nodes)) if p == base else real_listdir(p))
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace(base, str(nodes)), *a, **k))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpus() == 2
    monkeypatch.setattr(os, "listdir", lambda p: (_ for _ in ()).throw(OSError(p)) if p == base else real_listdir(p))
    assert bench.visible_gpus() is None                               # unknown: the ranks report a shortfall themselves


def test_command_line_documents_both_scaling_modes():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--batch", "--global-batch", "--workload", "--cpu-threads"):
        assert flag in out.stdout
    assert "--no-graph" not in out.stdout
