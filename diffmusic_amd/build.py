"""Builds libdiffmusic_hip.so (hipcc, gfx950 only) in-tree.  `python -m diffmusic_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "lib", "libdiffmusic_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("DMX_EXTRA_FLAGS", "").split()


# per-file additions (reasons: header note of the file)
FILE_FLAGS = {"flash_attn.hip": ["-fno-honor-nans", "-fno-slp-vectorize"]}
STAMP = os.path.join(OBJ, "flags.stamp")


def _flags_key(hipcc):
    """Everything besides the sources that decides what the objects contain: compiler path + the full flag list (DMX_EXTRA_FLAGS
    included).  A library built with other flags (e.g. an ablation build of a dev script) is rebuilt, never reused."""
    import hashlib
    return hashlib.sha256(("\0".join([hipcc] + FLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(FILE_FLAGS.items())])).encode()).hexdigest()


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "diffmusic_hip.h"))
    key = _flags_key(hipcc)
    try:
        with open(STAMP) as fh:
            same_flags = fh.read().strip() == key
    except OSError:
        same_flags = False
    if not same_flags:
        force = True                      # objects of unknown / different flags: rebuild everything
    objs, jobs = [], []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        # kernels must not touch scratch: a demoted register array once doubled the write traffic of every GEMM epilogue
        name = None
        for line in r.stderr.splitlines():
            if "Function Name:" in line:
                name = line.split("Function Name:")[1].split()[0]
            elif "ScratchSize [bytes/lane]:" in line and int(line.split("ScratchSize [bytes/lane]:")[1].split()[0]) > 0:
                print(f"WARNING: kernel {name} uses scratch memory ({line.split(':')[-1].split('[')[0].strip()} bytes/lane)", file=sys.stderr, flush=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    with open(STAMP, "w") as fh:
        fh.write(key + "\n")
    return LIB


TORCH_OPS_SRC = os.path.join(HERE, "csrc_torch", "torch_ops.cpp")
TORCH_OPS_LIB = os.path.join(HERE, "lib", "libdiffmusic_torch_ops.so")


def build_torch_ops(force=False, verbose=False):
    """libdiffmusic_torch_ops.so: TORCH_LIBRARY(diffmusic_hip) wrappers over the C ABI (csrc_torch/torch_ops.cpp).  Host code
    only (the kernels live in libdiffmusic_hip.so, which it links), so it is compiled with g++ against torch's headers."""
    import torch
    lib = build_library(force=False, verbose=verbose)
    hdr = os.path.join(os.path.dirname(HERE), "include", "diffmusic_hip.h")
    # the op library is ABI-bound to the torch it was compiled against: a stamp holds that version, another torch rebuilds it
    stamp = TORCH_OPS_LIB + ".torch_version"
    try:
        with open(stamp) as fh:
            same_torch = fh.read().strip() == torch.__version__
    except OSError:
        same_torch = False
    if not (force or not same_torch or _stale(TORCH_OPS_LIB, [TORCH_OPS_SRC, hdr, lib])):
        return TORCH_OPS_LIB
    tdir = os.path.dirname(torch.__file__)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w",
           f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include", f"-I{rocm}/include", TORCH_OPS_SRC, "-o", TORCH_OPS_LIB,
           f"-L{os.path.dirname(lib)}", "-ldiffmusic_hip", f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the torch op library failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    with open(stamp, "w") as fh:
        fh.write(torch.__version__ + "\n")
    return TORCH_OPS_LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_torch_ops(force="--force" in sys.argv, verbose=True))
