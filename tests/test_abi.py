"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/diffmusic_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "diffmusic_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dmx_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_all_declared_symbols():
    from diffmusic_amd.build import build_library
    lib = build_library()
    assert os.path.exists(lib)
    h = ctypes.CDLL(lib)
    names = _declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, missing
    h.dmx_abi_version.restype = ctypes.c_int
    assert h.dmx_abi_version() == 4
    assert h.dmx_act_dtype() in (0, 1)


def test_binding_signatures_cover_the_header():
    from diffmusic_amd import _lib
    bound = set(_lib._SIGS)
    declared = set(_declared_symbols())
    assert declared <= bound, sorted(declared - bound)


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    from diffmusic_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.lib()
        raise AssertionError("expected a RuntimeError")
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "diffmusic_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)


def test_torch_op_library_loads_and_registers_every_stage_op(monkeypatch, tmp_path):
    """TORCH_LIBRARY(diffmusic_hip) layer: builds, loads without a GPU, registers all stage ops, fails
    loudly when the library is missing, and refuses CPU tensors (no CPU fallback)."""
    import pytest
    import torch
    from diffmusic_amd.build import build_torch_ops
    from diffmusic_amd import ops
    assert os.path.exists(build_torch_ops())
    h = ops.load()
    for name in ops.OP_NAMES:
        assert hasattr(h, name), name
        assert str(getattr(h, name).default._schema).startswith(f"diffmusic_hip::{name}(")
    for want in ("unet_fwd", "vae_dec_fwd", "vae_dec_bwd", "hifigan_fwd", "hifigan_bwd", "logmel_fwd", "logmel_bwd", "stft_mag_fwd",
                 "stft_mag_bwd", "melscale_fwd", "resample_fwd", "resample_bwd", "mask_mul", "l2norm", "sched_update", "randn_philox"):
        assert want in ops.OP_NAMES                                   # the op list of SURVEY.md section 8b.4
    with pytest.raises(RuntimeError, match="GPU tensor"):
        h.sched_pred_x0(torch.zeros(2, 4), torch.zeros(2, 4), 0.5)
    monkeypatch.setattr(ops, "_loaded", False)
    monkeypatch.setattr(ops, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.load()
