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
    assert h.dmx_abi_version() == 1
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
