"""CPU-side checks of the C-ABI library: it loads, and exports every function
include/mzsearch.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mzsearch.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mzs_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from muzero_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with make -C muzero_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mzsearch.h but not exported"
    assert set(names) == set(_lib.EXPORTED_SYMBOLS)
    lib.mzs_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.mzs_version()


def test_weight_count_matches_layout():
    from muzero_amd import _lib
    from conftest import load_golden
    lib = _lib.load()
    for tag in ("tag123", "tag127", "tag133", "tag137", "synA4", "synA3"):
        w = load_golden(f"weights_{tag}.npz")
        d = _lib.Dims(int(w["obs_dim"]), int(w["A"]), int(w["S_dim"]), int(w["H"]), int(w["L"]))
        assert lib.mzs_weight_count(ctypes.byref(d)) == w["blob"].size


def test_create_rejects_bad_arguments_without_gpu():
    from muzero_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    cfg = _lib.Config(_lib.Dims(4, 100, 31, 64, 0), 1, 10, 0, 0, 0.997, 0.25, None)
    assert lib.mzs_create(ctypes.byref(cfg), ctypes.byref(h)) == -1   # n_actions > 64
    assert b"n_actions" in lib.mzs_last_error()
    assert lib.mzs_create(None, ctypes.byref(h)) == -1
