"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header
declares (no compute without a GPU)."""
import ctypes
import importlib
import os

import pytest
import torch


def test_build_and_symbols(rg):
    b = importlib.import_module("rag-gesture_amd.build")
    lib_path = b.build(verbose=False)
    assert os.path.exists(lib_path)
    lib = rg.capi.load_library()
    syms = rg.capi.header_symbols()
    assert "rg_create" in syms and "rg_ddim_update" in syms and len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    assert lib.rg_version() >= 100


def test_fails_loudly_without_gpu(rg):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rg.capi.RgError):
        rg.capi.Handle()
    with pytest.raises(rg.capi.RgError):
        rg.smoke.run()
