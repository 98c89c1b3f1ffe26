"""The C-ABI library loads without a GPU and exports every symbol that
include/jsmpeg_hip.h declares; creating a decoder without a device fails
loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, have_gpu
from jsmpeg_amd import batch, build, cabi, mp2


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "jsmpeg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:mpeg1_decoder|mp2_decoder|jsmpeg_hip)_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_reference_abi():
    names = declared_symbols()
    for n in cabi.ABI_SYMBOLS:          # the 15 functions of reference src/wasm/mpeg1.h:10-25
        assert n in names
    assert len(cabi.ABI_SYMBOLS) == 15
    for n in cabi.MP2_ABI_SYMBOLS:      # the 10 functions of reference src/wasm/mp2.h:10-20
        assert n in names
    assert len(cabi.MP2_ABI_SYMBOLS) == 10


def test_library_exports_every_declared_symbol(hip_lib):
    lib = ctypes.CDLL(hip_lib)
    for n in declared_symbols():
        assert hasattr(lib, n), n
    for n in batch.BATCH_SYMBOLS + mp2.MP2_BATCH_SYMBOLS + cabi.MP2_ABI_SYMBOLS:
        assert hasattr(lib, n), n


@pytest.mark.skipif(have_gpu(), reason="a GPU is present")
def test_no_device_means_loud_failure(hip_lib):
    lib = cabi.load(hip_lib)
    assert not lib.mpeg1_decoder_create(4096, cabi.MODE_EXPAND)
    L = batch.lib()
    assert b"no CPU fallback" in L.jsmpeg_hip_last_error()
    with pytest.raises(RuntimeError):
        batch.Batch(320, 240, 1, 4, 1 << 16)
    assert not cabi.load_mp2(hip_lib).mp2_decoder_create(4096, cabi.MODE_EXPAND)
    assert b"no CPU fallback" in L.jsmpeg_hip_last_error()
    with pytest.raises(RuntimeError):
        mp2.Mp2Batch(1, 1 << 16)


def test_hot_kernels_use_no_scratch():
    usage = build.check_kernel_resources()
    assert any("k_parse" in k for k in usage) and any("k_recon" in k for k in usage)
    assert any("k_mp2_matrix" in k for k in usage) and any("k_mp2_window" in k for k in usage)
