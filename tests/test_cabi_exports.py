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
    for n in batch.BATCH_SYMBOLS + mp2.MP2_BATCH_SYMBOLS + mp2.MP2_LIVE_SYMBOLS + cabi.MP2_ABI_SYMBOLS:
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
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mp2.Mp2Live(4)


def test_hot_kernels_use_no_scratch():
    usage = build.check_kernel_resources()
    assert any("k_parse" in k for k in usage) and any("k_recon" in k for k in usage)
    assert any("k_mp2_matrix" in k for k in usage) and any("k_mp2_window" in k for k in usage)


def test_graft_entry_functions_have_no_undefined_names():
    """smoke() only runs on the GPU box: a name it uses without importing it (it happened once) must fail HERE"""
    import ast
    import builtins
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    tree = ast.parse(src)
    module_names = set(dir(builtins))
    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            module_names |= {(a.asname or a.name).split(".")[0] for a in node.names}
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            module_names.add(node.name)
        elif isinstance(node, ast.Assign):
            for t in node.targets:
                module_names |= {n.id for n in ast.walk(t) if isinstance(n, ast.Name)}
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        local = {a.arg for a in fn.args.args}
        for node in ast.walk(fn):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                local |= {(a.asname or a.name).split(".")[0] for a in node.names}
            elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
                local.add(node.id)
            elif isinstance(node, (ast.FunctionDef, ast.Lambda)) and node is not fn:
                local |= {a.arg for a in node.args.args}
            elif isinstance(node, ast.ExceptHandler) and node.name:
                local.add(node.name)
            elif isinstance(node, ast.comprehension):
                local |= {n.id for n in ast.walk(node.target) if isinstance(n, ast.Name)}
        used = {n.id for n in ast.walk(fn) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
        assert not (used - local - module_names), "%s(): undefined %r" % (fn.name, sorted(used - local - module_names))


def test_hot_kernels_keep_their_occupancy():
    """registers / LDS of the hot kernels as the design counts on them: k_recon five workgroups per
    CU (96 registers, 25 LDS granules of 1280 bytes), k_parse two workgroups of eight wavefronts per CU (80 KB, four
    wavefronts per SIMD), k_scan five workgroups per CU; no scratch anywhere (build.check_kernel_resources raises)"""
    use = build.check_kernel_resources()
    recon = next(v for n, v in use.items() if "k_recon" in n)
    parse = next(v for n, v in use.items() if "7k_parse" in n)
    scan = next(v for n, v in use.items() if "k_scan" in n)
    assert recon["VGPRs"] <= 96 and recon["LDS Size"] <= 25 * 1280 and recon["Occupancy"] >= 5
    assert parse["VGPRs"] <= 128 and parse["LDS Size"] <= 81920 and parse["Occupancy"] >= 4
    assert scan["LDS Size"] <= 23 * 1280 and scan["VGPRs"] <= 96
    assert all(v.get("ScratchSize", 0) == 0 for v in use.values())
