"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/chirrup_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "chirrup_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([a-z][a-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(names))


def test_header_declares_the_boundary():
    names = _declared_functions()
    for must in ("wkv7_fwd_seq", "wkv7_fwd_one", "spmv_fp16", "mm8_seq", "mm8_one"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import chirrup_amd
    from chirrup_amd import lib

    if not os.path.exists(chirrup_amd.LIB_PATH):
        chirrup_amd.build()
    L = chirrup_amd.load()
    for name in _declared_functions():
        assert hasattr(L, name), f"{name} declared in include/chirrup_amd.h but not exported"
        assert name in lib.SIGNATURES, f"{name} has no ctypes signature in chirrup_amd/lib.py"
    header = open(os.path.join(ROOT, "include", "chirrup_amd.h")).read()
    assert int(re.search(r"#define\s+CHIRRUP_ABI_VERSION\s+(\d+)", header).group(1)) == lib.ABI_VERSION == L.chirrup_abi_version() == 4
    assert L.chirrup_target_arch() == b"gfx950"


def test_argument_validation_needs_no_gpu():
    """Bad arguments are rejected on the host before any launch (reference: assert(H*_N_==C),
    Albatross/cuda/rwkv7_state_fwd_fp16.cu:314)."""
    import chirrup_amd

    L = chirrup_amd.load()
    p = ctypes.c_void_p(4096)  # never dereferenced: validation fails first
    assert L.wkv7_fwd_seq(1, 1, 100, 2, p, p, p, p, p, p, p, p, p, None, 0, None) == -1   # C != H*64
    assert L.wkv7_fwd_seq(0, 1, 128, 2, p, p, p, p, p, p, p, p, p, None, 0, None) == -1   # B == 0
    assert L.wkv7_fwd_seq(1, 1, 128, 2, None, p, p, p, p, p, p, p, p, None, 0, None) == -2  # NULL state
    assert L.wkv7_fwd_seq(1, 1, 128, 2, ctypes.c_void_p(4098), p, p, p, p, p, p, p, p, None, 0, None) == -3
    assert L.wkv7_fwd_seq(1, 1, 128, 2, p, p, p, p, p, p, p, p, p, None, 4, None) == -3     # stride % 8
    assert L.wkv7_fwd_seq(1, 1, 128, 2, p, p, p, p, p, p, p, p, p, None, 4096, None) == -1  # stride < H*4096
    assert L.spmv_fp16(64, 100, p, p, p, p, None) == -1                                      # C % 8
    assert L.mm8_seq(1, 8, 8, p, 4, p, 8, p, p, p, p, p, 8, None, None) == -1               # x_stride < N


def test_ops_fail_loudly_without_gpu_tensors():
    import torch

    from chirrup_amd import ChirrupAmdError, ops

    t = torch.zeros(1, 1, 64, 64, dtype=torch.float16)
    v = torch.zeros(1, 64, dtype=torch.float16)
    with pytest.raises(ChirrupAmdError):
        ops.forward_one(1, 64, 1, t, v, v, v, v, v, v, v.clone(), torch.zeros(1, dtype=torch.int32))


def test_torch_op_registration_uses_reference_names():
    import torch

    from chirrup_amd import ops

    ops.register_torch_ops()
    for name in ("forward_one", "forward_seq", "spmv_forward"):
        assert hasattr(torch.ops.rwkv7_state_fwd_fp16, name)
    for name in ("mm8_seq", "mm8_one", "mm8_seq_opt", "gemm_fp16_cublas"):      # rwkv_pip_wrapper.cpp:206-211: all four
        assert hasattr(torch.ops.rwkv_pip, name)
