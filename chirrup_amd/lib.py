"""Loader for libchirrup_amd.so (C ABI declared in include/chirrup_amd.h).

The product path has NO fallback: if the HIP library is missing or fails to load, every op
raises.  torch is imported first on purpose -- the wheel bundles a HIP runtime with the same
SONAME (libamdhip64.so.7) the library needs, so the process ends up with exactly one runtime
and torch's stream handles are valid for our launches.
"""
import ctypes
import os
import subprocess
import threading

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHIRRUP_AMD_LIB: another build of the same ABI (A/B of kernel variants, tools/); default: the in-tree library
LIB_PATH = os.environ.get("CHIRRUP_AMD_LIB") or os.path.join(_HERE, "libchirrup_amd.so")
_lib = None
_lock = threading.Lock()      # worker threads may race to the first load

ABI_VERSION = 4               # CHIRRUP_ABI_VERSION of include/chirrup_amd.h this module's SIGNATURES were written against

E_NAMES = {-1: "CHIRRUP_E_SHAPE", -2: "CHIRRUP_E_NULL", -3: "CHIRRUP_E_ALIGN", -4: "CHIRRUP_E_UNSUPPORTED"}

# name -> (restype, argtypes); mirrors include/chirrup_amd.h one to one
_vp, _i, _i64, _f32p = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
SIGNATURES = {
    "chirrup_abi_version": (_i, []),
    "chirrup_target_arch": (ctypes.c_char_p, []),
    "wkv7_fwd_seq": (_i, [_i, _i, _i, _i] + [_vp] * 10 + [_i64, _vp]),
    "wkv7_fwd_seq_decayed": (_i, [_i, _i, _i, _i] + [_vp] * 10 + [_i64, _vp]),
    "wkv7_decay": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "wkv7_fwd_one": (_i, [_i, _i, _i] + [_vp] * 10 + [_i64, _vp]),
    "spmv_fp16_workspace_bytes": (_i64, [_i, _i]),
    "spmv_fp16": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mm8_seq_workspace_bytes": (_i64, [_i, _i, _i]),
    "mm8_seq": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "mm8_seq_opt": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "mm8_seq_direct": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mm8_packed_bytes": (_i64, [_i, _i]),
    "mm8_pack": (_i, [_i, _i, _vp, _i, _vp, _vp]),
    "mm8_one": (_i, [_i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _f32p, _vp]),
    "rwkv7_add_ln_mix": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _vp]),
    "rwkv7_add_ln_mix_mm8": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _vp, _vp]),
    "mm8_tile_parts": (_i, [_i]),
    "mm8t_gemm_fused": (_i, [_i, _i, _i, _vp, _i, _vp, _i64, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "mm8t_gemm_fused_split": (_i, [_i, _i, _i, _vp, _i, _vp, _i64, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "mm8t_gemm_partial": (_i, [_i, _i, _i, _vp, _i, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "mm8_reduce_rows": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mm8_row_parts": (_i, [_i]),
    "rwkv7_tmix_mid": (_i, [_i64, _i] + [_vp] * 9 + [_vp]),
    "rwkv7_tmix_post": (_i, [_i64, _i] + [_vp] * 8 + [ctypes.c_float, _vp, _vp]),
    "rwkv7_relu_sq": (_i, [_i64, _vp, _vp]),
    "rwkv7_tmix_wkv7_fused": (_i, [_i, _i, _i, _i] + [_vp] * 14 + [ctypes.c_float, _vp, _vp, _vp, _i64, _vp]),
    "rwkv7_tmix_wkv7_fused_mm8": (_i, [_i, _i, _i, _i] + [_vp] * 14 + [ctypes.c_float, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "rwkv7_lora_act": (_i, [_i, _i, _i64, _vp, _vp]),
    "rwkv7_penalize_argmax": (_i, [_i, _i] + [_vp] * 8),
    "rwkv7_commit_sampled": (_i, [_i, _i] + [_vp] * 7 + [_i64, _vp, _vp, _vp]),
    "rwkv7_commit_sampled_listed": (_i, [_i, _i] + [_vp] * 7 + [_i64, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "rwkv7_penalize_argmax_listed": (_i, [_i, _i] + [_vp] * 9 + [_i, _vp]),
    "rwkv7_sample_topp": (_i, [_i, _i] + [_vp] * 9),
    "rwkv7_embed_rows": (_i, [_i, _i, _i, _i] + [_vp] * 6 + [_i, _vp, _vp, _vp]),
    "rwkv7_advance_elapsed": (_i, [_i, _i, _vp, _vp, _vp]),
    "rwkv7_copy_slot_rows": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "skinny_gemm_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "skinny_gemm_splits": (_i, [_i, _i, _i, _i]),
    "skinny_gemm_batched_workspace_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "skinny_gemm_f16_batched": (_i, [_i, _i, _i, _i, _vp, _i, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "skinny_gemm_group_workspace_bytes": (_i64, [_i, _vp, _i, _i, _i]),
    "skinny_gemm_f16_group": (_i, [_i, _vp, _i, _i, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "skinny_gemm_pair_counters": (_i, []),
    "rwkv7_tmix_sync_words": (_i, []),
    "rwkv7_tmix_status_word": (_i, []),
    "rwkv7_tmix_gemms_workspace_bytes": (_i64, [_i, _i, _i, _vp, _i, _vp, _i]),
    "rwkv7_tmix_gemms_mm8": (_i, [_i, _i, _i, _i64, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "rwkv7_tmix_gemms": (_i, [_i, _i, _i, _i64, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "chirrup_device_cu_count": (_i, []),
    "skinny_gemm_clock_probe": (_i, [_vp, _i]),
    "skinny_gemm_warm_probe": (_i, [_i, _vp]),
    "chirrup_clock_probe": (_i, [_i, _vp, _vp]),
    "rwkv7_ln_probe": (_i, [_vp]),
    "chirrup_noop_launch": (_i, [_i, _i, _i, _i, _vp, _vp]),
    "skinny_gemm_f16_grouped": (_i, [_i, _i, _i, _i, _vp, _vp, _i, _i64, _vp, _i64, _i64, _i, _vp, _i64, _vp, _i, _i64, _i, _i, _i, _vp, _vp]),
    "skinny_gemm_f16_partial": (_i, [_i, _i, _i, _vp, _i, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "skinny_tile_weight": (_i, [_i, _i, _vp, _i64, _vp, _vp]),
    "skinny_untile_weight": (_i, [_i, _i, _vp, _vp, _i64, _vp]),
    "skinny_gemm_f16": (_i, [_i, _i, _i, _vp, _i, _vp, _i64, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mm8t_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "mm8t_seq": (_i, [_i, _i, _i, _vp, _i, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "mm8_dequant_f16": (_i, [_i, _i, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mm8t_exact_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "mm8t_seq_exact": (_i, [_i, _i, _i, _vp, _i, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "skinny_tile_weight_u8": (_i, [_i, _i, _vp, _i64, _vp, _vp]),
}


class ChirrupAmdError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    """Return the ctypes handle; raises ChirrupAmdError when the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ChirrupAmdError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C chirrup_amd/csrc). There is no CPU fallback.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise ChirrupAmdError(f"cannot load {LIB_PATH}: {e}") from e
        L.chirrup_abi_version.restype = ctypes.c_int
        have = L.chirrup_abi_version()
        if have != ABI_VERSION:       # a stale build: its entry points would read `stream` where the workspace now stands
            raise ChirrupAmdError(f"{LIB_PATH} has ABI version {have}, this package needs {ABI_VERSION}: rebuild it "
                                  "(make -C chirrup_amd/csrc)")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise ChirrupAmdError(f"{what}: {E_NAMES.get(rc, rc)}")
    raise ChirrupAmdError(f"{what}: HIP error {rc}")
