"""ctypes bindings of oracle/liboracle.so (the C restatement in oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker / timed CPU baseline.  The product package (chirrup_amd)
never imports anything from here.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle.c with gcc (make -C oracle).  Returns the library path."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("oracle.c", "wkv7_core.inc"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.oracle_wkv7_seq.argtypes = [i32] * 4 + [vp] * 10 + [i64, i32]
        L.oracle_wkv7_seq.restype = None
        L.oracle_wkv7_seq_f32acc.argtypes = [i32] * 4 + [vp] * 9
        L.oracle_wkv7_seq_f32acc.restype = None
        L.oracle_mm8_seq.argtypes = [i32, i32, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32]
        L.oracle_mm8_seq.restype = None
        L.oracle_mm8_one.argtypes = [i32, i32, vp, vp, i32, vp, vp, vp, vp, vp]
        L.oracle_mm8_one.restype = None
        L.oracle_spmv.argtypes = [i32, i32, vp, vp, vp]
        L.oracle_spmv.restype = None
        L.oracle_f2h_sw.argtypes = [ctypes.c_float]
        L.oracle_f2h_sw.restype = ctypes.c_uint16
        L.oracle_f2h_hw.argtypes = [ctypes.c_float]
        L.oracle_f2h_hw.restype = ctypes.c_uint16
        L.oracle_h2f_sw.argtypes = [ctypes.c_uint16]
        L.oracle_h2f_sw.restype = ctypes.c_float
        L.oracle_has_f16c.restype = ctypes.c_int
        L.oracle_decay_f32.argtypes = [ctypes.c_float, ctypes.c_int32]
        L.oracle_decay_f32.restype = ctypes.c_float
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _h(a):
    a = np.ascontiguousarray(a)
    assert a.dtype == np.float16, a.dtype
    return a


def wkv7_seq(state, r, w, k, v, a, b, elapsed_t, slot_idx=None, force_sw=False, f32acc=False):
    """state [n_slots,H,64,64] f16 (updated IN PLACE, must be C-contiguous), r..b [B,T,C] f16,
    elapsed_t [B] int32 -> y [B,T,C] f16."""
    assert state.dtype == np.float16 and state.flags["C_CONTIGUOUS"]
    r, w, k, v, a, b = (_h(t) for t in (r, w, k, v, a, b))
    B, T, C = r.shape
    H = C // 64
    assert state.shape[-3:] == (H, 64, 64)
    et = np.ascontiguousarray(elapsed_t, dtype=np.int32)
    y = np.empty((B, T, C), dtype=np.float16)
    if f32acc:
        assert slot_idx is None
        lib().oracle_wkv7_seq_f32acc(B, T, C, H, _p(state), _p(r), _p(w), _p(k), _p(v), _p(a), _p(b), _p(y), _p(et))
        return y
    si = None if slot_idx is None else np.ascontiguousarray(slot_idx, dtype=np.int32)
    lib().oracle_wkv7_seq(B, T, C, H, _p(state), _p(r), _p(w), _p(k), _p(v), _p(a), _p(b), _p(y), _p(et),
                          _p(si), 0, int(force_sw))
    return y


def mm8_seq(x, w, mx, rx, my, ry):
    """x [B,N] f16, w [N,M] u8, mx,rx [M] f16, my,ry [N] or [N,1] f16 -> y [B,M] f16."""
    x = _h(x)
    w = np.ascontiguousarray(w, dtype=np.uint8)
    B, N = x.shape
    M = w.shape[1]
    mx, rx, my, ry = _h(mx).reshape(M), _h(rx).reshape(M), _h(my).reshape(N), _h(ry).reshape(N)
    y = np.empty((B, M), dtype=np.float16)
    lib().oracle_mm8_seq(B, N, M, _p(x), N, _p(w), M, _p(mx), _p(rx), _p(my), _p(ry), _p(y), M)
    return y


def mm8_one(x, w, mx, rx, my, ry):
    """x [N] f16 -> y [M] float32 (the reference's mm8_one writes binary32)."""
    x = _h(x)
    w = np.ascontiguousarray(w, dtype=np.uint8)
    N, M = w.shape
    mx, rx, my, ry = _h(mx).reshape(M), _h(rx).reshape(M), _h(my).reshape(N), _h(ry).reshape(N)
    y = np.zeros((M,), dtype=np.float32)
    lib().oracle_mm8_one(N, M, _p(x), _p(w), M, _p(mx), _p(rx), _p(my), _p(ry), _p(y))
    return y


def spmv(vec, mat, out=None):
    """vec [D] f16, mat [D,C] f16 -> out [C] f16 (adds into `out` when given)."""
    vec, mat = _h(vec), _h(mat)
    D, C = mat.shape
    if out is None:
        out = np.zeros((C,), dtype=np.float16)
    assert out.dtype == np.float16 and out.flags["C_CONTIGUOUS"]
    lib().oracle_spmv(D, C, _p(vec), _p(mat), _p(out))
    return out
