"""Pin the WKV7 restatement (oracle/oracle.c) against the REFERENCE's own HIP kernel, compiled in the
build container from /root/reference/Albatross/hip/rwkv7_state_fwd_fp16.hip where it lies
(oracle/Makefile target `ref`; outputs in oracle/_ref/, git-ignored, shipped to the GPU box).

  strict build (-ffp-contract=off, no fast-math): must equal the oracle bit for bit wherever the
      result does not depend on the device's exp2f (w = -60000 makes the decay exactly the dither),
      and differ only through 1-ulp decay differences otherwise;
  fast build (the reference's own flags, rwkv7.py:51-52): FMA contraction + approximate exp2/div --
      reported, held to a loose bound (this is how far the reference is from its own source order).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from util import bits, wkv7_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEQ = "_Z16cuda_forward_seqiiiiPN3c104HalfES1_S1_S1_S1_S1_S1_S1_Pi"


def _ref(kind):
    path = os.path.join(ROOT, "oracle", "_ref", f"libref_wkv7_{kind}.so")
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (needs the reference tree: make -C oracle ref)")
    lib = ctypes.CDLL(path)
    fn = getattr(lib, SEQ)
    fn.restype = None
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 9
    return fn


def _run_ref(fn, state, r, w, k, v, a, b, et):
    B, T, C = r.shape
    ts = [torch.from_numpy(x.copy()).cuda() for x in (state, r, w, k, v, a, b)]
    te = torch.from_numpy(et).cuda()
    y = torch.empty((B, T, C), dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    fn(B, T, C, C // 64, *[ctypes.c_void_p(t.data_ptr()) for t in ts], ctypes.c_void_p(y.data_ptr()),
       ctypes.c_void_p(te.data_ptr()))                      # launches on the null stream (.hip:343)
    torch.cuda.synchronize()
    return y.cpu().numpy(), ts[0].cpu().numpy()


def test_strict_reference_kernel_equals_oracle_bitwise_when_decay_is_exact(oracle):
    fn = _ref("strict")
    B, T, C = 6, 3, 256
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=3, elapsed="big")
    w[:] = np.float16(-60000.0)          # sigmoid -> 0: w~ = dither exactly, whatever exp2f does
    S = state.copy()
    y_o = oracle.wkv7_seq(S, r, w, k, v, a, b, et)
    y_r, S_r = _run_ref(fn, state, r, w, k, v, a, b, et)
    assert np.array_equal(bits(y_r), bits(y_o))
    assert np.array_equal(bits(S_r), bits(S))


def test_strict_reference_kernel_vs_oracle_general_inputs(oracle):
    """General decay inputs: the only freedom left is the device exp2f (<= 1 ulp in binary32), which
    can move w~ by one binary16 ulp on a few channels."""
    fn = _ref("strict")
    B, T, C = 8, 1, 512
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=4)
    S = state.copy()
    y_o = oracle.wkv7_seq(S, r, w, k, v, a, b, et)
    y_r, S_r = _run_ref(fn, state, r, w, k, v, a, b, et)
    frac = float((bits(S_r) != bits(S)).mean())
    d = np.abs(S_r.astype(np.float32) - S.astype(np.float32))
    print(f"strict ref vs oracle: {frac:.5f} of state elements differ, max |d| {d.max():.3e}")
    assert frac < 0.02
    assert d.max() <= 2 ** -9 * max(1.0, float(np.abs(S.astype(np.float32)).max()))
    assert np.abs(y_r.astype(np.float32) - y_o.astype(np.float32)).max() <= 0.05


def test_fast_math_reference_kernel_stays_within_tolerance_of_oracle_and_of_our_kernel(oracle):
    from chirrup_amd import ops

    fn = _ref("fast")
    B, T, C = 8, 4, 512
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=5)
    S = state.copy()
    y_o = oracle.wkv7_seq(S, r, w, k, v, a, b, et)
    y_r, S_r = _run_ref(fn, state, r, w, k, v, a, b, et)
    tS = torch.from_numpy(state.copy()).cuda()
    ts = [torch.from_numpy(x).cuda() for x in (r, w, k, v, a, b)]
    y = torch.empty((B, T, C), dtype=torch.float16, device="cuda")
    ops.forward_seq(B, T, C, C // 64, tS, *ts, y, torch.from_numpy(et).cuda())
    assert np.array_equal(bits(tS.cpu().numpy()), bits(S))                       # ours == oracle, as always
    scale = max(1.0, float(np.abs(S.astype(np.float32)).max()))
    d = np.abs(S_r.astype(np.float32) - S.astype(np.float32)).max() / scale
    print(f"fast-math reference (as built, rwkv7.py:51-52) vs oracle and vs our kernel: state rel-Linf {d:.3e} after T={T}")
    # The reference AS BUILT (-ffast-math: FMA contraction, approximate exp2 / division) sits this far from its own
    # source order after 4 tokens; measured 1.1e-3 on gfx950.  north_star's 1e-3 is held against the strict build
    # (bit-exact, tests above) because the reference's CUDA and HIP fast-math builds already differ from each other
    # by this much -- there is no single "as built" result to match.  The bound is the measured distance with
    # headroom for another compiler's contraction choices, not a loose sanity bar.
    assert d <= 2e-3
    # one decode step (T = 1), the regime north_star's tolerance is quoted for: within 1e-3
    state1, r1, w1, k1, v1, a1, b1, et1 = wkv7_inputs(B, 1, C, seed=6)
    S1 = state1.copy()
    oracle.wkv7_seq(S1, r1, w1, k1, v1, a1, b1, et1)
    _, S1_r = _run_ref(fn, state1, r1, w1, k1, v1, a1, b1, et1)
    d1 = np.abs(S1_r.astype(np.float32) - S1.astype(np.float32)).max() / max(1.0, float(np.abs(S1.astype(np.float32)).max()))
    print(f"fast-math reference vs oracle, one decode step: state rel-Linf {d1:.3e}")
    assert d1 <= 1e-3
