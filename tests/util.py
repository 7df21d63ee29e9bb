"""Shared input generators for the parity tests (seeded, no reference access at run time)."""
import numpy as np


def wkv7_inputs(B, T, C, seed=0, n_slots=None, state_scale=0.5, elapsed="arange"):
    """Inputs shaped like what RWKV_x070_TMix_seq_batch hands the kernel (rwkv7.py:625-645):
    a = -kk with kk unit-norm per head, b = kk * sigmoid-range gate, w a pre-activation."""
    rng = np.random.default_rng(seed)
    H = C // 64
    n_slots = B if n_slots is None else n_slots
    f16 = np.float16
    state = (rng.standard_normal((n_slots, H, 64, 64)) * state_scale).astype(f16)
    r = rng.standard_normal((B, T, C)).astype(f16)
    k = rng.standard_normal((B, T, C)).astype(f16)
    v = rng.standard_normal((B, T, C)).astype(f16)
    w = rng.uniform(-8.0, 4.0, (B, T, C)).astype(f16)
    kk = rng.standard_normal((B, T, H, 64)).astype(np.float32)
    kk /= np.linalg.norm(kk, axis=-1, keepdims=True)
    gate = rng.uniform(0.0, 1.0, (B, T, H, 64)).astype(np.float32)
    a = (-kk).reshape(B, T, C).astype(f16)
    b = (kk * gate).reshape(B, T, C).astype(f16)
    if elapsed == "arange":
        et = (np.arange(B) * 7 + 3).astype(np.int32)
    elif elapsed == "big":
        et = rng.integers(0, 2**31 - 1 - T, size=B).astype(np.int32)
    else:
        et = np.zeros(B, np.int32)
    return state, r, w, k, v, a, b, et


def bits(x):
    return np.ascontiguousarray(x).view(np.uint16)
