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


def c768_fixture():
    """(fixture, checkpoint dict) of tests/golden/model_L3_C768.npz: outputs of the REFERENCE's
    forward_seq_batch_seperate (tests/golden/make_golden.py::gen_model_c768) on seed-generated weights that are
    rebuilt here and proven identical through the stored sha256."""
    import hashlib
    import os

    from chirrup_amd.synth import make_state_dict

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_L3_C768.npz"))
    L, C, V, B, seed = (int(v) for v in d["config"])
    zd = make_state_dict(L, C, V, seed=seed, varied_norms=True)
    h = hashlib.sha256()
    for k in sorted(zd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(zd[k].numpy()).tobytes())
    assert h.hexdigest() == d["weights_sha256"].tobytes().decode(), "synthetic weights differ from the fixture's (torch RNG changed?)"
    return d, zd


def c768_inputs(d, tag):
    """The seeded inputs of case `tag` (state tensors are regenerated, not stored)."""
    L, C, V, B, _ = (int(v) for v in d["config"])
    rng = np.random.default_rng(int(d[f"{tag}:seed"][0]))
    s0 = (rng.standard_normal((L, 2, B, C)) * 0.5).astype(np.float16)
    s1 = (rng.standard_normal((L, B, C // 64, 64, 64)) * 0.1).astype(np.float16)
    s2 = (np.arange(B) * 7 + 3).astype(np.int32)
    toks = rng.integers(1, V, size=d[f"{tag}:tokens"].shape).tolist()
    assert toks == d[f"{tag}:tokens"].tolist()
    return toks, [s0, s1, s2]
