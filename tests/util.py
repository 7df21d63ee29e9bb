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


def parity_stats(got, want):
    """The error figures the parity bars are written in: absolute L-inf, L-inf relative to max(1, max|want|) (the bar form of
    DESIGN.md section 2), max|want|, and the absolute error in units in the last place of binary16 at the tensor's TOP binade."""
    g, w = np.asarray(got, dtype=np.float32), np.asarray(want, dtype=np.float32)
    d = float(np.abs(g - w).max()) if w.size else 0.0
    m = float(np.abs(w).max()) if w.size else 0.0
    top_ulp = float(2.0 ** (np.floor(np.log2(max(m, 2.0 ** -14))) - 10))
    return {"abs_linf": d, "rel_linf": d / max(1.0, m), "max_abs_want": m, "top_binade_ulps": d / top_ulp}


def record_parity(case, **fields):
    """Append one measured-error record to the parity log (JSON lines; gpurun_out/parity/r04_parity_errors.jsonl, or
    $CHIRRUP_PARITY_LOG): VERDICT r3 item 2 -- every bar of the fixture / full-size / smoke tests sits beside the error that was
    measured against it; tools/parity_report.py turns the log into profiles/r04_parity_errors.txt."""
    import json
    import os

    path = os.environ.get("CHIRRUP_PARITY_LOG") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                "gpurun_out", "parity", "r04_parity_errors.jsonl")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"case": case, **fields}, sort_keys=True) + "\n")
    except OSError:
        pass


_BARS = None


def tight_bar(case, name, ceiling):
    """The bar a measurement is held to: tests/golden/parity_bars.json holds, per (case, tensor), 1.25 x the error measured on an
    MI355X with this round's kernels (written by `python tools/parity_report.py --write-bars` from the log of a full -m gpu run;
    the same log is committed as profiles/r04_parity_errors.txt) -- never above `ceiling`, the bar the test's own argument gives
    (north_star's 1e-3, a noise floor of two CPU evaluations, ...).  VERDICT r3 item 2: every bar <= 1.25 x its measurement."""
    global _BARS
    if _BARS is None:
        import json
        import os

        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_bars.json")
        _BARS = json.load(open(path)) if os.path.exists(path) else {}
    b = _BARS.get(f"{case}|{name}")
    return ceiling if b is None else min(ceiling, b)


def check_bar(case, name, stats, bar, key="rel_linf"):
    """Record (measurement, bar) and assert measurement <= bar (bar = tight_bar(...): 1.25 x the committed measurement, at most `bar`)."""
    bar = tight_bar(case, name, bar)
    record_parity(case, tensor=name, bar=bar, bar_on=key, **stats)
    assert stats[key] <= bar, (case, name, key, stats[key], bar)
