"""CPU tests of the oracle itself (no GPU, no reference access)."""
import numpy as np
import pytest

from util import bits, wkv7_inputs

F16 = np.float16


def test_half_conversions_match_numpy(oracle):
    L = oracle.lib()
    allh = np.arange(65536, dtype=np.uint16)
    ref = allh.view(F16).astype(np.float32)
    for h in range(0, 65536, 3):
        x = L.oracle_h2f_sw(h)
        assert (np.isnan(x) and np.isnan(ref[h])) or np.float32(x) == ref[h]
    hh = np.arange(0, 0x7C00, dtype=np.uint16).view(F16).astype(np.float64)
    mids = ((hh[:-1] + hh[1:]) / 2).astype(np.float32)
    rng = np.random.default_rng(1)
    xs = np.concatenate([mids[::5], -mids[::7], np.nextafter(mids[::11], np.float32(1e9)),
                         rng.standard_normal(3000).astype(np.float32) * 1e-6,
                         rng.standard_normal(3000).astype(np.float32) * 300, np.float32([0, -0.0, 65519.9, 65520, 1e9])])
    with np.errstate(over="ignore"):
        want = xs.astype(F16).view(np.uint16)
    for x, r in zip(xs, want):
        assert L.oracle_f2h_sw(float(x)) == r
        assert L.oracle_f2h_hw(float(x)) == r


def _wkv7_numpy(state, r, w, k, v, a, b, et):
    """Second, independent restatement of spec A1 in numpy binary16 arithmetic (numpy rounds every
    float16 operation once).  Follows Albatross/cuda/rwkv7_state_fwd_fp16.cu:56-82; vectorised over
    rows i, sequential over the 32 half2 columns exactly like the reference's unrolled loops."""
    B, T, C = r.shape
    H = C // 64
    y = np.zeros((B, T, C), F16)
    S = state.copy()
    for bb in range(B):
        for t in range(T):
            x = np.int64(et[bb]) + t
            prod = np.int64(np.uint32((np.uint64(2654435769) * np.uint64(x & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)))
            if prod >= 2**31:
                prod -= 2**32
            d = np.float32(4.547473508864641e-13) * np.float32(prod)
            for h in range(H):
                sl = slice(h * 64, h * 64 + 64)
                wf = w[bb, t, sl].astype(np.float32)
                e1 = np.exp2((np.float32(-1.4426950408889634) * wf).astype(np.float64)).astype(np.float32)
                q = np.float32(-0.8750387749145276) / (np.float32(1.0) + e1)
                e2 = np.exp2(q.astype(np.float64)).astype(np.float32)
                wt = ((e2 - np.float32(1.0)) + d).astype(F16)
                Sh = S[bb, h]                      # [64 rows i, 64 cols j]
                aa, kk_, bb_, rr = a[bb, t, sl], k[bb, t, sl], b[bb, t, sl], r[bb, t, sl]
                vv = v[bb, t, sl]                  # per row
                sa2 = np.zeros((64, 2), F16)
                for p in range(32):
                    sa2 = sa2 + aa[2 * p:2 * p + 2][None, :] * Sh[:, 2 * p:2 * p + 2]
                sa = sa2[:, 0] + sa2[:, 1]
                y2 = np.zeros((64, 2), F16)
                for p in range(32):
                    c = slice(2 * p, 2 * p + 2)
                    s = Sh[:, c]
                    s = s + ((s * wt[c][None, :] + kk_[c][None, :] * vv[:, None]) + sa[:, None] * bb_[c][None, :])
                    Sh[:, c] = s
                    y2 = y2 + s * rr[c][None, :]
                y[bb, t, sl] = y2[:, 0] + y2[:, 1]
    return y, S


@pytest.mark.parametrize("B,T,C,elapsed", [(1, 1, 64, "arange"), (2, 3, 128, "big"), (3, 1, 192, "zero")])
def test_wkv7_c_oracle_equals_numpy_restatement(oracle, B, T, C, elapsed):
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=B * 100 + T, elapsed=elapsed)
    y_np, S_np = _wkv7_numpy(state, r, w, k, v, a, b, et)
    S_c = state.copy()
    y_c = oracle.wkv7_seq(S_c, r, w, k, v, a, b, et)
    assert np.array_equal(bits(y_c), bits(y_np))
    assert np.array_equal(bits(S_c), bits(S_np))
    S_sw = state.copy()
    y_sw = oracle.wkv7_seq(S_sw, r, w, k, v, a, b, et, force_sw=True)
    assert np.array_equal(bits(y_sw), bits(y_c)) and np.array_equal(bits(S_sw), bits(S_c))


def test_wkv7_seq_equals_repeated_one(oracle):
    """T-step scan == T single steps with elapsed_t advanced (the reference's _seq vs _one kernels)."""
    B, T, C = 2, 4, 128
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=7)
    S1 = state.copy()
    y1 = oracle.wkv7_seq(S1, r, w, k, v, a, b, et)
    S2 = state.copy()
    for t in range(T):
        sl = slice(t, t + 1)
        yt = oracle.wkv7_seq(S2, r[:, sl], w[:, sl], k[:, sl], v[:, sl], a[:, sl], b[:, sl], et + t)
        assert np.array_equal(bits(yt[:, 0]), bits(y1[:, t]))
    assert np.array_equal(bits(S1), bits(S2))


def test_wkv7_slot_indirection(oracle):
    B, T, C = 3, 2, 128
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=11, n_slots=6)
    idx = np.array([4, 0, 3], np.int32)
    dense = np.ascontiguousarray(state[idx])
    y_d = oracle.wkv7_seq(dense, r, w, k, v, a, b, et)
    pool = state.copy()
    y_p = oracle.wkv7_seq(pool, r, w, k, v, a, b, et, slot_idx=idx)
    assert np.array_equal(bits(y_d), bits(y_p))
    assert np.array_equal(bits(pool[idx]), bits(dense))
    untouched = [s for s in range(6) if s not in idx]
    assert np.array_equal(bits(pool[untouched]), bits(state[untouched]))


def test_decay_range_and_dither(oracle):
    """1 + w~ stays in (0.545, 1) and |dither| <= 2^-10 (SURVEY spec A1)."""
    L = oracle.lib()
    for wf in (-60000.0, -20.0, -1.0, 0.0, 3.0, 20.0, 60000.0):
        for e in (0, 1, 12345, 2**31 - 1):
            d = L.oracle_decay_f32(wf, e)
            assert -0.4561 < d < 0.001


def test_mm8_oracle_matches_direct_formula(oracle):
    """oracle_mm8_seq / _one against the float64 value of the defining formula
    (scripts/test_mm8/benchmark.py:114-118)."""
    rng = np.random.default_rng(3)
    B, N, M = 5, 96, 80
    x = rng.standard_normal((B, N)).astype(F16)
    w = rng.integers(0, 256, (N, M)).astype(np.uint8)
    mx = (rng.standard_normal(M) * 0.1).astype(F16)
    rx = (rng.uniform(0.5, 1.5, M) / 16).astype(F16)
    my = (rng.standard_normal((N, 1)) * 0.1).astype(F16)
    ry = (rng.uniform(0.5, 1.5, (N, 1)) / 16).astype(F16)
    dq = (w.astype(np.float64) + 0.5) * ry.astype(np.float64) * rx.astype(np.float64) + my.astype(np.float64) + mx.astype(np.float64)
    want = x.astype(np.float64) @ dq
    y = oracle.mm8_seq(x, w, mx, rx, my, ry).astype(np.float64)
    assert np.allclose(y, want, rtol=2e-3, atol=2e-3)
    y1 = oracle.mm8_one(x[0], w, mx, rx, my, ry).astype(np.float64)
    assert np.allclose(y1, want[0], rtol=1e-5, atol=1e-4)


def test_spmv_oracle(oracle):
    rng = np.random.default_rng(4)
    D, C = 192, 128
    vec = np.maximum(rng.standard_normal(D), 0).astype(F16) ** 2
    vec[5] = F16(-0.0)
    mat = rng.standard_normal((D, C)).astype(F16)
    out = oracle.spmv(vec, mat)
    want = vec.astype(np.float64) @ mat.astype(np.float64)
    assert np.allclose(out.astype(np.float64), want, rtol=2e-3, atol=2e-3)
    out2 = oracle.spmv(vec, mat, out=out.copy())  # accumulates
    assert np.allclose(out2.astype(np.float64), 2 * want, rtol=4e-3, atol=4e-3)


def test_blas_mm8_and_split_accumulation_agree_with_the_as_coded_oracle(oracle):
    """The two helpers the full-size GPU tests lean on (oracle/rwkv7_np.py): mm8_seq_blas = the as-coded mm8 product
    with the j sum left to BLAS, and ACC_SPLIT = the same matmuls as ordered partial sums.  Both are the reference
    arithmetic in another summation order: results equal oracle_mm8_seq / the plain evaluation to binary16 rounding
    (at most one ulp on a few elements)."""
    from oracle import rwkv7_np as M

    rng = np.random.default_rng(9)
    B, N, Mo = 7, 256, 192
    w16 = (rng.standard_normal((N, Mo)) / 16).astype(F16)
    q, mx, rx, my, ry = M.quantize_weight(w16)
    x = rng.standard_normal((B, N)).astype(F16)
    want = oracle.mm8_seq(x, q, mx, rx, my, ry)
    got = M.mm8_seq_blas(x, q, mx, rx, my, ry)
    d = np.abs(got.astype(np.float32) - want.astype(np.float32))
    assert d.max() <= 2e-3 * max(1.0, float(np.abs(want.astype(np.float32)).max())) and (got != want).mean() < 0.02
    old = M.set_accumulation_split(4)
    try:
        got4 = M.mm8_seq_blas(x, q, mx, rx, my, ry)
        lin4 = M.linear(x, w16.T.copy())
    finally:
        M.set_accumulation_split(old)
    lin1 = M.linear(x, w16.T.copy())
    assert (got4 != want).mean() < 0.02 and (lin4 != lin1).mean() < 0.02
    assert np.abs(lin4.astype(np.float32) - lin1.astype(np.float32)).max() <= 2e-3
