"""GPU parity of the HIP WKV7 kernel against the CPU oracle, through the C ABI.

Bar: BIT-EXACT state and y (the kernel reproduces the reference's binary16 two-lane accumulation
order, spec A1 of SURVEY.md section 8; north_star tolerance is 1e-3, we hold 0)."""
import numpy as np
import pytest
import torch

from util import bits, wkv7_inputs

pytestmark = pytest.mark.gpu


def _run_gpu(state, r, w, k, v, a, b, et, slot_idx=None, one=False, split_decay=False):
    from chirrup_amd import ops

    dev = "cuda:0"
    tS = torch.from_numpy(state.copy()).to(dev)
    tr, tw, tk, tv, ta, tb = (torch.from_numpy(x).to(dev) for x in (r, w, k, v, a, b))
    te = torch.from_numpy(et).to(dev)
    B, T, C = r.shape
    y = torch.empty((B, T, C), dtype=torch.float16, device=dev)
    si = None if slot_idx is None else torch.from_numpy(slot_idx).to(dev)
    if one:
        assert T == 1
        ops.forward_one(B, C, C // 64, tS, tr, tw, tk, tv, ta, tb, y, te, si)
    else:
        ops.forward_seq(B, T, C, C // 64, tS, tr, tw, tk, tv, ta, tb, y, te, si, split_decay=split_decay)
    torch.cuda.synchronize()
    return y.cpu().numpy(), tS.cpu().numpy()


CASES = [
    # B, T, C, elapsed
    (1, 1, 64, "zero"),
    (1, 1, 768, "arange"),      # config 0: 0.1B, bsz 1
    (3, 1, 128, "big"),
    (3, 5, 128, "arange"),      # golden-fixture shape (T>1: chunked prefill path)
    (32, 1, 2048, "arange"),    # config 1: 1.5B bsz 32
    (7, 100, 256, "big"),       # max chunk of the worker's prefill (worker.py:177)
    (200, 1, 4096, "arange"),   # config 2: 7.2B bsz 200, one layer -- full bench size
]


@pytest.mark.parametrize("B,T,C,elapsed", CASES)
def test_wkv7_bit_exact_vs_oracle(oracle, B, T, C, elapsed):
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=B * 1000 + T * 10 + C // 64, elapsed=elapsed)
    S_ref = state.copy()
    y_ref = oracle.wkv7_seq(S_ref, r, w, k, v, a, b, et)
    y, S = _run_gpu(state, r, w, k, v, a, b, et, one=(T == 1 and B % 2 == 1))
    ny = int((bits(y) != bits(y_ref)).sum())
    ns = int((bits(S) != bits(S_ref)).sum())
    assert ny == 0 and ns == 0, f"{ny} y / {ns} state elements differ; max |dS| = " \
        f"{np.abs(S.astype(np.float32) - S_ref.astype(np.float32)).max()}"


@pytest.mark.parametrize("B,T,C,elapsed", [(7, 100, 256, "big"), (3, 5, 128, "arange"), (25, 40, 512, "big"), (2, 1, 64, "zero")])
def test_wkv7_decay_outside_the_scan_is_bit_exact(oracle, B, T, C, elapsed):
    """Chunked prefill: the per-token decay w~ (two exp2 per channel, dither of elapsed + t) computed for all rows by one
    row-parallel launch and the scan given w~ -- the same device function on the same inputs, so state and y keep the
    oracle's bits; slot indirection included."""
    n_slots = B + 3
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=B * 1000 + T * 10 + C // 64 + 1, elapsed=elapsed, n_slots=n_slots)
    idx = np.random.default_rng(B + T).permutation(n_slots)[:B].astype(np.int32)
    S_ref = state.copy()
    y_ref = oracle.wkv7_seq(S_ref, r, w, k, v, a, b, et, slot_idx=idx)
    y, S = _run_gpu(state, r, w, k, v, a, b, et, slot_idx=idx, split_decay=True)
    assert np.array_equal(bits(y), bits(y_ref)) and np.array_equal(bits(S), bits(S_ref))


def test_wkv7_slot_pool_indirection(oracle):
    B, T, C, n_slots = 5, 2, 256, 9
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=77, n_slots=n_slots)
    idx = np.array([8, 2, 5, 0, 3], np.int32)
    S_ref = state.copy()
    y_ref = oracle.wkv7_seq(S_ref, r, w, k, v, a, b, et, slot_idx=idx)
    y, S = _run_gpu(state, r, w, k, v, a, b, et, slot_idx=idx)
    assert np.array_equal(bits(y), bits(y_ref))
    assert np.array_equal(bits(S), bits(S_ref))       # includes the untouched slots


def test_wkv7_edge_values(oracle):
    """Zero state, +-0, subnormal halves, saturating decay inputs, extreme elapsed_t."""
    B, T, C = 4, 2, 128
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=5)
    state[0] = 0
    state[1] = np.float16(6e-8)          # subnormal state
    state[2, :, ::2] = np.float16(-0.0)
    w[0] = np.float16(60000.0)           # sigmoid -> 1
    w[1] = np.float16(-60000.0)          # sigmoid -> 0 : w~ = dither only
    w[2, :, :5] = np.float16(0.0)
    k[3] = 0
    et = np.array([0, 2**31 - 3, 1, 123456789], np.int32)   # elapsed+t wraps the int32 multiply
    S_ref = state.copy()
    y_ref = oracle.wkv7_seq(S_ref, r, w, k, v, a, b, et)
    y, S = _run_gpu(state, r, w, k, v, a, b, et)
    assert np.array_equal(bits(y), bits(y_ref))
    assert np.array_equal(bits(S), bits(S_ref))


def test_wkv7_view_into_layer_pool(oracle):
    """The worker hands state[1][layer][lo:hi] -- a contiguous VIEW of [L, n, H, 64, 64]
    (chirrup/worker.py:697-701); neighbours must stay untouched."""
    from chirrup_amd import ops

    L_, n, C = 3, 6, 128
    H = C // 64
    rng = np.random.default_rng(9)
    pool = (rng.standard_normal((L_, n, H, 64, 64)) * 0.5).astype(np.float16)
    _, r, w, k, v, a, b, et = wkv7_inputs(3, 1, C, seed=10)
    S_ref = pool[1, 2:5].copy()
    y_ref = oracle.wkv7_seq(S_ref, r, w, k, v, a, b, et)
    t_pool = torch.from_numpy(pool.copy()).cuda()
    tr, tw, tk, tv, ta, tb = (torch.from_numpy(x).cuda() for x in (r, w, k, v, a, b))
    y = torch.empty((3, 1, C), dtype=torch.float16, device="cuda")
    ops.forward_seq(3, 1, C, H, t_pool[1][2:5], tr, tw, tk, tv, ta, tb, y, torch.from_numpy(et).cuda())
    out = t_pool.cpu().numpy()
    want = pool.copy()
    want[1, 2:5] = S_ref
    assert np.array_equal(bits(out), bits(want))
    assert np.array_equal(bits(y.cpu().numpy()), bits(y_ref))


def test_wkv7_multi_step_decode_stays_exact(oracle):
    """64 consecutive decode steps on the same state: rounding must not drift from the oracle
    (the dither makes every step's decay differ)."""
    from chirrup_amd import ops

    B, C, steps = 4, 256, 64
    H = C // 64
    state, *_ = wkv7_inputs(B, 1, C, seed=21)
    S_ref = state.copy()
    tS = torch.from_numpy(state.copy()).cuda()
    y = torch.empty((B, 1, C), dtype=torch.float16, device="cuda")
    for s in range(steps):
        _, r, w, k, v, a, b, _ = wkv7_inputs(B, 1, C, seed=1000 + s)
        et = (np.arange(B) * 7 + 3 + s).astype(np.int32)
        y_ref = oracle.wkv7_seq(S_ref, r, w, k, v, a, b, et)
        tr, tw, tk, tv, ta, tb = (torch.from_numpy(x).cuda() for x in (r, w, k, v, a, b))
        ops.forward_one(B, C, H, tS, tr, tw, tk, tv, ta, tb, y, torch.from_numpy(et).cuda())
        assert np.array_equal(bits(y.cpu().numpy()), bits(y_ref)), f"step {s}"
    assert np.array_equal(bits(tS.cpu().numpy()), bits(S_ref))
    assert np.isfinite(S_ref.astype(np.float32)).all()


def test_torch_ops_route_to_hip_kernel(oracle):
    """torch.ops.rwkv7_state_fwd_fp16.forward_seq (the reference's op name) runs our kernel."""
    from chirrup_amd import ops

    ops.register_torch_ops()
    B, T, C = 2, 3, 128
    state, r, w, k, v, a, b, et = wkv7_inputs(B, T, C, seed=31)
    S_ref = state.copy()
    y_ref = oracle.wkv7_seq(S_ref, r, w, k, v, a, b, et)
    tS = torch.from_numpy(state.copy()).cuda()
    tr, tw, tk, tv, ta, tb = (torch.from_numpy(x).cuda() for x in (r, w, k, v, a, b))
    y = torch.empty((B, T, C), dtype=torch.float16, device="cuda")
    torch.ops.rwkv7_state_fwd_fp16.forward_seq(B, T, C, C // 64, tS, tr, tw, tk, tv, ta, tb, y, torch.from_numpy(et).cuda())
    assert np.array_equal(bits(y.cpu().numpy()), bits(y_ref))
    assert np.array_equal(bits(tS.cpu().numpy()), bits(S_ref))
