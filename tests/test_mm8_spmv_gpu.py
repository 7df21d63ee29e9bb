"""GPU parity of mm8 and the sparse vec x mat kernel against the CPU oracle (C ABI path)."""
import numpy as np
import pytest
import torch

from util import bits

pytestmark = pytest.mark.gpu
F16 = np.float16


def _mm8_case(B, N, M, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, N)).astype(F16)
    w = rng.integers(0, 256, (N, M)).astype(np.uint8)
    mx = (rng.standard_normal(M) * 0.05).astype(F16)
    rx = (rng.uniform(0.5, 1.5, M) / 16).astype(F16)
    my = (rng.standard_normal((N, 1)) * 0.05).astype(F16)
    ry = (rng.uniform(0.5, 1.5, (N, 1)) / 16).astype(F16)
    return x, w, mx, rx, my, ry


@pytest.mark.parametrize("B,N,M", [(1, 64, 256), (5, 96, 80), (16, 256, 512), (33, 300, 700), (200, 512, 1024)])
def test_mm8_seq_direct_kernel_bit_exact(oracle, B, N, M):
    """The direct kernel evaluates the as-coded expression in the oracle's order -> bit-exact."""
    from chirrup_amd import ops

    x, w, mx, rx, my, ry = _mm8_case(B, N, M, seed=B + N + M)
    want = oracle.mm8_seq(x, w, mx, rx, my, ry)
    t = [torch.from_numpy(z).cuda() for z in (x, w, mx, rx, my, ry)]
    y = torch.empty((B, M), dtype=torch.float16, device="cuda")
    ops.mm8_seq(B, N, M, *t, y)
    got = y.cpu().numpy()
    assert np.array_equal(bits(got), bits(want)), f"{(bits(got) != bits(want)).sum()} of {got.size} differ"


@pytest.mark.parametrize("N,M", [(64, 256), (300, 700), (1024, 4096)])
def test_mm8_one_within_reference_tolerance(oracle, N, M):
    """GEMV form adds slices with atomics (order undefined, like the reference); tolerance is the
    reference's own rtol 1e-3 (scripts/test_mm8/benchmark_pure_pytorch.py:92) with atol scaled to
    the row magnitude."""
    from chirrup_amd import ops

    x, w, mx, rx, my, ry = _mm8_case(1, N, M, seed=N + M)
    want = oracle.mm8_one(x[0], w, mx, rx, my, ry)
    t = [torch.from_numpy(z).cuda() for z in (x[0].copy(), w, mx, rx, my, ry)]
    y = torch.zeros((M,), dtype=torch.float32, device="cuda")
    ops.mm8_one(N, M, *t, y)
    got = y.cpu().numpy()
    scale = np.abs(want).max()
    assert np.allclose(got, want, rtol=1e-3, atol=1e-5 * scale)


@pytest.mark.parametrize("D,C,density", [(64, 256, 0.5), (3072, 768, 0.4), (192, 2048 + 64, 0.1), (256, 512, 0.0), (16384, 4096, 0.3)])
def test_spmv_vs_oracle(oracle, D, C, density):
    """fp32 accumulate on both sides, different association (per-64-row chunk) -> results may differ
    by one binary16 ulp on a few outputs; never more."""
    from chirrup_amd import ops

    rng = np.random.default_rng(D + C)
    vec = (rng.standard_normal(D) * (rng.uniform(size=D) < density)).astype(F16)
    if D > 5:
        vec[5] = F16(-0.0)
    mat = (rng.standard_normal((D, C)) / np.sqrt(max(D * max(density, 0.01), 1))).astype(F16)
    want = oracle.spmv(vec, mat)
    out = torch.zeros((C,), dtype=torch.float16, device="cuda")
    ops.spmv_forward(D, C, torch.from_numpy(vec).cuda(), torch.from_numpy(mat).cuda(), out)
    got = out.cpu().numpy()
    ulp = np.abs(bits(got).astype(np.int32) - bits(want).astype(np.int32))
    same_sign = (bits(got) & 0x8000) == (bits(want) & 0x8000)
    scale = float(np.abs(want.astype(np.float32)).max()) if want.size else 0.0
    big = (np.abs(want.astype(np.float32)) > 0.05 * scale) & same_sign & (scale > 0)
    assert ulp[big].max(initial=0) <= 1
    assert (ulp[big] == 1).mean() < 0.02 if big.any() else True
    # outputs that nearly cancel are only as good as binary32 accumulation of ~D terms
    assert np.allclose(got.astype(np.float32), want.astype(np.float32), rtol=1e-3, atol=1e-3 * scale + 1e-7)
    # accumulate semantics: a second call adds again
    ops.spmv_forward(D, C, torch.from_numpy(vec).cuda(), torch.from_numpy(mat).cuda(), out)
    assert np.allclose(out.cpu().numpy().astype(np.float32), 2 * want.astype(np.float32), rtol=4e-3, atol=2e-3 * scale + 1e-7)
    # Triton-surface wrapper returns a fresh tensor
    z = ops.rwkv_mm_sparsity(torch.from_numpy(vec).cuda(), torch.from_numpy(mat).cuda())
    assert np.array_equal(bits(z.cpu().numpy()), bits(got))
