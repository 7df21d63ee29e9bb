"""Skinny-M MFMA GEMM (fp16 and mm8/u8 weights) against fp64 math and the mm8 oracle."""
import numpy as np
import pytest
import torch

from oracle import rwkv7_np as M_

pytestmark = pytest.mark.gpu


def _ref(x, w, bias=None):
    y = x.double() @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    return y


@pytest.mark.parametrize("M,N,K,splits,bias,act", [
    (1, 128, 64, 1, False, 0), (7, 96, 128, 1, True, 0), (33, 480, 4096, 0, True, 0), (200, 4096, 4096, 0, False, 0),
    (200, 4096, 4096, 1, False, 0), (200, 16384, 4096, 0, False, 1), (200, 4096, 16384, 8, False, 0),
    (256, 132, 192, 3, True, 1), (64, 4096, 128, 1, True, 0), (200, 4096, 480 + 32, 0, True, 0)])
def test_skinny_linear_matches_fp64(M, N, K, splits, bias, act):
    from chirrup_amd import ops

    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda").half() if bias else None
    y = ops.skinny_linear(x, w, b, act=act, splits=splits)
    want = _ref(x, w, b)
    if act:
        want = torch.relu(want.half().double()) ** 2
    err = (y.double() - want).abs()
    # fp16 output rounding (+ fp32 accumulation); relu^2 doubles the relative error of its fp16 input
    tol = (4e-3 if act else 2e-3) * want.abs().clamp_min(1.0)
    assert bool((err <= tol).all()), float((err / want.abs().clamp_min(1.0)).max())
    # row-strided views (the model passes planes of packed tensors)
    wbig = torch.zeros(N, K + 64, device="cuda", dtype=torch.float16)
    wbig[:, :K] = w
    y2 = ops.skinny_linear(x, wbig[:, :K], b, act=act, splits=splits)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("B,N,M,splits,act", [(4, 256, 512, 1, 0), (3, 512, 128, 2, 0), (200, 4096, 1024, 0, 0),
                                              (200, 1024, 4096, 0, 1), (33, 320, 700 // 4 * 4, 1, 0),
                                              (600, 512, 256, 0, 1)])       # > 256 rows: 256-row blocks (prefill chunks)
def test_mm8t_matches_oracle(oracle, B, N, M, splits, act):
    """MFMA mm8 vs the as-coded oracle.  The split form rounds xs = x*ry to fp16 once (the reference's
    own Albatross decomposition does the same, benchmark.py:167), so the bar is the reference's stated
    rtol 1e-3 against the row scale, not bit equality."""
    from chirrup_amd import ops

    rng = np.random.default_rng(B + N + M)
    x = rng.standard_normal((B, N)).astype(np.float16)
    w16 = (rng.standard_normal((N, M)) / np.sqrt(N)).astype(np.float16)
    q, mx, rx, my, ry = M_.quantize_weight(w16)
    want = oracle.mm8_seq(x, q, mx, rx, my, ry).astype(np.float32)
    if act:
        want = np.maximum(want.astype(np.float16).astype(np.float32), 0) ** 2
    qT = np.ascontiguousarray(q.T)
    y = ops.mm8t_linear(torch.from_numpy(x).cuda(), torch.from_numpy(qT).cuda(), torch.from_numpy(mx).cuda(),
                        torch.from_numpy(rx).cuda(), torch.from_numpy(my.reshape(-1)).cuda(),
                        torch.from_numpy(ry.reshape(-1)).cuda(), act=act, splits=splits)
    got = y.cpu().numpy().astype(np.float32)
    scale = np.abs(want).max()
    assert np.allclose(got, want, rtol=2e-3, atol=2e-3 * scale), float(np.abs(got - want).max() / scale)
    # and the dequantised weights really approximate the fp16 matrix (quantisation error ~ 1/256 of the range)
    dense = x.astype(np.float32) @ w16.astype(np.float32)
    if not act:
        assert np.abs(got - dense).max() <= 0.05 * np.abs(dense).max()


@pytest.mark.parametrize("M,K,splits", [(200, 1024, 4), (64, 1024, 1), (256, 512, 2), (1, 256, 1)])
def test_both_kernels_give_the_same_bits(M, K, splits):
    """The library picks the 256-column kernel (8 compute waves, role-split loaders) for very wide problems (the head:
    N >= 32768) and the 128-column ring kernel otherwise; at the same K split every output element is the same chain of
    MFMAs, so a 512-column slice of a wide problem computed on its own (narrow kernel) must equal the wide launch bit
    for bit -- for binary16 and for uint8 weights."""
    from chirrup_amd import ops

    torch.manual_seed(M + K)
    N = 32768
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda").half()
    for tiled in (False, True):
        wide = ops.skinny_linear(x, ops.tile_weight(w) if tiled else w, b, splits=splits)
        for lo in (0, 512, 32768 - 512):
            ws = w[lo:lo + 512]
            narrow = ops.skinny_linear(x, ops.tile_weight(ws) if tiled else ws, b[lo:lo + 512].contiguous(), splits=splits)
            assert torch.equal(wide[:, lo:lo + 512], narrow), (tiled, lo)
    q = torch.randint(0, 256, (N, K), device="cuda", dtype=torch.uint8)
    mx, rx = torch.randn(N, device="cuda").half() * 0.01, (torch.rand(N, device="cuda") / 16 + 0.01).half()
    my, ry = torch.randn(K, device="cuda").half() * 0.01, (torch.rand(K, device="cuda") / 16 + 0.01).half()
    for tiled in (False, True):
        wide = ops.mm8t_linear(x, ops.tile_weight_u8(q) if tiled else q, mx, rx, my, ry, splits=splits, tiled=tiled)
        qs = q[256:768].contiguous()
        narrow = ops.mm8t_linear(x, ops.tile_weight_u8(qs) if tiled else qs, mx[256:768].contiguous(), rx[256:768].contiguous(), my, ry,
                                 splits=splits, tiled=tiled)
        assert torch.equal(wide[:, 256:768], narrow), tiled


@pytest.mark.parametrize("Z,M,N,K,splits,bias,act", [
    (3, 200, 4096, 4096, 2, False, 0), (3, 200, 4096, 4096, 0, False, 0), (4, 200, 512, 4096, 0, False, 4),
    (3, 200, 512, 4096, 16, False, 5), (4, 200, 4096, 512, 1, True, 0), (4, 33, 256, 128, 2, True, 0), (2, 7, 132, 64, 1, True, 1)])
def test_skinny_bmm_matches_fp64(Z, M, N, K, splits, bias, act):
    """Batched launch (receptance/key/value and the LoRA pairs): every problem against fp64, incl. strided planes
    of larger tensors and the fused LoRA activations (tanh on plane w, sigmoid on plane g)."""
    from chirrup_amd import ops

    torch.manual_seed(Z * M + N + K)
    xbig = torch.randn(Z + 1, M, K, device="cuda").half()
    x = xbig[1:]                                               # planes of a larger tensor, like mixed[2:6]
    w = (torch.randn(Z, N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(Z, 1, N, device="cuda").half() if bias else None
    y = ops.skinny_bmm(x, w, b, act=act, splits=splits)
    want = torch.bmm(x.double(), w.double().transpose(1, 2))
    if b is not None:
        want = want + b.double()
    if act == 1:
        want = torch.relu(want.half().double()) ** 2
    elif act >= 4:
        for z in range(Z):
            plane = z + act - 4
            if plane == 1:
                want[z] = torch.tanh(want[z].half().double())
            elif plane == 3:
                want[z] = torch.sigmoid(want[z].half().double())
    err = (y.double() - want).abs()
    tol = (4e-3 if act == 1 else 2e-3) * want.abs().clamp_min(1.0)
    assert bool((err <= tol).all()), float((err / want.abs().clamp_min(1.0)).max())
    # the same problems one at a time give the same bits when the split count is the same
    if act == 0 and splits > 0:
        for z in range(Z):
            y1 = ops.skinny_linear(x[z], w[z], None if b is None else b[z, 0], splits=splits)
            assert torch.equal(y1, y[z])


def test_grouped_reduction_lengths_skip_zero_padding():
    """k_of: problems zero-padded to a common K give the same bits whether the padding is streamed or not
    (RWKV-7 LoRA up-projections: ranks 128 / 128 / 128 / 512 packed as 512)."""
    from chirrup_amd import ops

    torch.manual_seed(5)
    Z, M, N, K, ks = 4, 200, 4096, 512, [128, 64, 128, 512]
    x = torch.zeros(Z, M, K, device="cuda", dtype=torch.float16)
    w = torch.zeros(Z, N, K, device="cuda", dtype=torch.float16)
    for z, k in enumerate(ks):
        x[z, :, :k] = torch.randn(M, k, device="cuda").half()
        w[z, :, :k] = (torch.randn(N, k, device="cuda") / k ** 0.5).half()
    b = torch.randn(Z, 1, N, device="cuda").half()
    full = ops.skinny_bmm(x, w, b, splits=1)
    x[:, :, :] = torch.where(x == 0, torch.full_like(x, 7.0), x)        # poison what must not be read
    for z, k in enumerate(ks):
        x[z, :, :k] = torch.where(x[z, :, :k] == 7.0, torch.zeros_like(x[z, :, :k]), x[z, :, :k])
    grouped = ops.skinny_bmm(x, w, b, splits=1, k_of=ks)
    assert torch.equal(full, grouped)
    want = torch.bmm(torch.where(x == 7.0, torch.zeros_like(x), x).double(), w.double().transpose(1, 2)) + b.double()
    assert bool(((grouped.double() - want).abs() <= 2e-3 * want.abs().clamp_min(1.0)).all())
    with pytest.raises(Exception):
        ops.skinny_bmm(x, w, b, splits=2, k_of=ks)


def test_grouped_launch_of_unequal_problems():
    """One launch for problems of different N with their own activations (R/K/V + the LoRA down-projections of a
    layer): each against fp64 and, for the plain ones, bit-equal to the single-problem kernel at the same split."""
    from chirrup_amd import ops

    torch.manual_seed(9)
    M, K = 200, 4096
    mixed = torch.randn(6, M, K, device="cuda").half()
    rkv = (torch.randn(3, 4096, K, device="cuda") / K ** 0.5).half()
    lora1 = (torch.randn(4, 512, K, device="cuda") / K ** 0.5).half()
    ranks = [128, 128, 128, 512]
    out_rkv = torch.empty(3, M, 4096, device="cuda", dtype=torch.float16)
    hid = torch.zeros(4, M, 512, device="cuda", dtype=torch.float16)
    acts = [None, "tanh", None, "sigmoid"]
    probs = [(mixed[j], rkv[j], out_rkv[j], None, None) for j in range(3)]
    probs += [(mixed[2 + j], lora1[j, :ranks[j]], hid[j, :, :ranks[j]], None, acts[j]) for j in range(4)]
    ops.skinny_group(probs, splits=2)
    for j in range(3):
        want = mixed[j].double() @ rkv[j].double().t()
        assert bool(((out_rkv[j].double() - want).abs() <= 2e-3 * want.abs().clamp_min(1.0)).all())
        assert torch.equal(out_rkv[j], ops.skinny_linear(mixed[j], rkv[j], splits=2))
    for j in range(4):
        want = (mixed[2 + j].double() @ lora1[j, :ranks[j]].double().t()).half().double()
        want = torch.tanh(want) if acts[j] == "tanh" else (torch.sigmoid(want) if acts[j] == "sigmoid" else want)
        got = hid[j, :, :ranks[j]].double()
        assert bool(((got - want).abs() <= 2e-3 * want.abs().clamp_min(1.0)).all()), j
        assert float(hid[j, :, ranks[j]:].abs().max()) == 0.0 if ranks[j] < 512 else True      # nothing written past N


@pytest.mark.parametrize("M,N,K", [(200, 4096, 4096), (200, 4096, 16384), (33, 128, 64), (130, 512, 128)])
def test_tile_image_weight_layout_gives_the_same_bits(M, N, K):
    """skinny_tile_weight re-lays W as consecutive 16-KiB tile images; every entry point must give bit-identical
    results with the tiled copy (same fragments, same accumulation order -- only the addresses of the loads differ)."""
    from chirrup_amd import ops

    torch.manual_seed(N + K)
    x = torch.randn(M, K, device="cuda").half()
    wbig = torch.zeros(N, K + 64, device="cuda", dtype=torch.float16)
    wbig[:, :K] = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    w = wbig[:, :K]                                              # row-strided source
    wt = ops.tile_weight(w)
    assert wt.shape == (N, K) and wt.data.numel() == N * K
    # the packing itself: tile (g, b), chunk c holds row g*128 + (c >> 3), logical chunk (c & 7) ^ ((row >> 1) & 7)
    img = wt.data.view(N // 128, K // 64, 128, 8, 8)
    nr = torch.arange(128, device="cuda")
    lc = torch.arange(8, device="cuda").view(1, 8) ^ ((nr >> 1) & 7).view(128, 1)
    want = w.contiguous().view(N // 128, 128, K // 64, 8, 8).permute(0, 2, 1, 3, 4)
    want = torch.gather(want, 3, lc.view(1, 1, 128, 8, 1).expand(N // 128, K // 64, 128, 8, 8))
    assert torch.equal(img, want)
    b = torch.randn(N, device="cuda").half()
    for splits, act in ((1, 0), (2, 1), (0, 0)):
        assert torch.equal(ops.skinny_linear(x, w, b, act=act, splits=splits), ops.skinny_linear(x, wt, b, act=act, splits=splits))
    pa, pb = torch.empty(16, M, N, device="cuda"), torch.empty(16, M, N, device="cuda")
    a_, b_ = ops.skinny_linear_partial(x, w, 8, pa), ops.skinny_linear_partial(x, wt, 8, pb)
    assert a_.shape == b_.shape and torch.equal(a_, b_)
    a_, b_ = ops.skinny_linear_partial(x, w, 0, pa), ops.skinny_linear_partial(x, wt, 0, pb)
    assert a_.shape[0] == ops.gemm_splits(N, K)
    assert a_.shape == b_.shape and torch.equal(a_, b_)
    ya, yb = torch.empty(M, N, device="cuda", dtype=torch.float16), torch.empty(M, N, device="cuda", dtype=torch.float16)
    sbig = torch.zeros(64, K + 64, device="cuda", dtype=torch.float16)      # same row stride as w (one ldw per launch)
    sbig[:, :K] = (torch.randn(64, K, device="cuda") / K ** 0.5).half()
    small = sbig[:, :K]
    sa, sb = torch.empty(M, 64, device="cuda", dtype=torch.float16), torch.empty(M, 64, device="cuda", dtype=torch.float16)
    ops.skinny_group([(x, w, ya, None, None), (x, small, sa, None, "tanh")], splits=1)
    ops.skinny_group([(x, wt, yb, None, None), (x, small, sb, None, "tanh")], splits=1)      # tiled and row-major problems mixed
    assert torch.equal(ya, yb) and torch.equal(sa, sb)


def test_u8_tile_image_layout_gives_the_same_bits(oracle):
    from chirrup_amd import ops
    from chirrup_amd.quant import untile_u8

    rng = np.random.default_rng(3)
    B, N, M = 200, 512, 256
    x = torch.from_numpy(rng.standard_normal((B, N)).astype(np.float16)).cuda()
    w16 = (rng.standard_normal((N, M)) / np.sqrt(N)).astype(np.float16)
    q, mx, rx, my, ry = M_.quantize_weight(w16)
    qT = torch.from_numpy(np.ascontiguousarray(q.T)).cuda()
    args = [torch.from_numpy(a.reshape(-1)).cuda() for a in (mx, rx, my, ry)]
    flat = ops.tile_weight_u8(qT)
    assert torch.equal(untile_u8(flat, M, N), qT)
    for act in (0, 1):
        assert torch.equal(ops.mm8t_linear(x, qT, *args, act=act), ops.mm8t_linear(x, flat, *args, act=act, tiled=True))


@pytest.mark.parametrize("M,N,K,bias,act", [(200, 16384, 4096, False, 1), (200, 4096, 4096, True, 0), (33, 132, 256, True, 1),
                                            (47, 128, 128, False, 0), (256, 1000, 512, True, 1), (40, 128, 64, True, 0),
                                            (255, 260, 192, False, 1)])
def test_row_halves_give_the_bits_of_the_whole_rows_launch(M, N, K, bias, act):
    """row_halves: two workgroups per tile and K-slice, one per half of the rows.  A row's sums never depend on the other
    rows, so at the same split count the results -- final values and partial planes -- are bit-identical."""
    from chirrup_amd import ops

    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda").half() if bias else None
    for splits in (1, 2):
        want = ops.skinny_linear(x, w, b, act=act, splits=splits)
        assert torch.equal(ops.skinny_linear(x, w, b, act=act, splits=splits, row_halves=True), want)
        if N % 128 == 0 and K % 64 == 0:
            assert torch.equal(ops.skinny_linear(x, ops.tile_weight(w), b, act=act, splits=splits, row_halves=True), want)
    pa, pb = torch.zeros(4, M, N, device="cuda"), torch.zeros(4, M, N, device="cuda")
    a_, b_ = ops.skinny_linear_partial(x, w, 2, pa), ops.skinny_linear_partial(x, w, 2, pb, row_halves=True)
    assert a_.shape == b_.shape and torch.equal(a_, b_) and float(pb[2:].abs().max()) == 0.0
    # the library's own choice with row halves: a result of the same quality (a different split count, so not the same bits)
    y = ops.skinny_linear(x, w, b, act=act, splits=0, row_halves=True)
    want = _ref(x, w, b)
    if act:
        want = torch.relu(want.half().double()) ** 2
    assert bool(((y.double() - want).abs() <= (4e-3 if act else 2e-3) * want.abs().clamp_min(1.0)).all())


@pytest.mark.parametrize("splits", [0, 1, 2])
def test_row_halves_of_a_grouped_launch(splits):
    """The layer's R/K/V + LoRA down-projection launch (seven problems, tanh / sigmoid on two of them): unsplit, bias and
    activations run in the GEMM epilogue (no partials, no reduce launch) -- the same arithmetic on the same binary32 sums
    as the reduce kernel applies, so whole rows vs row halves and epilogue vs reduce kernel agree bit for bit."""
    from chirrup_amd import ops

    torch.manual_seed(11)
    M, K = 200, 4096
    mixed = torch.randn(6, M, K, device="cuda").half()
    rkv = [ops.tile_weight((torch.randn(4096, K, device="cuda") / K ** 0.5).half()) for _ in range(3)]
    lora1 = (torch.randn(4, 512, K, device="cuda") / K ** 0.5).half()
    bias = torch.randn(4096, device="cuda").half()
    ranks, acts = [128, 128, 128, 512], [None, "tanh", None, "sigmoid"]

    def run(s, halves):
        out_rkv = torch.empty(3, M, 4096, device="cuda", dtype=torch.float16)
        hid = torch.zeros(4, M, 512, device="cuda", dtype=torch.float16)
        probs = [(mixed[j], rkv[j], out_rkv[j], bias if j == 1 else None, None) for j in range(3)]
        probs += [(mixed[2 + j], lora1[j, :ranks[j]], hid[j, :, :ranks[j]], None, acts[j]) for j in range(4)]
        ops.skinny_group(probs, splits=s, row_halves=halves)
        return out_rkv, hid

    if splits == 0:                 # the library's choice with halves is the unsplit launch (206 workgroups)
        want, got = run(1, False), run(0, True)
    else:
        want, got = run(splits, False), run(splits, True)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    for j in range(4):
        ref = (mixed[2 + j].double() @ lora1[j, :ranks[j]].double().t()).half().double()
        ref = torch.tanh(ref) if acts[j] == "tanh" else (torch.sigmoid(ref) if acts[j] == "sigmoid" else ref)
        assert bool(((got[1][j, :, :ranks[j]].double() - ref).abs() <= 2e-3 * ref.abs().clamp_min(1.0)).all()), j


@pytest.mark.parametrize("M", [200, 33, 256, 17, 32])
def test_row_halves_and_tiled_batch_give_the_same_bits(M):
    """The LoRA up-projection launch: per-problem K, tile-image weights, and each problem as two workgroup sets over the
    upper / lower half of the rows -- all bit-identical to the plain batched launch (row m's sums never depend on the
    other rows)."""
    from chirrup_amd import ops

    torch.manual_seed(M)
    Z, N, K, ks = 4, 1024, 512, [128, 128, 64, 512]
    x = torch.zeros(Z, M, K, device="cuda", dtype=torch.float16)
    w = torch.zeros(Z, N, K, device="cuda", dtype=torch.float16)
    for z, k in enumerate(ks):
        x[z, :, :k] = torch.randn(M, k, device="cuda").half()
        w[z, :, :k] = (torch.randn(N, k, device="cuda") / k ** 0.5).half()
    b = torch.randn(Z, 1, N, device="cuda").half()
    want = ops.skinny_bmm(x, w, b, splits=1, k_of=ks)
    wt = ops.tile_weight_batch(w)
    assert torch.equal(ops.skinny_bmm(x, w, b, splits=1, k_of=ks, row_halves=True), want)
    assert torch.equal(ops.skinny_bmm(x, wt, b, splits=1, k_of=ks), want)
    assert torch.equal(ops.skinny_bmm(x, wt, b, splits=1, k_of=ks, row_halves=True), want)
    assert torch.equal(ops.skinny_bmm(x[1:], wt[1:], b[1:], splits=1, k_of=ks[1:], row_halves=True), want[1:])
    out = torch.full((Z, M + 2, N), 3.0, device="cuda", dtype=torch.float16)       # nothing is written past row M
    ops.skinny_bmm(x, wt, b, splits=1, k_of=ks, row_halves=True, out=out[:, :M])
    assert torch.equal(out[:, :M], want) and bool((out[:, M:] == 3.0).all())


@pytest.mark.parametrize("B,K,N,halves,tiled", [(200, 1024, 4096, True, True), (200, 512, 1024, False, False), (33, 256, 384, True, False),
                                                 (131, 256, 1000, True, False), (64, 128, 1004, True, False), (256, 64, 136, True, False)])
def test_mm8_corrections_in_the_gemm_epilogue(B, K, N, halves, tiled):
    """mm8t_gemm_fused = mm8t_gemm_partial (unsplit) + mm8_reduce_rows in one launch: the same core sums and the same
    element arithmetic, so y and the next product's xs come out bit-identical; the next product's row sums are split per
    128-column tile instead of per 1024 columns (fp32 sums in another order)."""
    from chirrup_amd import ops

    torch.manual_seed(B + K + N)
    dev = "cuda"
    x = torch.randn(B, K, device=dev).half()
    q = torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8)
    mx, rx = torch.randn(N, device=dev).half() * 0.01, torch.rand(N, device=dev).half() / 64
    my, ry = torch.randn(K, device=dev).half() * 0.01, torch.rand(K, device=dev).half() / 16
    my2, ry2 = torch.randn(N, device=dev).half() * 0.01, torch.rand(N, device=dev).half() / 16
    xs = (x.float() * ry.float()).half()
    S = torch.stack([xs.float().sum(1), (x.float() * my.float()).sum(1), x.float().sum(1)], 1).contiguous()
    tiled = tiled and N % 128 == 0
    qT = ops.tile_weight_u8(q) if tiled else q
    if N % 8:
        with pytest.raises(Exception):
            ops.mm8t_gemm_fused(xs, qT, N, rx, mx, S, act=1, y=torch.empty(B, N, device=dev, dtype=torch.float16))
        return
    parts = ops.mm8t_gemm_partial(xs, qT, N, 1, torch.empty(1, B, N, device=dev), tiled=tiled)
    y_a, xs_a = torch.empty(B, N, device=dev, dtype=torch.float16), torch.empty(B, N, device=dev, dtype=torch.float16)
    S_a = torch.empty(B, ops.mm8_row_parts(N), 3, device=dev)
    ops.mm8_reduce_rows(parts, rx, mx, S, act=1, y=y_a, nxt=(ry2, my2, xs_a, S_a))
    y_b, xs_b = torch.zeros(B, N, device=dev, dtype=torch.float16), torch.zeros(B, N, device=dev, dtype=torch.float16)
    S_b = torch.zeros(B, ops.mm8_tile_parts(N), 3, device=dev)
    ops.mm8t_gemm_fused(xs, qT, N, rx, mx, S, act=1, y=y_b, nxt=(ry2, my2, xs_b, S_b), tiled=tiled, row_halves=halves)
    assert torch.equal(y_a, y_b) and torch.equal(xs_a, xs_b)
    want = torch.stack([xs_a.double().sum(1), (y_a.double() * my2.double()).sum(1), y_a.double().sum(1)], 1)
    scale = torch.stack([xs_a.double().abs().sum(1), (y_a.double() * my2.double()).abs().sum(1), y_a.double().abs().sum(1)], 1) + 1e-30
    assert float(((S_b.double().sum(1) - want).abs() / scale).max()) < 1e-5
    assert float(((S_a.double().sum(1) - want).abs() / scale).max()) < 1e-5
    # y only (no next product) and no relu^2
    y_c = torch.empty(B, N, device=dev, dtype=torch.float16)
    ops.mm8_reduce_rows(parts, rx, mx, S, act=0, y=y_c)
    y_d = torch.empty(B, N, device=dev, dtype=torch.float16)
    ops.mm8t_gemm_fused(xs, qT, N, rx, mx, S, act=0, y=y_d, tiled=tiled, row_halves=halves)
    assert torch.equal(y_c, y_d)


@pytest.mark.parametrize("M,N,K,bias,act", [(32, 8192, 2048, False, 1), (32, 4096, 4096, True, 0), (17, 132, 256, True, 1),
                                            (1, 128, 128, False, 0), (24, 1000, 512, True, 1), (64, 4096, 1024, True, 1)])
def test_in_launch_pair_reduction_gives_the_bits_of_the_reduce_launch(M, N, K, bias, act):
    """At <= 32 rows a 2..4-way K split reduces inside its launch: every K-slice of a tile writes its partial, the last
    workgroup to finish adds them in slice order (its own from on-chip sums) and applies the epilogue -- the reduce kernel's
    arithmetic, so the same bits whichever workgroup was last, launch after launch (the counters come back to zero).  (64 rows:
    the reduce launch either way.)"""
    from chirrup_amd import ops

    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda").half() if bias else None
    for splits in (2, 4):                              # 4: in the launch while 3 x M <= 96 rows of partials, else the reduce launch
        ops.PAIR_REDUCE = False
        try:
            want = ops.skinny_linear(x, w, b, act=act, splits=splits)
        finally:
            ops.PAIR_REDUCE = True
        for _ in range(4):
            got = ops.skinny_linear(x, w, b, act=act, splits=splits)
            assert torch.equal(got.view(torch.int16), want.view(torch.int16))         # bits, signed zeros included
        assert int(ops._tile_counters(x.device).abs().sum()) == 0
        if N % 128 == 0 and K % 64 == 0:
            assert torch.equal(ops.skinny_linear(x, ops.tile_weight(w), b, act=act, splits=splits), want)


def test_in_launch_pair_reduction_of_a_grouped_launch():
    """The layer's R/K/V + LoRA down-projection launch at 32 rows (seven problems, tanh / sigmoid on two of them) with the
    in-launch reduction against the same launch followed by the reduce kernel: equal bits, repeatedly."""
    from chirrup_amd import ops

    torch.manual_seed(11)
    M, K = 32, 2048
    mixed = torch.randn(6, M, K, device="cuda").half()
    rkv = [ops.tile_weight((torch.randn(2048, K, device="cuda") / K ** 0.5).half()) for _ in range(3)]
    lora1 = (torch.randn(4, 256, K, device="cuda") / K ** 0.5).half()
    ranks, acts = [64, 64, 64, 256], [None, "tanh", None, "sigmoid"]

    def run():
        out_rkv = torch.empty(3, M, 2048, device="cuda", dtype=torch.float16)
        hid = torch.zeros(4, M, 256, device="cuda", dtype=torch.float16)
        probs = [(mixed[j], rkv[j], out_rkv[j], None, None) for j in range(3)]
        probs += [(mixed[2 + j], lora1[j, :ranks[j]], hid[j, :, :ranks[j]], None, acts[j]) for j in range(4)]
        ops.skinny_group(probs, splits=splits)
        return out_rkv, hid

    for splits in (2, 4):
        ops.PAIR_REDUCE = False
        try:
            want = run()
        finally:
            ops.PAIR_REDUCE = True
        for _ in range(3):
            got = run()
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert int(ops._tile_counters(mixed.device).abs().sum()) == 0


@pytest.mark.parametrize("M,C,ranks,p0,halves", [(200, 4096, [128, 128, 128, 512], 0, True), (200, 4096, [128, 128, 128, 512], 1, True),
                                                 (130, 2048, [64, 128, 128, 256], 0, True), (256, 1024, [64, 64, 64, 192], 0, True),
                                                 (40, 768, [64, 64, 64, 128], 1, True), (64, 4096, [128, 128, 128, 512], 0, False),
                                                 (64, 4096, [128, 128, 128, 512], 0, True), (32, 2048, [128, 128, 64, 256], 0, False),
                                                 (17, 2048, [128, 128, 64, 256], 1, False), (1, 768, [64, 64, 64, 128], 0, False),
                                                 (128, 2048, [128, 128, 64, 256], 0, False)])
def test_time_mix_launch_with_the_lora_chain_inside(M, C, ranks, p0, halves):
    """rwkv7_tmix_gemms: R/K/V and the whole LoRA chain (down-projections, tanh / sigmoid, up-projections + bias) in ONE launch,
    the chain on the CUs the R/K/V tiles leave idle, its stages handed from workgroup to workgroup inside the launch.  Against
    the two launches it replaces (skinny_group + skinny_bmm): R/K/V bit-identical (the same unsplit sums); the hidden planes
    differ only by the binary32 order of a 4-way K split (<= 1 binary16 ulp before the activation); the up-projections within
    the GEMM bar of binary64.  hid / outputs are poisoned with NaN before every launch, so a consumer that read a tile before its
    producer published it shows as NaN; repeated 20 times, with a competing memory stream on another stream in half of the
    runs (uneven load); every hand-off word is back at zero and the status word clear after each launch.
    halves=False: whole-row tiles (<= 128 rows); at <= 64 rows the R/K/V tiles are then split over K and reduced inside the launch
    (the bits of skinny_group's in-launch reduction at the same split count are not asserted here: the split is the library's
    choice; the values are held to the GEMM bar)."""
    from chirrup_amd import ops

    torch.manual_seed(M + C + p0)
    K, dmax = C, (max(ranks) + 63) // 64 * 64
    mixed = torch.randn(6, M, K, device="cuda").half()
    rkv_w = [ops.tile_weight((torch.randn(C, K, device="cuda") / K ** 0.5).half()) for _ in range(3)]
    lora1 = torch.zeros(4, dmax, K, device="cuda", dtype=torch.float16)
    lora2 = torch.zeros(4, C, dmax, device="cuda", dtype=torch.float16)
    for j, r in enumerate(ranks):
        lora1[j, :r] = (torch.randn(r, K, device="cuda") / K ** 0.5).half()
        lora2[j, :, :r] = (torch.randn(C, r, device="cuda") / r ** 0.5).half()
    lora2_t = ops.tile_weight_batch(lora2)
    lbias = torch.randn(4, 1, C, device="cuda").half()
    acts = [None, "tanh", None, "sigmoid"]
    nz = 4 - p0

    def old():
        rkv = torch.empty(3, M, C, device="cuda", dtype=torch.float16)
        hid = torch.zeros(nz, M, dmax, device="cuda", dtype=torch.float16)
        probs = [(mixed[j], rkv_w[j], rkv[j], None, None) for j in range(3)]
        probs += [(mixed[2 + j], lora1[j, :ranks[j]], hid[j - p0, :, :ranks[j]], None, acts[j]) for j in range(p0, 4)]
        ops.skinny_group(probs, splits=1, row_halves=halves)          # unsplit: the sums the chain launch's R/K/V tiles form too
        up = ops.skinny_bmm(hid, lora2_t[p0:], lbias[p0:], splits=1, k_of=ranks[p0:], row_halves=halves)
        return rkv, hid, up

    want_rkv, want_hid, want_up = old()
    rkv = torch.empty(3, M, C, device="cuda", dtype=torch.float16)
    hid = torch.empty(nz, M, dmax, device="cuda", dtype=torch.float16)
    up = torch.empty(nz, M, C, device="cuda", dtype=torch.float16)
    main_p = [(mixed[j], rkv_w[j], rkv[j]) for j in range(3)]
    lora_p = [(mixed[2 + j], lora1[j, :ranks[j]], j - p0, lbias[j].view(-1), up[j - p0], acts[j], ranks[j]) for j in range(p0, 4)]
    side = torch.cuda.Stream()
    noise = torch.empty(64 << 20, device="cuda", dtype=torch.float16)
    ref_up = []
    for j in range(p0, 4):
        h = (mixed[2 + j].double() @ lora1[j, :ranks[j]].double().t()).half().double()
        h = torch.tanh(h) if acts[j] == "tanh" else (torch.sigmoid(h) if acts[j] == "sigmoid" else h)
        ref_up.append(h.half().double() @ lora2[j, :, :ranks[j]].double().t() + lbias[j].double())
    for it in range(20):
        rkv.fill_(float("nan")), hid.fill_(float("nan")), up.fill_(float("nan"))
        if it & 1:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                noise.add_(1)                                   # a competing kernel while the launch runs
        ops.tmix_gemms(main_p, lora_p, lora2_t[p0:], hid, row_halves=halves)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert ops.chain_status() == 0
        assert all(int(t.abs().sum()) == 0 for t in ops._chain_sync.values())
        if halves or M > 64:
            assert torch.equal(rkv, want_rkv)                   # unsplit R/K/V tiles: the same sums
        else:                                                   # split over K and reduced in the launch: another binary32 order
            assert not bool(torch.isnan(rkv).any())
            assert float((rkv.float() - want_rkv.float()).abs().max()) <= 2e-3 * max(1.0, float(want_rkv.float().abs().max()))
        for j in range(p0, 4):
            z, r = j - p0, ranks[j]
            got_h, want_h = hid[z, :, :r].float(), want_hid[z, :, :r].float()
            assert not bool(torch.isnan(got_h).any()), (it, j)
            assert float((got_h - want_h).abs().max()) <= 2e-3 * max(1.0, float(want_h.abs().max())), (it, j)
            assert not bool(torch.isnan(up[z]).any()), (it, j)
            assert bool(((up[z].double() - ref_up[z]).abs() <= 4e-3 * ref_up[z].abs().clamp_min(1.0)).all()), (it, j)
            assert float((up[z].float() - want_up[z].float()).abs().max()) <= 8e-3 * max(1.0, float(want_up[z].abs().max()))


@pytest.mark.parametrize("M,C,halves", [(200, 2048, True), (64, 2048, False), (64, 1024, True), (32, 2048, False), (17, 1024, False), (1, 768, False)])
def test_time_mix_launch_with_uint8_main_tiles(M, C, halves):
    """rwkv7_tmix_gemms_mm8: R/K/V as uint8 (mm8) tiles inside the time-mix launch -- activation prologues xs = fp16(x * ry) and row
    sums given, the rank-1 corrections in the tiles' epilogues (unsplit tiles) or after the in-launch reduction (<= 64 rows:
    split over K).  Against mm8t_linear per problem (the same split form: same xs, same uint8 bytes; only the binary32 order
    of the K split differs) and, for the LoRA chain, against binary64.  NaN-poisoned outputs, 10 launches."""
    from chirrup_amd import ops

    torch.manual_seed(M + C)
    K, ranks = C, [64, 128, 64, 128]
    dmax = 128
    x = torch.randn(3, M, K, device="cuda").half()
    mixed = torch.randn(6, M, K, device="cuda").half()
    q8 = []
    for j in range(3):
        qT = torch.randint(0, 256, (C, K), device="cuda", dtype=torch.uint8)
        rx, mx = torch.rand(C, device="cuda").half() / 64, torch.randn(C, device="cuda").half() * 0.01
        my, ry = torch.randn(K, device="cuda").half() * 0.01, torch.rand(K, device="cuda").half() / 16
        q8.append((qT, ops.tile_weight_u8(qT), mx, rx, my, ry))
    xs = torch.stack([(x[j].float() * q8[j][5].float()).half() for j in range(3)])
    S = torch.stack([torch.stack([xs[j].float().sum(1), (x[j].float() * q8[j][4].float()).sum(1), x[j].float().sum(1)], 1) for j in range(3)]).contiguous()
    want = [ops.mm8t_linear(x[j], q8[j][1], q8[j][2], q8[j][3], q8[j][4], q8[j][5], tiled=True) for j in range(3)]
    lora1 = torch.zeros(4, dmax, K, device="cuda", dtype=torch.float16)
    lora2 = torch.zeros(4, C, dmax, device="cuda", dtype=torch.float16)
    for j, r in enumerate(ranks):
        lora1[j, :r] = (torch.randn(r, K, device="cuda") / K ** 0.5).half()
        lora2[j, :, :r] = (torch.randn(C, r, device="cuda") / r ** 0.5).half()
    lora2_t = ops.tile_weight_batch(lora2)
    lbias = torch.randn(4, 1, C, device="cuda").half()
    acts = [None, "tanh", None, "sigmoid"]
    rkv = torch.empty(3, M, C, device="cuda", dtype=torch.float16)
    hid = torch.empty(4, M, dmax, device="cuda", dtype=torch.float16)
    up = torch.empty(4, M, C, device="cuda", dtype=torch.float16)
    main_p = [(xs[j], (q8[j][1], True), rkv[j], q8[j][3], q8[j][2], S[j]) for j in range(3)]
    lora_p = [(mixed[2 + j], lora1[j, :ranks[j]], j, lbias[j].view(-1), up[j], acts[j], ranks[j]) for j in range(4)]
    ref_up = []
    for j in range(4):
        h = (mixed[2 + j].double() @ lora1[j, :ranks[j]].double().t()).half().double()
        h = torch.tanh(h) if acts[j] == "tanh" else (torch.sigmoid(h) if acts[j] == "sigmoid" else h)
        ref_up.append(h.half().double() @ lora2[j, :, :ranks[j]].double().t() + lbias[j].double())
    for it in range(10):
        rkv.fill_(float("nan")), hid.fill_(float("nan")), up.fill_(float("nan"))
        ops.tmix_gemms(main_p, lora_p, lora2_t, hid, row_halves=halves, mm8=True)
        torch.cuda.synchronize()
        assert ops.chain_status() == 0 and all(int(t.abs().sum()) == 0 for t in ops._chain_sync.values())
        for j in range(3):
            assert not bool(torch.isnan(rkv[j]).any()), (it, j)
            scale = float(want[j].float().abs().max())
            assert float((rkv[j].float() - want[j].float()).abs().max()) <= 2e-3 * max(1.0, scale), (it, j)
        for j in range(4):
            assert bool(((up[j].double() - ref_up[j]).abs() <= 4e-3 * ref_up[j].abs().clamp_min(1.0)).all()), (it, j)


@pytest.mark.parametrize("B,K,N,splits,tiled", [(32, 2048, 8192, 0, True), (64, 4096, 4096, 2, True), (17, 1024, 1024, 4, False), (1, 512, 384, 2, False),
                                                 (48, 2048, 2048, 3, True), (24, 4096, 16384, 0, True)])
def test_mm8_fused_epilogue_behind_the_in_launch_reduction(B, K, N, splits, tiled):
    """mm8t_gemm_fused(splits=..., partials=...) at <= 64 rows: the product split over K, each tile's last workgroup adds the other
    slices' sums in slice order and runs the EPI_MM8 epilogue.  Same core sums as mm8t_gemm_partial at that split count reduced by
    mm8_reduce_rows: y and the next product's xs bit-identical, its row sums (per 128-column tile instead of per 1024 columns) equal in
    binary64 terms; repeated launches reproduce themselves and leave the tile counters at zero; where the in-launch reduction does not
    apply the wrapper's predicate says so."""
    from chirrup_amd import ops

    torch.manual_seed(B + K + N)
    dev = "cuda"
    x = torch.randn(B, K, device=dev).half()
    q = torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8)
    mx, rx = torch.randn(N, device=dev).half() * 0.01, torch.rand(N, device=dev).half() / 64
    my, ry = torch.randn(K, device=dev).half() * 0.01, torch.rand(K, device=dev).half() / 16
    my2, ry2 = torch.randn(N, device=dev).half() * 0.01, torch.rand(N, device=dev).half() / 16
    xs = (x.float() * ry.float()).half()
    S = torch.stack([xs.float().sum(1), (x.float() * my.float()).sum(1), x.float().sum(1)], 1).contiguous()
    tiled = tiled and N % 128 == 0
    qT = ops.tile_weight_u8(q) if tiled else q
    ok = ops.mm8_fused_split_ok(B, K, N, splits)
    s_used = ops.gemm_splits(N, K, 1, splits)
    if not ok:
        assert N % 128 or not (2 <= s_used <= 4) or (s_used - 1) * B > 96
        return
    parts = ops.mm8t_gemm_partial(xs, qT, N, splits, torch.empty(s_used, B, N, device=dev), tiled=tiled)
    assert parts.shape[0] == s_used
    y_a, xs_a = torch.empty(B, N, device=dev, dtype=torch.float16), torch.empty(B, N, device=dev, dtype=torch.float16)
    S_a = torch.empty(B, ops.mm8_row_parts(N), 3, device=dev)
    ops.mm8_reduce_rows(parts, rx, mx, S, act=1, y=y_a, nxt=(ry2, my2, xs_a, S_a))
    slabs = torch.empty(s_used, B, N, device=dev)
    for it in range(4):
        y_b = torch.full((B, N), float("nan"), device=dev, dtype=torch.float16)
        xs_b = torch.full((B, N), float("nan"), device=dev, dtype=torch.float16)
        S_b = torch.full((B, ops.mm8_tile_parts(N), 3), float("nan"), device=dev)
        slabs.fill_(float("nan"))
        ops.mm8t_gemm_fused(xs, qT, N, rx, mx, S, act=1, y=y_b, nxt=(ry2, my2, xs_b, S_b), tiled=tiled, splits=splits, partials=slabs)
        torch.cuda.synchronize()
        assert torch.equal(y_a, y_b) and torch.equal(xs_a, xs_b), it
        want = torch.stack([xs_a.double().sum(1), (y_a.double() * my2.double()).sum(1), y_a.double().sum(1)], 1)
        scale = torch.stack([xs_a.double().abs().sum(1), (y_a.double() * my2.double()).abs().sum(1), y_a.double().abs().sum(1)], 1) + 1e-30
        assert float(((S_b.double().sum(1) - want).abs() / scale).max()) < 1e-5
        assert int(ops._tile_counters(x.device).abs().sum()) == 0
    # y only, no relu^2
    y_c, y_d = torch.empty(B, N, device=dev, dtype=torch.float16), torch.empty(B, N, device=dev, dtype=torch.float16)
    ops.mm8_reduce_rows(parts, rx, mx, S, act=0, y=y_c)
    ops.mm8t_gemm_fused(xs, qT, N, rx, mx, S, act=0, y=y_d, tiled=tiled, splits=splits, partials=slabs)
    assert torch.equal(y_c, y_d)
