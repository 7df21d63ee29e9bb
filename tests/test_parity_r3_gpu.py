"""Round-3 parity additions ("Next round" item 1):
  * the reference's OWN stored mm8 / channel-mix outputs (tests/golden/mm8.npz, cmix.npz -- written by importing
    scripts/test_mm8/benchmark_pure_pytorch.py and Albatross/rwkv7.py, tests/golden/make_golden.py) fed to the HIP kernels;
  * the GEMM epilogues with operands the 16-byte store path must NOT take (Y / bias 8-byte aligned only, ldy % 8 == 4,
    N % 8 == 4), surrounded by canaries -- the regression test for the epilogue rewritten in commit 4599158;
  * the LN / token-shift kernel on rows much shorter than its 1024 lanes (C = 768) with every operand carved from ONE
    canary-filled arena, so a lane that runs past its row shows as a changed canary or a wrong value instead of a fault.
"""
import os

import numpy as np
import pytest
import torch

from util import bits

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F16, F32 = np.float16, np.float32


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------------------------------------ reference fixtures
@pytest.mark.parametrize("tag", ["wide", "tall"])
def test_reference_mm8_fixture_through_the_hip_kernels(tag, oracle):
    """mm8.npz: x, the quantised bytes and scales, and the outputs of the reference's original_mm8 / optimized_mm8
    (benchmark_pure_pytorch.py:11-53).  Every HIP form of the product -- the as-coded kernel, the reference-named op on
    the matrix cores (packed weights), the stateless C-ABI entry, the K-contiguous mm8t_linear -- against the reference's
    stored y (bars below) and against each other."""
    from chirrup_amd import ops

    d = np.load(os.path.join(G, "mm8.npz"))
    x, q = d[f"{tag}_x"], d[f"{tag}_q"]
    B, N = x.shape
    M = q.shape[1]
    mx, rx, my, ry = (cu(d[f"{tag}_{n}"].reshape(-1)) for n in ("mx", "rx", "my", "ry"))
    tx, tq = cu(x), cu(q)
    want = d[f"{tag}_y_original"].astype(F32)
    want_opt = d[f"{tag}_y_optimized"].astype(F32)
    scale = float(np.abs(want).max())
    # The stored outputs are binary16 PIPELINES evaluated by torch on CPU (weights dequantised to binary16, binary16
    # epilogue): they sit up to 4e-3 of the output scale from the binary32 as-coded kernel form -- the bar
    # tests/test_golden_cpu.py::test_mm8_quantisation_and_formulas_match_reference holds the oracle to against the same
    # fixture; the reference's own rtol 1e-3 / atol 1e-4 (benchmark_pure_pytorch.py:92) is between its two GPU forms.
    close = lambda g, ref: bool(np.allclose(g, ref, rtol=4e-3, atol=4e-3 * scale))

    y = torch.empty((B, M), dtype=torch.float16, device="cuda")
    ops.mm8_seq_direct(B, N, M, tx, tq, mx, rx, my, ry, y)
    got = y.cpu().numpy()
    want_oracle = oracle.mm8_seq(x, q, d[f"{tag}_mx"], d[f"{tag}_rx"], d[f"{tag}_my"], d[f"{tag}_ry"])
    assert np.array_equal(bits(got), bits(want_oracle))                      # as coded: the oracle's bits
    assert close(got.astype(F32), want) and close(got.astype(F32), want_opt)
    for name, run in (("mm8_seq op", lambda o: ops.mm8_seq(B, N, M, tx, tq, mx, rx, my, ry, o)),
                      ("mm8_seq C ABI", lambda o: ops.mm8_seq_stateless(B, N, M, tx, tq, mx, rx, my, ry, o)),
                      ("mm8t_linear", lambda o: ops.mm8t_linear(tx, tq.t().contiguous(), mx, rx, my, ry, out=o))):
        o = torch.zeros((B, M), dtype=torch.float16, device="cuda")
        run(o)
        g = o.cpu().numpy().astype(F32)
        assert close(g, want) and close(g, want_opt), (name, float(np.abs(g - want).max()), scale)
        # ... and the matrix-core forms against the as-coded kernel at the reference's own bar between its two forms
        assert bool(np.allclose(g, want_oracle.astype(F32), rtol=1e-3, atol=1e-4 + 2e-3 * scale)), name


@pytest.mark.parametrize("C", [128, 256])
def test_reference_cmix_fixture_through_the_hip_gemms(C):
    """cmix.npz: inputs and outputs of the reference's RWKV_x070_CMix_seq_batch (Albatross/rwkv7.py:673-679) on CPU.  The
    token-shift lerp is three element-wise binary16 ops (as the reference's eager ops); ffn.key + relu^2 and ffn.value run
    through the hand-written GEMM launches the decode step uses (row-major and tile-image weights, whole rows; the
    split-K partials of ffn.value summed like the next LN kernel sums them).  Bar of tests/test_golden_cpu.py: 2 ulp / 1e-3."""
    from chirrup_amd import ops

    d = np.load(os.path.join(G, "cmix.npz"))
    x, xp = cu(d[f"c{C}:x"]), cu(d[f"c{C}:x_prev_in"])
    x_k, K, V = cu(d[f"c{C}:x_k"]), cu(d[f"c{C}:K"]), cu(d[f"c{C}:V"])              # K [4C, C]; V [4C, C] (pre-transposed)
    B, T, _ = x.shape
    xx = torch.cat((xp[1].unsqueeze(1), x[:, :-1, :]), dim=1) - x
    k = (x + xx * x_k).view(B * T, C)
    assert np.array_equal(bits(x[:, -1].cpu().numpy()), bits(d[f"c{C}:x_prev_out"][1]))
    want = d[f"c{C}:y"].reshape(B * T, C).astype(F32)
    Vt = V.t().contiguous()                                                           # [C, 4C]: the NT form the model keeps
    for tiled in (False, True):
        Kw = ops.tile_weight(K) if tiled else K
        Vw = ops.tile_weight(Vt) if (tiled and C % 128 == 0) else Vt
        kf = ops.skinny_linear(k, Kw, act=1)
        parts = ops.skinny_linear_partial(kf, Vw, 0, torch.empty((16, B * T, C), dtype=torch.float32, device="cuda"))
        acc = torch.zeros((B * T, C), dtype=torch.float32, device="cuda")
        for s in range(parts.shape[0]):
            acc += parts[s]                                                           # plane order, as ln_row adds them
        got = acc.half().cpu().numpy().astype(F32)
        ulp = np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10
        assert bool((np.abs(got - want) <= np.maximum(2 * ulp, 1e-3 * np.abs(want) + 1e-4)).all()), float(np.abs(got - want).max())


# ------------------------------------------------------------------------------------------------ epilogue regression
def _carve(n_elems, offset, dtype=torch.float16, fill=7.0):
    """A canary-filled buffer and the view [offset, offset + n_elems) of it."""
    buf = torch.full((n_elems + offset + 64,), fill, dtype=dtype, device="cuda")
    return buf, buf[offset:offset + n_elems]


@pytest.mark.parametrize("M,N,K,ldy_pad,off,splits,halves", [
    (200, 260, 256, 0, 4, 1, True),       # N % 8 == 4, Y / bias 8-byte aligned only
    (200, 256, 256, 12, 4, 1, True),      # ldy % 8 == 4 with an N that would otherwise take the 16-byte path
    (200, 256, 256, 8, 4, 1, False),      # 16-byte ldy, misaligned base
    (24, 768, 768, 4, 0, 1, False),       # the aborted test's own shape (C = 768, 24 rows), ldy % 8 == 4
    (24, 772, 768, 0, 4, 2, False),       # ... through the reduce launch (N % 8 == 4: its 8-byte form)
    (130, 132, 128, 4, 4, 1, True),
])
def test_f16_epilogue_with_operands_the_16_byte_path_must_not_take(M, N, K, ldy_pad, off, splits, halves):
    """EPI_F16 (unsplit: bias / relu^2 in the GEMM epilogue) and the reduce launch with Y = base + 4 elements (8-byte
    aligned), a row stride with ldy % 8 == 4 and / or N % 8 == 4, bias = base + 4: the 16-byte f16x8 stores and vector bias
    loads of commit 4599158 must fall back to the 8-byte form.  Values against binary64, and not one element outside
    y[m, 0:N] may change (canaries before, between the rows and after)."""
    from chirrup_amd import ops

    torch.manual_seed(M + N + K + ldy_pad + off)
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    ldy = N + ldy_pad
    ybuf, yflat = _carve(M * ldy, off)
    y = yflat.view(M, ldy)[:, :N]
    bbuf, b = _carve(N, off, fill=0.0)
    b.copy_(torch.randn(N, device="cuda").half())
    assert y.data_ptr() % 16 == (8 if off == 4 else 0) and y.stride(0) == ldy
    for act in (0, 1):
        ybuf.fill_(7.0)
        ops.skinny_linear(x, w, b, act=act, splits=splits, out=y, row_halves=halves)
        ref = (x.double() @ w.double().t() + b.double())
        if act:
            ref = torch.relu(ref.half().double()) ** 2
        assert bool(((y.double() - ref).abs() <= (4e-3 if act else 2e-3) * ref.abs().clamp_min(1.0)).all())
        keep = torch.ones_like(ybuf, dtype=torch.bool)
        keep[off:off + M * ldy].view(M, ldy)[:, :N] = False
        assert bool((ybuf[keep] == 7.0).all()), "the epilogue wrote outside y[m, 0:N]"


def test_grouped_launch_with_unaligned_outputs():
    """The layer's grouped launch (R/K/V + LoRA down-projections) with output planes whose base is 8-byte aligned only and
    whose row stride has ldy % 8 == 4: per-problem fallback to the 8-byte stores, canaries intact."""
    from chirrup_amd import ops

    torch.manual_seed(5)
    M, K = 24, 768
    mixed = torch.randn(6, M, K, device="cuda").half()
    ws = [(torch.randn(768, K, device="cuda") / K ** 0.5).half() for _ in range(3)]
    lora1 = (torch.randn(4, 128, K, device="cuda") / K ** 0.5).half()
    ranks, acts = [64, 64, 64, 128], [None, "tanh", None, "sigmoid"]
    ld_r, ld_h, off = 768 + 4, 128 + 4, 4
    rbuf, rflat = _carve(3 * M * ld_r, off)
    hbuf, hflat = _carve(4 * M * ld_h, off)
    out_rkv, hid = rflat.view(3, M, ld_r), hflat.view(4, M, ld_h)
    probs = [(mixed[j], ws[j], out_rkv[j, :, :768], None, None) for j in range(3)]
    probs += [(mixed[2 + j], lora1[j, :ranks[j]], hid[j, :, :ranks[j]], None, acts[j]) for j in range(4)]
    for splits in (1, 3):
        rbuf.fill_(7.0), hbuf.fill_(7.0)
        ops.skinny_group(probs, splits=splits)
        for j in range(3):
            ref = mixed[j].double() @ ws[j].double().t()
            assert bool(((out_rkv[j, :, :768].double() - ref).abs() <= 2e-3 * ref.abs().clamp_min(1.0)).all())
        for j in range(4):
            ref = (mixed[2 + j].double() @ lora1[j, :ranks[j]].double().t()).half().double()
            ref = torch.tanh(ref) if acts[j] == "tanh" else (torch.sigmoid(ref) if acts[j] == "sigmoid" else ref)
            assert bool(((hid[j, :, :ranks[j]].double() - ref).abs() <= 2e-3 * ref.abs().clamp_min(1.0)).all()), j
        assert bool((out_rkv[:, :, 768:] == 7.0).all()) and bool((rbuf[:off] == 7.0).all()) and bool((rbuf[off + 3 * M * ld_r:] == 7.0).all())
        for j in range(4):
            assert bool((hid[j, :, ranks[j]:] == 7.0).all()), j
        assert bool((hbuf[:off] == 7.0).all()) and bool((hbuf[off + 4 * M * ld_h:] == 7.0).all())


def test_mm8_epilogue_refuses_what_its_vector_accesses_cannot_take():
    """EPI_MM8 (mm8t_gemm_fused) loads rx / mx / ry2 / my2 and stores y / xs2 sixteen bytes at a time: the C entry refuses
    N % 8 != 0, y_stride % 8 != 0 and any of those pointers off a 16-byte boundary BEFORE a launch (CHIRRUP_E_SHAPE /
    CHIRRUP_E_ALIGN), so there is no unaligned form to fall back to."""
    from chirrup_amd import lib

    L = lib.load()
    B, K, N = 40, 128, 256
    dev = "cuda"
    xs = torch.randn(B, K, device=dev).half()
    q = torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8)
    v = lambda n: torch.rand(n + 8, device=dev).half()
    rx, mx, ry2, my2 = v(N), v(N), v(N), v(N)
    S = torch.zeros(B, 1, 3, device=dev)
    y, xs2 = torch.zeros(B * (N + 8) + 8, device=dev).half(), torch.zeros(B * N + 8, device=dev).half()
    S2 = torch.zeros(B, L.mm8_tile_parts(N), 3, device=dev)

    def call(N_=N, rx_o=0, mx_o=0, y_o=0, y_stride=N, ry_o=0, my_o=0, xs2_o=0):
        p = lambda t, o: t.data_ptr() + 2 * o
        return L.mm8t_gemm_fused(B, K, N_, xs.data_ptr(), K, q.data_ptr(), K, 0, p(rx, rx_o), p(mx, mx_o), S.data_ptr(), 1, 1,
                                 p(y, y_o), y_stride, p(ry2, ry_o), p(my2, my_o), p(xs2, xs2_o), S2.data_ptr(), 1,
                                 torch.cuda.current_stream().cuda_stream)

    assert call() == 0
    assert call(N_=N - 4) == -1
    assert call(y_stride=N + 4) == -1
    for kw in ("rx_o", "mx_o", "y_o", "ry_o", "my_o", "xs2_o"):
        assert call(**{kw: 4}) == -3, kw
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ LN kernel, short rows
@pytest.mark.parametrize("C,B,T,n_mix,splits", [(768, 4, 6, 6, 12), (768, 4, 1, 1, 3), (768, 24, 1, 6, 12), (768, 4, 6, 0, 12),
                                                (128, 3, 5, 6, 4), (2048, 5, 1, 1, 8)])
def test_ln_kernel_lanes_past_the_row_touch_nothing(C, B, T, n_mix, splits):
    """rwkv7_add_ln_mix runs 1024 lanes per row, 8 channels per lane: at C = 768 only 96 lanes are inside the row (the
    configuration whose first fused forward aborted in round 2, gpurun_out/r2z/t12.log).  Every operand -- x, the split-K
    partial planes, x_out, the carry rows, out -- is a slice of ONE arena filled with a canary, with canary gaps between the
    slices; after the launch the gaps are intact and the values are the oracle's, in the prefill (T > 1: recomputed
    predecessor row) and decode forms, with partials and with a plain delta."""
    from chirrup_amd import ops
    from oracle import rwkv7_np as M

    rng = np.random.default_rng(C + B + T + n_mix)
    rows = B * T
    GAP = 4096                                                               # elements of canary between operands
    sizes = {"x": rows * C, "x_out": rows * C, "prev_in": B * C, "prev_out": B * C, "out": max(n_mix, 1) * rows * C}
    arena16 = torch.full((sum(sizes.values()) + GAP * (len(sizes) + 1),), 7.0, dtype=torch.float16, device="cuda")
    views, pos = {}, GAP
    for name, n in sizes.items():
        views[name] = arena16[pos:pos + n]
        pos += n + GAP
    arena32 = torch.full((splits * rows * C + 2 * GAP,), 7.0, dtype=torch.float32, device="cuda")
    dp_t = arena32[GAP:GAP + splits * rows * C].view(splits, rows, C)

    x = rng.standard_normal((B, T, C)).astype(F16)
    w = (1 + 0.1 * rng.standard_normal(C)).astype(F16)
    b = (0.1 * rng.standard_normal(C)).astype(F16)
    prev = rng.standard_normal((B, C)).astype(F16)
    mix = rng.uniform(0, 1, (max(n_mix, 1), C)).astype(F16)
    dp = (rng.standard_normal((splits, rows, C)) * 0.2).astype(F32)
    acc = np.zeros((rows, C), F32)
    for s in range(splits):
        acc += dp[s]
    xn = (x.astype(F32) + acc.astype(F16).reshape(B, T, C).astype(F32)).astype(F16)
    cur = M.layer_norm(xn, w, b)
    views["x"].copy_(cu(x).view(-1)), views["prev_in"].copy_(cu(prev).view(-1)), dp_t.copy_(cu(dp))
    tx, tx_out = views["x"].view(B, T, C), views["x_out"].view(B, T, C)
    tprev, tcarry = views["prev_in"].view(B, C), views["prev_out"].view(B, C)
    out = views["out"].view(max(n_mix, 1), B, T, C)
    if n_mix:
        ops.add_ln_mix(B, T, C, tx, None, tx_out, cu(w), cu(b), 1e-5, tprev, tcarry, cu(mix[:n_mix]), out, delta_partials=dp_t)
    else:
        ops.add_ln_mix(B, T, C, tx, None, None, cu(w), cu(b), 1e-5, None, None, None, out[0], delta_partials=dp_t)
    torch.cuda.synchronize()
    # canaries: the gaps of both arenas
    mask = torch.ones_like(arena16, dtype=torch.bool)
    pos = GAP
    for name, n in sizes.items():
        mask[pos:pos + n] = False
        pos += n + GAP
    assert bool((arena16[mask] == 7.0).all()), "a lane wrote outside its operand"
    assert bool((arena32[:GAP] == 7.0).all()) and bool((arena32[GAP + splits * rows * C:] == 7.0).all())
    assert np.array_equal(bits(tx.cpu().numpy()), bits(x)) and np.array_equal(dp_t.cpu().numpy(), dp)     # inputs untouched
    if n_mix:
        assert np.array_equal(bits(tx_out.cpu().numpy()), bits(xn))
        dx = np.concatenate([prev[:, None], cur[:, :-1]], 1) - cur
        want = np.stack([cur + dx * mix[m] for m in range(n_mix)])
        d = np.abs(out.cpu().numpy().astype(F32) - want.astype(F32))
        assert d.max() <= 2e-2 and (d > 4e-3).mean() < 2e-3, float(d.max())
        assert np.abs(tcarry.cpu().numpy().astype(F32) - cur[:, -1].astype(F32)).max() <= 8e-3
    else:
        assert bool((views["x_out"] == 7.0).all()) and bool((views["prev_out"] == 7.0).all())            # not given: not written
        d = np.abs(out[0].cpu().numpy().astype(F32) - cur.astype(F32))
        assert d.max() <= 8e-3, float(d.max())


@pytest.mark.gpu
@pytest.mark.parametrize("C", [768, 2048, 4096])
@pytest.mark.parametrize("n_mix", [0, 1, 6])
def test_ln_kernel_without_mm8_hooks_is_the_same_kernel(C, n_mix):
    """Launches without a chirrup_mm8_fuse record run an instantiation of add_ln_mix_kernel that was compiled without the mm8
    hooks ("Bugs found in round 3": as run-time branches they cost the binary16 model registers).  With an all-null
    record the other instantiation runs: the two must write the same bits (decode form with folded split-K partials, and a
    four-token chunk)."""
    import ctypes

    from chirrup_amd import ops

    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(C + n_mix)
    for B, T, splits in ((5, 1, 8), (2, 4, 0)):
        x = torch.randn(B, T, C, generator=g).half().to(dev)
        part = (torch.randn(splits, B * T, C, generator=g) * 0.3).to(dev) if splits else None
        delta = None if splits else (torch.randn(B, T, C, generator=g) * 0.3).half().to(dev)
        ln_w, ln_b = (1 + 0.1 * torch.randn(C, generator=g)).half().to(dev), (0.1 * torch.randn(C, generator=g)).half().to(dev)
        prev = torch.randn(B, C, generator=g).half().to(dev)
        mix = torch.rand(n_mix, C, generator=g).half().to(dev) if n_mix else None
        outs = []
        for with_record in (False, True):
            x_out, prev_out = torch.empty_like(x), torch.empty_like(prev)
            out = torch.full((max(n_mix, 1), B, T, C), float("nan"), dtype=torch.float16, device=dev)
            if not with_record:
                ops.add_ln_mix(B, T, C, x, delta, x_out, ln_w, ln_b, 1e-5, prev if n_mix else None, prev_out if n_mix else None, mix, out,
                               delta_partials=part)
            else:
                fz = ops._Mm8Fuse()                    # every hook null: the instantiation with the hooks compiled in
                rc = ops._lib.load().rwkv7_add_ln_mix_mm8(
                    B, T, C, n_mix, ops._ptr(x), ops._ptr(delta), ops._ptr(x_out), ops._ptr(ln_w), ops._ptr(ln_b), 1e-5,
                    ops._ptr(prev if n_mix else None), ops._ptr(prev_out if n_mix else None), ops._ptr(mix), ops._ptr(out), B * T * C, None,
                    ops._ptr(part), splits, ctypes.addressof(fz), ops._stream())
                assert rc == 0
            torch.cuda.synchronize()
            outs.append((out.clone(), x_out.clone(), prev_out.clone() if n_mix else None))
        same = lambda u, v: torch.equal(u.view(torch.int16), v.view(torch.int16))
        assert same(outs[0][0], outs[1][0]) and same(outs[0][1], outs[1][1])
        if n_mix:
            assert same(outs[0][2], outs[1][2])
        assert not torch.isnan(outs[0][0].float()).any()
