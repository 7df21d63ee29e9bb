"""Unit parity of the fused element-wise HIP kernels (csrc/elementwise.hip) against the numpy
oracle's per-op restatement (oracle/rwkv7_np.py).  Element-wise parts must be bit-identical; the
binary32 reductions (norms, head sums) may differ in summation order, which can move a result by
one binary16 ulp on a small fraction of elements."""
import numpy as np
import pytest
import torch

from oracle import rwkv7_np as M
from util import bits

pytestmark = pytest.mark.gpu
F16, F32 = np.float16, np.float32


def ulp_diff(a, b):
    def key(x):
        u = bits(x).astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u)
    return np.abs(key(a) - key(b))


def assert_close_ulps(got, want, max_ulp=1, max_frac=0.01, name="", atol=1e-3):
    """Every element within max_ulp binary16 ulps of the oracle, or -- where operands of magnitude
    ~1 cancel to a result near zero, so that an ulp of the RESULT is far below the rounding of its
    inputs -- within atol (= one ulp at magnitude 1); and at most max_frac of elements differ at all."""
    d = ulp_diff(got, want)
    ad = np.abs(got.astype(F32) - want.astype(F32))
    bad = (d > max_ulp) & ~(ad <= atol)
    assert not bad.any(), f"{name}: {int(bad.sum())} elements off, worst {d[bad].max()} ulps / {ad[bad].max()} abs"
    assert (d > 0).mean() <= max_frac, f"{name}: {(d > 0).mean():.4f} of elements differ"


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("B,T,C,n_mix,with_delta", [(3, 1, 128, 6, True), (2, 4, 768, 6, True), (5, 1, 4096, 1, True),
                                                    (2, 3, 256, 1, False), (4, 1, 2048, 0, True), (200, 1, 4096, 6, True)])
def test_add_ln_mix(B, T, C, n_mix, with_delta):
    from chirrup_amd import ops

    rng = np.random.default_rng(B * 7 + T + C)
    x = rng.standard_normal((B, T, C)).astype(F16)
    delta = (rng.standard_normal((B, T, C)) * 0.5).astype(F16) if with_delta else None
    w = (1 + 0.1 * rng.standard_normal(C)).astype(F16)
    b = (0.1 * rng.standard_normal(C)).astype(F16)
    prev = rng.standard_normal((B, C)).astype(F16)
    mix = rng.uniform(0, 1, (max(n_mix, 1), C)).astype(F16)
    # oracle
    xn = x + delta if with_delta else x
    cur = M.layer_norm(xn, w, b)
    if n_mix:
        dx = np.concatenate([prev[:, None], cur[:, :-1]], 1) - cur
        want = np.stack([cur + dx * mix[m] for m in range(n_mix)])
    else:
        want = cur[None]
    tx, tprev = cu(x), cu(prev)
    out = torch.empty((max(n_mix, 1), B, T, C), dtype=torch.float16, device="cuda")
    prev_out = tprev if T == 1 else torch.empty_like(tprev)
    x_new = tx if T == 1 else torch.empty_like(tx)       # in place only for T == 1 (T > 1: rows re-read their predecessor)
    ops.add_ln_mix(B, T, C, tx, cu(delta) if with_delta else None, x_new if with_delta else None, cu(w), cu(b), 1e-5,
                   tprev if n_mix else None, prev_out if n_mix else None, cu(mix[:n_mix]) if n_mix else None, out)
    got = out.cpu().numpy()
    if with_delta:
        assert np.array_equal(bits(x_new.cpu().numpy()), bits(xn))       # residual add is exact
    # LN outputs reach |x| ~ 4-5 (ulp 3.9e-3); a one-ulp flip there can survive a cancelling lerp
    assert_close_ulps(got, want, 2, 0.02, "mixed", atol=4e-3)
    if n_mix:
        assert_close_ulps(prev_out.cpu().numpy(), cur[:, -1], 1, 0.01, "carry")


@pytest.mark.parametrize("rows,C,layer0", [(3, 128, True), (7, 768, False), (200, 4096, False)])
def test_tmix_mid(rows, C, layer0):
    from chirrup_amd import ops

    rng = np.random.default_rng(rows + C)
    H = C // 64
    k = rng.standard_normal((rows, C)).astype(F16)
    v = rng.standard_normal((rows, C)).astype(F16)
    a_pre = rng.standard_normal((rows, C)).astype(F16)
    vg_pre = rng.standard_normal((rows, C)).astype(F16)
    v_first = rng.standard_normal((rows, C)).astype(F16)
    k_k = (0.85 + 0.05 * rng.standard_normal(C)).astype(F16)
    k_a = (1 + 0.05 * rng.standard_normal(C)).astype(F16)
    # oracle: oracle/rwkv7_np.tmix lines for rwkv7.py:629-637
    a = M.sigmoid_h(a_pre)
    kk_in = (k * k_k).reshape(rows, H, 64)
    nrm = np.sqrt((kk_in.astype(F32) ** 2).sum(-1, keepdims=True, dtype=F32)).astype(F16)
    kk = (kk_in / np.maximum(nrm, F16(1e-12))).reshape(rows, C)
    k_want = k * (F16(1.0) + (a - F16(1.0)) * k_a)
    kka_want = kk * a
    v_want = v if layer0 else v + (v_first - v) * M.sigmoid_h(vg_pre)
    tk, tv = cu(k), cu(v)
    nk, kka = torch.empty_like(tk), torch.empty_like(tk)
    ops.tmix_mid(rows, C, tk, tv, cu(a_pre), None if layer0 else cu(vg_pre), None if layer0 else cu(v_first), cu(k_k),
                 cu(k_a), nk, kka)
    assert_close_ulps(tk.cpu().numpy(), k_want, 1, 0.005, "k")          # sigmoid: device expf vs numpy
    assert_close_ulps(nk.cpu().numpy(), -kk, 1, 0.01, "neg_kk")
    assert_close_ulps(kka.cpu().numpy(), kka_want, 2, 0.02, "kka")
    assert_close_ulps(tv.cpu().numpy(), v_want, 1, 0.005, "v")


@pytest.mark.parametrize("rows,C", [(3, 128), (7, 768), (200, 4096)])
def test_tmix_post(rows, C):
    from chirrup_amd import ops

    rng = np.random.default_rng(rows * 3 + C)
    H = C // 64
    y = (rng.standard_normal((rows, C)) * 2).astype(F16)
    r, k, v, g = (rng.standard_normal((rows, C)).astype(F16) for _ in range(4))
    r_k = (0.1 * rng.standard_normal(C)).astype(F16)
    w = (1 + 0.1 * rng.standard_normal(C)).astype(F16)
    b = (0.1 * rng.standard_normal(C)).astype(F16)
    gn = M.group_norm_heads(y, H, w, b)
    bonus = ((r * k * r_k).reshape(rows, H, 64).astype(F32).sum(-1, keepdims=True, dtype=F32)).astype(F16)
    want = (gn + (bonus * v.reshape(rows, H, 64)).reshape(rows, C)) * g
    out = torch.empty((rows, C), dtype=torch.float16, device="cuda")
    ops.tmix_post(rows, C, cu(y), cu(r), cu(k), cu(v), cu(g), cu(r_k), cu(w), cu(b), 64e-5, out)
    assert_close_ulps(out.cpu().numpy(), want, 2, 0.03, "out")


def test_relu_sq_exact():
    from chirrup_amd import ops

    rng = np.random.default_rng(0)
    x = (rng.standard_normal((200, 16384)) * 3).astype(F16)
    x[0, :4] = [F16(-0.0), F16(0.0), F16(250.0), F16(300.0)]          # 300^2 overflows to inf like torch
    t = cu(x)
    ops.relu_sq_(t)
    with np.errstate(over="ignore"):
        want = np.maximum(x, F16(0)) * np.maximum(x, F16(0))
    assert np.array_equal(bits(t.cpu().numpy()), bits(want))


@pytest.mark.parametrize("B,T,C,layer0", [(3, 1, 128, True), (4, 3, 256, False), (200, 1, 4096, False)])
def test_fused_tmix_core_equals_three_kernel_path(oracle, B, T, C, layer0):
    """rwkv7_tmix_wkv7_fused (gating + WKV7 + group-norm/bonus/gate in one kernel) against tmix_mid ->
    wkv7_fwd_seq -> tmix_post.  Same per-op roundings; only the order of the binary32 head reductions
    differs (8-lane partials vs a 64-lane butterfly), so outputs and state may differ by an ulp on a
    small fraction of elements, never more."""
    from chirrup_amd import ops
    from util import wkv7_inputs

    rng = np.random.default_rng(B + T + C)
    H, rows = C // 64, B * T
    state, r, w, k, v, _, _, et = wkv7_inputs(B, T, C, seed=B * 3 + C)
    mk = lambda s=1.0: (rng.standard_normal((B, T, C)) * s).astype(F16)
    a_pre, vg_pre, v_first, g = mk(), mk(), mk(), mk()
    k_k = (0.85 + 0.05 * rng.standard_normal(C)).astype(F16)
    k_a = (1 + 0.05 * rng.standard_normal(C)).astype(F16)
    r_k = (0.1 * rng.standard_normal(C)).astype(F16)
    lw = (1 + 0.1 * rng.standard_normal(C)).astype(F16)
    lb = (0.1 * rng.standard_normal(C)).astype(F16)
    # three-kernel path
    tk, tv = cu(k), cu(v)
    nk, kka = torch.empty_like(tk), torch.empty_like(tk)
    ops.tmix_mid(rows, C, tk, tv, cu(a_pre), None if layer0 else cu(vg_pre), None if layer0 else cu(v_first), cu(k_k), cu(k_a), nk, kka)
    S3 = cu(state)
    y = torch.empty((B, T, C), dtype=torch.float16, device="cuda")
    ops.forward_seq(B, T, C, H, S3, cu(r), cu(w), tk, tv, nk, kka, y, cu(et))
    out3 = torch.empty_like(y)
    ops.tmix_post(rows, C, y, cu(r), tk, tv, cu(g), cu(r_k), cu(lw), cu(lb), 64e-5, out3)
    # fused
    S1 = cu(state)
    out1 = torch.empty((B, T, C), dtype=torch.float16, device="cuda")
    ops.tmix_wkv7_fused(B, T, C, H, S1, cu(r), cu(w), cu(k), cu(v), cu(a_pre), None if layer0 else cu(vg_pre),
                        None if layer0 else cu(v_first), cu(g), cu(k_k), cu(k_a), cu(r_k), cu(lw), cu(lb), 64e-5, out1, cu(et))
    s_a, s_b = S1.cpu().numpy(), S3.cpu().numpy()
    scale = max(1.0, float(np.abs(s_b.astype(F32)).max()))
    assert (bits(s_a) != bits(s_b)).mean() < 0.02
    assert float(np.abs(s_a.astype(F32) - s_b.astype(F32)).max()) <= 2e-3 * scale
    assert_close_ulps(out1.cpu().numpy(), out3.cpu().numpy(), 2, 0.05, "out", atol=4e-3 * max(1.0, float(out3.abs().max())))


@pytest.mark.parametrize("B,T,C,layer0", [(3, 1, 128, True), (4, 3, 256, False), (64, 1, 4096, False)])
def test_fused_tmix_core_with_the_mm8_prologue_is_the_same_kernel(B, T, C, layer0):
    """The mm8 prologue of the fused time-mix core is a template form of its own (as a run-time branch it cost the binary16 form
    a wave per SIMD).  With ry = 1 the prologue's xs = binary16(o * ry) IS o: both forms must write the same output
    and state bits, and the prologue's third row sum must be the sum of o over the head."""
    from chirrup_amd import ops
    from util import wkv7_inputs

    rng = np.random.default_rng(B + T + C + 7)
    H = C // 64
    state, r, w, k, v, _, _, et = wkv7_inputs(B, T, C, seed=B * 5 + C)
    mk = lambda s=1.0: (rng.standard_normal((B, T, C)) * s).astype(F16)
    a_pre, vg_pre, v_first, g = mk(), mk(), mk(), mk()
    k_k = (0.85 + 0.05 * rng.standard_normal(C)).astype(F16)
    k_a = (1 + 0.05 * rng.standard_normal(C)).astype(F16)
    r_k = (0.1 * rng.standard_normal(C)).astype(F16)
    lw = (1 + 0.1 * rng.standard_normal(C)).astype(F16)
    lb = (0.1 * rng.standard_normal(C)).astype(F16)
    res = []
    for prologue in (False, True):
        S = cu(state)
        out = torch.empty((B, T, C), dtype=torch.float16, device="cuda")
        mm8_out = None
        if prologue:
            sums = torch.full((B * T, H, 3), float("nan"), dtype=torch.float32, device="cuda")
            mm8_out = (torch.ones(C, dtype=torch.float16, device="cuda"), torch.zeros(C, dtype=torch.float16, device="cuda"), sums)
        ops.tmix_wkv7_fused(B, T, C, H, S, cu(r), cu(w), cu(k), cu(v), cu(a_pre), None if layer0 else cu(vg_pre),
                            None if layer0 else cu(v_first), cu(g), cu(k_k), cu(k_a), cu(r_k), cu(lw), cu(lb), 64e-5, out, cu(et),
                            mm8_out=mm8_out)
        torch.cuda.synchronize()
        res.append((out, S, mm8_out[2] if prologue else None))
    assert torch.equal(res[0][0].view(torch.int16), res[1][0].view(torch.int16))
    assert torch.equal(res[0][1].view(torch.int16), res[1][1].view(torch.int16))
    sums = res[1][2]
    want = res[0][0].float().view(B * T, H, 64).sum(-1)
    assert torch.allclose(sums[..., 0], want, rtol=1e-5, atol=1e-3) and torch.allclose(sums[..., 2], want, rtol=1e-5, atol=1e-3)
    assert float(sums[..., 1].abs().max()) == 0.0


@pytest.mark.parametrize("rows,C", [(5, 128), (200, 1024)])
def test_mm8_chain_folded_into_neighbouring_kernels_equals_mm8t_linear(rows, C):
    """Decode-regime mm8 FFN: the activation prologue of ffn.key written by the LN kernel, corrections + relu^2 of ffn.key
    and the prologue of ffn.value in ONE row kernel, corrections of ffn.value in the next LN kernel -- the same operations
    and rounding points as the stand-alone mm8t_linear calls (bit-identical up to the order of the binary32 row sums)."""
    from chirrup_amd import ops
    from chirrup_amd.quant import quantize_linear

    torch.manual_seed(rows + C)
    dev = "cuda"
    K8 = quantize_linear((torch.randn(4 * C, C, device=dev) / C ** 0.5).half())
    V8 = quantize_linear((torch.randn(C, 4 * C, device=dev) / (4 * C) ** 0.5).half())
    x = torch.randn(rows, 1, C, device=dev).half()
    att = (torch.randn(rows, 1, C, device=dev) * 0.3).half()
    w, b = (1 + 0.1 * torch.randn(C, device=dev)).half(), (0.1 * torch.randn(C, device=dev)).half()
    prev, mix = torch.randn(rows, C, device=dev).half(), torch.rand(1, C, device=dev).half()
    # reference: separate launches
    x_ref, prev_ref = x.clone(), prev.clone()
    kin_ref = torch.empty(1, rows, 1, C, dtype=torch.float16, device=dev)
    ops.add_ln_mix(rows, 1, C, x_ref, att, x_ref, w, b, 1e-5, prev_ref, prev_ref, mix, kin_ref)
    kf_ref = ops.mm8t_linear(kin_ref[0].view(rows, C), *K8, act=1)
    d_ref = ops.mm8t_linear(kf_ref, *V8)
    x2_ref, out_ref = x_ref.clone(), torch.empty(1, rows, 1, C, dtype=torch.float16, device=dev)
    ops.add_ln_mix(rows, 1, C, x2_ref, d_ref.view(rows, 1, C), x2_ref, w, b, 1e-5, None, None, None, out_ref)
    # fused chain
    f32 = dict(dtype=torch.float32, device=dev)
    x_f, prev_f = x.clone(), prev.clone()
    kin = torch.empty(1, rows, 1, C, dtype=torch.float16, device=dev)
    xs_k, S_k = torch.empty(rows, C, dtype=torch.float16, device=dev), torch.empty(rows, 3, **f32)
    xs_v, S_v = torch.empty(rows, 4 * C, dtype=torch.float16, device=dev), torch.empty(rows, ops.mm8_row_parts(4 * C), 3, **f32)
    ops.add_ln_mix(rows, 1, C, x_f, att, x_f, w, b, 1e-5, prev_f, prev_f, mix, kin, mm8_out=(K8.ry, K8.my, xs_k, S_k))
    assert torch.equal(kin, kin_ref) and torch.equal(prev_f, prev_ref)
    pk = torch.empty(16, rows, 4 * C, **f32)
    kparts = ops.mm8t_gemm_partial(xs_k, K8.qT, 4 * C, 0, pk)
    kf = torch.empty(rows, 4 * C, dtype=torch.float16, device=dev)
    ops.mm8_reduce_rows(kparts, K8.rx, K8.mx, S_k, act=1, y=kf, nxt=(V8.ry, V8.my, xs_v, S_v))
    assert torch.equal(kf, kf_ref)
    pv = torch.empty(16, rows, C, **f32)
    vparts = ops.mm8t_gemm_partial(xs_v, V8.qT, C, 0, pv)
    out = torch.empty(1, rows, 1, C, dtype=torch.float16, device=dev)
    ops.add_ln_mix(rows, 1, C, x_f, None, x_f, w, b, 1e-5, None, None, None, out, delta_partials=vparts.view(-1, rows, C),
                   mm8_in=(V8.rx, V8.mx, S_v))
    # the row sums S of ffn.value's prologue are binary32 sums in another order here (1024-lane row kernel vs the
    # stand-alone prologue kernel): a delta may move by one binary16 ulp on a few elements, never more
    dx = (x_f.float() - x2_ref.float()).abs()
    assert float(dx.max()) <= 4e-3 and float((dx > 0).float().mean()) < 0.10      # (the 1024*S0 term of the u8 offset makes delta that sensitive to the order of the S0 sum)
    assert float((out.float() - out_ref.float()).abs().max()) <= 8e-3


@pytest.mark.parametrize("B,T,C,V", [(5, 1, 128, 1000), (3, 4, 768, 65536), (200, 1, 4096, 65536)])
def test_embed_rows_and_advance_elapsed_equal_the_torch_ops(B, T, C, V):
    """The two ends of a decode step as one launch each (include/chirrup_amd.h: rwkv7_embed_rows, rwkv7_advance_elapsed) against
    the torch statements they replace: the embedding gather with fed-back ids for negative tokens (rwkv7.py:503-517 + the worker's
    id feedback), the slots' step counters in row order, the zeroing of the launch-sync words, and state[2] += T over a slot list."""
    from chirrup_amd import ops

    g = torch.Generator(device="cpu").manual_seed(B + T + C)
    dev = "cuda"
    emb = torch.randn(V, C, generator=g).half().to(dev)
    n_slots = B + 7
    slots = torch.randperm(n_slots, generator=g)[:B].to(torch.int32).to(dev)
    tokens = torch.randint(0, V, (B, T), generator=g).to(dev)
    tokens[::2, 0] = -1                                    # these rows take the id fed back for their slot
    feedback = torch.randint(0, V, (n_slots,), generator=g).to(torch.int32).to(dev)
    pool = torch.randint(0, 1000, (n_slots,), generator=g).to(torch.int32).to(dev)
    arena, _ = ops._sync_arena(emb.device)
    arena.fill_(7)
    x, rows = ops.embed_rows(emb, tokens, slots, feedback, zero_sync=True, elapsed_pool=pool)
    want_tok = torch.where(tokens < 0, feedback[slots.long()].long().view(B, 1), tokens)
    assert torch.equal(x.view(torch.int16), emb[want_tok].view(torch.int16))
    assert torch.equal(rows, pool[slots.long()])
    assert int(arena.abs().sum()) == 0
    # without slots / feedback / extras: the plain gather; an id outside the table gives a zero row
    tok2 = torch.randint(0, V, (B, T), generator=g).to(dev)
    assert torch.equal(ops.embed_rows(emb, tok2).view(torch.int16), emb[tok2].view(torch.int16))
    tok2[0, 0] = V + 5
    assert float(ops.embed_rows(emb, tok2)[0, 0].abs().max()) == 0.0
    # state[2] += T
    before = pool.clone()
    ops.advance_elapsed(pool, T, slots)
    want = before.clone()
    want[slots.long()] += T
    assert torch.equal(pool, want)
    flat = torch.arange(B, dtype=torch.int32, device=dev)
    ops.advance_elapsed(flat, 3)
    assert torch.equal(flat, torch.arange(B, dtype=torch.int32, device=dev) + 3)


def test_copy_slot_rows_equals_index_copy():
    from chirrup_amd import ops

    g = torch.Generator(device="cpu").manual_seed(3)
    for n_slots, B, C in ((9, 4, 128), (201, 200, 4096)):
        src = torch.randn(n_slots, C, generator=g).half().cuda()
        dst = torch.randn(n_slots, C, generator=g).half().cuda()
        slots = torch.randperm(n_slots, generator=g)[:B].to(torch.int32).cuda()
        want = dst.clone()
        want.index_copy_(0, slots.long(), src.index_select(0, slots.long()))
        ops.copy_slot_rows(src, dst, slots)
        assert torch.equal(dst.view(torch.int16), want.view(torch.int16))
