"""The diagnostic entry points behind tools/ (include/chirrup_amd.h: skinny_gemm_clock_probe with its timeline, skinny_gemm_warm_probe,
rwkv7_ln_probe, chirrup_noop_launch): they must not change any result, must switch off cleanly, and their stamps must be ordered --
the profiles under profiles/r03_* rest on them."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gemm_clock_probe_and_timeline_are_ordered_and_leave_results_alone():
    from chirrup_amd import lib, ops

    torch.manual_seed(0)
    M, N, K = 200, 2048, 1024
    x = torch.randn(M, K, device="cuda").half()
    w = ops.tile_weight((torch.randn(N, K, device="cuda") / K ** 0.5).half())
    want = ops.skinny_linear(x, w, act=1, splits=1, row_halves=True)
    L = lib.load()
    pairs = 1024
    buf = torch.zeros(2 * pairs, dtype=torch.int64, device="cuda")
    L.skinny_gemm_clock_probe(buf.data_ptr(), pairs)
    try:
        got = ops.skinny_linear(x, w, act=1, splits=1, row_halves=True)
        torch.cuda.synchronize()
    finally:
        L.skinny_gemm_clock_probe(None, 0)
    assert torch.equal(got, want)
    wgs = 2 * (N // 128)
    dur = buf[: 2 * wgs].view(wgs, 2)
    assert bool((dur > 0).all())
    mhz = dur[:, 0].double() / dur[:, 1].double() * 100.0
    assert 500.0 < float(mhz.median()) < 3000.0        # shader clock from the two counters
    tl = buf[2 * wgs: 2 * wgs + 4 * wgs].view(wgs, 4)
    assert bool((tl[:, 0] <= tl[:, 1]).all() and (tl[:, 1] <= tl[:, 2]).all() and (tl[:, 2] <= tl[:, 3]).all())
    assert float((tl[:, 3].max() - tl[:, 0].min())) / 100.0 < 1000.0        # one launch: well under a millisecond
    buf.zero_()
    ops.skinny_linear(x, w, act=1, splits=1, row_halves=True)              # switched off: nothing is written
    torch.cuda.synchronize()
    assert int(buf.abs().sum()) == 0


def test_warm_probe_changes_no_result():
    from chirrup_amd import lib, ops

    torch.manual_seed(1)
    x = torch.randn(32, 2048, device="cuda").half()
    w = ops.tile_weight((torch.randn(1024, 2048, device="cuda") / 45).half())
    wr = (torch.randn(640, 2048, device="cuda") / 45).half()
    want = ops.skinny_linear(x, w, splits=0), ops.skinny_linear(x, wr, splits=2)
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    L = lib.load()
    for mode in (3, 103, 303, 201):
        L.skinny_gemm_warm_probe(mode, sink.data_ptr())
        try:
            got = ops.skinny_linear(x, w, splits=0), ops.skinny_linear(x, wr, splits=2)
            torch.cuda.synchronize()
        finally:
            L.skinny_gemm_warm_probe(0, None)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), mode
    assert int(sink.abs().sum()) == 0


def test_ln_probe_stamps_are_ordered():
    from chirrup_amd import lib, ops

    torch.manual_seed(2)
    B, C = 24, 1024
    dev = "cuda"
    x = torch.randn(B, 1, C, device=dev).half()
    delta = torch.randn(B, 1, C, device=dev).half()
    ln_w, ln_b = torch.rand(C, device=dev).half(), torch.randn(C, device=dev).half() * 0.1
    prev = torch.randn(B, C, device=dev).half()
    mix = torch.rand(6, C, device=dev).half()

    def run(probe):
        xo, po = torch.empty_like(x), prev.clone()
        out = torch.empty(6, B, 1, C, device=dev, dtype=torch.float16)
        buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
        L = lib.load()
        if probe:
            L.rwkv7_ln_probe(buf.data_ptr())
        try:
            ops.add_ln_mix(B, 1, C, x, delta, xo, ln_w, ln_b, 1e-5, po, po, mix, out)
            torch.cuda.synchronize()
        finally:
            L.rwkv7_ln_probe(None)
        return xo, po, out, buf

    a, b = run(False), run(True)
    assert all(torch.equal(p, q) for p, q in zip(a[:3], b[:3]))
    assert int(a[3].abs().sum()) == 0
    st = b[3].view(-1, 8)[:B, :7]
    assert bool((st > 0).all()) and bool((st[:, 1:] >= st[:, :-1]).all())


def test_noop_launch_validates_and_runs():
    from chirrup_amd import lib

    L = lib.load()
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert L.chirrup_noop_launch(256, 512, 150 << 10, 0, sink.data_ptr(), st) == 0
    assert L.chirrup_noop_launch(8, 64, 0, 1, sink.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert L.chirrup_noop_launch(0, 64, 0, 0, sink.data_ptr(), st) < 0
    assert L.chirrup_noop_launch(8, 2048, 0, 0, sink.data_ptr(), st) < 0
    assert L.chirrup_noop_launch(8, 64, 200 << 10, 0, sink.data_ptr(), st) < 0
    assert int(sink.abs().sum()) == 0
