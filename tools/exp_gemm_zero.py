"""The same GEMM launches (7.2B decode shapes, bsz 200, as shipped) on RANDOM and on ALL-ZERO operands: identical instructions,
addresses and HBM bytes -- only the energy per MFMA / per transferred bit differs.  If the launches were bound by their
structure (issue slots, barriers, latencies, HBM or L2 bandwidth) the two would take the same time.
    python tools/exp_gemm_zero.py [M=200]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C, dev, NW = 4096, "cuda:0", 8
torch.manual_seed(0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, N, K, call in (("ffn.value (split 8, partials)", C, 4 * C, "partial"), ("ffn.key (unsplit, row halves, relu^2)", 4 * C, C, "key"),
                         ("att.output (split 4, row halves)", C, C, "att")):
    part = torch.empty(8, M, N, device=dev, dtype=torch.float32)
    res = {}
    for kind in ("random", "zero W", "zero x", "zero both"):
        Ws = [ops.tile_weight((torch.randn(N, K, device=dev) / K ** 0.5).half() if "W" not in kind and kind != "zero both" else torch.zeros(N, K, device=dev, dtype=torch.float16))
              for _ in range(NW)]
        x = torch.randn(M, K, device=dev).half() if kind in ("random", "zero W") else torch.zeros(M, K, device=dev, dtype=torch.float16)
        if call == "partial":
            fn = lambda: [ops.skinny_linear_partial(x, W, 8, part) for W in Ws]
        elif call == "key":
            fn = lambda: [ops.skinny_linear(x, W, act=1, splits=0, row_halves=True) for W in Ws]
        else:
            fn = lambda: [ops.skinny_linear_partial(x, W, 0, part, row_halves=True) for W in Ws]
        res[kind] = timeit(fn) / NW
        del Ws
    print(f"{name:40s} M={M}: " + "  ".join(f"{k} {v:5.1f} us" for k, v in res.items()), flush=True)
