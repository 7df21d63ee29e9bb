"""Kernel-only timing of the FFN / att.output GEMM shapes (7.2B, tile-image weights, partial form) -- the quick A/B used
with alternative builds of the library (CHIRRUP_AMD_LIB=tools/build/lib....so) and CHIRRUP_GEMM_BN.
    python tools/exp_gemm_quick.py [M] [splits_key] [splits_value] [splits_att]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sv = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sa = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev, C = "cuda:0", 4096
torch.manual_seed(0)


def timeit(fn, iters=20):
    for _ in range(2):
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
    return e0.elapsed_time(e1) / iters


out = [os.path.basename(os.environ.get("CHIRRUP_AMD_LIB", "in-tree")), "BN=" + os.environ.get("CHIRRUP_GEMM_BN", "auto")]
for name, N, K, s in (("ffn.key", 4 * C, C, sk), ("ffn.value", C, 4 * C, sv), ("att.out", C, C, sa)):
    nw = 6 if N * K > C * C else 12
    Wt = [ops.tile_weight((torch.randn(N, K, device=dev) / K ** 0.5).half()) for _ in range(nw)]
    x = torch.randn(M, K, device=dev).half()
    part = torch.empty(16, M, N, device=dev, dtype=torch.float32)
    t = timeit(lambda: [ops.skinny_linear_partial(x, W, s, part) for W in Wt]) / nw
    out.append(f"{name} s{ops.gemm_splits(N, K, 1, s)}: {t*1e3:6.1f} us ({N*K*2/t/1e6:5.0f} GB/s)")
    if os.environ.get("EXP_STAMPS"):      # builds with WIDE_EXP & 16: per-workgroup (cycles, 100-MHz ticks) of the main loop in plane 15
        torch.cuda.synchronize()
        st = part[15].view(torch.int32).view(-1)[: 512].view(256, 2).cpu().double()
        clk = (st[:, 0] / st[:, 1].clamp_min(1)) * 0.1
        out[-1] += f" [main loop: {st[:, 1].median() / 100:.1f} us, clock {clk.median():.2f} GHz (min {clk.min():.2f} max {clk.max():.2f})]"
    del Wt, part
print(" | ".join(out), flush=True)
