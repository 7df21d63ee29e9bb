"""Launch time of the layer's largest GEMM (ffn.key, 16384 x 4096) against the number of token rows, binary16 and uint8 (mm8)
weights, as the decode step launches them (>= 128 rows: unsplit, two row halves per tile, epilogue in the launch): the data
behind "what the uint8 weights can and cannot buy at bsz 200" (DESIGN.md).  Graph replay over rotating weights.

    python tools/exp_rows_sweep.py [out.txt]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import ops

dev, C = "cuda:0", 4096
N, K = 4 * C, C
torch.manual_seed(0)
NW = 6
W16 = [ops.tile_weight((torch.randn(N, K, device=dev) / K ** 0.5).half()) for _ in range(NW)]
W8 = [ops.tile_weight_u8(torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8)) for _ in range(NW)]
rx, mx = torch.rand(N, device=dev).half() / 64, torch.randn(N, device=dev).half() * 0.01
ry2, my2 = torch.rand(N, device=dev).half() / 16, torch.randn(N, device=dev).half() * 0.01
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / NW * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


say("# ffn.key (N = 16384, K = 4096), unsplit launch with two row halves per tile, us per launch (median of 10 graph replays over 6 rotating weights)")
say(f"{'rows':>5} {'f16 us':>8} {'u8 us':>8} {'u8 / f16':>9}")
pts = []
for M in (128, 144, 160, 176, 192, 200, 208, 224, 240, 256):
    x = torch.randn(M, K, device=dev).half()
    S = torch.zeros(M, 1, 3, device=dev)
    xs2, S2 = torch.empty(M, N, device=dev, dtype=torch.float16), torch.empty(M, ops.mm8_tile_parts(N), 3, device=dev)
    t16 = timeit(lambda: [ops.skinny_linear(x, w, act=1, splits=0, row_halves=True) for w in W16])
    t8 = timeit(lambda: [ops.mm8t_gemm_fused(x, w, N, rx, mx, S, act=1, nxt=(ry2, my2, xs2, S2), tiled=True) for w in W8])
    pts.append((M, t16, t8))
    say(f"{M:5d} {t16:8.2f} {t8:8.2f} {t8 / t16:9.3f}")
# least-squares line t = a + b * rows over the sweep
import numpy as np

m = np.array([p[0] for p in pts], float)
for name, col in (("f16", 1), ("u8", 2)):
    t = np.array([p[col] for p in pts])
    b, a = np.polyfit(m, t, 1)
    say(f"{name}: t = {a:.1f} us + {b:.4f} us/row x rows   (weights: {N * K * (2 if name == 'f16' else 1) / 1e6:.0f} MB; at 6.3 TB/s: {N * K * (2 if name == 'f16' else 1) / 6.3e6:.1f} us)")
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
