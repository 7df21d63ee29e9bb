"""Timeline of ONE ring-GEMM launch from in-kernel 100-MHz stamps (include/chirrup_amd.h: skinny_gemm_clock_probe): when the
workgroups enter, start and end their main loops and finish their epilogues, against the launch's duration as events see it.
    python tools/gemm_timeline.py <key|value|out|key8|value8> [rows] [C]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import lib, ops

shape = sys.argv[1] if len(sys.argv) > 1 else "key"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 200
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
dev, NW = "cuda:0", 10
N, K = {"key": (4 * C, C), "value": (C, 4 * C), "out": (C, C), "key8": (4 * C, C), "value8": (C, 4 * C)}[shape]
torch.manual_seed(0)
if shape.endswith("8"):        # uint8 (mm8) weights as the decode step launches them
    W = [ops.tile_weight_u8(torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8)) for _ in range(NW)]
    rx, mx = torch.rand(N, device=dev).half() / 64, torch.randn(N, device=dev).half() * 0.01
    ry2, my2 = torch.rand(N, device=dev).half() / 16, torch.randn(N, device=dev).half() * 0.01
    S = torch.zeros(M, 1, 3, device=dev)
    xs2, S2 = torch.empty(M, N, device=dev, dtype=torch.float16), torch.empty(M, ops.mm8_tile_parts(N), 3, device=dev)
else:
    W = [ops.tile_weight((torch.randn(N, K, device=dev) / K ** 0.5).half()) for _ in range(NW)]
x = torch.randn(M, K, device=dev).half()
parts = torch.empty(8 * M * N, dtype=torch.float32, device=dev)
halves = M >= 128


def one(w):
    if shape == "key":
        ops.skinny_linear(x, w, act=1, splits=0, row_halves=halves)
    elif shape == "value":
        ops.skinny_linear_partial(x, w, 8, parts)
    elif shape == "key8":
        ops.mm8t_gemm_fused(x, w, N, rx, mx, S, act=1, nxt=(ry2, my2, xs2, S2), tiled=True)
    elif shape == "value8":
        ops.mm8t_gemm_partial(x, w, N, 8, parts, tiled=True)
    else:
        ops.skinny_linear_partial(x, w, 0, parts, row_halves=halves)


def run():
    for w in W:
        one(w)


run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    g.replay()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 / NW * 1e3
L_ = lib.load()
pairs = 8192
cb = torch.zeros((2 * pairs,), dtype=torch.int64, device=dev)
for _ in range(10):
    g.replay()
L_.skinny_gemm_clock_probe(cb.data_ptr(), pairs)
run()                                                  # eager, back to back: the LAST launch's stamps stay in the buffer
torch.cuda.synchronize()
L_.skinny_gemm_clock_probe(None, 0)
v = cb.view(pairs, 2)
n = int(((v[:, 1] > 0) & (v[:, 1] < 10 ** 7) & (v[:, 0] < 10 ** 9)).sum())      # workgroups that ran a main loop: durations, not absolute stamps
wg_total = n
# the pairs of padding workgroups are zero; the timeline starts behind 2 x (all workgroups of the grid): find it by the launch's grid
for cand in range(n, n + 16):
    tl = cb[2 * cand: 2 * cand + 4 * cand].view(cand, 4)
    if int((tl[:, 0] > 0).sum()) == n and int(tl[:, 0][tl[:, 0] > 0].min()) > 10 ** 6:
        wg_total = cand
        break
tl = cb[2 * wg_total: 2 * wg_total + 4 * wg_total].view(wg_total, 4).double()
tl = tl[tl[:, 0] > 0]
t0 = tl[:, 0].min()
tl = (tl - t0) / 100.0                                 # us since the first workgroup's entry
q = lambda c, f: float(tl[:, c].sort().values[min(len(tl) - 1, int(f * len(tl)))])
print(f"{shape} rows {M} C {C}: {us:.2f} us per launch (events, graph replay over {NW} rotating weights); {len(tl)} workgroups, stamps of one launch (us since the first entry)")
for c, name in enumerate(("kernel entry", "main loop start", "main loop end", "epilogue done")):
    print(f"  {name:16s} min {q(c, 0):6.2f}  median {q(c, 0.5):6.2f}  p90 {q(c, 0.9):6.2f}  max {float(tl[:, c].max()):6.2f}")
d = tl[:, 2] - tl[:, 1]
print(f"  main loop: median {float(d.median()):.2f} us, max {float(d.max()):.2f};  entry -> loop start median {float((tl[:, 1] - tl[:, 0]).median()):.2f};  "
      f"loop end -> done median {float((tl[:, 3] - tl[:, 2]).median()):.2f};  first entry -> last done {float(tl[:, 3].max()):.2f}")
