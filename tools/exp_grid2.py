"""2 x 2 wave grid of the ring GEMM (a compile-time variant of skinny_gemm.hip measured in round 4, profiles/r04_gemm_2x2_wave_grid_u8.txt; the tool is a digest + timing A/B of any two library builds) against the strip layout: the ffn.key / ffn.value launches of a
7.2B decode step at `rows` rows, uint8 and binary16 weights -- a digest of every launch's output (the layouts must agree bit for
bit) and us per launch (graph replay over rotating weights).  Run once per library build (CHIRRUP_AMD_LIB=...)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import lib, ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C, dev, NW = 4096, "cuda:0", 8
torch.manual_seed(0)
halves = rows >= 128


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]


def timed(run, n):
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 / n * 1e3


out = []
for shape, (N, K) in (("ffn.key", (4 * C, C)), ("ffn.value", (C, 4 * C))):
    x = torch.randn(rows, K, device=dev).half()
    # binary16
    W = [ops.tile_weight((torch.randn(N, K, device=dev) / K ** 0.5).half()) for _ in range(NW)]
    parts = torch.empty(8 * rows * N, dtype=torch.float32, device=dev)
    if shape == "ffn.key":
        f16 = lambda w: ops.skinny_linear(x, w, act=1, splits=1, row_halves=halves)
    else:
        f16 = lambda w: ops.skinny_linear_partial(x, w, 8, parts)
    y = f16(W[0])
    out.append((shape + " f16", digest(y), timed(lambda: [f16(w) for w in W], NW)))
    del W
    # uint8
    Q = [ops.tile_weight_u8(torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8)) for _ in range(NW)]
    xs = (x * 0.06).half()
    if shape == "ffn.key":
        rx, mx = (torch.rand(N, device=dev) / 16 + 0.03).half(), (torch.randn(N, device=dev) * 0.05).half()
        ry2, my2 = (torch.rand(N, device=dev) / 16 + 0.03).half(), (torch.randn(N, device=dev) * 0.05).half()
        S = torch.randn(rows, 3, device=dev)
        xs2 = torch.empty((rows, N), dtype=torch.float16, device=dev)
        S2 = torch.empty((rows, ops.mm8_tile_parts(N), 3), dtype=torch.float32, device=dev)
        u8 = lambda q: ops.mm8t_gemm_fused(xs, q, N, rx, mx, S, act=1, nxt=(ry2, my2, xs2, S2), tiled=True)
        u8(Q[0])
        d = digest(xs2) + "/" + digest(S2)
    else:
        u8 = lambda q: ops.mm8t_gemm_partial(xs, q, N, 8, parts, tiled=True)
        d = digest(u8(Q[0]))
    out.append((shape + " u8", d, timed(lambda: [u8(q) for q in Q], NW)))
    del Q
name = os.path.basename(lib.LIB_PATH)
for shape, d, us in out:
    print(f"{name:32s} rows {rows} {shape:14s} digest {d}  {us:7.2f} us/launch", flush=True)
