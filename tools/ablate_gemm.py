"""What bounds the ring GEMM's main loop: the same launch from the product library and from the diagnostic builds with parts of
the loop removed (csrc/Makefile: `make ablate A=<bits>`, skinny_gemm.hip RING_ABLATE: 1 no MFMAs, 2 no loads, 4 no fragment reads).
Run once per library (CHIRRUP_AMD_LIB=...):  python tools/ablate_gemm.py <key|value> [rows]
Prints: us per launch (graph replay over rotating weights), main-loop us / shader cycles / MHz (median over workgroups, in-kernel
stamps of the last of the back-to-back launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import lib, ops

shape = sys.argv[1] if len(sys.argv) > 1 else "key"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 200
C, dev, NW = 4096, "cuda:0", 10
N, K = (4 * C, C) if shape.startswith("key") else (C, 4 * C)
if shape.startswith("key") and len(shape) > 3:       # key<N>: fewer tiles -> fewer workgroups (2 per 128 columns), same K
    N = int(shape[3:])
    NW = 40
torch.manual_seed(0)
W = [ops.tile_weight((torch.randn(N, K, device=dev) / K ** 0.5).half()) for _ in range(NW)]
x = torch.randn(M, K, device=dev).half()
parts = torch.empty(8 * M * N, dtype=torch.float32, device=dev)
halves = M >= 128


def run():
    for w in W:
        if shape.startswith("key"):
            ops.skinny_linear(x, w, act=1, splits=1, row_halves=halves)
        else:
            ops.skinny_linear_partial(x, w, 8, parts)


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
pairs = 4096
cb = torch.zeros((2 * pairs,), dtype=torch.int64, device=dev)
for _ in range(20):
    g.replay()
L_.skinny_gemm_clock_probe(cb.data_ptr(), pairs)
run()
torch.cuda.synchronize()
L_.skinny_gemm_clock_probe(None, 0)
v = cb.view(pairs, 2)
v = v[(v[:, 1] > 0) & (v[:, 1] < 10 ** 7) & (v[:, 0] < 10 ** 9)].double()       # (durations: the timeline's absolute stamps follow the pairs)
cyc = v[:, 0].sort().values[len(v) // 2]
rt = (v[:, 1] / 100.0).sort().values[len(v) // 2]
mhz = (v[:, 0] / v[:, 1] * 100.0).sort().values[len(v) // 2]
print(f"{os.path.basename(lib.LIB_PATH):30s} {shape} rows {M}: {us:7.2f} us/launch; main loop {float(rt):6.2f} us = {float(cyc) / 1e3:6.1f} k cycles at {float(mhz):6.0f} MHz ({len(v)} workgroups)", flush=True)
