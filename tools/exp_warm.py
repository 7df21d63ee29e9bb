"""Upside of warming the L2 for a GEMM's first K-blocks from the kernel BEFORE it (skinny_gemm_warm_probe): run under
rocprofv3 --kernel-trace --stats and compare the ring_gemm_kernel average with stages = 0 and stages > 0.

    rocprofv3 --kernel-trace --stats -d out -- python3 tools/exp_warm.py <stages> <rows> <C> <shape: key|value|out>
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import ops
from chirrup_amd import lib as _lib_mod

stages, M, C, shape = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev = "cuda:0"
N, K = {"key": (4 * C, C), "value": (C, 4 * C), "out": (C, C)}[shape]
NW = max(6, int(1.5e9 / (N * K * 2)))            # rotate over 1.5 GB of weights: nothing survives in the L2s or the MALL
NW = min(NW, 64)
torch.manual_seed(0)
W = [ops.tile_weight((torch.randn(N, K, device=dev) / K ** 0.5).half()) for _ in range(NW)]
x = torch.randn(M, K, device=dev).half()
sink = torch.zeros(4, dtype=torch.int32, device=dev)
halves = M >= 128
parts = torch.empty(8 * M * N, dtype=torch.float32, device=dev)


def run():
    for w in W:
        if shape == "key":
            ops.skinny_linear(x, w, act=1, splits=0, row_halves=halves)
        elif shape == "value":
            ops.skinny_linear_partial(x, w, 8, parts)
        else:
            ops.skinny_linear(x, w, splits=0, row_halves=halves)


_lib_mod.load().skinny_gemm_warm_probe(stages, sink.data_ptr())
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
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"stages {stages} rows {M} C {C} {shape}: {e0.elapsed_time(e1) / 10 / NW * 1e3:.2f} us per GEMM (+ warm launch if any), {NW} weights")
_lib_mod.load().skinny_gemm_warm_probe(0, None)
