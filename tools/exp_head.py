"""The head GEMM (65536 x 4096) at decode batch sizes: 256-column kernel (default) vs the 128-column ring kernel (CHIRRUP_GEMM_BN=128),
whole rows or two row halves per tile.   python tools/exp_head.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev, C, V = "cuda:0", 4096, 65536
torch.manual_seed(0)
W = [ops.tile_weight((torch.randn(V, C, device=dev) / C ** 0.5).half()) for _ in range(3)]
x = torch.randn(M, C, device=dev).half()
for halves in (False, True):
    def run():
        for w in W:
            ops.skinny_linear(x, w, splits=1, row_halves=halves)
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"BN={os.environ.get('CHIRRUP_GEMM_BN', 'auto')} rows {M} row_halves={halves}: {e0.elapsed_time(e1) / 10 / len(W) * 1e3:.1f} us per head GEMM", flush=True)
