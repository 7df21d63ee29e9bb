"""tools/gemm_intercept.py [rows]: what a ring-GEMM launch costs besides its K-blocks.  The ffn.key launch (N = 16384, unsplit, two row
halves: 256 workgroups, one per CU) timed in a HIP graph for K = 64 .. 4096 over rotating weights; a line through the points gives
the per-K-block time (the per-CU ingest) and the intercept -- dispatch, ring fill, epilogue and completion of a launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = 16384
dev = torch.device("cuda:0")
torch.manual_seed(0)
pts = []
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    nw = max(4, min(32, (512 << 20) // (N * K * 2)))          # > 256 MB of weights in rotation where that fits
    ws = [ops.tile_weight((torch.randn(N, K, device=dev) * 0.02).half()) for _ in range(nw)]
    x = (torch.randn(M, K, device=dev)).half()
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    for w in ws[:2]:
        ops.skinny_linear(x, w, act=1, splits=1, out=y, row_halves=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for w in ws:
            ops.skinny_linear(x, w, act=1, splits=1, out=y, row_halves=True)
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / nw)
    ts.sort()
    pts.append((K // 64, ts[len(ts) // 2]))
    print("K %5d (%2d K-blocks): %6.2f us per launch" % (K, K // 64, pts[-1][1]))
    del ws, g
    torch.cuda.empty_cache()
# least squares over the points with >= 8 K-blocks (below that the ring never fills)
sel = [(k, t) for k, t in pts if k >= 8]
n = len(sel); sx = sum(k for k, _ in sel); sy = sum(t for _, t in sel); sxx = sum(k * k for k, _ in sel); sxy = sum(k * t for k, t in sel)
slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
print("fit over K >= 512: %.3f us per K-block (%.1f GB/s per CU at %d + 128 rows x 128 B per block) + %.2f us per launch"
      % (slope, ((min(M, 112) + 128) * 128) / slope / 1e3, min(M, 112), icpt))
