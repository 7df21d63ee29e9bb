"""What a kernel node of a captured graph costs beyond its workgroups' work: empty workgroups of several shapes, 64 nodes per
graph, us per node (events around graph replays).  include/chirrup_amd.h: chirrup_noop_launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import lib

L = lib.load()
dev = "cuda:0"
sink = torch.zeros(4, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
print("grid x block, LDS per workgroup, idle per workgroup -> us per kernel node (graph of 64 nodes, median of 10 replays)")
for grid, block, lds, sleep in ((1, 64, 0, 0), (256, 64, 0, 0), (256, 512, 0, 0), (256, 512, 64 << 10, 0), (256, 512, 150 << 10, 0), (512, 512, 150 << 10, 0),
                                (2048, 512, 150 << 10, 0), (12800, 64, 9 << 10, 0), (200, 1024, 0, 0), (256, 512, 150 << 10, 1), (256, 512, 150 << 10, 4)):
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: [L.chirrup_noop_launch(grid, block, lds, sleep, sink.data_ptr(), torch.cuda.current_stream().cuda_stream) for _ in range(64)]
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 64 * 1e3)
    ts.sort()
    print(f"  {grid:6d} x {block:4d}, {lds >> 10:4d} KiB, {sleep} x 4 us: {ts[len(ts) // 2]:7.2f} us")
