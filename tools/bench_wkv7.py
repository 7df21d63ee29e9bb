"""Micro-benchmark of the WKV7 kernel alone at a model's layer shape (HIP-event timed).
usage: python tools/bench_wkv7.py [B] [C] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd import ops

FUSED = "--fused" in sys.argv
sys.argv = [a for a in sys.argv if a != "--fused"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
L = 32  # rotate over L distinct layer states so the 256 MiB Infinity Cache cannot hold them
H = C // 64
dev = "cuda:0"
torch.manual_seed(0)
state = (torch.randn(L, B, H, 64, 64, device=dev) * 0.1).half()
mk = lambda: torch.randn(B, 1, C, device=dev).half()
r, k, v = mk(), mk(), mk()
w = (torch.rand(B, 1, C, device=dev) * 12 - 8).half()
a = (torch.randn(B, 1, C, device=dev) * 0.125).half()
b = (torch.randn(B, 1, C, device=dev) * 0.06).half()
y = torch.empty(B, 1, C, device=dev, dtype=torch.float16)
et = (torch.arange(B, device=dev, dtype=torch.int32) * 7 + 3)
vg, vf, gg = mk(), mk(), mk()
pk = lambda s, o=0.0: (torch.randn(C, device=dev) * s + o).half()
k_k, k_a, r_k, lw_, lb_ = pk(0.05, 0.85), pk(0.05, 1.0), pk(0.1), pk(0.1, 1.0), pk(0.1)
if FUSED:
    _plain = ops.forward_seq
    def _fused(B_, T_, C_, H_, st, r_, w_, k_, v_, a_, b_, y_, et_):
        ops.tmix_wkv7_fused(B_, T_, C_, H_, st, r_, w_, k_, v_, a_, vg, vf, gg, k_k, k_a, r_k, lw_, lb_, 64e-5, y_, et_)
    ops.forward_seq = _fused
for l in range(L):
    ops.forward_seq(B, 1, C, H, state[l], r, w, k, v, a, b, y, et)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(iters):
    for l in range(L):
        ops.forward_seq(B, 1, C, H, state[l], r, w, k, v, a, b, y, et)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (iters * L)
byts = B * ((274 if FUSED else 270) * C + 4)
print(f"wkv7{' fused core' if FUSED else ''} B={B} C={C}: {ms*1e3:.1f} us/launch  {byts/ms/1e6:.1f} GB/s algorithmic  ({byts/ms/1e6/8000*100:.1f}% of 8 TB/s)")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for l in range(L):
        ops.forward_seq(B, 1, C, H, state[l], r, w, k, v, a, b, y, et)
g.replay(); torch.cuda.synchronize()
e0.record()
for it in range(iters):
    g.replay()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (iters * L)
print(f"  graph replay: {ms*1e3:.1f} us/launch  {byts/ms/1e6:.1f} GB/s")
