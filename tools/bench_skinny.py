"""The decode step's GEMM launches (7.2B shapes) through the hand-written kernels vs torch (hipBLASLt), graph-replayed over
rotating tile-image weights (>> the 256 MiB Infinity Cache), for a sweep of K-split factors.  The kernel (128- or
256-column tiles) is the library's choice; CHIRRUP_GEMM_BN=128|256 in the environment forces one for an A/B.

    python tools/bench_skinny.py [M]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from chirrup_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
torch.manual_seed(0)
print("CHIRRUP_GEMM_BN =", os.environ.get("CHIRRUP_GEMM_BN", "(library's choice)"), " M =", M, flush=True)


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


C = 4096
for name, N, K, splits_list, act in [("att.output", C, C, (0, 2, 4, 8, 16), 0), ("ffn.key", 4 * C, C, (0, 1, 2, 4), 1),
                                     ("ffn.value", C, 4 * C, (0, 4, 8, 16), 0), ("head", 65536, C, (1,), 0)]:
    nw = 12 if N * K * 2 < 200e6 else 3
    Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(nw)]
    Wt = [ops.tile_weight(W) for W in Ws]
    x = torch.randn(M, K, device=dev).half()
    t0 = timeit(lambda: [F.linear(x, W) for W in Ws]) / nw
    line = f"{name:12s} N={N:6d} K={K:6d}: torch {t0*1e3:7.1f} us | full (GEMM + reduce):"
    for s in splits_list:
        t = timeit(lambda: [ops.skinny_linear(x, W, act=act, splits=s) for W in Wt]) / nw
        line += f" s{ops.gemm_splits(N, K, 1, s)}: {t*1e3:6.1f} us ({N*K*2/t/1e6:5.0f} GB/s)"
    print(line, flush=True)
    if name != "head":
        line = f"{'':12s} {'':35s} | GEMM kernel only (partials): "
        part = torch.empty(16, M, N, device=dev, dtype=torch.float32)
        for s in splits_list:
            if ops.gemm_splits(N, K, 1, s) == 1:
                continue
            t = timeit(lambda: [ops.skinny_linear_partial(x, W, s, part) for W in Wt]) / nw
            line += f" s{ops.gemm_splits(N, K, 1, s)}: {t*1e3:6.1f} us ({N*K*2/t/1e6:5.0f} GB/s)"
        print(line, flush=True)
        del part
        Q = [ops.tile_weight_u8(torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8)) for _ in range(nw)]
        mx, rx = torch.randn(N, device=dev).half() * 0.01, torch.rand(N, device=dev).half() / 16
        my, ry = torch.randn(K, device=dev).half() * 0.01, torch.rand(K, device=dev).half() / 16
        line = f"{'  mm8t (u8)':12s} {'':35s} | prep + GEMM + reduce:        "
        for s in splits_list:
            t = timeit(lambda: [ops.mm8t_linear(x, q, mx, rx, my, ry, act=act, splits=s, tiled=True) for q in Q]) / nw
            line += f" s{ops.gemm_splits(N, K, 1, s)}: {t*1e3:6.1f} us ({N*K/t/1e6:5.0f} GB/s)"
        print(line, flush=True)
        del Q
    del Ws, Wt
# the grouped launch of the time-mix block: R/K/V + the four LoRA down-projections (+ reduce with activations)
ranks = (96, 128, 128, 480)
nw = 6
Ws = [([ops.tile_weight((torch.randn(C, C, device=dev) / C ** 0.5).half()) for _ in range(3)],
       [(torch.randn(512, C, device=dev) / C ** 0.5).half() for _ in range(4)]) for _ in range(nw)]
mixed = torch.randn(6, M, C, device=dev).half()
rkv, hid = torch.empty(3, M, C, device=dev, dtype=torch.float16), torch.empty(4, M, 512, device=dev, dtype=torch.float16)


def group(W, s):
    probs = [(mixed[j], W[0][j], rkv[j], None, None) for j in range(3)]
    for j in range(4):
        kj = (ranks[j] + 63) // 64 * 64
        probs.append((mixed[2 + j], W[1][j][:kj], hid[j, :, :kj], None, "tanh" if j == 1 else ("sigmoid" if j == 3 else None)))
    ops.skinny_group(probs, splits=s)


line = "rkv+lora dn  3xCxC + 4 ranks            | GEMM + reduce:"
for s in (0, 1, 2, 4, 8):
    t = timeit(lambda: [group(W, s) for W in Ws]) / nw
    line += f" s{s}: {t*1e3:6.1f} us"
print(line, flush=True)
# the batched LoRA up-projections (per-problem K)
lora2 = [(torch.randn(4, C, 512, device=dev) / 512 ** 0.5).half() for _ in range(nw)]
lb = torch.randn(4, 1, C, device=dev).half()
ks = [(r + 63) // 64 * 64 for r in ranks]
t = timeit(lambda: [ops.skinny_bmm(hid, W2, lb, splits=1, k_of=ks) for W2 in lora2]) / nw
print(f"lora up x4   N=  4096 K<=512 (batched, bias in the epilogue): {t*1e3:6.1f} us", flush=True)
