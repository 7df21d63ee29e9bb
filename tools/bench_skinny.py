"""skinny_linear / mm8t vs torch (hipBLASLt) on the decode GEMM shapes, graph-replayed, rotating weights."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from chirrup_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if len(sys.argv) > 2:
    from chirrup_amd import lib
    lib.load().skinny_gemm_select(int(sys.argv[2]))      # kernel variant, see include/chirrup_amd.h
    print("kernel mode", sys.argv[2], flush=True)
dev = "cuda:0"
torch.manual_seed(0)
def timeit(fn, iters=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, N, K, splits_list in [("att CxC", 4096, 4096, (1, 2, 4, 8)), ("ffn.key", 16384, 4096, (1, 2)), ("ffn.value", 4096, 16384, (2, 4, 8, 16)),
                                ("lora dn 480", 480, 4096, (8, 16, 32)), ("lora up 480", 4096, 512, (1,)), ("head", 65536, 4096, (1,))]:
    nw = 12 if N * K * 2 < 200e6 else 3
    Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(nw)]
    x = torch.randn(M, K, device=dev).half()
    t0 = timeit(lambda: [F.linear(x, W) for W in Ws]) / nw
    line = f"{name:12s} N={N:6d} K={K:6d}: torch {t0*1e3:7.1f} us |"
    for s in splits_list:
        t = timeit(lambda: [ops.skinny_linear(x, W, splits=s) for W in Ws]) / nw
        line += f" s{s}: {t*1e3:6.1f} us ({N*K*2/t/1e6:5.0f} GB/s)"
    print(line, flush=True)
    if N * K <= 16384 * 4096:
        Q = [torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8) for _ in range(nw)]
        mx = torch.randn(N, device=dev).half() * 0.01; rx = torch.rand(N, device=dev).half() / 16
        my = torch.randn(K, device=dev).half() * 0.01; ry = torch.rand(K, device=dev).half() / 16
        line = f"{'  mm8t':12s} {'':25s}                |"
        for s in splits_list:
            t = timeit(lambda: [ops.mm8t_linear(x, q, mx, rx, my, ry, splits=s) for q in Q]) / nw
            line += f" s{s}: {t*1e3:6.1f} us ({N*K/t/1e6:5.0f} GB/s)"
        print(line, flush=True)
        del Q
    del Ws
# batched launches of the time-mix block (7.2B: C 4096, LoRA ranks padded to 512)
for name, Z, N, K, splits_list, act in [("rkv 3xCxC", 3, 4096, 4096, (1, 2, 4), 0), ("lora dn x4", 4, 512, 4096, (4, 8, 16), 4),
                                        ("lora up x4", 4, 4096, 512, (1, 2), 0)]:
    nw = 8
    Ws = [(torch.randn(Z, N, K, device=dev) / K ** 0.5).half() for _ in range(nw)]
    x = torch.randn(Z, M, K, device=dev).half()
    bias = torch.randn(Z, 1, N, device=dev).half()
    if name.startswith("lora up"):
        t0 = timeit(lambda: [torch.baddbmm(bias, x, W.transpose(1, 2)) for W in Ws]) / nw
    else:
        t0 = timeit(lambda: [torch.bmm(x, W.transpose(1, 2)) for W in Ws]) / nw
    line = f"{name:12s} N={N:6d} K={K:6d}: torch {t0*1e3:7.1f} us |"
    for s in splits_list:
        b = bias if name.startswith("lora up") else None
        t = timeit(lambda: [ops.skinny_bmm(x, W, b, act=act, splits=s) for W in Ws]) / nw
        line += f" s{s}: {t*1e3:6.1f} us ({Z*N*K*2/t/1e6:5.0f} GB/s)"
    print(line, flush=True)
    del Ws
