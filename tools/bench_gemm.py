"""Micro-benchmark of the decode-step GEMM shapes through torch (hipBLASLt / rocBLAS), graph-replayed
so host launch cost is excluded.  usage: python tools/bench_gemm.py [M]"""
import os, sys, torch
import torch.nn.functional as F

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
torch.manual_seed(0)
shapes = [("att CxC", 4096, 4096), ("ffn.key", 16384, 4096), ("ffn.value", 4096, 16384), ("lora down 128", 128, 4096),
          ("lora down 480", 480, 4096), ("lora up 128", 4096, 128), ("lora up 480", 4096, 480), ("head", 65536, 4096)]
NW = 24   # rotate over NW weight copies (> 256 MiB in total for the big ones) so L2/MALL cannot serve them


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


print("tunableop:", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"), "prefer hipblaslt:", os.environ.get("TORCH_BLAS_PREFER_HIPBLASLT"))
for name, N, K in shapes:
    nw = NW if N * K * 2 < 200e6 else 3
    Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(nw)]
    x = torch.randn(M, K, device=dev).half()
    xT = x.t().contiguous()

    def f_linear():
        for W in Ws:
            F.linear(x, W)

    def f_swapped():
        for W in Ws:
            torch.mm(W, xT)          # y^T [N, M]

    t1 = timeit(f_linear) / nw
    t2 = timeit(f_swapped) / nw
    byts = N * K * 2
    print(f"{name:14s} N={N:6d} K={K:6d}: F.linear {t1*1e3:8.1f} us ({byts/t1/1e6:7.0f} GB/s)   W@x^T {t2*1e3:8.1f} us ({byts/t2/1e6:7.0f} GB/s)")
    del Ws
# batched R,K,V
Wb = [(torch.randn(3, 4096, 4096, device=dev) / 64).half() for _ in range(8)]
xb = torch.randn(3, M, 4096, device=dev).half()
def f_bmm():
    for W in Wb:
        torch.bmm(xb, W.transpose(1, 2))
t = timeit(f_bmm) / 8
print(f"bmm 3x(CxC): {t*1e3:.1f} us per 3 GEMMs ({3*4096*4096*2/t/1e6:.0f} GB/s)")
