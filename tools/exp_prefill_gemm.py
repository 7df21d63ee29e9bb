"""Library-GEMM formulations of the prefill projections (rows x K) . (N x K)^T, graph-free event timing over rotating weights:
python tools/exp_prefill_gemm.py [rows=2500]"""
import sys
import torch
import torch.nn.functional as F

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K in (("ffn.key", 16384, 4096), ("ffn.value", 4096, 16384), ("att.output", 4096, 4096)):
    Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(6)]
    x = torch.randn(rows, K, device=dev).half()
    i = [0]

    def nxt():
        i[0] = (i[0] + 1) % len(Ws)
        return Ws[i[0]]

    fl = 2 * rows * N * K
    res = {}
    res["F.linear"] = timeit(lambda: F.linear(x, nxt()))
    res["split N/2"] = timeit(lambda: [F.linear(x, w_) for w_ in nxt().split(N // 2)])
    res["split N/4"] = timeit(lambda: [F.linear(x, w_) for w_ in nxt().split(N // 4)])
    res["split rows/2"] = timeit(lambda: [F.linear(x_, W) for W in [nxt()] for x_ in x.split((rows + 1) // 2)])
    out = torch.empty(rows, N, device=dev, dtype=torch.float16)
    res["mm(out=)"] = timeit(lambda: torch.mm(x, nxt().t(), out=out))
    Wts = [w_.t().contiguous() for w_ in Ws]          # [K, N] row-major ("NN")
    j = [0]

    def nxt_t():
        j[0] = (j[0] + 1) % len(Wts)
        return Wts[j[0]]

    res["x @ W^T stored [K,N]"] = timeit(lambda: x @ nxt_t())
    try:
        torch.backends.cuda.preferred_blas_library("cublas")
        res["rocBLAS F.linear"] = timeit(lambda: F.linear(x, nxt()))
        res["rocBLAS x @ [K,N]"] = timeit(lambda: x @ nxt_t())
    finally:
        torch.backends.cuda.preferred_blas_library("cublaslt")
    print(name, f"rows={rows} N={N} K={K}:", "  ".join(f"{k_} {v:.0f} us ({fl / v / 1e9:.2f} PF/s)" for k_, v in res.items()), flush=True)
    del Ws, Wts
