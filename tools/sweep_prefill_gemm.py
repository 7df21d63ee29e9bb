"""Which formulation of a prefill projection the library runs fastest, by row count: (rows x K) . (N x K)^T as ONE call, as two row
halves, as two column halves -- hipBLASLt's own choice of kernel per shape is erratic (1575 rows: ffn.value 391 us whole, 251 us as
two row halves; 2500 rows: 264 vs 318).  Prints one line per (shape, rows); RWKV_x070's static rule (rwkv7.py: _split_rows_rule)
is read off this table (profiles/r04_prefill_gemm_formulations.txt).
python tools/sweep_prefill_gemm.py [C=4096]"""
import sys

import torch

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, n=12):
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


for name, N, K in (("ffn.key", 4 * C, C), ("ffn.value", C, 4 * C), ("att.output / r / k / v", C, C)):
    Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(4)]
    i = [0]

    def nxt():
        i[0] = (i[0] + 1) % len(Ws)
        return Ws[i[0]]

    for rows in (300, 400, 512, 640, 768, 900, 1024, 1200, 1400, 1575, 1600, 1800, 2000, 2200, 2500):
        x = torch.randn(rows, K, device=dev).half()
        out = torch.empty(rows, N, device=dev, dtype=torch.float16)
        half = (rows + 1) // 2
        res = {"whole": timeit(lambda: torch.mm(x, nxt().t(), out=out)),
               "rows/2": timeit(lambda: [[torch.mm(x[a:b], W.t(), out=out[a:b]) for a, b in ((0, half), (half, rows))] for W in [nxt()]]),
               "rows/3": timeit(lambda: [[torch.mm(x[a:b], W.t(), out=out[a:b]) for a, b in ((0, rows // 3), (rows // 3, 2 * rows // 3), (2 * rows // 3, rows))] for W in [nxt()]])}
        best = min(res, key=res.get)
        fl = 2 * rows * N * K
        print(f"{name:24s} rows {rows:5d}: " + "  ".join(f"{k_} {v:6.0f} us" for k_, v in res.items()) + f"   best {best} ({fl / res[best] / 1e9:.2f} PF/s, {res['whole'] / res[best]:.2f}x whole)", flush=True)
    del Ws
