"""The three forms of the rwkv_pip mm8_seq operator at the ffn.key shape (200, 4096, 16384), after >= 50 ms of GPU work, 50 launches
each (the speed claim that used to be a wall-clock assertion inside a parity test)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from chirrup_amd import ops                                  # noqa: E402
from chirrup_amd.quant import quantize_weight                # noqa: E402

B, N, M = 200, 4096, 16384
torch.manual_seed(1)
x = torch.randn(B, N, device="cuda").half()
w16 = (torch.randn(N, M, device="cuda") / N ** 0.5).half()
q, mx, rx, my, ry = quantize_weight(w16)
my, ry = my.reshape(-1).contiguous(), ry.reshape(-1).contiguous()
y = torch.empty((B, M), dtype=torch.float16, device="cuda")
args = (B, N, M, x, q.contiguous(), mx, rx, my, ry)
for name, fn, n in (("mm8_seq (two-pass exact split, the reference kernel's arithmetic)", ops.mm8_seq, 50),
                    ("mm8_seq_opt (one-pass split form)", ops.mm8_seq_opt, 50),
                    ("mm8_seq_stateless (C ABI mm8_seq: packs per call)", ops.mm8_seq_stateless, 50),
                    ("mm8_seq_direct (as-coded scalar kernel)", ops.mm8_seq_direct, 5)):
    for _ in range(300 if n == 50 else 3):
        fn(*args, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(*args, y)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:70s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us per call", flush=True)
