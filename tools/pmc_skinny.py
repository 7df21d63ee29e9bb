"""Run the ffn.value-shaped ring GEMM a few times (no graph) so that rocprofv3 --pmc can attribute counters to it.
usage (on the GPU box):  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... --output-format csv -d out -- python3 tools/pmc_skinny.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd import ops
M, N, K, s = 200, 4096, 16384, 8
dev = "cuda:0"
torch.manual_seed(0)
Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(6)]
x = torch.randn(M, K, device=dev).half()
part = torch.empty(s, M, N, device=dev, dtype=torch.float32)
for _ in range(2):
    for W in Ws:
        ops.skinny_linear_partial(x, W, s, part)
torch.cuda.synchronize()
