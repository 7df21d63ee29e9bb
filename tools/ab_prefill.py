"""Chunked prefill (forward_slots, B sequences x T tokens) on a synthetic model, with and without the row-parallel time-mix split:
python tools/ab_prefill.py [model=7.2B] [B=25] [T=100]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd.rwkv7 import RWKV_x070, model_args
from chirrup_amd.synth import CONFIGS, make_state_dict

name = sys.argv[1] if len(sys.argv) > 1 else "7.2B"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
L, C = CONFIGS[name]
dev = torch.device("cuda", 0)
zd = make_state_dict(L, C, 65536, seed=42, device=dev)
model = RWKV_x070(model_args("synthetic"), state_dict=zd, device=dev)
del zd
pool = model.generate_zero_state(B + 8)
idx = torch.arange(B, dtype=torch.int32, device=dev)
tok = torch.randint(1, 65536, (B, T), device=dev)


def run(label, n=4):
    for _ in range(2):
        model.forward_slots(tok, pool, idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        model.forward_slots(tok, pool, idx)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"prefill {name} B={B} T={T} [{label}]: {ms:.2f} ms per chunk -> {B * T / ms * 1e3:.0f} prompt tokens/s", flush=True)


for rep in range(2):
    model.split_tmix_min_T = 0
    run("one fused time-mix kernel")
    model.split_tmix_min_T = 32
    run("row-parallel gating / decay / group norm around the scan")
