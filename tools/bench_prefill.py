"""Chunked-prefill forward (T tokens for B sequences through forward_slots) on a synthetic model."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd.rwkv7 import RWKV_x070, model_args
from chirrup_amd.synth import CONFIGS, make_state_dict
name = sys.argv[1] if len(sys.argv) > 1 else "7.2B"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 25
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
mm8 = len(sys.argv) > 4 and sys.argv[4] == "mm8"        # uint8 (w8a16) ffn.key / ffn.value, the recommended decode configuration
L, C = CONFIGS[name]
dev = torch.device("cuda", 0)
zd = make_state_dict(L, C, 65536, seed=42, device=dev)
model = RWKV_x070(model_args("synthetic"), state_dict=zd, device=dev, ffn_dtype=torch.int8 if mm8 else torch.float16)
del zd
pool = model.generate_zero_state(B + 8)
idx = torch.arange(B, dtype=torch.int32, device=dev)
tok = torch.randint(1, 65536, (B, T), device=dev)
for _ in range(2):
    model.forward_slots(tok, pool, idx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n):
    model.forward_slots(tok, pool, idx)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"prefill {name}{' mm8 (u8 ffn)' if mm8 else ' fp16'}: B={B} T={T}: {ms:.2f} ms per chunk -> {B*T/ms*1e3:.0f} prompt tokens/s")
