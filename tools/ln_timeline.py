"""Phases of the add_ln_mix launches of a decode step from in-kernel 100-MHz stamps (include/chirrup_amd.h: rwkv7_ln_probe): us since
the launch's first workgroup entered, median / max over workgroups, for the LAST LN1 / LN2 launch of an eager step.
    python tools/ln_timeline.py [model=7.2B] [bsz=200]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import lib, ops
from chirrup_amd.rwkv7 import RWKV_x070, model_args
from chirrup_amd.synth import CONFIGS, make_state_dict

name = sys.argv[1] if len(sys.argv) > 1 else "7.2B"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L, C = CONFIGS[name]
L = min(L, 4)                                          # a few layers are enough
dev = torch.device("cuda", 0)
model = RWKV_x070(model_args("synthetic"), state_dict=make_state_dict(L, C, 65536, seed=1, device=dev), device=dev)
state = model.generate_zero_state(B)
tok = torch.randint(1, 65536, (B, 1), device=dev)
for _ in range(3):
    model.forward_seq_batch(tok, state)
torch.cuda.synchronize()
L_ = lib.load()
buf = torch.zeros(8 * 2048, dtype=torch.int64, device=dev)
names = ("entry", "row data arrived", "mean", "variance", "normalised", "stores issued", "stores acknowledged")

# which launch writes last: run the step with the probe on only around one kind of LN at a time by patching ops.add_ln_mix
orig = ops.add_ln_mix
for kind, n_mix in (("LN1 (six lerps, reads ffn.value's 8 partial planes)", 6), ("LN2 (one lerp, reads att.output's partial planes)", 1)):
    def wrapped(*a, **k):
        mixw = a[11]                                   # the lerp coefficients: [6, C] or [1, C]
        on = mixw is not None and mixw.shape[0] == n_mix
        if on:
            buf.zero_()
            L_.rwkv7_ln_probe(buf.data_ptr())
        try:
            return orig(*a, **k)
        finally:
            if on:
                L_.rwkv7_ln_probe(None)
    ops.add_ln_mix = wrapped
    model.forward_seq_batch(tok, state)
    torch.cuda.synchronize()
    ops.add_ln_mix = orig
    v = buf.view(-1, 8)
    v = v[v[:, 0] > 0].double()
    t0 = v[:, 0].min()
    v = (v - t0) / 100.0
    print(f"{name} bsz {B}: {kind}: {len(v)} workgroups; us since the first workgroup's entry (median / max)")
    for i, nm in enumerate(names):
        col = v[:, i]
        print(f"  {nm:22s} {float(col.median()):6.2f} {float(col.max()):6.2f}")
