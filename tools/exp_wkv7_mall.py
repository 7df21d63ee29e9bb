"""tools/exp_wkv7_mall.py: does a WKV7 launch run faster when its state was just read (i.e. sits in the 256-MB memory-side
cache)?  The bound of an in-step prefetch of layer l's state (105 MB at 7.2B bsz 200) under the GEMMs in front of it.
A: WKV7 launches over 32 rotating layer states, cold.  B: a reading pass over the layer's state (torch sum) right before each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd import ops

B, C, L = 200, 4096, 32
H = C // 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
state = (torch.randn(L, B, H, 64, 64, device=dev) * 0.1).half()
mk = lambda s: (torch.randn(B, 1, C, device=dev) * s).half()
r, k, v, a, b = mk(1), mk(1), mk(1), mk(.125), mk(.06)
w = (torch.rand(B, 1, C, device=dev) * 12 - 8).half()
y = torch.empty(B, 1, C, device=dev, dtype=torch.float16)
et = torch.arange(B, device=dev, dtype=torch.int32) * 7 + 3
sink = torch.zeros(L, device=dev)


def run(prefetch, part=1.0, other=0):
    ts = []
    n = int(B * part)
    for rep in range(3):
        for l in range(L):
            if prefetch:
                sink[l] = state[(l + other) % L][:n].float().sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.forward_seq(B, 1, C, H, state[l], r, w, k, v, a, b, y, et)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


print("cold              : %.1f us per launch (events around one launch)" % run(False))
print("state just read   : %.1f us" % run(True))
print("half of it read   : %.1f us" % run(True, 0.5))
print("ANOTHER layer's state just read (control: same queue state, nothing of this layer cached): %.1f us" % run(True, 1.0, 16))
print("cold again        : %.1f us" % run(False))
