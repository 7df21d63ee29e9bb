"""Phase timings of the time-mix launch (chain_gemm_kernel) at the 7.2B / bsz-200 shape: in-kernel 100-MHz stamps per workgroup
(include/chirrup_amd.h: skinny_gemm_clock_probe).  usage: python tools/chain_stamps.py [M] [C]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import lib, ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 200
HALVES = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ranks = {4096: [128, 128, 128, 512], 2048: [128, 128, 64, 256], 768: [64, 64, 64, 128]}[C]
dev = "cuda"
torch.manual_seed(0)
K, dmax = C, max(ranks)
NW = 6
mixed = torch.randn(6, M, K, device=dev).half()
sets = []
for _ in range(NW):
    rkv_w = [ops.tile_weight((torch.randn(C, K, device=dev) / K ** 0.5).half()) for _ in range(3)]
    lora1 = torch.zeros(4, dmax, K, device=dev, dtype=torch.float16)
    lora2 = torch.zeros(4, C, dmax, device=dev, dtype=torch.float16)
    for j, r in enumerate(ranks):
        lora1[j, :r] = (torch.randn(r, K, device=dev) / K ** 0.5).half()
        lora2[j, :, :r] = (torch.randn(C, r, device=dev) / r ** 0.5).half()
    sets.append((rkv_w, lora1, ops.tile_weight_batch(lora2)))
lbias = torch.randn(4, 1, C, device=dev).half()
rkv = torch.empty(3, M, C, device=dev, dtype=torch.float16)
hid = torch.empty(4, M, dmax, device=dev, dtype=torch.float16)
up = torch.empty(4, M, C, device=dev, dtype=torch.float16)
acts = [None, "tanh", None, "sigmoid"]


def run(s):
    rkv_w, lora1, lora2_t = s
    main_p = [(mixed[j], rkv_w[j], rkv[j]) for j in range(3)]
    lora_p = [(mixed[2 + j], lora1[j, :ranks[j]], j, lbias[j].view(-1), up[j], acts[j], ranks[j]) for j in range(4)]
    ops.tmix_gemms(main_p, lora_p, lora2_t, hid, row_halves=HALVES)


for _ in range(3):
    for s in sets:
        run(s)
torch.cuda.synchronize()
L_ = lib.load()
pairs = 4096
buf = torch.zeros(2 * pairs, dtype=torch.int64, device=dev)
L_.skinny_gemm_clock_probe(buf.data_ptr(), pairs)
for s in sets:
    run(s)
torch.cuda.synchronize()
L_.skinny_gemm_clock_probe(None, 0)
st = buf.view(-1, 8).cpu()
live = st[:, 0] > 0
st = st[live].double()
t0 = st[:, 0].min()
n_chain = int(((st[:, 2] > 0) | (st[:, 6] > 0)).sum())
chain, rest = st[:n_chain], st[n_chain:]
us = lambda v: (v - t0) / 100.0
names = ["start", "down loop done", "slab + ticket", "combine + signal", "first acquire", "first up item", "up share done", "end"]
print(f"M={M} C={C}: {n_chain} chain workgroups, {len(rest)} R/K/V workgroups; us since the first workgroup's start (median / max over workgroups)")
for i, nm in enumerate(names):
    v = chain[:, i]
    v = v[v > 0]
    if len(v):
        print(f"  chain  {nm:18s} {float(us(v).median()):7.2f} {float(us(v).max()):7.2f}   (n={len(v)})")
for i, nm in ((0, "start"), (1, "main loop done"), (7, "end")):
    v = rest[:, i]
    v = v[v > 0]
    print(f"  R/K/V  {nm:18s} {float(us(v).median()):7.2f} {float(us(v).max()):7.2f}   (n={len(v)})")
print(f"  launch span {float(us(st.max())):.2f} us;  status {ops.chain_status()}")
