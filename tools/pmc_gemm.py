"""Run ONE ring-GEMM shape of the 7.2B / bsz-200 decode step a few times (eager, rotating weights) so that
`rocprofv3 --pmc ...` can attribute counters to it.  The calls are the model's own (tile-image weights, same splits).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/pmc_gemm.py ffn.value
shapes: ffn.key | ffn.value | att.output | tmix_chain (R/K/V + the LoRA chain, one launch) | rkv_lora (the grouped launch it
        replaced) | lora_up | head | ffn.key.u8 | ffn.value.u8
(ffn.key, att.output and lora_up as shipped: two workgroups per tile over the two halves of the rows)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd import ops

shape = sys.argv[1] if len(sys.argv) > 1 else "ffn.value"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 200
C, dev, NW = 4096, "cuda:0", 6
torch.manual_seed(0)
rnd = lambda n, k: (torch.randn(n, k, device=dev) / k ** 0.5).half()
if shape == "ffn.value":
    Ws = [ops.tile_weight(rnd(C, 4 * C)) for _ in range(NW)]
    x, part = torch.randn(M, 4 * C, device=dev).half(), torch.empty(8, M, C, device=dev, dtype=torch.float32)
    run = lambda W: ops.skinny_linear_partial(x, W, 8, part)
elif shape == "ffn.key":
    Ws = [ops.tile_weight(rnd(4 * C, C)) for _ in range(NW)]
    x = torch.randn(M, C, device=dev).half()
    run = lambda W: ops.skinny_linear(x, W, act=1, splits=0, row_halves=True)          # unsplit, 128 tiles x 2 row halves
elif shape == "att.output":
    Ws = [ops.tile_weight(rnd(C, C)) for _ in range(NW * 2)]
    x, part = torch.randn(M, C, device=dev).half(), torch.empty(16, M, C, device=dev, dtype=torch.float32)   # (room for the split count
    run = lambda W: ops.skinny_linear_partial(x, W, 0, part, row_halves=True)           #  the size check assumes: whole rows) split 4 x 32 tiles x 2 row halves
elif shape == "head":
    Ws = [ops.tile_weight(rnd(65536, C)) for _ in range(2)]
    x = torch.randn(M, C, device=dev).half()
    run = lambda W: ops.skinny_linear(x, W, splits=1)
elif shape == "rkv_lora":
    ranks = (96, 128, 128, 480)
    Ws = [([ops.tile_weight(rnd(C, C)) for _ in range(3)], [rnd(512, C) for _ in range(4)]) for _ in range(NW)]
    mixed = torch.randn(6, M, C, device=dev).half()
    rkv, hid = torch.empty(3, M, C, device=dev, dtype=torch.float16), torch.empty(4, M, 512, device=dev, dtype=torch.float16)

    def run(W):
        probs = [(mixed[j], W[0][j], rkv[j], None, None) for j in range(3)]
        for j in range(4):
            kj = (ranks[j] + 63) // 64 * 64
            probs.append((mixed[2 + j], W[1][j][:kj], hid[j, :, :kj], None, "tanh" if j == 1 else None))
        ops.skinny_group(probs, splits=2)
elif shape == "tmix_chain":
    # the shipped time-mix launch: R/K/V tiles + the whole LoRA chain (rwkv7_tmix_gemms), row halves
    ranks = (128, 128, 128, 512)
    Ws = []
    for _ in range(NW):
        l1 = torch.zeros(4, 512, C, device=dev, dtype=torch.float16)
        l2 = torch.zeros(4, C, 512, device=dev, dtype=torch.float16)
        for j, r in enumerate(ranks):
            l1[j, :r], l2[j, :, :r] = rnd(r, C), rnd(C, r)
        Ws.append(([ops.tile_weight(rnd(C, C)) for _ in range(3)], l1, ops.tile_weight_batch(l2)))
    mixed = torch.randn(6, M, C, device=dev).half()
    rkv, hid = torch.empty(3, M, C, device=dev, dtype=torch.float16), torch.empty(4, M, 512, device=dev, dtype=torch.float16)
    up, lb = torch.empty(4, M, C, device=dev, dtype=torch.float16), torch.randn(4, 1, C, device=dev).half()

    def run(W):
        main_p = [(mixed[j], W[0][j], rkv[j]) for j in range(3)]
        lora_p = [(mixed[2 + j], W[1][j, :ranks[j]], j, lb[j].view(-1), up[j], "tanh" if j == 1 else ("sigmoid" if j == 3 else None), ranks[j])
                  for j in range(4)]
        ops.tmix_gemms(main_p, lora_p, W[2], hid, row_halves=True)
elif shape == "lora_up":
    ranks = [128, 128, 128, 512]
    Ws = [ops.tile_weight_batch((torch.randn(4, C, 512, device=dev) / 512 ** 0.5).half()) for _ in range(NW * 2)]
    hid, lb = torch.randn(4, M, 512, device=dev).half(), torch.randn(4, 1, C, device=dev).half()
    run = lambda W: ops.skinny_bmm(hid, W, lb, splits=1, k_of=ranks, row_halves=True)
elif shape in ("ffn.key.u8", "ffn.value.u8"):
    n, k = (4 * C, C) if shape == "ffn.key.u8" else (C, 4 * C)
    Ws = [ops.tile_weight_u8(torch.randint(0, 256, (n, k), device=dev, dtype=torch.uint8)) for _ in range(NW)]
    x = torch.randn(M, k, device=dev).half()
    mx, rx = torch.randn(n, device=dev).half() * 0.01, torch.rand(n, device=dev).half() / 16
    my, ry = torch.randn(k, device=dev).half() * 0.01, torch.rand(k, device=dev).half() / 16
    run = lambda W: ops.mm8t_linear(x, W, mx, rx, my, ry, act=int(shape == "ffn.key.u8"), tiled=True)
else:
    raise SystemExit("unknown shape " + shape)
for _ in range(3):
    for W in Ws:
        run(W)
torch.cuda.synchronize()
