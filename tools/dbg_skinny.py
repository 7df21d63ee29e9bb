import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd import ops
torch.manual_seed(0)
for (M, N, K, s) in [(32, 128, 64, 1), (32, 128, 128, 1), (32, 128, 256, 1), (200, 256, 512, 1), (32,128,256,2)]:
    x = torch.randn(M, K, device="cuda").half()
    w = torch.randn(N, K, device="cuda").half() / K ** 0.5
    w = w.half()
    y = ops.skinny_linear(x, w, splits=s)
    ref = (x.double() @ w.double().t())
    err = (y.double() - ref).abs()
    print(M, N, K, s, "max err", float(err.max()), "frac bad", float((err > 0.01).float().mean()), "nan", bool(torch.isnan(y).any()))
    if float(err.max()) > 0.01:
        # which k-blocks contribute? use block-indicator inputs
        for kb in range(K // 64):
            xx = torch.zeros(M, K, device="cuda").half(); xx[:, kb*64:(kb+1)*64] = 1
            ww = torch.ones(N, K, device="cuda").half()
            yy = ops.skinny_linear(xx, ww, splits=s)
            print("   kblock", kb, "sum ->", yy[0, 0].item(), yy[M-1, N-1].item(), "(want 64)")
