import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from chirrup_amd.rwkv7 import RWKV_x070
from oracle import rwkv7_np as M
d = np.load("tests/golden/model_L2_C128.npz")
zd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
z_np = M.prepare_weights({k[2:]: d[k] for k in d.files if k.startswith("w:")})
args = lambda: types.SimpleNamespace(vocab_size=320, head_size=64, MODEL_NAME="unused")
ms = RWKV_x070(args(), state_dict=zd, device="cuda:0", sparse_bsz1=True)
md = RWKV_x070(args(), state_dict=zd, device="cuda:0")
mu = RWKV_x070(args(), state_dict=zd, device="cuda:0", fused=False)
for row in (0, 1):
    st = {n: m.generate_zero_state(0) for n, m in (("s", ms), ("d", md), ("u", mu))}
    st_np = [np.zeros((2, 2, 1, 128), np.float16), np.zeros((2, 1, 2, 64, 64), np.float16), np.zeros((1,), np.int32)]
    p = d["greedy:prompt"][row].tolist()
    lg = {"s": ms.forward(p, st["s"]), "d": md.forward(p, st["d"]), "u": mu.forward(p, st["u"])}
    lg_np = M.forward_seq_batch(z_np, [p], st_np, 2)[0]
    for step in range(16):
        gold = d["greedy:step_logits"][row, step].astype(np.float32)
        errs = {n: float(np.abs(l.float().cpu().numpy() - gold).max()) for n, l in lg.items()}
        e_np = float(np.abs(lg_np.astype(np.float32) - gold).max())
        top2 = np.sort(gold)[-2:]
        tok = int(d["greedy:ids"][row, step])
        print(row, step, "margin %.3f" % (top2[1]-top2[0]), "err vs golden: sparse %.4f dense %.4f unfused %.4f numpy1 %.4f" % (errs["s"], errs["d"], errs["u"], e_np),
              "argmax", {n: int(l.float().argmax()) for n, l in lg.items()}, "gold", tok)
        lg = {"s": ms.forward([tok], st["s"]), "d": md.forward([tok], st["d"]), "u": mu.forward([tok], st["u"])}
        lg_np = M.forward_seq_batch(z_np, [[tok]], st_np, 2)[0]
