"""tools/sampler_bench.py [rows] [V]: time of one rwkv7_sample_topp launch (HIP events over 50 launches) on logits shaped like a
language model's (a few dominant tokens over a wide body), for the sampling settings the serving loop meets:
default (T 1.0, top_p 0.3), hot (T 1.5), with top-k, and near-greedy.  Also checks each draw against the oracle's kept set."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd import ops

rows_n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
V = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
body = torch.randn(rows_n, V, generator=g) * 2.0
peaks = torch.randint(0, V, (rows_n, 12), generator=g)
body.scatter_(1, peaks, torch.rand(rows_n, 12, generator=g) * 6 + 6)
logits = body.to(torch.float16).to(dev)
rows = torch.arange(rows_n, dtype=torch.int32, device=dev)
ids = torch.zeros(rows_n, dtype=torch.int32, device=dev)
uni = torch.rand(rows_n, generator=g).to(dev)


def run(T, P, K, label):
    t = torch.full((rows_n,), T, dtype=torch.float16, device=dev)
    p = torch.full((rows_n,), P, dtype=torch.float16, device=dev)
    k = torch.full((rows_n,), K, dtype=torch.int32, device=dev)
    for _ in range(5):
        ops.sample_topp(logits, rows, t, p, k, uni, ids)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        ops.sample_topp(logits, rows, t, p, k, uni, ids)
    b.record()
    torch.cuda.synchronize()
    if os.environ.get("CHIRRUP_AMD_LIB", "").endswith("s9.so"):          # the stamp build: ids hold interval times, 10 ns ticks
        names = ("load+max", "masses+Z", "pass compares", "pass reductions", "top-k", "kept mass", "scan+draw", "all")
        t = ids.float().view(-1, 8).mean(0) / 100.0
        print("%-28s %7.1f us per launch   " % (label, a.elapsed_time(b) * 1e3 / 50) + "  ".join("%s %.1f" % (n, v) for n, v in zip(names, t.tolist())))
        return
    # every id must lie in the kept set: softmax mass of tokens >= the drawn one's logit, minus the drawn logit's ties, < top_p
    prob = torch.softmax(logits.float(), dim=1)
    drawn = logits.gather(1, ids.long().unsqueeze(1))
    above = (prob * (logits > drawn)).sum(1)
    ok_p = bool((above < float(p[0]) + 1e-3).all()) or P == 0
    ok_k = K <= 0 or bool(((logits > drawn).sum(1) < K).all())
    print("%-28s %7.1f us per launch   kept-set check: %s" % (label, a.elapsed_time(b) * 1e3 / 50, "ok" if ok_p and ok_k else "FAILED"))


run(1.0, 0.3, 0, "default (T 1.0, top_p 0.3)")
run(1.5, 0.9, 0, "hot (T 1.5, top_p 0.9)")
run(1.0, 0.9, 50, "top-k 50 (top_p 0.9)")
run(1.0, 0.0, 0, "top_p 0 (greedy corner)")
