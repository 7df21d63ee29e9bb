"""tools/exp_two_half_batches.py [model] [bsz]: would TWO half-batch decode steps on two streams run concurrently faster than one
full-batch step?  The step is a chain of launches that are each bound by something else (GEMM main loops: one CU's ingest, at
~3 TB/s of HBM reads; WKV7: HBM; LN: latency) -- two half batches a launch apart could fill each other's idle resource, at the
price of streaming every weight twice.  The time-mix launch keeps library-global hand-off words, so both variants run with the
two-launch time-mix form (chain_tmix_gemms = False); the one-graph baseline is reported with and without it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "7.2B"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = bench.build_model(name, dev, fused=True)


def time_graphs(graphs, streams, iters=40):
    toks = [torch.randint(1, 65536, (g.B, 1), device=dev) for g in graphs]
    def once():
        for g, s, t in zip(graphs, streams, toks):
            with torch.cuda.stream(s):
                g.step(t)
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


cur = torch.cuda.current_stream()
full = model.capture_decode_graph(bench.make_state(model, B))
print("one graph, %d rows, single-launch time-mix : %.3f ms per step" % (B, time_graphs([full], [cur])))
del full
model.chain_tmix_gemms = False
full = model.capture_decode_graph(bench.make_state(model, B))
print("one graph, %d rows, two-launch time-mix    : %.3f ms per step" % (B, time_graphs([full], [cur])))
del full
torch.cuda.empty_cache()
h = B // 2
ga = model.capture_decode_graph(bench.make_state(model, h, seed=1))
gb = model.capture_decode_graph(bench.make_state(model, h, seed=2))
print("one graph, %d rows                          : %.3f ms per step" % (h, time_graphs([ga], [cur])))
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
print("two graphs of %d rows on two streams        : %.3f ms per pair of steps" % (h, time_graphs([ga, gb], [s1, s2])))
print("the same two graphs on ONE stream            : %.3f ms per pair of steps" % time_graphs([ga, gb], [cur, cur]))
