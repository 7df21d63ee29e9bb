"""Prefix-cache hit cost: HBM arena (device-to-device) vs the reference's host-resident states (two PCIe copies).
usage: python tools/bench_state_cache.py [model=13.3B] [entries=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd.state_cache import HbmStateArena, SimpleStateCache
from chirrup_amd.synth import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "13.3B"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L, C = CONFIGS[name]
H = C // 64
dev = torch.device("cuda", 0)
pool = [torch.zeros((L, 2, 8, C), dtype=torch.float16, device=dev), torch.zeros((L, 8, H, 64, 64), dtype=torch.float16, device=dev),
        torch.zeros((8,), dtype=torch.int32, device=dev)]
export = lambda slot: [pool[0][:, :, [slot], :], pool[1][:, [slot], :, :], pool[2][[slot]]]      # Worker._export_state


def install(state, slot):                                                                           # Worker._install
    pool[0][:, :, [slot], :] = state[0].to(dev, non_blocking=True)
    pool[1][:, [slot], :, :] = state[1].to(dev, non_blocking=True)
    pool[2][[slot]] = state[2].to(dev, non_blocking=True)


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


arena = HbmStateArena(L, C, n, dev)
cache = SimpleStateCache(max_size=n, arena=arena)
mb = arena.bytes_per_state / 1e6
print(f"{name}: state {mb:.1f} MB, arena of {n} rows = {n * mb / 1e3:.2f} GB in HBM", flush=True)
t_put = timed(lambda i: cache.cache((i + 1, 7, 7), export(i % 8)), n)
t_hit = timed(lambda i: install(cache.check([i % n + 1, 7, 7, 0])[1], i % 8), 2 * n)
host = [[t.cpu() for t in export(i % 8)] for i in range(8)]
t_host_put = timed(lambda i: [t.to("cpu") for t in export(i % 8)], 16)
t_host_hit = timed(lambda i: install(host[i % 8], i % 8), 16)
print(f"  cache a prefix : arena {t_put:.3f} ms   host copy (reference, worker.py:427-429) {t_host_put:.3f} ms")
print(f"  hit -> slot    : arena {t_hit:.3f} ms   host copy (reference, worker.py:591-597) {t_host_hit:.3f} ms")
