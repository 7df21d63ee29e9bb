"""BASELINE.json config 5 as a workload on ONE MI355X: RWKV7-g1 13.3B (L=61, C=4096; synthetic weights), one worker with
bsz 64, a prefix-state cache whose states live in an HBM arena of >= 2048 resident 33-MB rows, and a
request mix in which half of the requests hit a cached prefix (reference: chirrup/utils/state_cache.py:85-124, :286;
chirrup/worker.py:421-435, :591-597).

  phase A  32 long prompts, uncached: chunked prefill + greedy decode; the worker exports each prompt's prefix state
           (cache_prefill) straight into an arena row; the cache adopts the rows.
  phase B  64 requests at once: the 32 prompts again (cache hits: the state is installed row -> slot with one device copy,
           only the last few prompt tokens are fed) + 32 new prompts (misses).
  checks   every hit request generates exactly the ids of its uncached run; the arena holds >= 2048 resident rows.
  reports  decode iteration time of the 64-row mix, prefill + admission time of hits vs misses, hit-install cost.

usage: python tools/bench_config5.py [rows=2048] [prompt_len=384] [new_tokens=48] [out.txt]
"""
import os
import queue
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from chirrup_amd.core_structure import ModelLoadConfig, Task
from chirrup_amd.rwkv7 import RWKV_x070, model_args
from chirrup_amd.state_cache import ArenaRef, HbmStateArena, SimpleStateCache
from chirrup_amd.synth import CONFIGS, make_state_dict
from chirrup_amd.worker import Worker

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
plen = int(sys.argv[2]) if len(sys.argv) > 2 else 384
new = int(sys.argv[3]) if len(sys.argv) > 3 else 48
out_path = sys.argv[4] if len(sys.argv) > 4 else None
name, bsz, pad = os.environ.get("CONFIG5_MODEL", "13.3B"), 64, 3
L, C = CONFIGS[name]
dev = torch.device("cuda", 0)
lines = []


def say(msg):
    print(msg, flush=True)
    lines.append(msg)


t0 = time.perf_counter()
zd = make_state_dict(L, C, 65536, seed=42, device=dev)
model = RWKV_x070(model_args("synthetic"), state_dict=zd, device=dev)
del zd
torch.cuda.empty_cache()
arena = HbmStateArena.for_model(model, capacity=rows + bsz)          # cached rows + one per request in flight
cache = SimpleStateCache(max_size=rows, arena=arena)
say(f"# config 5 on one MI355X: RWKV7-g1 {name} (L={L}, C={C}), bsz {bsz}, prefix cache of {rows} rows x {arena.bytes_per_state / 1e6:.2f} MB "
    f"= {rows * arena.bytes_per_state / 1e9:.1f} GB resident in HBM (arena {arena.capacity} rows); model + arena ready in {time.perf_counter() - t0:.0f} s; "
    f"HBM in use {torch.cuda.memory_allocated() / 1e9:.1f} GB")


class Tok:
    def decode(self, ids, utf8_errors="strict"):
        return "x"


class Sink:
    def __init__(self):
        self.items, self.t_first = [], None

    def put_nowait(self, x):
        if x[0] == "token_generated" and self.t_first is None:
            self.t_first = time.perf_counter()
        self.items.append(x)


cfg = ModelLoadConfig(model_path="synthetic", vocab_path="none", vocab_size=65536, head_size=64)
tq, mq = queue.Queue(), queue.Queue()
w = Worker("w0", [0], cfg, tq, mq, None, batch_size=bsz + 1, model=model, tokenizer=Tok(), state_arena=arena)
w.max_prefill_count = bsz
w._init_worker()

# fill the cache to `rows - 32` resident prefixes (synthetic states: what matters is that the rows are live in HBM)
g = torch.Generator().manual_seed(7)
t0 = time.perf_counter()
filler = [torch.randn((L, 2, 1, C), device=dev).half() * 0.5, torch.randn((L, 1, C // 64, 64, 64), device=dev).half() * 0.1,
          torch.tensor([plen], dtype=torch.int32, device=dev)]
for i in range(rows - 32):
    cache.cache((65000, i // 60000 + 1, i % 60000 + 1), filler)
torch.cuda.synchronize()
say(f"filled {rows - 32} rows in {time.perf_counter() - t0:.2f} s ({(rows - 32) * arena.bytes_per_state / 1e9 / (time.perf_counter() - t0):.0f} GB/s of device copies)")

mk = lambda toks, state=None, **kw: Task(output_queue=Sink(), task_event_queue=queue.Queue(), prompt_str="", prefill_tokens=list(toks),
                                         state=state, temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0,
                                         stop_tokens=[], max_tokens=new, **kw)
ids_of = lambda t: [x[1][0] for x in t.output_queue.items if x[0] == "token_generated"]


def drain(tasks, label):
    """Run the worker until idle; returns (wall seconds, decode-iteration times once every request is decoding)."""
    for t in tasks:
        tq.put(t)
    t_start, it = time.perf_counter(), []
    while True:
        a = time.perf_counter()
        busy = w.step()
        if not busy:
            break
        cats = w._organize_batch()
        from chirrup_amd.worker import StateCategory
        if len(cats[StateCategory.FORWARD_ONE_DECODE]) == len(tasks) and not cats[StateCategory.FORWARD_SEQ] and not cats[StateCategory.FORWARD_ONE_PREFILL]:
            torch.cuda.synchronize()
            it.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_start
    it = sorted(it[2:]) if len(it) > 4 else it
    med = it[len(it) // 2] * 1e3 if it else float("nan")
    say(f"{label}: {len(tasks)} requests, {sum(len(ids_of(t)) for t in tasks)} tokens, wall {wall:.2f} s, "
        f"median decode iteration with all {len(tasks)} rows live {med:.2f} ms ({len(tasks) / med * 1e3 if it else 0:.0f} tok/s)")
    return wall, med


prompts = [torch.randint(1, 65536, (plen,), generator=g).tolist() for _ in range(64)]
# ---- phase A: the first 32 prompts, uncached, exporting their prefix states
tA = [mk(p, cache_prefill=True, cache_prefill_padding=pad) for p in prompts[:32]]
drain(tA, "phase A (32 uncached prompts, prefix export)")
for t in tA:
    ex = [x[1] for x in t.output_queue.items if x[0] == "cache_prefill"]
    assert len(ex) == 1 and isinstance(ex[0]["state"], ArenaRef), "the worker exports straight into the arena"
    cache.cache(tuple(ex[0]["prefilled_tokens"]), ex[0]["state"])
assert len(cache) == rows and arena.free_rows == bsz, (len(cache), arena.free_rows)
say(f"cache now holds {len(cache)} prefixes = {len(cache) * arena.bytes_per_state / 1e9:.1f} GB resident; free arena rows {arena.free_rows}")

# ---- phase B: 64 requests, 50 % prefix hits
hits, t_inst = [], 0.0
for p in prompts[:32]:
    rest, state, n = cache.check(list(p))
    assert n == plen - pad and isinstance(state, ArenaRef), (n, type(state))
    hits.append(mk(rest, state=state))
misses = [mk(p) for p in prompts[32:]]
mix = [t for pair in zip(hits, misses) for t in pair]
wallB, medB = drain(mix, "phase B (64 requests, 32 hits + 32 misses)")
bad = [i for i, (a, b) in enumerate(zip(hits, tA)) if ids_of(a) != ids_of(b)]
assert not bad, f"hit requests {bad} diverge from their uncached run"
assert all(len(ids_of(t)) == new for t in mix)
say(f"ids of all 32 hit requests equal their uncached run ({new} greedy tokens each); all pins released: free arena rows {arena.free_rows}")
t_hit = sorted(t.output_queue.t_first for t in hits)
t_miss = sorted(t.output_queue.t_first for t in misses)
# ---- the same 64 prompts with nothing cached, for the admission / prefill cost the hits avoid
tC = [mk(p) for p in prompts]
wallC, medC = drain(tC, "phase C (the same 64 prompts, all uncached)")
say(f"prefill work avoided by the hits: phase B wall {wallB:.2f} s vs phase C {wallC:.2f} s; decode iteration {medB:.2f} vs {medC:.2f} ms")
# ---- cost of a hit install (row -> slot) and of an export (slot -> row), one copy each
torch.cuda.synchronize()
ref = cache.check(list(prompts[0]))[1]
reps = 20
t0 = time.perf_counter()
for i in range(reps):
    ref.install_into(w.batch_state, i % bsz)
torch.cuda.synchronize()
t_in = (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter()
for i in range(reps):
    arena.export_slot(w.batch_state, i % bsz).release()
torch.cuda.synchronize()
t_ex = (time.perf_counter() - t0) / reps * 1e3
ref.release()
mb = arena.bytes_per_state / 1e6
say(f"hit install (arena row -> slot, one strided copy per tensor): {t_in:.3f} ms = {mb / t_in:.0f} GB/s;  "
    f"export (slot -> arena row): {t_ex:.3f} ms = {mb / t_ex:.0f} GB/s;  peak HBM in use {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")
if out_path:
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
