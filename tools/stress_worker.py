"""Worker stress run on a synthetic-weight model: staggered arrivals, prompts of every length class, mixed sampling
parameters, aborts; checks every request's message stream for protocol consistency (not for token values).
usage: python tools/stress_worker.py [model=1.5B] [n_requests=300] [slots=65]"""
import os, queue, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chirrup_amd.core_structure import ModelLoadConfig, RequestStatus, Task
from chirrup_amd.rwkv7 import RWKV_x070, model_args
from chirrup_amd.synth import CONFIGS, make_state_dict
from chirrup_amd.worker import Worker

name = sys.argv[1] if len(sys.argv) > 1 else "1.5B"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 65
L, C = CONFIGS[name]
dev = torch.device("cuda", 0)
model = RWKV_x070(model_args("synthetic"), state_dict=make_state_dict(L, C, 65536, seed=42, device=dev), device=dev)


class Tok:
    def decode(self, ids, utf8_errors="strict"):
        return f"<{ids[0]}>"


class Sink:
    def __init__(self):
        self.items = []

    def put_nowait(self, x):
        self.items.append(x)


cfg = ModelLoadConfig(model_path="synthetic", vocab_path="none", vocab_size=65536, head_size=64)
tq, mq = queue.Queue(), queue.Queue()
w = Worker("w0", [0], cfg, tq, mq, None, batch_size=slots, model=model, tokenizer=Tok())
w._init_worker()
rng = random.Random(7)
tasks, pending = [], []
for i in range(N):
    plen = rng.choice([1, 2, 5, 9, 10, 11, 30, 99, 100, 101, 250])
    greedy = rng.random() < 0.5
    t = Task(output_queue=Sink(), task_event_queue=queue.Queue(), prompt_str="", prefill_tokens=[rng.randrange(1, 65536) for _ in range(plen)],
             state=None, temperature=0.0 if greedy else 1.0, top_p=0.0 if greedy else rng.choice([0.3, 0.9, 1.0]), top_k=rng.choice([0, 0, 20]),
             frequency_penalty=rng.choice([0.0, 0.5]), presence_penalty=rng.choice([0.0, 0.5]), penalty_decay=0.996,
             stop_tokens=[] if rng.random() < 0.7 else [rng.randrange(1, 65536) for _ in range(2000)], max_tokens=rng.randrange(1, 60),
             cache_prefill=rng.random() < 0.2, cache_prefill_padding=rng.choice([0, 1, 3]))
    t._abort_at = rng.randrange(2, 40) if rng.random() < 0.1 else None
    tasks.append(t)
    pending.append(t)
t0 = time.perf_counter()
it = 0
while True:
    for _ in range(rng.randrange(0, 6)):            # staggered arrivals
        if pending:
            tq.put(pending.pop())
    for t in tasks:
        if t._abort_at is not None and it == t._abort_at * 3:
            t.task_event_queue.put(("abort", None))
    busy = w.step()
    it += 1
    if not busy and not pending:
        break
    assert it < 200000
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n_tok = n_abort = n_stop = n_cap = n_cache = 0
for t in tasks:
    kinds = [k for k, _ in t.output_queue.items]
    assert kinds.count("task_completed") == 1 and kinds[-1] == "task_completed", kinds[-3:]
    toks = [p[0] for k, p in t.output_queue.items if k == "token_generated"]
    assert toks == t.generated_tokens and len(toks) <= t.max_tokens
    assert all(0 <= x < 65536 for x in toks)
    n_tok += len(toks)
    n_cache += kinds.count("cache_prefill")
    st = t.request_status
    assert RequestStatus.is_finished(st)
    if st == RequestStatus.FINISHED_ABORTED:
        n_abort += 1
    elif st == RequestStatus.FINISHED_STOPPED:
        n_stop += 1
        assert len(toks) < t.max_tokens
    else:
        n_cap += 1
        assert len(toks) == t.max_tokens
assert all(td["task"] is None for td in w.state_slot.values())
print(f"stress {name}: {N} requests over {slots - 1} slots, {it} iterations, {n_tok} tokens in {dt:.1f}s; "
      f"length-capped {n_cap}, stopped {n_stop}, aborted {n_abort}, prefix exports {n_cache}; all streams consistent")
