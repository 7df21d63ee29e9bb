"""Host logic of the continuous-batching worker (row A10) with a fake backend on CPU, the way the
reference's own tests fake the model (tests/test_worker_state_category.py:122-127)."""
import json
import os
import queue

import numpy as np
import pytest
import torch

from chirrup_amd.core_structure import ModelLoadConfig, RequestStatus, Task
from chirrup_amd.worker import StateCategory, Worker, min_swaps_to_target_fast

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
V = 64


def test_state_category_order_matches_reference():
    assert [c.name for c in sorted(StateCategory)] == ["FORWARD_ONE_DECODE", "FORWARD_ONE_PREFILL", "FORWARD_ONE_SUSPENDED",
                                                        "FORWARD_SEQ", "FINISHED", "EMPTY"]


def test_min_swaps_matches_reference_fixture():
    for c in json.load(open(os.path.join(G, "scheduler.json")))["cases"]:
        lst = [StateCategory(v) for v in c["input"]]
        swaps, offsets = min_swaps_to_target_fast(lst, list(sorted(StateCategory)))
        assert [list(s) for s in swaps] == c["swaps"] and [list(o) for o in offsets] == c["offsets"]
        assert [int(v) for v in lst] == c["after"]


class FakeModel:
    """Next-token logits are a pure function of the slot's whole token history, carried in the state
    pool exactly where the real model keeps it, so any mix-up of slots / order / counts shows."""

    def generate_zero_state(self, n):
        return [torch.zeros((1, 2, n, 4), dtype=torch.float16), torch.zeros((1, n, 1, 1, 1), dtype=torch.float32),
                torch.zeros((n,), dtype=torch.int32)]

    @staticmethod
    def advance(h, tok):
        return (h * 31 + tok + 7) % 9973

    def forward_slots(self, tokens, pool, slot_idx, full_output=False):
        out = torch.zeros((len(tokens), V), dtype=torch.float16)
        for b, (toks, s) in enumerate(zip(tokens, slot_idx.tolist())):
            h = int(pool[1][0, s, 0, 0, 0])
            for t in toks:
                h = self.advance(h, int(t))
            pool[1][0, s, 0, 0, 0] = h
            pool[2][s] += len(toks)
            out[b, h % V] = 5.0
            out[b, (h + 1) % V] = 4.0            # runner-up, promoted when the winner is penalised away
        return out


class FakeTok:
    def decode(self, ids, utf8_errors="strict"):
        return "".join(f"<{i}>" for i in ids)


def cpu_penalize_argmax(logits, occ, alpha, decay, freq, slot_idx):
    s = slot_idx.long()
    occ[s] = occ[s] * decay[s].float().unsqueeze(1)
    pen = (logits.float() - (alpha[s] + occ[s] * freq[s].float().unsqueeze(1))).half()
    logits.copy_(pen)
    return pen.float().argmax(-1).to(torch.int32)


def expected_stream(prompt, n_new, stop=(), freq=0.0, pres=0.0, decay=1.0, state_h=0):
    h = state_h
    for t in prompt:
        h = FakeModel.advance(h, t)
    out, occ, alpha = [], np.zeros(V, np.float32), np.zeros(V, np.float32)
    for _ in range(n_new):
        lg = np.zeros(V, np.float32)
        lg[h % V], lg[(h + 1) % V] = 5.0, 4.0
        occ = (occ * np.float32(np.float16(decay))).astype(np.float32)
        lg = (lg - (alpha + occ * np.float32(np.float16(freq)))).astype(np.float16).astype(np.float32)
        tok = int(lg.argmax())
        if tok in stop:
            break
        out.append(tok)
        if tok not in {33, 10, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58}:
            occ[tok] += 1.0
        alpha[tok] = pres
        h = FakeModel.advance(h, tok)
    return out


class Sink:
    def __init__(self):
        self.items = []

    def put_nowait(self, x):
        self.items.append(x)


RUN_AHEAD = True


@pytest.fixture(autouse=True, params=[True, False], ids=["run_ahead", "in_step"])
def _both_scheduling_modes(request):
    """Every worker test runs with the ids handled one forward behind (run-ahead) and in the same step."""
    global RUN_AHEAD
    RUN_AHEAD = request.param
    yield


def make_worker(batch_size=6):
    cfg = ModelLoadConfig(model_path="fake", vocab_path="fake", vocab_size=V, head_size=64)
    tq, mq, wq = queue.Queue(), queue.Queue(), queue.Queue()
    w = Worker("w0", [], cfg, tq, mq, wq, batch_size=batch_size, model=FakeModel(), tokenizer=FakeTok(),
               penalize_argmax=cpu_penalize_argmax, run_ahead=RUN_AHEAD)
    w._init_worker()
    return w, tq, mq, wq


def new_task(prompt, **kw):
    kw.setdefault("temperature", 0.0)
    kw.setdefault("frequency_penalty", 0.0)
    kw.setdefault("presence_penalty", 0.0)
    kw.setdefault("penalty_decay", 1.0)
    kw.setdefault("stop_tokens", [])
    return Task(output_queue=Sink(), task_event_queue=queue.Queue(), prompt_str="", prefill_tokens=list(prompt), state=None, **kw)


def run_until_idle(w, max_iter=2000):
    for _ in range(max_iter):
        if not w.step():
            return
    raise AssertionError("worker did not drain")


def tokens_of(task):
    return [p[0] for kind, p in task.output_queue.items if kind == "token_generated"]


def test_continuous_batching_streams_are_per_request_exact():
    """More requests than slots, prompts of every length class (1 token -> decode at once, < 10 ->
    single-token prefill, long -> chunked prefill incl. a > 100 token prompt), different max_tokens."""
    w, tq, _, wq = make_worker(batch_size=6)         # 5 usable slots
    assert wq.get_nowait()[1] == "worker_loaded"
    rng = np.random.default_rng(0)
    specs = [(1, 5), (3, 9), (9, 4), (25, 7), (130, 6), (11, 12), (2, 3), (250, 5), (40, 8)]
    tasks = []
    for plen, n_new in specs:
        t = new_task(rng.integers(1, V, plen).tolist(), max_tokens=n_new)
        t._prompt = list(t.prefill_tokens)
        tasks.append(t)
        tq.put(t)
    run_until_idle(w)
    for t, (plen, n_new) in zip(tasks, specs):
        assert tokens_of(t) == expected_stream(t._prompt, n_new), (plen, n_new)
        assert t.generated_tokens == tokens_of(t)
        assert t.request_status == RequestStatus.FINISHED_LENGTH_CAPPED
        assert t.output_queue.items[-1] == ("task_completed", t)
        assert t.decoded_texts == [f"<{i}>" for i in t.generated_tokens]
    assert all(td["state_category"] == StateCategory.EMPTY for td in w.state_slot.values())
    kinds = set()
    while not wq.empty():
        kinds.add(wq.get_nowait()[1])
    assert "worker_performance" in kinds


def test_stop_tokens_penalties_and_abort():
    w, tq, _, _ = make_worker(batch_size=4)
    p = [5, 6, 7]
    free = expected_stream(p, 20)
    stop_tok = free[4]
    t_stop = new_task(p, max_tokens=20, stop_tokens=[stop_tok])
    t_pen = new_task(p, max_tokens=20, frequency_penalty=2.0, presence_penalty=1.0, penalty_decay=0.996)
    t_abort = new_task(p, max_tokens=10 ** 6)
    for t in (t_stop, t_pen, t_abort):
        tq.put(t)
    for _ in range(8):
        w.step()
    t_abort.task_event_queue.put(("abort", None))
    run_until_idle(w)
    assert tokens_of(t_stop) == free[:free.index(stop_tok)] and t_stop.request_status == RequestStatus.FINISHED_STOPPED
    assert tokens_of(t_pen) == expected_stream(p, 20, freq=2.0, pres=1.0, decay=0.996)
    assert tokens_of(t_pen) != free                      # the penalties really changed the stream
    assert t_abort.request_status == RequestStatus.FINISHED_ABORTED and 0 < len(tokens_of(t_abort)) < 20
    assert tokens_of(t_abort) == free[:len(tokens_of(t_abort))] or len(free) < len(tokens_of(t_abort))


def test_prefix_state_export_and_reuse():
    """cache_prefill emits the slot's state `cache_prefill_padding` tokens before the prompt end
    (worker.py:411-435, :457-476); feeding it back as Task.state continues the same stream."""
    w, tq, _, _ = make_worker(batch_size=4)
    prompt = list(range(1, 31))
    t1 = new_task(prompt, max_tokens=6, cache_prefill=True, cache_prefill_padding=3)
    tq.put(t1)
    run_until_idle(w)
    caches = [p for kind, p in t1.output_queue.items if kind == "cache_prefill"]
    assert len(caches) == 1
    st, seen = caches[0]["state"], list(caches[0]["prefilled_tokens"])
    assert [tuple(x.shape) for x in st] == [(1, 2, 1, 4), (1, 1, 1, 1, 1), (1,)]
    assert seen == prompt[:len(seen)] and len(prompt) - len(seen) in (2, 3)
    assert int(st[2][0]) == len(seen)
    t2 = new_task(prompt[len(seen):], max_tokens=6)
    t2.state = [x.clone() for x in st]
    tq.put(t2)
    run_until_idle(w)
    assert tokens_of(t2) == tokens_of(t1) == expected_stream(prompt, 6)


def test_return_logits_and_forbidden_tokens():
    w, tq, _, _ = make_worker(batch_size=3)
    p = [9, 8]
    free = expected_stream(p, 3)
    t = new_task(p, max_tokens=3, return_logits=True, forbidden_tokens=[free[0]])
    tq.put(t)
    run_until_idle(w)
    toks = [x for x in t.output_queue.items if x[0] == "token_generated"]
    assert len(toks[0][1]) == 3 and toks[0][1][2].shape == (V,)          # (id, text, raw logits before masking)
    assert float(toks[0][1][2].max()) == 5.0
    assert toks[0][1][0] != free[0]                                        # the forbidden arg-max was masked out


def test_shutdown_event_stops_the_loop():
    w, tq, mq, _ = make_worker(batch_size=3)
    mq.put({"type": "shutdown"})
    w.start()                                                              # returns instead of spinning
    assert w.shutdown_flag
