"""GPU: fused penalty+arg-max kernel vs the oracle, slot-pool forward vs the dense forward, and the
worker end to end on a tiny real model (continuous batching must not change any request's ids)."""
import os
import queue
import types

import numpy as np
import pytest
import torch

from oracle import rwkv7_np as M
from util import bits

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_penalize_argmax_matches_reference_fixture():
    """Same inputs as the golden penalty fixture (chirrup/worker.py:724-728 run by the reference)."""
    from chirrup_amd import ops

    d = np.load(os.path.join(G, "sampler.npz"))
    lg = torch.from_numpy(d["pen_logits"].copy()).cuda()
    occ = torch.from_numpy(d["pen_occurrence"].copy()).cuda()
    alpha = torch.from_numpy(d["pen_alpha"].copy()).cuda()
    decay = torch.full((3,), 0.996, dtype=torch.float16, device="cuda")
    freq = torch.full((3,), 0.5, dtype=torch.float16, device="cuda")
    ids = ops.penalize_argmax(lg, occ, alpha, decay, freq)
    assert np.array_equal(ids.cpu().numpy().astype(np.int64), d["pen_ids"])
    assert np.array_equal(bits(lg.cpu().numpy()), bits(d["pen_after"]))          # bit-exact fp16 logits
    assert np.array_equal(occ.cpu().numpy(), d["pen_occ_after"])                 # bit-exact fp32 occurrence


def test_penalize_argmax_slots_ties_and_plain_argmax():
    from chirrup_amd import ops

    rng = np.random.default_rng(0)
    B, V, n = 5, 65536, 9
    lg = (rng.standard_normal((B, V)) * 2).astype(np.float16)
    lg[0, 100] = lg[0, 77] = np.float16(30.0)                 # tie -> lowest id
    occ = rng.integers(0, 4, (n, V)).astype(np.float32) * (rng.uniform(size=(n, V)) < 0.01)
    alpha = 0.5 * (occ > 0).astype(np.float32)
    decay = rng.uniform(0.9, 1.0, n).astype(np.float16)
    freq = rng.uniform(0.0, 1.0, n).astype(np.float16)
    idx = np.array([7, 0, 3, 8, 2], np.int32)
    want_lg, want_occ = M.apply_penalties(lg, occ[idx], alpha[idx], decay[idx, None], freq[idx, None])
    t_lg, t_occ = torch.from_numpy(lg.copy()).cuda(), torch.from_numpy(occ.copy()).cuda()
    ids = ops.penalize_argmax(t_lg, t_occ, torch.from_numpy(alpha).cuda(), torch.from_numpy(decay).cuda(),
                              torch.from_numpy(freq).cuda(), torch.from_numpy(idx).cuda())
    assert np.array_equal(bits(t_lg.cpu().numpy()), bits(want_lg))
    got_occ = t_occ.cpu().numpy()
    assert np.array_equal(got_occ[idx], want_occ)
    untouched = [s for s in range(n) if s not in idx]
    assert np.array_equal(got_occ[untouched], occ[untouched])
    assert np.array_equal(ids.cpu().numpy(), M.greedy_sample(want_lg).astype(np.int32))
    plain = ops.penalize_argmax(torch.from_numpy(lg.copy()).cuda())
    assert plain.cpu().tolist() == lg.astype(np.float32).argmax(-1).tolist() and plain[0].item() == 77


def _tiny_model():
    from chirrup_amd.rwkv7 import RWKV_x070

    d = np.load(os.path.join(G, "model_L2_C128.npz"))
    zd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
    return d, RWKV_x070(types.SimpleNamespace(vocab_size=320, head_size=64, MODEL_NAME="unused"), state_dict=zd, device="cuda:0")


def test_forward_slots_equals_dense_forward():
    """Slot-pool addressing (no gathers, no swaps) gives bit-identical logits and states to running
    the same rows as a dense batch; untouched slots stay untouched.  T = 1 and T = 5."""
    d, m = _tiny_model()
    for tag in ("b3t1", "b3t5"):
        ins = [torch.from_numpy(d[f"{tag}:{n}_in"].copy()).cuda() for n in ("s0", "s1", "s2")]
        dense = [t.clone() for t in ins]
        lg_dense = m.forward_seq_batch_seperate(d[f"{tag}:tokens"].tolist(), dense)
        pool = m.generate_zero_state(8)
        torch.manual_seed(0)
        pool[0].copy_(torch.randn_like(pool[0].float()).half())
        pool[1].copy_((torch.randn_like(pool[1].float()) * 0.1).half())
        pool[2].copy_(torch.arange(8, dtype=torch.int32) * 3)
        before = [t.clone() for t in pool]
        slots = torch.tensor([6, 1, 4], dtype=torch.int32, device="cuda")
        sl = slots.long()
        pool[0][:, :, sl] = ins[0]
        pool[1][:, sl] = ins[1]
        pool[2][sl] = ins[2]
        lg = m.forward_slots(d[f"{tag}:tokens"].tolist(), pool, slots)
        assert torch.equal(lg, lg_dense)
        assert torch.equal(pool[0][:, :, sl], dense[0]) and torch.equal(pool[1][:, sl], dense[1]) and torch.equal(pool[2][sl], dense[2])
        rest = torch.tensor([0, 2, 3, 5, 7], device="cuda")
        for k in range(2):
            a = pool[k].index_select(2 if k == 0 else 1, rest)
            b = before[k].index_select(2 if k == 0 else 1, rest)
            assert torch.equal(a, b)
        assert torch.equal(pool[2][rest], before[2][rest])


class _Tok:
    def decode(self, ids, utf8_errors="strict"):
        return "".join(chr(65 + i % 26) for i in ids)


class _Sink:
    def __init__(self):
        self.items = []

    def put_nowait(self, x):
        self.items.append(x)


@pytest.mark.parametrize("run_ahead", [True, False], ids=["run_ahead", "in_step"])
def test_worker_end_to_end_matches_unbatched_greedy(run_ahead):
    """Requests with short / medium / long prompts through the continuous-batching worker (slot pool,
    fused sampler, chunked prefill) produce exactly the ids of decoding each request alone.
    The two golden prompts (top-2 margin >= 0.03 at every step) also match the REFERENCE's ids."""
    from chirrup_amd.core_structure import ModelLoadConfig, Task
    from chirrup_amd.worker import Worker

    d, m = _tiny_model()
    rng = np.random.default_rng(5)
    prompts = [d["greedy:prompt"][0].tolist(), d["greedy:prompt"][1].tolist()]
    prompts += [rng.integers(1, 320, n).tolist() for n in (1, 4, 14, 37, 120)]
    n_new = 16

    def solo(prompt):
        st = m.generate_zero_state(1)
        lg = m.forward_seq_batch_seperate([prompt], st)
        out, margins = [], []
        for _ in range(n_new):
            top2 = torch.topk(lg.float(), 2, dim=-1).values[0]
            margins.append(float(top2[0] - top2[1]))
            tok = int(lg.float().argmax(-1))
            out.append(tok)
            lg = m.forward_seq_batch_seperate([[tok]], st)
        return out, min(margins)

    cfg = ModelLoadConfig(model_path="unused", vocab_path="unused", vocab_size=320, head_size=64)
    tq, mq = queue.Queue(), queue.Queue()
    w = Worker("w0", [0], cfg, tq, mq, None, batch_size=5, model=m, tokenizer=_Tok(), run_ahead=run_ahead)
    w._init_worker()
    tasks = []
    for p in prompts:
        t = Task(output_queue=_Sink(), task_event_queue=queue.Queue(), prompt_str="", prefill_tokens=list(p), state=None,
                 temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[], max_tokens=n_new)
        tasks.append(t)
        tq.put(t)
    for _ in range(4000):
        if not w.step():
            break
    for i, (t, p) in enumerate(zip(tasks, prompts)):
        want, margin = solo(p)
        got = [x[1][0] for x in t.output_queue.items if x[0] == "token_generated"]
        if margin >= 0.02:          # ids are only defined where the arg-max is separated from fp16 noise
            assert got == want, (i, len(p), margin)
        else:
            k = next((j for j, (a, b) in enumerate(zip(got, want)) if a != b), n_new)
            assert k >= 1
    got0 = [x[1][0] for x in tasks[0].output_queue.items if x[0] == "token_generated"]
    got1 = [x[1][0] for x in tasks[1].output_queue.items if x[0] == "token_generated"]
    assert got0 == d["greedy:ids"][0].tolist() and got1 == d["greedy:ids"][1].tolist()


@pytest.mark.parametrize("V,temp,top_p,top_k", [(1024, 1.0, 0.3, 0), (1024, 1.0, 0.9, 0), (4096, 1.5, 1.0, 5), (65536, 0.7, 0.8, 10),
                                                (1024, 1.0, 0.0, 0), (1024, 1.0, 1.0, 1)])
def test_sample_topp_distribution(V, temp, top_p, top_k):
    """The sort-free sampler draws from the distribution the reference's algorithm defines
    (softmax -> top-p cutoff by value -> top-k -> p**(1/T)); frequencies over 20 000 draws match it to
    the reference's own statistical bar (tests/test_sampler_equivalence.py:110-143: 0.05; here 0.02),
    nothing outside the kept set is ever drawn, and the greedy corner cases return the arg-max."""
    from chirrup_amd import ops

    torch.manual_seed(V + int(top_p * 100) + top_k)
    N = 20000
    logits = (torch.randn(1, V) * 2.5).half().cuda()
    p = torch.softmax(logits.float(), -1)[0].cpu()
    sp, order = torch.sort(p, descending=True)
    cs = torch.cumsum(sp, 0)
    cut = sp[min(int(torch.searchsorted(cs, torch.tensor(float(torch.tensor(top_p).half())))), V - 1)]
    q = torch.where(p < cut, torch.zeros_like(p), p)
    if top_k > 0:
        kth = sp[top_k - 1]
        q = torch.where(p < kth, torch.zeros_like(q), q)          # ties at the boundary are kept (documented)
    if temp != 1.0:
        q = q ** (1.0 / temp)
    q = q / q.sum()
    rows = torch.zeros(N, dtype=torch.int32, device="cuda")
    u = torch.rand(N, device="cuda")
    t = torch.full((1,), temp, dtype=torch.float16, device="cuda")
    tp = torch.full((1,), top_p, dtype=torch.float16, device="cuda")
    tk = torch.full((1,), top_k, dtype=torch.int32, device="cuda")
    # N blocks on the same row, each with its own uniform; every block writes ids[0] -> give each its own row
    big = logits.expand(N, V).contiguous() if V <= 4096 else None
    if big is not None:
        rows = torch.arange(N, dtype=torch.int32, device="cuda")
        ids = torch.full((N,), -1, dtype=torch.int32, device="cuda")
        ops.sample_topp(big, rows, t.expand(N).contiguous(), tp.expand(N).contiguous(), tk.expand(N).contiguous(), u, ids)
        draws = ids.cpu().long()
    else:
        draws = []
        ids = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        for i in range(400):
            ops.sample_topp(logits, rows[:1], t, tp, tk, u[i:i + 1], ids)
            draws.append(int(ids[0]))
        draws = torch.tensor(draws)
        N = 400
    assert int(draws.min()) >= 0
    freq = torch.bincount(draws, minlength=V).float() / N
    assert float(freq[q == 0].sum()) == 0.0
    if N >= 20000:
        assert float((freq - q).abs().max()) <= 0.02
    if top_p == 0.0 or top_k == 1:
        assert bool((draws == int(p.argmax())).all())


@pytest.mark.parametrize("V", [1000, 50304, 65536])
def test_sample_topp_kept_set_and_the_ends_of_the_walk(V):
    """Per-row settings through slot tables, vocabularies that do not fill the kernel's 8192-token strides, and the two ends of
    the inverse-CDF walk: with u = 0 the draw is the FIRST token (in id order) of the kept set the reference's algorithm defines
    (chirrup/utils/samplers.py:171-255: cutoff by value at the first crossing of top_p, ties kept, then top-k), with u just
    below 1 the LAST one; nothing outside the set is drawn for u in between."""
    from chirrup_amd import ops

    torch.manual_seed(V)
    top_p = [0.1, 0.5, 0.9, 0.3, 0.7, 1.0]
    top_k = [0, 0, 0, 3, 50, 0]
    n = len(top_p)
    logits = (torch.randn(n, V) * 2.5)
    logits[:, :8] -= 30.0                 # the first and last tokens far outside every kept set but top_p = 1's
    logits[:, -8:] -= 30.0
    logits = logits.half().cuda()
    prob = torch.softmax(logits.float(), -1).cpu()
    kept = []                             # per row: the kept sets for the crossing index and its two neighbours (the kernel adds the
    for i in range(n):                    # same masses in another order: a crossing within ~1e-5 of top_p may fall one token either way)
        sp, _ = torch.sort(prob[i], descending=True)
        cs = torch.cumsum(sp, 0)
        tpv = float(torch.tensor(top_p[i]).half())
        j = min(int(torch.searchsorted(cs, torch.tensor(tpv))), V - 1)
        near = [j] + [jj for jj in (j - 1, j + 1) if 0 <= jj < V and min(abs(float(cs[max(jj, j) - 1]) - tpv), abs(float(cs[min(jj, j)]) - tpv)) < 1e-5]
        sets = []
        for jj in near:
            keep = prob[i] >= sp[jj]
            if top_k[i] > 0:
                keep &= prob[i] >= sp[top_k[i] - 1]
            sets.append(keep)
        kept.append(sets)
    # slots in a permuted order, rows listed out of order
    slot_idx = torch.tensor([4, 2, 5, 0, 1, 3], dtype=torch.int32, device="cuda")
    t = torch.ones(n, dtype=torch.float16, device="cuda")
    tp = torch.zeros(n, dtype=torch.float16, device="cuda")
    tk = torch.zeros(n, dtype=torch.int32, device="cuda")
    for i in range(n):
        tp[int(slot_idx[i])] = top_p[i]
        tk[int(slot_idx[i])] = top_k[i]
    rows = torch.tensor([3, 0, 5, 1, 4, 2], dtype=torch.int32, device="cuda")

    def draw(u):
        ids = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        ops.sample_topp(logits, rows, t, tp, tk, torch.full((n,), u, dtype=torch.float32, device="cuda"), ids, slot_idx=slot_idx)
        return ids.cpu().tolist()

    first, last, mid = draw(0.0), draw(1.0 - 2.0 ** -24), draw(0.37)
    for i in range(n):
        if top_p[i] == 1.0:             # where binary32 cumsum first reaches 1.0 is rounding in the reference too (it drops < 1e-6 of mass
            assert all(0 <= v < V for v in (first[i], mid[i], last[i]))      # there or nothing); the kernel keeps every token
            continue
        ok = False
        for keep in kept[i]:
            idx = torch.nonzero(keep).flatten()
            ok |= first[i] == int(idx[0]) and bool(keep[mid[i]]) and last[i] == int(idx[-1])
        assert ok, (i, first[i], mid[i], last[i], [int(torch.nonzero(k).flatten()[0]) for k in kept[i]], [int(k.sum()) for k in kept[i]])


@pytest.mark.parametrize("worker_has_arena", [False, True], ids=["clone_export", "arena_export"])
def test_prefix_cache_in_hbm_arena_round_trip(worker_has_arena):
    """Row 8f-1 end to end: the worker exports a prefix state (device-resident; with an arena straight into a free row:
    ONE copy), the arena-backed cache keeps it, a later request with the same prefix gets a pinned row handle, the
    worker installs row -> slot with ONE copy and produces exactly the ids of the uncached request; evicting the entry
    while that hit is queued does not disturb it."""
    from chirrup_amd.core_structure import ModelLoadConfig, Task
    from chirrup_amd.state_cache import ArenaRef, HbmStateArena, SimpleStateCache
    from chirrup_amd.worker import Worker

    d, m = _tiny_model()
    arena = HbmStateArena.for_model(m, capacity=3)            # max_size + one row for a state in flight
    cache = SimpleStateCache(max_size=2, arena=arena)
    cfg = ModelLoadConfig(model_path="unused", vocab_path="unused", vocab_size=320, head_size=64)
    tq, mq = queue.Queue(), queue.Queue()
    w = Worker("w0", [0], cfg, tq, mq, None, batch_size=4, model=m, tokenizer=_Tok(), state_arena=arena if worker_has_arena else None)
    w._init_worker()
    rng = np.random.default_rng(3)
    prompt = rng.integers(1, 320, 40).tolist()

    def run(task):
        tq.put(task)
        for _ in range(2000):
            if not w.step():
                break
        return [x[1][0] for x in task.output_queue.items if x[0] == "token_generated"]

    mk = lambda toks, state=None, **kw: Task(output_queue=_Sink(), task_event_queue=queue.Queue(), prompt_str="",
                                             prefill_tokens=list(toks), state=state, temperature=0.0, frequency_penalty=0.0,
                                             presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[], max_tokens=8, **kw)
    t1 = mk(prompt, cache_prefill=True, cache_prefill_padding=3)
    ids1 = run(t1)
    exported = [x[1] for x in t1.output_queue.items if x[0] == "cache_prefill"]
    assert len(exported) == 1
    ex_state = exported[0]["state"]
    assert isinstance(ex_state, ArenaRef) == worker_has_arena
    ex_tensors = ex_state.tensors() if worker_has_arena else ex_state
    assert ex_tensors[1].is_cuda and tuple(ex_tensors[1].shape) == (2, 1, 2, 64, 64)
    seen = tuple(exported[0]["prefilled_tokens"])
    cache.cache(seen, ex_state)                               # adopts the worker's row (arena export) or copies once
    assert arena.free_rows == 2
    rest, hit, n = cache.check(list(prompt))
    assert n == len(seen) and rest == prompt[n:] and isinstance(hit, ArenaRef)
    got = hit.tensors()
    assert torch.equal(got[1], ex_tensors[1]) and torch.equal(got[0], ex_tensors[0]) and int(got[2][0]) == n
    cache.cache((1, 2, 3), ex_tensors)
    cache.cache((4, 5, 6), ex_tensors)                        # evicts `seen` while the hit is still queued: its row stays pinned
    assert cache.check(list(prompt))[1] is None and arena.free_rows == 0
    ids2 = run(mk(rest, state=hit))                           # Worker._install: row -> slot, then the pin is released
    assert ids2 == ids1 and len(ids1) == 8
    assert arena.free_rows == 1 and not hit._live


def test_worker_stress_protocol_consistency():
    """Staggered arrivals, every prompt-length class, greedy and sampled rows, penalties, stop tokens, aborts and
    prefix exports on a 0.1B synthetic model over 16 slots (run-ahead on): every request must see exactly one
    task_completed, as its last message; tokens must match the task's own record and respect max_tokens; all slots
    must be free at the end.  (tools/stress_worker.py is the larger form of this.)"""
    import random

    from chirrup_amd.core_structure import ModelLoadConfig, RequestStatus, Task
    from chirrup_amd.rwkv7 import RWKV_x070, model_args
    from chirrup_amd.synth import CONFIGS, make_state_dict
    from chirrup_amd.worker import Worker

    L, C = CONFIGS["0.1B"]
    dev = torch.device("cuda", 0)
    model = RWKV_x070(model_args("synthetic"), state_dict=make_state_dict(L, C, 65536, seed=3, device=dev), device=dev)
    cfg = ModelLoadConfig(model_path="synthetic", vocab_path="none", vocab_size=65536, head_size=64)
    tq, mq = queue.Queue(), queue.Queue()
    w = Worker("w0", [0], cfg, tq, mq, None, batch_size=17, model=model, tokenizer=_Tok())
    w._init_worker()
    rng = random.Random(11)
    tasks, pending = [], []
    for _ in range(120):
        greedy = rng.random() < 0.5
        t = Task(output_queue=_Sink(), task_event_queue=queue.Queue(), prompt_str="",
                 prefill_tokens=[rng.randrange(1, 65536) for _ in range(rng.choice([1, 2, 9, 10, 11, 40, 100, 101, 230]))], state=None,
                 temperature=0.0 if greedy else 1.0, top_p=0.0 if greedy else rng.choice([0.3, 0.9, 1.0]), top_k=rng.choice([0, 0, 20]),
                 frequency_penalty=rng.choice([0.0, 0.5]), presence_penalty=rng.choice([0.0, 0.5]), penalty_decay=0.996,
                 stop_tokens=[] if rng.random() < 0.7 else [rng.randrange(1, 65536) for _ in range(3000)], max_tokens=rng.randrange(1, 40),
                 cache_prefill=rng.random() < 0.2, cache_prefill_padding=rng.choice([0, 1, 3]))
        t._abort_at = rng.randrange(2, 60) if rng.random() < 0.1 else None
        tasks.append(t)
        pending.append(t)
    it = 0
    while True:
        for _ in range(rng.randrange(0, 4)):
            if pending:
                tq.put(pending.pop())
        for t in tasks:
            if t._abort_at == it:
                t.task_event_queue.put(("abort", None))
        busy = w.step()
        it += 1
        if not busy and not pending:
            break
        assert it < 20000
    outcomes = set()
    for t in tasks:
        kinds = [k for k, _ in t.output_queue.items]
        assert kinds.count("task_completed") == 1 and kinds[-1] == "task_completed"
        toks = [p[0] for k, p in t.output_queue.items if k == "token_generated"]
        assert toks == t.generated_tokens and len(toks) <= t.max_tokens and all(0 <= x < 65536 for x in toks)
        assert RequestStatus.is_finished(t.request_status)
        if t.request_status == RequestStatus.FINISHED_LENGTH_CAPPED:
            assert len(toks) == t.max_tokens
        elif t.request_status == RequestStatus.FINISHED_STOPPED:
            assert len(toks) < t.max_tokens
        outcomes.add(t.request_status)
    assert all(td["task"] is None for td in w.state_slot.values())
    assert {RequestStatus.FINISHED_LENGTH_CAPPED, RequestStatus.FINISHED_ABORTED} <= outcomes


def test_worker_values_in_the_hand_written_gemm_regime():
    """Round-1 advisor finding: the worker's VALUES were only checked on the L2/C128 toy model with 5 slots.  Here a
    0.1B-shaped stack (C = 768, real LoRA ranks, every GEMM hand-written: skinny_min_embd = 0) serves 36 requests at
    once -- 39 slots plus the parking slot, graph buckets up to 64 rows, prompts of 120-250 tokens through the chunked
    prefill (up to 12 x 100 rows per forward) interleaved with decode steps -- and every request's greedy stream must
    equal that request decoded ALONE (one prefill of the whole prompt, then single steps), up to the first step
    whose solo top-2 logit margin is below fp16 noise (0.03: different batch compositions take different GEMM tiles)."""
    from chirrup_amd.core_structure import ModelLoadConfig, Task
    from chirrup_amd.rwkv7 import RWKV_x070
    from chirrup_amd.synth import make_state_dict
    from chirrup_amd.worker import Worker

    L, C, V, n_req, new = 3, 768, 1024, 36, 64
    zd = make_state_dict(L, C, V, seed=21, varied_norms=True)
    m = RWKV_x070(types.SimpleNamespace(vocab_size=V, head_size=64, MODEL_NAME="unused"), state_dict=zd, device="cuda:0",
                  skinny_min_embd=0)
    assert m._layers[0].rkv_t is not None
    rng = np.random.default_rng(17)
    prompts = [rng.integers(1, V, int(rng.integers(120, 251))).tolist() for _ in range(n_req)]
    # solo decode: ids + the margin of every decision
    solo = []
    for p in prompts:
        st = m.generate_zero_state(1)
        lg = m.forward_seq_batch_seperate([p], st)
        ids, margins = [], []
        for _ in range(new):
            top2 = torch.topk(lg[0].float(), 2).values
            margins.append(float(top2[0] - top2[1]))
            tok = int(lg[0].float().argmax())
            ids.append(tok)
            lg = m.forward_seq_batch_seperate([[tok]], st)
        solo.append((ids, margins))
    cfg = ModelLoadConfig(model_path="unused", vocab_path="unused", vocab_size=V, head_size=64)
    tq, mq = queue.Queue(), queue.Queue()
    w = Worker("w0", [0], cfg, tq, mq, None, batch_size=40, model=m, tokenizer=_Tok())
    w.max_prefill_count = 12                            # up to 12 x 100 prompt rows per prefill forward (library-GEMM regime)
    w._init_worker()
    tasks = [Task(output_queue=_Sink(), task_event_queue=queue.Queue(), prompt_str="", prefill_tokens=list(p), state=None,
                  temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[], max_tokens=new)
             for p in prompts]
    for i, t in enumerate(tasks):                       # staggered arrivals: prefill chunks and decode rows share iterations
        tq.put(t)
        if i % 9 == 8:
            for _ in range(4):
                w.step()
    for _ in range(5000):
        if not w.step():
            break
    assert max(w._graphs) >= 32                         # the wide graph buckets were used
    exact = 0
    for t, (ids, margins) in zip(tasks, solo):
        got = [x[1][0] for x in t.output_queue.items if x[0] == "token_generated"]
        assert len(got) == new
        for k in range(new):
            if got[k] != ids[k]:
                assert margins[k] < 0.03, (k, got, ids, margins[k])
                break
        else:
            exact += 1
    assert exact >= n_req * 3 // 4, exact               # near-ties are rare: most streams agree to the last token


def test_commit_sampled_kernel_matches_the_torch_ops():
    """ops.commit_sampled == Worker._commit_sampled's torch form (chirrup/worker.py:527-535): last ids, occurrence increments by the
    per-token weight (0 for the no-penalty ids), presence entries -- on a slot subset in scrambled order, bit for bit; a repeated
    slot accumulates like index_put_(accumulate=True); an id outside the vocabulary only lands in last_ids."""
    from chirrup_amd import ops

    torch.manual_seed(0)
    dev, n_slots, V, n = "cuda", 37, 4096, 21
    occ = torch.rand(n_slots, V, device=dev)
    alpha = torch.rand(n_slots, V, device=dev)
    pw = torch.ones(V, device=dev)
    pw[[0, 11, 33, 261]] = 0.0
    presence = torch.rand(n_slots, 1, device=dev)
    last = torch.full((n_slots,), -5, dtype=torch.int32, device=dev)
    slots = torch.randperm(n_slots, device=dev)[:n].to(torch.int32)
    ids = torch.randint(0, V, (n,), device=dev, dtype=torch.int32)
    ids[:4] = torch.tensor([0, 11, 33, 261], dtype=torch.int32, device=dev)          # no-penalty ids
    want_occ, want_alpha, want_last = occ.clone(), alpha.clone(), last.clone()
    dl, il = slots.long(), ids.long()
    want_last.index_copy_(0, dl, ids)
    want_occ.index_put_((dl, il), pw[il], accumulate=True)
    want_alpha[dl, il] = presence[dl, 0]
    ops.commit_sampled(ids, slots, last, occ, pw, alpha, presence)
    assert torch.equal(last, want_last) and torch.equal(occ, want_occ) and torch.equal(alpha, want_alpha)
    # no slot_idx: rows are slots
    occ2, alpha2, last2 = torch.zeros(n, V, device=dev), torch.zeros(n, V, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
    ops.commit_sampled(ids, None, last2, occ2, pw, alpha2, presence[:n].contiguous())
    assert torch.equal(last2, ids) and float(occ2.sum()) == float(pw[il].sum())
    assert torch.equal(alpha2[torch.arange(n, device=dev), il], presence[:n, 0])
    # a slot twice in one call accumulates; an id outside [0, V) touches no table
    occ3, alpha3, last3 = torch.zeros(4, V, device=dev), torch.zeros(4, V, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
    ops.commit_sampled(torch.tensor([7, 7, V + 3, -1], dtype=torch.int32, device=dev), torch.tensor([2, 2, 1, 3], dtype=torch.int32, device=dev),
                       last3, occ3, pw, alpha3, torch.ones(4, 1, device=dev))
    assert float(occ3[2, 7]) == 2.0 and float(occ3.sum()) == 2.0 and float(alpha3.sum()) == 1.0
    assert last3.tolist() == [0, V + 3, 7, -1]
