"""Engine façade (B4) with fake-backend workers on CPU: two worker threads sharing one task queue,
streaming, full completion, abort, prefix-cache events, telemetry, shutdown."""
import asyncio
import queue

import pytest

from chirrup_amd.core_structure import ModelLoadConfig
from chirrup_amd.engine_core import AsyncEngineCore
from chirrup_amd.worker import Worker
from test_worker_cpu import FakeModel, FakeTok, V, cpu_penalize_argmax, expected_stream


class _Tok(FakeTok):
    def encode(self, s):
        return [ord(c) % (V - 1) + 1 for c in s]


def _factory(**kw):
    return Worker(model=FakeModel(), tokenizer=_Tok(), penalize_argmax=cpu_penalize_argmax, **kw)


def test_engine_end_to_end():
    async def main():
        eng = AsyncEngineCore(worker_factory=_factory, tokenizer=_Tok())
        cfg = ModelLoadConfig(model_path="fake", vocab_path="fake", vocab_size=V, head_size=64)
        await eng.init(worker_num=2, model_config=cfg, batch_size=4)
        assert len(eng.workers) == 2 and {w.gpu_id[0] for w in eng.workers} == {0, 1}
        kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[])
        # 1) streamed events
        prompt = [3, 1, 4, 1, 5]
        c = eng.completion("", prefill_tokens=list(prompt), max_tokens=7, **kw)
        toks = [ev[1] async for ev in c if ev[0] == "token"]
        assert toks == expected_stream(prompt, 7) and c.is_finished
        assert c.task.generated_tokens == toks
        # 2) many concurrent requests over both workers, text via get_full_completion, prompt_str tokenised
        cs = [eng.completion("hello %d" % i, max_tokens=5, **kw) for i in range(9)]
        texts = await asyncio.gather(*[x.get_full_completion() for x in cs])
        for i, (x, text) in enumerate(zip(cs, texts)):
            want = expected_stream(_Tok().encode("hello %d" % i), 5)
            assert text == "".join(f"<{t}>" for t in want)
        # 3) abort
        c = eng.completion("", prefill_tokens=[9, 9], max_tokens=10 ** 6, **kw)
        n = 0
        async for ev in c:
            n += 1
            if n == 5:
                c.abort()
        assert 5 <= n < 200 and str(c.task.request_status) == "FINISHED_ABORTED"
        # 4) cache_prefill event carries a state that resumes the same stream
        p = list(range(1, 25))
        c = eng.completion("", prefill_tokens=list(p), max_tokens=4, cache_prefill=True, cache_prefill_padding=3, **kw)
        evs = [ev async for ev in c]
        cache = [e[1] for e in evs if e[0] == "cache_prefill"]
        assert len(cache) == 1
        seen = list(cache[0]["prefilled_tokens"])
        c2 = eng.completion("", prefill_tokens=p[len(seen):], state=cache[0]["state"], max_tokens=4, **kw)
        assert [ev[1] async for ev in c2 if ev[0] == "token"] == expected_stream(p, 4)
        # 5) telemetry
        async for perf in eng.iter_worker_performance(timeout=0.2):
            assert {"worker_id", "avg_loop_time", "state_size", "task_details"} <= set(perf)
            break
        eng.shutdown()
        assert all(not t.is_alive() for t in eng.worker_threads)
        with pytest.raises(RuntimeError):
            eng.completion("x")

    asyncio.run(main())


@pytest.mark.filterwarnings("ignore::pytest.PytestUnhandledThreadExceptionWarning")
def test_worker_failure_releases_every_client():
    """A worker whose loop dies (here: the backend raises) must not leave clients waiting: requests in its slots
    and, when it was the last worker, requests still queued are completed as aborted; a load failure fails init."""
    class Boom(FakeModel):
        def forward_slots(self, *a, **k):
            raise RuntimeError("backend failure")

    def factory(**kw):
        return Worker(model=Boom(), tokenizer=_Tok(), penalize_argmax=cpu_penalize_argmax, **kw)

    async def main():
        eng = AsyncEngineCore(worker_factory=factory, tokenizer=_Tok())
        cfg = ModelLoadConfig(model_path="fake", vocab_path="fake", vocab_size=V, head_size=64)
        await eng.init(worker_num=1, model_config=cfg, batch_size=3)
        kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[], max_tokens=5)
        cs = [eng.completion("", prefill_tokens=[1, 2, 3], **kw) for _ in range(6)]     # more than the slots
        outs = await asyncio.wait_for(asyncio.gather(*[c.get_full_completion() for c in cs]), 10.0)
        assert outs == [""] * 6
        assert all(str(c.task.request_status) == "FINISHED_ABORTED" for c in cs)
        eng.shutdown()

    asyncio.run(main())


def _process_factory(**kw):
    """Runs INSIDE a spawned worker process (module-level: picklable)."""
    return Worker(model=FakeModel(), tokenizer=_Tok(), penalize_argmax=cpu_penalize_argmax, **kw)


def test_engine_with_one_process_per_worker():
    """worker_mode="process" ("on a GIL build use one process per GPU"): two spawned worker processes on the
    fake backend behind the same engine API -- streams equal the single-worker expectation whichever worker pulls a
    request, abort reaches the right process, prefix states cross the process boundary as host tensors, telemetry and
    shutdown work.  The engine process itself never calls into the GPU runtime."""
    import torch

    async def main():
        eng = AsyncEngineCore(worker_factory=_process_factory, tokenizer=_Tok(), worker_mode="process")
        cfg = ModelLoadConfig(model_path="fake", vocab_path="fake", vocab_size=V, head_size=64)
        await asyncio.wait_for(eng.init(worker_num=2, model_config=cfg, batch_size=4), 120)
        assert len(eng.workers) == 2 and all(w.is_alive() for w in eng.workers) and {w.gpu_id[0] for w in eng.workers} == {0, 1}
        kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[])
        prompt = [3, 1, 4, 1, 5]
        c = eng.completion("", prefill_tokens=list(prompt), max_tokens=7, **kw)
        toks = [ev[1] async for ev in c if ev[0] == "token"]
        assert toks == expected_stream(prompt, 7) and c.is_finished and c.task.generated_tokens == toks
        assert str(c.task.request_status) == "FINISHED_LENGTH_CAPPED"
        cs = [eng.completion("hello %d" % i, max_tokens=5, **kw) for i in range(12)]       # more than one worker's slots
        texts = await asyncio.wait_for(asyncio.gather(*[x.get_full_completion() for x in cs]), 60)
        for i, text in enumerate(texts):
            assert text == "".join(f"<{t}>" for t in expected_stream(_Tok().encode("hello %d" % i), 5))
        assert len(set(eng._router.owner.values())) == 0                                    # nothing left in flight
        c = eng.completion("", prefill_tokens=[9, 9], max_tokens=10 ** 6, **kw)
        n = 0
        async for ev in c:
            n += 1
            if n == 5:
                c.abort()
        assert 5 <= n < 5000 and str(c.task.request_status) == "FINISHED_ABORTED"
        p = list(range(1, 25))
        c = eng.completion("", prefill_tokens=list(p), max_tokens=4, cache_prefill=True, cache_prefill_padding=3, **kw)
        evs = [ev async for ev in c]
        cache = [e[1] for e in evs if e[0] == "cache_prefill"]
        assert len(cache) == 1 and all(isinstance(t, torch.Tensor) and t.device.type == "cpu" for t in cache[0]["state"])
        seen = list(cache[0]["prefilled_tokens"])
        c2 = eng.completion("", prefill_tokens=p[len(seen):], state=cache[0]["state"], max_tokens=4, **kw)
        assert [ev[1] async for ev in c2 if ev[0] == "token"] == expected_stream(p, 4)      # state shipped to whichever worker pulled it
        async for perf in eng.iter_worker_performance(timeout=0.5):
            assert {"worker_id", "avg_loop_time", "state_size", "task_details"} <= set(perf)
            break
        eng.shutdown()
        assert all(not w.is_alive() for w in eng.workers)

    asyncio.run(main())
    assert not torch.cuda.is_initialized()


def test_abort_of_a_request_that_no_worker_has_pulled_yet():
    """Process mode, round-2 advisor finding: the abort of a request that still sits in the shared queue (every slot busy) is
    broadcast by task id before any worker owns the task.  The workers remember such ids (bounded), and the one that pulls the
    task later finds the abort in the task's own event queue on admission -- FINISHED_ABORTED, as in thread mode and the
    reference (interface.py:140-142, worker.py:443), instead of running to max_tokens."""
    async def main():
        eng = AsyncEngineCore(worker_factory=_process_factory, tokenizer=_Tok(), worker_mode="process")
        cfg = ModelLoadConfig(model_path="fake", vocab_path="fake", vocab_size=V, head_size=64)
        await asyncio.wait_for(eng.init(worker_num=1, model_config=cfg, batch_size=3), 120)      # 2 request slots + the scratch slot
        kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[])
        busy = [eng.completion("", prefill_tokens=[1, 2, 3 + i], max_tokens=10 ** 6, **kw) for i in range(2)]    # fill both slots
        its = [b.__aiter__() for b in busy]
        for it in its:
            await asyncio.wait_for(it.__anext__(), 30)                      # both are running
        queued = eng.completion("", prefill_tokens=[7, 7], max_tokens=10 ** 6, **kw)
        qit = queued.__aiter__()                                             # puts the task on the shared queue
        first = asyncio.ensure_future(qit.__anext__())
        await asyncio.sleep(0.3)
        assert queued.task.task_id not in eng._router.owner                  # nobody has pulled it: no free slot
        queued.abort()
        await asyncio.sleep(0.3)
        for b in busy:                                                       # now free the slots
            b.abort()
        for it in its:
            async for _ in it:
                pass
        n = 0
        try:
            await asyncio.wait_for(first, 30)
            n += 1
            async for _ in qit:
                n += 1
        except StopAsyncIteration:
            pass
        assert str(queued.task.request_status) == "FINISHED_ABORTED" and n < 50, (queued.task.request_status, n)
        eng.shutdown()

    asyncio.run(main())


class _DyingModel(FakeModel):
    """Ends its process the hard way in the middle of a request (what a HIP abort on a GPU fault looks like from outside)."""
    calls = 0

    def forward_slots(self, *a, **k):
        type(self).calls += 1
        if type(self).calls >= 4:
            import os

            os._exit(1)
        return super().forward_slots(*a, **k)


def _dying_factory(**kw):
    return Worker(model=_DyingModel(), tokenizer=_Tok(), penalize_argmax=cpu_penalize_argmax, **kw)


def _ctor_raises_factory(**kw):
    raise RuntimeError("cannot build the worker")


def test_hard_exit_of_a_worker_process_releases_its_clients_and_fails_init_fast():
    """Process mode, round-2 advisor finding: a worker process that dies without a goodbye (os._exit here; a HIP abort, a
    segmentation fault or an OOM kill in production) is noticed by the engine's liveness monitor on the process sentinel:
    running and queued requests complete as aborted (nothing is restarted); and a worker whose CONSTRUCTOR raises fails
    init() at once instead of after the 300-s load timeout."""
    import time

    async def main():
        eng = AsyncEngineCore(worker_factory=_dying_factory, tokenizer=_Tok(), worker_mode="process")
        cfg = ModelLoadConfig(model_path="fake", vocab_path="fake", vocab_size=V, head_size=64)
        await asyncio.wait_for(eng.init(worker_num=1, model_config=cfg, batch_size=3), 120)
        kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[], max_tokens=10 ** 6)
        cs = [eng.completion("", prefill_tokens=[1, 2, 3], **kw) for _ in range(4)]         # two running, two queued
        await asyncio.wait_for(asyncio.gather(*[c.get_full_completion() for c in cs]), 60)
        assert all(str(c.task.request_status) == "FINISHED_ABORTED" for c in cs)
        assert not eng.workers[0].is_alive() and eng.workers[0].process.exitcode == 1
        eng.shutdown()

        eng = AsyncEngineCore(worker_factory=_ctor_raises_factory, tokenizer=_Tok(), worker_mode="process")
        t0 = time.time()
        with pytest.raises(RuntimeError, match="failed to load"):
            await asyncio.wait_for(eng.init(worker_num=1, model_config=cfg, batch_size=2), 120)
        assert time.time() - t0 < 60
        eng.shutdown()

    asyncio.run(main())


def test_process_mode_prefix_states_travel_as_arena_row_addresses():
    """worker_mode="process" with state_arena_rows: every worker process owns an arena (here on the CPU, shared memory standing
    in for HIP IPC -- the GPU form is tests/test_engine_gpu.py) and ONLY row addresses cross the process boundary.  A prefill
    exported by one worker is cached in the engine process as (worker, row); hits queued with affinity are installed by the
    owner, hits queued without affinity are also installed by the OTHER process out of the owner's arena; all streams equal the
    uncached stream; evicting a prefix hands the row back to its worker exactly once, and only after the last hit on it has
    reported its copy complete."""
    from chirrup_amd.remote_arena import RemoteStateRef
    from chirrup_amd.state_cache import SimpleStateCache

    async def run(affinity):
        eng = AsyncEngineCore(worker_factory=_process_factory, tokenizer=_Tok(), worker_mode="process", state_arena_rows=4,
                              prefix_affinity=affinity)
        cfg = ModelLoadConfig(model_path="fake", vocab_path="fake", vocab_size=V, head_size=64)
        await asyncio.wait_for(eng.init(worker_num=2, model_config=cfg, batch_size=4), 120)
        cache = SimpleStateCache(max_size=2, arena=eng.state_arena)
        kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[])
        prompts = [list(range(1 + 3 * i, 25 + 3 * i)) for i in range(3)]
        for p in prompts[:2]:
            c = eng.completion("", prefill_tokens=list(p), max_tokens=4, cache_prefill=True, cache_prefill_padding=3, **kw)
            evs = [ev async for ev in c]
            hit = [e[1] for e in evs if e[0] == "cache_prefill"]
            assert len(hit) == 1 and isinstance(hit[0]["state"], RemoteStateRef)          # an address, no tensor
            assert [e[1] for e in evs if e[0] == "token"] == expected_stream(p, 4)
            cache.cache(hit[0]["prefilled_tokens"], hit[0]["state"])
        assert len(cache) == 2 and eng.state_arena.freed == []
        # hits: many at once, so that with the shared queue both processes pull some of them
        for rounds in range(3):
            cs = []
            for p in prompts[:2] * 3:
                rest, state, n_hit = cache.check(list(p))
                assert isinstance(state, RemoteStateRef) and n_hit == len(p) - 3
                cs.append((p, eng.completion("", prefill_tokens=rest, state=state, max_tokens=5, **kw)))
            outs = await asyncio.wait_for(asyncio.gather(*[c.get_full_completion() for _, c in cs]), 60)
            for (p, _), text in zip(cs, outs):
                assert text == "".join(f"<{t}>" for t in expected_stream(p, 5))
        inst = dict(eng._router.installs)
        assert inst["local"] + inst["peer"] == 18
        # a third prefix evicts the least recently used one: its row goes back to its worker (exactly one free message)
        c = eng.completion("", prefill_tokens=list(prompts[2]), max_tokens=2, cache_prefill=True, cache_prefill_padding=3, **kw)
        evs = [ev async for ev in c]
        hit = [e[1] for e in evs if e[0] == "cache_prefill"][0]
        cache.cache(hit["prefilled_tokens"], hit["state"])
        assert len(cache) == 2 and len(eng.state_arena.freed) == 1
        # ... and both arenas can take new exports for ever: 12 more prefixes through a 2-entry cache over 2 x 4 rows
        for i in range(12):
            p = list(range(2 + i, 30 + i))
            c = eng.completion("", prefill_tokens=list(p), max_tokens=2, cache_prefill=True, cache_prefill_padding=3, **kw)
            evs = [ev async for ev in c]
            hit = [e[1] for e in evs if e[0] == "cache_prefill"][0]
            assert isinstance(hit["state"], RemoteStateRef)                               # never the host-tensor fallback of a full arena
            cache.cache(hit["prefilled_tokens"], hit["state"])
        assert len(eng.state_arena.freed) == 13
        eng.shutdown()
        return inst

    with_affinity = asyncio.run(run(True))
    assert with_affinity["peer"] == 0 or with_affinity["local"] > 0      # the owner pulls its own hits first
    without = asyncio.run(run(False))
    assert without["peer"] > 0                                            # the other process installed some through the owner's arena


def test_a_dead_workers_arena_rows_are_dropped_and_never_opened():
    """Round-3 advisor finding (engine_core.py:128): when a worker process ends, the rows of ITS arena must die with it in the
    engine's books.  Two worker processes with arenas; the owner of a cached prefix is killed.  Afterwards: a cache lookup of
    that prefix misses (the entry is dropped, the request prefills again on the survivor and streams the right tokens); a hit
    that was handed out BEFORE the death and is submitted after it completes as aborted at once (it holds only the tokens
    behind the lost prefix) instead of waiting in a dead process's queue; the survivor keeps serving."""
    import time

    from chirrup_amd.remote_arena import RemoteStateRef
    from chirrup_amd.state_cache import SimpleStateCache

    async def main():
        eng = AsyncEngineCore(worker_factory=_process_factory, tokenizer=_Tok(), worker_mode="process", state_arena_rows=4)
        cfg = ModelLoadConfig(model_path="fake", vocab_path="fake", vocab_size=V, head_size=64)
        await asyncio.wait_for(eng.init(worker_num=2, model_config=cfg, batch_size=4), 120)
        cache = SimpleStateCache(max_size=4, arena=eng.state_arena)
        kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[])
        p = list(range(1, 25))
        c = eng.completion("", prefill_tokens=list(p), max_tokens=4, cache_prefill=True, cache_prefill_padding=3, **kw)
        evs = [ev async for ev in c]
        hit = [e[1] for e in evs if e[0] == "cache_prefill"][0]
        assert isinstance(hit["state"], RemoteStateRef)
        owner = hit["state"].worker_id
        cache.cache(hit["prefilled_tokens"], hit["state"])
        rest, early_ref, n_hit = cache.check(list(p))              # a hit handed out while the owner is alive ...
        assert isinstance(early_ref, RemoteStateRef) and n_hit == len(p) - 3
        victim = [w for w in eng.workers if w.worker_id == owner][0]
        victim.process.kill()
        t_end = time.time() + 30
        while not eng.state_arena.is_dead(owner) and time.time() < t_end:
            await asyncio.sleep(0.05)
        assert eng.state_arena.is_dead(owner)
        c = eng.completion("", prefill_tokens=rest, state=early_ref, max_tokens=4, **kw)   # ... submitted after its death
        await asyncio.wait_for(c.get_full_completion(), 30)
        assert str(c.task.request_status) == "FINISHED_ABORTED"
        rest2, state2, n_hit2 = cache.check(list(p))               # the lookup now misses and forgets the prefix
        assert state2 is None and n_hit2 == 0 and rest2 == p and len(cache) == 0
        c = eng.completion("", prefill_tokens=list(p), max_tokens=5, **kw)
        assert await asyncio.wait_for(c.get_full_completion(), 60) == "".join(f"<{t}>" for t in expected_stream(p, 5))
        eng.shutdown()

    asyncio.run(main())


def test_pending_exports_post_a_row_address_only_behind_its_copies():
    """Round-3 advisor finding (engine_process.py:64): the ("cache_prefill", row address) message of an export into a device
    arena must not leave the worker before the export's copies have completed.  Fake events stand in for the HIP events."""
    from chirrup_amd.engine_process import PendingExports, ResultSink

    class Ev:
        def __init__(self):
            self.done, self.synced = False, False

        def query(self):
            return self.done

        def synchronize(self):
            self.done = self.synced = True

    class Arena:
        def __init__(self):
            self.ev = Ev()

        def export_event(self, row):
            return self.ev

        def adopt(self, ref):
            return ref.row

    class Ref:
        def __init__(self, arena, row):
            self.arena, self.row = arena, row

    q = queue.Queue()
    pend, arena = PendingExports(q), Arena()
    sink = ResultSink(q, "t1", worker_id="worker_0", arena=arena, exports=pend)
    sink.put_nowait(("cache_prefill", {"state": Ref(arena, 3), "prefilled_tokens": (1, 2)}))
    assert q.empty()                                   # copies in flight: nothing has left
    pend.poll()
    assert q.empty()
    sink.put_nowait(("token_generated", (5, "x")))     # other messages are not held back
    assert q.get_nowait() == ("t1", ("token_generated", (5, "x")))
    arena.ev.done = True
    pend.poll()
    tid, (kind, payload) = q.get_nowait()
    assert (tid, kind) == ("t1", "cache_prefill") and payload["state"] == {"__remote_row__": ("worker_0", 3)}
    # a request that completes while its export is in flight: the address goes out FIRST (waited for), then the completion
    arena.ev = Ev()
    sink.put_nowait(("cache_prefill", {"state": Ref(arena, 1), "prefilled_tokens": (1,)}))
    assert q.empty()
    from chirrup_amd.core_structure import Task
    t = Task(output_queue=None, task_event_queue=None, prompt_str="", prefill_tokens=[], state=None)
    sink.put_nowait(("task_completed", t))
    assert arena.ev.synced
    assert q.get_nowait()[1][0] == "cache_prefill" and q.get_nowait()[1][0] == "task_completed"
