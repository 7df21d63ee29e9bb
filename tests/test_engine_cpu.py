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
