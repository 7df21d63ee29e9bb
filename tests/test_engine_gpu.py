"""The engine façade on the real path: checkpoint file -> loader -> worker thread on the GPU -> tokenizer ->
streamed completion (boundary B4, chirrup/engine_core.py + chirrup/interface.py)."""
import asyncio
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_engine_from_checkpoint_file_matches_direct_decode(tmp_path):
    from chirrup_amd.core_structure import ModelLoadConfig
    from chirrup_amd.engine_core import AsyncEngineCore
    from chirrup_amd.rwkv7 import RWKV_x070
    from chirrup_amd.tokenizer import TRIE_TOKENIZER

    d = np.load(os.path.join(G, "model_L2_C128.npz"))
    zd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
    ckpt = os.path.join(tmp_path, "tiny.pth")
    torch.save(zd, ckpt)
    vocab = os.path.join(G, "mini_vocab.txt")
    tok = TRIE_TOKENIZER(vocab)
    prompts = ["hello world", "abc abc abc abc abc abc abc abc", "the quick brown fox"]
    n_new = 12

    ref = RWKV_x070(types.SimpleNamespace(vocab_size=320, head_size=64, MODEL_NAME=ckpt[:-4]), device="cuda:0")   # loads the file itself

    def solo(text):
        ids = tok.encode(text)
        st = ref.generate_zero_state(1)
        lg = ref.forward_seq_batch_seperate([ids], st)
        out, safe = [], 0                    # safe = ids before the first step whose top-2 margin is within fp16 noise
        for i in range(n_new):
            top2 = torch.topk(lg.float(), 2, dim=-1).values[0]
            if float(top2[0] - top2[1]) >= 0.02 and safe == i:
                safe = i + 1
            t = int(lg.float().argmax(-1))
            out.append(t)
            lg = ref.forward_seq_batch_seperate([[t]], st)
        return out, safe

    async def main():
        eng = AsyncEngineCore()
        cfg = ModelLoadConfig(model_path=ckpt, vocab_path=vocab, vocab_size=320, head_size=64)
        await eng.init(worker_num=1, model_config=cfg, batch_size=4)
        kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[], max_tokens=n_new)
        cs = [eng.completion(p, **kw) for p in prompts]
        streams = []
        for c in cs:
            streams.append([(ev[1], ev[2]) async for ev in c if ev[0] == "token"])
        eng.shutdown()
        return streams

    streams = asyncio.run(main())
    checked = 0
    for p, got in zip(prompts, streams):
        want, safe = solo(p)
        checked += safe
        ids = [g[0] for g in got]
        assert len(ids) == n_new
        assert ids[:safe] == want[:safe], (p, safe)     # ids are defined where the arg-max is clear of fp16 noise
        def text_of(i):                      # the tiny model's vocab (320) is larger than the vocabulary file (164 ids)
            try:
                return tok.decode([i], utf8_errors="ignore")
            except KeyError:
                return ""

        assert [g[1] for g in got] == [text_of(i) for i in ids]
    assert checked >= 3          # the comparison must not be vacuous


def test_engine_process_mode_on_the_gpu_matches_thread_mode(tmp_path):
    """worker_mode="process" with the REAL worker: a spawned process loads the checkpoint onto the GPU, pulls requests from the
    shared queue and streams tokens back; greedy ids equal the thread-mode engine's on the same prompts, an abort reaches the
    worker process, and a prefix state exported by it comes back as host tensors that a later request continues from."""
    from chirrup_amd.core_structure import ModelLoadConfig
    from chirrup_amd.engine_core import AsyncEngineCore

    d = np.load(os.path.join(G, "model_L2_C128.npz"))
    zd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
    ckpt = os.path.join(tmp_path, "tiny.pth")
    torch.save(zd, ckpt)
    vocab = os.path.join(G, "mini_vocab.txt")
    prompts = ["hello world", "abc abc abc abc abc abc abc abc", "the quick brown fox", "a b c d e f g"]
    kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[])

    async def run(mode):
        eng = AsyncEngineCore(worker_mode=mode)
        cfg = ModelLoadConfig(model_path=ckpt, vocab_path=vocab, vocab_size=320, head_size=64)
        await asyncio.wait_for(eng.init(worker_num=1, model_config=cfg, batch_size=4), 300)
        cs = [eng.completion(p, max_tokens=10, **kw) for p in prompts]
        ids = []
        for c in cs:
            ids.append([ev[1] async for ev in c if ev[0] == "token"])
        extra = {}
        if mode == "process":
            assert eng.workers[0].process.is_alive() and eng.workers[0].process.pid != os.getpid()
            c = eng.completion(prompts[0], max_tokens=10 ** 6, **kw)
            n = 0
            async for ev in c:
                n += 1
                if n == 4:
                    c.abort()
            assert 4 <= n < 2000 and str(c.task.request_status) == "FINISHED_ABORTED"
            toks = eng.tokenizer.encode(prompts[1])
            c = eng.completion("", prefill_tokens=list(toks), max_tokens=6, cache_prefill=True, cache_prefill_padding=2, **kw)
            evs = [ev async for ev in c]
            cache = [e[1] for e in evs if e[0] == "cache_prefill"]
            assert len(cache) == 1 and all(t.device.type == "cpu" for t in cache[0]["state"])
            seen = len(cache[0]["prefilled_tokens"])
            c2 = eng.completion("", prefill_tokens=toks[seen:], state=cache[0]["state"], max_tokens=6, **kw)
            extra = {"first": [e[1] for e in evs if e[0] == "token"], "resumed": [ev[1] async for ev in c2 if ev[0] == "token"]}
        eng.shutdown()
        return ids, extra

    ids_t, _ = asyncio.run(run("thread"))
    ids_p, extra = asyncio.run(run("process"))
    assert all(len(x) == 10 for x in ids_p)
    assert ids_p == ids_t
    assert extra["resumed"] == extra["first"] == ids_p[1][:6]


def test_process_mode_prefix_states_never_leave_hbm(tmp_path):
    """VERDICT r2 item 4.  Two worker PROCESSES on cuda:0, each with its own HBM arena (state_arena_rows): a prefill exported by
    one of them reaches the engine process as a row address (no tensor), is cached there, and every hit -- queued without
    affinity ("avoid": queued for the process that does NOT own the row) -- is installed by a device copy out of the owner's arena
    (opened through a HIP IPC handle in the other process).  The resumed streams equal the uncached stream's tail, the engine
    counts local and cross-process installs, eviction hands rows back.  No state byte travels through host memory: the only
    objects that cross a process boundary are (worker id, row) pairs -- asserted on the events -- and the arenas' IPC handles."""
    from chirrup_amd.core_structure import ModelLoadConfig
    from chirrup_amd.engine_core import AsyncEngineCore
    from chirrup_amd.remote_arena import RemoteStateRef
    from chirrup_amd.state_cache import SimpleStateCache

    d = np.load(os.path.join(G, "model_L2_C128.npz"))
    zd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
    ckpt = os.path.join(tmp_path, "tiny.pth")
    torch.save(zd, ckpt)
    vocab = os.path.join(G, "mini_vocab.txt")
    kw = dict(temperature=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, stop_tokens=[])
    texts = ["abc abc abc abc abc abc abc abc", "the quick brown fox jumps over the lazy dog", "a b c d e f g h i j k l m n"]

    async def run(affinity):
        eng = AsyncEngineCore(worker_mode="process", state_arena_rows=4, prefix_affinity=affinity, gpu_ids=[0, 0])
        cfg = ModelLoadConfig(model_path=ckpt, vocab_path=vocab, vocab_size=320, head_size=64)
        await asyncio.wait_for(eng.init(worker_num=2, model_config=cfg, batch_size=4), 300)
        cache = SimpleStateCache(max_size=2, arena=eng.state_arena)
        first = {}
        for text in texts[:2]:
            toks = eng.tokenizer.encode(text)
            c = eng.completion("", prefill_tokens=list(toks), max_tokens=8, cache_prefill=True, cache_prefill_padding=2, **kw)
            evs = [ev async for ev in c]
            hit = [e[1] for e in evs if e[0] == "cache_prefill"]
            assert len(hit) == 1 and isinstance(hit[0]["state"], RemoteStateRef)          # an address: (worker id, row)
            cache.cache(hit[0]["prefilled_tokens"], hit[0]["state"])
            first[text] = [e[1] for e in evs if e[0] == "token"]
        for _ in range(3):
            cs = []
            for text in texts[:2] * 3:
                toks = eng.tokenizer.encode(text)
                rest, state, n_hit = cache.check(list(toks))
                assert isinstance(state, RemoteStateRef) and n_hit == len(toks) - 2
                cs.append((text, eng.completion("", prefill_tokens=rest, state=state, max_tokens=8, **kw)))
            for text, c in cs:
                assert [ev[1] async for ev in c if ev[0] == "token"] == first[text]      # resumed from the arena row: the same ids
        inst = dict(eng._router.installs)
        assert inst["local"] + inst["peer"] == 18
        toks = eng.tokenizer.encode(texts[2])
        c = eng.completion("", prefill_tokens=list(toks), max_tokens=2, cache_prefill=True, cache_prefill_padding=2, **kw)
        hit = [e[1] for e in [ev async for ev in c] if e[0] == "cache_prefill"][0]
        cache.cache(hit["prefilled_tokens"], hit["state"])
        assert len(eng.state_arena.freed) == 1                                           # the evicted prefix's row went back to its worker
        eng.shutdown()
        return inst

    # the worker processes inherit this process's stderr: collect it at the descriptor level to read what THEY print at exit
    import sys
    import tempfile
    import threading

    sys.stderr.flush()
    saved, log = os.dup(2), tempfile.TemporaryFile()
    os.dup2(log.fileno(), 2)
    try:
        a = asyncio.run(run(True))
        b = asyncio.run(run("avoid"))
    finally:
        sys.stderr.flush()
        os.dup2(saved, 2)
        os.close(saved)
    log.seek(0)
    child_err = log.read().decode(errors="replace")
    sys.stderr.write(child_err)
    print("installs with affinity", a, "queued for the non-owner", b)
    assert a["local"] > 0
    assert b["peer"] > 0                    # another process copied rows out of the owner's arena through its IPC handle
    # the consumers drop their views of a peer's arena BEFORE its owner ends (two-phase shutdown,
    # engine_process.worker_process_main) -- torch's IPC bookkeeping has nothing to complain about
    assert "Producer process has been terminated" not in child_err, child_err[-2000:]
    # ... and the engine leaves no thread behind in THIS process (a leftover busy thread can hold the GIL for a whole 5-ms switch
    # interval inside someone else's timed region: the most likely cause of round 3's 486-us mm8 timing)
    left = [t.name for t in threading.enumerate() if t.name.startswith("chirrup:") and t.is_alive()]
    assert not left, left


def test_openai_routes_over_the_real_worker_on_the_gpu(tmp_path):
    """SURVEY 8 f/4 on hardware: checkpoint file -> worker PROCESS on the GPU with an HBM arena -> AsyncEngineCore ->
    chirrup_amd.web_service: a chat completion streamed and as one body, the second identical request served from the cached prefix
    state (an arena row address: the state never leaves HBM), both equal to the engine's own stream for the same tokens."""
    import json

    import httpx

    from chirrup_amd.core_structure import ModelLoadConfig
    from chirrup_amd.engine_core import AsyncEngineCore
    from chirrup_amd.remote_arena import RemoteStateRef
    from chirrup_amd.web_service import ChatMessage, chat_prompt, create_app

    d = np.load(os.path.join(G, "model_L2_C128.npz"))
    zd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
    ckpt = os.path.join(tmp_path, "tiny.pth")
    torch.save(zd, ckpt)
    vocab = os.path.join(G, "mini_vocab.txt")
    msgs = [{"role": "user", "content": "the quick brown fox jumps over the lazy dog"}]
    req = {"model": "rwkv-latest", "messages": msgs, "temperature": 0.0, "top_p": 0.0, "presence_penalty": 0.0, "frequency_penalty": 0.0,
           "penalty_decay": 1.0, "max_tokens": 8, "stop": []}

    async def main():
        eng = AsyncEngineCore(worker_mode="process", state_arena_rows=4)
        cfg = ModelLoadConfig(model_path=ckpt, vocab_path=vocab, vocab_size=320, head_size=64)
        await asyncio.wait_for(eng.init(worker_num=1, model_config=cfg, batch_size=4), 300)
        prompt, pad = chat_prompt([ChatMessage(**m) for m in msgs], req["model"])
        toks = [0] + eng.tokenizer.encode(prompt)
        kw = dict(temperature=0.0, top_p=0.0, frequency_penalty=0.0, presence_penalty=0.0, penalty_decay=1.0, max_tokens=8)
        c = eng.completion(prompt, prefill_tokens=list(toks), **kw)                 # the engine's own stream (default stop tokens, like the route)
        want_text = "".join([ev[2] async for ev in c if ev[0] == "token"])
        app = create_app(eng, state_cache_size=2)
        outs = []
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as cl:
            assert (await cl.get("/health")).json()["model_loaded"] is True
            for stream in (False, True, True):
                r = await cl.post("/v1/chat/completions", json=dict(req, stream=stream), timeout=120)
                assert r.status_code == 200
                if stream:
                    chunks = [json.loads(ln[6:]) for ln in r.text.split("\n\n") if ln.startswith("data: ") and ln != "data: [DONE]"]
                    outs.append("".join(ch["choices"][0]["delta"].get("content", "") + ch["choices"][0]["delta"].get("reasoning_content", "")
                                        for ch in chunks))
                else:
                    m = r.json()["choices"][0]["message"]
                    outs.append(m["content"] + m["reasoning_content"])
            cache = app.state.state_cache
            assert len(cache) == 1                                                   # cached once, by the first request; hits afterwards
            (key,) = cache.keys()
            assert list(key) == toks[:len(toks) - pad]
            _, st, n_hit = cache.check(list(toks))
            assert isinstance(st, RemoteStateRef) and n_hit == len(toks) - pad       # an arena row address, no tensor
            st.release()
        eng.shutdown()
        return want_text, outs

    want_text, outs = asyncio.run(main())
    # the `\n\n` rule of the splitter may end the visible answer early; whatever is shown is a prefix of the engine's text
    assert all(want_text.startswith(o) or o == want_text for o in outs), (want_text, outs)
    assert outs[0] == outs[1] == outs[2]                                             # uncached, streamed, and from the cached prefix: one answer
