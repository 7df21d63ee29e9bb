"""The OpenAI-compatible surface (SURVEY.md section 8 f/4; reference chirrup/web_service/app.py:128-330) over the fake-backend
engine on CPU: routes, request defaults, chat template, `<think>` stream splitting, streamed and one-body answers, prefix-state
cache hand-over."""
import asyncio
import json
import os

import httpx
import pytest

from chirrup_amd.core_structure import ModelLoadConfig
from chirrup_amd.engine_core import AsyncEngineCore
from chirrup_amd.web_service import ChatCompletionRequest, ChatMessage, ThinkSplitter, chat_prompt, create_app
from test_engine_cpu import _Tok, _factory
from test_worker_cpu import V, expected_stream

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_think_splitter_matches_the_reference_parser_on_seeded_piece_streams():
    """tests/golden/web_service.json: outputs of the reference's StreamingStringParser (rule set TRIE_THINK_NO_TRIGGER) primed with
    each assistant cue, on texts cut into random pieces (tests/golden/make_golden.py::gen_web)."""
    cases = json.load(open(os.path.join(G, "web_service.json")))["splits"]
    assert len(cases) >= 60
    for c in cases:
        sp = ThinkSplitter()
        sp.feed(c["cue"])
        runs = [r for piece in c["pieces"] for r in sp.feed(piece)]
        assert "".join(t for t, st in runs if st == "content") == c["content"], c
        assert "".join(t for t, st in runs if st == "reasoning_content") == c["reasoning_content"], c


def test_chat_template_and_request_defaults():
    """chirrup/utils/prompt_formatters.py:8-45 (read as text: the module does not import on Python 3.10) and api_model.py:14-62."""
    msgs = [ChatMessage(role="system", content="Be brief.\n\n\nVery brief."), ChatMessage(role="user", content="  What is 2+2?  "),
            ChatMessage(role="assistant", content="4"), ChatMessage(role="tool", content="x"), ChatMessage(role="user", content="")]
    body = "System: Be brief.\nVery brief.\n\nUser: What is 2+2?\n\nAssistant: 4\n\ntool: x\n\nUser: "
    assert chat_prompt(msgs, "rwkv-latest") == (body + "\n\nAssistant:<think>\n</think>", 7)
    assert chat_prompt(msgs, "rwkv-latest:thinking") == (body + "\n\nAssistant:<think>", 3)
    assert chat_prompt(msgs, "rwkv-latest:no-thinking") == (body + "\n\nAssistant:", 0)
    r = ChatCompletionRequest(messages=[{"role": "user", "content": "hi"}])
    assert (r.model, r.stream, r.temperature, r.top_p, r.presence_penalty, r.frequency_penalty, r.penalty_decay, r.max_tokens) == \
        ("rwkv-latest", False, 1.0, 0.3, 0.5, 0.5, 0.996, 8192)
    assert r.pad_zero and r.use_state_cache and r.cache_prefill and r.stop is None
    with pytest.raises(Exception):
        ChatCompletionRequest(messages=[], temperature=3.0)


def test_routes_over_the_fake_backend_engine():
    async def main():
        eng = AsyncEngineCore(worker_factory=_factory, tokenizer=_Tok())
        cfg = ModelLoadConfig(model_path="fake", vocab_path="fake", vocab_size=V, head_size=64)
        await eng.init(worker_num=2, model_config=cfg, batch_size=4)
        app = create_app(eng, state_cache_size=8)
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as cl:
            h = (await cl.get("/health")).json()
            assert h["status"] == "healthy" and h["model_loaded"] is True
            m = (await cl.get("/v1/models")).json()
            assert m["object"] == "list" and [x["id"] for x in m["data"]] == ["rwkv-latest", "rwkv-latest:thinking", "rwkv-latest:no-thinking"]
            msgs = [{"role": "user", "content": "tell me a long story about slots and states"}]
            req = {"model": "rwkv-latest:no-thinking", "messages": msgs, "temperature": 0.0, "top_p": 0.0, "presence_penalty": 0.0,
                   "frequency_penalty": 0.0, "penalty_decay": 1.0, "max_tokens": 6, "stop": ["\x07"]}
            prompt, pad = chat_prompt([ChatMessage(**x) for x in msgs], req["model"])
            toks = [0] + _Tok().encode(prompt)
            want = "".join(f"<{t}>" for t in expected_stream(toks, 6))
            # one JSON body
            r = await cl.post("/v1/chat/completions", json=req)
            assert r.status_code == 200 and r.headers["content-type"].startswith("application/json")
            j = r.json()
            assert j["object"] == "chat.completion" and j["model"] == req["model"] and j["id"].startswith("chatcmpl-")
            assert j["choices"][0]["message"] == {"role": "assistant", "content": want, "reasoning_content": ""}
            assert j["choices"][0]["finish_reason"] == "stop"
            assert j["usage"] == {"prompt_tokens": len(toks), "completion_tokens": 6, "total_tokens": len(toks) + 6}
            # the worker exports the state `cache_prefill_padding` tokens before the prompt's end (chirrup/worker.py:407-434): with
            # padding 0 (:no-thinking) that is the whole prompt
            assert app.state.state_cache.keys() == [tuple(toks[:-1])]        # (the last token is the next forward's input)
            req2 = dict(req, model="rwkv-latest", stream=True)
            prompt2, pad2 = chat_prompt([ChatMessage(**x) for x in msgs], "rwkv-latest")
            toks2 = [0] + _Tok().encode(prompt2)
            want2 = "".join(f"<{t}>" for t in expected_stream(toks2, 6))
            for attempt in range(2):                            # the second one starts from the cached prefix state: same stream
                r = await cl.post("/v1/chat/completions", json=req2)
                assert r.status_code == 200 and r.headers["content-type"].startswith("text/event-stream")
                events = [ln[6:] for ln in r.text.split("\n\n") if ln.startswith("data: ")]
                assert events[-1] == "[DONE]"
                chunks = [json.loads(e) for e in events[:-1]]
                assert all(c["object"] == "chat.completion.chunk" and c["model"] == "rwkv-latest" for c in chunks)
                assert chunks[-1]["choices"][0] == {"index": 0, "delta": {}, "finish_reason": "stop"}
                assert "".join(c["choices"][0]["delta"].get("content", "") for c in chunks[:-1]) == want2
                assert len(app.state.state_cache) == 2          # ... and the default model's prefix was cached once, by the first attempt
            key = app.state.state_cache.keys()[-1]
            assert list(key) == toks2[:len(key)] and len(key) == len(toks2) - pad2      # everything but the 7-token assistant cue
            # thinking variant: the cue opens the reasoning part -- everything generated is reasoning until a `</think>`
            r = await cl.post("/v1/chat/completions", json=dict(req, model="rwkv-latest:thinking"))
            msg = r.json()["choices"][0]["message"]
            assert msg["content"] == "" and len(msg["reasoning_content"]) > 0
            # validation errors are 422 like any FastAPI app; an engine that is shut down answers 503
            assert (await cl.post("/v1/chat/completions", json={"messages": msgs, "temperature": 9})).status_code == 422
            eng.shutdown()
            assert (await cl.post("/v1/chat/completions", json=req)).status_code == 503
            assert (await cl.get("/health")).json()["model_loaded"] is False

    asyncio.run(main())
