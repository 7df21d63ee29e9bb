"""OpenAI-compatible surface over AsyncEngineCore (SURVEY.md section 8 f/4; reference: chirrup/web_service/app.py:128-330,
api_model.py:8-112, chirrup/utils/prompt_formatters.py, streaming_string_parser.py).  Thin by design: `GET /health`,
`GET /v1/models`, `POST /v1/chat/completions` (streamed as server-sent events, or one JSON body) -- the reference's request
fields, chat template (`User:/Assistant:` paragraphs, model suffixes `:thinking` / `:no-thinking`), `[0]` + encode,
prefix-state cache protocol (`check_and_wait_prefill` -> `completion(state=...)` -> `cache()` + `awake_hang_up_prefills` on the
("cache_prefill", ...) event; in process mode the states are arena row addresses and never leave HBM) and 10-s keep-alive comments.
Not here: the settings CLI and the batch translate / rollout routes (out of scope).

    app = create_app(engine, state_cache_size=50)       # engine: an initialised AsyncEngineCore
"""
import asyncio
import json
import re
import time
import uuid
from typing import Any, Dict, List, Optional, Tuple, Union

from fastapi import FastAPI, HTTPException
from fastapi.responses import StreamingResponse
from pydantic import BaseModel, Field

from .core_structure import DEFAULT_SAMPLING_CONFIG, DEFAULT_STOP_TOKENS
from .state_cache import SimpleStateCache

KEEP_ALIVE_S = 10.0
_SSE_HEADERS = {"Cache-Control": "no-cache", "Connection": "keep-alive", "X-Accel-Buffering": "no"}


class ChatMessage(BaseModel):
    role: str
    content: str
    reasoning_content: Optional[str] = None


class ChatCompletionRequest(BaseModel):                      # api_model.py:14-62: same fields, defaults and ranges
    model: str = "rwkv-latest"
    messages: List[ChatMessage]
    stream: bool = False
    temperature: float = Field(default=DEFAULT_SAMPLING_CONFIG["temperature"], ge=0.0, le=2.0)
    top_p: float = Field(default=DEFAULT_SAMPLING_CONFIG["top_p"], ge=0.0, le=1.0)
    presence_penalty: float = Field(default=DEFAULT_SAMPLING_CONFIG["presence_penalty"], ge=0, le=2.0)
    frequency_penalty: float = Field(default=DEFAULT_SAMPLING_CONFIG["frequency_penalty"], ge=0, le=2.0)
    penalty_decay: float = Field(default=DEFAULT_SAMPLING_CONFIG["penalty_decay"], ge=0.0, le=1.0)
    max_tokens: int = Field(default=DEFAULT_SAMPLING_CONFIG["max_tokens"], ge=1)
    stop: Optional[Union[str, List[str]]] = None
    pad_zero: bool = True
    use_state_cache: bool = True
    cache_prefill: bool = True


def chat_prompt(messages: List[ChatMessage], model: str) -> Tuple[str, int]:
    """(prompt, cache_prefill_padding) of app.py:150-161: one `Role: text` paragraph per message (blank lines inside a message
    collapsed), then the assistant cue of the model variant."""
    names = {"user": "User", "assistant": "Assistant", "system": "System"}
    parts = [f"{names.get(m.role, m.role)}: {re.sub(chr(10) + '+', chr(10), (m.content or '').strip())}" for m in messages]
    variant = model.split(":")
    cue, pad = ("Assistant:<think>", 3) if "thinking" in variant else (("Assistant:", 0) if "no-thinking" in variant else ("Assistant:<think>\n</think>", 7))
    return "\n\n".join(parts + [cue]), pad


class ThinkSplitter:
    """Splits the generated text into ("content" | "reasoning_content", piece) runs: `<think>` opens the reasoning part,
    `</think>` closes it, and a blank line in the content part ends the answer (later text is dropped) -- the rule set
    TRIE_THINK_NO_TRIGGER of streaming_string_parser.py:176-182, markers not echoed.  A marker split over several pieces is held
    back until it is complete or ruled out."""
    MARKS = {"content": (("<think>", "reasoning_content"), ("\n\n", "end")), "reasoning_content": (("</think>", "content"),), "end": ()}

    def __init__(self):
        self.state, self.held = "content", ""

    def feed(self, text: str) -> List[Tuple[str, str]]:
        out: List[Tuple[str, str]] = []
        buf = self.held + text
        self.held = ""
        while buf:
            marks = self.MARKS[self.state]
            hits = [(buf.find(m), m, nxt) for m, nxt in marks if m in buf]
            if hits:
                at, m, nxt = min(hits)
                if at:
                    out.append((buf[:at], self.state))
                if nxt == "end":
                    out.append((m, nxt))
                buf, self.state = buf[at + len(m):], nxt
                continue
            keep = max((k for m, _ in marks for k in range(1, len(m)) if buf.endswith(m[:k])), default=0)   # a marker may be arriving
            if len(buf) > keep:
                out.append((buf[:len(buf) - keep], self.state))
            self.held = buf[len(buf) - keep:] if keep else ""
            break
        merged: List[Tuple[str, str]] = []
        for piece, st in out:
            if merged and merged[-1][1] == st:
                merged[-1] = (merged[-1][0] + piece, st)
            else:
                merged.append((piece, st))
        return merged


def create_app(engine, state_cache: Optional[SimpleStateCache] = None, state_cache_size: int = 50, model_ids=None) -> FastAPI:
    """The FastAPI application over an INITIALISED engine.  state_cache: default a SimpleStateCache of state_cache_size prefixes
    (over engine.state_arena when the engine has one: prefix states stay in the workers' HBM); 0 disables caching."""
    app = FastAPI(title="RWKV OpenAI Compatible API", version="1.0.0")
    if state_cache is None and state_cache_size > 0:
        arena = getattr(engine, "state_arena", None)
        state_cache = SimpleStateCache(state_cache_size, arena=arena if (arena is not None and arena.capacity >= state_cache_size) else None)
    created = int(time.time())
    models = [{"id": i, "object": "model", "created": created, "owned_by": "chirrup"}
              for i in (model_ids or ("rwkv-latest", "rwkv-latest:thinking", "rwkv-latest:no-thinking"))]
    app.state.engine, app.state.state_cache = engine, state_cache

    @app.get("/health")
    async def health():
        return {"status": "healthy", "timestamp": int(time.time()), "model_loaded": bool(engine.is_initialized and not engine.is_shutdown)}

    @app.get("/v1/models")
    async def list_models():
        return {"object": "list", "data": models}

    async def events_of(completion, splitter, head):
        """(kind, text) pieces of a completion; the prefix-cache hand-over happens here (app.py:285-295).  head: the prompt tokens the
        request did NOT feed because it started from a cached state.  The worker reports the tokens fed SINCE admission
        (chirrup/worker.py:662 starts `prefilled_tokens` empty, as does chirrup_amd's), and the reference files the exported state under
        those alone (app.py:295) -- after a hit that is a suffix of the real prefix: the state is never found again and the requests
        waiting for the real prefix (check_and_wait_prefill) are never woken.  Here the key is the whole prefix."""
        async for ev in completion:
            if ev[0] == "token":
                for piece, st in splitter.feed(ev[2]):
                    if st in ("content", "reasoning_content"):
                        yield st, piece
            elif ev[0] == "cache_prefill" and state_cache is not None:
                key = tuple(head) + tuple(ev[1]["prefilled_tokens"])
                node = state_cache.cache(key, ev[1]["state"], return_trie_node=True)
                if node is not None:
                    await state_cache.awake_hang_up_prefills(key)

    @app.post("/v1/chat/completions")
    async def chat(request: ChatCompletionRequest):
        if not engine.is_initialized or engine.is_shutdown:
            raise HTTPException(status_code=503, detail="model not loaded")
        try:
            prompt, pad = chat_prompt(request.messages, request.model)
            tokens = ([0] if request.pad_zero else []) + engine.tokenizer.encode(prompt)
            stops: List[int] = []
            for word in ([request.stop] if isinstance(request.stop, str) else (request.stop or [])):
                stops.extend(engine.tokenizer.encode(word))
            rest, state, n_cached = tokens, None, 0
            if request.use_state_cache and state_cache is not None:
                rest, state, n_cached = await state_cache.check_and_wait_prefill(tokens, pad)
            completion = engine.completion(prompt_str=prompt, prefill_tokens=rest, state=state, temperature=request.temperature,
                                           top_p=request.top_p, max_tokens=request.max_tokens, presence_penalty=request.presence_penalty,
                                           frequency_penalty=request.frequency_penalty, penalty_decay=request.penalty_decay,
                                           stop_tokens=set(DEFAULT_STOP_TOKENS + stops),
                                           cache_prefill=state_cache is not None and request.cache_prefill, cache_prefill_padding=pad)
        except HTTPException:
            raise
        except Exception as e:                                  # noqa: BLE001 -- app.py:231
            raise HTTPException(status_code=500, detail=f"generation failed: {e}")
        splitter = ThinkSplitter()
        splitter.feed(prompt.split("\n\n")[-1])                 # the assistant cue sets the splitter's starting part (app.py:243)
        cid, created_at = f"chatcmpl-{uuid.uuid4().hex}", int(time.time())

        def chunk(delta: Dict[str, Any], finish=None) -> str:
            body = {"id": cid, "object": "chat.completion.chunk", "created": created_at, "model": request.model,
                    "choices": [{"index": 0, "delta": delta, "finish_reason": finish}]}
            return f"data: {json.dumps(body, ensure_ascii=False)}\n\n"

        async def produce(q: asyncio.Queue):
            try:
                content, reasoning = [], []
                async for st, piece in events_of(completion, splitter, tokens[:n_cached]):
                    (content if st == "content" else reasoning).append(piece)
                    if request.stream:
                        q.put_nowait(chunk({"content": piece} if st == "content" else {"content": "", "reasoning_content": piece}))
                if request.stream:
                    q.put_nowait(chunk({}, "stop"))
                    q.put_nowait("data: [DONE]\n\n")
                else:
                    n_out = len(completion.task.generated_tokens)
                    q.put_nowait(json.dumps({"id": cid, "object": "chat.completion", "created": created_at, "model": request.model,
                                             "choices": [{"index": 0, "finish_reason": "stop",
                                                          "message": {"role": "assistant", "content": "".join(content),
                                                                      "reasoning_content": "".join(reasoning)}}],
                                             "usage": {"prompt_tokens": len(tokens), "completion_tokens": n_out,
                                                       "total_tokens": len(tokens) + n_out}}, ensure_ascii=False))
            except Exception as e:                              # noqa: BLE001 -- reported in-band like the reference (app.py:297-301)
                err = json.dumps({"error": {"message": str(e), "type": "internal_error"}})
                q.put_nowait(f"data: {err}\n\ndata: [DONE]\n\n" if request.stream else err)
            q.put_nowait(None)

        async def body():
            q: asyncio.Queue = asyncio.Queue()
            task = asyncio.create_task(produce(q))
            try:
                while True:
                    try:
                        item = await asyncio.wait_for(q.get(), timeout=KEEP_ALIVE_S)
                    except asyncio.TimeoutError:
                        if request.stream:
                            yield ":\n\n"                         # SSE comment: keeps proxies from closing an idle stream (app.py:325-330)
                        continue
                    if item is None:
                        return
                    yield item
            finally:                                            # client gone or done: stop generating
                task.cancel()
                completion.abort()
                if state_cache is not None and request.use_state_cache and n_cached + pad != len(tokens):
                    # this request was the one prefilling its prefix: if it ended without exporting it (abort, failure), the requests
                    # parked on that prefix must not wait for ever (the reference leaves them waiting) -- they look again and prefill
                    await state_cache.awake_hang_up_prefills(tuple(tokens[:-pad]) if pad else tuple(tokens))

        return StreamingResponse(body(), media_type="text/event-stream" if request.stream else "application/json", headers=_SSE_HEADERS)

    return app


__all__ = ["create_app", "ChatCompletionRequest", "ChatMessage", "ThinkSplitter", "chat_prompt"]
