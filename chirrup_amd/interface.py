"""Per-request handle of the engine façade (boundary B4).

Same surface as the reference's chirrup/interface.py:31-142: an async iterator over
("token", id, text[, logits]) / ("cache_prefill", {"state", "prefilled_tokens"}) events, lazily
submitting its Task on first iteration, with ``get_full_completion()`` and ``abort()``."""
import asyncio
import queue
from typing import Any, List, Optional

from .core_structure import DEFAULT_SAMPLING_CONFIG, DEFAULT_STOP_TOKENS, Task


class AsyncEngineCompletion:
    def __init__(self, prompt_str: str, prefill_tokens: List[int], state, task_queue: "queue.Queue[Task]", result_channel,
                 task_id: str, priority: int = 0, temperature: float = DEFAULT_SAMPLING_CONFIG["temperature"],
                 top_p: float = DEFAULT_SAMPLING_CONFIG["top_p"], top_k: int = DEFAULT_SAMPLING_CONFIG["top_k"],
                 presence_penalty: float = DEFAULT_SAMPLING_CONFIG["presence_penalty"],
                 frequency_penalty: float = DEFAULT_SAMPLING_CONFIG["frequency_penalty"],
                 penalty_decay: float = DEFAULT_SAMPLING_CONFIG["penalty_decay"],
                 stop_tokens: Optional[List[int]] = DEFAULT_STOP_TOKENS, forbidden_tokens: Optional[List[int]] = None,
                 max_tokens: Optional[int] = DEFAULT_SAMPLING_CONFIG["max_tokens"], cache_prefill: bool = False,
                 cache_prefill_padding: int = 0, return_logits: bool = False, task_event_queue=None):
        self.task_id = task_id
        # the worker polls it for ("abort", None); a process-mode engine passes a channel that reaches the worker's process
        self.task_event_queue = task_event_queue if task_event_queue is not None else queue.Queue()
        self._result_queue: asyncio.Queue = result_channel.queue
        self.task = Task(output_queue=result_channel, task_event_queue=self.task_event_queue, prompt_str=prompt_str,
                         prefill_tokens=prefill_tokens, state=state, task_id=task_id, priority=priority,
                         temperature=temperature, top_p=top_p, top_k=top_k, presence_penalty=presence_penalty,
                         frequency_penalty=frequency_penalty, penalty_decay=penalty_decay, max_tokens=max_tokens,
                         stop_tokens=stop_tokens, forbidden_tokens=list(forbidden_tokens or []),
                         cache_prefill=cache_prefill, cache_prefill_padding=cache_prefill_padding,
                         return_logits=return_logits)
        self._task_queue = task_queue
        self._submitted = False
        self.is_finished = False

    def start(self):
        self._submitted = True
        self._task_queue.put_nowait(self.task)

    def __aiter__(self):
        if not self._submitted:
            self.start()
        return self

    async def __anext__(self):
        if self.is_finished:
            raise RuntimeError("Already finished")
        while True:
            msg: Any = await self._result_queue.get()
            if not (isinstance(msg, tuple) and len(msg) == 2):
                continue
            kind, payload = msg
            if kind == "token_generated":
                return ("token", *payload)
            if kind == "cache_prefill":
                return ("cache_prefill", payload)
            if kind == "task_completed":
                self.is_finished = True
                self.task = payload
                raise StopAsyncIteration

    def get_full_completion(self) -> "asyncio.Task[str]":
        async def collect() -> str:
            parts = []
            async for ev in self:
                if ev[0] == "token":
                    parts.append(ev[2])
            return "".join(parts)

        return asyncio.create_task(collect())

    def abort(self):
        self.task_event_queue.put_nowait(("abort", None))
