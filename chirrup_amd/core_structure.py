"""Request / configuration types of the engine boundary (B4 of SURVEY.md section 8b).

Pure data classes; names, fields, defaults and enum values are those of the reference's
chirrup/core_structure.py (Task :92-169, ModelLoadConfig :182-232, RequestStatus :61-82,
FinishReason :40-55, defaults :13-36) because the worker protocol and the OpenAI surface are
written against them.  No arithmetic lives here.
"""
import enum
import queue
import uuid
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch

FINISH_REASON_STRINGS = ("stop", "length", "abort")
DEFAULT_STOP_TOKENS = [0, 261, 24281]
DEFAULT_SAMPLING_CONFIG: Dict[str, Any] = {
    "temperature": 1.0,
    "top_p": 0.3,
    "top_k": 0,
    "presence_penalty": 0.5,
    "frequency_penalty": 0.5,
    "penalty_decay": 0.996,
    "max_tokens": 8192,
}


class FinishReason(enum.IntEnum):
    STOP = 0
    LENGTH = 1
    ABORT = 2

    def __str__(self):
        return FINISH_REASON_STRINGS[self.value]


class RequestStatus(enum.IntEnum):
    WAITING = enum.auto()
    RUNNING = enum.auto()
    FINISHED = enum.auto()              # everything above this value is a terminal state
    FINISHED_STOPPED = enum.auto()
    FINISHED_LENGTH_CAPPED = enum.auto()
    FINISHED_ABORTED = enum.auto()

    def __str__(self):
        return self.name

    @staticmethod
    def is_finished(status: "RequestStatus") -> bool:
        return status > RequestStatus.FINISHED

    @staticmethod
    def get_finished_reason(status: "RequestStatus") -> Optional[FinishReason]:
        return {RequestStatus.FINISHED_STOPPED: FinishReason.STOP,
                RequestStatus.FINISHED_LENGTH_CAPPED: FinishReason.LENGTH,
                RequestStatus.FINISHED_ABORTED: FinishReason.ABORT}.get(status)


@dataclass
class Task:
    """One generation request.  ``output_queue`` receives ("token_generated", (id, text[, logits])),
    ("cache_prefill", {"state", "prefilled_tokens"}) and ("task_completed", task) tuples;
    ``task_event_queue`` carries ("abort", payload) from the front end."""

    output_queue: Any
    task_event_queue: queue.Queue
    prompt_str: str
    prefill_tokens: List[int]
    state: Optional[List[torch.Tensor]]
    task_id: Optional[str] = None
    priority: int = 0

    temperature: float = DEFAULT_SAMPLING_CONFIG["temperature"]
    top_p: float = DEFAULT_SAMPLING_CONFIG["top_p"]
    top_k: int = DEFAULT_SAMPLING_CONFIG["top_k"]
    presence_penalty: float = DEFAULT_SAMPLING_CONFIG["presence_penalty"]
    frequency_penalty: float = DEFAULT_SAMPLING_CONFIG["frequency_penalty"]
    penalty_decay: float = DEFAULT_SAMPLING_CONFIG["penalty_decay"]
    max_tokens: Optional[int] = DEFAULT_SAMPLING_CONFIG["max_tokens"]

    stop_tokens: List[int] = field(default_factory=lambda: DEFAULT_STOP_TOKENS)
    forbidden_tokens: List[int] = field(default_factory=list)

    cache_prefill: bool = False
    cache_prefill_padding: int = 0
    return_logits: bool = False

    event_list: List = field(init=False, default_factory=list)
    request_status: RequestStatus = field(init=False, default=RequestStatus.WAITING)
    generated_tokens: List[int] = field(init=False, default_factory=list)
    decoded_texts: List[str] = field(init=False, default_factory=list)

    def __post_init__(self):
        if self.task_id is None:
            self.task_id = str(uuid.uuid4())

    def is_finished(self) -> bool:
        return RequestStatus.is_finished(self.request_status)


@dataclass
class ModelLoadConfig:
    model_path: str
    vocab_path: str
    vocab_size: int
    head_size: int
    dtype: torch.dtype = torch.float16       # torch.int8 selects the mm8 (w8a16) channel-mix path
    att_dtype: torch.dtype = torch.float16   # (not in the reference) torch.int8: receptance / key / value / output and the head as mm8 too

    n_head: Optional[int] = field(default=None, init=False)
    n_embd: Optional[int] = field(default=None, init=False)
    n_layer: Optional[int] = field(default=None, init=False)

    @property
    def param_byte(self) -> int:
        return {torch.float16: 2, torch.float32: 4, torch.bfloat16: 2, torch.int8: 1}.get(self.dtype, 2)

    def load_params(self, n_head: int, n_embd: int, n_layer: int) -> None:
        self.n_head, self.n_embd, self.n_layer = n_head, n_embd, n_layer

    def get_state_size_mb(self) -> float:
        """Per-request state in MB, with the reference's formula (core_structure.py:210-232: it counts
        vocab_size elements for state[2])."""
        if self.n_layer is None or self.n_embd is None:
            raise ValueError("n_layer and n_embd must be set through load_params first")
        n = self.n_layer * 2 * self.n_embd + self.n_layer * (self.n_embd // self.head_size) * self.head_size ** 2
        return (n + self.vocab_size) * self.param_byte / (1024 * 1024)
