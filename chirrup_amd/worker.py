"""Continuous-batching worker for one GPU (row A10 of SURVEY.md section 8a; boundary B4).

Keeps the reference worker's protocol and scheduling semantics (chirrup/worker.py):
  * ``Worker(worker_id, gpu_id, model_config, task_queue, master_event_queue, worker_event_queue,
    batch_size).start()`` runs the loop in the calling thread until a {"type": "shutdown"} event;
  * slots are classified with ``StateCategory`` (same members / order, :86-92); decode rows and
    single-token prefill rows share one forward per iteration (:671-742), chunked prefill of at most
    ``max_forward_seq_len_per_forward`` tokens runs every ``decode_prefill_ratio`` iterations for at
    most ``max(batch_size/8, 1)`` sequences (:143, :744-776, :854-856);
  * messages: ("token_generated", (id, text[, logits])), ("task_completed", task),
    ("cache_prefill", {"state", "prefilled_tokens"}) on the task's output queue (:422-434, :498-501,
    :556-558), (worker_id, "worker_loaded" | "worker_performance", {...}) on the worker event queue.

What is different is the DEVICE DISPATCH, redesigned for a 288 GB MI355X:
  * the slot table never moves.  The reference sorts slots by category with physical swaps of three
    17 MB state blocks per swap (`_switch_batch`, :304-360) so that a category is a contiguous slice;
    here every forward takes an int32 ``slot_idx`` vector and the kernels index the pool directly
    (RWKV_x070.forward_slots).  ``min_swaps_to_target_fast`` is still exported (tested API).
  * penalties + greedy sampling of all decode rows are one kernel and one D2H copy of B ids
    (ops.penalize_argmax) instead of ~8 torch kernels and B ``.item()`` syncs (:719-740); rows with
    real sampling parameters go through samplers.sample_logits_rwkv_pip_compatible.
  * prefix states exported for the cache stay in HBM by default (device clones; `state_cache_device="cpu"`
    restores the reference's host copies) -- a cache hit is then a 17-33 MB device-to-device copy.
  * RUN-AHEAD: the sampled ids never have to reach the host before the next forward.  They are scattered into a
    per-slot device vector (``last_ids``) that the next forward gathers its decode inputs from, and the penalty
    state (occurrence += 1, presence) is updated on the device from the same ids.  ``step()`` therefore launches
    forward k and only then reads and handles the ids of forward k-1 (text decode, stop / length checks, messages,
    admission): the ~1 ms of per-token host bookkeeping runs under the ~9 ms forward instead of in series with it.
    A request that turns out to have finished at k-1 has one speculative row in forward k; that row's result is
    dropped (the slot is recycled afterwards, in stream order).  ``run_ahead=False`` handles each forward's ids in
    the same ``step()`` call, like the reference's loop; both modes produce the same token streams.
"""
import queue
import time
import types
from collections import deque
from enum import IntEnum, auto
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .core_structure import ModelLoadConfig, RequestStatus, Task
from .samplers import is_greedy_row, sample_logits_rwkv_pip_compatible


class StateCategory(IntEnum):
    FORWARD_ONE_DECODE = auto()
    FORWARD_ONE_PREFILL = auto()
    FORWARD_ONE_SUSPENDED = auto()
    FORWARD_SEQ = auto()
    FINISHED = auto()
    EMPTY = auto()


def min_swaps_to_target_fast(lst, elements):
    """Partition ``lst`` in place so that equal categories are contiguous in the order of
    ``elements``; returns (swaps, offsets) like chirrup/worker.py:43-78.  This worker does not need
    the swaps (it uses slot indices) but the function is part of the reference's tested surface."""
    swaps: List[Tuple[int, int]] = []
    offsets: List[Tuple[int, int]] = []
    lo = 0
    for cat in elements:
        here = [i for i in range(lo, len(lst)) if lst[i] == cat]
        hi = lo + len(here)
        offsets.append((lo, hi))
        inside = set(here)
        holes = [i for i in range(lo, hi) if i not in inside]
        for hole, src in zip(holes, [i for i in here if i >= hi]):
            swaps.append((hole, src))
            lst[hole], lst[src] = lst[src], lst[hole]
        lo = hi
    return swaps, offsets


def _empty_slot() -> dict:
    return {"task": None, "is_prefilling": None, "new_token": None, "next_input_token": None,
            "state_category": StateCategory.EMPTY, "prefilled_tokens": [], "prefill_cached": False,
            "feedback": False}      # feedback: the slot's next decode input is last_ids[slot] on the device


class Worker:
    def __init__(self, worker_id: str, gpu_id: List[int], model_config: ModelLoadConfig, task_queue: queue.Queue,
                 master_event_queue: queue.Queue, worker_event_queue: Optional[queue.Queue], batch_size: int = 32,
                 model=None, tokenizer=None, penalize_argmax=None, state_cache_device=None, run_ahead: bool = True,
                 state_arena=None, state_arena_rows: int = 0, peer_arenas=None):
        self.worker_id, self.gpu_id, self.model_config = worker_id, gpu_id, model_config
        self.task_queue, self.master_event_queue, self.worker_event_queue = task_queue, master_event_queue, worker_event_queue
        self.real_state_size = batch_size
        self.max_batch_size = batch_size - 1          # the reference keeps one scratch slot; kept for parity of capacity
        self.max_prefill_count = max(int(batch_size * 0.125), 1)
        self.state_slot: Dict[int, dict] = {i: _empty_slot() for i in range(self.max_batch_size)}
        self.model, self.tokenizer = model, tokenizer   # may be injected (tests use a fake backend)
        self._slot_cache: Dict[tuple, torch.Tensor] = {}
        self._max_mem_gb = None
        self._penalize_argmax = penalize_argmax if penalize_argmax is not None else ops.penalize_argmax
        self._sample_topp = ops.sample_topp if penalize_argmax is None else None     # fake backends use the torch sampler
        self._commit_kernel = ops.commit_sampled if penalize_argmax is None else None  # ... and torch ops for the commit
        self.batch_state = None
        # where exported prefix states live: None = the pool's own device (HBM-resident prefix cache:
        # 17-33 MB device-to-device copies instead of two PCIe transfers per cache hit); "cpu" = the
        # reference's behaviour (worker.py:427-429)
        self.state_cache_device = state_cache_device
        # an HbmStateArena on this worker's device: prefix states are exported straight into a free arena row (one copy)
        # and travel as ArenaRef handles; without it they are device clones (or host copies, state_cache_device="cpu")
        self.state_arena = state_arena
        # process mode: build an arena of this many rows on this worker's own device (after the model has loaded) and publish it
        # to the other worker processes (remote_arena.PeerArenas): prefix states then never leave HBM
        self.state_arena_rows, self.peer_arenas = int(state_arena_rows), peer_arenas
        self._pending_installs = []                   # (event, task id): copies out of ANOTHER worker's arena still in flight
        self.run_ahead = run_ahead
        self.on_fatal = None                          # callable(worker, exception), set by the engine
        self._inflight = None                         # the forward whose sampled ids the host has not handled yet
        self.no_penalty_token_ids = {33, 10, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58}
        self.min_forward_seq_len = 10
        self.max_forward_seq_len_per_forward = 100
        self.seq_forward_count_down = 0
        self.decode_prefill_ratio = 5
        # NOT the reference's policy (off by default): while fewer than half of the slots are decoding, run a prefill chunk in
        # EVERY iteration instead of every decode_prefill_ratio-th -- a batch that arrives at once reaches full width after
        # n_chunks iterations instead of 5 x n_chunks (bench.py --serving-fill-first)
        self.prefill_when_underfilled = False
        self.shutdown_flag = False
        self.loop_time_recorder = deque(maxlen=10)
        self.iterations = 0

    # ------------------------------------------------------------------ set-up
    def _load_model(self):
        from .rwkv7 import RWKV_x070, model_args
        from .tokenizer import TRIE_TOKENIZER

        if self.model is None:
            self.model = RWKV_x070(model_args(self.model_config.model_path, self.model_config.vocab_size,
                                              self.model_config.head_size),
                                   ffn_dtype=torch.int8 if self.model_config.dtype == torch.int8 else torch.float16,
                                   att_dtype=torch.int8 if getattr(self.model_config, "att_dtype", None) == torch.int8 else torch.float16)
        if self.tokenizer is None:
            self.tokenizer = TRIE_TOKENIZER(self.model_config.vocab_path)
        self._post({"status": "success", "worker_id": self.worker_id, "gpu_id": self.gpu_id,
                    "model_path": self.model_config.model_path}, "worker_loaded")

    def _post(self, payload, kind):
        if self.worker_event_queue is not None:
            try:
                self.worker_event_queue.put_nowait((self.worker_id, kind, payload))
            except queue.Full:      # performance messages are droppable (engine_core.py:42-47)
                pass

    def _init_worker(self):
        if self.gpu_id and torch.cuda.is_available():
            torch.cuda.set_device(self.gpu_id[0])
        self._load_model()
        n, V = self.real_state_size, self.model_config.vocab_size
        self.batch_state = self.model.generate_zero_state(n)
        dev = self.batch_state[0].device
        self.device = dev
        f32 = dict(dtype=torch.float32, device=dev)
        self.occurrence = torch.zeros((n, V), **f32)
        self.alpha_presence_vector = torch.zeros((n, V), **f32)
        self.temperature_tensor = torch.zeros((n, 1), dtype=torch.float16, device=dev)
        self.top_p_tensor = torch.zeros((n, 1), dtype=torch.float16, device=dev)
        self.top_k_tensor = torch.zeros((n, 1), dtype=torch.int32, device=dev)
        self.frequency_penalty_tensor = torch.zeros((n, 1), dtype=torch.float16, device=dev)
        self.penalty_decay_tensor = torch.zeros((n, 1), dtype=torch.float16, device=dev)
        self.presence_penalty_tensor = torch.zeros((n, 1), **f32)
        self._greedy = [True] * n
        # device-side feedback of the sampled ids (run-ahead): next decode inputs and the penalty bookkeeping
        self.last_ids = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.penalty_weight = torch.ones((V,), **f32)           # occurrence increment per token id (worker.py:531)
        self.penalty_weight[[t for t in self.no_penalty_token_ids if t < V]] = 0.0
        self._ids_host = ([torch.zeros((n + 1,), dtype=torch.int32).pin_memory() for _ in range(2)]     # (+ 1: the launch-status word)
                          if dev.type == "cuda" else None)
        self._launches = 0
        # the ids each slot has sampled (ops.PenaltyLists): the penalty step then touches those entries of the dense tables only
        self._pen_lists = (ops.PenaltyLists(n, V, dev) if (self._commit_kernel is not None and dev.type == "cuda" and V % 32 == 0
                                                           and self._penalize_argmax is ops.penalize_argmax) else None)
        if self.state_arena is None and self.state_arena_rows > 0:
            from .state_cache import HbmStateArena

            s0, s1 = self.batch_state[0], self.batch_state[1]
            self.state_arena = HbmStateArena(s0.shape[0], s0.shape[3], self.state_arena_rows, dev, dtype=s0.dtype,
                                             wkv_shape=s1.shape[2:], wkv_dtype=s1.dtype)
            if hasattr(self.task_queue, "arena"):
                self.task_queue.arena = self.state_arena          # exports into it leave this process as row addresses
            if self.peer_arenas is not None:
                self.peer_arenas.publish(self.state_arena)
        # decode-step HIP graphs per batch bucket (captured lazily); slot n-1 is the parking slot
        self._graphs = {}
        self.use_graph = bool(getattr(self.model, "fused", False)) and self.device.type == "cuda"

    # ------------------------------------------------------------------ per-slot bookkeeping (host only)
    def _process_events(self) -> bool:
        while True:
            try:
                ev = self.master_event_queue.get_nowait()
            except queue.Empty:
                return False
            if ev.get("type") == "shutdown":
                self.shutdown_flag = True
                return True
            if ev.get("type") == "arena_free" and self.state_arena is not None:
                self.state_arena.release(int(ev["row"]))       # the engine's cache dropped the prefix: the row may be reused
            if ev.get("type") == "peer_dead" and self.peer_arenas is not None:
                self.peer_arenas.forget(ev["worker"])          # another worker process has ended: its arena is gone

    @staticmethod
    def _is_task_aborted(td) -> bool:
        q = td["task"].task_event_queue
        if q.empty():                       # the common case, without the cost of raising queue.Empty
            return False
        try:
            kind, _ = q.get_nowait()
            return kind == "abort"
        except queue.Empty:
            return False

    def _export_state(self, slot: int):
        """[state0[:, :, [s]], state1[:, [s]], state2[[s]]] on the CPU (worker.py:426-430)."""
        s0, s1, s2 = self.batch_state
        if self.state_arena is not None and self.state_arena.free_rows > 0:
            return self.state_arena.export_slot(self.batch_state, slot)      # slot -> arena row, one strided copy per tensor
        parts = [s0[:, :, [slot], :], s1[:, [slot], :, :], s2[[slot]]]      # advanced indexing: fresh copies
        if self.state_cache_device is None:
            return parts
        return [p.to(self.state_cache_device, non_blocking=True) for p in parts]

    def _maybe_cache_prefill(self, td, slot: int):
        t: Task = td["task"]
        if t.cache_prefill and not td["prefill_cached"] and len(t.prefill_tokens) == max(t.cache_prefill_padding - 1, 0):
            t.output_queue.put_nowait(("cache_prefill", {"state": self._export_state(slot),
                                                         "prefilled_tokens": tuple(td["prefilled_tokens"])}))
            td["prefill_cached"] = True
            return True
        return False

    def _handle_forward_seq(self, td, slot: int):
        t: Task = td["task"]
        if self._maybe_cache_prefill(td, slot):
            td["state_category"] = StateCategory.FORWARD_ONE_PREFILL
        if len(t.prefill_tokens) == 0:
            td["state_category"], td["is_prefilling"] = StateCategory.FORWARD_ONE_DECODE, False
        elif len(t.prefill_tokens) < self.min_forward_seq_len:
            td["state_category"] = StateCategory.FORWARD_ONE_PREFILL

    def _handle_forward_one_prefill_phase(self, td, slot: int):
        t: Task = td["task"]
        td["prefilled_tokens"].append(td["next_input_token"])
        td["next_input_token"] = t.prefill_tokens.pop(0)
        if len(t.prefill_tokens) == 0:
            td["is_prefilling"], td["state_category"] = False, StateCategory.FORWARD_ONE_DECODE
        self._maybe_cache_prefill(td, slot)

    def _handle_forward_one_decode_phase(self, td, tok: int, raw_logits=None):
        """One sampled token of a live request (chirrup/worker.py:503-560): stop / length checks, text, message.
        The penalty state was already advanced on the device when the token was sampled."""
        t: Task = td["task"]
        td["new_token"] = tok
        if tok in t.stop_tokens:
            t.request_status = RequestStatus.FINISHED_STOPPED
            return
        try:
            text = self.tokenizer.decode([tok], utf8_errors="ignore")
        except KeyError:                 # an id the vocabulary file does not define (the model's vocab is padded)
            text = ""
        t.generated_tokens.append(tok)
        t.decoded_texts.append(text)
        if t.return_logits and raw_logits is not None:
            t.output_queue.put_nowait(("token_generated", (tok, text, raw_logits.detach().cpu())))
        else:
            t.output_queue.put_nowait(("token_generated", (tok, text)))
        if len(t.generated_tokens) >= t.max_tokens:
            t.request_status = RequestStatus.FINISHED_LENGTH_CAPPED
            return
        td["next_input_token"] = tok

    def _commit_sampled(self, ids: torch.Tensor, didx: torch.Tensor, status_out=None):
        """Device-side consequences of sampling `ids` for the slots `didx`: the next decode input and the
        repetition-penalty state (worker.py:527-535: occurrence += 1 except for the no-penalty ids, presence).
        status_out: the element behind the ids that receives the device's sticky launch-status word in the same launch."""
        if self._commit_kernel is not None:                # one launch instead of ~12 eager ones behind every decode step
            self._commit_kernel(ids, didx, self.last_ids, self.occurrence, self.penalty_weight, self.alpha_presence_vector,
                                self.presence_penalty_tensor, status_out=status_out, **({"lists": self._pen_lists} if self._pen_lists is not None else {}))
            return
        dl, il = didx.long(), ids.long()
        self.last_ids.index_copy_(0, dl, ids)
        self.occurrence.index_put_((dl, il), self.penalty_weight[il], accumulate=True)
        self.alpha_presence_vector[dl, il] = self.presence_penalty_tensor[dl, 0]

    def _handle_results(self, rec):
        """Host half of a forward: read its sampled ids and run the per-token bookkeeping.  Rows whose request
        is gone (finished or aborted while this forward was in flight) are dropped."""
        if rec is None:
            return
        if rec["event"] is not None:
            rec["event"].synchronize()
        n_rows = len(rec["rows"])
        host_ids = rec["ids"][: n_rows + (1 if rec.get("status") else 0)].tolist()        # ONE device->host copy for the whole batch
        if rec.get("status") and host_ids[n_rows] != 0:
            # the time-mix launch's bounded in-launch waits (include/chirrup_amd.h: rwkv7_tmix_gemms): a wait that gave up left the LoRA
            # outputs of a step at or before this one undefined -- never seen on a GPU this process has to itself.  The sticky status
            # word arrives behind every step's ids, so the step that gave up is fatal BEFORE any of its tokens is sent to a client
            # (round 3 looked at a word every replay had zeroed, every 256 iterations).
            raise RuntimeError(f"a time-mix launch gave up waiting for its own workgroups (status word {host_ids[n_rows]}): results are undefined")
        done = []
        for j, (slot, task) in enumerate(rec["rows"]):
            td = self.state_slot[slot]
            if td["task"] is not task or RequestStatus.is_finished(task.request_status):
                continue
            self._handle_forward_one_decode_phase(td, int(host_ids[j]), rec["raw"].get(j))
            if RequestStatus.is_finished(task.request_status):
                done.append(slot)
        self._process_accomplished_tasks(done)

    def _process_accomplished_tasks(self, slots):
        for s in slots:
            t = self.state_slot[s]["task"]
            t.output_queue.put_nowait(("task_completed", t))
            self.state_slot[s] = _empty_slot()

    def _fill_task_pool(self):
        prefills = 0
        for slot in range(self.max_batch_size):
            if prefills >= self.max_prefill_count:
                break
            td = self.state_slot[slot]
            if td["state_category"] != StateCategory.EMPTY:
                prefills += td["state_category"] == StateCategory.FORWARD_SEQ
                continue
            prefills += 1
            try:
                task: Task = self.task_queue.get_nowait()
            except queue.Empty:
                break
            self._install(task, slot)

    def _install(self, task: Task, slot: int):
        s0, s1, s2 = self.batch_state
        if task.state is None:
            s0[:, :, slot].zero_(), s1[:, slot].zero_(), s2[slot].zero_()
        elif hasattr(task.state, "install_into"):   # prefix-cache hit out of an HbmStateArena: row -> slot, one copy, unpin
            task.state.install_into(self.batch_state, slot)
            task.state.release()
        elif isinstance(task.state, dict) and "__remote_row__" in task.state:
            # process mode: the address of a row in a worker's arena -- this worker's own (one local copy) or another worker's,
            # read through its IPC handle (remote_arena.PeerArenas); the engine learns when the copy has COMPLETED
            wid, row = task.state["__remote_row__"]
            if self.peer_arenas is None:
                raise RuntimeError("a remote arena row arrived at a worker without peer arenas")
            try:
                self.peer_arenas.install(wid, int(row), self.batch_state, slot)
            except Exception as exc:              # noqa: BLE001 -- the owner of the row has ended (its IPC handle cannot be opened,
                # or was never published): that is the end of THIS request -- it holds only the tokens behind the lost prefix --
                # not of this worker and the requests it serves (round-3 advisor finding: the failure used to be fatal for the
                # stealer, one dead worker cascading through the survivors)
                import sys

                print(f"{self.worker_id}: prefix state ({wid}, {row}) is unreachable ({type(exc).__name__}: {exc}); request aborted", file=sys.stderr)
                self.task_queue.installed(task.task_id, wid != self.worker_id)
                task.request_status = RequestStatus.FINISHED_ABORTED
                task.output_queue.put_nowait(("task_completed", task))
                return
            peer = wid != self.worker_id
            if peer and s1.device.type == "cuda":
                ev = torch.cuda.Event()
                ev.record()
                self._pending_installs.append((ev, task.task_id))
            else:                                    # own row: later writers of the row queue behind this copy on this stream
                self.task_queue.installed(task.task_id, peer)
        else:                                    # prefix-cache hit: [L,2,1,C], [L,1,H,64,64], [1]
            s0[:, :, [slot], :] = task.state[0].to(s0.device, non_blocking=True)
            s1[:, [slot], :, :] = task.state[1].to(s1.device, non_blocking=True)
            s2[[slot]] = task.state[2].to(s2.device, non_blocking=True)
        self.occurrence[slot].zero_()
        self.alpha_presence_vector[slot].zero_()
        if getattr(self, "_pen_lists", None) is not None:
            self._pen_lists.reset(slot)
        self.temperature_tensor[slot, 0] = task.temperature if task.temperature > 0 else 1.0
        self.top_p_tensor[slot, 0] = task.top_p
        self.top_k_tensor[slot, 0] = task.top_k
        self.frequency_penalty_tensor[slot, 0] = task.frequency_penalty
        self.penalty_decay_tensor[slot, 0] = task.penalty_decay
        self.presence_penalty_tensor[slot, 0] = task.presence_penalty
        self._greedy[slot] = is_greedy_row(task.temperature, task.top_p, task.top_k)
        first = task.prefill_tokens.pop(0)
        if len(task.prefill_tokens) == 0:
            cat, prefilling = StateCategory.FORWARD_ONE_DECODE, False
        elif len(task.prefill_tokens) - max(task.cache_prefill_padding - 1, 0) < self.min_forward_seq_len:
            cat, prefilling = StateCategory.FORWARD_ONE_PREFILL, True
        else:
            cat, prefilling = StateCategory.FORWARD_SEQ, True
        td = _empty_slot()
        td.update(task=task, is_prefilling=prefilling, next_input_token=first, state_category=cat)
        self.state_slot[slot] = td

    def _organize_batch(self):
        """Slot ids per category, in slot order (the reference returns contiguous ranges after
        swapping; here the lists ARE the addressing)."""
        by_cat: Dict[StateCategory, List[int]] = {c: [] for c in StateCategory}
        for slot in range(self.max_batch_size):
            by_cat[self.state_slot[slot]["state_category"]].append(slot)
        return by_cat

    # ------------------------------------------------------------------ device dispatch
    def _slot_tensor(self, slots: List[int]) -> torch.Tensor:
        """Device copy of a list of slot / row ids, cached: torch.tensor(list, device=...) is a pageable host-to-device copy, i.e.
        the host blocks until everything already enqueued on the stream has run -- once per iteration that turned the run-ahead loop into
        a lock-step one (the GPU idled ~200 us per 7-ms iteration while the host enqueued the next).  The decode slot list is the same
        from one iteration to the next until a request arrives or leaves; the tensors are never written."""
        key = tuple(slots)
        t = self._slot_cache.get(key)
        if t is None:
            if len(self._slot_cache) >= 16:
                self._slot_cache.pop(next(iter(self._slot_cache)))
            t = self._slot_cache[key] = torch.tensor(slots, dtype=torch.int32, device=self.device)
        return t

    def _run_forward_one(self, decode_slots: List[int], prefill_slots: List[int]):
        """Enqueue one forward + sampling for these slots; returns the record `_handle_results` consumes."""
        slots = decode_slots + prefill_slots
        if not slots:
            return None
        idx = self._slot_tensor(slots)
        nd = len(decode_slots)
        # every small host->device tensor is created BEFORE the forward is enqueued: a pageable H2D copy is
        # synchronous and would otherwise stall the host behind the whole decode step
        sampled = [j for j, s in enumerate(decode_slots) if not self._greedy[s]]
        rows = u = None
        if sampled:
            rows = self._slot_tensor(sampled)
            u = torch.rand((len(sampled),), device=self.device, dtype=torch.float32)
        tokens = []
        for j, s in enumerate(slots):
            td = self.state_slot[s]
            tokens.append(-1 if (j < nd and td["feedback"]) else td["next_input_token"])    # -1: last_ids[slot]
            if j < nd:
                td["feedback"] = True
        if self.use_graph:
            out = self._graph_for(len(slots)).run(tokens, slots)
        else:
            tok = torch.tensor(tokens, dtype=torch.long, device=self.device).view(-1, 1)
            tok = torch.where(tok < 0, self.last_ids[idx.long()].long().view(-1, 1), tok)
            out = self.model.forward_slots(tok, self.batch_state, idx)
        self._launches += 1
        if nd == 0:
            return None
        logits = out[:nd]
        raw = {}
        for j, s in enumerate(decode_slots):
            t: Task = self.state_slot[s]["task"]
            if t.return_logits:
                raw[j] = logits[j].clone()
            for tok_id in t.forbidden_tokens:
                logits[j, tok_id] -= 1e10
        didx = idx[:nd]
        # penalties for every decode row + arg-max, one kernel; occurrence is decayed in place
        status_out = None
        if self._commit_kernel is not None and self.device.type == "cuda":
            buf = torch.empty((nd + 1,), dtype=torch.int32, device=self.device)     # the ids, and the launch-status word behind them
            ids, status_out = buf[:nd], buf[nd:]
            self._penalize_argmax(logits, self.occurrence, self.alpha_presence_vector, self.penalty_decay_tensor.view(-1),
                                  self.frequency_penalty_tensor.view(-1), didx, out=ids, **({"lists": self._pen_lists} if self._pen_lists is not None else {}))
        else:
            ids = self._penalize_argmax(logits, self.occurrence, self.alpha_presence_vector, self.penalty_decay_tensor.view(-1),
                                        self.frequency_penalty_tensor.view(-1), didx)
        if sampled:
            if self._sample_topp is not None and logits.shape[1] <= 65536:
                # sort-free top-p / top-k / temperature kernel, one workgroup per sampled row
                self._sample_topp(logits, rows, self.temperature_tensor.view(-1), self.top_p_tensor.view(-1),
                                  self.top_k_tensor.view(-1), u, ids, didx)
            else:
                lrows = rows.long()
                srows = didx.long()[lrows]
                ids[lrows] = sample_logits_rwkv_pip_compatible(logits[lrows], self.temperature_tensor[srows],
                                                               self.top_p_tensor[srows], self.top_k_tensor[srows]).to(torch.int32)
        self._commit_sampled(ids, didx, status_out)
        event = None
        if self._ids_host is not None:
            host = self._ids_host[self._launches & 1]
            if status_out is not None:
                host[:nd + 1].copy_(buf, non_blocking=True)
            else:
                host[:nd].copy_(ids, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            ids = host
        return {"rows": [(s, self.state_slot[s]["task"]) for s in decode_slots], "ids": ids, "event": event, "raw": raw,
                "status": status_out is not None and self._ids_host is not None}

    def _graph_for(self, n: int):
        """Smallest captured bucket >= n (buckets: powers of two up to the slot count)."""
        from .rwkv7 import SlotDecodeGraph

        b = 1
        while b < n:
            b *= 2
        b = min(b, self.max_batch_size)
        g = self._graphs.get(b)
        if g is None:
            g = self._graphs[b] = SlotDecodeGraph(self.model, self.batch_state, b, parking_slot=self.real_state_size - 1,
                                                  feedback=self.last_ids)
        return g

    def _run_forward_seq(self, seq_slots: List[int]):
        lens = [len(self.state_slot[s]["task"].prefill_tokens) - max(self.state_slot[s]["task"].cache_prefill_padding - 1, 0)
                for s in seq_slots]
        n_tok = min(self.max_forward_seq_len_per_forward, *lens)
        assert n_tok > 0
        batch = []
        for s in seq_slots:
            td = self.state_slot[s]
            t: Task = td["task"]
            chunk = [td["next_input_token"]] + t.prefill_tokens[: n_tok - 1]
            td["prefilled_tokens"].extend(chunk)
            t.prefill_tokens = t.prefill_tokens[n_tok - 1:]
            td["next_input_token"] = t.prefill_tokens.pop(0)
            batch.append(chunk)
        self.model.forward_slots(batch, self.batch_state, self._slot_tensor(seq_slots))   # logits discarded (:776)

    # ------------------------------------------------------------------ main loop
    def step(self) -> bool:
        """One loop iteration (chirrup/worker.py:793-884).  Returns False when idle.
        Order: (1) host-only slot bookkeeping that needs no sampled token (aborts, prompt cursors, admission),
        (2) enqueue this iteration's forward(s), (3) handle sampled ids -- those of the PREVIOUS iteration's
        forward when running ahead, else this one's."""
        t0 = time.perf_counter()
        poll_exports = getattr(self.task_queue, "poll_exports", None)
        if poll_exports is not None:                   # process mode: row addresses whose export copies have completed may leave now
            poll_exports()
        if self._pending_installs:                     # copies out of another worker's arena: report those that have completed
            still = []
            for ev, task_id in self._pending_installs:
                if ev.query():
                    self.task_queue.installed(task_id, True)
                else:
                    still.append((ev, task_id))
            self._pending_installs = still
        done = []
        for slot in range(self.max_batch_size):
            td = self.state_slot[slot]
            cat = td["state_category"]
            assert cat != StateCategory.FINISHED
            if cat == StateCategory.EMPTY:
                continue
            if self._is_task_aborted(td):
                td["task"].request_status = RequestStatus.FINISHED_ABORTED
                td["state_category"] = StateCategory.FINISHED
            elif cat == StateCategory.FORWARD_SEQ:
                self._handle_forward_seq(td, slot)
            elif cat == StateCategory.FORWARD_ONE_PREFILL:
                self._handle_forward_one_prefill_phase(td, slot)
            if RequestStatus.is_finished(td["task"].request_status):
                done.append(slot)
        self._process_accomplished_tasks(done)
        self._fill_task_pool()
        cats = self._organize_batch()
        dec, pre, seq = (cats[c] for c in (StateCategory.FORWARD_ONE_DECODE, StateCategory.FORWARD_ONE_PREFILL,
                                           StateCategory.FORWARD_SEQ))
        if not dec and not pre and not seq:
            if self._inflight is None:
                return False
            rec, self._inflight = self._inflight, None           # drain the pipeline
            self._handle_results(rec)
            return True
        rec = None
        if dec or pre:
            rec = self._run_forward_one(dec, pre)
            self.seq_forward_count_down -= 1
        else:
            self.seq_forward_count_down = 0
        if seq and (self.seq_forward_count_down < 1 or (self.prefill_when_underfilled and 2 * len(dec) < self.max_batch_size)):
            self._run_forward_seq(seq)
            self.seq_forward_count_down = max(1, self.decode_prefill_ratio)
        if self.run_ahead:
            rec, self._inflight = self._inflight, rec
        self._handle_results(rec)
        self.iterations += 1
        self.loop_time_recorder.append(time.perf_counter() - t0)
        if (self.iterations & 63) == 1 or self._max_mem_gb is None:      # (memory_stats() walks a dictionary of ~150 counters: 0.12 ms per call)
            self._max_mem_gb = (torch.cuda.max_memory_allocated() / 1024 ** 3) if torch.cuda.is_available() else 0.0
        self._post({"avg_loop_time": sum(self.loop_time_recorder) / len(self.loop_time_recorder),
                    "state_size": self.real_state_size,
                    "state_offset_details": {"decode_slots": dec, "one_prefill_slots": pre, "seq_prefill_slots": seq},
                    "task_details": {"decode_count": len(dec), "one_prefill_count": len(pre), "seq_prefill_count": len(seq)},
                    "max_allocated_memory_GB": self._max_mem_gb},
                   "worker_performance")
        return True

    def start(self):
        """The worker thread's entry point.  A failure anywhere in the loop must not strand the clients: it is
        reported on the worker event queue and on stderr, and every request this worker holds is completed as
        aborted before the thread ends (the reference's loop has no such path: a dead worker thread leaves its
        requests waiting forever)."""
        try:
            if self.batch_state is None:
                self._init_worker()
            while True:
                if self._process_events():
                    break
                if not self.step():
                    time.sleep(0.05)
        except BaseException as exc:             # noqa: BLE001 -- report, release the clients, then re-raise
            import traceback

            traceback.print_exc()
            self._post({"status": "failed", "worker_id": self.worker_id, "error": repr(exc)}, "worker_error")
            self._fail_all_requests()
            if self.on_fatal is not None:
                self.on_fatal(self, exc)         # the engine releases queued requests if no worker is left
            raise
        finally:
            self._cleanup()

    def _fail_all_requests(self):
        for td in list(getattr(self, "state_slot", {}).values()):
            t = td.get("task")
            if t is not None and not RequestStatus.is_finished(t.request_status):
                t.request_status = RequestStatus.FINISHED_ABORTED
                try:
                    t.output_queue.put_nowait(("task_completed", t))
                except Exception:                # noqa: BLE001
                    pass

    def _cleanup(self):
        for name in ("state_slot", "batch_state", "occurrence", "alpha_presence_vector", "model"):
            if hasattr(self, name):
                delattr(self, name)


def model_load_config_for(model_path: str, vocab_path: str, vocab_size: int = 65536) -> ModelLoadConfig:
    return ModelLoadConfig(model_path=model_path, vocab_path=vocab_path, vocab_size=vocab_size, head_size=64)


__all__ = ["Worker", "StateCategory", "min_swaps_to_target_fast", "types"]
