"""Async engine façade over N single-GPU workers (boundary B4 of SURVEY.md section 8b).

API of the reference's chirrup/engine_core.py: ``AsyncEngineCore().init(worker_num, model_config,
batch_size)`` -> asyncio.Task that completes when every worker has loaded; ``.completion(prompt_str,
...)`` -> AsyncEngineCompletion; ``.shutdown()``; ``.iter_worker_performance()``.  One shared host
task queue, workers pull from it (replica data parallelism, worker k on GPU k); worker -> asyncio
traffic goes through ``call_soon_threadsafe`` (ThreadSafeAsyncQueue, reference :30-57).

``worker_mode="thread"`` (default) runs the workers as daemon threads like the reference, which relies on a
free-threaded interpreter for that (README.md:52).  On a GIL build the per-iteration host work of N worker threads
serialises (~1 ms each per ~7.5 ms step), so ``worker_mode="process"`` gives every worker its own process, spawned
before the engine process makes any GPU call -- same messages, same pull-based balancing, queues over
``multiprocessing`` (chirrup_amd/engine_process.py).  ``worker_mode="auto"`` picks "process" for more than one worker on
an interpreter with a GIL.
"""
import sys
import asyncio
import queue
import threading
import uuid
from typing import Any, AsyncIterator, Callable, Dict, List, Optional

from .core_structure import DEFAULT_SAMPLING_CONFIG, DEFAULT_STOP_TOKENS, ModelLoadConfig, Task
from .interface import AsyncEngineCompletion


class ThreadSafeAsyncQueue:
    """asyncio.Queue that worker threads may put into."""

    def __init__(self, event_loop: asyncio.AbstractEventLoop, q: Optional[asyncio.Queue] = None):
        self.event_loop = event_loop
        self.queue: asyncio.Queue = q if q is not None else asyncio.Queue()

    def put_nowait(self, item):
        if self.event_loop.is_closed():
            return

        def _put(x):
            try:
                self.queue.put_nowait(x)
            except asyncio.QueueFull:
                pass                      # bounded queues carry droppable telemetry only

        try:
            self.event_loop.call_soon_threadsafe(_put, item)
        except RuntimeError:
            pass                          # loop already closed

    def empty(self) -> bool:
        return self.queue.empty()

    def get_nowait(self):
        return self.queue.get_nowait()


class AsyncEngineCore:
    LOAD_TIMEOUT_S = 300

    def __init__(self, worker_factory: Optional[Callable[..., Any]] = None, tokenizer=None, worker_mode: str = "thread",
                 worker_kwargs: Optional[Dict[str, Any]] = None, state_arena_rows: int = 0, prefix_affinity: bool = True,
                 gpu_ids: Optional[List[int]] = None):
        """worker_factory(**worker_args) -> object with .start(): defaults to chirrup_amd.worker.Worker; in process mode it is
        called INSIDE the worker process and must be picklable (a module-level function).  worker_kwargs: extra Worker
        arguments (e.g. run_ahead).  state_arena_rows (process mode): rows of the HBM prefix-state arena EVERY worker process
        builds on its GPU; ``self.state_arena`` is then the engine-side view to hand to ``SimpleStateCache(arena=...)`` -- prefix
        states never leave HBM (chirrup_amd/remote_arena.py).  prefix_affinity=False queues hits on the shared queue instead of
        the owning worker's, "avoid" on ANOTHER worker's (tests: every hit is then installed through the owner's IPC handle).  gpu_ids: device of
        worker k (default k)."""
        if worker_mode not in ("thread", "process", "auto"):
            raise ValueError("worker_mode must be 'thread', 'process' or 'auto'")
        self.worker_mode = worker_mode
        self._worker_kwargs = dict(worker_kwargs or {})
        self.state_arena_rows, self.prefix_affinity, self.gpu_ids = int(state_arena_rows), prefix_affinity, gpu_ids
        self.state_arena = None                       # process mode with state_arena_rows > 0: a remote_arena.RemoteArena
        self._router = None
        self._monitor = None
        self._result_q = None
        self.workers: List[Any] = []
        self.worker_threads: List[threading.Thread] = []
        self.task_queue: "queue.Queue[Task]" = queue.Queue()
        self.event_queue: "queue.Queue[Dict[str, Any]]" = queue.Queue()
        self.worker_id_set = set()
        self.worker_event_queue: Optional[ThreadSafeAsyncQueue] = None
        self.event_loop: Optional[asyncio.AbstractEventLoop] = None
        self.is_initialized = False
        self.is_shutdown = False
        self.tokenizer = tokenizer
        self._worker_factory = worker_factory

    def _make_worker(self, **kw):
        kw.update(self._worker_kwargs)
        if self._worker_factory is not None:
            return self._worker_factory(**kw)
        from .worker import Worker

        return Worker(**kw)

    @staticmethod
    def _gil_enabled() -> bool:
        return getattr(sys, "_is_gil_enabled", lambda: True)()

    def _start_process_workers(self, worker_num, model_config, batch_size):
        """One process per GPU, spawned before anything in THIS process has touched a GPU (it never has to)."""
        from . import engine_process as ep

        self._result_q, mp_task_q = ep.make_queues()
        kwargs = dict(self._worker_kwargs)
        if self.state_arena_rows > 0:
            from .remote_arena import RemoteArena

            kwargs["state_arena_rows"] = self.state_arena_rows
            kwargs["arena_steal"] = self.prefix_affinity != "avoid"      # ("avoid": tests -- the owner must not take its hits back)
            self.state_arena = RemoteArena(self.state_arena_rows, worker_num, self._send_arena_free)
        self._router = ep.ResultRouter(self._result_q, self.worker_event_queue, self._on_process_worker_exit, self.state_arena)
        self.workers, affinity_qs = ep.spawn_workers(worker_num, model_config, batch_size, self._worker_factory, kwargs,
                                                     self._result_q, mp_task_q, self.gpu_ids)
        self.task_queue = ep.ProcessTaskQueue(mp_task_q, self._router, affinity_qs, self.prefix_affinity)
        self._router.start()
        self._monitor = ep.LivenessMonitor(self.workers, self._result_q, lambda: self.is_shutdown)
        self._monitor.start()

    def _send_arena_free(self, worker_id: str, row: int) -> None:
        """RemoteArena: the cache has dropped a prefix and no hit on it is in flight -- its worker may reuse the row."""
        for w in self.workers:
            if w.worker_id == worker_id and w.is_alive():
                w.control_q.put({"type": "arena_free", "row": int(row)})

    def _on_process_worker_exit(self, worker_id: str, kind: str) -> None:
        """Router thread: a worker process reported an error or ended -- by its own message, or, for a hard exit (HIP abort,
        segmentation fault, kill), by the LivenessMonitor that watches the process sentinels.  Its requests are completed as
        aborted (its own loop does that when it can), and so are the queued ones when no worker is left."""
        if self.is_shutdown or self._router is None:
            return
        for task in self._router.tasks_of(worker_id):
            self._router.finish_aborted(task)
        others = [w for w in self.workers if w.worker_id != worker_id and w.is_alive()]
        if self.state_arena is not None:
            # the worker's arena went with its process (round-3 advisor finding): its rows are dead in the engine's books
            # (cache lookups drop them and the requests prefill again), the hits still queued for it complete as aborted
            # (a hit carries only the tokens behind its prefix), and the surviving workers forget their views of its memory
            self.state_arena.worker_dead(worker_id)
            for tid in self.task_queue.drain_affinity(worker_id):
                task = self._router.task(tid)
                if task is not None:
                    self._router.finish_aborted(task)
            for w in others:
                w.control_q.put({"type": "peer_dead", "worker": worker_id})
        if not others:
            for task in self._router.pending():
                self._router.finish_aborted(task)
            while True:
                try:
                    self.task_queue.get_nowait()
                except Exception:                # noqa: BLE001 -- queue.Empty
                    break

    def init(self, worker_num: int, model_config: ModelLoadConfig, batch_size: int = 32) -> "asyncio.Task":
        if self.is_initialized:
            raise RuntimeError("Workers already initialized")
        if self.is_shutdown:
            raise RuntimeError("Engine has been shutdown")
        try:
            self.event_loop = asyncio.get_running_loop()
        except RuntimeError:
            self.event_loop = asyncio.new_event_loop()
            asyncio.set_event_loop(self.event_loop)
        self.worker_event_queue = ThreadSafeAsyncQueue(self.event_loop, asyncio.Queue(maxsize=worker_num * 100))
        self.is_initialized = True
        if self.tokenizer is None:
            from .tokenizer import TRIE_TOKENIZER

            self.tokenizer = TRIE_TOKENIZER(model_config.vocab_path)

        mode = self.worker_mode
        if mode == "auto":
            mode = "process" if (worker_num > 1 and self._gil_enabled()) else "thread"
        self.worker_mode = mode

        async def wait_loaded():
            self.worker_id_set = {f"worker_{i}" for i in range(worker_num)}
            if mode == "process":
                self._start_process_workers(worker_num, model_config, batch_size)
            for k, wid in enumerate(sorted(self.worker_id_set) if mode == "thread" else []):
                w = self._make_worker(worker_id=wid, gpu_id=[k], model_config=model_config, task_queue=self.task_queue,
                                      master_event_queue=self.event_queue, worker_event_queue=self.worker_event_queue,
                                      batch_size=batch_size)
                w.on_fatal = self._on_worker_fatal
                self.workers.append(w)
                t = threading.Thread(target=w.start, daemon=True, name=f"chirrup:{wid}")
                t.start()
                self.worker_threads.append(t)
            loaded, budget = set(), self.LOAD_TIMEOUT_S
            while len(loaded) < worker_num and budget > 0:
                try:
                    wid, kind, payload = await asyncio.wait_for(self.worker_event_queue.queue.get(), timeout=1.0)
                except asyncio.TimeoutError:
                    budget -= 1
                    continue
                if kind == "worker_error" or (kind == "worker_loaded" and payload.get("status") != "success"):
                    raise RuntimeError(f"Worker {wid} failed to load: {payload}")
                if kind == "worker_loaded":
                    loaded.add(wid)
            if len(loaded) < worker_num:
                raise RuntimeError(f"workers timed out while loading: {self.worker_id_set - loaded}")

        return asyncio.create_task(wait_loaded())

    def completion(self, prompt_str: str, prefill_tokens: Optional[List[int]] = None, state=None, priority: int = 0,
                   temperature: float = DEFAULT_SAMPLING_CONFIG["temperature"], top_p: float = DEFAULT_SAMPLING_CONFIG["top_p"],
                   top_k: int = DEFAULT_SAMPLING_CONFIG["top_k"],
                   presence_penalty: float = DEFAULT_SAMPLING_CONFIG["presence_penalty"],
                   frequency_penalty: float = DEFAULT_SAMPLING_CONFIG["frequency_penalty"],
                   penalty_decay: float = DEFAULT_SAMPLING_CONFIG["penalty_decay"],
                   stop_tokens: Optional[List[int]] = DEFAULT_STOP_TOKENS, forbidden_tokens: Optional[List[int]] = None,
                   max_tokens: Optional[int] = DEFAULT_SAMPLING_CONFIG["max_tokens"], task_id: Optional[str] = None,
                   cache_prefill: bool = False, cache_prefill_padding: int = 0,
                   return_logits: bool = False) -> AsyncEngineCompletion:
        assert not (state is not None and prefill_tokens is None), "prefill_tokens cannot be None when state is not None"
        if not self.is_initialized:
            raise RuntimeError("Engine not initialized")
        if self.is_shutdown:
            raise RuntimeError("Engine has been shutdown")
        if not prefill_tokens:
            prefill_tokens = self.tokenizer.encode(prompt_str)
        task_id = task_id or str(uuid.uuid4())
        abort_channel = None
        if self.worker_mode == "process":
            from .engine_process import AbortChannel

            abort_channel = AbortChannel(task_id, [w.abort_q for w in self.workers])
        return AsyncEngineCompletion(prompt_str=prompt_str, prefill_tokens=list(prefill_tokens), state=state,
                                     task_queue=self.task_queue, result_channel=ThreadSafeAsyncQueue(self.event_loop),
                                     task_id=task_id, priority=priority, temperature=temperature, task_event_queue=abort_channel,
                                     top_p=top_p, top_k=top_k, presence_penalty=presence_penalty,
                                     frequency_penalty=frequency_penalty, penalty_decay=penalty_decay,
                                     stop_tokens=stop_tokens, forbidden_tokens=forbidden_tokens, max_tokens=max_tokens,
                                     cache_prefill=cache_prefill, cache_prefill_padding=cache_prefill_padding,
                                     return_logits=return_logits)

    def _on_worker_fatal(self, worker, exc) -> None:
        """Called on a dying worker's thread.  When it was the last live worker, nobody will ever pick up the
        requests still in the shared queue: complete them as aborted instead of leaving their clients waiting."""
        me = threading.current_thread()
        if any(t.is_alive() and t is not me for t in self.worker_threads):
            return
        from .core_structure import RequestStatus

        while True:
            try:
                task = self.task_queue.get_nowait()
            except Exception:                    # noqa: BLE001 -- queue.Empty
                break
            task.request_status = RequestStatus.FINISHED_ABORTED
            task.output_queue.put_nowait(("task_completed", task))

    def shutdown(self) -> None:
        if self.is_shutdown:
            return
        self.is_shutdown = True
        if self.worker_mode == "process" and self._router is not None:
            if self._monitor is not None:
                self._monitor.stop()
            for w in self.workers:
                w.control_q.put({"type": "shutdown"})
                w.abort_q.put(None)
            if self.state_arena is not None:
                # two phases: every worker first drops its views of the other workers' arenas ("peers_released"), only then
                # may the owners of those arenas end (engine_process.worker_process_main)
                import time

                alive = {w.worker_id for w in self.workers if w.is_alive()}
                t_end = time.time() + 15.0
                with self._router.peers_released_cv:
                    while not alive <= self._router.peers_released and time.time() < t_end:
                        self._router.peers_released_cv.wait(timeout=0.2)
                        alive = {w.worker_id for w in self.workers if w.is_alive()}
                for w in self.workers:
                    w.control_q.put({"type": "exit"})
            for w in self.workers:
                w.process.join(timeout=10)
                if w.process.is_alive():
                    w.process.terminate()
            self._result_q.put(None)
            self._router.join(timeout=5)
            return
        for _ in range(max(1, len(self.workers))):       # every worker consumes one shutdown event
            self.event_queue.put_nowait({"type": "shutdown"})
        for t in self.worker_threads:
            if t.is_alive():
                t.join(timeout=5)

    async def iter_worker_performance(self, timeout: float = 1.0) -> AsyncIterator[Dict[str, Any]]:
        if self.worker_event_queue is None:
            raise RuntimeError("Engine not initialized")
        while not self.is_shutdown:
            try:
                wid, kind, payload = await asyncio.wait_for(self.worker_event_queue.queue.get(), timeout=timeout)
            except asyncio.TimeoutError:
                continue
            if kind == "worker_performance":
                yield dict(worker_id=wid, **{k: payload[k] for k in ("avg_loop_time", "state_size", "state_offset_details",
                                                                       "task_details", "max_allocated_memory_GB")})

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
