"""Process-per-GPU transport for AsyncEngineCore (``worker_mode="process"``).

The reference runs its N workers as threads of one free-threaded (3.14t) interpreter (chirrup/engine_core.py:135-153,
README.md:52).  On an interpreter with a GIL the per-iteration host work of N worker threads serialises, so here each
worker gets its own PROCESS, spawned before the parent has made any GPU call (the parent never touches a GPU in this
mode).  The worker loop, the message tuples and the pull-based balancing are unchanged -- only the queues between the
engine and the workers are replaced by adapters over ``multiprocessing`` queues:

    engine process                                         worker process k (GPU k)
    ------------------------------------------------       ------------------------------------------------------
    ProcessTaskQueue.put_nowait(Task)  --- task dicts --->  RemoteTaskQueue.get_nowait()  (pulled when a slot is free)
    router thread  <--- (task_id, message) / events -----  ResultSink.put_nowait(message), WorkerEventSink
    AbortChannel.put_nowait(("abort", ..)) - task ids --->  abort listener -> the task's local task_event_queue
    control queue  ---------- {"type": "shutdown"} ------>  ControlQueue.get_nowait()

Prefix states: with ``state_arena_rows > 0`` (``AsyncEngineCore(..., state_arena_rows=N)``) every worker process owns an
``HbmStateArena`` on its GPU and only row ADDRESSES cross the process boundary -- export, cache, hit and install without a
state byte on the host; a hit is queued for the worker that owns the row (affinity) and an idle worker that takes it instead
copies the row out of the owner's arena through a HIP IPC handle (chirrup_amd/remote_arena.py).  Without it a worker exports
with ``state_cache_device="cpu"`` and states travel as host tensors (torch's shared-memory pickling), as in round 2.
"""
import multiprocessing as mp
import multiprocessing.connection
import queue
import threading
from collections import OrderedDict
from typing import Any, Callable, Dict, List, Optional

from .core_structure import ModelLoadConfig, RequestStatus, Task
from .remote_arena import WIRE_KEY, PeerArenas, RemoteStateRef, is_wire_row

_TASK_FIELDS = ("prompt_str", "prefill_tokens", "state", "task_id", "priority", "temperature", "top_p", "top_k", "presence_penalty",
                "frequency_penalty", "penalty_decay", "max_tokens", "stop_tokens", "forbidden_tokens", "cache_prefill",
                "cache_prefill_padding", "return_logits")
_RESULT_FIELDS = ("request_status", "generated_tokens", "decoded_texts", "prefill_tokens")


def task_to_wire(task: Task) -> Dict[str, Any]:
    d = {k: getattr(task, k) for k in _TASK_FIELDS}
    if isinstance(d["state"], RemoteStateRef):                           # a row of a worker's arena: its address travels, the
        d["state"] = d["state"].wire()                                   # handle stays pinned in the router until it is installed
        return d
    if d["state"] is not None and hasattr(d["state"], "tensors"):       # an ArenaRef cannot leave its process: ship copies
        ref, d["state"] = d["state"], d["state"].tensors()
        ref.release()
    if d["state"] is not None:
        d["state"] = [t.detach().to("cpu") for t in d["state"]]
    return d


# ----------------------------------------------------------------------------------------- worker-process side
class ResultSink:
    """What a Task's ``output_queue`` is inside a worker process."""

    def __init__(self, result_q, task_id: str, on_done=None, worker_id: Optional[str] = None, arena=None, exports=None):
        self._q, self._id, self._on_done, self._wid, self._arena = result_q, task_id, on_done, worker_id, arena
        self._exports = exports                         # the worker's PendingExports (device arenas)

    def put_nowait(self, msg):
        kind, payload = msg
        if kind == "task_completed":                    # the reference hands back the Task object: ship its final fields
            payload = {k: (int(getattr(payload, k)) if k == "request_status" else getattr(payload, k)) for k in _RESULT_FIELDS}
            if self._exports is not None:
                self._exports.flush(self._id)           # a row address must not arrive behind its request's completion
            if self._on_done is not None:
                self._on_done(self._id)
        elif kind == "cache_prefill" and self._arena is not None and getattr(payload["state"], "arena", None) is self._arena:
            # exported straight into a row of THIS worker's arena: the row now belongs to whoever caches the prefix in the
            # engine process (freed by an {"type": "arena_free"} control message); only its address leaves the process --
            # and only once the copies into the row have COMPLETED: they are non-blocking copies on this worker's stream,
            # behind its in-flight forward, and a peer worker reads the row on a stream of its own, in another process, with
            # nothing to order the two (round-3 advisor finding).  Installs were event-gated already; exports are now too.
            ev = self._arena.export_event(payload["state"].row) if hasattr(self._arena, "export_event") else None
            row = self._arena.adopt(payload["state"])
            payload = {"state": {WIRE_KEY: (self._wid, row)}, "prefilled_tokens": tuple(payload["prefilled_tokens"])}
            if ev is not None and self._exports is not None and not ev.query():
                self._exports.add(ev, self._id, (self._id, (kind, payload)))
                return
        elif kind == "cache_prefill":
            st = payload["state"]
            if hasattr(st, "tensors"):
                ref, st = st, st.tensors()
                ref.release()
            payload = {"state": [t.detach().to("cpu") for t in st], "prefilled_tokens": tuple(payload["prefilled_tokens"])}
        elif kind == "token_generated" and len(payload) > 2:
            payload = (payload[0], payload[1], payload[2].detach().to("cpu"))
        self._q.put((self._id, (kind, payload)))


class PendingExports:
    """("cache_prefill", row address) messages whose device copies are still in flight: posted by the worker's loop
    (``poll``, every iteration, like its pending installs) once their event has passed."""

    def __init__(self, result_q):
        self._q, self._items = result_q, []

    def add(self, event, task_id: str, item) -> None:
        self._items.append((event, task_id, item))

    def poll(self) -> None:
        if not self._items:
            return
        still = []
        for ev, tid, item in self._items:
            if ev.query():
                self._q.put(item)
            else:
                still.append((ev, tid, item))
        self._items = still

    def flush(self, task_id: Optional[str] = None) -> None:
        """Wait out and post the exports of one task (all tasks: None) -- at the task's completion and at shutdown."""
        keep = []
        for ev, tid, item in self._items:
            if task_id is None or tid == task_id:
                ev.synchronize()
                self._q.put(item)
            else:
                keep.append((ev, tid, item))
        self._items = keep


class WorkerEventSink:
    def __init__(self, result_q):
        self._q = result_q

    def put_nowait(self, item):
        self._q.put(("__worker_event__", item))


class ControlQueue:
    """master_event_queue of a worker process: non-blocking reads of the engine's control messages."""

    def __init__(self, q):
        self._q = q

    def get_nowait(self):
        return self._q.get_nowait()                     # raises queue.Empty like queue.Queue


class RemoteTaskQueue:
    """task_queue of a worker process: a task leaves the shared queue only when this worker has a free slot for it, so
    the pull-based balancing of the reference (worker.py:583) carries over unchanged."""

    EARLY_ABORTS = 4096                                 # ids remembered for tasks nobody has pulled yet (oldest dropped first)

    def __init__(self, task_q, result_q, worker_id: str, affinity_qs: Optional[Dict[str, Any]] = None, steal: bool = True):
        self._q, self._result_q, self._wid, self._steal = task_q, result_q, worker_id, steal
        self._affinity = dict(affinity_qs or {})        # worker id -> queue of hits on rows of that worker's arena
        self.arena = None                               # set by the worker once it has built its arena
        self.exports = PendingExports(result_q)         # row addresses whose export copies are still in flight
        self.local_events: Dict[str, queue.Queue] = {}
        self._early_aborts: "OrderedDict[str, None]" = OrderedDict()
        self._lock = threading.Lock()

    def _pull(self) -> Dict[str, Any]:
        """Hits on this worker's own rows first (one local copy), then the shared queue, then -- rather than idling -- hits
        queued for another worker (their rows are read through its IPC handle)."""
        mine = self._affinity.get(self._wid)
        if mine is not None:
            try:
                return mine.get_nowait()
            except queue.Empty:
                pass
        try:
            return self._q.get_nowait()
        except queue.Empty:
            pass
        for wid, q_ in (self._affinity.items() if self._steal else ()):
            if wid != self._wid:
                try:
                    return q_.get_nowait()
                except queue.Empty:
                    pass
        raise queue.Empty

    def poll_exports(self) -> None:
        self.exports.poll()

    def installed(self, task_id: str, peer: bool) -> None:
        """The copy of an arena row into this worker's slot has completed: the engine may let the row go."""
        self._result_q.put((task_id, ("__installed__", {"peer": bool(peer), "worker": self._wid})))

    def get_nowait(self) -> Task:
        d = self._pull()                                # queue.Empty when there is nothing to pull
        ev = queue.Queue()
        with self._lock:
            self.local_events[d["task_id"]] = ev
            # an abort that was broadcast while the task still sat in the shared queue (or between the pull above and this
            # insert): the task's own event queue carries it, as in thread mode and in the reference (interface.py:140-142),
            # and the worker sees it on admission (worker.py:443)
            if self._early_aborts.pop(d["task_id"], 0) is None:
                ev.put_nowait(("abort", None))
        self._result_q.put((d["task_id"], ("__accepted__", self._wid)))
        return Task(output_queue=ResultSink(self._result_q, d["task_id"], on_done=self.forget, worker_id=self._wid, arena=self.arena,
                                            exports=self.exports),
                    task_event_queue=ev, **d)

    def deliver_abort(self, task_id: str) -> None:
        with self._lock:
            ev = self.local_events.get(task_id)
            if ev is None:                              # not pulled by THIS worker (yet): remember it, bounded
                self._early_aborts[task_id] = None
                self._early_aborts.move_to_end(task_id)
                while len(self._early_aborts) > self.EARLY_ABORTS:
                    self._early_aborts.popitem(last=False)
        if ev is not None:
            ev.put_nowait(("abort", None))

    def forget(self, task_id: str) -> None:
        with self._lock:
            self.local_events.pop(task_id, None)


def worker_process_main(worker_id: str, gpu_id: List[int], model_config: ModelLoadConfig, batch_size: int, task_q, result_q,
                        control_q, abort_q, worker_factory: Optional[Callable[..., Any]], worker_kwargs: Dict[str, Any],
                        affinity_qs: Optional[Dict[str, Any]] = None, peer_qs: Optional[Dict[str, Any]] = None) -> None:
    """Entry point of a worker process (spawned: nothing of the parent's CUDA/HIP state is inherited)."""
    worker_kwargs = dict(worker_kwargs)
    tasks = RemoteTaskQueue(task_q, result_q, worker_id, affinity_qs, steal=worker_kwargs.pop("arena_steal", True))
    if worker_kwargs.get("state_arena_rows", 0) > 0 and peer_qs:
        import torch.multiprocessing  # noqa: F401 -- registers the tensor reductions (IPC handles) with multiprocessing's pickler

        worker_kwargs["peer_arenas"] = PeerArenas(worker_id, peer_qs[worker_id], peer_qs)

    def listen():
        while True:
            tid = abort_q.get()
            if tid is None:
                return
            tasks.deliver_abort(tid)

    threading.Thread(target=listen, daemon=True, name=f"chirrup:{worker_id}:aborts").start()
    peers = worker_kwargs.get("peer_arenas")
    try:                                                # construction included: an exception there must reach the engine too
        kw = dict(worker_id=worker_id, gpu_id=gpu_id, model_config=model_config, task_queue=tasks,
                  master_event_queue=ControlQueue(control_q), worker_event_queue=WorkerEventSink(result_q), batch_size=batch_size)
        kw.update(worker_kwargs)
        if worker_factory is not None:
            w = worker_factory(**kw)
        else:
            from .worker import Worker

            kw.setdefault("state_cache_device", "cpu")
            w = Worker(**kw)
        w.start()
    except BaseException as e:                          # noqa: BLE001 -- reported, then re-raised
        result_q.put(("__worker_event__", (worker_id, "worker_error", {"error": f"{type(e).__name__}: {e}"})))
        raise
    finally:
        # Shutdown order: a worker drops the views it has opened of the OTHER workers' arenas and says so;
        # it ends -- and its own arena's memory with it -- only when the engine has heard that from every worker (an {"type":
        # "exit"} control message; bounded wait).  Before, every worker left on the first shutdown message and an arena's owner
        # could be gone while a peer still held its IPC handles ("Producer process has been terminated before all shared CUDA
        # tensors released" in every GPU engine test's log).
        try:
            tasks.exports.flush()
        except Exception:                               # noqa: BLE001 -- a dead device: nothing to post
            pass
        if peers is not None:
            try:
                peers.close()
            except Exception:                           # noqa: BLE001
                pass
            result_q.put(("__worker_event__", (worker_id, "peers_released", {})))
            import time as _t

            t_end = _t.time() + 15.0
            while _t.time() < t_end:
                try:
                    if control_q.get(timeout=0.2).get("type") == "exit":
                        break
                except queue.Empty:
                    pass
        result_q.put(("__worker_event__", (worker_id, "worker_exit", {})))


# ----------------------------------------------------------------------------------------- engine-process side
class AbortChannel:
    """task_event_queue of a request in the engine process: an abort is broadcast to the workers by task id (only the
    worker that holds the task has a local queue for it)."""

    def __init__(self, task_id: str, abort_qs):
        self._id, self._qs = task_id, abort_qs

    def put_nowait(self, item):
        if item and item[0] == "abort":
            for q_ in self._qs:
                q_.put(self._id)

    def empty(self) -> bool:
        return True


class ProcessTaskQueue:
    """task_queue of the engine in process mode: registers the request's result channel, ships the task."""

    def __init__(self, task_q, router: "ResultRouter", affinity_qs: Optional[Dict[str, Any]] = None, prefix_affinity: bool = True):
        self._q, self._router, self._affinity, self.prefix_affinity = task_q, router, dict(affinity_qs or {}), prefix_affinity

    def put_nowait(self, task: Task):
        self._router.register(task)
        if isinstance(task.state, RemoteStateRef) and task.state.arena.is_dead(task.state.worker_id):
            # the row's worker ended after the hit was handed out: the state no longer exists, and the request carries only the
            # tokens BEHIND the cached prefix -- it cannot be prefilled again here.  Completed as aborted; the client retries
            # (its next cache lookup misses: RemoteArena.ref returns None for a dead worker's rows).
            self._router.finish_aborted(task)
            return
        wire = task_to_wire(task)
        if isinstance(task.state, RemoteStateRef):
            self._router.pin(task.task_id, task.state)   # released on "__installed__" (or when the task ends)
            q_ = self._affinity.get(task.state.worker_id) if self.prefix_affinity else None
            if self.prefix_affinity == "avoid":          # (tests) queue the hit for a worker that does NOT own the row: every
                others = [w for w in sorted(self._affinity) if w != task.state.worker_id]   # install is a cross-process copy
                q_ = self._affinity[others[0]] if others else q_
            if q_ is not None:                           # the worker whose arena holds the row pulls it first
                q_.put(wire)
                return
        self._q.put(wire)

    def get_nowait(self):                               # (used when the last worker died: drain what nobody will pull)
        return self._q.get_nowait()

    def drain_affinity(self, worker_id: str) -> List[str]:
        """Task ids of the hits still queued for `worker_id` (its process has ended: nobody may open its rows any more)."""
        q_, out = self._affinity.get(worker_id), []
        while q_ is not None:
            try:
                out.append(q_.get_nowait()["task_id"])
            except queue.Empty:
                break
            except Exception:                           # noqa: BLE001 -- a queue whose peer died mid-write
                break
        return out


class ResultRouter(threading.Thread):
    """Moves worker messages from the shared result queue to the asyncio-side channels of their requests."""

    def __init__(self, result_q, worker_event_queue, on_worker_exit, remote_arena=None):
        super().__init__(daemon=True, name="chirrup:router")
        self._q, self._events, self._on_exit = result_q, worker_event_queue, on_worker_exit
        self._tasks: Dict[str, Task] = {}
        self.owner: Dict[str, str] = {}                 # task id -> worker id, once pulled
        self._lock = threading.Lock()
        self.remote_arena = remote_arena
        self._pinned: Dict[str, RemoteStateRef] = {}    # task id -> the arena row its request starts from, until installed
        self.installs = {"local": 0, "peer": 0}         # arena rows installed by their own worker / by another one (IPC)
        self.peers_released = set()                     # workers that have dropped their views of the other workers' arenas
        self.peers_released_cv = threading.Condition()

    def pin(self, task_id: str, ref: RemoteStateRef) -> None:
        with self._lock:
            self._pinned[task_id] = ref

    def _unpin(self, task_id: str) -> None:
        with self._lock:
            ref = self._pinned.pop(task_id, None)
        if ref is not None:
            ref.release()

    def register(self, task: Task) -> None:
        with self._lock:
            self._tasks[task.task_id] = task

    def task(self, task_id: str) -> Optional[Task]:
        with self._lock:
            return self._tasks.get(task_id)

    def tasks_of(self, worker_id: str) -> List[Task]:
        with self._lock:
            return [self._tasks[t] for t, w in self.owner.items() if w == worker_id and t in self._tasks]

    def pending(self) -> List[Task]:
        with self._lock:
            return [t for tid, t in self._tasks.items() if tid not in self.owner]

    def finish_aborted(self, task: Task) -> None:
        task.request_status = RequestStatus.FINISHED_ABORTED
        self._unpin(task.task_id)
        with self._lock:
            self._tasks.pop(task.task_id, None)
            self.owner.pop(task.task_id, None)
        task.output_queue.put_nowait(("task_completed", task))

    def run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            tid, msg = item
            if tid == "__worker_event__":
                wid, kind, payload = msg
                if kind == "peers_released":
                    with self.peers_released_cv:
                        self.peers_released.add(wid)
                        self.peers_released_cv.notify_all()
                    continue
                if kind in ("worker_exit", "worker_error"):
                    self._on_exit(wid, kind)
                    with self.peers_released_cv:        # (a worker that is gone holds no handles either)
                        self.peers_released.add(wid)
                        self.peers_released_cv.notify_all()
                if kind != "worker_exit":
                    self._events.put_nowait(msg)
                continue
            kind, payload = msg
            with self._lock:
                task = self._tasks.get(tid)
                if kind == "__accepted__":
                    self.owner[tid] = payload
                    continue
                if kind == "task_completed":
                    self._tasks.pop(tid, None)
                    self.owner.pop(tid, None)
            if kind == "__installed__":
                self.installs["peer" if payload.get("peer") else "local"] += 1
                self._unpin(tid)
                continue
            if kind == "task_completed":
                self._unpin(tid)                        # (a request that ended before its row was installed)
            if task is None:
                continue
            if kind == "cache_prefill" and is_wire_row(payload.get("state")) and self.remote_arena is not None:
                wid, row = payload["state"][WIRE_KEY]
                ref = self.remote_arena.incoming(wid, row)
                if ref is None:                         # its worker has ended since: there is no state to cache
                    continue
                payload = {"state": ref, "prefilled_tokens": payload["prefilled_tokens"]}
            if kind == "task_completed":
                task.request_status = RequestStatus(payload["request_status"])
                task.generated_tokens, task.decoded_texts = payload["generated_tokens"], payload["decoded_texts"]
                task.prefill_tokens = payload["prefill_tokens"]
                task.output_queue.put_nowait(("task_completed", task))
            else:
                task.output_queue.put_nowait((kind, payload))


class ProcessWorkerHandle:
    """What ``AsyncEngineCore.workers`` holds in process mode."""

    def __init__(self, worker_id: str, gpu_id: List[int], process, control_q, abort_q):
        self.worker_id, self.gpu_id, self.process, self.control_q, self.abort_q = worker_id, gpu_id, process, control_q, abort_q

    def is_alive(self) -> bool:
        return self.process.is_alive()


class LivenessMonitor(threading.Thread):
    """Engine-side watch over the worker PROCESSES.  A worker that ends without its own goodbye -- a HIP abort or a
    segmentation fault on a GPU fault, an out-of-memory kill, os._exit -- never sends "worker_exit" (that message comes from a
    `finally` of the worker's Python code); its clients would wait forever and init() would sit out its 300-s load timeout.
    This thread waits on the processes' sentinels and, for a process that ended while the engine is not shutting down, posts a
    "worker_error" event (init() fails fast on it) and "worker_exit" (the router completes the worker's requests as aborted).
    Nothing is restarted."""

    def __init__(self, handles, result_q, is_shutdown: Callable[[], bool]):
        super().__init__(daemon=True, name="chirrup:liveness")
        self._handles, self._q, self._is_shutdown = list(handles), result_q, is_shutdown
        self._stop_r, self._stop_w = mp.Pipe(duplex=False)

    def stop(self):
        try:
            self._stop_w.send(None)
        except (OSError, ValueError):
            pass

    def run(self):
        live = {h.process.sentinel: h for h in self._handles}
        while live:
            ready = multiprocessing.connection.wait(list(live) + [self._stop_r])
            if self._stop_r in ready:
                return
            for s_ in ready:
                h = live.pop(s_)
                if self._is_shutdown():
                    continue
                code = h.process.exitcode
                self._q.put(("__worker_event__", (h.worker_id, "worker_error",
                                                  {"error": f"worker process ended unexpectedly (exit code {code})"})))
                self._q.put(("__worker_event__", (h.worker_id, "worker_exit", {})))


def spawn_workers(worker_num: int, model_config: ModelLoadConfig, batch_size: int, worker_factory, worker_kwargs, result_q, task_q,
                  gpu_ids: Optional[List[int]] = None):
    """Start one process per GPU with the ``spawn`` method (a forked child of a process that has initialised HIP is not
    usable, and the parent must not have to).  Returns (handles, affinity queues by worker id).  gpu_ids: device of worker k
    (default k; rehearsals put several workers on one GPU)."""
    ctx = mp.get_context("spawn")
    wids = [f"worker_{k}" for k in range(worker_num)]
    arena_mode = worker_kwargs.get("state_arena_rows", 0) > 0
    affinity_qs = {wid: ctx.Queue() for wid in wids} if arena_mode else {}
    peer_qs = {wid: ctx.Queue() for wid in wids} if arena_mode else {}
    handles = []
    for k, wid in enumerate(wids):
        gpu = [gpu_ids[k] if gpu_ids is not None else k]
        control_q, abort_q = ctx.Queue(), ctx.Queue()
        p = ctx.Process(target=worker_process_main, name=f"chirrup:{wid}", daemon=True,
                        args=(wid, gpu, model_config, batch_size, task_q, result_q, control_q, abort_q, worker_factory, worker_kwargs,
                              affinity_qs, peer_qs))
        p.start()
        h = ProcessWorkerHandle(wid, gpu, p, control_q, abort_q)
        h.keep = (affinity_qs, peer_qs)          # the parent must hold every queue it handed over: with the spawn method a queue's
        handles.append(h)                        # semaphores are unlinked when the parent's object dies, possibly before the child opens them
    return handles, affinity_qs


def make_queues():
    ctx = mp.get_context("spawn")
    return ctx.Queue(), ctx.Queue()
