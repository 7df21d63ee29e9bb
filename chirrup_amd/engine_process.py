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

Prefix states cross the process boundary as host tensors (torch's shared-memory pickling): a worker in this mode exports
with ``state_cache_device="cpu"``; an HBM-resident arena needs the workers in the engine's own process (thread mode).
"""
import multiprocessing as mp
import multiprocessing.connection
import queue
import threading
from collections import OrderedDict
from typing import Any, Callable, Dict, List, Optional

from .core_structure import ModelLoadConfig, RequestStatus, Task

_TASK_FIELDS = ("prompt_str", "prefill_tokens", "state", "task_id", "priority", "temperature", "top_p", "top_k", "presence_penalty",
                "frequency_penalty", "penalty_decay", "max_tokens", "stop_tokens", "forbidden_tokens", "cache_prefill",
                "cache_prefill_padding", "return_logits")
_RESULT_FIELDS = ("request_status", "generated_tokens", "decoded_texts", "prefill_tokens")


def task_to_wire(task: Task) -> Dict[str, Any]:
    d = {k: getattr(task, k) for k in _TASK_FIELDS}
    if d["state"] is not None and hasattr(d["state"], "tensors"):       # an ArenaRef cannot leave its process: ship copies
        ref, d["state"] = d["state"], d["state"].tensors()
        ref.release()
    if d["state"] is not None:
        d["state"] = [t.detach().to("cpu") for t in d["state"]]
    return d


# ----------------------------------------------------------------------------------------- worker-process side
class ResultSink:
    """What a Task's ``output_queue`` is inside a worker process."""

    def __init__(self, result_q, task_id: str, on_done=None):
        self._q, self._id, self._on_done = result_q, task_id, on_done

    def put_nowait(self, msg):
        kind, payload = msg
        if kind == "task_completed":                    # the reference hands back the Task object: ship its final fields
            payload = {k: (int(getattr(payload, k)) if k == "request_status" else getattr(payload, k)) for k in _RESULT_FIELDS}
            if self._on_done is not None:
                self._on_done(self._id)
        elif kind == "cache_prefill":
            st = payload["state"]
            if hasattr(st, "tensors"):
                ref, st = st, st.tensors()
                ref.release()
            payload = {"state": [t.detach().to("cpu") for t in st], "prefilled_tokens": tuple(payload["prefilled_tokens"])}
        elif kind == "token_generated" and len(payload) > 2:
            payload = (payload[0], payload[1], payload[2].detach().to("cpu"))
        self._q.put((self._id, (kind, payload)))


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

    def __init__(self, task_q, result_q, worker_id: str):
        self._q, self._result_q, self._wid = task_q, result_q, worker_id
        self.local_events: Dict[str, queue.Queue] = {}
        self._early_aborts: "OrderedDict[str, None]" = OrderedDict()
        self._lock = threading.Lock()

    def get_nowait(self) -> Task:
        d = self._q.get_nowait()                        # queue.Empty when there is nothing to pull
        ev = queue.Queue()
        with self._lock:
            self.local_events[d["task_id"]] = ev
            # an abort that was broadcast while the task still sat in the shared queue (or between the pull above and this
            # insert): the task's own event queue carries it, as in thread mode and in the reference (interface.py:140-142),
            # and the worker sees it on admission (worker.py:443)
            if self._early_aborts.pop(d["task_id"], 0) is None:
                ev.put_nowait(("abort", None))
        self._result_q.put((d["task_id"], ("__accepted__", self._wid)))
        return Task(output_queue=ResultSink(self._result_q, d["task_id"], on_done=self.forget), task_event_queue=ev, **d)

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
                        control_q, abort_q, worker_factory: Optional[Callable[..., Any]], worker_kwargs: Dict[str, Any]) -> None:
    """Entry point of a worker process (spawned: nothing of the parent's CUDA/HIP state is inherited)."""
    tasks = RemoteTaskQueue(task_q, result_q, worker_id)

    def listen():
        while True:
            tid = abort_q.get()
            if tid is None:
                return
            tasks.deliver_abort(tid)

    threading.Thread(target=listen, daemon=True, name=f"chirrup:{worker_id}:aborts").start()
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

    def __init__(self, task_q, router: "ResultRouter"):
        self._q, self._router = task_q, router

    def put_nowait(self, task: Task):
        self._router.register(task)
        self._q.put(task_to_wire(task))

    def get_nowait(self):                               # (used when the last worker died: drain what nobody will pull)
        return self._q.get_nowait()


class ResultRouter(threading.Thread):
    """Moves worker messages from the shared result queue to the asyncio-side channels of their requests."""

    def __init__(self, result_q, worker_event_queue, on_worker_exit):
        super().__init__(daemon=True, name="chirrup:router")
        self._q, self._events, self._on_exit = result_q, worker_event_queue, on_worker_exit
        self._tasks: Dict[str, Task] = {}
        self.owner: Dict[str, str] = {}                 # task id -> worker id, once pulled
        self._lock = threading.Lock()

    def register(self, task: Task) -> None:
        with self._lock:
            self._tasks[task.task_id] = task

    def tasks_of(self, worker_id: str) -> List[Task]:
        with self._lock:
            return [self._tasks[t] for t, w in self.owner.items() if w == worker_id and t in self._tasks]

    def pending(self) -> List[Task]:
        with self._lock:
            return [t for tid, t in self._tasks.items() if tid not in self.owner]

    def finish_aborted(self, task: Task) -> None:
        task.request_status = RequestStatus.FINISHED_ABORTED
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
                if kind in ("worker_exit", "worker_error"):
                    self._on_exit(wid, kind)
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
            if task is None:
                continue
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


def spawn_workers(worker_num: int, model_config: ModelLoadConfig, batch_size: int, worker_factory, worker_kwargs, result_q, task_q):
    """Start one process per GPU with the ``spawn`` method (a forked child of a process that has initialised HIP is not
    usable, and the parent must not have to).  Returns the handles."""
    ctx = mp.get_context("spawn")
    handles = []
    for k in range(worker_num):
        wid = f"worker_{k}"
        control_q, abort_q = ctx.Queue(), ctx.Queue()
        p = ctx.Process(target=worker_process_main, name=f"chirrup:{wid}", daemon=True,
                        args=(wid, [k], model_config, batch_size, task_q, result_q, control_q, abort_q, worker_factory, worker_kwargs))
        p.start()
        handles.append(ProcessWorkerHandle(wid, [k], p, control_q, abort_q))
    return handles


def make_queues():
    ctx = mp.get_context("spawn")
    return ctx.Queue(), ctx.Queue()
