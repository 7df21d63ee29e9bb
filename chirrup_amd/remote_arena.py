"""Prefix states that stay in HBM when the workers are PROCESSES (``AsyncEngineCore(worker_mode="process")``).

Thread mode keeps one ``HbmStateArena`` in the engine's own process (state_cache.py).  In process mode -- the only mode a
multi-GPU node can use on an interpreter with a GIL -- the engine process never touches a GPU, so round 2 shipped prefix
states through it as host tensors (17-33 MB per export and per hit, pickled through shared memory).  Here every worker
process owns an ``HbmStateArena`` on ITS GPU and only row ADDRESSES travel:

    export   worker k: slot -> a free row of its own arena (one device copy)            ("cache_prefill", {"state": (k, row)})
    cache    engine:   SimpleStateCache(arena=RemoteArena) adopts the address; eviction sends {"type": "arena_free"} to k
    hit      engine:   the task carries (k, row) and is queued for worker k first (affinity); an idle worker j may steal it
    install  worker k: row -> slot, one device copy;   worker j != k: the same copy out of k's arena, which j has opened
                       through a HIP IPC handle (torch's CUDA tensor sharing) -- on another GPU a peer copy over xGMI
                       (reference sizing: scripts/benchmark_nvlink_bandwidth.py:36, chirrup/utils/state_cache.py:284-287)
    confirm  worker:   ("__installed__") once the copy has COMPLETED (an event, polled by the loop): only then may the engine
                       let k reuse the row

No state byte crosses the host.  Reference: chirrup/worker.py:421-435 (export to CPU), :583-597 (upload), chirrup/utils/
state_cache.py:51-215 (policy, kept in the engine process unchanged).
"""
import threading
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

RowId = Tuple[str, int]            # (worker id, row of that worker's arena)
WIRE_KEY = "__remote_row__"


def is_wire_row(state) -> bool:
    return isinstance(state, dict) and WIRE_KEY in state


class RemoteStateRef:
    """Engine-process handle on one row of a worker's arena (what a prefix-cache hit hands to ``completion(state=...)``, and
    what a worker's ("cache_prefill", ...) event carries).  Holds a pin: the row is not handed back to its worker for reuse
    before ``release()``."""
    __slots__ = ("arena", "row", "_live")

    def __init__(self, arena: "RemoteArena", row: RowId):
        self.arena, self.row, self._live = arena, row, True

    @property
    def worker_id(self) -> str:
        return self.row[0]

    def wire(self) -> dict:
        return {WIRE_KEY: (self.row[0], int(self.row[1]))}

    def tensors(self) -> List[torch.Tensor]:
        raise RuntimeError("this prefix state lives in a worker's HBM arena; the engine process holds its address only")

    def release(self) -> None:
        if self._live:
            self._live = False
            self.arena._unpin(self.row)

    def __del__(self):
        try:
            self.release()
        except Exception:           # noqa: BLE001 -- interpreter shutdown
            pass


class RemoteArena:
    """Engine-side bookkeeping over the workers' arenas: the interface ``SimpleStateCache(arena=...)`` uses (``adopt``, ``ref``,
    ``release``), with rows addressed as (worker id, row).  It owns no memory; ``send_free(worker_id, row)`` tells a worker that
    a row may be reused -- only when the cache has dropped the prefix AND no hit on it is still on its way."""
    accepts_tensors = False        # host tensors handed to cache() stay host tensors (there is no device in this process)

    def __init__(self, rows_per_worker: int, n_workers: int, send_free: Callable[[str, int], None]):
        self.rows_per_worker, self.n_workers = rows_per_worker, n_workers
        self._send_free = send_free
        self._pins: Dict[RowId, int] = {}
        self._doomed = set()
        self._dead = set()                    # workers that have ended: their arenas (and every row address into them) are gone
        self._lock = threading.Lock()
        self.freed: List[RowId] = []          # (for tests / telemetry)

    @property
    def capacity(self) -> int:
        return self.rows_per_worker * self.n_workers

    @property
    def free_rows(self) -> int:               # allocation happens in the workers; the cache only needs "not exhausted here"
        return self.capacity

    def worker_dead(self, worker_id: str) -> None:
        """The worker process has ended (round-3 advisor finding: its rows stayed live in the engine; a hit on one was queued
        for a dead process, and a stealer then opened the IPC handle of freed memory).  From here on ``ref`` on one of its
        rows returns None -- SimpleStateCache drops the entry and the request prefills again -- and nothing is sent to it."""
        with self._lock:
            self._dead.add(worker_id)

    def is_dead(self, worker_id: str) -> bool:
        with self._lock:
            return worker_id in self._dead

    def incoming(self, worker_id: str, row: int) -> Optional[RemoteStateRef]:
        """A row a worker has just exported into (nobody owns it yet: it goes back unless a cache adopts it)."""
        rid = (worker_id, int(row))
        with self._lock:
            if worker_id in self._dead:
                return None
            self._pins[rid] = self._pins.get(rid, 0) + 1
            self._doomed.add(rid)
        return RemoteStateRef(self, rid)

    def adopt(self, ref: RemoteStateRef) -> RowId:
        with self._lock:
            self._doomed.discard(ref.row)
        ref.release()
        return ref.row

    def ref(self, row: RowId) -> Optional[RemoteStateRef]:
        with self._lock:
            if row[0] in self._dead:
                return None                   # the memory is gone with its process
            self._pins[row] = self._pins.get(row, 0) + 1
        return RemoteStateRef(self, row)

    def release(self, row: RowId) -> None:
        with self._lock:
            if self._pins.get(row, 0) > 0:
                self._doomed.add(row)
                return
        self._free(row)

    def _unpin(self, row: RowId) -> None:
        with self._lock:
            n = self._pins.get(row, 0) - 1
            if n > 0:
                self._pins[row] = n
                return
            self._pins.pop(row, None)
            if row not in self._doomed:
                return
            self._doomed.discard(row)
        self._free(row)

    def _free(self, row: RowId) -> None:
        self.freed.append(row)
        if not self.is_dead(row[0]):
            self._send_free(row[0], row[1])


class PeerArenas:
    """Worker-process side: this worker's own arena plus views of the other workers' arenas, opened through HIP IPC handles
    (torch's CUDA tensor sharing -- the handle of the arena's allocation travels, no data).  ``publish`` is called once by
    every worker after it has built its arena; a peer's tensors are picked up lazily, the first time a hit on one of its
    rows arrives."""

    def __init__(self, worker_id: str, inbox, outboxes: Dict[str, object]):
        self.worker_id, self._inbox, self._outboxes = worker_id, inbox, outboxes
        self.local = None
        self._peers: Dict[str, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}
        self._gone = set()                       # peers that have ended (engine: {"type": "peer_dead"})

    def publish(self, arena) -> None:
        self.local = arena
        if arena.shift.device.type == "cpu":                  # CPU rehearsals: a host arena travels as shared memory.  Move it
            for t in (arena.shift, arena.wkv, arena.elapsed):  # there NOW, on this thread: left to the queue's feeder thread the move
                t.share_memory_()                              # could race with this worker's first export into a row
        for wid, q in self._outboxes.items():
            if wid != self.worker_id:
                q.put((self.worker_id, arena.shift, arena.wkv, arena.elapsed))     # IPC handles, not bytes

    def _drain(self) -> None:
        import queue as _q

        while True:
            try:
                wid, shift, wkv, elapsed = self._inbox.get_nowait()
            except _q.Empty:
                return
            except Exception:          # noqa: BLE001 -- a publication whose sender has ended since cannot be opened (its handle / its
                continue               # shared-memory descriptor died with it): skip it, installs from that worker fail one by one
            self._peers[wid] = (shift, wkv, elapsed)

    def close(self) -> None:
        """Drop every view of the other workers' arenas (their IPC handles) -- BEFORE the owning processes end: a producer that
        exits while a consumer still holds its shared tensors prints "Producer process has been terminated before all shared
        CUDA tensors released" (torch CudaIPCTypes.cpp) and leaves the release to process teardown order."""
        self._drain()
        self._peers.clear()
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
            torch.cuda.ipc_collect()

    def forget(self, worker_id: str) -> None:
        """A peer has ended: its handles are stale, installs from it fail from here on."""
        self._peers.pop(worker_id, None)
        self._gone.add(worker_id)

    def tensors_of(self, worker_id: str, wait_s: float = 1.0):
        """(An arena is published before any address of one of its rows can exist, so the publication is already in the inbox
        when a hit arrives: the wait only covers the queue's feeder thread.  It used to be 30 s -- inside the serving loop.)"""
        if worker_id == self.worker_id:
            a = self.local
            return a.shift, a.wkv, a.elapsed
        if worker_id in self._gone:
            raise RuntimeError(f"{self.worker_id}: worker {worker_id} has ended, its arena is gone")
        if worker_id not in self._peers:
            import time

            t_end = time.time() + wait_s
            while worker_id not in self._peers and time.time() < t_end:
                self._drain()
                if worker_id not in self._peers:
                    time.sleep(0.01)
        if worker_id not in self._peers:
            raise RuntimeError(f"{self.worker_id}: the arena of {worker_id} was never published")
        return self._peers[worker_id]

    def install(self, worker_id: str, row: int, pool: Sequence[torch.Tensor], slot: int) -> None:
        """Arena row -> slot of this worker's state tables: one strided device copy per tensor (another process's memory on
        the same GPU, or a peer copy from another GPU)."""
        shift, wkv, elapsed = self.tensors_of(worker_id)
        pool[0][:, :, slot, :].copy_(shift[row], non_blocking=True)
        pool[1][:, slot].copy_(wkv[row], non_blocking=True)
        pool[2][slot: slot + 1].copy_(elapsed[row: row + 1], non_blocking=True)
