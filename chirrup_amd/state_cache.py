"""Prefix-state cache (reference: chirrup/utils/state_cache.py:51-215).

Two parts:

* ``SimpleStateCache`` -- the reference's interface and observable behaviour (``check``, ``cache``, ``remove``,
  ``check_and_wait_prefill``, ``awake_hang_up_prefills``; LRU over at most ``max_size`` prefixes, a prefix only
  counts as a hit when at least one prompt token is left to feed, a lookup refreshes the entry's recency, caching
  an already cached prefix keeps the first state).  It is checked operation by operation against traces of the
  reference (tests/golden/state_cache.json).  Written as a reference-counted trie: every node counts the cached
  prefixes that pass through it, and a branch disappears when its count reaches zero.
  One deliberate difference: requests waiting for someone else's prefill are parked under the exact token
  prefix they wait for, not on whichever trie node the lookup happened to stop at, so a waiter is always woken
  by the ``cache()`` of that prefix (the reference can strand a waiter when the lookup walked past the last
  cached prefix, state_cache.py:95-106).

* ``HbmStateArena`` -- where the states live on an MI355X.  The reference keeps ``[L,2,1,C]``, ``[L,1,H,64,64]``,
  ``[1]`` CPU tensors per entry (worker.py:426-430) and pays two PCIe transfers of 17-33 MB per hit.  With 288 GB
  of HBM the cache is a preallocated, index-addressed pool on the device: ``capacity`` rows of
  ``[L,2,C] + [L,H,64,64] + [1]``.  States move between the worker's slot table and the arena by ONE device-to-device
  copy each way: the worker exports a prefix straight into a free row (``export_slot``), a hit hands the request an
  ``ArenaRef`` -- a pinned handle on the row, no data -- and ``Worker._install`` copies the row straight into the slot
  (``ArenaRef.install_into``) and unpins it.  A row that is evicted while a hit on it is still queued is only returned
  to the free list when the last pin is gone.  4096 rows of a 13.3B state (33 MB) are 135 GB.
  ``SimpleStateCache(max_size, arena=HbmStateArena(...))`` stores its states there.
"""
import asyncio
import threading
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch


class ArenaRef:
    """A pinned handle on one arena row: what a prefix-cache hit hands to the request (``Task.state``).  The row cannot be
    reused until ``release()`` (``Worker._install`` calls it after its copy is enqueued; dropping the handle releases too)."""
    __slots__ = ("arena", "row", "_live")

    def __init__(self, arena: "HbmStateArena", row: int):
        self.arena, self.row, self._live = arena, row, True

    def install_into(self, pool: Sequence[torch.Tensor], slot: int) -> None:
        """One strided device-to-device copy per state tensor: arena row -> slot `slot` of the worker's tables
        [L,2,n,C], [L,n,H,64,64], [n] (works across devices: a peer copy over xGMI)."""
        a = self.arena
        a.wait_ready(self.row)               # the export into the row may still be queued on ANOTHER worker's stream
        pool[0][:, :, slot, :].copy_(a.shift[self.row], non_blocking=True)
        pool[1][:, slot].copy_(a.wkv[self.row], non_blocking=True)
        pool[2][slot: slot + 1].copy_(a.elapsed[self.row: self.row + 1], non_blocking=True)

    def tensors(self) -> List[torch.Tensor]:
        """Copies in the reference's exported layout [L,2,1,C], [L,1,H,64,64], [1] (for consumers that want tensors)."""
        return self.arena.get(self.row)

    def release(self) -> None:
        if self._live:
            self._live = False
            self.arena._unpin(self.row)

    def __del__(self):
        try:
            self.release()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


class HbmStateArena:
    """Preallocated pool of ``capacity`` RWKV-7 request states on one device."""

    def __init__(self, n_layer: int, n_embd: int, capacity: int, device, head_size: int = 64, dtype=torch.float16,
                 wkv_shape=None, wkv_dtype=None):
        """wkv_shape / wkv_dtype: per-layer shape and dtype of state[1] when they are not the model's [H, 64, 64] binary16
        (host-logic tests with a fake backend)."""
        if capacity <= 0:
            raise ValueError("capacity must be positive")
        self.n_layer, self.n_embd, self.capacity = n_layer, n_embd, capacity
        H = n_embd // head_size
        wkv_shape = tuple(wkv_shape) if wkv_shape is not None else (H, head_size, head_size)
        self.shift = torch.empty((capacity, n_layer, 2, n_embd), dtype=dtype, device=device)            # state[0] rows
        self.wkv = torch.empty((capacity, n_layer) + wkv_shape, dtype=wkv_dtype or dtype, device=device)  # state[1] rows
        self.elapsed = torch.zeros((capacity,), dtype=torch.int32, device=device)                       # state[2]
        self._free = list(range(capacity - 1, -1, -1))
        # row -> event recorded behind the copies that filled it (device arenas): a reader on another stream -- another worker
        # thread's install, this worker's peers in process mode (via export_event) -- must not read the row before it (round-3
        # advisor finding: exports are non-blocking copies on the exporting worker's stream, behind its in-flight forward)
        self._ready: Dict[int, "torch.cuda.Event"] = {}
        self._pins = [0] * capacity
        self._doomed = set()          # rows released by their owner while pinned: freed by the last unpin
        self._lock = threading.Lock()  # the cache runs on the engine's thread, installs on the worker's

    @classmethod
    def for_model(cls, model, capacity: int):
        """Geometry from a chirrup_amd.rwkv7.RWKV_x070 (n_layer, n_embd, head_size, device)."""
        return cls(model.n_layer, model.n_embd, capacity, model.device, head_size=model.head_size)

    @property
    def bytes_per_state(self) -> int:
        return (self.shift[0].numel() + self.wkv[0].numel()) * self.shift.element_size() + 4

    @property
    def free_rows(self) -> int:
        return len(self._free)

    def _take_row(self) -> int:
        with self._lock:
            if not self._free:
                raise RuntimeError("HbmStateArena is full (the cache's max_size must not exceed the arena's capacity)")
            return self._free.pop()

    def put(self, state: Sequence[torch.Tensor]) -> int:
        """Copy one exported state ([L,2,1,C], [L,1,H,64,64], [1]; any device) into a free row; returns the row."""
        s0, s1, s2 = state
        if tuple(s0.shape) != (self.n_layer, 2, 1, self.n_embd) or s1.shape[0] != self.n_layer or s1.shape[1] != 1:
            raise ValueError(f"state shapes {tuple(s0.shape)}, {tuple(s1.shape)} do not match the arena")
        row = self._take_row()
        self.shift[row].copy_(s0[:, :, 0, :], non_blocking=True)
        self.wkv[row].copy_(s1[:, 0], non_blocking=True)
        self.elapsed[row: row + 1].copy_(s2.reshape(1), non_blocking=True)
        self._mark_filled(row)
        return row

    def _mark_filled(self, row: int) -> None:
        if self.shift.is_cuda:
            with torch.cuda.device(self.shift.device):
                ev = torch.cuda.Event()
                ev.record()                    # on the filling thread's current stream: the one the copies were enqueued on
            self._ready[row] = ev

    def export_event(self, row: int):
        """The event behind the copies that filled `row` (None on a host arena, or when it has been waited out)."""
        return self._ready.get(row)

    def wait_ready(self, row: int) -> None:
        """Order the CURRENT stream behind the copies that filled `row` (device-side wait; nothing on a host arena)."""
        ev = self._ready.get(row)
        if ev is not None:
            if ev.query():
                self._ready.pop(row, None)     # complete: nothing left to order against
            else:
                torch.cuda.current_stream().wait_event(ev)

    def export_slot(self, pool: Sequence[torch.Tensor], slot: int) -> "ArenaRef":
        """Copy slot `slot` of a worker's state tables straight into a free row (one strided copy per tensor) and return a
        handle on it -- what ``Worker._export_state`` sends with ("cache_prefill", ...) when it has an arena; hand it to
        ``SimpleStateCache.cache`` (which takes the row over) or release it."""
        if tuple(pool[0].shape[:2]) != (self.n_layer, 2) or pool[0].shape[3] != self.n_embd:
            raise ValueError("state tables do not match the arena")
        row = self._take_row()
        self.shift[row].copy_(pool[0][:, :, slot, :], non_blocking=True)
        self.wkv[row].copy_(pool[1][:, slot], non_blocking=True)
        self.elapsed[row: row + 1].copy_(pool[2][slot: slot + 1], non_blocking=True)
        self._mark_filled(row)
        with self._lock:
            self._pins[row] += 1
            self._doomed.add(row)           # nobody owns it yet: the row goes back when this handle is released ...
        return ArenaRef(self, row)

    def adopt(self, ref: "ArenaRef") -> int:
        """... unless a cache adopts it (``SimpleStateCache.cache``): the row now lives until the cache releases it."""
        with self._lock:
            self._doomed.discard(ref.row)
        ref.release()
        return ref.row

    def ref(self, row: int) -> "ArenaRef":
        """A pinned handle on a cached row (a prefix-cache hit)."""
        with self._lock:
            self._pins[row] += 1
        return ArenaRef(self, row)

    def _unpin(self, row: int) -> None:
        with self._lock:
            self._pins[row] -= 1
            if self._pins[row] == 0 and row in self._doomed:
                self._doomed.discard(row)
                self._free.append(row)

    def get(self, row: int) -> List[torch.Tensor]:
        """A device copy of row ``row`` in the layout the reference exports ([L,2,1,C], [L,1,H,64,64], [1])."""
        self.wait_ready(row)
        return [self.shift[row].unsqueeze(2).clone(), self.wkv[row].unsqueeze(1).clone(), self.elapsed[row: row + 1].clone()]

    def release(self, row: int) -> None:
        """The owner (the cache) gives the row up; it is reused once no hit holds it any more."""
        with self._lock:
            if self._pins[row] > 0:
                self._doomed.add(row)
            else:
                self._free.append(row)


class _ArenaRow:
    """What the LRU holds for an arena-backed entry."""
    __slots__ = ("row",)

    def __init__(self, row: int):
        self.row = row


class TrieNode:
    """One token of a cached (or awaited) prefix.  ``entries`` = cached prefixes that pass through or end here."""
    __slots__ = ("children", "state", "entries")

    def __init__(self):
        self.children: Dict[int, "TrieNode"] = {}
        self.state = False
        self.entries = 0

    @property
    def depend_count(self) -> int:          # the reference's name for the same counter
        return self.entries


class SimpleStateCache:
    def __init__(self, max_size: int = 100, arena=None):
        """arena: an HbmStateArena (thread mode: the states live in this process's HBM pool) or the engine's
        remote_arena.RemoteArena (process mode: they live in the worker processes' pools, this cache keeps row addresses)."""
        if max_size <= 0:
            raise ValueError("capacity must be positive")
        if arena is not None and arena.capacity < max_size:
            raise ValueError("the arena must have at least max_size rows")
        self.max_size = max_size
        self.arena = arena
        self.root = TrieNode()
        self._lru: "OrderedDict[Tuple[int, ...], object]" = OrderedDict()
        self._waiting: Dict[Tuple[int, ...], asyncio.Condition] = {}     # prefixes somebody is prefilling right now
        self.prefill_lock = asyncio.Lock()

    # ------------------------------------------------------------------ bookkeeping
    def __len__(self) -> int:
        return len(self._lru)

    def keys(self) -> List[Tuple[int, ...]]:
        """Cached prefixes, least recently used first."""
        return list(self._lru.keys())

    def _materialise(self, stored):
        if isinstance(stored, _ArenaRow):
            return self.arena.ref(stored.row)        # a pinned handle, no copy: Worker._install copies row -> slot once
        return stored

    def _drop_path(self, tokens: Tuple[int, ...]) -> None:
        """Forget one cached prefix in the trie: decrement the counters along its path, cut the branch that only
        this prefix kept alive."""
        node = self.root
        node.entries -= 1
        for tok in tokens:
            child = node.children[tok]
            child.entries -= 1
            if child.entries == 0:
                del node.children[tok]
                return
            node = child
        node.state = False

    def _discard_state(self, stored) -> None:
        if isinstance(stored, _ArenaRow):
            self.arena.release(stored.row)
        elif isinstance(stored, list):
            del stored[:]                    # like the reference: the tensors are released now, not at GC time

    # ------------------------------------------------------------------ the reference's interface
    def check(self, tokens: List[int], return_trie_node: bool = False):
        """Longest cached proper prefix of ``tokens`` -> (tokens still to feed, its state or None, its length
        [, deepest trie node reached])."""
        node, depth, hit = self.root, 0, 0
        while depth < len(tokens):
            if node.state:
                hit = depth
            nxt = node.children.get(tokens[depth])
            if nxt is None or nxt.entries == 0:
                break
            node = nxt
            depth += 1
        key = tuple(tokens[:hit])
        state = None
        if key in self._lru:
            self._lru.move_to_end(key)
            state = self._materialise(self._lru[key])
            if state is None:                      # the row's worker process has ended (RemoteArena.ref): the prefix is gone --
                self.remove(list(key))             # forget it and look again (a shorter prefix may live elsewhere)
                return self.check(tokens, return_trie_node)
        if return_trie_node:
            return tokens[hit:], state, hit, node
        return tokens[hit:], state, hit

    def cache(self, tokens: Tuple[int, ...], state: object, return_trie_node: bool = False):
        """Remember ``state`` as the state after ``tokens``; evicts the least recently used prefix when full."""
        tokens = tuple(tokens)
        if not tokens:
            return None
        if tokens in self._lru:                    # already cached: refresh its recency, keep the first state
            self._lru.move_to_end(tokens)
            node = self.root
            for tok in tokens:
                node = node.children[tok]
            return node if return_trie_node else None
        node = self.root
        node.entries += 1
        for tok in tokens:
            node = node.children.setdefault(tok, TrieNode())
            node.entries += 1
        node.state = True
        handle = hasattr(state, "row") and hasattr(state, "release")      # an ArenaRef, or a remote_arena.RemoteStateRef
        in_arena = self.arena is not None and getattr(self.arena, "accepts_tensors", True)
        if self.arena is not None and handle and state.arena is self.arena:
            if len(self._lru) >= self.max_size:      # the state already sits in a row (Worker export): take the row over
                old_key, old_state = self._lru.popitem(last=False)
                self._drop_path(old_key)
                self._discard_state(old_state)
            self._lru[tokens] = _ArenaRow(self.arena.adopt(state))
        else:
            if handle:
                state = state.tensors()
            self._lru[tokens] = _ArenaRow(self._put_after_eviction(state)) if in_arena else state
        if not in_arena and len(self._lru) > self.max_size:
            old_key, old_state = self._lru.popitem(last=False)
            self._drop_path(old_key)
            self._discard_state(old_state)
        return node if return_trie_node else None

    def _put_after_eviction(self, state) -> int:
        """Arena-backed insert: make room first (the arena may be exactly max_size rows).  A row whose prefix is evicted
        while a hit on it is still queued stays occupied until that hit is installed, so an arena without spare rows can
        be momentarily full: further least-recently-used prefixes are given up until a row is free."""
        if len(self._lru) >= self.max_size:
            old_key, old_state = self._lru.popitem(last=False)
            self._drop_path(old_key)
            self._discard_state(old_state)
        while self.arena.free_rows == 0 and self._lru:
            old_key, old_state = self._lru.popitem(last=False)
            self._drop_path(old_key)
            self._discard_state(old_state)
        return self.arena.put(state)

    def remove(self, tokens: List[int]) -> None:
        key = tuple(tokens)
        if key not in self._lru:
            return
        stored = self._lru.pop(key)
        self._drop_path(key)
        self._discard_state(stored)

    async def check_and_wait_prefill(self, tokens: List[int], cache_prefill_padding: int):
        """Admission with de-duplicated prefills (state_cache.py:85-124): the first request for an uncached prompt
        gets it back to prefill; identical requests arriving meanwhile wait and then start from the cached state."""
        async with self.prefill_lock:
            rest, state, hit = self.check(tokens)
            if hit + cache_prefill_padding == len(tokens):
                return rest, state, hit
            wanted = tuple(tokens[:-cache_prefill_padding]) if cache_prefill_padding else tuple(tokens)
            cond = self._waiting.get(wanted)
            if cond is None:
                self._waiting[wanted] = asyncio.Condition()
                return rest, state, hit
        async with cond:
            await cond.wait()
        if wanted in self._lru:
            self._lru.move_to_end(wanted)
            got = self._materialise(self._lru[wanted])
            if got is not None:
                return list(tokens[len(wanted):]), got, len(wanted)
            self.remove(list(wanted))              # (its worker died in between)
            return self.check(tokens)
        return rest, state, hit

    async def awake_hang_up_prefills(self, node) -> bool:
        """Wake the requests waiting for the prefix that ends at ``node`` (as returned by
        ``cache(..., return_trie_node=True)``, or the prefix itself as a tuple of tokens)."""
        key = node if isinstance(node, tuple) else self._prefix_of(node)
        cond = self._waiting.pop(key, None) if key is not None else None
        if cond is None:
            return False
        async with cond:
            cond.notify_all()
        return True

    def _prefix_of(self, node: TrieNode) -> Optional[Tuple[int, ...]]:
        for key in self._waiting:                   # few prefills are in flight at a time
            walk = self.root
            for tok in key:
                walk = walk.children.get(tok)
                if walk is None:
                    break
            if walk is node:
                return key
        return None


__all__ = ["SimpleStateCache", "HbmStateArena", "ArenaRef", "TrieNode"]
