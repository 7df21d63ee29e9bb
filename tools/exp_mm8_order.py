"""Why did the cached-pack rwkv_pip::mm8_seq time 486.6 us beside 68.8 us for pack-per-call on the driver's round-3 box
(GPUTEST_r03.json; the builder's box printed 63.0 us for the same line)?  VERDICT r3 item 1(a).

Phase A replays the record: seconds of CPU BLAS work with the GPU idle, H2D copies, ONE warm-up call, then 10 / 5 / 2 timed
calls in the order cached, stateless, direct -- with the HOST time of the launch loop beside the event time, so that a
host-bound region and a slow-clock region can be told apart.
Phase B: the three variants in both orders after >= 50 ms of GPU work, 50 iterations each.
Phase C: the cached path again after a 3-s idle with 1, 2, 5, 10, 50 iterations back to back (how long the ramp is).
Prints one line per region; nothing is asserted."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from chirrup_amd import ops                      # noqa: E402
from oracle import rwkv7_np as M_                # noqa: E402  (tools/: the oracle is only used to reproduce the CPU pause)


def region(fn, args, out, n, warm=1):
    for _ in range(warm):
        fn(*args, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn(*args, out)
    e1.record()
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, (h1 - h0) / n * 1e6


def main():
    B, N, M = 200, 4096, 16384
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, N)).astype(np.float16)
    w16 = (rng.standard_normal((N, M)) / np.sqrt(N)).astype(np.float16)
    q, mx, rx, my, ry = M_.quantize_weight(w16)
    my, ry = my.reshape(-1), ry.reshape(-1)
    t0 = time.perf_counter()
    M_.mm8_seq_blas(x, q, mx, rx, my, ry)        # the seconds of CPU work in front of the first timed region
    print(f"cpu pause {time.perf_counter() - t0:.1f} s", flush=True)
    t = [torch.from_numpy(z).cuda() for z in (x, q, mx, rx, my, ry)]
    y, yd = (torch.empty((B, M), dtype=torch.float16, device="cuda") for _ in range(2))
    args = (B, N, M, *t)
    names = {"cached": ops.mm8_seq, "stateless": ops.mm8_seq_stateless, "direct": ops.mm8_seq_direct}

    print("A: the record's sequence (1 warm-up call each, right after the CPU pause)")
    for k, n in (("cached", 10), ("stateless", 5), ("direct", 2)):
        ev, host = region(names[k], args, yd if k == "direct" else y, n)
        print(f"  {k:10s} n={n:3d}  event {ev:8.1f} us/call   host {host:8.1f} us/call", flush=True)

    for order in (("cached", "stateless", "direct"), ("direct", "stateless", "cached")):
        for _ in range(800):                      # >= 50 ms of GPU work
            ops.mm8_seq(*args, y)
        torch.cuda.synchronize()
        print("B: after >= 50 ms of GPU work, order " + " -> ".join(order))
        for k in order:
            ev, host = region(names[k], args, yd if k == "direct" else y, 50 if k != "direct" else 10, warm=3)
            print(f"  {k:10s} event {ev:8.1f} us/call   host {host:8.1f} us/call", flush=True)

    print("C: cached path after a 3-s idle, n calls back to back (no warm-up call)")
    for n in (1, 2, 5, 10, 50, 200):
        torch.cuda.synchronize()
        time.sleep(3.0)
        ev, host = region(ops.mm8_seq, args, y, n, warm=0)
        print(f"  n={n:3d}  event {ev:8.1f} us/call   host {host:8.1f} us/call", flush=True)


if __name__ == "__main__":
    main()
