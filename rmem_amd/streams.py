"""HIP stream helper: a second stream that really runs beside the current one.

HIP maps streams onto a small pool of hardware queues (4 by default); two streams that share a
queue execute in order, however independent their work is (observed with rocprofv3: the
encoder-prefetch stream and the default stream both landed on queue 1 and the "concurrent"
encoder pass ran strictly before the frame it was meant to overlap).  The mapping cannot be
queried, so a candidate stream is accepted only after a short on-device spin on both streams
is seen to overlap.  Placement is used for speed only; ordering always comes from events."""
from __future__ import annotations

import torch

import time

_SPIN = 1_000_000      # cycles per probe kernel (~0.4 ms)


def _probe(cur: "torch.cuda.Stream", other: "torch.cuda.Stream") -> float:
    """Wall time (ms) of one spin on each stream, launched back to back."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(other):
        torch.cuda._sleep(_SPIN)
    with torch.cuda.stream(cur):
        torch.cuda._sleep(_SPIN)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def concurrent_stream(device, tries: int = 8) -> "torch.cuda.Stream":
    """A new stream on `device` whose kernels overlap with the current stream's (falls back to
    the last candidate if none of `tries` streams is seen to overlap)."""
    cur = torch.cuda.current_stream(device)
    keep = []
    with torch.cuda.device(device):
        _probe(cur, cur)                         # warm-up
        serial = min(_probe(cur, cur) for _ in range(2))   # both spins on one stream: the in-order time
        for _ in range(tries):
            cand = torch.cuda.Stream(device=device)
            keep.append(cand)                    # keep rejected ones alive so the pool advances
            _probe(cur, cand)                    # first use creates the queue
            if min(_probe(cur, cand) for _ in range(2)) < 0.75 * serial:
                return cand
    return keep[-1]
