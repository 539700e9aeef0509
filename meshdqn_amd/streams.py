"""HIP streams that really run beside each other.

HIP maps streams round-robin onto `GPU_MAX_HW_QUEUES` hardware queues (4 by default; the package sets 8), and two
streams that land on the same queue serialise - the flow leg of the S3 step or the optimiser chain of the learning
loop then runs BEHIND the smoothing kernel instead of beside it (measured: 2.8 instead of 1.8 ms per batched step,
in roughly one process out of ten, depending on how many streams the process had created before).  `concurrent_stream`
creates streams until one demonstrably overlaps with the given ones: a spin kernel with more one-per-CU workgroups than
CUs on the other stream (its queue's dispatcher stays busy), a tiny kernel on the candidate, and the candidate has to
finish at once.  (Queues that merely share a dispatcher pipe pass a test with single-workgroup kernels and still
serialise kernels that wait for free CUs, like the smoothing kernel and the pressure factorisation.)"""
from __future__ import annotations

import torch


def _overlaps(cand: torch.cuda.Stream, other: torch.cuda.Stream, device) -> bool:
    """True when a CHAIN of kernels on `cand` (six launches of one-workgroup-per-half-the-CUs spin kernels with 150 KB of
    LDS, 40 us each: the shape of the flow leg / optimiser chain) runs at its own pace while `other` is busy the way the
    main chain is busy (a 1 ms kernel of the same shape - the smoothing kernel - with the next kernel queued behind
    it).  Same queue: the chain waits for the long kernel.  Queues that share a dispatcher pipe: single kernels overlap,
    but every launch of the chain is held up behind the other queue's pending barrier packet (measured on the env
    step: 2.6 instead of 1.85 ms)."""
    from . import _lib
    lib = _lib.load()
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    probe2 = torch.zeros(1, device=device)
    half = max(1, torch.cuda.get_device_properties(device).multi_processor_count // 2)
    torch.cuda.synchronize(device)
    with torch.cuda.stream(other):
        c0.record(other)                      # (the clock starts in front of the long kernel: an event of `cand` would
        _lib.check(lib.mdq_spin(half, 150 * 1024, 100_000, _lib.stream_ptr()), "mdq_spin")        # 1 ms   only start
        probe2.add_(1.0)                      # the next packets of that queue                           when cand does)
        probe2.add_(1.0)
    with torch.cuda.stream(cand):
        for _ in range(6):
            _lib.check(lib.mdq_spin(half, 150 * 1024, 4_000, _lib.stream_ptr()), "mdq_spin")       # 40 us each
        c1.record(cand)
    c1.synchronize()
    ok = c0.elapsed_time(c1) < 0.6
    torch.cuda.synchronize(device)
    return ok


def concurrent_stream(device, others=(), attempts: int = 12) -> torch.cuda.Stream:
    """A new stream on `device` that runs concurrently with the current stream and with every stream in `others`
    (falls back to the last candidate if none of `attempts` passes, e.g. with GPU_MAX_HW_QUEUES=1)."""
    device = torch.device(device)
    with torch.cuda.device(device):
        refs = [torch.cuda.current_stream(device)] + [s for s in others if s is not None]
        cand = None
        rejected = []                         # (kept alive until the end: a destroyed stream's queue slot is handed out again)
        for _ in range(max(1, attempts)):
            cand = torch.cuda.Stream(device=device)
            if all(_overlaps(cand, r, device) for r in refs):
                return cand
            rejected.append(cand)
        return cand
