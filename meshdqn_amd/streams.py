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


_PROBE_WARM = False


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
    global _PROBE_WARM
    if not _PROBE_WARM:          # the first launch of the probe kernel loads its code object (milliseconds): not part of the test
        _lib.check(lib.mdq_spin(1, 150 * 1024, 10, _lib.stream_ptr()), "mdq_spin")
        probe2.add_(1.0)
        _PROBE_WARM = True
    for st in (cand, other):     # (and a stream's first packet sets its queue up)
        with torch.cuda.stream(st):
            _lib.check(lib.mdq_spin(1, 150 * 1024, 10, _lib.stream_ptr()), "mdq_spin")
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
    c0.synchronize()                          # (another process on the GPU can hold `other` back past the end of `cand`'s chain:
    ok = c0.elapsed_time(c1) < 0.6            #  elapsed_time of an event that has not completed raises)
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


# ---------------------------------------------------------------------------- process-level stream roles
#
# The engines of this package need three streams that run beside each other: MAIN (the env step's kernel chain; rollouts
# move off the legacy default stream), FLOW (the IPCS leg of the S3 step) and OPT (the optimiser chain of the learning
# loop).  They are created ONCE per process and device, in this fixed order, the first time an engine asks: the
# stream -> hardware-queue mapping is round-robin in creation order, so every process that uses the package the same way
# gets the same mapping (it used to depend on how many environments / trainers a process had built before).  ONE probe
# per pair (`_overlaps`) verifies the roles; a pair that fails it is replaced through `concurrent_stream`.  The choice of
# a flow stream by timing real env steps (`VecEnv2DAirfoil.calibrate_streams`) remains as the second line of defence - a
# few pairs overlap only partly and pass every synthetic probe - but its verdict is kept here, so it runs once per
# process and main stream, not once per environment object.  `LOG` records which path was taken (bench.py prints it).
_ROLES = {}
_CALIBRATED = {}
LOG = []


def profiler_attached() -> bool:
    """True when the process runs under rocprofv3 / rocprof (their launchers preload a tool library and pass their options
    through ROCPROF_* / ROCP_* variables)."""
    import os
    for var in ("LD_PRELOAD", "HSA_TOOLS_LIB"):          # (HSA_TOOLS_LIB: the tool libraries of rocprof v1 / v2, roctracer)
        val = os.environ.get(var, "").lower()
        if "rocprof" in val or "roctracer" in val:
            return True
    return any(k.startswith(("ROCPROF_", "ROCP_TOOL", "ROCPROFILER_")) for k in os.environ)


def _cu_partition(dev):
    """Compute-unit ranges (lo, hi) of the three roles, or None (plain torch streams).  `MDQ_CU_PARTITION`:
      "full" (default)  every role gets a stream created with a CU mask that covers the WHOLE chip.  Such a stream owns a
                        hardware queue of its own instead of a slot in the round-robin pool of `GPU_MAX_HW_QUEUES`
                        queues, which is what removes the stream lottery: measured over repeated processes, plain
                        streams gave 123 k env-steps/s or - flow leg behind the main chain - 85 k depending on the
                        creation history, full-mask streams 123-125 k every time with every probe passing.
      "0"               plain torch streams (the round-2 behaviour: probes + calibration have to sort the pairs out).  Chosen
                        automatically under rocprofv3 (`profiler_attached`): the profiler of ROCm 7.2 dies at exit, before
                        writing its output, in a process that owns CU-masked queues (a four-line script reproduces it;
                        destroying the streams first cures the script but not a process with events / graphs on them)
      "1"               main on the first half of the driver's CU numbering, flow + optimiser chain on the second half
                        (the numbering goes round-robin over XCDs and shader engines: each half is half of every shader
                        engine).  Measured SLOWER (97-120 k): the chains do not suffer from sharing compute units.
      "lo:hi,lo:hi,lo:hi"  explicit ranges of main, flow, opt ("-" = a plain stream)."""
    import os
    spec = os.environ.get("MDQ_CU_PARTITION")
    if spec is None:
        spec = _CU_PARTITION_DEFAULT
        if profiler_attached():      # (an explicit MDQ_CU_PARTITION still wins)
            if not any(e.get("event") == "profiler attached" for e in LOG):
                LOG.append(dict(event="profiler attached", note="plain torch streams + calibration instead of CU-mask streams: "
                                "rocprofv3 (ROCm 7.2) dies at exit - before it writes its output - in a process that owns CU-masked queues"))
            spec = "0"
    if spec in ("", "0"):
        return None
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    if spec == "full":
        return dict(main=(0, ncu), flow=(0, ncu), opt=(0, ncu))
    if spec == "1":
        return dict(main=(0, ncu // 2), flow=(ncu // 2, ncu), opt=(ncu // 2, ncu))
    out = {}
    for k, part in zip(("main", "flow", "opt"), spec.split(",")):
        if part.strip() == "-":
            out[k] = None
        else:
            lo, hi = (int(v) for v in part.split(":"))
            if not 0 <= lo < hi <= ncu:
                raise ValueError(f"MDQ_CU_PARTITION: range {part!r} outside 0..{ncu}")
            out[k] = (lo, hi)
    if len(out) != 3:
        raise ValueError("MDQ_CU_PARTITION: three ranges (main, flow, opt) expected")
    return out


def _masked_stream(dev, lo, hi):
    """A torch stream object around a HIP stream restricted to compute units lo..hi-1 (kept for the process's lifetime)."""
    import ctypes as C
    from . import _lib
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (ncu + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(lo, hi):
        mask[i // 32] |= 1 << (i % 32)
    out = C.c_void_p()
    _lib.check(_lib.load().mdq_stream_create_cu_mask(mask, words, C.byref(out)), "mdq_stream_create_cu_mask")
    return torch.cuda.ExternalStream(out.value, device=dev)      # (kept for the process's lifetime)


_CU_PARTITION_DEFAULT = "full"


def new_flow_candidate(dev):
    """Another stream for the flow role (the calibration's candidates): same compute-unit range as the role's stream."""
    import os
    part = _cu_partition(dev)
    if part is not None and part["flow"] is not None:
        return _masked_stream(dev, *part["flow"])
    return torch.cuda.Stream(device=dev, priority=int(os.environ.get("MDQ_FLOW_PRIORITY", "0")))


def role_streams(device) -> dict:
    """dict(main=, flow=, opt=) of torch streams for `device` (created and verified on first use)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    r = _ROLES.get(idx)
    if r is not None:
        return r
    dev = torch.device("cuda", idx)
    with torch.cuda.device(dev):
        import os
        fp = int(os.environ.get("MDQ_FLOW_PRIORITY", "0"))     # (experiment knob: -1 = high priority for the flow stream)
        part = _cu_partition(dev)
        if part is not None:
            try:
                r = {k: (_masked_stream(dev, *part[k]) if part[k] is not None else torch.cuda.Stream(device=dev)) for k in ("main", "flow", "opt")}
            except Exception as e:       # (a runtime that refuses CU masks: plain streams, probes and calibration as before)
                LOG.append(dict(device=idx, event="CU-mask streams unavailable", error=str(e)))
                part = None
        if part is None:
            r = dict(main=torch.cuda.Stream(device=dev), flow=torch.cuda.Stream(device=dev, priority=fp), opt=torch.cuda.Stream(device=dev))
        how = {}
        for a, b in (("flow", "main"), ("opt", "main"), ("opt", "flow")):
            if _overlaps(r[a], r[b], dev):
                how[f"{a}/{b}"] = "probe ok"
            else:                                     # same queue / shared dispatcher pipe: another stream for role a
                with torch.cuda.stream(r["main"]):
                    r[a] = concurrent_stream(dev, [v for k, v in r.items() if k not in (a, "main")])
                how[f"{a}/{b}"] = "probe failed: replaced"
    _ROLES[idx] = r
    LOG.append(dict(device=idx, event="roles created in fixed order", probes=how,
                    cu_partition=None if part is None else {k: (list(v) if v else None) for k, v in part.items()}))
    return r


def roles_own_queues(device) -> bool:
    """True when the roles of `device` are CU-mask streams (a hardware queue each) and every pair passed its probe: there
    is nothing left for a calibration by timing to decide, and creating more such streams only uses up hardware queues
    (a process that had created ~25 of them ran its last measurements at half speed: the queues were time-sliced)."""
    idx = device_index(device)
    for e in LOG:
        if e.get("event") == "roles created in fixed order" and e.get("device") == idx:
            return e.get("cu_partition") is not None and all(v == "probe ok" for v in e.get("probes", {}).values())
    return False


def device_index(device) -> int:
    """The one normalisation of a device to its index for every per-device dictionary of this package ('cuda' without an
    index = the current device; round 3 keyed some tables on `torch.device(device).index`, which is None there)."""
    idx = torch.device(device).index
    return torch.cuda.current_device() if idx is None else int(idx)


def calibrated_flow_stream(device, main):
    """The flow stream `calibrate_streams` chose for rollouts on `main` in this process (None: not calibrated yet)."""
    for (idx, m), st in _CALIBRATED.items():
        if idx == device_index(device) and m == main:
            return st
    return None


def remember_flow_stream(device, main, flow, ms, how):
    _CALIBRATED[(device_index(device), main)] = flow
    LOG.append(dict(device=device_index(device), event="flow stream calibrated", how=how, ms_per_step=[round(v, 3) for v in ms]))
