"""GPU: the process's stream roles (meshdqn_amd/streams.py) and the CU-mask stream entry points of the C ABI."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_role_streams_are_full_mask_streams_that_overlap(lib_built):
    """Default `MDQ_CU_PARTITION=full`: main / flow / opt are three different streams created through
    `mdq_stream_create_cu_mask` with a mask over the whole chip (a hardware queue of their own each), every pair passes
    the overlap probe at the first try, and the log says so."""
    from meshdqn_amd import streams
    dev = torch.device("cuda", 0)
    r = streams.role_streams(dev)
    assert set(r) == {"main", "flow", "opt"}
    assert len({int(s.cuda_stream) for s in r.values()}) == 3
    created = [e for e in streams.LOG if e.get("event") == "roles created in fixed order" and e.get("device") == 0]
    assert created, streams.LOG
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    if created[-1]["cu_partition"] is not None:          # (None only with MDQ_CU_PARTITION=0 in the environment)
        assert created[-1]["cu_partition"] == {k: [0, ncu] for k in ("main", "flow", "opt")}
        assert set(created[-1]["probes"]) == {"flow/main", "opt/main", "opt/flow"}
    for a, b in (("flow", "main"), ("opt", "main"), ("opt", "flow")):
        # (a timing probe: one retry for a box that is busy with something else at that moment)
        assert streams._overlaps(r[a], r[b], dev) or streams._overlaps(r[a], r[b], dev), (a, b)
    # work on a role stream is ordinary torch work
    with torch.cuda.stream(r["flow"]):
        x = torch.arange(1000, device=dev, dtype=torch.float64).sum()
    r["flow"].synchronize()
    assert float(x) == 499500.0


def test_cu_mask_stream_entry_points(lib_built):
    """`mdq_stream_create_cu_mask` / `mdq_stream_destroy`: a stream restricted to half of the compute units runs kernels
    (the spin kernel with one 150 KB workgroup per CU of the half) and torch work through `ExternalStream`; bad arguments
    are refused with a message."""
    from meshdqn_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (ncu + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(ncu // 2):
        mask[i // 32] |= 1 << (i % 32)
    out = C.c_void_p()
    _lib.check(lib.mdq_stream_create_cu_mask(mask, words, C.byref(out)), "mdq_stream_create_cu_mask")
    assert out.value
    st = torch.cuda.ExternalStream(out.value, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        _lib.check(lib.mdq_spin(1, 150 * 1024, 10, C.c_void_p(out.value)), "mdq_spin")      # (code object load, queue set-up)
        torch.ones(8, device=dev).sum()
    st.synchronize()
    with torch.cuda.stream(st):
        e0.record()
        _lib.check(lib.mdq_spin(ncu // 2, 150 * 1024, 20_000, C.c_void_p(out.value)), "mdq_spin")     # 0.2 ms
        y = torch.ones(4096, device=dev).sum()
        e1.record()
    st.synchronize()
    assert float(y) == 4096.0
    assert 0.15 < e0.elapsed_time(e1) < 5.0
    del st
    torch.cuda.synchronize()
    _lib.check(lib.mdq_stream_destroy(C.c_void_p(out.value)), "mdq_stream_destroy")
    assert lib.mdq_stream_create_cu_mask(None, words, C.byref(out)) != 0
    assert lib.mdq_stream_create_cu_mask(mask, 0, C.byref(out)) != 0
    assert lib.mdq_stream_destroy(None) == 0
