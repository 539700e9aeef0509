"""CPU: the parts of meshdqn_amd/streams.py that need no GPU - profiler detection (under rocprofv3 the stream roles fall
back to plain torch streams: the profiler of ROCm 7.2 dies at exit in a process that owns CU-mask queues)."""


def test_profiler_detection_from_the_environment(monkeypatch):
    from meshdqn_amd import streams
    for k in list(__import__("os").environ):
        if k.startswith(("ROCPROF_", "ROCP_TOOL", "ROCPROFILER_")):
            monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LD_PRELOAD", "/usr/local/lib/libsomething_else.so")
    assert not streams.profiler_attached()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert streams.profiler_attached()
    monkeypatch.setenv("LD_PRELOAD", "")
    monkeypatch.delenv("HSA_TOOLS_LIB", raising=False)
    assert not streams.profiler_attached()
    monkeypatch.setenv("HSA_TOOLS_LIB", "/opt/rocm/lib/librocprofiler64.so.1")
    assert streams.profiler_attached()
    monkeypatch.delenv("HSA_TOOLS_LIB")
    monkeypatch.setenv("ROCPROF_KERNEL_TRACE", "1")
    assert streams.profiler_attached()
    monkeypatch.delenv("ROCPROF_KERNEL_TRACE")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "x.so")
    assert streams.profiler_attached()


def test_roles_own_queues_reads_the_log(monkeypatch):
    """`roles_own_queues` is true only for CU-mask roles whose probes all passed (then no calibration by timing runs)."""
    from meshdqn_amd import streams
    monkeypatch.setattr(streams, "LOG", [])
    assert not streams.roles_own_queues("cuda:0")
    streams.LOG.append(dict(device=0, event="roles created in fixed order", probes={"flow/main": "probe ok", "opt/main": "probe ok"},
                            cu_partition=None))
    assert not streams.roles_own_queues("cuda:0")
    streams.LOG[:] = [dict(device=0, event="roles created in fixed order", probes={"flow/main": "probe failed: replaced"},
                           cu_partition={"main": [0, 256], "flow": [0, 256], "opt": [0, 256]})]
    assert not streams.roles_own_queues("cuda:0")
    streams.LOG[:] = [dict(device=0, event="roles created in fixed order", probes={"flow/main": "probe ok", "opt/flow": "probe ok"},
                           cu_partition={"main": [0, 256], "flow": [0, 256], "opt": [0, 256]})]
    assert streams.roles_own_queues("cuda:0")
    assert not streams.roles_own_queues("cuda:1")


def test_bench_profiler_test_agrees_with_the_package(monkeypatch):
    """bench.py's CPU-baseline interpreters decide `skip the multi-process legs under a profiler` with the same environment
    test as the package's stream roles (the two guards had drifted apart)."""
    import importlib.util
    import os
    from meshdqn_amd import streams
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in list(os.environ):
        if k.startswith(("ROCPROF_", "ROCP_TOOL", "ROCPROFILER_")):
            monkeypatch.delenv(k, raising=False)
    for var, val in (("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"), ("LD_PRELOAD", "libfoo.so"),
                     ("HSA_TOOLS_LIB", "librocprofiler64.so.1"), ("ROCPROF_KERNEL_TRACE", "1"), ("ROCP_TOOL_LIBRARIES", "x.so")):
        monkeypatch.setenv(var, val)
        assert bench._profiler_attached() == streams.profiler_attached(), (var, val)
        monkeypatch.delenv(var)
