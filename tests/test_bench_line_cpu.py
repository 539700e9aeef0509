"""CPU: the ONE stdout line of bench.py stays inside what the driver reads (round 5's line had grown to 22.9 KB and the
driver recorded `parsed: null`).  `compact_line` is fed the full result of a committed run (profiles/r05_bench.json = the
detail of that round) and must return the contract keys, numbers only for the side rates, no prose, <= 8 KB."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "rates", "steady_state_ms_per_step", "value_min", "value_max"}


def _full_result():
    res = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    res["config"].update(workload_short="ys930 (876 vertices / 1570 triangles), 128 envs per GPU, S3 env step", mesh="ys930")
    res["roofline"].update(floor_ms=0.2, launch_over_floor=2.1, traffic_source_short="profiles/r05_pmc_summary.json")
    return res


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, (list, tuple)):
        for v in o:
            yield from _strings(v)


def test_line_is_small_and_complete():
    res = _full_result()
    line = bench.compact_line(res, "/somewhere/bench_detail.json")
    txt = json.dumps(line)
    assert len(txt) <= bench.LINE_LIMIT == 8192 and len(txt) < 5000, len(txt)
    assert json.loads(txt) == line
    assert CONTRACT <= set(line)
    assert all(len(s) <= 128 for s in _strings(line)), [s for s in _strings(line) if len(s) > 128]
    assert line["value"] == float(f"{res['value']:.6g}") and line["n_gpus"] == 1 and line["detail"] == "bench_detail.json"
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_ms", "algorithmic_bytes_per_launch",
            "traffic_source", "floor_ms", "launch_over_floor"} <= set(roof)
    assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-6 and roof["traffic"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and cpu["parallel_cores"] == 12 and cpu["s2_ipcs"]["value"] > 0
    # one number per side rate
    assert set(line["rates"]) == set(res["rates"])
    assert all(v is None or isinstance(v, (int, float)) or v == "error" for v in line["rates"].values())
    assert line["rates"]["S3_north_star_step"] == line["value"]
    assert line["config"]["krylov_iters_per_ipcs_step"]["pressure_cg"] > 0 and "workload" in line["config"]


def test_a_failed_side_measurement_stays_a_word():
    res = _full_result()
    res["rates"]["C5_s3_refined_mesh"] = dict(error="RuntimeError('x' * 5000)", where=["a", "b"])
    res.pop("cpu_baseline")
    line = bench.compact_line(res, "d.json")
    assert line["rates"]["C5_s3_refined_mesh"] == "error" and "cpu_baseline" not in line and "c5_s3_krylov_iters" not in line
    assert len(json.dumps(line)) < 5000


def test_resource_table_is_tied_to_the_library(tmp_path):
    """libmeshdqn_hip.resources.json carries the sha256 of the .so it was built with; any other library is refused."""
    import pytest
    from meshdqn_amd import build as b
    if not os.path.exists(b.RESOURCES) or not os.path.exists(b.LIB):
        pytest.skip("library not built")
    table = b.library_resources(b.LIB)
    assert len(table) >= 55 and "_library_sha256" not in table
    other = tmp_path / "other.so"
    other.write_bytes(b"not the library")
    with pytest.raises(RuntimeError, match="does not describe"):
        b.library_resources(str(other))
