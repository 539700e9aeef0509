"""CPU: the sibling watchdog of `bench.py --gpus N` (`run_watched`): every child is polled; the first one that exits
non-zero tears the job down within seconds (the others are terminated), names the failed rank and keeps its last
output lines - instead of rank 0 sitting in a collective until the c10d timeout (reference: the Ray trainer ends the job
when a worker dies, airfoil_dqn.py:508-514)."""
import importlib.util
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _py(code):
    return ([sys.executable, "-c", code], dict(os.environ))


def test_all_children_succeed_and_rank0_stdout_is_collected(capfd):
    bench = _bench()
    cmds = [_py("import sys; print('{\"rank\": 0}'); sys.stderr.write('hello from zero\\n')")] + [
        _py(f"import sys; print('noise of rank {r}')") for r in range(1, 8)]
    failed, why, out0, tails, rcs = bench.run_watched(cmds, 60.0)
    assert failed is None and rcs == [0] * 8
    assert [l.strip() for l in out0] == ['{"rank": 0}']
    err = capfd.readouterr().err
    assert "[rank 0] hello from zero" in err and "[rank 5] noise of rank 5" in err      # prefixed relay


def test_a_failing_rank_ends_the_job_within_seconds(capfd):
    """Rank 3 of 8 dies one second after the start while the others would run for ten minutes (a rank stuck in a barrier)."""
    bench = _bench()
    sleeper = "import time; time.sleep(600)"
    cmds = [_py(sleeper) for _ in range(8)]
    cmds[3] = _py("import sys, time; time.sleep(1); sys.stderr.write('RuntimeError: no GPU for me\\n'); sys.exit(3)")
    t0 = time.monotonic()
    failed, why, out0, tails, rcs = bench.run_watched(cmds, 600.0)
    took = time.monotonic() - t0
    assert failed == 3 and "code 3" in why and took < 15.0, (failed, why, took)
    assert rcs[3] == 3 and all(rc is not None and rc != 0 for rc in rcs)             # every sibling was stopped
    assert any("no GPU for me" in l for l in tails[3])


def test_a_child_that_ignores_sigterm_is_killed(capfd):
    bench = _bench()
    stubborn = "import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); print('up', flush=True); time.sleep(600)"
    cmds = [_py(stubborn), _py("import sys, time; time.sleep(1.5); sys.exit(1)")]
    t0 = time.monotonic()
    failed, why, out0, tails, rcs = bench.run_watched(cmds, 600.0)
    assert failed == 1 and time.monotonic() - t0 < 20.0
    assert rcs[0] == -9                                                              # SIGKILL after the grace period


def test_launch_timeout_names_a_rank_that_is_still_running():
    bench = _bench()
    failed, why, out0, tails, rcs = bench.run_watched([_py("import time; time.sleep(600)")], 1.0)
    assert failed == 0 and "timeout" in why and rcs[0] != 0


def test_gpus_flag_on_a_box_without_gpus_fails_fast_with_the_rank_named():
    """The whole path through `python bench.py --gpus 2` on this CPU box: the ranks find no GPU and exit 2; the parent
    reports which rank failed, prints no JSON line and exits non-zero (no c10d timeout involved)."""
    try:
        import torch
        if torch.cuda.is_available():
            import pytest
            pytest.skip("CPU-box case")
    except ImportError:
        pass
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MDQ_SHARE_GPU"] = "1"
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--no-cpu-baseline"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 1 and time.monotonic() - t0 < 120
    assert "exited with code 2" in out.stderr and "no result line" in out.stderr and "[rank " in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
