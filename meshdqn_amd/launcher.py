"""Rank launcher shared by `bench.py --gpus N` and `train.py --gpus N`: the parent process - which never imports torch or
touches a GPU - starts N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their
environment, 127.0.0.1 rendezvous) and WATCHES ALL OF THEM: the first rank that exits non-zero (or the launch timeout) tears the
job down within seconds - only processes this parent started are terminated - and the failed rank is named with its last
output lines (the reference's Ray trainer ends the job when a worker dies as well: airfoil_dqn.py:508-514).  Never a re-exec
of a process that has initialised the GPU."""
import os
import sys
import time


def _relay(stream, prefix, sink, tail, keep=None):
    """Reader thread of one child pipe: every line goes to `sink` with the rank prefix; the last lines are kept in `tail`
    (what the parent prints when that rank fails) and - rank 0's stdout - all lines in `keep`."""
    for line in iter(stream.readline, ""):
        if keep is not None:
            keep.append(line)
        else:
            sink.write(prefix + line)
            sink.flush()
        tail.append(line)
        del tail[:-30]
    stream.close()


def _stop_children(procs, grace=5.0):
    """Terminate exactly the processes this parent started (SIGTERM, then SIGKILL after `grace` seconds)."""
    import subprocess
    for p_ in procs:
        if p_.poll() is None:
            p_.terminate()
    t_end = time.monotonic() + grace
    for p_ in procs:
        try:
            p_.wait(timeout=max(0.05, t_end - time.monotonic()))
        except subprocess.TimeoutExpired:
            p_.kill()
            p_.wait()


def run_watched(cmds, timeout_s, poll=0.2):
    """Start one child per (argv, env) entry and watch ALL of them: returns (failed, why, out0, tails, exit codes) where
    `failed` is None when every child exited 0, else the index of the first child that exited non-zero (or that was still
    running at the timeout) - in which case the other children have been terminated.  stdout of child 0 is collected in
    `out0`, everything else is relayed to stderr with a `[rank r]` prefix; `tails[r]` = the last lines of child r."""
    import subprocess
    import threading
    procs, threads, tails, out0 = [], [], [], []
    for r, (argv_r, env_r) in enumerate(cmds):
        p_ = subprocess.Popen(argv_r, env=env_r, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p_)
        tails.append([])
        for stream, keep in ((p_.stdout, out0 if r == 0 else None), (p_.stderr, None)):
            th = threading.Thread(target=_relay, args=(stream, f"[rank {r}] ", sys.stderr, tails[r], keep), daemon=True)
            th.start()
            threads.append(th)
    deadline = time.monotonic() + timeout_s
    failed, why = None, ""
    while True:
        rcs = [p_.poll() for p_ in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed, why = bad[0], f"exited with code {rcs[bad[0]]}"
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed = next(r for r, rc in enumerate(rcs) if rc is None)
            why = "still running at the launch timeout (MDQ_LAUNCH_TIMEOUT)"
            break
        time.sleep(poll)
    if failed is not None:
        _stop_children(procs)
    for th in threads:
        th.join(timeout=5)
    return failed, why, out0, tails, [p_.returncode for p_ in procs]



def start_ranks(script, argv, n, env=None, timeout_s=None, tag="launcher"):
    """Run `python script argv` as n ranks on this node; returns (exit code, stdout lines of rank 0)."""
    import socket
    env = dict(os.environ if env is None else env)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=env.get("MASTER_PORT", str(port)))
    cmds = [([sys.executable, os.path.abspath(script)] + list(argv), dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    if timeout_s is None:
        timeout_s = float(os.environ.get("MDQ_LAUNCH_TIMEOUT", "1700"))
    failed, why, out0, tails, rcs = run_watched(cmds, timeout_s)
    if failed is not None:
        sys.stderr.write(f"[{tag}] rank {failed} of {n} {why}; the other ranks were terminated "
                         f"(exit codes {rcs}): no result line.  Last output of rank {failed}:\n")
        sys.stderr.write("".join("    " + l for l in tails[failed][-15:]))
        return 1, out0
    return 0, out0


def visible_gpus():
    """GPUs the kernel driver exposes (kfd topology nodes with SIMDs), without touching the HIP runtime; None if the
    topology cannot be read (the ranks then find out themselves)."""
    import glob
    n, seen = 0, False
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(l.split()[:2] for l in open(f).read().splitlines() if len(l.split()) >= 2)
        except OSError:
            continue
        seen = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n if (seen and n > 0) else None       # (0 GPU nodes found: treat the topology as unreadable)
