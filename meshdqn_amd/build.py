"""Build the gfx950 HIP library in-tree (`meshdqn_amd/libmeshdqn_hip.so`).

hipcc cross-compiles for gfx950 without a GPU present; the built .so is
git-ignored but travels with the working tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmeshdqn_hip.so")
SOURCES = ["mdq_lib.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def _stale() -> bool:
    if os.environ.get("MDQ_LIB_PATH"):      # an explicitly chosen library: nothing to build
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "meshdqn_hip.h"))
    return any(os.path.getmtime(p) > t for p in deps if os.path.isfile(p))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    cmd = [_hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
           "-I", os.path.join(HERE, "..", "include"), "-I", CSRC]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    cmd += os.environ.get("MDQ_CFLAGS", "").split()
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libmeshdqn_hip.so")
    if verbose:
        sys.stderr.write(res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
