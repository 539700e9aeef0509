#!/bin/bash
# Build a variant of the library (extra -D flags) into tools/micro/bin/libmdq_NAME.so (git-ignored; travels with gpurun);
# use it with MDQ_LIB_PATH=tools/micro/bin/libmdq_NAME.so.   tools/micro/build_variant.sh NAME -DMDQ_TOPO_TRACE ...
# Same translation units as meshdqn_amd/build.py, objects under build/obj_NAME; linked WITHOUT the version script: the
# trace builds export their read-back entry points (mdq_*_trace_host; default visibility: some carry no MDQ_API) beside the header's symbols.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/bin"
name="$1"; shift
cd "$ROOT"
python - "$name" "$@" <<'PY'
import concurrent.futures as cf, os, subprocess, sys
from meshdqn_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join("tools", "micro", "bin", f"libmdq_{name}.so")
objd = os.path.join("build", f"obj_{name}")
os.makedirs(objd, exist_ok=True)
base = ["-O3", "-std=c++17", f"--offload-arch={b.ARCH}", "-fPIC", "-I", "include", "-I", b.CSRC] + flags
def comp(unit):
    src, extra, _ = b.UNITS[unit]
    obj = os.path.join(objd, unit + ".o")
    subprocess.run([b._hipcc()] + base + extra + ["-c", os.path.join(b.CSRC, src), "-o", obj], check=True)
    return obj
with cf.ThreadPoolExecutor(min(len(b.UNITS), os.cpu_count() or 4)) as ex:
    objs = list(ex.map(comp, b.UNITS))
subprocess.run([b._hipcc(), f"--offload-arch={b.ARCH}", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
print("built", out)
PY
