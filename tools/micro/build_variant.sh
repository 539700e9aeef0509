#!/bin/bash
# Build a variant of the library (extra -D flags) into tools/micro/bin/libmdq_NAME.so (git-ignored; travels with gpurun);
# use it with MDQ_LIB_PATH=tools/micro/bin/libmdq_NAME.so.   tools/micro/build_variant.sh NAME -DMDQ_TOPO_TRACE ...
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/bin"
name="$1"; shift
cd "$ROOT"
python - "$name" "$@" <<'PY'
import os, subprocess, sys
from meshdqn_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join("tools", "micro", "bin", f"libmdq_{name}.so")
cmd = [b._hipcc(), "-O3", "-std=c++17", f"--offload-arch={b.ARCH}", "-fPIC", "-shared", "-I", "include", "-I", b.CSRC] + flags
cmd += [os.path.join(b.CSRC, s) for s in b.SOURCES] + ["-o", out]
subprocess.run(cmd, check=True)
print("built", out)
PY
