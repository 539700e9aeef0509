"""Build the gfx950 HIP library in-tree (`meshdqn_amd/libmeshdqn_hip.so`).

hipcc cross-compiles for gfx950 without a GPU present; the built .so is
git-ignored but travels with the working tree to the GPU box.

The library is several translation units compiled in parallel (objects cached
under `build/obj`, recompiled when their sources or any header change) and
linked with `-fvisibility=hidden`: only the `MDQ_API` entry points of
`include/meshdqn_hip.h` are exported.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmeshdqn_hip.so")
RESOURCES = os.path.join(HERE, "libmeshdqn_hip.resources.json")   # per-kernel VGPRs / scratch / occupancy of THIS build (bench.py)
OBJ = os.path.join(ROOT, "build", "obj")
ARCH = "gfx950"

# translation unit -> (source, extra flags, the .hip files it includes besides itself).  mdq_ipcs.hip is compiled in
# parts (-DMDQ_IPCS_PART=k: each part instantiates the kernels of some operator modes and their launchers; part 0 also
# holds the entry points) because it alone took a minute as one unit.
UNITS = {
    "ipcs0": ("mdq_ipcs.hip", ["-DMDQ_IPCS_PART=0"], []),
    "ipcs1": ("mdq_ipcs.hip", ["-DMDQ_IPCS_PART=1"], []),
    "ipcs2": ("mdq_ipcs.hip", ["-DMDQ_IPCS_PART=2"], []),
    "ipcs3": ("mdq_ipcs.hip", ["-DMDQ_IPCS_PART=3"], []),
    "pressure_factor": ("mdq_pressure_factor.hip", [], []),
    "tilemaps": ("mdq_tilemaps.hip", [], []),
    "gcn": ("mdq_tu_gcn.hip", [], ["mdq_gcn.hip", "mdq_gcn_train.hip"]),
    "replay": ("mdq_replay.hip", [], []),
    "mesh": ("mdq_mesh.hip", [], []),
    "smooth": ("mdq_tu_smooth.hip", [], ["mdq_smooth_big.hip", "mdq_smooth.hip", "mdq_smooth_linear.hip"]),
    "topology": ("mdq_topology.hip", [], []),
    "remesh": ("mdq_remesh.hip", [], []),
    "host_mesh": ("mdq_host_mesh.hip", [], []),
}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def _headers() -> list:
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "meshdqn_hip.h"))
    return hs


def _flags(verbose: bool) -> list:
    fl = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fvisibility=hidden",
          "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Rpass-analysis=kernel-resource-usage"]
    return fl + os.environ.get("MDQ_CFLAGS", "").split()


def parse_resources(log: str) -> dict:
    """kernel (demangled, without arguments) -> dict(vgprs, agprs, sgprs, scratch_bytes_per_lane, occupancy, lds_bytes) from the
    `-Rpass-analysis=kernel-resource-usage` remarks of a compile."""
    import re
    rows, cur = [], None
    for line in log.splitlines():
        m = re.search(r"remark: (?:\S+ )?\s*(Function Name|Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]|SGPRs): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k in ("Function Name", "Name"):
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0]] = v
    if not rows:
        return {}
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    out = {}
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        out[n] = dict(vgprs=int(r.get("VGPRs", -1)), agprs=int(r.get("AGPRs", -1)), sgprs=int(r.get("SGPRs", -1)),
                      scratch_bytes_per_lane=int(r.get("ScratchSize", -1)), occupancy=int(r.get("Occupancy", -1)),
                      lds_bytes=int(r.get("LDS", -1)))
    return out


def _unit_key(name: str, flags: list) -> str:
    src, extra, incs = UNITS[name]
    h = hashlib.sha256()
    h.update(" ".join(flags + extra).encode())
    for p in [os.path.join(CSRC, src)] + [os.path.join(CSRC, i) for i in incs] + _headers():
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    return h.hexdigest()


def _stale() -> bool:
    if os.environ.get("MDQ_LIB_PATH"):      # an explicitly chosen library: nothing to build
        return False
    if not os.path.exists(LIB):
        return True
    try:        # built with other flags (MDQ_CFLAGS experiments)?
        if open(os.path.join(OBJ, "flags.txt")).read() != " ".join(_flags(False)):
            return True
    except OSError:     # no record of the flags (a tree copied without build/): stale only if flags were asked for
        if os.environ.get("MDQ_CFLAGS"):
            return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(ROOT, "include", "meshdqn_hip.h"))
    return any(os.path.getmtime(p) > t for p in deps if os.path.isfile(p))


def declared_symbols() -> list:
    """The entry points `include/meshdqn_hip.h` declares (MDQ_API)."""
    import re
    with open(os.path.join(ROOT, "include", "meshdqn_hip.h")) as f:
        return sorted(set(re.findall(r"^MDQ_API\s+[A-Za-z0-9_\* ]*?\b(mdq_[a-z0-9_]+)\s*\(", f.read(), flags=re.M)))


def _version_script() -> str:
    """Linker version script: the header's entry points (and the *_trace_host helpers of the -DMDQ_*_TRACE debug builds) are
    the library's only dynamic symbols - kernel stubs, libstdc++ instantiations and __hip_cuid_* stay local."""
    path = os.path.join(OBJ, "exports.map")
    body = "{\n  global:\n" + "".join(f"    {n};\n" for n in declared_symbols()) + "    mdq_*_trace_host;\n  local:\n    *;\n};\n"
    with open(path, "w") as f:
        f.write(body)
    return path


def _compile(name: str, flags: list, force: bool) -> tuple:
    src, extra, _ = UNITS[name]
    obj = os.path.join(OBJ, name + ".o")
    keyf = obj + ".key"
    key = _unit_key(name, flags)
    if not force and os.path.exists(obj) and os.path.exists(keyf) and os.path.exists(obj + ".log") and open(keyf).read() == key:
        return name, obj, open(obj + ".log").read()
    cmd = [_hipcc()] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", obj + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src} ({name}):\n{res.stdout}{res.stderr}")
    os.replace(obj + ".tmp", obj)
    with open(obj + ".log", "w") as f:
        f.write(res.stderr)
    with open(keyf, "w") as f:
        f.write(key)
    return name, obj, res.stderr


def build(force: bool = False, verbose: bool = False) -> str:
    """`force` recompiles every unit; otherwise only units whose sources / headers / flags changed."""
    if not force and not verbose and not _stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    flags = _flags(verbose)
    jobs = int(os.environ.get("MDQ_BUILD_JOBS", "0")) or min(len(UNITS), os.cpu_count() or 4)
    objs, logs = {}, {}
    with cf.ThreadPoolExecutor(jobs) as ex:
        futs = [ex.submit(_compile, n, flags, force) for n in UNITS]
        for f in futs:
            try:
                n, o, log = f.result()
            except RuntimeError as e:
                sys.stderr.write(str(e))
                raise RuntimeError("hipcc failed building libmeshdqn_hip.so") from None
            objs[n], logs[n] = o, log
    if verbose:
        for n in UNITS:
            sys.stderr.write(logs[n])
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fvisibility=hidden",
           f"-Wl,--version-script={_version_script()}"] + [objs[n] for n in UNITS] + ["-o", LIB + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed linking libmeshdqn_hip.so")
    os.replace(LIB + ".tmp", LIB)
    with open(os.path.join(OBJ, "flags.txt"), "w") as f:
        f.write(" ".join(flags))
    import json
    table = {}
    for n in UNITS:
        table.update(parse_resources(logs[n]))
    # stamped with the sha256 of the library it describes: `library_resources` refuses the table for any other .so
    table["_library_sha256"] = file_sha256(LIB)
    with open(RESOURCES, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    return LIB


def file_sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def library_resources(lib_path: str) -> dict:
    """The per-kernel resource table of `lib_path` - only if the table beside the in-tree library was written by the build
    that linked exactly that file (sha256 stamp); raises otherwise (a prebuilt, variant or MDQ_LIB_PATH library has no
    table of its own)."""
    import json
    with open(RESOURCES) as f:
        table = json.load(f)
    stamp = table.pop("_library_sha256", None)
    if stamp is None or stamp != file_sha256(lib_path):
        raise RuntimeError(f"{os.path.basename(RESOURCES)} does not describe {lib_path} (sha256 stamp "
                           f"{'missing' if stamp is None else 'differs'}): rebuild with `python -m meshdqn_amd.build --force`")
    return table


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
