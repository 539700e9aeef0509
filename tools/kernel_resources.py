"""Table of the per-kernel resource report (`-Rpass-analysis=kernel-resource-usage`) of a library build.
usage: python -m meshdqn_amd.build --force -v 2> /tmp/res.txt ; python tools/kernel_resources.py /tmp/res.txt"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark: (?:\S+ )?\s*(Function Name|Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k in ("Function Name", "Name"):
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'occ':>4s}")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)
    print(f"{n[:70]:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('ScratchSize','?'):>8s} {r.get('Occupancy','?'):>4s}")
