"""Dependency depth of one Gauss-Seidel smoothing sweep (DOLFIN smooth, index order) after CHAIN CONTRACTION.

A sweep updates the interior vertices in index order: v needs the NEW position of its interior neighbours w < v.  The mesh
numbering runs along rings (the 113 consecutively numbered vertices around the airfoil are one), so the DAG is deep (113
levels on ys930) but most of its depth is chains v, v+1, v+2, ... in which every vertex's only unresolved same-sweep
dependency is its predecessor.  In full-step mode the update along such a chain is the affine recurrence
x_v = (x_{v-1} + c_v) / k_v, which a parallel scan over (a, b) pairs solves in log2(length) steps.

This script segments the interior vertices into maximal SEGMENTS of consecutive indices [s, e] such that every member
v > s is adjacent to v - 1, and no member depends (directly, through a lower-numbered interior neighbour outside the
segment) on something that depends on an earlier member of the same segment - i.e. every outside dependency w of a member
must be schedulable before the segment starts: level(w) < level(segment).  Greedy: extend while that holds.  Reports the
number of segments, the DAG depth over segments, the length histogram, and a cost model for one wave walking the
segments (list-scheduled, several short segments side by side in one pass of 64 lanes).

    python tools/chain_depth.py [tools/micro/data/*.bin]
"""
import glob
import os
import sys

import numpy as np


def load(path):
    with open(path, "rb") as f:
        nv, nt = np.fromfile(f, np.int32, 2)
        coords = np.fromfile(f, np.float64, 2 * nv).reshape(nv, 2)
        cells = np.fromfile(f, np.int32, 3 * nt).reshape(nt, 3)
    return coords, cells


def analyse(cells, nv):
    nbr = [set() for _ in range(nv)]
    cnt = {}
    for a, b, c in cells:
        for u, w in ((a, b), (b, c), (a, c)):
            nbr[u].add(w)
            nbr[w].add(u)
            e = (min(u, w), max(u, w))
            cnt[e] = cnt.get(e, 0) + 1
    boundary = np.zeros(nv, bool)
    for (u, w), n in cnt.items():
        if n == 1:
            boundary[u] = boundary[w] = True
    interior = [v for v in range(nv) if not boundary[v] and nbr[v]]
    isint = np.zeros(nv, bool)
    isint[interior] = True
    # plain level schedule (what bounds the per-vertex design)
    level = np.zeros(nv, np.int64)
    for v in interior:
        lower = [w for w in nbr[v] if w < v and isint[w]]
        level[v] = 1 + max((level[w] for w in lower), default=0)
    depth_plain = int(level[interior].max())
    # greedy segmentation; seg_level[v] = level of the segment holding v
    seg_of = -np.ones(nv, np.int64)
    segs = []            # (start, end, level)
    i = 0
    while i < len(interior):
        s = interior[i]
        lvl = 1 + max((segs[seg_of[w]][2] for w in nbr[s] if w < s and isint[w]), default=0)
        e = s
        j = i + 1
        while j < len(interior):
            v = interior[j]
            if v != e + 1 or (v - 1) not in nbr[v]:
                break
            outside = [w for w in nbr[v] if w < s and isint[w]]
            inside = [w for w in nbr[v] if s <= w < v - 1 and isint[w]]
            if inside:            # depends on an earlier member other than its predecessor: not a simple recurrence
                break
            need = 1 + max((segs[seg_of[w]][2] for w in outside), default=0)
            lvl = max(lvl, need)  # the whole segment waits for its latest outside dependency
            e = v
            j += 1
        k = len(segs)
        for v in interior[i:j]:
            seg_of[v] = k
        segs.append((s, e, lvl))
        i = j
    # raising a segment's level for a late member may have been unnecessary for the early members, but correctness holds;
    # recompute exact levels over the segment DAG
    lv = []
    for k, (s, e, _) in enumerate(segs):
        deps = {seg_of[w] for v in range(s, e + 1) for w in nbr[v] if w < s and isint[w]}
        lv.append(1 + max((lv[d] for d in deps), default=0))
    lens = np.array([e - s + 1 for s, e, _ in segs])
    depth_seg = max(lv)
    # cost model: one wave, per DAG level the segments of that level run side by side; a pass handles up to 64 chain
    # members (scan width = next power of two of the longest segment in the pass); longer segments take ceil(len / 64)
    # sequential scans
    passes = 0
    scan_steps = 0
    for L in range(1, depth_seg + 1):
        ls = sorted((lens[k] for k in range(len(segs)) if lv[k] == L), reverse=True)
        # first-fit decreasing into 64-lane passes
        bins = []
        for n in ls:
            for c in range(int(np.ceil(n / 64))):
                m = min(64, n - 64 * c)
                for b in bins:
                    if b[0] + m <= 64 and c == 0:
                        b[0] += m
                        b[1] = max(b[1], m)
                        break
                else:
                    bins.append([m, m])
        passes += len(bins)
        scan_steps += sum(int(np.ceil(np.log2(max(b[1], 2)))) for b in bins)
    return dict(n_int=len(interior), depth_plain=depth_plain, n_segs=len(segs), depth_seg=depth_seg, passes=passes,
                scan_steps=scan_steps, longest=int(lens.max()), singles=int((lens == 1).sum()),
                hist=np.bincount(np.minimum(lens, 20)).tolist())


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    paths = sys.argv[1:] or sorted(glob.glob(os.path.join(here, "micro", "data", "*.bin")))
    for p in paths:
        coords, cells = load(p)
        r = analyse(cells, len(coords))
        print(f"{os.path.basename(p):20s} interior {r['n_int']:4d}  levels {r['depth_plain']:4d}  segments {r['n_segs']:4d} "
              f"(singles {r['singles']}, longest {r['longest']})  segment-DAG depth {r['depth_seg']:3d}  "
              f"64-lane passes {r['passes']:3d}  scan steps {r['scan_steps']:4d}")
        print("    length histogram (1..19, >=20):", r["hist"][1:])
