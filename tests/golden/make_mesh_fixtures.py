"""Generate the mesh + known-answer fixtures under tests/golden/.

Run in the build container (where /root/reference exists):
    python tests/golden/make_mesh_fixtures.py

Outputs (data only, no reference source):
  * ys930.npz, ah93w145.npz  - coords (nv,2) f8, cells (nt,3) i4 exactly as
    stored in the reference's xdmf_files/*.h5 (file order, unsorted cells).
  * xdmf/ah93w145_0.14000_triangle.{xdmf,h5} - the smaller of the reference's two mesh DATA files, byte for byte (an
    XDMF3 descriptor + the meshio-written HDF5 container: superblock v0, chunked + deflate datasets), so that the
    dependency-free reader `meshdqn_amd/io_xdmf.py` is tested on the format the reference ships.
  * kat_rows.json - the two benchmark rows of the reference's
    training_results/benchmark_results/*.csv that correspond to the shipped
    meshes: NUM_COORDS, DRAG, LIFT after 5000 IPCS steps.
"""
import csv
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from meshdqn_amd.io_xdmf import read_xdmf_mesh  # noqa: E402

REF = os.environ.get("MESHDQN_REFERENCE", "/root/reference")

MESHES = {
    "ys930": ("xdmf_files/ys930_0.15000_triangle.xdmf",
              "training_results/benchmark_results/smooth_ys930_1.0_0.001_smooth_benchmark.csv"),
    "ah93w145": ("xdmf_files/ah93w145_0.14000_triangle.xdmf",
                 "training_results/benchmark_results/smooth_ah93w145_1.0_0.001_smooth_benchmark.csv"),
}


def main():
    import shutil
    os.makedirs(os.path.join(HERE, "xdmf"), exist_ok=True)
    for ext in ("xdmf", "h5"):
        shutil.copyfile(os.path.join(REF, f"xdmf_files/ah93w145_0.14000_triangle.{ext}"),
                        os.path.join(HERE, "xdmf", f"ah93w145_0.14000_triangle.{ext}"))
        os.chmod(os.path.join(HERE, "xdmf", f"ah93w145_0.14000_triangle.{ext}"), 0o644)
    kat = {}
    for name, (xdmf, bench_csv) in MESHES.items():
        coords, cells = read_xdmf_mesh(os.path.join(REF, xdmf))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), coords=coords, cells=cells)
        with open(os.path.join(REF, bench_csv)) as fh:
            rows = list(csv.DictReader(fh))
        hit = [r for r in rows if int(r["NUM_COORDS"]) == coords.shape[0]]
        assert len(hit) == 1, (name, len(hit))
        r = hit[0]
        kat[name] = dict(num_coords=int(r["NUM_COORDS"]), resolution=float(r["RESOLUTION"]),
                         time_s=float(r["TIME"]), drag=float(r["DRAG"]), lift=float(r["LIFT"]),
                         csv_line=rows.index(r) + 2, source=bench_csv,
                         solver_steps=5000, dt=1e-3, mu=1e-3, rho=1.0, smooth=True)
        # whole convergence table (sanity band for refined meshes)
        kat[name]["table"] = [[int(q["NUM_COORDS"]), float(q["DRAG"]), float(q["LIFT"])] for q in rows]
        print(name, coords.shape, cells.shape, kat[name]["drag"], kat[name]["lift"])
    with open(os.path.join(HERE, "kat_rows.json"), "w") as fh:
        json.dump(kat, fh, indent=1)


if __name__ == "__main__":
    main()
