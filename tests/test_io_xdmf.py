"""CPU: the dependency-free XDMF3 / HDF5 reader (meshdqn_amd/io_xdmf.py, replacing `XDMFFile(mesh_file).read(mesh)` of
flow_solver.py:58-62) on one of the reference's own mesh data files (tests/golden/xdmf/, copied byte for byte by
tests/golden/make_mesh_fixtures.py): superblock v0, chunked datasets behind a v1 B-tree, deflate filter."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
XDMF = os.path.join(HERE, "golden", "xdmf", "ah93w145_0.14000_triangle.xdmf")


def test_reads_the_reference_mesh_file():
    from meshdqn_amd.io_xdmf import read_xdmf_mesh
    coords, cells = read_xdmf_mesh(XDMF)
    z = np.load(os.path.join(HERE, "golden", "ah93w145.npz"))
    assert coords.dtype == np.float64 and coords.shape == (797, 2) and cells.shape == (1431, 3)
    assert np.array_equal(coords, z["coords"]) and np.array_equal(cells, z["cells"])
    # 163 boundary vertices with the lowest ids, channel [-0.5, 3] x [-0.5, 0.5]
    assert coords[:, 0].min() == -0.5 and coords[:, 0].max() == 3.0 and abs(coords[:, 1]).max() == 0.5
    assert cells.min() == 0 and cells.max() == 796


def test_load_mesh_accepts_xdmf_and_npz():
    from meshdqn_amd.flow_solver import load_mesh
    c1, t1 = load_mesh(XDMF)
    c2, t2 = load_mesh(os.path.join(HERE, "golden", "ah93w145.npz"))
    assert np.array_equal(c1, c2) and np.array_equal(t1, t2)


def test_rejects_a_file_that_is_not_hdf5(tmp_path):
    import pytest
    from meshdqn_amd.io_xdmf import HDF5Error, read_xdmf_mesh
    bad = tmp_path / "m.h5"
    bad.write_bytes(b"not an hdf5 file" * 10)
    (tmp_path / "m.xdmf").write_text(open(XDMF).read().replace("ah93w145_0.14000_triangle.h5", "m.h5"))
    with pytest.raises(HDF5Error):
        read_xdmf_mesh(str(tmp_path / "m.xdmf"))
