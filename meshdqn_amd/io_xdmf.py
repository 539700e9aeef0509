"""XDMF3 + HDF5 mesh reader with no h5py / DOLFIN dependency.

Replaces the `XDMFFile(mesh_file).read(mesh)` call of the reference
(`flow_solver.py:58-62`).  The reference's mesh files
(`xdmf_files/*_triangle.{xdmf,h5}`) are XDMF3 descriptors whose DataItems
point into an HDF5 file written by meshio: superblock version 0, datasets in
the root group, chunked layout (v1 B-tree chunk index), deflate filter.

Only the subset of HDF5 that those files use is implemented:
  * superblock v0, 8-byte offsets/lengths
  * old-style groups (symbol table: v1 B-tree node type 0 + local heap + SNOD)
  * v1 object headers with messages: dataspace(1), datatype(3), layout(8),
    filter pipeline(11), continuation(16)
  * layout class 1 (contiguous) and 2 (chunked, v1 B-tree node type 1)
  * filters: deflate(1), shuffle(2)
  * datatypes: fixed-point and IEEE float, little endian
"""
from __future__ import annotations

import os
import re
import struct
import zlib

import numpy as np

_UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5Error(RuntimeError):
    pass


class _H5File:
    def __init__(self, path: str):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise HDF5Error(f"{path}: not an HDF5 file")
        if b[8] != 0:
            raise HDF5Error(f"{path}: superblock version {b[8]} unsupported (need 0)")
        if b[13] != 8 or b[14] != 8:
            raise HDF5Error("only 8-byte offsets/lengths supported")
        # superblock v0: 8 sig, 8 version bytes, 2 leaf k, 2 internal k, 4 flags,
        # then base, free-space, eof, driver addresses, then root symbol table entry
        self.base = struct.unpack_from("<Q", b, 24)[0]
        root_entry = 24 + 4 * 8
        self.root = self._symtab_entry(root_entry)

    # --- low-level -------------------------------------------------------
    def _symtab_entry(self, off):
        name_off, hdr_addr, cache_type = struct.unpack_from("<QQI", self.buf, off)
        scratch = self.buf[off + 24: off + 40]
        return dict(name_off=name_off, hdr=hdr_addr, cache=cache_type, scratch=scratch)

    def _messages(self, hdr_addr):
        """Yield (type, bytes) for every message of a v1 object header."""
        b = self.buf
        ver, _, nmsg, _refc, hsize = struct.unpack_from("<BBHII", b, hdr_addr)
        if ver != 1:
            raise HDF5Error(f"object header version {ver} unsupported")
        blocks = [(hdr_addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, pos)
                body = b[pos + 8: pos + 8 + msize]
                if mtype == 0x10:  # continuation
                    caddr, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((caddr, clen))
                out.append((mtype, body))
                pos += 8 + msize
        return out

    def _group_children(self, entry):
        """name -> object header address for an old-style group."""
        btree_addr = heap_addr = None
        if entry["cache"] == 1:
            btree_addr, heap_addr = struct.unpack_from("<QQ", entry["scratch"], 0)
        else:
            for mtype, body in self._messages(entry["hdr"]):
                if mtype == 0x11:
                    btree_addr, heap_addr = struct.unpack_from("<QQ", body, 0)
        if btree_addr is None:
            raise HDF5Error("group without symbol table message")
        b = self.buf
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise HDF5Error("bad local heap")
        heap_data = struct.unpack_from("<Q", b, heap_addr + 24)[0]
        names = {}

        def walk(addr):
            if b[addr:addr + 4] == b"TREE":
                ntype, level, nused = struct.unpack_from("<BBH", b, addr + 4)
                pos = addr + 8 + 16  # skip left/right siblings
                # keys are 8 bytes (heap offset), children 8 bytes
                for i in range(nused):
                    child = struct.unpack_from("<Q", b, pos + 8 + i * 16)[0]
                    walk(child)
            elif b[addr:addr + 4] == b"SNOD":
                nsym = struct.unpack_from("<H", b, addr + 6)[0]
                for i in range(nsym):
                    e = self._symtab_entry(addr + 8 + i * 40)
                    s = heap_data + e["name_off"]
                    name = b[s:b.index(b"\0", s)].decode()
                    names[name] = e["hdr"]
            else:
                raise HDF5Error("bad group b-tree node")

        walk(btree_addr)
        return names

    # --- datasets --------------------------------------------------------
    def dataset(self, name: str) -> np.ndarray:
        name = name.strip("/")
        children = self._group_children(self.root)
        if name not in children:
            raise HDF5Error(f"dataset {name!r} not found (have {sorted(children)})")
        shape = dtype = layout = None
        filters = []
        for mtype, body in self._messages(children[name]):
            if mtype == 0x1:
                ver, rank, flags = struct.unpack_from("<BBB", body, 0)
                off = 8 if ver == 1 else 4
                shape = struct.unpack_from("<%dQ" % rank, body, off)
            elif mtype == 0x3:
                cls = body[0] & 0x0F
                bits0 = body[1]
                size = struct.unpack_from("<I", body, 4)[0]
                if bits0 & 1:
                    raise HDF5Error("big-endian data unsupported")
                if cls == 0:
                    signed = bool(bits0 & 0x08)
                    dtype = np.dtype(("<i" if signed else "<u") + str(size))
                elif cls == 1:
                    dtype = np.dtype("<f" + str(size))
                else:
                    raise HDF5Error(f"datatype class {cls} unsupported")
            elif mtype == 0x8:
                layout = body
            elif mtype == 0xB:
                ver, nf = body[0], body[1]
                pos = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid, nlen, _fl, ncd = struct.unpack_from("<HHHH", body, pos)
                    pos += 8
                    if ver == 1 or fid >= 256:
                        pos += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                    cd = struct.unpack_from("<%dI" % ncd, body, pos)
                    pos += 4 * ncd
                    if ver == 1 and ncd % 2:
                        pos += 4
                    filters.append((fid, cd))
        if shape is None or dtype is None or layout is None:
            raise HDF5Error(f"dataset {name!r}: incomplete header")
        if layout[0] != 3:
            raise HDF5Error(f"layout message version {layout[0]} unsupported")
        lclass = layout[1]
        n = int(np.prod(shape)) if shape else 1
        if lclass == 1:
            addr, size = struct.unpack_from("<QQ", layout, 2)
            return np.frombuffer(self.buf, dtype, n, addr).reshape(shape).copy()
        if lclass != 2:
            raise HDF5Error(f"layout class {lclass} unsupported")
        rank1 = layout[2]
        btree = struct.unpack_from("<Q", layout, 3)[0]
        cdims = struct.unpack_from("<%dI" % rank1, layout, 11)[:-1]
        out = np.zeros(shape, dtype)
        self._read_chunks(btree, rank1, cdims, filters, out)
        return out

    def _read_chunks(self, addr, rank1, cdims, filters, out):
        b = self.buf
        if addr == _UNDEF:
            return
        if b[addr:addr + 4] != b"TREE":
            raise HDF5Error("bad chunk b-tree node")
        ntype, level, nused = struct.unpack_from("<BBH", b, addr + 4)
        if ntype != 1:
            raise HDF5Error("expected raw-data chunk b-tree")
        keysize = 8 + 8 * rank1
        pos = addr + 24
        for _ in range(nused):
            csize, _mask = struct.unpack_from("<II", b, pos)
            offs = struct.unpack_from("<%dQ" % rank1, b, pos + 8)[:-1]
            child = struct.unpack_from("<Q", b, pos + keysize)[0]
            pos += keysize + 8
            if level > 0:
                self._read_chunks(child, rank1, cdims, filters, out)
                continue
            raw = b[child:child + csize]
            for fid, cd in reversed(filters):
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cd[0]
                    a = np.frombuffer(raw, np.uint8).reshape(es, -1)
                    raw = a.T.copy().tobytes()
                else:
                    raise HDF5Error(f"filter {fid} unsupported")
            chunk = np.frombuffer(raw, out.dtype, int(np.prod(cdims))).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, out.shape))
            csl = tuple(slice(0, s.stop - s.start) for s in sl)
            out[sl] = chunk[csl]


def read_h5_dataset(path: str, name: str) -> np.ndarray:
    return _H5File(path).dataset(name)


def read_xdmf_mesh(xdmf_path: str):
    """Return (coords (nv,2) f8, cells (nt,3) i4) of a 2-D triangle XDMF3 mesh.

    Mirrors what DOLFIN's `XDMFFile.read(mesh)` delivers to
    `flow_solver.py:58-62`: vertex coordinates in file order and triangle
    connectivity (cell vertex lists are *not* yet sorted here).
    """
    with open(xdmf_path, "r") as fh:
        text = fh.read()
    geo = re.search(r"<Geometry[^>]*>\s*<DataItem[^>]*>([^<]+)</DataItem>", text)
    topo = re.search(r"<Topology[^>]*>\s*<DataItem[^>]*>([^<]+)</DataItem>", text)
    if not geo or not topo:
        raise HDF5Error(f"{xdmf_path}: could not find Geometry/Topology DataItems")
    base = os.path.dirname(os.path.abspath(xdmf_path))

    def load(ref):
        fname, dset = ref.strip().split(":")
        return read_h5_dataset(os.path.join(base, fname), dset)

    coords = np.ascontiguousarray(load(geo.group(1)), dtype=np.float64)[:, :2]
    cells = np.ascontiguousarray(load(topo.group(1)), dtype=np.int32)
    return np.ascontiguousarray(coords), cells


def load_mesh(path: str):
    """Load a mesh from `.xdmf` (+`.h5`) or from the `.npz` fixture format."""
    if path.endswith(".npz"):
        z = np.load(path)
        return np.ascontiguousarray(z["coords"], np.float64), np.ascontiguousarray(z["cells"], np.int32)
    return read_xdmf_mesh(path)
