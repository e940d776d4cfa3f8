"""Minimal HDF5 reader for Keras-3 ``model.weights.h5`` files (no h5py on the MI355X box).

A `.keras` archive is a stored zip whose ``model.weights.h5`` member is written by
h5py with the oldest, simplest on-disk structures:

* superblock version 0, 8-byte offsets and lengths;
* "old style" groups: object header v1 -> symbol-table message (type 0x0011) ->
  v1 B-tree (``TREE``) -> symbol nodes (``SNOD``) whose link names live in a local
  heap (``HEAP``);
* datasets: dataspace (0x0001), datatype (0x0003), data layout (0x0008, version 3,
  class 1 = contiguous) messages, possibly spread over continuation blocks (0x0010);
* no chunking, compression or filters.

Anything outside that subset raises ``ValueError``.  The reference loads the same file
through ``tf.keras.models.load_model`` (reference: birdnet_stm32/models/runners.py:109-113).
"""

from __future__ import annotations

import struct

import numpy as np

_SIGNATURE = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5File:
    """Read-only view of an HDF5 byte string restricted to the subset described above."""

    def __init__(self, raw: bytes):
        if raw[:8] != _SIGNATURE:
            raise ValueError("not an HDF5 file")
        self.raw = raw
        sb_version = raw[8]
        if sb_version != 0:
            raise ValueError(f"HDF5 superblock version {sb_version} is not supported (expected 0)")
        size_offsets, size_lengths = raw[13], raw[14]
        if size_offsets != 8 or size_lengths != 8:
            raise ValueError("only 8-byte offsets/lengths are supported")
        # superblock v0: 8 sig + 8 version bytes + 2*u16 group k + u32 flags = 24, then four addresses
        base_addr = self._u64(24)
        if base_addr != 0:
            raise ValueError("non-zero HDF5 base address is not supported")
        # root group symbol-table entry starts after base, free-space, eof, driver-info addresses
        root_entry = 24 + 4 * 8
        self.root_header = self._u64(root_entry + 8)

    # -- primitive accessors ----------------------------------------------
    def _u8(self, p):
        return self.raw[p]

    def _u16(self, p):
        return struct.unpack_from("<H", self.raw, p)[0]

    def _u32(self, p):
        return struct.unpack_from("<I", self.raw, p)[0]

    def _u64(self, p):
        return struct.unpack_from("<Q", self.raw, p)[0]

    # -- object headers ------------------------------------------------------
    def _messages(self, addr: int) -> list[tuple[int, int, int]]:
        """Return ``(type, body_pos, body_size)`` for every message of a v1 object header."""
        if self._u8(addr) != 1:
            raise ValueError(f"object header at {addr}: version {self._u8(addr)} unsupported (expected 1)")
        n_msgs = self._u16(addr + 2)
        first_size = self._u32(addr + 8)
        blocks = [(addr + 16, first_size)]  # 12-byte prefix padded to 8 -> 16
        out: list[tuple[int, int, int]] = []
        bi = 0
        while bi < len(blocks) and len(out) < n_msgs:
            pos, size = blocks[bi]
            end = pos + size
            while pos + 8 <= end and len(out) < n_msgs:
                mtype = self._u16(pos)
                msize = self._u16(pos + 2)
                body = pos + 8
                out.append((mtype, body, msize))
                if mtype == 0x0010:  # continuation: address + length of the next block
                    blocks.append((self._u64(body), self._u64(body + 8)))
                pos = body + msize
            bi += 1
        return out

    # -- groups --------------------------------------------------------------
    def _heap_data(self, heap_addr: int) -> int:
        if self.raw[heap_addr : heap_addr + 4] != b"HEAP":
            raise ValueError("bad local heap signature")
        return self._u64(heap_addr + 24)

    def _walk_btree(self, node: int, heap_data: int, out: dict[str, int]) -> None:
        if self.raw[node : node + 4] != b"TREE":
            raise ValueError("bad B-tree signature")
        node_type, level, used = self._u8(node + 4), self._u8(node + 5), self._u16(node + 6)
        if node_type != 0:
            raise ValueError("only group B-trees (type 0) are supported")
        # header: sig 4 + type 1 + level 1 + entries 2 + left 8 + right 8 = 24; then key0, child0, key1, ...
        p = node + 24
        for k in range(used):
            child = self._u64(p + 8 + k * 16)
            if level > 0:
                self._walk_btree(child, heap_data, out)
            else:
                self._read_snod(child, heap_data, out)

    def _read_snod(self, addr: int, heap_data: int, out: dict[str, int]) -> None:
        if self.raw[addr : addr + 4] != b"SNOD":
            raise ValueError("bad symbol node signature")
        n = self._u16(addr + 6)
        for k in range(n):
            e = addr + 8 + 40 * k
            name_off = self._u64(e)
            header = self._u64(e + 8)
            s = heap_data + name_off
            name = self.raw[s : self.raw.index(b"\x00", s)].decode("utf-8")
            out[name] = header

    def children(self, header_addr: int) -> dict[str, int] | None:
        """Links of a group as ``{name: object_header_address}``; ``None`` if not a group."""
        for mtype, body, _ in self._messages(header_addr):
            if mtype == 0x0011:
                btree, heap = self._u64(body), self._u64(body + 8)
                out: dict[str, int] = {}
                self._walk_btree(btree, self._heap_data(heap), out)
                return out
        return None

    # -- datasets ------------------------------------------------------------
    def dataset(self, header_addr: int) -> np.ndarray | None:
        """Decode a contiguous little-endian dataset; ``None`` if the object is not a dataset."""
        shape = None
        dtype = None
        data_addr = data_size = None
        for mtype, body, _size in self._messages(header_addr):
            if mtype == 0x0001:  # dataspace
                ver, rank = self._u8(body), self._u8(body + 1)
                if ver == 1:
                    dims_at = body + 8
                elif ver == 2:
                    dims_at = body + 4
                else:
                    raise ValueError(f"dataspace version {ver} unsupported")
                shape = tuple(self._u64(dims_at + 8 * i) for i in range(rank))
            elif mtype == 0x0003:  # datatype
                cls = self._u8(body) & 0x0F
                bits0 = self._u8(body + 1)
                nbytes = self._u32(body + 4)
                if bits0 & 1:
                    raise ValueError("big-endian datasets are not supported")
                if cls == 1:
                    dtype = {2: np.float16, 4: np.float32, 8: np.float64}.get(nbytes)
                elif cls == 0:
                    signed = bool(bits0 & 0x08)
                    dtype = np.dtype(f"{'i' if signed else 'u'}{nbytes}").type
                if dtype is None:
                    raise ValueError(f"datatype class {cls} size {nbytes} unsupported")
            elif mtype == 0x0008:  # data layout
                ver = self._u8(body)
                if ver != 3:
                    raise ValueError(f"data layout version {ver} unsupported")
                lclass = self._u8(body + 1)
                if lclass == 1:
                    data_addr, data_size = self._u64(body + 2), self._u64(body + 10)
                elif lclass == 0:  # compact: size u16 then raw bytes
                    data_size = self._u16(body + 2)
                    data_addr = body + 4
                else:
                    raise ValueError("chunked datasets are not supported")
        if shape is None or dtype is None or data_addr is None:
            return None
        count = int(np.prod(shape)) if shape else 1
        if data_addr == _UNDEF or count == 0:
            return np.zeros(shape, dtype=dtype)
        itemsize = np.dtype(dtype).itemsize
        if data_size < count * itemsize:
            raise ValueError("dataset storage smaller than its dataspace")
        arr = np.frombuffer(self.raw, dtype=np.dtype(dtype).newbyteorder("<"), count=count, offset=data_addr)
        return arr.astype(dtype).reshape(shape)

    # -- whole-file walk -----------------------------------------------------
    def walk(self, skip_prefixes: tuple[str, ...] = ()) -> dict[str, np.ndarray]:
        """Return every dataset as ``{"/path/to/dataset": array}``."""
        found: dict[str, np.ndarray] = {}

        def rec(addr: int, path: str) -> None:
            kids = self.children(addr)
            if kids is None:
                arr = self.dataset(addr)
                if arr is not None:
                    found[path] = arr
                return
            for name, child in kids.items():
                sub = f"{path}/{name}"
                if any(sub.startswith(pfx) for pfx in skip_prefixes):
                    continue
                rec(child, sub)

        rec(self.root_header, "")
        return found


def read_h5_datasets(raw: bytes, skip_prefixes: tuple[str, ...] = ("/optimizer",)) -> dict[str, np.ndarray]:
    """Decode all datasets of an HDF5 byte string (optimizer state skipped by default)."""
    return H5File(raw).walk(skip_prefixes)
