"""Minimal HDF5 reader for the files the reference's data preparation writes.

`code/dataloaders/acdc_data_processing.py:53-60,108-113` stores every slice / volume with h5py defaults
(`create_dataset(name, data=..., compression="gzip")`): superblock version 0, a root group addressed through a symbol table
(B-tree v1 + local heap), version-1 object headers, and datasets that are contiguous or chunked with the deflate (and
optionally shuffle) filter.  h5py is not in the image, so this module reads exactly that subset with `struct` + `zlib`:

    with File(path) as f:
        image = f["image"][:]          # numpy array, like h5py
        names = f.keys()

Anything outside the subset (new-style groups, v2 object headers, other filters, compound types) raises `H5Error`
instead of guessing.  Format reference: "HDF5 File Format Specification Version 2.0/3.0" (superblock 0, symbol-table
groups, v1 B-trees, object header v1 messages 0x0001 / 0x0003 / 0x0008 / 0x000B / 0x0010 / 0x0011)."""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(Exception):
    pass


class _Dataset(object):
    def __init__(self, f, shape, dtype, layout, filters):
        self._f, self.shape, self.dtype, self._layout, self._filters = f, tuple(shape), np.dtype(dtype), layout, filters

    def __getitem__(self, key):
        return self._read()[key]

    def _read(self):
        f, lay = self._f, self._layout
        n = int(np.prod(self.shape)) if self.shape else 1
        if lay[0] == "contiguous":
            addr, size = lay[1], lay[2]
            if addr == UNDEF:
                return np.zeros(self.shape, self.dtype)
            return np.frombuffer(f._b, self.dtype, n, addr).reshape(self.shape).copy()
        if lay[0] == "compact":
            return np.frombuffer(lay[1], self.dtype, n).reshape(self.shape).copy()
        _, btree, cdims = lay
        out = np.zeros(self.shape, self.dtype)
        rank = len(self.shape)
        for offs, raw, fmask in f._chunks(btree, rank):
            for k, (fid, cd) in enumerate(reversed(self._filters)):
                if fmask & (1 << (len(self._filters) - 1 - k)):
                    continue                                   # this filter was skipped for this chunk
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                                 # shuffle: bytes of each element are stored de-interleaved
                    es = cd[0] if cd else self.dtype.itemsize
                    a = np.frombuffer(raw, np.uint8)
                    m = len(a) // es
                    raw = a[:m * es].reshape(es, m).T.tobytes() + a[m * es:].tobytes()
                elif fid == 3:                                 # fletcher32 checksum: 4 trailing bytes
                    raw = raw[:-4]
                else:
                    raise H5Error("unsupported HDF5 filter id %d" % fid)
            chunk = np.frombuffer(raw, self.dtype, int(np.prod(cdims))).reshape(cdims)
            sel_o = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, self.shape))
            sel_c = tuple(slice(0, s.stop - s.start) for s in sel_o)
            out[sel_o] = chunk[sel_c]
        return out


class File(object):
    def __init__(self, path, mode="r"):
        if mode != "r":
            raise H5Error("h5lite is read-only")
        with open(path, "rb") as fh:
            self._b = fh.read()
        b = self._b
        if b[:8] != SIGNATURE:
            raise H5Error("%s: not an HDF5 file" % path)
        if b[8] != 0:
            raise H5Error("%s: superblock version %d (only version 0 is read)" % (path, b[8]))
        if b[13] != 8 or b[14] != 8:
            raise H5Error("%s: offsets/lengths of %d/%d bytes (only 8/8 is read)" % (path, b[13], b[14]))
        base = self._u64(24)
        if base != 0:
            raise H5Error("non-zero base address")
        # root group symbol-table entry starts after the four addresses: 24 + 4*8 = 56
        ent = 56
        cache_type = self._u32(ent + 16)
        if cache_type == 1:
            btree, heap = self._u64(ent + 24), self._u64(ent + 32)
        else:                                                   # find the symbol-table message in the root object header
            msgs = self._messages(self._u64(ent + 8))
            st = [m for t, m in msgs if t == 0x0011]
            if not st:
                raise H5Error("root group without a symbol table (new-style groups are not read)")
            btree, heap = struct.unpack_from("<QQ", st[0], 0)
        self._links = dict(self._group_links(btree, heap))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        pass

    def keys(self):
        return sorted(self._links)

    def __contains__(self, name):
        return name in self._links

    def __getitem__(self, name):
        if name not in self._links:
            raise KeyError(name)
        return self._dataset(self._links[name])

    # ------------------------------------------------------------------ low level
    def _u16(self, o):
        return struct.unpack_from("<H", self._b, o)[0]

    def _u32(self, o):
        return struct.unpack_from("<I", self._b, o)[0]

    def _u64(self, o):
        return struct.unpack_from("<Q", self._b, o)[0]

    def _group_links(self, btree, heap):
        b = self._b
        if b[heap:heap + 4] != b"HEAP":
            raise H5Error("bad local heap")
        heap_data = self._u64(heap + 24)

        def name_at(off):
            p = heap_data + off
            return b[p:b.index(b"\0", p)].decode()

        def walk(node):
            if b[node:node + 4] != b"TREE" or b[node + 4] != 0:
                raise H5Error("bad group B-tree node")
            level, used = b[node + 5], self._u16(node + 6)
            p = node + 8 + 16                                   # skip the sibling addresses
            for i in range(used):
                child = self._u64(p + 8)                        # key_i (8) child_i (8) ... key_used (8)
                p += 16
                if level:
                    for x in walk(child):
                        yield x
                else:
                    if b[child:child + 4] != b"SNOD":
                        raise H5Error("bad symbol-table node")
                    ns = self._u16(child + 6)
                    for k in range(ns):
                        e = child + 8 + 40 * k
                        yield name_at(self._u64(e)), self._u64(e + 8)
        return walk(btree)

    def _messages(self, addr):
        b = self._b
        if b[addr] != 1:
            raise H5Error("object header version %d (only version 1 is read)" % b[addr])
        nmsg, size = self._u16(addr + 2), self._u32(addr + 8)
        blocks, out = [(addr + 16, size)], []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = self._u16(p), self._u16(p + 2)
                data = b[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:                             # continuation
                    blocks.append(struct.unpack_from("<QQ", data, 0))
                out.append((mtype, data))
        return out

    def _dataset(self, addr):
        shape = dtype = layout = None
        filters = []
        for mtype, d in self._messages(addr):
            if mtype == 0x0001:                                 # dataspace
                ver, rank, flags = d[0], d[1], d[2]
                o = 8 if ver == 1 else 4
                shape = struct.unpack_from("<%dQ" % rank, d, o) if rank else ()
            elif mtype == 0x0003:                               # datatype
                cls, bits0, size = d[0] & 0x0F, d[1], struct.unpack_from("<I", d, 4)[0]
                if bits0 & 1:
                    raise H5Error("big-endian data is not read")
                if cls == 0:
                    dtype = ("i" if bits0 & 0x08 else "u") + str(size)
                elif cls == 1:
                    dtype = "f" + str(size)
                else:
                    raise H5Error("datatype class %d is not read" % cls)
            elif mtype == 0x0008:                               # data layout
                if d[0] != 3:
                    raise H5Error("data layout message version %d (only 3 is read)" % d[0])
                lc = d[1]
                if lc == 1:
                    layout = ("contiguous",) + struct.unpack_from("<QQ", d, 2)
                elif lc == 2:
                    nd = d[2]
                    bt = struct.unpack_from("<Q", d, 3)[0]
                    dims = struct.unpack_from("<%dI" % nd, d, 11)
                    layout = ("chunked", bt, tuple(dims[:-1]))
                elif lc == 0:
                    n = struct.unpack_from("<H", d, 2)[0]
                    layout = ("compact", d[4:4 + n])
                else:
                    raise H5Error("layout class %d" % lc)
            elif mtype == 0x000B:                               # filter pipeline
                ver, nf = d[0], d[1]
                o = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid = struct.unpack_from("<H", d, o)[0]
                    if ver == 1 or fid >= 256:
                        nlen = struct.unpack_from("<H", d, o + 2)[0]
                        o += 4
                    else:
                        nlen = 0
                        o += 2
                    _flags, ncd = struct.unpack_from("<HH", d, o)
                    o += 4
                    o += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                    cd = struct.unpack_from("<%dI" % ncd, d, o)
                    o += 4 * ncd
                    if ver == 1 and ncd % 2:
                        o += 4
                    filters.append((fid, cd))
        if shape is None or dtype is None or layout is None:
            raise H5Error("object at %d is not a simple dataset" % addr)
        return _Dataset(self, shape, dtype, layout, filters)

    def _chunks(self, node, rank):
        b = self._b
        if node == UNDEF:
            return
        if b[node:node + 4] != b"TREE" or b[node + 4] != 1:
            raise H5Error("bad chunk B-tree node")
        level, used = b[node + 5], self._u16(node + 6)
        ksz = 8 + 8 * (rank + 1)
        p = node + 8 + 16
        for _ in range(used):
            csize, fmask = struct.unpack_from("<II", b, p)
            offs = struct.unpack_from("<%dQ" % rank, b, p + 8)
            child = self._u64(p + ksz)
            p += ksz + 8
            if level:
                for x in self._chunks(child, rank):
                    yield x
            else:
                yield offs, b[child:child + csize], fmask
