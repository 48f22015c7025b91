"""Read-only access to the label file of scripts/prepro_labels.py without h5py.

``prepro_labels.py:158-163`` writes ``<name>_label.h5`` with ``h5py.File(path, 'w')`` + four ``create_dataset(name, dtype='uint32',
data=...)`` calls: the oldest on-disk format HDF5 still writes by default -- version-0 superblock, version-1 object headers,
a version-1 group B-tree over one local heap, and CONTIGUOUS, unchunked, uncompressed datasets.  That subset is ~100 lines of
struct unpacking (HDF5 File Format Specification 2.0/3.0, sections II.A superblock, III.A B-trees, III.B/C symbol table nodes,
III.D local heaps, IV.A object headers + the dataspace / datatype / layout / continuation messages); anything else (chunked or
compressed datasets, newer superblocks, dense groups) raises ``H5LiteError`` naming the feature, and
``feature_loader.load_labels`` then asks for h5py or for the one-time ``tools/convert_labels.py``.
"""
import struct

import numpy as np

SIG = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5LiteError(ValueError):
    pass


class H5File:
    def __init__(self, path):
        with open(path, 'rb') as f:
            self.b = f.read()
        self.base = -1
        off = 0
        while off < len(self.b):                  # the superblock sits at 0, 512, 1024, ... (spec II.A)
            if self.b[off:off + 8] == SIG:
                self.base = off
                break
            off = 512 if off == 0 else off * 2
        if self.base < 0:
            raise H5LiteError('%s is not an HDF5 file' % path)
        ver = self.b[self.base + 8]
        if ver not in (0, 1):
            raise H5LiteError('superblock version %d (a file written with libver="latest"?): only the default version-0/1 layout '
                              'of h5py.File(path, "w") is read here' % ver)
        so, sl = self.b[self.base + 13], self.b[self.base + 14]
        if (so, sl) != (8, 8):
            raise H5LiteError('offsets / lengths of %d / %d bytes' % (so, sl))
        p = self.base + 24 + (4 if ver == 1 else 0)           # v1 adds the indexed-storage K + 2 reserved bytes
        base_addr, _free, _eof, _drv = struct.unpack_from('<4Q', self.b, p)
        self.base_addr = base_addr
        # root group symbol table entry: link name offset, object header address, cache type, reserved, 16 bytes of scratch
        _name, root_hdr, cache, _r = struct.unpack_from('<QQII', self.b, p + 32)
        if cache == 1:
            self.root_btree, self.root_heap = struct.unpack_from('<QQ', self.b, p + 32 + 24)
        else:
            self.root_btree, self.root_heap = self._group_of(root_hdr)
        self._entries = dict(self._walk(self.root_btree, self._heap_data(self.root_heap)))

    # ---- low level
    def _at(self, addr):
        return self.base_addr + addr

    def _heap_data(self, addr):
        p = self._at(addr)
        if self.b[p:p + 4] != b'HEAP':
            raise H5LiteError('local heap signature missing')
        _size, _free, data = struct.unpack_from('<QQQ', self.b, p + 8)
        return self._at(data)

    def _name(self, heap_data, off):
        end = self.b.index(b'\0', heap_data + off)
        return self.b[heap_data + off:end].decode('utf-8')

    def _walk(self, btree, heap_data):
        p = self._at(btree)
        if self.b[p:p + 4] != b'TREE':
            raise H5LiteError('group B-tree signature missing (a "dense" new-style group?)')
        ntype, level, used = struct.unpack_from('<BBH', self.b, p + 4)
        if ntype != 0:
            raise H5LiteError('B-tree node type %d where a group node was expected' % ntype)
        q = p + 8 + 16                              # left / right sibling addresses
        for i in range(used):                       # key_i (heap offset), child_i, ..., key_used
            child = struct.unpack_from('<Q', self.b, q + 8 + 16 * i)[0]
            if level > 0:
                yield from self._walk(child, heap_data)
            else:
                s = self._at(child)
                if self.b[s:s + 4] != b'SNOD':
                    raise H5LiteError('symbol table node signature missing')
                n = struct.unpack_from('<H', self.b, s + 6)[0]
                for j in range(n):
                    name_off, hdr = struct.unpack_from('<QQ', self.b, s + 8 + 40 * j)
                    yield self._name(heap_data, name_off), hdr

    def _messages(self, hdr):
        """(type, payload bytes) of a version-1 object header, continuation blocks followed"""
        p = self._at(hdr)
        if self.b[p] != 1:
            raise H5LiteError('object header version %d (libver="latest" writes version 2): not read here' % self.b[p])
        nmsg, _refs, size = struct.unpack_from('<HII', self.b, p + 2)
        blocks, out = [(p + 16, size)], []
        while blocks and len(out) < nmsg:
            q, left = blocks.pop(0)
            while left >= 8 and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from('<HHB', self.b, q)
                body = self.b[q + 8:q + 8 + msize]
                if mtype == 0x0010:                 # continuation: offset, length
                    coff, clen = struct.unpack_from('<QQ', body, 0)
                    blocks.append((self._at(coff), clen))
                out.append((mtype, body))
                q += 8 + msize
                left -= 8 + msize
        return out

    def _group_of(self, hdr):
        for mtype, body in self._messages(hdr):
            if mtype == 0x0011:
                return struct.unpack_from('<QQ', body, 0)
        raise H5LiteError('the root object carries no symbol-table message (a new-style "dense" group): not read here')

    # ---- datasets
    def keys(self):
        return sorted(self._entries)

    def __contains__(self, name):
        return name in self._entries

    def __getitem__(self, name):
        if name not in self._entries:
            raise KeyError(name)
        shape = dtype = None
        addr = size = None
        for mtype, body in self._messages(self._entries[name]):
            if mtype == 0x0001:                     # dataspace
                ver, rank, flags = body[0], body[1], body[2]
                if ver == 1:
                    shape = struct.unpack_from('<%dQ' % rank, body, 8)
                elif ver == 2:
                    shape = struct.unpack_from('<%dQ' % rank, body, 4)
                else:
                    raise H5LiteError('dataspace message version %d' % ver)
            elif mtype == 0x0003:                   # datatype
                cls, bits0, dsize = body[0] & 15, body[1], struct.unpack_from('<I', body, 4)[0]
                if bits0 & 1:
                    raise H5LiteError('big-endian dataset %r' % name)
                if cls == 0:
                    dtype = np.dtype('%s%d' % ('i' if bits0 & 8 else 'u', dsize))
                elif cls == 1:
                    dtype = np.dtype('f%d' % dsize)
                else:
                    raise H5LiteError('dataset %r has datatype class %d (only integers and floats are read here)' % (name, cls))
            elif mtype == 0x0008:                   # data layout
                ver = body[0]
                if ver == 3:
                    lclass = body[1]
                    if lclass == 1:
                        addr, size = struct.unpack_from('<QQ', body, 2)
                    elif lclass == 0:               # compact: the data sit in the message
                        n = struct.unpack_from('<H', body, 2)[0]
                        addr, size = ('compact', body[4:4 + n]), n
                    else:
                        raise H5LiteError('dataset %r is chunked (chunks= / compression= / maxshape= in create_dataset): '
                                          'read it with h5py or convert it once with tools/convert_labels.py' % name)
                else:
                    raise H5LiteError('data layout message version %d of dataset %r' % (ver, name))
            elif mtype == 0x000B:
                raise H5LiteError('dataset %r has a filter pipeline (compression): read it with h5py' % name)
        if shape is None or dtype is None or addr is None:
            raise H5LiteError('%r is not a simple dataset' % name)
        count = int(np.prod(shape)) if len(shape) else 1
        if isinstance(addr, tuple):
            raw = addr[1]
        elif addr == UNDEF or count == 0:
            raw = b''
        else:
            raw = self.b[self._at(addr):self._at(addr) + size]
        arr = np.frombuffer(raw, dtype=dtype, count=min(count, len(raw) // dtype.itemsize)).copy()
        if arr.size != count:
            raise H5LiteError('dataset %r: %d of %d elements on disk' % (name, arr.size, count))
        return arr.reshape(shape)


def read_datasets(path, names=None):
    f = H5File(path)
    return {k: f[k] for k in (names or f.keys())}
