"""Minimal HDF5 access for the reference's data contract: one contiguous dataset `exchange/data`, complex64 stored
as the h5py compound {r: f32, i: f32} (or a plain float dataset), shape (n_theta, Y, X) for full-field
(cnn_propagator/fullfield.py:139-140, writer cnn_propagator/simulation.py:124-126) or (n_theta, n_pos, py, px) for
ptychography (cnn_propagator/ptychography.py:91-92, simulation.py:363).

h5py is used when it is importable; otherwise the built-in parser/writer below handles exactly the subset h5py's
default settings produce for such files: superblock v0, v1 object headers, symbol-table groups, contiguous layout.
`File(path)['exchange/data']` mimics the two access patterns of the reference: `[...]` and lazy fancy indexing.
"""
import struct

import numpy as np

try:                                      # pragma: no cover - depends on the interpreter
    import h5py as _h5py
except Exception:                         # noqa: BLE001
    _h5py = None

_SIG = b'\x89HDF\r\n\x1a\n'
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5FormatError(IOError):
    pass


# --------------------------------------------------------------------------------------------------
# reader
# --------------------------------------------------------------------------------------------------
class _Reader(object):
    def __init__(self, path):
        self.path = path
        with open(path, 'rb') as f:
            self.buf = f.read(1 << 20)            # metadata of such files sits in the first MiB
        if self.buf[:8] != _SIG:
            raise H5FormatError('{}: not an HDF5 file'.format(path))
        ver = self.buf[8]
        if ver not in (0, 1):
            raise H5FormatError('{}: superblock version {} not supported (write with libver="earliest")'.format(path, ver))
        if self.buf[13] != 8 or self.buf[14] != 8:
            raise H5FormatError('only 8-byte offsets/lengths are supported')
        off = 24 if ver == 0 else 28
        self.base = struct.unpack_from('<Q', self.buf, off)[0]
        entry = off + 32                         # root group symbol table entry
        self.root_header = struct.unpack_from('<Q', self.buf, entry + 8)[0]

    def _messages(self, addr):
        """Yield (type, data bytes) of a version-1 object header, following continuation blocks."""
        b = self.buf
        if b[addr] != 1:
            raise H5FormatError('object header version {} not supported'.format(b[addr]))
        nmsg = struct.unpack_from('<H', b, addr + 2)[0]
        size = struct.unpack_from('<I', b, addr + 8)[0]
        blocks = [(addr + 16, size)]
        seen = 0
        while blocks and seen < nmsg:
            pos, length = blocks.pop(0)
            end = pos + length
            while pos + 8 <= end and seen < nmsg:
                mtype, msize = struct.unpack_from('<HH', b, pos)
                data = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                seen += 1
                if mtype == 0x0010:
                    coff, clen = struct.unpack_from('<QQ', data, 0)
                    blocks.append((self.base + coff, clen))
                else:
                    yield mtype, data

    def _group_entries(self, header_addr):
        btree = heap = None
        for mtype, data in self._messages(header_addr):
            if mtype == 0x0011:
                btree, heap = struct.unpack_from('<QQ', data, 0)
        if btree is None:
            raise H5FormatError('object at {:#x} is not an old-style group'.format(header_addr))
        b = self.buf
        heap += self.base
        if b[heap:heap + 4] != b'HEAP':
            raise H5FormatError('bad local heap')
        heap_data = self.base + struct.unpack_from('<Q', b, heap + 24)[0]
        out = {}

        def walk(node):
            node += self.base
            if b[node:node + 4] == b'TREE':
                level = b[node + 5]
                n = struct.unpack_from('<H', b, node + 6)[0]
                for e in range(n):
                    child = struct.unpack_from('<Q', b, node + 24 + 8 + e * 16)[0]
                    walk(child) if level > 0 else leaf(child)
            else:
                raise H5FormatError('bad B-tree node')

        def leaf(addr):
            addr += self.base
            if b[addr:addr + 4] != b'SNOD':
                raise H5FormatError('bad symbol table node')
            n = struct.unpack_from('<H', b, addr + 6)[0]
            for e in range(n):
                name_off, ohdr = struct.unpack_from('<QQ', b, addr + 8 + e * 40)
                s = heap_data + name_off
                name = b[s:b.index(b'\0', s)].decode()
                out[name] = self.base + ohdr

        walk(btree)
        return out

    def find(self, name):
        addr = self.root_header + self.base
        for part in name.strip('/').split('/'):
            entries = self._group_entries(addr)
            if part not in entries:
                raise KeyError('{}: no object {!r}'.format(self.path, name))
            addr = entries[part]
        return addr

    @staticmethod
    def _parse_dtype(data):
        cls = data[0] & 0x0F
        ver = data[0] >> 4
        bits0 = data[1]
        size = struct.unpack_from('<I', data, 4)[0]
        if cls == 1:                              # floating point
            if bits0 & 1:
                raise H5FormatError('big-endian floats not supported')
            return np.dtype('<f{}'.format(size)), 8 + 12
        if cls == 0:
            return np.dtype('<{}{}'.format('i' if data[1] & 0x08 else 'u', size)), 8 + 4
        if cls == 6:                              # compound: expect {r, i}
            nmemb = struct.unpack_from('<H', data, 1)[0]
            pos = 8
            fields = []
            for _ in range(nmemb):
                end = data.index(b'\0', pos)
                fname = data[pos:end].decode()
                if ver < 3:
                    pos += ((end - pos) // 8 + 1) * 8
                    moff = struct.unpack_from('<I', data, pos)[0]
                    pos += 4 + (28 if ver == 1 else 0)
                else:
                    pos = end + 1
                    nb = 1 if size < 256 else (2 if size < 65536 else 4)
                    moff = int.from_bytes(data[pos:pos + nb], 'little')
                    pos += nb
                mdt, used = _Reader._parse_dtype(data[pos:])
                pos += used
                fields.append((fname, mdt, moff))
            dt = np.dtype({'names': [f[0] for f in fields], 'formats': [f[1] for f in fields],
                           'offsets': [f[2] for f in fields], 'itemsize': size})
            return dt, pos
        raise H5FormatError('datatype class {} not supported'.format(cls))

    def dataset(self, name):
        addr = self.find(name)
        shape = dtype = data_addr = None
        for mtype, data in self._messages(addr):
            if mtype == 0x0001:
                ver, rank = data[0], data[1]
                start = 8 if ver == 1 else 4
                shape = struct.unpack_from('<{}Q'.format(rank), data, start)
            elif mtype == 0x0003:
                dtype, _ = self._parse_dtype(data)
            elif mtype == 0x0008:
                if data[0] != 3:
                    raise H5FormatError('data layout message version {} not supported'.format(data[0]))
                if data[1] != 1:
                    raise H5FormatError('only contiguous datasets are supported (no chunking/compression)')
                data_addr = struct.unpack_from('<Q', data, 2)[0]
        if shape is None or dtype is None or data_addr is None:
            raise H5FormatError('{}: {} is not a dataset'.format(self.path, name))
        if data_addr == _UNDEF:
            return np.zeros(shape, dtype=dtype)
        return np.memmap(self.path, dtype=dtype, mode='r', offset=self.base + data_addr, shape=tuple(shape))


def _as_complex(arr):
    if arr.dtype.names and set(arr.dtype.names) == {'r', 'i'}:
        return arr['r'] + 1j * arr['i'] if arr.dtype['r'] != np.float32 else (arr['r'] + 1j * arr['i']).astype(np.complex64)
    return arr


class Dataset(object):
    """Array-like over a contiguous dataset; `[...]` loads it, other indices read only what is asked for."""

    def __init__(self, raw):
        self._raw = raw
        self.shape = tuple(raw.shape)
        self.dtype = np.dtype(np.complex64) if raw.dtype.names else raw.dtype

    def __getitem__(self, key):
        return _as_complex(np.asarray(self._raw[key]))

    def __len__(self):
        return self.shape[0]


class File(object):
    """`h5py.File(path, 'r')` look-alike for reading; uses h5py itself when present."""

    def __init__(self, path, mode='r'):
        if mode != 'r':
            raise ValueError('File() reads only; use write_dataset() to create files')
        self.path = path
        self._h5 = _h5py.File(path, 'r') if _h5py is not None else None
        self._reader = None if self._h5 is not None else _Reader(path)

    def __getitem__(self, name):
        if self._h5 is not None:
            return self._h5[name]
        return Dataset(self._reader.dataset(name))

    def close(self):
        if self._h5 is not None:
            self._h5.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def read_dataset(path, name='exchange/data'):
    with File(path) as f:
        return np.asarray(f[name][...])


# --------------------------------------------------------------------------------------------------
# writer (superblock v0, one dataset at <group>/<name>, contiguous)
# --------------------------------------------------------------------------------------------------
def _dtype_message(dt):
    def flt(size):
        if size == 4:
            prop = struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)
            bits = (0x20, 0x1f, 0x00)
        else:
            prop = struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)
            bits = (0x20, 0x3f, 0x00)
        return struct.pack('<BBBBI', 0x11, *bits, size) + prop
    if dt.kind == 'f':
        return flt(dt.itemsize)
    if dt.kind == 'c':
        half = dt.itemsize // 2
        body = b''
        for fname, off in (('r', 0), ('i', half)):
            body += fname.encode() + b'\0' * 7 + struct.pack('<IB3xI4x4I', off, 0, 0, 0, 0, 0, 0) + flt(half)
        return struct.pack('<BBBBI', 0x16, 2, 0, 0, dt.itemsize) + body
    raise TypeError('cannot store dtype {}'.format(dt))


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _object_header(messages):
    body = b''
    for mtype, data in messages:
        data = _pad8(data)
        body += struct.pack('<HHB3x', mtype, len(data), 0) + data
    return struct.pack('<BxHII4x', 1, len(messages), 1, len(body)) + body


def write_dataset(path, name, array):
    """Write `array` (float32/float64/complex64/complex128) as the only dataset `name` ('group/data')."""
    array = np.ascontiguousarray(array)
    parts = name.strip('/').split('/')
    if len(parts) != 2:
        raise ValueError("name must look like 'exchange/data'")
    gname, dname = parts
    LEAF_K, INT_K = 4, 16
    btree_size = 24 + (2 * INT_K + 1) * 8 + 2 * INT_K * 8
    snod_size = 8 + 2 * LEAF_K * 40
    heap_data_size = 88

    def group_blocks(ohdr_addr, child_name, child_ohdr):
        """object header + B-tree + heap + SNOD of a group with one member; returns (bytes, end address)."""
        ohdr = _object_header([(0x0011, b'\0' * 16)])          # patched below
        bt = ohdr_addr + len(ohdr)
        hp = bt + btree_size
        hd = hp + 32
        sn = hd + heap_data_size
        ohdr = _object_header([(0x0011, struct.pack('<QQ', bt, hp))])
        name_off = 8
        heap_data = (b'\0' * 8 + child_name.encode() + b'\0').ljust(heap_data_size, b'\0')
        free_off = 8 + ((len(child_name) + 1 + 7) // 8) * 8
        heap_data = heap_data[:free_off] + struct.pack('<QQ', 1, heap_data_size - free_off) + heap_data[free_off + 16:]
        tree = b'TREE' + struct.pack('<BBHQQ', 0, 0, 1, _UNDEF, _UNDEF) + struct.pack('<QQQ', 0, sn, name_off)
        tree = tree.ljust(btree_size, b'\0')
        heap = b'HEAP' + struct.pack('<B3xQQQ', 0, heap_data_size, free_off, hd)
        snod = (b'SNOD' + struct.pack('<BxH', 1, 1) + struct.pack('<QQII16x', name_off, child_ohdr, 0, 0)).ljust(snod_size, b'\0')
        return ohdr, bt, hp, ohdr + tree + heap + heap_data + snod, sn + snod_size

    root_addr = 96
    # two-pass: sizes do not depend on addresses, so lay out with dummy child addresses first
    _, _, _, blob, g_addr = group_blocks(root_addr, gname, 0)
    _, _, _, gblob, d_addr = group_blocks(g_addr, dname, 0)
    dspace = struct.pack('<BBB5x', 1, array.ndim, 0) + struct.pack('<{}Q'.format(array.ndim), *array.shape)
    fill = struct.pack('<BBBB', 2, 2, 2, 0)
    msgs = [(0x0001, dspace), (0x0003, _dtype_message(array.dtype)), (0x0005, fill), (0x0008, b'\0' * 18)]
    data_addr = d_addr + len(_object_header(msgs))
    data_addr += -data_addr % 8
    msgs[3] = (0x0008, struct.pack('<BBQQ', 3, 1, data_addr, array.nbytes))
    dhdr = _object_header(msgs).ljust(data_addr - d_addr, b'\0')
    rohdr, rbt, rhp, blob, _ = group_blocks(root_addr, gname, g_addr)
    _, _, _, gblob, _ = group_blocks(g_addr, dname, d_addr)
    eof = data_addr + array.nbytes
    sb = _SIG + struct.pack('<BBBxBBBxHHI', 0, 0, 0, 0, 8, 8, LEAF_K, INT_K, 0)
    sb += struct.pack('<QQQQ', 0, _UNDEF, eof, _UNDEF)
    sb += struct.pack('<QQII', 0, root_addr, 1, 0) + struct.pack('<QQ', rbt, rhp)
    assert len(sb) == root_addr
    with open(path, 'wb') as f:
        f.write(sb + blob + gblob + dhdr)
        f.write(array.tobytes())
