"""Tiny TIFF reader/writer standing in for the dxchange calls of the reconstruction loops
(dxchange.write_tiff / read_tiff / read_tiff_stack, cnn_propagator/fullfield.py:223-236,355-357,387-390):
uncompressed strips, one sample per pixel, 2-D images or 3-D stacks as multi-page files."""
import os
import struct

import numpy as np

_FMT = {(3, 32): np.float32, (3, 64): np.float64, (1, 8): np.uint8, (1, 16): np.uint16, (1, 32): np.uint32,
        (2, 8): np.int8, (2, 16): np.int16, (2, 32): np.int32}


def _with_ext(fname):
    root, ext = os.path.splitext(fname)
    return fname if ext.lower() in ('.tif', '.tiff') else fname + '.tiff'


def write_tiff(data, fname='tmp/data', dtype=None, overwrite=False, force_bigtiff=False):
    """dxchange.write_tiff look-alike: a 3-D array becomes a multi-page file."""
    arr = np.asarray(data)
    arr = arr.astype(dtype) if dtype is not None else arr
    if arr.dtype not in (np.float32, np.float64, np.uint8, np.uint16, np.int16, np.int32, np.uint32):
        arr = arr.astype(np.float32)
    arr = arr.astype(arr.dtype.newbyteorder('<'))
    fname = _with_ext(fname)
    folder = os.path.dirname(fname)
    if folder and not os.path.exists(folder):
        os.makedirs(folder)
    if not overwrite and os.path.exists(fname):
        root, ext = os.path.splitext(fname)
        i = 1
        while os.path.exists('{}-{}{}'.format(root, i, ext)):
            i += 1
        fname = '{}-{}{}'.format(root, i, ext)
    pages = arr[None] if arr.ndim == 2 else arr.reshape((-1,) + arr.shape[-2:])
    fmt = {'f': 3, 'u': 1, 'i': 2}[arr.dtype.kind]
    bits = arr.dtype.itemsize * 8
    h, w = pages.shape[1:]
    nbytes = h * w * arr.dtype.itemsize
    ntags = 10
    # classic TIFF holds 32-bit offsets: a 1024^3 float32 volume (4.3 GB) needs BigTIFF (magic 43, 64-bit offsets), which is
    # what tifffile / dxchange switch to as well
    big = force_bigtiff or len(pages) * (nbytes + 256) + 16 >= (1 << 32) - (1 << 20)
    ifd_size = (8 + ntags * 20 + 8) if big else (2 + ntags * 12 + 4)
    with open(fname, 'wb') as f:
        if big:
            f.write(struct.pack('<2sHHHQ', b'II', 43, 8, 0, 16))
            pos = 16
        else:
            f.write(struct.pack('<2sHI', b'II', 42, 8))
            pos = 8
        for p, page in enumerate(pages):
            data_off = pos + ifd_size
            end = data_off + nbytes
            end += end % 2
            nxt = end if p + 1 < len(pages) else 0
            tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1), (273, 16 if big else 4, 1, data_off),
                    (277, 3, 1, 1), (278, 4, 1, h), (279, 16 if big else 4, 1, nbytes), (339, 3, 1, fmt)]
            if big:
                f.write(struct.pack('<Q', ntags))
                for tag, typ, cnt, val in tags:
                    f.write(struct.pack('<HHQQ', tag, typ, cnt, val))       # little endian: a short / long value sits in the low bytes
                f.write(struct.pack('<Q', nxt))
            else:
                f.write(struct.pack('<H', ntags))
                for tag, typ, cnt, val in tags:
                    f.write(struct.pack('<HHI', tag, typ, cnt) + (struct.pack('<HH', val, 0) if typ == 3 else struct.pack('<I', val)))
                f.write(struct.pack('<I', nxt))
            f.write(page.tobytes())
            if (data_off + nbytes) % 2:
                f.write(b'\0')
            pos = end
    return fname


def read_tiff(fname):
    fname = fname if os.path.exists(fname) else _with_ext(fname)
    with open(fname, 'rb') as f:
        buf = f.read()
    bo = {b'II': '<', b'MM': '>'}.get(buf[:2])
    magic = struct.unpack_from(bo + 'H', buf, 2)[0] if bo else 0
    if magic not in (42, 43):
        raise IOError('{}: not a TIFF file'.format(fname))
    big = magic == 43
    if big:
        off = struct.unpack_from(bo + 'Q', buf, 8)[0]
        cnt_fmt, ent_fmt, ent_size, val_size, off_fmt = 'Q', 'HHQ', 20, 8, 'Q'
    else:
        off = struct.unpack_from(bo + 'I', buf, 4)[0]
        cnt_fmt, ent_fmt, ent_size, val_size, off_fmt = 'H', 'HHI', 12, 4, 'I'
    head = struct.calcsize(bo + cnt_fmt)
    pages = []
    while off:
        n = struct.unpack_from(bo + cnt_fmt, buf, off)[0]
        t = {}
        for e in range(n):
            epos = off + head + e * ent_size
            tag, typ, cnt = struct.unpack_from(bo + ent_fmt, buf, epos)
            size = {1: 1, 2: 1, 3: 2, 4: 4, 16: 8}.get(typ, 4) * cnt
            vpos = epos + ent_size - val_size
            if size > val_size:
                vpos = struct.unpack_from(bo + off_fmt, buf, vpos)[0]
            code = {1: 'B', 3: 'H', 4: 'I', 16: 'Q'}.get(typ)
            t[tag] = struct.unpack_from(bo + str(cnt) + code, buf, vpos) if code else None
        if t.get(259, (1,))[0] != 1:
            raise IOError('{}: compressed TIFF is not supported'.format(fname))
        w, h = t[256][0], t[257][0]
        dt = np.dtype(_FMT[(t.get(339, (1,))[0], t[258][0])]).newbyteorder(bo)
        raw = b''.join(buf[o:o + c] for o, c in zip(t[273], t[279]))
        pages.append(np.frombuffer(raw, dtype=dt, count=w * h).reshape(h, w).astype(dt.newbyteorder('=')))
        off = struct.unpack_from(bo + off_fmt, buf, off + head + n * ent_size)[0]
    return pages[0] if len(pages) == 1 else np.stack(pages)


def read_tiff_stack(fname, ind, digit=5):
    """dxchange.read_tiff_stack look-alike: fname names one member (e.g. mask_00000.tiff); `ind` lists the indices."""
    root, ext = os.path.splitext(fname)
    prefix = root[:-digit]
    return np.stack([read_tiff('{}{:0{}d}{}'.format(prefix, i, digit, ext)) for i in ind])


def write_tiff_stack(data, fname='tmp/data', dtype=None, overwrite=False, digit=5):
    """dxchange.write_tiff_stack look-alike: one file per slice along axis 0, `<fname>_00000.tiff` ..."""
    out = []
    for i, page in enumerate(np.asarray(data)):
        out.append(write_tiff(page, '{}_{:0{}d}'.format(os.path.splitext(fname)[0], i, digit), dtype=dtype, overwrite=overwrite))
    return out
