"""Minimal reader for scanline OpenEXR files with PIZ (or no / ZIP) compression and FLOAT / HALF channels -- enough to read the one image
the reference ships (examples/evaluation/cbox.exr, rendered with Mitsuba 2) without OpenEXR / OpenImageIO.  Written from the published
file-format description (OpenEXR "Technical Introduction" / file layout documents); used only by tools/make_cbox_fixture.py."""
import struct
import zlib

import numpy as np


def _read_header(d):
    assert d[:4] == b"v/1\x01", "not an OpenEXR file"
    version, = struct.unpack("<I", d[4:8])
    assert version & 0xFF == 2 and not (version & 0x200), "only single-part scanline files"
    pos, attrs = 8, {}
    while True:
        e = d.index(b"\0", pos)
        name = d[pos:e].decode()
        pos = e + 1
        if not name:
            break
        e = d.index(b"\0", pos)
        typ = d[pos:e].decode()
        pos = e + 1
        size, = struct.unpack("<I", d[pos:pos + 4])
        pos += 4
        attrs[name] = (typ, d[pos:pos + size])
        pos += size
    chans, v, p = [], attrs["channels"][1], 0
    while v[p] != 0:
        e = v.index(b"\0", p)
        cname = v[p:e].decode()
        p = e + 1
        ptype, = struct.unpack("<i", v[p:p + 4])
        xs, ys = struct.unpack("<ii", v[p + 8:p + 16])
        assert xs == 1 and ys == 1, "subsampled channels are not supported"
        p += 16
        chans.append((cname, ptype))
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    return pos, chans, attrs["compression"][1][0], (x0, y0, x1, y1)


class _Bits:
    def __init__(self, data, pos):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n):
        while self.lc < n:
            self.c = (self.c << 8) | self.d[self.p]
            self.p += 1
            self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)


def _huf_uncompress(buf, n_raw):
    im, iM, _tl, n_bits, _res = struct.unpack("<5I", buf[:20])
    hcode = [0] * 65537
    b = _Bits(buf, 20)
    i = im
    while i <= iM:  # packed code lengths, 6 bits each, with zero-run escapes
        l = b.get(6)
        hcode[i] = l
        if l == 63:
            run = b.get(8) + 6
            for k in range(run):
                hcode[i + k] = 0
            i += run - 1
        elif l >= 59:
            run = l - 59 + 2
            for k in range(run):
                hcode[i + k] = 0
            i += run - 1
        i += 1
    n = [0] * 59  # canonical code assignment
    for i in range(65537):
        n[hcode[i]] += 1
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        n[i] = c
        c = nc
    table = {}
    for i in range(65537):
        l = hcode[i]
        if l > 0:
            table[(l, n[l])] = i
            n[l] += 1
    data, pos = buf, b.p  # the code table ends on a byte boundary of its own bit reader
    bits = _Bits(data, pos)
    out, rlc, total = [], iM, 0
    code, length = 0, 0
    while total < n_bits and len(out) < n_raw:
        code = (code << 1) | bits.get(1)
        length += 1
        total += 1
        sym = table.get((length, code))
        if sym is None:
            assert length < 59, "corrupt Huffman stream"
            continue
        if sym == rlc:
            run = bits.get(8)
            total += 8
            out.extend([out[-1]] * run)
        else:
            out.append(sym)
        code, length = 0, 0
    assert len(out) == n_raw, (len(out), n_raw)
    return np.array(out, dtype=np.uint16)


def _wdec14(l, h):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    return (ai & 0xFFFF).astype(np.uint16), ((ai - hs) & 0xFFFF).astype(np.uint16)


def _wdec16(l, h):
    m = l.astype(np.int32)
    d = h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xFFFF
    aa = (d + bb - (1 << 15)) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a, mx):
    """In-place inverse wavelet transform of a 2-D uint16 array (ny, nx)."""
    ny, nx = a.shape
    dec = _wdec14 if mx < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2)
        xs = np.arange(0, nx - p2 + 1, p2)
        if len(ys) and len(xs):
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            i00, i10 = dec(a[Y, X], a[Y + p, X])
            i01, i11 = dec(a[Y, X + p], a[Y + p, X + p])
            a[Y, X], a[Y, X + p] = dec(i00, i01)
            a[Y + p, X], a[Y + p, X + p] = dec(i10, i11)
        if nx & p and len(ys):
            x = len(xs) * p2  # first column the blocks did not cover
            v0, v1 = dec(a[ys, x], a[ys + p, x])
            a[ys, x], a[ys + p, x] = v0, v1
        if ny & p:
            y = len(ys) * p2
            if len(xs):
                v0, v1 = dec(a[y, xs], a[y, xs + p])
                a[y, xs], a[y, xs + p] = v0, v1
        p2 = p
        p >>= 1


def _piz_block(buf, nx, ny, sizes):
    """`sizes`: 16-bit words per pixel of each channel (1 = HALF, 2 = FLOAT / UINT).  Returns the raw little-endian scanline bytes."""
    n_raw = sum(nx * ny * s for s in sizes)
    mn, mxz = struct.unpack("<HH", buf[:4])
    bitmap = bytearray(8192)
    pos = 4
    if mn <= mxz:
        bitmap[mn:mxz + 1] = buf[pos:pos + mxz - mn + 1]
        pos += mxz - mn + 1
    lut = np.zeros(65536, dtype=np.uint16)
    k = 0
    for i in range(65536):
        if i == 0 or (bitmap[i >> 3] & (1 << (i & 7))):
            lut[k] = i
            k += 1
    max_value = k - 1
    length, = struct.unpack("<i", buf[pos:pos + 4])
    raw = _huf_uncompress(buf[pos + 4:pos + 4 + length], n_raw)
    chans, off = [], 0
    for s in sizes:
        c = raw[off:off + nx * ny * s].reshape(ny, nx, s).copy()
        off += nx * ny * s
        for j in range(s):
            plane = np.ascontiguousarray(c[:, :, j])
            _wav2_decode(plane, max_value)
            c[:, :, j] = plane
        chans.append(lut[c])
    rows = []
    for y in range(ny):
        for c in chans:
            rows.append(c[y].reshape(-1))
    return np.concatenate(rows).astype("<u2").tobytes()


def _zip_block(buf):
    raw = bytearray(zlib.decompress(buf))
    for i in range(1, len(raw)):  # predictor
        raw[i] = (raw[i - 1] + raw[i] - 128) & 0xFF
    half = (len(raw) + 1) // 2
    out = bytearray(len(raw))
    out[0::2] = raw[:half]
    out[1::2] = raw[half:]
    return bytes(out)


def read_exr(path):
    """Returns {channel name: float32 array (H, W)}."""
    d = open(path, "rb").read()
    pos, chans, comp, (x0, y0, x1, y1) = _read_header(d)
    W, H = x1 - x0 + 1, y1 - y0 + 1
    lines = {0: 1, 2: 1, 3: 16, 4: 32}[comp]
    n_blocks = (H + lines - 1) // lines
    offsets = struct.unpack("<%dQ" % n_blocks, d[pos:pos + 8 * n_blocks])
    sizes = [1 if t == 1 else 2 for _, t in chans]
    out = {name: np.zeros((H, W), np.float32) for name, _ in chans}
    for off in offsets:
        y, size = struct.unpack("<ii", d[off:off + 8])
        buf = d[off + 8:off + 8 + size]
        ny = min(lines, y1 - y + 1)
        expect = sum(W * ny * s * 2 for s in sizes)
        if size == expect or comp == 0:
            raw = buf
        elif comp == 4:
            raw = _piz_block(buf, W, ny, sizes)
        else:
            raw = _zip_block(buf)
        p = 0
        for row in range(ny):
            for (name, t), s in zip(chans, sizes):
                n = W * s * 2
                seg = raw[p:p + n]
                p += n
                if t == 1:
                    vals = np.frombuffer(seg, dtype="<f2").astype(np.float32)
                elif t == 2:
                    vals = np.frombuffer(seg, dtype="<f4")
                else:
                    vals = np.frombuffer(seg, dtype="<u4").astype(np.float32)
                out[name][y - y0 + row] = vals
    return out
