"""Minimal decoders for the files the export entry points write (PNG 8/16-bit, BMP 24/32 bpp, PAM P7) — test
infrastructure only. PIL is used as a second opinion where it can represent the image (8-bit)."""
import struct
import zlib

import numpy as np


def decode_png(path):
    """Minimal PNG decoder (non-interlaced, 8/16-bit, colour types 0/2/6): PIL flattens 16-bit RGB to 8."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        ln, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        pos += 12 + ln
    w, h, depth, ctype, _, _, interlace = hdr
    assert interlace == 0 and depth in (8, 16)
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    bpp = ch * depth // 8
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * bpp)
    out = np.zeros((h, w * bpp), dtype=np.uint8)
    prev = np.zeros(w * bpp, dtype=np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        cur = np.zeros(w * bpp, dtype=np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        elif f == 1:
            cur = line.copy()
            for k in range(bpp, w * bpp):
                cur[k] = (cur[k] + cur[k - bpp]) & 255
        elif f in (3, 4):
            for k in range(w * bpp):
                a = cur[k - bpp] if k >= bpp else 0
                b = prev[k]
                c = prev[k - bpp] if k >= bpp else 0
                if f == 3:
                    pred = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[k] = (line[k] + pred) & 255
        out[y] = cur
        prev = cur
    if depth == 16:
        return out.reshape(h, w, ch, 2).astype(np.uint16)[..., 0] << 8 | out.reshape(h, w, ch, 2)[..., 1]
    return out.reshape(h, w, ch)


def decode_bmp(path):
    """BITMAPINFOHEADER 24 bpp BI_RGB or BITMAPV4HEADER 32 bpp BI_BITFIELDS, bottom-up -> (H, W, 3|4) uint8 RGB(A)."""
    d = open(path, "rb").read()
    assert d[:2] == b"BM"
    size, off = struct.unpack("<I4xI", d[2:14])
    assert size == len(d)
    dib, w, h, planes, bpp, comp = struct.unpack("<IiiHHI", d[14:34])
    assert planes == 1 and h > 0
    ch = bpp // 8
    row = (w * ch + 3) & ~3
    px = np.frombuffer(d[off:off + row * h], dtype=np.uint8).reshape(h, row)[::-1, :w * ch].reshape(h, w, ch)
    if ch == 3:
        assert comp == 0 and dib == 40
        return px[..., ::-1].copy()
    assert comp == 3 and dib == 108
    masks = struct.unpack("<4I", d[14 + 40:14 + 56])
    out = np.zeros((h, w, 4), dtype=np.uint8)
    word = px.astype(np.uint32)
    word = word[..., 0] | (word[..., 1] << 8) | (word[..., 2] << 16) | (word[..., 3] << 24)
    for c, m in enumerate(masks):
        shift = (m & -m).bit_length() - 1
        out[..., c] = (word & m) >> shift
    return out


def decode_pam(path):
    d = open(path, "rb").read()
    head, _, body = d.partition(b"ENDHDR\n")
    f = dict(line.split(None, 1) for line in head.decode().splitlines()[1:] if line)
    assert head.startswith(b"P7\n")
    w, h, depth, maxval = int(f["WIDTH"]), int(f["HEIGHT"]), int(f["DEPTH"]), int(f["MAXVAL"])
    assert maxval == 255 and f["TUPLTYPE"] == ("RGB_ALPHA" if depth == 4 else "RGB")
    assert len(body) == w * h * depth
    return np.frombuffer(body, dtype=np.uint8).reshape(h, w, depth).copy()
