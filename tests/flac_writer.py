"""A small FLAC *encoder* for the tests of audio/_flac.py (test infrastructure only): writes valid streams that exercise every
feature the decoder implements — chosen per frame by the caller — straight from the format specification (RFC 9639).  It is not a
compressor: predictor orders, Rice parameters and stereo modes are whatever the test asks for."""

import hashlib
import struct

import numpy as np


class Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, v: int, k: int):
        if k == 0:
            return
        self.acc = (self.acc << k) | (int(v) & ((1 << k) - 1))
        self.n += k
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, q: int):
        for _ in range(q):
            self.put(0, 1)
        self.put(1, 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    n = 2
    while v >= 1 << (5 * n + 1):
        n += 1
    out = [((0xFF << (8 - n)) & 0xFF) | (v >> (6 * (n - 1)))]
    for i in range(n - 2, -1, -1):
        out.append(0x80 | ((v >> (6 * i)) & 0x3F))
    return bytes(out)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def residual(bits: Bits, res, blocksize, order, po, rice2=False, escape_parts=()):
    bits.put(1 if rice2 else 0, 2)
    bits.put(po, 4)
    pb, esc = (5, 31) if rice2 else (4, 15)
    idx = 0
    for p in range(1 << po):
        cnt = (blocksize >> po) - (order if p == 0 else 0)
        part = [int(v) for v in res[idx : idx + cnt]]
        idx += cnt
        if p in escape_parts:
            nb = max([1] + [(abs(v) if v >= 0 else abs(v + 1)).bit_length() + 1 for v in part])
            bits.put(esc, pb)
            bits.put(nb, 5)
            for v in part:
                bits.put(v, nb)
            continue
        mean = (sum(abs(v) for v in part) / max(len(part), 1)) if part else 0
        k = min(max(int(mean).bit_length(), 0), esc - 1)
        bits.put(k, pb)
        for v in part:
            u = (v << 1) if v >= 0 else ((-v) << 1) - 1
            bits.unary(u >> k)
            bits.put(u & ((1 << k) - 1), k)


def subframe(bits: Bits, s, bps, kind, order=0, po=0, rice2=False, escape_parts=(), lpc=None, wasted=0):
    s = [int(v) for v in s]
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in s)
        s = [v >> wasted for v in s]
        bps -= wasted
    bits.put(0, 1)
    code = {"constant": 0, "verbatim": 1}.get(kind, None)
    if kind == "fixed":
        code = 8 + order
    if kind == "lpc":
        order = len(lpc[0])
        code = 32 + order - 1
    bits.put(code, 6)
    if wasted:
        bits.put(1, 1)
        bits.unary(wasted - 1)
    else:
        bits.put(0, 1)
    n = len(s)
    if kind == "constant":
        bits.put(s[0], bps)
    elif kind == "verbatim":
        for v in s:
            bits.put(v, bps)
    elif kind == "fixed":
        for v in s[:order]:
            bits.put(v, bps)
        c = FIXED[order]
        res = [s[i] - sum(c[k] * s[i - 1 - k] for k in range(order)) for i in range(order, n)]
        residual(bits, res, n, order, po, rice2, escape_parts)
    else:
        coefs, prec, shift = lpc
        for v in s[:order]:
            bits.put(v, bps)
        bits.put(prec - 1, 4)
        bits.put(shift, 5)
        for cf in coefs:
            bits.put(cf, prec)
        res = [s[i] - (sum(coefs[k] * s[i - 1 - k] for k in range(order)) >> shift) for i in range(order, n)]
        residual(bits, res, n, order, po, rice2, escape_parts)


BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
BPS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode(samples: np.ndarray, sample_rate: int, bps: int, frames: list[dict], with_md5=True, id3=False, total_known=True) -> bytes:
    """``samples`` int [n, channels]; ``frames``: per frame ``{"n": blocksize, "mode": "indep"|"ls"|"sr"|"ms", "sub": [subframe kwargs per channel]}``."""
    x = np.asarray(samples, np.int64)
    n, ch = x.shape
    out = bytearray()
    if id3:
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, 10]) + b"\x00" * 10
    out += b"fLaC"
    width = (bps + 7) // 8
    pcm = x.astype("<i4").view(np.uint8).reshape(n, ch, 4)[:, :, :width].tobytes()
    md5 = hashlib.md5(pcm).digest() if with_md5 else b"\x00" * 16
    sizes = [f["n"] for f in frames]
    si = struct.pack(">HH", min(sizes), max(sizes)) + b"\x00" * 6
    packed = (sample_rate << 44) | ((ch - 1) << 41) | ((bps - 1) << 36) | (n if total_known else 0)
    si += packed.to_bytes(8, "big") + md5
    out += bytes([0x00]) + len(si).to_bytes(3, "big") + si            # STREAMINFO, not last
    pad = b"\x00" * 7
    out += bytes([0x80 | 1]) + len(pad).to_bytes(3, "big") + pad      # PADDING, last
    pos = 0
    for fi, f in enumerate(frames):
        bs = f["n"]
        blk = x[pos : pos + bs]
        pos += bs
        hdr = Bits()
        hdr.put(0b11111111111110, 14)
        hdr.put(0, 1)
        hdr.put(0, 1)  # fixed block size stream: frame number
        code = BLOCK_CODES.get(bs, 6 if bs <= 256 else 7)
        hdr.put(code, 4)
        hdr.put(0, 4)  # sample rate from STREAMINFO
        mode = f.get("mode", "indep")
        hdr.put({"indep": ch - 1, "ls": 8, "sr": 9, "ms": 10}[mode], 4)
        hdr.put(BPS_CODES[bps] if f.get("explicit_bps") else 0, 3)
        hdr.put(0, 1)
        for b in utf8_number(f.get("number", fi)):
            hdr.put(b, 8)
        if code == 6:
            hdr.put(bs - 1, 8)
        elif code == 7:
            hdr.put(bs - 1, 16)
        head = bytes(hdr.out)
        body = Bits()
        chans = [blk[:, c] for c in range(ch)]
        extra = [0] * ch
        if mode == "ls":
            chans, extra = [chans[0], chans[0] - chans[1]], [0, 1]
        elif mode == "sr":
            chans, extra = [chans[0] - chans[1], chans[1]], [1, 0]
        elif mode == "ms":
            chans, extra = [(chans[0] + chans[1]) >> 1, chans[0] - chans[1]], [0, 1]
        for c in range(ch):
            subframe(body, chans[c], bps + extra[c], **f["sub"][c])
        body.align()
        frame = head + bytes([crc8(head)]) + bytes(body.out)
        out += frame + crc16(frame).to_bytes(2, "big")
    assert pos == n
    return bytes(out)
