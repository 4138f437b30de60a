"""Second, independently structured restatement of the reference hot path (numpy).

TEST INFRASTRUCTURE ONLY.  Purpose: catch reading errors in ``alice_oracle.c``.
Where the C oracle follows the reference's loop structure (gather a column,
run the 1-D transform, scatter it back), this file states each lifting step as
one whole-axis array operation and the entropy coder as plain Python integers,
so the two share no code shape.  Small inputs only (the rANS loops are Python).

Citations are into the reference checkout (file:line).
"""
from __future__ import annotations

import numpy as np

CDF53, CDF97, HAAR = 0, 1, 2

# src/wavelet.rs:66-127 -- (coeff, predict) lists, scale 2^12
STEPS = {
    CDF97: [(-6497, True), (-217, False), (3616, True), (1817, False)],
    HAAR: [(-4096, True), (2048, False)],
    CDF53: [(-4096, True), (1024, False)],
}


def _wrap32(a):
    return ((np.asarray(a, dtype=np.int64) + 2**31) % 2**32 - 2**31)


def _delta(a, b, coeff):
    # src/wavelet.rs:193-194: avg wraps in i32; product/rounding in i64; floor shift; `as i32`
    avg = _wrap32(a.astype(np.int64) + b.astype(np.int64))
    return _wrap32((avg * coeff + 4096) >> 13)


def _lift_axis(v: np.ndarray, axis: int, steps, inverse: bool) -> np.ndarray:
    """One 1-D transform along ``axis`` for every line at once (n = v.shape[axis])."""
    v = np.moveaxis(v.astype(np.int64), axis, 0).copy()
    n = v.shape[0]
    if n < 2:  # src/wavelet.rs:135,159
        return np.moveaxis(v, 0, axis)
    half = n // 2
    if inverse:  # interleave, src/wavelet.rs:236-248 (odd tail becomes 0)
        t = np.zeros_like(v)
        t[0:2 * half:2] = v[:half]
        t[1:2 * half:2] = v[half:2 * half]
        v = t
        steps = [(-c, p) for (c, p) in reversed(steps)]  # src/wavelet.rs:167-174
    for coeff, predict in steps:
        even = v[0:2 * half:2]
        odd = v[1:2 * half:2]
        if predict:  # src/wavelet.rs:184-196
            right = np.empty_like(even)
            right[:-1] = even[1:]
            # i*2+2 < n ?  for the last pair: 2*half < n only when n is odd
            right[-1] = v[2 * half] if 2 * half < n else even[-1]
            v[1:2 * half:2] = _wrap32(odd + _delta(even, right, coeff))
        else:  # src/wavelet.rs:205-216
            left = np.empty_like(odd)
            left[1:] = odd[:-1]
            left[0] = odd[0]
            v[0:2 * half:2] = _wrap32(even + _delta(left, odd, coeff))
    if not inverse:  # deinterleave, src/wavelet.rs:220-233 (odd tail dropped -> 0)
        t = np.zeros_like(v)
        t[:half] = v[0:2 * half:2]
        t[half:2 * half] = v[1:2 * half:2]
        v = t
    return np.moveaxis(v, 0, axis)


def wavelet1d(kind, signal, inverse=False):
    return _lift_axis(np.asarray(signal, np.int64), 0, STEPS[kind], inverse).astype(np.int32)


def wavelet3d(kind, volume, width, height, depth, inverse=False):
    v = np.asarray(volume, np.int64).reshape(depth, height, width)
    st = STEPS[kind]
    if not inverse:  # src/wavelet.rs:392-438: rows, columns (per frame), then temporal
        v = _lift_axis(v, 2, st, False)
        v = _lift_axis(v, 1, st, False)
        v = _lift_axis(v, 0, st, False)
    else:  # src/wavelet.rs:441-484: temporal, columns, rows
        v = _lift_axis(v, 0, st, True)
        v = _lift_axis(v, 1, st, True)
        v = _lift_axis(v, 2, st, True)
    return v.reshape(-1).astype(np.int32)


def rgb_to_ycocg_r(rgb):
    # src/color.rs:221-228 (values stay far inside i16, no wrap possible from u8 input)
    p = np.asarray(rgb, np.int64).reshape(-1, 3)
    r, g, b = p[:, 0], p[:, 1], p[:, 2]
    co = r - b
    t = b + (co >> 1)
    cg = g - t
    y = t + (cg >> 1)
    return y, co, cg


def _wrap16(a):
    return ((np.asarray(a, np.int64) + 2**15) % 2**16 - 2**15)


def ycocg_r_to_rgb(y, co, cg):
    # src/color.rs:266-273, i16 wrapping arithmetic then clamp
    y, co, cg = (_wrap16(a) for a in (y, co, cg))
    t = _wrap16(y - (cg >> 1))
    g = _wrap16(cg + t)
    b = _wrap16(t - (co >> 1))
    r = _wrap16(co + b)
    out = np.stack([r, g, b], axis=1)
    return np.clip(out, 0, 255).astype(np.uint8).reshape(-1)


def quantize(v, step, dead_zone):
    # src/quant.rs:89-97: trunc-toward-zero division
    v = np.asarray(v, np.int64)
    half = int(dead_zone / 2)  # trunc toward zero like Rust `/`
    mag = np.abs(v)
    pos = (v - half)
    neg = (v + half)
    adj = np.where(v >= 0, pos, neg)
    q = np.sign(adj) * (np.abs(adj) // step)
    return np.where(mag < dead_zone, 0, q)


def to_symbols(q):
    # src/quant.rs:555-560
    q = np.asarray(q, np.int64)
    s = np.where(q == 0, 0, np.where(q > 0, 2 * q - 1, -2 * q))
    return (s % 256).astype(np.uint8)


def from_symbols(s):
    # src/quant.rs:580-588
    s = np.asarray(s, np.int64)
    return np.where(s == 0, 0, np.where(s % 2 == 1, (s + 1) // 2, -(s // 2)))


def freq_table(hist):
    """src/rans.rs:102-150 (and :158-189 for the all-zero case) -> (cum[256], freq[256], cum_to_sym[4096])"""
    hist = [int(x) for x in hist]
    n = len(hist)
    total = sum(hist)
    cum, freq = [], []
    if total == 0:
        fps = (4096 // n) & 0xFFFF
        c = 0
        for _ in range(n):
            cum.append(c); freq.append(fps); c = (c + fps) & 0xFFFF
        freq[-1] = (4096 - cum[-1]) & 0xFFFF
    else:
        c = 0; nt = 0
        for count in hist:
            f = 1 if count == 0 else max(count * 4096 // total, 1)
            nt += f
            cum.append(c & 0xFFFF); freq.append(f & 0xFFFF); c += f
        if nt != 4096:
            freq[-1] = (freq[-1] + (4096 - nt)) & 0xFFFF
    c2s = [0] * 4096
    for sym in range(n):
        for slot in range(cum[sym], min(cum[sym] + freq[sym], 4096)):
            c2s[slot] = sym & 0xFF
    return cum, freq, c2s


def rans_encode(symbols, cum, freq) -> bytes:
    # src/rans.rs:269-308
    x = 1 << 23
    out = bytearray()
    for s in reversed([int(v) for v in symbols]):
        f, c = freq[s], cum[s]
        if f == 0:
            raise ZeroDivisionError("reference diverges: freq 0")
        x_max = (((1 << 23) >> 12) << 8) * f
        while x >= x_max:
            out.append(x & 0xFF); x >>= 8
        x = (((x // f) << 12) + (x % f) + c) & 0xFFFFFFFF
    for k in range(4):
        out.append((x >> (8 * k)) & 0xFF)
    out.reverse()
    return bytes(out)


def rans_decode(data: bytes, n: int, cum, freq, c2s):
    # src/rans.rs:330-381
    data = bytes(data)
    x, pos = 0, 0
    if len(data) >= 4:
        x = int.from_bytes(data[:4], "big"); pos = 4
    out = []
    for _ in range(n):
        slot = x & 4095
        s = c2s[slot]
        x = (freq[s] * (x >> 12) + slot - cum[s]) & 0xFFFFFFFF
        while x < (1 << 23) and pos < len(data):
            x = ((x << 8) | data[pos]) & 0xFFFFFFFF; pos += 1
        out.append(s)
    return np.array(out, np.uint8)


def quality_to_step(q):  # src/pipeline.rs:456-457
    return max(64 - (min(q, 100) * 63) // 100, 1)


def _pad(ch, w, h, f):
    # src/pipeline.rs:77-114 stated as edge replication
    pw, ph = w + (w & 1), h + (h & 1)
    pf = 2 if f == 1 else f + (f & 1)
    v = np.asarray(ch).reshape(f, h, w)
    v = np.pad(v, ((0, pf - f), (0, ph - h), (0, pw - w)), mode="edge")
    return v, pw, ph, pf


def encode(rgb, w, h, f, quality, kind=CDF53) -> bytes:
    """src/pipeline.rs:377-507 + :200-226."""
    import struct
    rgb = np.asarray(rgb, np.uint8).reshape(-1)
    n = w * h * f
    hdr = bytearray(b"ALCC" + bytes([1, kind]) + struct.pack("<III", w, h, f))
    if n == 0:
        assert rgb.size == 0
        for _ in range(3):
            hdr += struct.pack("<IiiI", 0, 1, 1, 0) + bytes(1024)
        return bytes(hdr)
    assert rgb.size == 3 * n and w > 0 and h > 0
    step = quality_to_step(quality)
    payload = bytearray()
    for ch in rgb_to_ycocg_r(rgb):
        v, pw, ph, pf = _pad(ch, w, h, f)
        coef = wavelet3d(kind, v, pw, ph, pf)
        sym = to_symbols(quantize(coef, step, step))
        hist = np.bincount(sym, minlength=256)
        cum, freq, _ = freq_table(hist)
        stream = rans_encode(sym, cum, freq)
        hdr += struct.pack("<IiiI", len(stream), step, step, pw * ph * pf)
        hdr += struct.pack("<256I", *[int(x) for x in hist])
        payload += stream
    return bytes(hdr) + bytes(payload)


def decode(alc: bytes) -> np.ndarray:
    """src/pipeline.rs:235-313 + :537-624 (valid input only)."""
    import struct
    alc = bytes(alc)
    assert alc[:4] == b"ALCC" and alc[4] == 1
    kind = alc[5]
    w, h, f = struct.unpack_from("<III", alc, 6)
    off = 18
    hdrs = []
    for _ in range(3):
        clen, step, dz, nsym = struct.unpack_from("<IiiI", alc, off)
        hist = struct.unpack_from("<256I", alc, off + 16)
        hdrs.append((clen, step, dz, nsym, hist))
        off += 1040
    if w * h * f == 0:
        return np.zeros(0, np.uint8)
    pw, ph = w + (w & 1), h + (h & 1)
    pf = 2 if f == 1 else f + (f & 1)
    chans = []
    for clen, step, dz, nsym, hist in hdrs:
        assert nsym == pw * ph * pf
        cum, freq, c2s = freq_table(hist)
        sym = rans_decode(alc[off:off + clen], nsym, cum, freq, c2s)
        off += clen
        q = from_symbols(sym)
        coef = _wrap32(q * step)
        vol = wavelet3d(kind, coef, pw, ph, pf, inverse=True).reshape(pf, ph, pw)
        chans.append(_wrap16(vol[:f, :h, :w].reshape(-1)))
    return ycocg_r_to_rgb(*chans)
