"""CPU oracle for the ALICE-Codec hot path -- TEST INFRASTRUCTURE ONLY.

ctypes front end over ``oracle/alice_oracle.c`` (a scalar C restatement of the
reference Rust sources; every C function cites the file:line it follows).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product path under
``alice-codec_amd/`` never does, and fails loudly without its HIP library.

Pinning status: stage level pinned by the reference's own exact-value tests;
whole-bitstream ``.alc`` bytes are **parity unpinned** (no reference golden
exists and the Rust crate cannot be built here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libalice_oracle.so")
_lock = threading.Lock()
_lib = None

CDF53, CDF97, HAAR = 0, 1, 2
OK = 0
ERR_INVALID_BUFFER_SIZE = 1
ERR_INVALID_DIMENSIONS = 2
ERR_DIMENSION_OVERFLOW = 3
ERR_INVALID_BITSTREAM = 4
ERR_INVALID_QUANT_STEP = 5
ERR_REFERENCE_DIVERGES = 6

_u8p = C.POINTER(C.c_uint8)
_i16p = C.POINTER(C.c_int16)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


class OracleError(Exception):
    def __init__(self, code: int):
        super().__init__(f"oracle error code {code}")
        self.code = code


class _FreqTable(C.Structure):
    _fields_ = [
        ("n_symbols", C.c_size_t),
        ("cum_freq", C.POINTER(C.c_uint16)),
        ("freq", C.POINTER(C.c_uint16)),
        ("cum_to_sym", C.c_uint8 * 4096),
    ]


class _FastQ(C.Structure):
    _fields_ = [("reciprocal", C.c_uint64), ("shift", C.c_uint32), ("step", C.c_int32),
                ("dead_zone", C.c_int32)]


def build(force: bool = False, so_path: str | None = None, cflags: str | None = None) -> str:
    """Compile the C oracle (gcc).  Returns the path of the shared object."""
    so = so_path or _SO
    src = os.path.join(_HERE, "alice_oracle.c")
    hdr = os.path.join(_HERE, "alice_oracle.h")
    stale = (not os.path.exists(so)) or any(
        os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        os.makedirs(os.path.dirname(so), exist_ok=True)
        flags = (cflags or "-O3 -fPIC -std=c11").split()
        subprocess.check_call(["gcc", *flags, "-pthread", "-shared", "-o", so, src, "-lm"])
    return so


def _bind(lib):
    lib.ao_rgb_bytes_to_ycocg_r.argtypes = [_u8p, C.c_size_t, _i16p, _i16p, _i16p, C.c_size_t]
    lib.ao_ycocg_r_to_rgb_bytes.argtypes = [_i16p, _i16p, _i16p, C.c_size_t, _u8p, C.c_size_t]
    for name in ("ao_wavelet1d_forward", "ao_wavelet1d_inverse"):
        getattr(lib, name).argtypes = [C.c_int, _i32p, C.c_size_t]
        getattr(lib, name).restype = None
    for name in ("ao_wavelet2d_forward", "ao_wavelet2d_inverse"):
        getattr(lib, name).argtypes = [C.c_int, _i32p, C.c_size_t, C.c_size_t]
        getattr(lib, name).restype = None
    for name in ("ao_wavelet3d_forward", "ao_wavelet3d_inverse"):
        getattr(lib, name).argtypes = [C.c_int, _i32p, C.c_size_t, C.c_size_t, C.c_size_t]
        getattr(lib, name).restype = None
    lib.ao_quantize.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.ao_quantize.restype = C.c_int32
    lib.ao_dequantize.argtypes = [C.c_int32, C.c_int32]
    lib.ao_dequantize.restype = C.c_int32
    lib.ao_quantize_buffer.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, C.c_size_t]
    lib.ao_quantize_buffer.restype = None
    lib.ao_dequantize_buffer.argtypes = [C.c_int32, _i32p, _i32p, C.c_size_t]
    lib.ao_dequantize_buffer.restype = None
    lib.ao_fast_quantizer_new.argtypes = [C.c_int32, C.POINTER(_FastQ)]
    lib.ao_fast_quantizer_with_dead_zone.argtypes = [C.c_int32, C.c_int32, C.POINTER(_FastQ)]
    lib.ao_fast_quantize.argtypes = [C.POINTER(_FastQ), C.c_int32]
    lib.ao_fast_quantize.restype = C.c_int32
    lib.ao_fast_quantize_buffer.argtypes = [C.POINTER(_FastQ), _i32p, _i32p, C.c_size_t]
    lib.ao_fast_quantize_buffer.restype = None
    lib.ao_to_symbols.argtypes = [_i32p, _u8p, C.c_size_t]
    lib.ao_to_symbols.restype = None
    lib.ao_from_symbols.argtypes = [_u8p, _i32p, C.c_size_t]
    lib.ao_from_symbols.restype = None
    lib.ao_build_histogram.argtypes = [_u8p, C.c_size_t, _u32p]
    lib.ao_build_histogram.restype = None
    lib.ao_freq_table_from_histogram.argtypes = [_u32p, C.c_size_t, C.POINTER(_FreqTable)]
    lib.ao_freq_table_uniform.argtypes = [C.c_size_t, C.POINTER(_FreqTable)]
    lib.ao_freq_table_free.argtypes = [C.POINTER(_FreqTable)]
    lib.ao_freq_table_free.restype = None
    lib.ao_rans_encode.argtypes = [_u8p, C.c_size_t, C.POINTER(_FreqTable), C.POINTER(_u8p),
                                   C.POINTER(C.c_size_t)]
    lib.ao_rans_decode.argtypes = [_u8p, C.c_size_t, C.c_size_t, C.POINTER(_FreqTable), _u8p]
    lib.ao_rans_decode.restype = None
    lib.ao_rans_encode_interleaved.argtypes = lib.ao_rans_encode.argtypes
    lib.ao_rans_decode_interleaved.argtypes = [_u8p, C.c_size_t, C.c_size_t,
                                               C.POINTER(_FreqTable), _u8p]
    lib.ao_quality_to_step.argtypes = [C.c_uint8]
    lib.ao_quality_to_step.restype = C.c_int32
    lib.ao_encode.argtypes = [_u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint8,
                              C.c_int, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    lib.ao_decode.argtypes = [_u8p, C.c_size_t, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    lib.ao_encode_par3.argtypes = lib.ao_encode.argtypes
    lib.ao_decode_par3.argtypes = lib.ao_decode.argtypes
    lib.ao_encode_symbols.argtypes = [_u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint8, C.c_int, C.POINTER(_u8p), C.POINTER(C.c_size_t)]
    lib.ao_ssim.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]
    lib.ao_ms_ssim.argtypes = lib.ao_ssim.argtypes
    lib.ao_rdo_target_bpp.argtypes = [C.c_uint8]
    lib.ao_rdo_target_bpp.restype = C.c_double
    lib.ao_subband_quant_strength.argtypes = [C.c_int]
    lib.ao_rdo_estimate_variance.argtypes = [_i32p, C.c_size_t]
    lib.ao_rdo_estimate_variance.restype = C.c_double
    lib.ao_rdo_compute_quantizer.argtypes = [C.c_double, _i32p, C.c_size_t, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.ao_rdo_compute_quantizer.restype = None
    lib.ao_psnr.argtypes = [_u8p, _u8p, C.c_size_t]
    lib.ao_psnr.restype = C.c_double
    lib.ao_free.argtypes = [C.c_void_p]
    lib.ao_free.restype = None
    return lib


def lib(so_path: str | None = None):
    """The loaded oracle library (built on first use)."""
    global _lib
    if so_path is not None:
        return _bind(C.CDLL(so_path))
    with _lock:
        if _lib is None:
            _lib = _bind(C.CDLL(build()))
        return _lib


def _copy_out(ptr, n: int) -> np.ndarray:
    """Copies n bytes at a C pointer into a fresh array.  (ctypes.string_at takes its size as a C int on this
    Python, which silently truncates buffers of 2 GiB and more.)"""
    if n == 0:
        return np.zeros(0, np.uint8)
    addr = C.cast(ptr, C.c_void_p).value
    return np.frombuffer((C.c_uint8 * n).from_address(addr), dtype=np.uint8).copy()


def _u8(a) -> np.ndarray:
    return np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray, memoryview)) else a, dtype=np.uint8)


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


def _check(rc: int):
    if rc != OK:
        raise OracleError(rc)


def _take(pp, n) -> bytes:
    try:
        return _copy_out(pp, n).tobytes()
    finally:
        lib().ao_free(C.cast(pp, C.c_void_p))


# ---- stage wrappers ------------------------------------------------------------

def rgb_to_ycocg_r(rgb):
    rgb = _u8(rgb).reshape(-1)
    n = rgb.size // 3
    y = np.zeros(n, np.int16); co = np.zeros(n, np.int16); cg = np.zeros(n, np.int16)
    _check(lib().ao_rgb_bytes_to_ycocg_r(_ptr(rgb, _u8p), rgb.size, _ptr(y, _i16p), _ptr(co, _i16p),
                                         _ptr(cg, _i16p), n))
    return y, co, cg


def ycocg_r_to_rgb(y, co, cg):
    y = np.ascontiguousarray(y, np.int16); co = np.ascontiguousarray(co, np.int16)
    cg = np.ascontiguousarray(cg, np.int16)
    out = np.zeros(y.size * 3, np.uint8)
    _check(lib().ao_ycocg_r_to_rgb_bytes(_ptr(y, _i16p), _ptr(co, _i16p), _ptr(cg, _i16p), y.size,
                                         _ptr(out, _u8p), out.size))
    return out


def wavelet1d(kind: int, signal, inverse: bool = False) -> np.ndarray:
    s = np.array(signal, dtype=np.int32).reshape(-1).copy()
    fn = lib().ao_wavelet1d_inverse if inverse else lib().ao_wavelet1d_forward
    fn(kind, _ptr(s, _i32p), s.size)
    return s


def wavelet2d(kind: int, image, width: int, height: int, inverse: bool = False) -> np.ndarray:
    s = np.array(image, dtype=np.int32).reshape(-1).copy()
    assert s.size == width * height
    fn = lib().ao_wavelet2d_inverse if inverse else lib().ao_wavelet2d_forward
    fn(kind, _ptr(s, _i32p), width, height)
    return s


def wavelet3d(kind: int, volume, width: int, height: int, depth: int, inverse: bool = False) -> np.ndarray:
    s = np.array(volume, dtype=np.int32).reshape(-1).copy()
    assert s.size == width * height * depth
    fn = lib().ao_wavelet3d_inverse if inverse else lib().ao_wavelet3d_forward
    fn(kind, _ptr(s, _i32p), width, height, depth)
    return s


def quantize(step: int, value: int, dead_zone: int | None = None) -> int:
    return lib().ao_quantize(step, step if dead_zone is None else dead_zone, value)


def dequantize(step: int, q: int) -> int:
    return lib().ao_dequantize(step, q)


def quantize_buffer(step: int, values, dead_zone: int | None = None) -> np.ndarray:
    v = np.ascontiguousarray(values, np.int32).reshape(-1)
    out = np.zeros_like(v)
    lib().ao_quantize_buffer(step, step if dead_zone is None else dead_zone, _ptr(v, _i32p),
                             _ptr(out, _i32p), v.size)
    return out


def dequantize_buffer(step: int, values) -> np.ndarray:
    v = np.ascontiguousarray(values, np.int32).reshape(-1)
    out = np.zeros_like(v)
    lib().ao_dequantize_buffer(step, _ptr(v, _i32p), _ptr(out, _i32p), v.size)
    return out


def fast_quantizer(step: int, dead_zone: int | None = None) -> _FastQ:
    q = _FastQ()
    if dead_zone is None:
        _check(lib().ao_fast_quantizer_new(step, C.byref(q)))
    else:
        _check(lib().ao_fast_quantizer_with_dead_zone(step, dead_zone, C.byref(q)))
    return q


def fast_quantize(q: _FastQ, value: int) -> int:
    return lib().ao_fast_quantize(C.byref(q), value)


def fast_quantize_buffer(q: _FastQ, values) -> np.ndarray:
    v = np.ascontiguousarray(values, np.int32).reshape(-1)
    out = np.zeros_like(v)
    lib().ao_fast_quantize_buffer(C.byref(q), _ptr(v, _i32p), _ptr(out, _i32p), v.size)
    return out


def to_symbols(coeffs) -> np.ndarray:
    v = np.ascontiguousarray(coeffs, np.int32).reshape(-1)
    out = np.zeros(v.size, np.uint8)
    lib().ao_to_symbols(_ptr(v, _i32p), _ptr(out, _u8p), v.size)
    return out


def from_symbols(symbols) -> np.ndarray:
    s = _u8(symbols).reshape(-1)
    out = np.zeros(s.size, np.int32)
    lib().ao_from_symbols(_ptr(s, _u8p), _ptr(out, _i32p), s.size)
    return out


def build_histogram(symbols) -> np.ndarray:
    s = _u8(symbols).reshape(-1)
    out = np.zeros(256, np.uint32)
    lib().ao_build_histogram(_ptr(s, _u8p), s.size, _ptr(out, _u32p))
    return out


class FrequencyTable:
    """rans.rs:85-219 FrequencyTable (from_histogram / uniform)."""

    def __init__(self, histogram=None, uniform: int | None = None):
        self._t = _FreqTable()
        if uniform is not None:
            _check(lib().ao_freq_table_uniform(uniform, C.byref(self._t)))
        else:
            h = np.ascontiguousarray(histogram, np.uint32).reshape(-1)
            _check(lib().ao_freq_table_from_histogram(_ptr(h, _u32p), h.size, C.byref(self._t)))

    def __del__(self):
        try:
            lib().ao_freq_table_free(C.byref(self._t))
        except Exception:
            pass

    def __len__(self):
        return self._t.n_symbols

    @property
    def freq(self) -> np.ndarray:
        return np.array([self._t.freq[i] for i in range(len(self))], np.uint16)

    @property
    def cum_freq(self) -> np.ndarray:
        return np.array([self._t.cum_freq[i] for i in range(len(self))], np.uint16)

    @property
    def cum_to_sym(self) -> np.ndarray:
        return np.frombuffer(bytes(self._t.cum_to_sym), np.uint8).copy()


def rans_encode(symbols, table: FrequencyTable, interleaved: bool = False) -> bytes:
    s = _u8(symbols).reshape(-1)
    out = _u8p(); n = C.c_size_t()
    fn = lib().ao_rans_encode_interleaved if interleaved else lib().ao_rans_encode
    _check(fn(_ptr(s, _u8p), s.size, C.byref(table._t), C.byref(out), C.byref(n)))
    return _take(out, n.value)


def rans_decode(data, n: int, table: FrequencyTable, interleaved: bool = False) -> np.ndarray:
    d = _u8(data).reshape(-1)
    out = np.zeros(n, np.uint8)
    if interleaved:
        _check(lib().ao_rans_decode_interleaved(_ptr(d, _u8p), d.size, n, C.byref(table._t),
                                                _ptr(out, _u8p)))
    else:
        lib().ao_rans_decode(_ptr(d, _u8p), d.size, n, C.byref(table._t), _ptr(out, _u8p))
    return out


class RansEncoder:
    """rans.rs:238-309 RansEncoder as an object: encode / encode_symbols may be called any number of times before finish."""

    def __init__(self):
        L = lib()
        L.ao_rans_encoder_new.restype = C.c_void_p
        L.ao_rans_encoder_state.restype = C.c_uint32
        self._h = C.c_void_p(L.ao_rans_encoder_new())

    def encode(self, cum_freq: int, freq: int) -> None:
        _check(lib().ao_rans_encoder_encode(self._h, C.c_uint16(cum_freq), C.c_uint16(freq)))

    def encode_symbols(self, symbols, table: FrequencyTable) -> None:
        s = _u8(symbols).reshape(-1)
        _check(lib().ao_rans_encoder_encode_symbols(self._h, _ptr(s, _u8p), C.c_size_t(s.size), C.byref(table._t)))

    @property
    def state(self) -> int:
        return int(lib().ao_rans_encoder_state(self._h))

    def finish(self) -> bytes:
        out = _u8p(); n = C.c_size_t()
        h, self._h = self._h, None
        _check(lib().ao_rans_encoder_finish(h, C.byref(out), C.byref(n)))
        return _take(out, n.value)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ao_rans_encoder_free(self._h)


class RansDecoder:
    """rans.rs:321-389 RansDecoder as an object: decode_n continues from the current position."""

    def __init__(self, data):
        L = lib()
        L.ao_rans_decoder_new.restype = C.c_void_p
        L.ao_rans_decoder_state.restype = C.c_uint32
        L.ao_rans_decoder_pos.restype = C.c_size_t
        d = _u8(data).reshape(-1)
        self._h = C.c_void_p(L.ao_rans_decoder_new(_ptr(d, _u8p), C.c_size_t(d.size)))

    def decode_n(self, n: int, table: FrequencyTable) -> np.ndarray:
        out = np.zeros(n, np.uint8)
        lib().ao_rans_decoder_decode_n(self._h, C.c_size_t(n), C.byref(table._t), _ptr(out, _u8p))
        return out

    def is_empty(self) -> bool:
        return bool(lib().ao_rans_decoder_is_empty(self._h))

    @property
    def state(self) -> int:
        return int(lib().ao_rans_decoder_state(self._h))

    @property
    def pos(self) -> int:
        return int(lib().ao_rans_decoder_pos(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ao_rans_decoder_free(self._h)


def quality_to_step(q: int) -> int:
    return lib().ao_quality_to_step(q)


# ---- pipeline ----------------------------------------------------------------------

def encode(rgb, width: int, height: int, frames: int, quality: int, wavelet: int = CDF53,
           _lib=None, three_threads: bool = False) -> bytes:
    """FrameEncoder::with_wavelet(quality, wavelet).encode(rgb, w, h, f).to_bytes().
    three_threads: the non-reference variant with Y, Co, Cg on three threads (same bytes)."""
    L = _lib or lib()
    r = _u8(rgb).reshape(-1)
    out = _u8p(); n = C.c_size_t()
    fn = L.ao_encode_par3 if three_threads else L.ao_encode
    _check(fn(_ptr(r, _u8p), r.size, width, height, frames, quality, wavelet,
              C.byref(out), C.byref(n)))
    try:
        return _copy_out(out, n.value).tobytes()
    finally:
        L.ao_free(C.cast(out, C.c_void_p))


def decode(alc, _lib=None, three_threads: bool = False) -> np.ndarray:
    """FrameDecoder::new().decode(&EncodedChunk::from_bytes(alc)?)."""
    L = _lib or lib()
    d = _u8(alc).reshape(-1)
    out = _u8p(); n = C.c_size_t()
    _check((L.ao_decode_par3 if three_threads else L.ao_decode)(_ptr(d, _u8p), d.size, C.byref(out), C.byref(n)))
    try:
        return _copy_out(out, n.value)
    finally:
        L.ao_free(C.cast(out, C.c_void_p))


def encode_symbols(rgb, width: int, height: int, frames: int, quality: int, wavelet: int = CDF53):
    """Front half of encode: (3, padded_pixels) u8 symbols, channel-major."""
    r = _u8(rgb).reshape(-1)
    out = _u8p(); n = C.c_size_t()
    _check(lib().ao_encode_symbols(_ptr(r, _u8p), r.size, width, height, frames, quality, wavelet,
                                   C.byref(out), C.byref(n)))
    if n.value == 0:
        return np.zeros((3, 0), np.uint8)
    raw = _take(out, 3 * n.value)
    return np.frombuffer(raw, np.uint8).reshape(3, n.value).copy()


def ssim(a, b, width: int, height: int, multi_scale: bool = False) -> float:
    a = _u8(a).reshape(-1); b = _u8(b).reshape(-1)
    out = C.c_double()
    _check((lib().ao_ms_ssim if multi_scale else lib().ao_ssim)(_ptr(a, _u8p), a.size, _ptr(b, _u8p), b.size, width, height, C.byref(out)))
    return out.value


def rdo_target_bpp(quality: int) -> float:
    return lib().ao_rdo_target_bpp(min(int(quality), 255))


def subband_quant_strength(subband: int) -> int:
    return lib().ao_subband_quant_strength(int(subband))


def rdo_estimate_variance(coeffs) -> float:
    c = np.ascontiguousarray(coeffs, dtype=np.int32).reshape(-1)
    return lib().ao_rdo_estimate_variance(_ptr(c, _i32p), c.size)


def rdo_compute_quantizer(target_bpp: float, coeffs, subband: int):
    """AnalyticalRDO::compute_quantizer -> (step, dead_zone)."""
    c = np.ascontiguousarray(coeffs, dtype=np.int32).reshape(-1)
    st, dz = C.c_int32(), C.c_int32()
    lib().ao_rdo_compute_quantizer(float(target_bpp), _ptr(c, _i32p), c.size, int(subband), C.byref(st), C.byref(dz))
    return st.value, dz.value


def psnr(a, b) -> float:
    a = _u8(a).reshape(-1); b = _u8(b).reshape(-1)
    if a.size != b.size:
        return -1.0
    return lib().ao_psnr(_ptr(a, _u8p), _ptr(b, _u8p), a.size)


def make_gradient(w: int, h: int, f: int) -> np.ndarray:
    """The reference tests' input generator (src/pipeline.rs:673-683), as data."""
    n = w * h * f
    v = ((np.arange(n, dtype=np.int64) * 7) % 256).astype(np.uint8)
    rgb = np.empty(n * 3, np.uint8)
    rgb[0::3] = v
    rgb[1::3] = v + np.uint8(30)
    rgb[2::3] = v + np.uint8(60)
    return rgb
