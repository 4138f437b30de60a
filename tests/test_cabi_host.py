"""CPU-only checks of the product library's host side: it loads, exports every symbol
include/alice_codec.h declares, honours the reference FFI's null/failure conventions
(src/ffi.rs:357-484), parses and re-serialises `.alc` headers on the host, and fails
loudly -- not silently on a CPU -- when no GPU is present.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(codec):
    lib = codec.load_library()
    import glob
    hdr = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))))
    names = sorted(set(re.findall(r"\b(alice_codec_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 60
    for n in names:
        assert hasattr(lib, n), n
    for n in codec.EXPORTED_SYMBOLS_PART1:  # the 20 drop-in functions of src/ffi.rs
        assert n in names
    assert len(codec.EXPORTED_SYMBOLS_PART1) == 20


def test_version_and_string_free(codec):  # src/ffi.rs:467-475
    assert codec.version() == "0.1.2"


def test_null_safety(codec):  # src/ffi.rs:357-363, 432-446, 478-484
    lib = codec.load_library()
    lib.alice_codec_wavelet1d_forward(None, None, 0)
    lib.alice_codec_wavelet1d_inverse(None, None, 0)
    lib.alice_codec_wavelet1d_destroy(None)
    lib.alice_codec_encoder_destroy(None)
    lib.alice_codec_chunk_destroy(None)
    lib.alice_codec_data_free(None, 0)
    lib.alice_codec_string_free(None)
    assert not lib.alice_codec_encode(None, None, 0, 0, 0, 0)
    n = C.c_uint32(7)
    assert not lib.alice_codec_decode(None, C.byref(n)) and n.value == 7  # *out_len untouched on failure
    assert not lib.alice_codec_chunk_to_bytes(None, C.byref(n))
    assert lib.alice_codec_chunk_width(None) == 0
    assert lib.alice_codec_chunk_height(None) == 0
    assert lib.alice_codec_chunk_frames(None) == 0
    a = (C.c_uint8 * 3)(100, 150, 200)
    assert lib.alice_codec_psnr(None, a, 3) == -1.0


def test_handles_create_destroy(codec):  # src/ffi.rs:345-354
    lib = codec.load_library()
    for ctor in (lib.alice_codec_wavelet1d_haar, lib.alice_codec_wavelet1d_cdf53, lib.alice_codec_wavelet1d_cdf97):
        h = ctor()
        assert h
        lib.alice_codec_wavelet1d_destroy(h)
    e = lib.alice_codec_encoder_create(80)
    assert e and lib.alice_codec_encoder_wavelet(e) == 0 and lib.alice_codec_encoder_quality(e) == 80
    lib.alice_codec_encoder_destroy(e)
    assert not lib.alice_codec_encoder_create_ex(80, 3)


def test_chunk_from_bytes_errors(codec):  # src/ffi.rs:421-429, src/pipeline.rs:772-781
    lib = codec.load_library()
    assert not lib.alice_codec_chunk_from_bytes(None, 0)
    bad = (C.c_uint8 * 3)(*b"BAD")
    assert not lib.alice_codec_chunk_from_bytes(bad, 3)
    assert lib.alice_codec_last_error() == 4
    for data in (b"ALCC", b"BADD" + b"x" * 4000, b"ALCC\x02" + bytes(4000), b"ALCC\x01\x07" + bytes(4000)):
        with pytest.raises(codec.CodecError) as e:
            codec.EncodedChunk.from_bytes(data)
        assert e.value.kind == "InvalidBitstream"


def test_chunk_bytes_roundtrip_on_host(codec, oracle_mod):
    """from_bytes -> to_bytes is the identity on well-formed chunks (host-only code path)."""
    for w, h, f, q, k in ((4, 4, 2, 80, 1), (3, 5, 1, 90, 0), (0, 0, 0, 50, 0)):
        alc = oracle_mod.encode(oracle_mod.make_gradient(w, h, f), w, h, f, q, k)
        c = codec.EncodedChunk.from_bytes(alc)
        assert (c.width, c.height, c.frames, int(c.wavelet_type)) == (w, h, f, k)
        assert c.compressed_size() == len(alc) - 3138
        assert c.to_bytes() == alc
    # trailing bytes beyond the declared payload are dropped (src/pipeline.rs:303)
    alc = oracle_mod.encode(oracle_mod.make_gradient(4, 4, 2), 4, 4, 2, 80)
    assert codec.EncodedChunk.from_bytes(alc + b"junk").to_bytes() == alc
    with pytest.raises(codec.CodecError):
        codec.EncodedChunk.from_bytes(alc[:-1])  # truncated payload (:296-301)


def test_validation_precedes_device_use(codec):
    """Argument errors carry the reference's CodecError variants (src/pipeline.rs:388-427, 786-797)."""
    enc = codec.FrameEncoder(50)
    with pytest.raises(codec.CodecError) as e:
        enc.encode(np.zeros(10, np.uint8), 4, 4, 2)
    assert e.value.kind == "InvalidBufferSize"
    with pytest.raises(codec.CodecError) as e:
        enc.encode(np.zeros(0, np.uint8), 2**32 - 1, 2**32 - 1, 2**32 - 1)
    assert e.value.kind == "DimensionOverflow"
    with pytest.raises(codec.CodecError) as e:
        codec.FastQuantizer(0)
    assert e.value.kind == "InvalidQuantStep"
    with pytest.raises(codec.CodecError) as e:
        codec.FastQuantizer(-5)
    assert e.value.kind == "InvalidQuantStep"
    with pytest.raises(codec.CodecError) as e:
        codec.to_symbols(np.zeros(4, np.int32), out_len=2)
    assert e.value.kind == "InvalidBufferSize"
    # the empty chunk needs no device at all (src/pipeline.rs:391-412, 738-743)
    c = enc.encode(np.zeros(0, np.uint8), 0, 0, 0)
    assert c.compressed_size() == 0 and len(c.to_bytes()) == 3138
    assert codec.FrameDecoder().decode(c).size == 0
    with pytest.raises(codec.CodecError) as e:
        enc.encode(np.zeros(3, np.uint8), 0, 0, 0)
    assert e.value.kind == "InvalidBufferSize"


def test_dimensions_of_2_pow_32_minus_1(codec):
    """0xFFFFFFFF pads to 2^32 in the reference's usize arithmetic (src/pipeline.rs:437-440, 548-551), so the padded
    pixel count can never equal a u32 num_symbols: decode answers InvalidBitstream (:566-571) and nothing may be
    launched for such a shape.  A u32 padding would wrap to 0 and let a crafted header with num_symbols = 0 through."""
    big = 0xFFFFFFFF
    for w, h, f in ((big, 1, 1), (1, big, 1), (1, 1, big), (big, big, 1)):
        hdr = bytearray(3138)
        hdr[0:4] = b"ALCC"; hdr[4] = 1; hdr[5] = 0
        hdr[6:10] = w.to_bytes(4, "little"); hdr[10:14] = h.to_bytes(4, "little"); hdr[14:18] = f.to_bytes(4, "little")
        for c in range(3):
            hdr[18 + 1040 * c + 4: 18 + 1040 * c + 8] = (1).to_bytes(4, "little")    # step
            hdr[18 + 1040 * c + 8: 18 + 1040 * c + 12] = (1).to_bytes(4, "little")   # dead zone
        chunk = codec.EncodedChunk.from_bytes(bytes(hdr))                              # the header itself is well formed
        with pytest.raises(codec.CodecError) as e:
            codec.FrameDecoder().decode(chunk)
        assert e.value.kind == ("InvalidBitstream" if w * h * f < 2**64 else "DimensionOverflow"), (w, h, f)
        with pytest.raises(codec.CodecError) as e:
            codec.Batch(w, h, f, 1, 80)
        assert e.value.kind == "DimensionOverflow"
    lib = codec.load_library()
    fake = C.c_void_p(4096)   # never dereferenced: the shape is refused before any device call
    assert lib.alice_codec_dev_forward_symbols(fake, big, 1, 1, 0, 80, fake, fake, None) == 3
    step = (C.c_int32 * 3)(1, 1, 1)
    assert lib.alice_codec_dev_inverse_symbols(fake, big, 1, 1, 0, step, fake, None) == 3


def test_quality_must_fit_u8(codec):
    """quality is a u8 in the reference (src/pipeline.rs:347, src/bin/main.rs): 300 must not encode as 44."""
    for bad in (256, 300, -1):
        with pytest.raises(ValueError):
            codec.FrameEncoder(bad)
        with pytest.raises(ValueError):
            codec.FrameEncoder.with_wavelet(bad, codec.WaveletType.Cdf97)
    assert codec.FrameEncoder(255).quality == 255
    from alice_codec_amd import cli
    for bad in ("300", "-1"):
        with pytest.raises(SystemExit):
            cli.main(["encode", "in.rgb", "-o", "out.alc", "-W", "4", "-H", "4", "-q", bad])


def test_no_cpu_fallback(codec):
    """Without a HIP device a compute call must fail loudly with DeviceError, never compute on the host."""
    if codec.device_count() > 0:
        pytest.skip("a GPU is present; the failure mode is exercised on CPU-only hosts")
    with pytest.raises(codec.CodecError) as e:
        codec.FrameEncoder(80).encode(np.zeros(96, np.uint8), 4, 4, 2)
    assert e.value.kind == "DeviceError"
    with pytest.raises(codec.CodecError) as e:
        codec.Wavelet3D.cdf53().forward(np.zeros(8, np.int32), 2, 2, 2)
    assert e.value.kind == "DeviceError"
    with pytest.raises(codec.CodecError):
        codec.build_histogram(np.zeros(8, np.uint8))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under alice-codec_amd/ may reference it."""
    pkg = os.path.join(ROOT, "alice-codec_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) == "build":
            continue
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "alice_oracle" not in text and "import oracle" not in text and "from oracle" not in text, fn


def _enc_entry(f, c):
    """Python restatement of make_enc_entry() in alice-codec_amd/csrc/rans.hip (fast path only)."""
    if f == 1:
        return 0xFFFFFFFF, 0, 4095, c + 4095
    L = (f - 1).bit_length()
    rcp = ((1 << (31 + L)) + f - 1) // f
    assert rcp < 2**32
    return rcp, L - 1, 4096 - f, c


def test_encode_reciprocal_is_exact():
    """x' = y + (umulhi(y, rcp) >> rsh) * (4096 - f) + cbias must equal the reference update
    ((y / f) << 12) + y % f + c (src/rans.rs:282-284) for every f in 1..4096 and every
    renormalised state y in [f << 11, f << 19)."""
    rng = np.random.default_rng(3)
    for f in range(1, 4097):
        rcp, rsh, g, cb = _enc_entry(f, 17)
        lo, hi = f << 11, (f << 19) - 1
        ys = np.concatenate([
            np.array([lo, lo + 1, hi, hi - 1, hi - f, hi - f + 1], dtype=np.uint64),
            (rng.integers(lo // f, hi // f, 64).astype(np.uint64) * np.uint64(f)),          # exact multiples
            (rng.integers(lo // f + 1, hi // f, 64).astype(np.uint64) * np.uint64(f)) - np.uint64(1),  # just below
            rng.integers(lo, hi, 64).astype(np.uint64),
        ])
        ys = ys[(ys >= lo) & (ys <= hi)]
        q = ((ys * np.uint64(rcp)) >> np.uint64(32)) >> np.uint64(rsh)
        got = ys + q * np.uint64(g) + np.uint64(cb)
        ref = ((ys // np.uint64(f)) << np.uint64(12)) + ys % np.uint64(f) + np.uint64(17)
        assert np.array_equal(got, ref), f


def test_many_devices_plan_is_round_robin(codec):
    """alice_codec_many_devices_plan: chunk k -> devices[k mod n] (src/pipeline.rs:461-497: chunks are independent), in
    chunk order; no device is touched, so this runs without a GPU."""
    assert codec.plan_devices(7, [0, 1, 2]) == [0, 1, 2, 0, 1, 2, 0]
    assert codec.plan_devices(5, [3]) == [3] * 5
    assert codec.plan_devices(4, [1, 1, 0]) == [1, 1, 0, 1]          # a device may be listed twice (two host threads)
    assert codec.plan_devices(0, [0, 1]) == []
    assert codec.plan_devices(3, list(range(8))) == [0, 1, 2]
    for bad in ([], [-1], [0, -2]):
        with pytest.raises(codec.CodecError) as e:
            codec.plan_devices(4, bad)
        assert e.value.kind == "DeviceError"


def test_many_devices_validate_before_touching_a_gpu(codec):
    """Argument errors of the multi-device calls come back in the reference's order and need no device."""
    lib = codec.load_library()
    enc = lib.alice_codec_encoder_create(80)
    devs = (C.c_int * 2)(0, 1)
    out = (C.c_void_p * 2)()
    buf = (C.c_uint8 * 10)()
    assert lib.alice_codec_encode_many_devices(enc, buf, 10, 4, 4, 2, 2, devs, 2, out) == 1      # InvalidBufferSize
    assert lib.alice_codec_encode_many_devices(enc, buf, 0, 0, 4, 2, 2, devs, 2, out) == 2       # InvalidDimensions
    assert lib.alice_codec_encode_many_devices(None, buf, 10, 4, 4, 2, 2, devs, 2, out) == 9     # null
    assert lib.alice_codec_encode_many_devices(enc, buf, 0, 4, 4, 2, 0, devs, 2, out) == 0       # no chunks: nothing to do
    assert lib.alice_codec_decode_many_devices(None, 1, devs, 2, buf, 10) == 9
    lib.alice_codec_encoder_destroy(enc)


def test_adopted_result_buffers_are_freed_once_with_the_last_view(codec):
    """FrameDecoder.decode hands out the library's own buffer (no 398 MB copy): the array's base owns it and gives it to
    the free function when the last view dies, not before and not twice."""
    import gc
    libc = C.CDLL("libc.so.6")
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    freed = []

    def free(p, n):
        freed.append((p, n))
        libc.free(C.c_void_p(p))
    p = libc.malloc(1000)
    C.memset(p, 7, 1000)
    a = codec._adopt(p, 1000, free)
    assert a.dtype == np.uint8 and a.size == 1000 and int(a.sum()) == 7000 and a.flags.writeable
    view = a[100:200]
    del a
    gc.collect()
    assert freed == [] and int(view.sum()) == 700
    del view
    gc.collect()
    assert freed == [(p, 1000)]
    q = libc.malloc(1)
    assert codec._adopt(q, 0, free).size == 0 and freed[-1] == (q, 0)     # an empty result: released at once
