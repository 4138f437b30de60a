"""Band-wise transform launches (csrc/transform.hip, "Bands") on small shapes: the suite shrinks the band target
(alice_codec_test_set_tuning) so that frames of a few hundred rows are cut into many bands -- every band boundary,
the inverse bands' halo rows, bands that end in an overhanging tile row, the pixel <-> symbol-plane segment mapping -- and
compares every byte with the oracle.  Results must not depend on the band plan at all."""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def smooth(w, h, f, seed):
    rng = np.random.default_rng(seed)
    t, y, x = np.meshgrid(np.arange(f), np.arange(h), np.arange(w), indexing="ij")
    base = 128 + 70 * np.sin((x + 2 * t) / 9.0 + seed) * np.cos((y - t) / 7.0)
    rgb = np.stack([base, base * 0.8 + 20, 255 - base], axis=-1) + rng.integers(-6, 7, (f, h, w, 3))
    return np.clip(rgb, 0, 255).astype(np.uint8).reshape(-1)


@pytest.fixture
def tuning(gpu_codec):
    lib = gpu_codec.load_library()
    yield lambda band_kb: lib.alice_codec_test_set_tuning(band_kb)
    lib.alice_codec_test_set_tuning(1024 * 1024)


# (w, h, f): padded widths are multiples of 4 (only those are cut); heights with a partial last tile row for both tile
# heights (40 forward, 32 inverse), an odd height, an odd width (pad column), odd and single frame counts
SHAPES = [(256, 250, 10), (255, 251, 5), (128, 321, 4), (260, 96, 1), (512, 200, 6)]
# band target in KiB: one tile row per band, two or three, and (0) the uncut chunk
TUNINGS = [96, 200, 400, 0]


@pytest.mark.parametrize("k,q", [(1, 80), (1, 90), (0, 80), (2, 100), (1, 100)])
def test_banded_chunks_match_the_oracle(gpu_codec, oracle_mod, tuning, k, q):
    for (w, h, f) in SHAPES:
        rgb = smooth(w, h, f, seed=w + h + f + k)
        ref = oracle_mod.encode(rgb, w, h, f, q, k)
        want = oracle_mod.decode(ref)
        for tn in TUNINGS:
            tuning(tn)
            chunk = gpu_codec.FrameEncoder.with_wavelet(q, gpu_codec.WaveletType(k)).encode(rgb, w, h, f)
            got = chunk.to_bytes()
            assert hashlib.sha256(got).hexdigest() == hashlib.sha256(ref).hexdigest(), (w, h, f, k, q, tn)
            assert np.array_equal(gpu_codec.FrameDecoder().decode(chunk), want), (w, h, f, k, q, tn)


def test_banded_decode_of_foreign_steps(gpu_codec, oracle_mod, tuning):
    """Headers whose quantiser steps force the i32 band slots (no i16 bound) and the exact-product instances."""
    w, h, f = 256, 250, 6
    rgb = smooth(w, h, f, seed=9)
    blob = bytearray(oracle_mod.encode(rgb, w, h, f, 80, 1))
    for steps in ((300, 14, 14), (70000, 3, 1 << 20)):
        for c, s in enumerate(steps):
            blob[18 + 1040 * c + 4:18 + 1040 * c + 8] = int(s).to_bytes(4, "little")
        want = oracle_mod.decode(bytes(blob))
        for tn in TUNINGS[:2]:
            tuning(tn)
            got = gpu_codec.FrameDecoder().decode(gpu_codec.EncodedChunk.from_bytes(bytes(blob)))
            assert np.array_equal(got, want), (steps, tn)


def test_banded_batch_pipelines_across_chunks(gpu_codec, oracle_mod, tuning):
    """A batch of banded chunks: in-place decode puts the pixels of chunk i on the symbols of chunk i - 1 (chunk 0 in a
    spare buffer) when the chunks are cut, on the chunk's own symbols when they are not."""
    import torch
    w, h, f, B = 256, 250, 8, 4
    chunks = [smooth(w, h, f, seed=40 + i) for i in range(B)]
    refs = [oracle_mod.encode(c, w, h, f, 80, 1) for c in chunks]
    rgb = torch.from_numpy(np.stack(chunks)).cuda()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    for tn in (96, 200, 0):
        tuning(tn)
        bt = gpu_codec.Batch(w, h, f, B, 80, gpu_codec.WaveletType.Cdf97)
        st = torch.cuda.current_stream().cuda_stream
        for rep in range(2):
            bt.encode(rgb.data_ptr(), st)
            sizes = bt.encode_finish()
            for i in range(B):
                t = torch.empty(int(sizes[i]), dtype=torch.uint8, device="cuda")
                assert hip.hipMemcpy(t.data_ptr(), bt.alc_ptr(i), int(sizes[i]), 3) == 0
                assert bytes(t.cpu().numpy()) == refs[i], (tn, rep, i)
            bt.decode(bt.alc_ptr(0), bt.alc_stride, None, st)
            bt.decode_finish()
            for i in range(B):
                t = torch.empty(w * h * f * 3, dtype=torch.uint8, device="cuda")
                assert hip.hipMemcpy(t.data_ptr(), bt.rgb_ptr(i), t.numel(), 3) == 0
                assert np.array_equal(t.cpu().numpy(), oracle_mod.decode(refs[i])), (tn, rep, i)
        del bt


def test_transform_timer_and_valu_probe_run(gpu_codec):
    """alice_codec_test_transform_ms: the timer the profiles use (real kernels and the VALU-floor twins) returns sane numbers."""
    import torch
    lib = gpu_codec.load_library()
    w, h, f = 512, 256, 16
    px = w * h * f
    rgb = torch.randint(0, 256, (2, px * 3), dtype=torch.uint8, device="cuda")
    sym = torch.empty((2, px * 3), dtype=torch.uint8, device="cuda")
    out = torch.empty_like(rgb)
    ms = (C.c_float * 2)()
    for probe in (0, 1, 2, 3):
        rc = lib.alice_codec_test_transform_ms(rgb.data_ptr(), sym.data_ptr(), out.data_ptr(), 2, w, h, f, 1, 80, 4, 2, probe, ms,
                                               torch.cuda.current_stream().cuda_stream)
        assert rc == 0, gpu_codec.last_error_message() if hasattr(gpu_codec, "last_error_message") else rc
        assert 0 < ms[0] < 50 and 0 < ms[1] < 50


def test_chain_kernels_keep_one_wave_per_simd(gpu_codec):
    """The one-chain-per-SIMD placement rests on the compiler counting the kernels' AGPR clobber: ask the runtime."""
    out = (C.c_uint32 * 6)()
    assert gpu_codec.load_library().alice_codec_test_chain_occupancy(out) == 0
    enc_regs, enc_lds, enc_wg, dec_regs, dec_lds, dec_wg = list(out)
    assert enc_regs > 256 and dec_regs > 256, list(out)      # more than half of a SIMD's 512 registers per lane
    assert 1 <= enc_wg <= 4 and 1 <= dec_wg <= 4, list(out)  # at most one single-wave workgroup per SIMD


def test_batch_reuses_capacities_and_recovers_when_content_grows(gpu_codec, oracle_mod):
    """From its second encode on a batch keeps the .alc capacities of the encode before and queues the whole step without
    a host round trip; content that needs more (noise after smooth frames) overflows on the device, and encode_finish
    re-sizes from the new histograms and runs again.  Either way the bytes are the oracle's."""
    import torch
    w, h, f, B = 128, 96, 8, 3
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rng = np.random.default_rng(3)
    quiet = [smooth(w, h, f, seed=70 + i) for i in range(B)]
    loud = [rng.integers(0, 256, w * h * f * 3, dtype=np.uint8) for _ in range(B)]
    bt = gpu_codec.Batch(w, h, f, B, 80, gpu_codec.WaveletType.Cdf97)
    st = torch.cuda.current_stream().cuda_stream
    strides = []
    for chunks in (quiet, quiet, loud, quiet, loud):
        rgb = torch.from_numpy(np.stack(chunks)).cuda()
        bt.encode(rgb.data_ptr(), st)
        sizes = bt.encode_finish()
        strides.append(bt.alc_stride)
        for i in range(B):
            t = torch.empty(int(sizes[i]), dtype=torch.uint8, device="cuda")
            assert hip.hipMemcpy(t.data_ptr(), bt.alc_ptr(i), int(sizes[i]), 3) == 0
            assert bytes(t.cpu().numpy()) == oracle_mod.encode(chunks[i], w, h, f, 80, 1), (len(strides), i)
    assert strides[0] == strides[1] < strides[2] == strides[3] == strides[4]    # grown once, never shrunk
