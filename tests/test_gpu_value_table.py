"""The forward temporal kernel quantises through a value -> symbol table in LDS for coefficients in [-r, r) and through the
arithmetic of Quantizer::quantize + to_symbols (reference src/quant.rs:89-97,555-560) for a wavefront that holds any value
outside it (csrc/transform.hip, kQLutR).  With the default r = 2048 no 8-bit RGB content leaves the table (largest
coefficient: 2040), so the suite shrinks r: wavefronts over the loud half of the frames then take the arithmetic, those
over the quiet half the table, and every byte must be the oracle's whatever r is."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def loud_and_quiet(w, h, f, seed):
    """Left half: Co = +-255 checkerboard in x, y and t (the largest coefficients 8-bit RGB can make); right half: a smooth
    ramp with a little noise."""
    rng = np.random.default_rng(seed)
    t, y, x = np.meshgrid(np.arange(f), np.arange(h), np.arange(w), indexing="ij")
    s = ((x + y + t) & 1) == 1
    ramp = (x * 255 // max(w - 1, 1) + rng.integers(-4, 5, x.shape)).clip(0, 255)
    left = x < w // 2
    r = np.where(left, np.where(s, 255, 0), ramp)
    b = np.where(left, np.where(s, 0, 255), 255 - ramp)
    g = np.where(left, 128, ramp)
    return np.stack([r, g, b], axis=-1).astype(np.uint8).reshape(-1)


@pytest.fixture
def radius(gpu_codec):
    lib = gpu_codec.load_library()
    yield lambda r: lib.alice_codec_test_set_value_table_radius(r)
    lib.alice_codec_test_set_value_table_radius(2048)


def test_eight_bit_rgb_stays_inside_the_default_table(oracle_mod):
    """The bound the kernel's comment states, on the oracle: |coefficient| <= 2040 < 2048 for the worst content."""
    w, h, f = 64, 32, 8
    rgb = loud_and_quiet(w, h, f, 0)
    worst = 0
    for k in (0, 1, 2):
        for ch in oracle_mod.rgb_to_ycocg_r(rgb):
            c = oracle_mod.wavelet3d(k, np.asarray(ch, np.int32), w, h, f)
            worst = max(worst, int(np.abs(c).max()))
    assert 1500 < worst < 2048, worst


@pytest.mark.parametrize("k,q", [(1, 80), (1, 100), (0, 80), (2, 95), (1, 30)])
def test_table_and_arithmetic_agree_with_the_oracle(gpu_codec, oracle_mod, radius, k, q):
    for (w, h, f) in ((256, 96, 16), (130, 75, 7), (512, 64, 32)):
        rgb = loud_and_quiet(w, h, f, seed=w + q)
        ref = oracle_mod.encode(rgb, w, h, f, q, k)
        for r in (2048, 700, 64, 3, 1):
            radius(r)
            chunk = gpu_codec.FrameEncoder.with_wavelet(q, gpu_codec.WaveletType(k)).encode(rgb, w, h, f)
            assert hashlib.sha256(chunk.to_bytes()).hexdigest() == hashlib.sha256(ref).hexdigest(), (w, h, f, k, q, r)
        assert np.array_equal(gpu_codec.FrameDecoder().decode(chunk), oracle_mod.decode(ref)), (w, h, f, k, q)
