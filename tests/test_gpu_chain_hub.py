"""The chain hub (csrc/codec.hip, ChainHub): whole-chunk host calls of concurrent threads hand their rANS chains to merged
launches.  Whatever the threads do at the same time -- more callers than lane streams, chunks of different shapes and
wavelets, encodes beside decodes beside many-chunk calls, a call that fails its validation while the others run, a call
whose first stream capacity overflows and is retried -- every result must be the oracle's bytes, as with one call at a
time (FrameEncoder / EncodedChunk are Send + Sync in the reference, src/pipeline.rs:635-644)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def smooth(w, h, f, seed):
    rng = np.random.default_rng(seed)
    t, y, x = np.meshgrid(np.arange(f), np.arange(h), np.arange(w), indexing="ij")
    base = 128 + 80 * np.sin((x + 2 * t) / 9.0 + seed) * np.cos((y - t) / 7.0)
    rgb = np.stack([base, base * 0.8 + 20, 255 - base], axis=-1) + rng.integers(-6, 7, (f, h, w, 3))
    return np.clip(rgb, 0, 255).astype(np.uint8).reshape(-1)


def run_threads(fns):
    errs = []

    def guard(fn):
        try:
            fn()
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=guard, args=(fn,)) for fn in fns]
    [t.start() for t in th]
    [t.join() for t in th]
    return errs


def test_more_callers_than_lanes_mixed_shapes(gpu_codec, oracle_mod):
    shapes = [(64, 32, 8, 1, 80), (130, 75, 7, 0, 90), (256, 96, 16, 1, 30), (32, 32, 2, 2, 100), (200, 64, 9, 1, 80), (96, 41, 5, 0, 75)]
    n = 18
    jobs = [(shapes[i % len(shapes)], smooth(*shapes[i % len(shapes)][:3], seed=300 + i)) for i in range(n)]
    refs = [oracle_mod.encode(rgb, w, h, f, q, k) for ((w, h, f, k, q), rgb) in jobs]
    out = [None] * n

    def work(i):
        (w, h, f, k, q), rgb = jobs[i]
        for rep in range(3):
            c = gpu_codec.FrameEncoder.with_wavelet(q, gpu_codec.WaveletType(k)).encode(rgb, w, h, f)
            out[i] = (c.to_bytes(), np.array(gpu_codec.FrameDecoder().decode(c)))
            assert out[i][0] == refs[i], (i, rep)
    assert run_threads([lambda i=i: work(i) for i in range(n)]) == []
    for i in range(n):
        assert out[i][0] == refs[i], i
        assert np.array_equal(out[i][1], oracle_mod.decode(refs[i])), i


def test_encodes_beside_decodes_beside_many_chunk_calls_and_failures(gpu_codec, oracle_mod):
    w, h, f = 128, 64, 8
    enc = gpu_codec.FrameEncoder.with_wavelet(80, gpu_codec.WaveletType.Cdf97)
    rgbs = [smooth(w, h, f, seed=500 + i) for i in range(6)]
    refs = [oracle_mod.encode(r, w, h, f, 80, 1) for r in rgbs]
    wants = [oracle_mod.decode(r) for r in refs]
    chunks = [gpu_codec.EncodedChunk.from_bytes(r) for r in refs]
    bad_blob = bytearray(refs[0]); bad_blob[18 + 12:18 + 16] = (123).to_bytes(4, "little")   # num_symbols != padded pixels
    results = {}

    def encoder(i):
        for _ in range(4):
            assert enc.encode(rgbs[i], w, h, f).to_bytes() == refs[i]
        results[("e", i)] = True

    def decoder(i):
        for _ in range(4):
            assert np.array_equal(gpu_codec.FrameDecoder().decode(chunks[i]), wants[i])
        results[("d", i)] = True

    def many():
        got = gpu_codec.encode_many(enc, np.stack(rgbs), w, h, f)
        assert [c.to_bytes() for c in got] == refs
        assert np.array_equal(gpu_codec.decode_many(got), np.stack(wants))
        results["many"] = True

    def failing():
        for _ in range(6):
            with pytest.raises(gpu_codec.CodecError):
                gpu_codec.FrameDecoder().decode(gpu_codec.EncodedChunk.from_bytes(bytes(bad_blob)))
            with pytest.raises(gpu_codec.CodecError):
                enc.encode(rgbs[0][:-3], w, h, f)            # buffer size mismatch (src/pipeline.rs:422-427)
        results["fail"] = True

    fns = [lambda i=i: encoder(i) for i in range(6)] + [lambda i=i: decoder(i) for i in range(6)] + [many, many, failing]
    assert run_threads(fns) == []
    assert len(results) == 6 + 6 + 2


def test_overflow_retry_inside_a_merged_launch(gpu_codec, oracle_mod):
    """alice_codec_test_force_first_cap is thread-local: two of the six concurrent encoders start with stream regions that
    are far too small, overflow inside the merged launch, and come back through the hub with the next capacity."""
    w, h, f = 96, 64, 8
    lib = gpu_codec.load_library()
    rng = np.random.default_rng(9)
    rgbs = [rng.integers(0, 256, w * h * f * 3, dtype=np.uint8) for _ in range(6)]
    refs = [oracle_mod.encode(r, w, h, f, 90, 1) for r in rgbs]

    def work(i):
        if i < 2:
            lib.alice_codec_test_force_first_cap(4352)
        try:
            for _ in range(3):
                assert gpu_codec.FrameEncoder.with_wavelet(90, gpu_codec.WaveletType.Cdf97).encode(rgbs[i], w, h, f).to_bytes() == refs[i]
        finally:
            lib.alice_codec_test_force_first_cap(0)
    assert run_threads([lambda i=i: work(i) for i in range(6)]) == []


def test_admission_queues_calls_instead_of_failing_them(gpu_codec, oracle_mod):
    """More concurrent calls than the admission budget holds wait their turn (alice_codec_test_set_admission_budget: a budget
    of one byte admits one call at a time, a budget of three calls' worth three); every call still succeeds with the oracle's
    bytes.  0 restores the measured budget."""
    w, h, f = 96, 64, 8
    import ctypes as C
    lib = gpu_codec.load_library()
    lib.alice_codec_test_set_admission_budget.argtypes = [C.c_uint64]
    rgbs = [smooth(w, h, f, seed=700 + i) for i in range(10)]
    refs = [oracle_mod.encode(r, w, h, f, 80, 1) for r in rgbs]
    per_call = w * h * f * 3 * 4
    try:
        for budget in (1, 3 * per_call):
            assert lib.alice_codec_test_set_admission_budget(budget) == 0
            out = [None] * 10

            def work(i):
                c = gpu_codec.FrameEncoder.with_wavelet(80, gpu_codec.WaveletType.Cdf97).encode(rgbs[i], w, h, f)
                out[i] = (c.to_bytes(), np.array(gpu_codec.FrameDecoder().decode(c)))
            assert run_threads([lambda i=i: work(i) for i in range(10)]) == []
            for i in range(10):
                assert out[i][0] == refs[i] and np.array_equal(out[i][1], oracle_mod.decode(refs[i])), (budget, i)
    finally:
        lib.alice_codec_test_set_admission_budget(0)
