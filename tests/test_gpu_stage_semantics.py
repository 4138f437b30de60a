"""Stage-API objects that live across calls, against the oracle's objects doing the same call sequence
(reference src/rans.rs:238-309, 321-389): RansEncoder::encode / repeated encode_symbols continue one state, RansDecoder::decode /
decode_n continue from the current position; FrequencyTable over n != 256 symbols (:102-104,158-189; the reference's own
test_uniform_table_small, :934-944, uses 2); quantize_subband / dequantize_subband (src/quant.rs:518-545)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def skewed(rng, n, n_sym=256):
    p = 1.0 / (1.0 + np.arange(n_sym)) ** 1.3
    return rng.choice(n_sym, size=n, p=p / p.sum()).astype(np.uint8)


def tables(gpu_codec, oracle_mod, hist):
    return gpu_codec.FrequencyTable.from_histogram(hist), oracle_mod.FrequencyTable(hist)


def test_repeated_encode_symbols_continue_the_state(gpu_codec, oracle_mod):
    rng = np.random.default_rng(5)
    for sizes in ((3, 2), (1, 1, 1), (5000, 7), (64, 4096, 1023, 1), (9000, 9000)):
        parts = [skewed(rng, n) for n in sizes]
        hist = np.bincount(np.concatenate(parts), minlength=256).astype(np.uint32)
        tg, to = tables(gpu_codec, oracle_mod, hist)
        eg, eo = gpu_codec.RansEncoder(), oracle_mod.RansEncoder()
        for p in parts:
            eg.encode_symbols(p, tg)
            eo.encode_symbols(p, to)
            assert eg.state == eo.state, sizes
        got, ref = eg.finish(), eo.finish()
        assert got == ref, sizes
        # s1 then s2 is one call on s2 || s1 (src/rans.rs:288-294)
        assert got == oracle_mod.rans_encode(np.concatenate(parts[::-1]), to), sizes


def test_single_symbol_encode_and_mixed_calls(gpu_codec, oracle_mod):
    rng = np.random.default_rng(6)
    sym = skewed(rng, 300)
    hist = np.bincount(sym, minlength=256).astype(np.uint32)
    tg, to = tables(gpu_codec, oracle_mod, hist)
    eg, eo = gpu_codec.RansEncoder.with_capacity(1024), oracle_mod.RansEncoder()
    for s in sym[:40][::-1]:                      # encode(&RansSymbol), last symbol first (src/rans.rs:269-285)
        rs = tg.get_symbol(int(s))
        eg.encode(rs)
        eo.encode(int(to.cum_freq[s]), int(to.freq[s]))
    eg.encode_symbols(sym[40:], tg); eo.encode_symbols(sym[40:], to)
    eg.encode(gpu_codec.RansSymbol(100, 3000)); eo.encode(100, 3000)     # any (cum, freq) pair, table or not
    eg.encode(gpu_codec.RansSymbol(0, 1)); eo.encode(0, 1)
    eg.encode(gpu_codec.RansSymbol(7, 60000)); eo.encode(7, 60000)       # a frequency above 4096 (exact path)
    assert eg.state == eo.state
    assert eg.finish() == eo.finish()
    # an encoder that never saw a symbol: the four bytes of 2^23
    assert gpu_codec.RansEncoder().finish() == oracle_mod.RansEncoder().finish() == bytes([0, 0x80, 0, 0])
    # a zero frequency: the reference divides by zero
    with pytest.raises(gpu_codec.CodecError):
        gpu_codec.RansEncoder().encode(gpu_codec.RansSymbol(5, 0))


def test_decoder_continues_from_its_position(gpu_codec, oracle_mod):
    rng = np.random.default_rng(7)
    sym = skewed(rng, 30000)
    hist = np.bincount(sym, minlength=256).astype(np.uint32)
    tg, to = tables(gpu_codec, oracle_mod, hist)
    data = oracle_mod.rans_encode(sym, to)
    for calls in ((2, 3), (1, 1, 1, 1), (5000, 1, 9000, 4096, 4095), (4096, 4096), (29999, 1, 500)):
        dg, do = gpu_codec.RansDecoder(data), oracle_mod.RansDecoder(data)
        assert dg.state == do.state and dg.position == do.pos
        for n in calls:
            a, b = dg.decode_n(n, tg), do.decode_n(n, to)
            assert np.array_equal(a, b), (calls, n)
            assert dg.state == do.state and dg.position == do.pos and dg.is_empty() == do.is_empty(), (calls, n)
    dg, do = gpu_codec.RansDecoder(data), oracle_mod.RansDecoder(data)
    assert [dg.decode(tg) for _ in range(5)] == [int(do.decode_n(1, to)[0]) for _ in range(5)]
    # short and empty inputs (no state bytes: state 0, position 0; is_empty at once)
    for blob in (b"", b"\x01\x02\x03", b"\x00\x80\x00\x00"):
        dg, do = gpu_codec.RansDecoder(blob), oracle_mod.RansDecoder(blob)
        assert dg.is_empty() == do.is_empty() and dg.state == do.state and dg.position == do.pos
        assert np.array_equal(dg.decode_n(10, tg), do.decode_n(10, to))
        assert dg.is_empty() == do.is_empty() and dg.state == do.state and dg.position == do.pos


@pytest.mark.parametrize("n", [1, 2, 3, 7, 100, 255, 256])
def test_frequency_tables_of_n_symbols(gpu_codec, oracle_mod, n):
    rng = np.random.default_rng(n)
    cases = [np.zeros(n, np.uint32), rng.integers(0, 50, n).astype(np.uint32), rng.integers(0, 2, n).astype(np.uint32) * 1000,
             np.full(n, 4_000_000_000 // max(n, 1), np.uint32)]
    for hist in cases:
        tg, to = tables(gpu_codec, oracle_mod, hist)
        assert len(tg) == len(to) == n
        assert np.array_equal(tg.freq[:n], to.freq) and np.array_equal(tg.cum_freq[:n], to.cum_freq), (n, hist[:8])
        assert not tg.freq[n:].any() and not tg.cum_freq[n:].any()
    u = gpu_codec.FrequencyTable.uniform(n)
    uo = oracle_mod.FrequencyTable(uniform=n)
    assert np.array_equal(u.freq[:n], uo.freq) and np.array_equal(u.cum_freq[:n], uo.cum_freq)
    assert int(u.freq[:n].astype(np.int64).sum()) == 4096                                  # src/rans.rs:934-944
    # round trip through the coders with this alphabet
    sym = rng.integers(0, n, 5000).astype(np.uint8)
    enc = gpu_codec.RansEncoder(); enc.encode_symbols(sym, u)
    data = enc.finish()
    assert data == oracle_mod.rans_encode(sym, uo)
    assert np.array_equal(gpu_codec.RansDecoder(data).decode_n(5000, u), oracle_mod.rans_decode(data, 5000, uo))
    if n < 256:   # a symbol the table does not hold: the reference indexes out of bounds
        with pytest.raises(gpu_codec.CodecError):
            e = gpu_codec.RansEncoder(); e.encode_symbols(np.array([n], np.uint8), u)
        with pytest.raises(IndexError):
            u.get_symbol(n)


def test_empty_and_oversized_alphabets_are_errors(gpu_codec):
    with pytest.raises(gpu_codec.CodecError):
        gpu_codec.FrequencyTable.from_histogram(np.zeros(0, np.uint32))      # uniform(0): the reference divides by zero
    with pytest.raises(gpu_codec.CodecError):
        gpu_codec.FrequencyTable.from_histogram(np.ones(257, np.uint32))     # symbols are u8


def test_subband_quantisers(gpu_codec, oracle_mod):
    rng = np.random.default_rng(8)
    v = rng.integers(-5000, 5000, 10001).astype(np.int32)
    v[:4] = [0, 2**31 - 1, -2**31, -1]
    for step, dz in ((8, 8), (14, 21), (1, 1), (3, 0), (-5, 4)):
        q = gpu_codec.Quantizer.with_dead_zone(step, dz)
        got = gpu_codec.quantize_subband(v, q)
        assert np.array_equal(got, oracle_mod.quantize_buffer(step, v, dz)), (step, dz)
        assert np.array_equal(gpu_codec.dequantize_subband(got, q), oracle_mod.dequantize_buffer(step, got)), (step, dz)
    with pytest.raises(gpu_codec.CodecError):                                 # output smaller than the input (src/quant.rs:516)
        gpu_codec.quantize_subband(v, gpu_codec.Quantizer(8), out_len=10)


def test_many_devices_entry_points_match_the_single_device_calls(gpu_codec, oracle_mod):
    """alice_codec_encode_many_devices / decode_many_devices (chunk k -> devices[k mod n], one host thread per entry):
    with one device listed once, and listed three times (three host threads sharing the card -- the only multi-thread
    arrangement a one-GPU box offers), the chunks are byte-identical to alice_codec_encode_many and to the oracle, in
    chunk order."""
    w, h, f, n = 96, 64, 8, 7
    rng = np.random.default_rng(11)
    chunks = rng.integers(0, 256, (n, w * h * f * 3), dtype=np.uint8)
    enc = gpu_codec.FrameEncoder.with_wavelet(80, gpu_codec.WaveletType.Cdf97)
    base = [c.to_bytes() for c in gpu_codec.encode_many(enc, chunks, w, h, f)]
    for i in range(n):
        assert base[i] == oracle_mod.encode(chunks[i], w, h, f, 80, 1), i
    for devs in ([0], [0, 0, 0]):
        got = gpu_codec.encode_many(enc, chunks, w, h, f, devices=devs)
        assert [c.to_bytes() for c in got] == base, devs
        dec = gpu_codec.decode_many(got, devices=devs)
        assert np.array_equal(dec, gpu_codec.decode_many(got)), devs
        for i in (0, n - 1):
            assert np.array_equal(dec[i], oracle_mod.decode(base[i])), (devs, i)
    with pytest.raises(gpu_codec.CodecError):      # a device the node does not have
        gpu_codec.encode_many(enc, chunks, w, h, f, devices=[0, 99])
