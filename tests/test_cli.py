"""CLI parity with the reference tool (src/bin/main.rs:34-196): same sub-commands, options and `info` report."""
import os

import numpy as np
import pytest


def test_info_matches_reference_report(codec, oracle_mod, tmp_path, capsys):
    from alice_codec_amd import cli
    p = tmp_path / "a.alc"
    p.write_bytes(oracle_mod.encode(oracle_mod.make_gradient(4, 4, 2), 4, 4, 2, 90, 1))
    assert cli.main(["info", str(p)]) == 0
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "ALICE-Codec Bitstream Info"
    assert "  File size:   3180 bytes" in out and "  Wavelet:     CDF 9/7" in out
    assert "  Payload:     42 bytes" in out and "  Raw size:    96 bytes (uncompressed RGB)" in out
    assert "  Ratio:       43.8%" in out


def test_cli_errors(codec, tmp_path, capsys):
    from alice_codec_amd import cli
    bad = tmp_path / "bad.alc"
    bad.write_bytes(b"ALCC")
    assert cli.main(["info", str(bad)]) == 1
    assert cli.main(["decode", str(tmp_path / "missing.alc"), "-o", str(tmp_path / "x")]) == 1
    raw = tmp_path / "in.rgb"
    raw.write_bytes(bytes(96))
    assert cli.main(["encode", str(raw), "-o", str(tmp_path / "o.alc"), "-W", "4", "-H", "4", "-f", "2", "-w", "dct"]) == 1
    assert "unknown wavelet" in capsys.readouterr().err


@pytest.mark.gpu
def test_cli_encode_decode_files(gpu_codec, oracle_mod, tmp_path):
    from alice_codec_amd import cli
    w, h, f = 48, 32, 70
    rgb = np.random.default_rng(3).integers(0, 256, w * h * f * 3, dtype=np.uint8)
    raw = tmp_path / "in.rgb"; rgb.tofile(raw)
    alc = tmp_path / "out.alc"
    assert cli.main(["encode", str(raw), "-o", str(alc), "-W", str(w), "-H", str(h), "-f", str(f), "-q", "80", "-w", "cdf97"]) == 0
    ref = oracle_mod.encode(rgb, w, h, f, 80, 1)
    assert alc.read_bytes() == ref
    dec = tmp_path / "dec.rgb"
    assert cli.main(["decode", str(alc), "-o", str(dec)]) == 0
    assert np.array_equal(np.fromfile(dec, np.uint8), oracle_mod.decode(ref))
    # multi-chunk driver: 64-frame chunks + a 6-frame tail, each an independent .alc
    assert cli.main(["encode-chunks", str(raw), "-o", str(tmp_path / "c"), "-W", str(w), "-H", str(h), "-q", "80", "-w", "cdf97"]) == 0
    fb = w * h * 3
    assert (tmp_path / "c.00000.alc").read_bytes() == oracle_mod.encode(rgb[:64 * fb], w, h, 64, 80, 1)
    assert (tmp_path / "c.00001.alc").read_bytes() == oracle_mod.encode(rgb[64 * fb:], w, h, 6, 80, 1)


@pytest.mark.gpu
def test_chunk_driver_groups_and_many_api(gpu_codec, oracle_mod, tmp_path):
    """encode-chunks pushes whole chunks through alice_codec_encode_many in groups; every .alc equals the
    single-chunk encode, the tail chunk included; decode_many returns what FrameDecoder::decode returns."""
    from alice_codec_amd import cli
    w, h, c, n_frames = 40, 24, 8, 8 * 5 + 3          # five whole 8-frame chunks and a 3-frame tail
    rgb = np.random.default_rng(9).integers(0, 256, w * h * n_frames * 3, dtype=np.uint8)
    raw = tmp_path / "in.rgb"; rgb.tofile(raw)
    assert cli.main(["encode-chunks", str(raw), "-o", str(tmp_path / "g"), "-W", str(w), "-H", str(h), "-c", str(c),
                     "--in-flight", "2", "-q", "85", "-w", "cdf97"]) == 0
    fb = w * h * 3
    for k in range(6):
        f = c if k < 5 else 3
        part = rgb[k * c * fb:(k * c + f) * fb]
        assert (tmp_path / f"g.{k:05d}.alc").read_bytes() == oracle_mod.encode(part, w, h, f, 85, 1), k
    enc = gpu_codec.FrameEncoder.with_wavelet(85, gpu_codec.WaveletType.Cdf97)
    chunks = gpu_codec.encode_many(enc, rgb[:5 * c * fb], w, h, c)
    assert [ch.to_bytes() for ch in chunks] == [oracle_mod.encode(rgb[k * c * fb:(k + 1) * c * fb], w, h, c, 85, 1) for k in range(5)]
    dec = gpu_codec.decode_many(chunks)
    for k in range(5):
        assert np.array_equal(dec[k], oracle_mod.decode(chunks[k].to_bytes())), k
    with pytest.raises(gpu_codec.CodecError):
        gpu_codec.encode_many(enc, rgb[:5 * c * fb + 3], w, h, c)
