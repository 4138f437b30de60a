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
