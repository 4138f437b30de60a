"""Command-line front end with the reference CLI's interface (src/bin/main.rs:34-196):
    python -m alice_codec_amd.cli encode INPUT -o OUT.alc -W 1920 -H 1080 -f 64 [-q 90] [-w cdf53|cdf97|haar]
    python -m alice_codec_amd.cli decode INPUT.alc -o OUT.rgb
    python -m alice_codec_amd.cli info INPUT.alc
Like the reference it treats the whole input file as ONE chunk (src/bin/main.rs:117-122).  `encode-chunks`
is the extension SURVEY.md section 8f asks for: it cuts a long raw-RGB file into 64-frame chunks
(DEFAULT_CHUNK_SIZE, src/lib.rs:110) and writes one .alc per chunk."""
from __future__ import annotations

import argparse
import sys

import numpy as np

from . import DEFAULT_CHUNK_SIZE, CodecError, EncodedChunk, FrameDecoder, FrameEncoder, WaveletType, encode_many

WAVELETS = {"cdf53": WaveletType.Cdf53, "cdf97": WaveletType.Cdf97, "haar": WaveletType.Haar}
WAVELET_NAMES = {WaveletType.Cdf53: "CDF 5/3", WaveletType.Cdf97: "CDF 9/7", WaveletType.Haar: "Haar"}


def parse_wavelet(s: str) -> WaveletType:
    if s not in WAVELETS:
        raise ValueError(f"unknown wavelet '{s}'; expected cdf53, cdf97, or haar")
    return WAVELETS[s]


def cmd_encode(a) -> None:
    wt = parse_wavelet(a.wavelet)
    rgb = np.fromfile(a.input, dtype=np.uint8)
    chunk = FrameEncoder.with_wavelet(a.quality, wt).encode(rgb, a.width, a.height, a.frames)
    data = chunk.to_bytes()
    with open(a.output, "wb") as f:
        f.write(data)
    ratio = 0.0 if rgb.size == 0 else len(data) / rgb.size
    print(f"encoded {a.width}x{a.height}x{a.frames} ({rgb.size} bytes) -> {len(data)} bytes "
          f"({ratio * 100:.1f}% ratio, quality={a.quality}, wavelet={a.wavelet})", file=sys.stderr)


def cmd_encode_chunks(a) -> None:
    wt = parse_wavelet(a.wavelet)
    frame_bytes = a.width * a.height * 3
    rgb = np.memmap(a.input, dtype=np.uint8, mode="r")
    if frame_bytes == 0 or rgb.size % frame_bytes:
        raise ValueError("input size is not a whole number of frames")
    n_frames = rgb.size // frame_bytes
    enc = FrameEncoder.with_wavelet(a.quality, wt)
    starts = list(range(0, n_frames, a.chunk))
    k = 0
    i = 0
    while i < len(starts):
        # whole chunks go through one call per group (their entropy chains run side by side); a short tail chunk alone
        group = [s0 for s0 in starts[i:i + a.in_flight] if n_frames - s0 >= a.chunk] or [starts[i]]
        f = min(a.chunk, n_frames - group[0])
        part = np.ascontiguousarray(rgb[group[0] * frame_bytes:(group[-1] + f) * frame_bytes])
        for start, chunk in zip(group, encode_many(enc, part, a.width, a.height, f)):
            data = chunk.to_bytes()
            with open(f"{a.output}.{k:05d}.alc", "wb") as out:
                out.write(data)
            print(f"chunk {k}: frames {start}..{start + f - 1} -> {len(data)} bytes", file=sys.stderr)
            k += 1
        i += len(group)


def cmd_decode(a) -> None:
    data = np.fromfile(a.input, dtype=np.uint8)
    chunk = EncodedChunk.from_bytes(data)
    rgb = FrameDecoder().decode(chunk)
    rgb.tofile(a.output)
    print(f"decoded {chunk.width}x{chunk.height}x{chunk.frames} -> {rgb.size} bytes (raw RGB)", file=sys.stderr)


def cmd_info(a) -> None:
    data = np.fromfile(a.input, dtype=np.uint8)
    chunk = EncodedChunk.from_bytes(data)
    raw = chunk.width * chunk.height * chunk.frames * 3
    ratio = 0.0 if raw == 0 else chunk.compressed_size() / raw
    print("ALICE-Codec Bitstream Info")
    print(f"  File:        {a.input}")
    print(f"  File size:   {data.size} bytes")
    print(f"  Width:       {chunk.width}")
    print(f"  Height:      {chunk.height}")
    print(f"  Frames:      {chunk.frames}")
    print(f"  Wavelet:     {WAVELET_NAMES[chunk.wavelet_type]}")
    print(f"  Payload:     {chunk.compressed_size()} bytes")
    print(f"  Raw size:    {raw} bytes (uncompressed RGB)")
    print(f"  Ratio:       {ratio * 100:.1f}%")


def _u8(text: str) -> int:
    """clap parses `quality: u8` (src/bin/main.rs:44-46): 300 or -1 is a usage error, not quality 44 or 255."""
    v = int(text)
    if not 0 <= v <= 255:
        raise argparse.ArgumentTypeError(f"{text} is not in 0..255")
    return v


def main(argv=None) -> int:
    p = argparse.ArgumentParser(prog="alice-codec", description="ALICE-Codec: 3D wavelet video codec (MI355X path)")
    sub = p.add_subparsers(dest="command", required=True)
    for name in ("encode", "encode-chunks"):
        e = sub.add_parser(name)
        e.add_argument("input")
        e.add_argument("-o", "--output", required=True)
        e.add_argument("-W", "--width", type=int, required=True)
        e.add_argument("-H", "--height", type=int, required=True)
        if name == "encode":
            e.add_argument("-f", "--frames", type=int, default=1)
        else:
            e.add_argument("-c", "--chunk", type=int, default=DEFAULT_CHUNK_SIZE)
            e.add_argument("--in-flight", type=int, default=16, help="chunks encoded per call (GPU memory: about 2.4x the raw size of a chunk each)")
        e.add_argument("-q", "--quality", type=_u8, default=90)
        e.add_argument("-w", "--wavelet", default="cdf53")
    d = sub.add_parser("decode")
    d.add_argument("input")
    d.add_argument("-o", "--output", required=True)
    i = sub.add_parser("info")
    i.add_argument("input")
    a = p.parse_args(argv)
    try:
        {"encode": cmd_encode, "encode-chunks": cmd_encode_chunks, "decode": cmd_decode, "info": cmd_info}[a.command](a)
    except (CodecError, ValueError, OSError) as e:
        print(f"error: {e}", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
