"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/alice_oracle.c).

The reference (Rust) cannot be built or run in this image, and it ships no golden
`.alc` files, so these fixtures are the ORACLE's outputs: they pin the HIP path and
guard the oracle itself against regressions; they are not reference outputs
("whole-bitstream parity unpinned", see oracle/alice_oracle.h).  Inputs are either the
reference tests' own gradient generator (src/pipeline.rs:673-683, restated as data in
oracle.make_gradient) or seeded numpy noise.  Each .npz holds: rgb, w, h, f, quality,
wavelet, alc (encoder bytes), decoded (decoder output on those bytes).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as o  # noqa: E402

CASES = [
    # name, w, h, f, quality, wavelet, generator
    ("grad_4x4x2_q80_cdf53", 4, 4, 2, 80, 0, "grad"),
    ("grad_4x4x2_q90_cdf53", 4, 4, 2, 90, 0, "grad"),
    ("grad_4x4x2_q90_cdf97", 4, 4, 2, 90, 1, "grad"),
    ("grad_3x5x1_q90_cdf53", 3, 5, 1, 90, 0, "grad"),
    ("pixel_1x1x1_q100_cdf53", 1, 1, 1, 100, 0, "pixel"),
    ("grad_8x8x2_q100_haar", 8, 8, 2, 100, 2, "grad"),
    ("grad_64x64x8_q100_haar", 64, 64, 8, 100, 2, "grad"),      # BASELINE config 0
    ("solid_4x4x2_q80_cdf53", 4, 4, 2, 80, 0, "solid128"),
    ("noise_33x17x5_q80_cdf97", 33, 17, 5, 80, 1, "noise"),
    ("noise_70x50x6_q75_cdf97", 70, 50, 6, 75, 1, "noise"),
    ("noise_64x48x16_q100_cdf97", 64, 48, 16, 100, 1, "noise"),  # step 1: u8 symbol wrap + table warts
    ("smooth_96x64x16_q80_cdf97", 96, 64, 16, 80, 1, "smooth"),
    ("smooth_96x64x16_q80_cdf53", 96, 64, 16, 80, 0, "smooth"),
]


def make_input(kind, w, h, f, seed):
    n = w * h * f
    if kind == "grad":
        return o.make_gradient(w, h, f)
    if kind == "pixel":
        return np.array([128, 200, 50], np.uint8)
    if kind == "solid128":
        return np.full(n * 3, 128, np.uint8)
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, n * 3, dtype=np.uint8)
    t, y, x = np.meshgrid(np.arange(f), np.arange(h), np.arange(w), indexing="ij")
    out = np.empty((f, h, w, 3), np.float64)
    for c, (s, ph) in enumerate(((23, 0), (31, 1), (17, 2))):
        out[..., c] = 128 + 90 * np.sin((x + 2 * t) / s + ph) * np.cos((y - t) / (0.7 * s))
    out += rng.integers(-4, 5, out.shape)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8).reshape(-1)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for i, (name, w, h, f, q, wt, kind) in enumerate(CASES):
        rgb = make_input(kind, w, h, f, 1000 + i)
        alc = np.frombuffer(o.encode(rgb, w, h, f, q, wt), np.uint8)
        dec = o.decode(alc)
        np.savez_compressed(os.path.join(here, name + ".npz"), rgb=rgb, w=w, h=h, f=f, quality=q, wavelet=wt,
                            alc=alc, decoded=dec)
        print(f"{name}: {alc.size} bytes, psnr {o.psnr(rgb, dec):.2f} dB")


if __name__ == "__main__":
    main()
