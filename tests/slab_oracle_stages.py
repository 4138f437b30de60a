"""Test-only compute provider for alice_codec_amd.slab: the CPU oracle behind the five stage calls, so the
exchange logic of the row-slab path can be rehearsed with `gloo` ranks on machines without a GPU.
Never imported by the product."""
import numpy as np
import torch

import oracle as o


class OracleStages:
    def forward_symbols(self, rgb, wavelet, quality):
        f, h, w, _ = rgb.shape
        pw, ph, pf = w + (w & 1), h + (h & 1), 2 if f == 1 else f + (f & 1)
        sym = o.encode_symbols(rgb.contiguous().numpy().reshape(-1), w, h, f, quality, wavelet)
        return torch.from_numpy(sym.reshape(3, pf, ph, pw))

    def inverse_symbols(self, sym, w, h, f, wavelet, steps):
        pw, ph, pf = w + (w & 1), h + (h & 1), 2 if f == 1 else f + (f & 1)
        s = sym.contiguous().numpy().reshape(3, -1)
        planes = []
        for c in range(3):
            q = o.from_symbols(s[c])
            v = o.dequantize_buffer(int(steps[c]), q)
            v = o.wavelet3d(wavelet, v, pw, ph, pf, inverse=True).reshape(pf, ph, pw)
            planes.append(v[:f, :h, :w].astype(np.int16).reshape(-1))     # `as i16`, src/pipeline.rs:105-113
        rgb = o.ycocg_r_to_rgb(planes[0], planes[1], planes[2])
        return torch.from_numpy(np.asarray(rgb, dtype=np.uint8).reshape(f, h, w, 3).copy())

    def histogram(self, sym):
        return torch.from_numpy(o.build_histogram(sym.contiguous().numpy().reshape(-1)).astype(np.int64))

    def rans_encode(self, sym, hist):
        t = o.FrequencyTable(np.asarray(hist, dtype=np.uint32))
        return torch.from_numpy(np.frombuffer(o.rans_encode(sym.contiguous().numpy().reshape(-1), t), np.uint8).copy())

    def rans_decode(self, stream, hist, n):
        t = o.FrequencyTable(np.asarray(hist, dtype=np.uint32))
        return torch.from_numpy(np.asarray(o.rans_decode(stream.contiguous().numpy(), n, t), dtype=np.uint8).copy())
