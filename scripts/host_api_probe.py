"""Developer timing on a GPU box: the drop-in host-buffer calls (alice_codec_encode64 / alice_codec_decode64 through the
Python mirror) on one 1920x1080x64 chunk, CDF 9/7 q=80, pageable host memory -- the PCIe-inclusive rate of the path.
Prints one JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import alice_codec_amd as a  # noqa: E402
import bench  # noqa: E402

W, H, F = bench.W, bench.H, bench.F
rgb = bench.synth_chunk(torch.device("cuda:0"), 0).reshape(-1).cpu().numpy()
enc = a.FrameEncoder.with_wavelet(80, a.WaveletType.Cdf97)
dec = a.FrameDecoder()
chunk = enc.encode(rgb, W, H, F)          # warm-up: library load, pool
dec.decode(chunk)
t0 = time.perf_counter(); chunk = enc.encode(rgb, W, H, F); t1 = time.perf_counter()
out = dec.decode(chunk); t2 = time.perf_counter()
d = torch.empty(rgb.size, dtype=torch.uint8, device="cuda:0")
h = torch.from_numpy(rgb)
torch.cuda.synchronize(); t3 = time.perf_counter(); d.copy_(h); torch.cuda.synchronize(); t4 = time.perf_counter()
back = d.cpu(); torch.cuda.synchronize(); t5 = time.perf_counter()
px = W * H * F
print(json.dumps({"chunk": f"{W}x{H}x{F} CDF 9/7 q=80, pageable host buffers",
                  "encode_s": round(t1 - t0, 3), "decode_s": round(t2 - t1, 3),
                  "encode_mpix_s": round(px / (t1 - t0) / 1e6, 2), "decode_mpix_s": round(px / (t2 - t1) / 1e6, 2),
                  "h2d_398MB_pageable_s": round(t4 - t3, 4), "d2h_398MB_pageable_s": round(t5 - t4, 4),
                  "alc_bytes": len(chunk.to_bytes())}))
