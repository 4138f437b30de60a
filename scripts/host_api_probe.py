"""Developer timing on a GPU box: the drop-in host-buffer calls (alice_codec_encode64 / alice_codec_decode64) from T host
threads, each on its own 1920x1080x64 chunk in pageable host memory -- bench.host_api_measure, the same measurement the
bench line carries as `host_api` -- plus the bare PCIe copies of one chunk.  WRITES its JSON to the file named on the
command line (and prints it).

    python scripts/host_api_probe.py profiles/r03_host_api_1080p64.json [T ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (no import-time side effects)

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/host_api.json"
threads = tuple(int(x) for x in sys.argv[2:]) or (1, 8, 64)
bench._mods()
import torch  # noqa: E402
import alice_codec_amd as a  # noqa: E402

a.set_device(0)
res = bench.host_api_measure(threads)
W, H, F = bench.W, bench.H, bench.F
if os.environ.get("ALICE_PROBE_MANY"):
    # the many-chunk entry points from ONE thread: n chunks in host memory, one call each way
    import numpy as np
    n = int(os.environ["ALICE_PROBE_MANY"])
    dev = torch.device("cuda:0")
    rgb = np.stack([bench.synth_chunk(dev, 5000 + i).reshape(-1).cpu().numpy() for i in range(n)])
    torch.cuda.empty_cache()
    enc = a.FrameEncoder.with_wavelet(bench.QUALITY, bench.WAVELET)
    t0 = time.perf_counter(); chunks = a.encode_many(enc, rgb, W, H, F); t1 = time.perf_counter()
    out = a.decode_many(chunks); t2 = time.perf_counter()
    px = W * H * F
    res["many_chunk_calls_one_thread"] = {"chunks": n, "encode_s": round(t1 - t0, 3), "decode_s": round(t2 - t1, 3),
                                          "encode_mpix_s": round(n * px / (t1 - t0) / 1e6, 1), "decode_mpix_s": round(n * px / (t2 - t1) / 1e6, 1),
                                          "decoded_shape": list(out.shape)}
    print("[probe] many-chunk calls:", res["many_chunk_calls_one_thread"], file=sys.stderr)
    del rgb, chunks, out
h = bench.synth_chunk(torch.device("cuda:0"), 0).reshape(-1).cpu()
d = torch.empty(h.numel(), dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize(); t3 = time.perf_counter(); d.copy_(h); torch.cuda.synchronize(); t4 = time.perf_counter()
back = d.cpu(); torch.cuda.synchronize(); t5 = time.perf_counter()
res["pcie_one_chunk_pageable"] = {"bytes": int(h.numel()), "h2d_s": round(t4 - t3, 4), "d2h_s": round(t5 - t4, 4)}
with open(out_path, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
