"""Developer timing on a GPU box: Wavelet3D::forward / inverse of one 1920x1080x64 i32 volume through the device-pointer
stage entry points (tile kernels) and, for comparison, the per-axis kernels the same entry point uses for shapes the
tiles do not cover (forced here with an odd width: 1919).  Prints one JSON object.

    python scripts/stage_wavelet_probe.py [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import alice_codec_amd as a  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = a.load_library()
dev = torch.device("cuda:0")
res = {"volume": "1920x1080x64 i32 (530.8 MB)", "reps": reps, "note": "host-timed around the call, which synchronises its stream"}
for label, (w, h, d) in (("tile_kernels_1920x1080x64", (1920, 1080, 64)), ("per_axis_kernels_1919x1080x64", (1919, 1080, 64))):
    vol = torch.randint(-2000, 2000, (w * h * d,), dtype=torch.int32, device=dev)
    tmp = torch.empty_like(vol)
    for k, kn in ((1, "cdf97"), (0, "cdf53")):
        for name, fn in (("forward", lib.alice_codec_dev_wavelet3d_forward), ("inverse", lib.alice_codec_dev_wavelet3d_inverse)):
            assert fn(k, vol.data_ptr(), tmp.data_ptr(), w, h, d, None) == 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                assert fn(k, vol.data_ptr(), tmp.data_ptr(), w, h, d, None) == 0
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            res.setdefault(label, {})[f"{kn}_{name}_ms"] = round(ms, 3)
            res[label][f"{kn}_{name}_gsamples_per_s"] = round(w * h * d / ms / 1e6, 2)
    del vol, tmp
print(json.dumps(res))
