"""Developer sweep on a GPU box: milliseconds per 1920x1080x64 chunk of the forward and inverse transform launches alone
(alice_codec_test_transform_ms: HIP events around the launches a batch uses, chains excluded) over the band plan's
target and the probe twins of the kernels (global loads and / or stores replaced by register moves: the VALU floor and
what each kind of access costs on top of it).  Writes one JSON file.

    python scripts/transform_sweep.py out.json [wavelet 0|1|2] [quality] [quick]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import alice_codec_amd as a  # noqa: E402
import bench  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/transform_sweep.json"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
q = int(sys.argv[3]) if len(sys.argv) > 3 else 80
quick = len(sys.argv) > 4
W, H, F = bench.W, bench.H, bench.F
dev = torch.device("cuda:0")
lib = a.load_library()
NB, NCH, REPS = 4, 16, 3
px = W * H * F
rgb = torch.stack([bench.synth_chunk(dev, i).reshape(-1) for i in range(NB)]).contiguous()
sym = torch.empty((NB, 3 * px), dtype=torch.uint8, device=dev)
out = torch.empty_like(rgb)
st = torch.cuda.current_stream().cuda_stream
ms = (C.c_float * 2)()
rows = []


def run(band_kb, probe=0):
    lib.alice_codec_test_set_tuning(band_kb)
    rc = lib.alice_codec_test_transform_ms(rgb.data_ptr(), sym.data_ptr(), out.data_ptr(), NB, W, H, F, k, q, NCH, REPS, probe, ms, st)
    assert rc == 0, rc
    r = {"band_kb": band_kb, "probe": probe, "forward_ms": round(ms[0], 4), "inverse_ms": round(ms[1], 4),
         "forward_frac_of_8TBs": round(6 * px / (ms[0] * 1e-3) / 8e12, 4), "inverse_frac_of_8TBs": round(6 * px / (ms[1] * 1e-3) / 8e12, 4)}
    rows.append(r)
    print(r, flush=True)
    return r


run(0)            # warm-up of every instance
rows.clear()
base = run(0)                            # the uncut chunk: one tile launch and one temporal launch per direction
run(0, probe=1)                          # its VALU floor: loads and stores replaced by register moves
run(0, probe=2)                          # only the loads replaced
run(0, probe=3)                          # only the stores replaced
for band in (() if quick else (32768, 65536, 131072, 262144, 409600)):
    run(band)
lib.alice_codec_test_set_tuning(1024 * 1024)
json.dump({"chunk": f"{W}x{H}x{F}", "wavelet": k, "quality": q, "chunks_per_pass": NCH, "passes": REPS, "distinct_buffers": NB,
           "algorithmic_bytes_per_chunk": 6 * px, "rows": rows, "uncut": base,
           "probe_modes": {"1": "loads and stores replaced by register moves (VALU floor)", "2": "loads replaced", "3": "stores replaced"}},
          open(out_path, "w"), indent=1)
