"""Developer probe on a GPU box: do the tile kernels (instruction-issue bound) and the temporal kernels (memory bound)
overlap when they come from different streams?  T host threads, each with a stream of its own, run the forward (or
inverse) transform stage call on their own 1920x1080x64 chunk; the aggregate time per chunk is compared with T = 1.
Prints one JSON object.

    python scripts/overlap_probe.py [reps] [out.json]"""
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import alice_codec_amd as a  # noqa: E402
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H, F = bench.W, bench.H, bench.F
dev = torch.device("cuda:0")
lib = a.load_library()
px = W * H * F
TMAX = 3
rgbs = [bench.synth_chunk(dev, i).contiguous() for i in range(TMAX)]
syms = [torch.empty(3 * px, dtype=torch.uint8, device=dev) for _ in range(TMAX)]
hists = [torch.zeros(768, dtype=torch.int32, device=dev) for _ in range(TMAX)]
outs = [torch.empty_like(r) for r in rgbs]
streams = [torch.cuda.Stream(device=dev) for _ in range(TMAX)]
step = (C.c_int32 * 3)(14, 14, 14)
res = {"chunk": f"{W}x{H}x{F} CDF 9/7 q=80", "reps_per_thread": reps}


def run(kind, T):
    def work(i):
        a.set_device(0)
        st = streams[i].cuda_stream
        for _ in range(reps):
            if kind == "forward":
                rc = lib.alice_codec_dev_forward_symbols(rgbs[i].data_ptr(), W, H, F, 1, 80, syms[i].data_ptr(), hists[i].data_ptr(), st)
            else:
                rc = lib.alice_codec_dev_inverse_symbols(syms[i].data_ptr(), W, H, F, 1, step, outs[i].data_ptr(), st)
            assert rc == 0
    ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * T)
    return dt


for kind in ("forward", "inverse"):
    run(kind, 1)
    for T in (1, 2, 3):
        dt = run(kind, T)
        res[f"{kind}_ms_per_chunk_T{T}"] = round(dt * 1e3, 4)
        print(f"{kind} T={T}: {dt * 1e3:.3f} ms per chunk aggregate ({6 * px / dt / 8e12:.3f} of 8 TB/s algorithmic)", flush=True)
print(json.dumps(res))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
