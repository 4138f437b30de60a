"""Developer timing on a GPU box: the forward and inverse transform stage calls alone (no chains), one 1920x1080x64
chunk, CDF 9/7 q=80 unless told otherwise.  Not the bench.

    python scripts/transform_probe.py [wavelet 0|1|2] [reps]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import alice_codec_amd as a  # noqa: E402
import bench  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H, F = bench.W, bench.H, bench.F
dev = torch.device("cuda:0")
rgb = bench.synth_chunk(dev, 0).contiguous()
px = W * H * F
sym = torch.empty(3 * px, dtype=torch.uint8, device=dev)
hist = torch.zeros(768, dtype=torch.int32, device=dev)
out = torch.empty_like(rgb)
lib = a.load_library()
st = torch.cuda.current_stream().cuda_stream
step = (C.c_int32 * 3)(14, 14, 14)
for name, fn in (("forward", lambda: lib.alice_codec_dev_forward_symbols(rgb.data_ptr(), W, H, F, k, 80, sym.data_ptr(), hist.data_ptr(), st)),
                 ("inverse", lambda: lib.alice_codec_dev_inverse_symbols(sym.data_ptr(), W, H, F, k, step, out.data_ptr(), st))):
    assert fn() == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        assert fn() == 0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt * 1e3:.3f} ms per chunk (host-timed, includes one stream sync and scratch allocation from the pool), "
          f"{6 * px / dt / 1e9:.0f} GB/s algorithmic = {6 * px / dt / 8e12:.3f} of 8 TB/s", flush=True)
