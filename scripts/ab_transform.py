"""Developer A/B workload for rocprofv3 --kernel-trace --stats: N forward and N inverse transform stage calls
(alice_codec_dev_forward_symbols / alice_codec_dev_inverse_symbols) on one 1920x1080x64 chunk, CDF 9/7 q=80, with the
library named by ALICE_CODEC_LIB (default: the in-tree build).  ALICE_AB_PROBE=1: the VALU-floor twins (new builds only).

    ALICE_CODEC_LIB=ab_libs/libalice_r2.so rocprofv3 --kernel-trace --stats ... -- python scripts/ab_transform.py 20"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import alice_codec_amd as a  # noqa: E402
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H, F = bench.W, bench.H, bench.F
dev = torch.device("cuda:0")
lib = C.CDLL(a.LIB_PATH)
vp = C.c_void_p
lib.alice_codec_dev_forward_symbols.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, vp, vp, vp]
lib.alice_codec_dev_inverse_symbols.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint8, C.POINTER(C.c_int32), vp, vp]
px = W * H * F
rgb = bench.synth_chunk(dev, 0).contiguous()
sym = torch.empty(3 * px, dtype=torch.uint8, device=dev)
hist = torch.zeros(768, dtype=torch.int32, device=dev)
out = torch.empty_like(rgb)
st = torch.cuda.current_stream().cuda_stream
step = (C.c_int32 * 3)(14, 14, 14)
if os.environ.get("ALICE_AB_TUNING"):
    lib.alice_codec_test_set_tuning.argtypes = [C.c_long, C.c_long, C.c_long]
    lib.alice_codec_test_set_tuning(*[int(x) for x in os.environ["ALICE_AB_TUNING"].split(",")])
for _ in range(reps):
    assert lib.alice_codec_dev_forward_symbols(rgb.data_ptr(), W, H, F, 1, 80, sym.data_ptr(), hist.data_ptr(), st) == 0
for _ in range(reps):
    assert lib.alice_codec_dev_inverse_symbols(sym.data_ptr(), W, H, F, 1, step, out.data_ptr(), st) == 0
torch.cuda.synchronize()
print("done", a.LIB_PATH)
