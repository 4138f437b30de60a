"""Workload for rocprofv3 runs: one 1920x1080x64 CDF 9/7 q=80 chunk, encode + decode through the batch API,
plus calibration copies of a known byte count at 4 / 8 / 16 bytes per lane (the gfx950 FETCH_SIZE counter
is width dependent, MI355X_MICROARCH.md section HBM).  Usage: rocprofv3 ... -- python scripts/profile_run.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import alice_codec_amd as a
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bench.F = F
dev = torch.device("cuda:0")
rgb = bench.synth_chunk(dev, 0).unsqueeze(0).contiguous()
out = torch.empty_like(rgb)
WT = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bt = a.Batch(bench.W, bench.H, F, 1, 80, a.WaveletType(WT))
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    bt.encode(rgb.data_ptr(), st); sizes = bt.encode_finish()
    bt.decode(bt.alc_ptr(0), bt.alc_stride, out.data_ptr(), st); bt.decode_finish()
print("sizes", sizes, bt.stage_ms())
# calibration: 256 MiB copies (read 256 MiB + write 256 MiB each) with element widths 4, 8, 16 bytes
n = 256 << 20
src = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
for dt in (torch.int32, torch.int64, torch.complex128):
    s = src.view(dt); d = torch.empty_like(s)
    d.copy_(s); torch.cuda.synchronize()
print("calibration done")
