"""The streamed gather of bench.py's N > 1 path (alice-codec_amd/multi.py: stream_alc_to_root) on the GPU, as far as a
one-GPU box allows: a single `nccl` (= RCCL) rank on cuda:0 feeds the root logic from DeviceView views of a real batch's
library-owned `.alc` buffers on a NON-default stream, and a sink that copies every chunk on to a ring of pinned host
buffers (as bench.py's does) checks content and order.  A multi-rank RCCL run needs a multi-GPU node (two ranks cannot
share one device); the exchange itself is covered by the world-2/3 `gloo` tests in tests/test_distributed_cpu.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import alice_codec_amd as ac, oracle as o
from alice_codec_amd import multi
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
torch.cuda.set_device(0); ac.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
w, h, f, B = 96, 64, 8, 5
rng = np.random.default_rng(21)
chunks = [rng.integers(0, 256, w * h * f * 3, dtype=np.uint8) for _ in range(B)]
rgb = torch.from_numpy(np.stack(chunks)).to(dev)
bt = ac.Batch(w, h, f, B, 80, ac.WaveletType.Cdf97)
main = torch.cuda.current_stream().cuda_stream
bt.encode(rgb.data_ptr(), main)
sizes = bt.encode_finish()
side = torch.cuda.Stream(device=dev)
ring = [torch.empty(int(sizes.max()), dtype=torch.uint8).pin_memory() for _ in range(2)]
seen, got = [], []
def sink(r, i, t):
    assert t.is_cuda and torch.cuda.current_stream(dev) == side      # device bytes, on the caller's side stream
    slot = ring[len(seen) %% 2]
    slot[:t.numel()].copy_(t, non_blocking=True)
    side.synchronize()                                               # (the test reads the slot at once; bench.py does not)
    seen.append((r, i)); got.append(bytes(slot[:t.numel()].numpy()))
with torch.cuda.stream(side):
    stride = bt.alc_stride
    res = multi.stream_alc_to_root(lambda i: multi.DeviceView(bt.alc_ptr(i), stride).tensor(dev),
                                   torch.from_numpy(sizes.astype(np.int64)), sink, dst=0)
side.synchronize()
assert seen == [(0, i) for i in range(B)], seen                      # chunk-major delivery, every chunk once
assert res.shape == (1, B) and res[0].tolist() == [int(s) for s in sizes]
for i in range(B):
    assert got[i] == o.encode(chunks[i], w, h, f, 80, 1), i
dist.barrier(); dist.destroy_process_group()
print("GPU STREAM GATHER OK")
'''


@pytest.mark.gpu
def test_streamed_gather_root_logic_on_the_gpu(gpu_codec):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "GPU STREAM GATHER OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
