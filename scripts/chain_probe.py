"""Developer probe on a GPU box: per-chain rANS encode/decode rate against the number of chains in flight.

    python scripts/chain_probe.py W H F chunks [chunks ...]

One batch per chunk count (3 chains per chunk); prints ns/symbol of the encode and decode chain kernels (HIP-event
stage times of the batch) and, from the encoder's own s_memtime / s_memrealtime stamps (ALICE_CODEC_DEBUG), the shader
cycles per symbol and the clock the chains ran at.  Not the bench; see bench.py."""
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ALICE_CODEC_DEBUG"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402
import alice_codec_amd as a  # noqa: E402

W, H, F = (int(v) for v in sys.argv[1:4])
counts = [int(v) for v in sys.argv[4:]]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1234)
t = torch.arange(F, device=dev).view(F, 1, 1, 1).float()
y = torch.arange(H, device=dev).view(1, H, 1, 1).float()
x = torch.arange(W, device=dev).view(1, 1, W, 1).float()
s = torch.tensor([23., 31., 17.], device=dev).view(1, 1, 1, 3)
ph = torch.tensor([0., 1., 2.], device=dev).view(1, 1, 1, 3)
base = 128 + 90 * torch.sin((x + 2 * t) / s + ph) * torch.cos((y - t) / (0.7 * s))
one = (base + torch.randint(-4, 5, (F, H, W, 3), device=dev, generator=g)).clamp(0, 255).to(torch.uint8)
del base
st = torch.cuda.current_stream().cuda_stream
px = W * H * F
for B in counts:
    rgb = one.unsqueeze(0).expand(B, F, H, W, 3).contiguous()
    bt = a.Batch(W, H, F, B, 80, a.WaveletType.Cdf97)
    log = tempfile.TemporaryFile(mode="w+")
    sys.stderr.flush()
    saved = os.dup(2)
    os.dup2(log.fileno(), 2)
    try:
        for _ in range(2):
            bt.encode(rgb.data_ptr(), st)
            bt.encode_finish()
            ms_e = bt.stage_ms()
            bt.decode(bt.alc_ptr(0), bt.alc_stride, None, st)
            bt.decode_finish()
            ms = bt.stage_ms()
    finally:
        os.dup2(saved, 2)
        os.close(saved)
    log.seek(0)
    mc, ghz, where, dmc, dwhere = [], [], [], [], []
    for line in log.read().splitlines():
        m = re.search(r"encode chain \d+: \d+ bytes, ([0-9.]+) Mcycles, [0-9.]+ ms of 100 MHz ticks => ([0-9.]+) GHz, xcc (\d+) se (\d+) cu (\d+) simd (\d+)", line)
        if m:
            mc.append(float(m.group(1)))
            ghz.append(float(m.group(2)))
            where.append(tuple(int(v) for v in m.groups()[2:]))
        m = re.search(r"decode chain \d+: .* ([0-9.]+) Mcycles, xcc (\d+) se (\d+) cu (\d+) simd (\d+)", line)
        if m:
            dmc.append(float(m.group(1)))
            dwhere.append(tuple(int(v) for v in m.groups()[1:]))
    mc, ghz, where = np.array(mc[-3 * B:]), np.array(ghz[-3 * B:]), where[-3 * B:]
    dmc, dwhere = np.array(dmc[-3 * B:]), dwhere[-3 * B:]
    if os.environ.get("PROBE_PLACEMENT"):
        from collections import Counter
        for name, cyc, wh in (("encode", mc, where), ("decode", dmc, dwhere)):
            per_simd = Counter(wh)
            per_cu = Counter(w[:3] for w in wh)
            print(f"  {name}: waves per SIMD histogram {sorted(Counter(per_simd.values()).items())}, per CU {sorted(Counter(per_cu.values()).items())}, "
                  f"CUs used {len(per_cu)}")
            slow = cyc > 1.2 * np.median(cyc)
            shared = np.array([per_simd[w] > 1 for w in wh])
            print(f"  {name}: slow chains {int(slow.sum())}, of them on a shared SIMD {int((slow & shared).sum())}; chains on a shared SIMD {int(shared.sum())}; "
                  f"cycles/sym median {np.median(cyc) * 1e6 / px:.1f}, on shared SIMDs {cyc[shared].mean() * 1e6 / px if shared.any() else 0:.1f}")
    print(f"chunks {B:4d} chains {3 * B:5d}: encode {ms_e['rans_encode'] / px * 1e6:6.2f} ns/sym  decode {ms['rans_decode'] / px * 1e6:6.2f} ns/sym | "
          f"encoder cycles/sym mean {mc.mean() * 1e6 / px:5.1f} max {mc.max() * 1e6 / px:5.1f}  clock mean {ghz.mean():.2f} min {ghz.min():.2f} GHz | "
          f"aggregate enc {3 * B * px / ms_e['rans_encode'] / 1e3:7.0f} dec {3 * B * px / ms['rans_decode'] / 1e3:7.0f} Msym/s", flush=True)
    del bt, rgb
    a.load_library().alice_codec_trim()
    torch.cuda.empty_cache()
