"""Row-slab sharded encode/decode of one chunk on the real kernels: several ranks share
the one GPU of the test box, the exchanges run over `gloo` with host staging (RCCL refuses two ranks on one
device; on a multi-GPU node the same code runs with backend "nccl" and device tensors).  The sharded `.alc`
must equal the single-GPU encode of the whole chunk byte for byte, which in turn equals the oracle's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rgb(w, h, f, seed):
    rng = np.random.default_rng(seed)
    t, y, x = np.meshgrid(np.arange(f), np.arange(h), np.arange(w), indexing="ij")
    base = 128 + 90 * np.sin((x + 2 * t) / 23.0) * np.cos((y - t) / 16.1)
    rgb = np.stack([base, np.roll(base, 3, 2), np.roll(base, 5, 1)], -1) + rng.integers(-4, 5, (f, h, w, 3))
    return np.clip(np.rint(rgb), 0, 255).astype(np.uint8)


def _worker(rank, world, port, case, use_oracle, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import alice_codec_amd as ac
    from alice_codec_amd import slab
    w, h, f, q, wavelet = case
    rgb = _rgb(w, h, f, 11)
    geo = slab.SlabGeometry(w, h, f, world)
    a, b = geo.owned_real(rank)
    dev = torch.device("cuda", 0)
    mine = torch.from_numpy(rgb[:, a:b].copy()).to(dev)
    st = slab.DeviceStages(dev)
    alc = slab.encode_sharded(mine, w, h, f, q, wavelet, st, dst=0)
    ok = True
    want = None
    if rank == 0:
        enc = ac.FrameEncoder.with_wavelet(q, ac.WaveletType(wavelet))
        want = enc.encode(rgb.reshape(-1), w, h, f).to_bytes()
        got = alc.cpu().numpy().tobytes()
        ok &= got == want
        if use_oracle:
            import oracle as o
            ok &= got == o.encode(rgb.reshape(-1), w, h, f, q, wavelet)
    out, geo2 = slab.decode_sharded(alc, st, dev, src=0)
    whole = slab.gather_rows(out, geo2, dst=0)
    if rank == 0:
        full = ac.FrameDecoder().decode(ac.EncodedChunk.from_bytes(want)).reshape(f, h, w, 3)
        ok &= np.array_equal(whole.cpu().numpy(), full)
    ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    (2, (96, 64, 4, 80, 1), True),
    (3, (130, 75, 5, 90, 1), True),       # odd height / frames / unaligned width
    (3, (64, 48, 2, 70, 0), True),
    (4, (320, 272, 8, 80, 1), False),     # several tiles per slab; compared with the single-GPU encode
]


@pytest.mark.parametrize("world,case,use_oracle", CASES)
def test_slab_sharded_on_gpu(world, case, use_oracle):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 33000 + (os.getpid() % 2000) + 11 * world + case[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, use_oracle, ret)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert all(ret.get(timeout=5) for _ in range(world))
