"""`gloo` rehearsal of the row-slab sharded path (one chunk split across ranks): halo
exchange, histogram all-reduce, symbol-row gather to the chain ranks, stream gather, and the mirror-image
decode.  The compute behind the exchange is the CPU oracle here (tests/slab_oracle_stages.py); the result must
be byte-identical to the oracle's single-process encode/decode of the whole chunk."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rgb(w, h, f, seed):
    rng = np.random.default_rng(seed)
    t, y, x = np.meshgrid(np.arange(f), np.arange(h), np.arange(w), indexing="ij")
    base = 128 + 90 * np.sin((x + 2 * t) / 5.0) * np.cos((y - t) / 3.5)
    rgb = np.stack([base, np.roll(base, 3, 2), np.roll(base, 5, 1)], -1) + rng.integers(-4, 5, (f, h, w, 3))
    return np.clip(np.rint(rgb), 0, 255).astype(np.uint8)


def _worker(rank, world, port, case, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as o
    from alice_codec_amd import slab
    from slab_oracle_stages import OracleStages
    w, h, f, q, wavelet = case
    rgb = _rgb(w, h, f, 7)
    geo = slab.SlabGeometry(w, h, f, world)
    a, b = geo.owned_real(rank)
    mine = torch.from_numpy(rgb[:, a:b].copy())
    st = OracleStages()
    alc = slab.encode_sharded(mine, w, h, f, q, wavelet, st, dst=0)
    want = o.encode(rgb.reshape(-1), w, h, f, q, wavelet)
    ok = True
    if rank == 0:
        ok &= alc.numpy().tobytes() == want
    else:
        ok &= alc is None
    out, geo2 = slab.decode_sharded(alc, st, "cpu", src=0)
    full = o.decode(want).reshape(f, h, w, 3)
    ok &= np.array_equal(out.numpy(), full[:, a:b])
    whole = slab.gather_rows(out, geo2, dst=0)
    if rank == 0:
        ok &= np.array_equal(whole.numpy(), full)
    ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    (2, (20, 22, 4, 80, 1)),      # CDF 9/7, two slabs
    (3, (18, 21, 3, 90, 1)),      # odd height and odd frame count: pad row lives on the last slab
    (3, (16, 26, 2, 70, 0)),      # CDF 5/3
    (2, (12, 16, 2, 100, 2)),     # "Haar", step 1
    (4, (10, 6, 2, 80, 1)),       # more ranks than row pairs per rank can fill: some slabs are thinner than the halo
    (5, (8, 4, 2, 80, 1)),        # a rank that owns nothing
]


@pytest.mark.parametrize("world,case", CASES)
def test_slab_sharded_matches_whole_chunk(world, case):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 31000 + (os.getpid() % 2000) + 7 * world + case[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, ret)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(180) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert all(ret.get(timeout=5) for _ in range(world))


def test_slab_bounds_partition():
    from alice_codec_amd import slab
    for ph in (2, 4, 6, 22, 1080, 4320):
        for world in (1, 2, 3, 5, 8):
            b = slab.slab_bounds(ph, world)
            assert b[0][0] == 0 and b[-1][1] == ph
            assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert all(x[0] % 2 == 0 and x[1] % 2 == 0 and x[1] >= x[0] for x in b)


def test_header_round_trip():
    from alice_codec_amd import slab
    hists = np.arange(768, dtype=np.uint32).reshape(3, 256)
    raw = slab.build_header(1920, 1080, 64, 1, 14, [10, 20, 30], 1920 * 1080 * 64, hists)
    h = slab.parse_header(raw)
    assert (h.w, h.h, h.f, h.wavelet, h.lens, h.steps, h.dead_zones) == (1920, 1080, 64, 1, [10, 20, 30], [14] * 3, [14] * 3)
    assert np.array_equal(h.hists, hists)
