"""world_size-2 `gloo` rehearsal of the multi-GPU path: chunks shard round-robin across ranks with no
data-path collective, and one variable-length gather reassembles every rank's `.alc` blobs on rank 0.
The blobs here come from the CPU oracle (this is a test of the exchange, not of the kernels)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, chunks_total, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as o
    import alice_codec_amd  # noqa: F401  (package import must not need a GPU)
    from alice_codec_amd import multi
    mine = multi.shard_chunks(chunks_total, rank, world)
    w, h, f = 16, 8, 4
    blobs = []
    for k in mine:
        rgb = np.random.default_rng(k).integers(0, 256, w * h * f * 3, dtype=np.uint8)
        blobs.append(o.encode(rgb, w, h, f, 80, 1))
    sizes = torch.tensor([len(b) for b in blobs], dtype=torch.int64)
    packed = torch.from_numpy(np.frombuffer(b"".join(blobs) + bytes(64), np.uint8).copy())  # slack past the payload
    if rank == 0 and chunks_total == 6:
        # the root writes its own bytes straight into its slice of the blob (no staging buffer), as bench.py does
        mine_n = int(sizes.sum())
        res = multi.gather_alc_start(None, sizes, dst=0, device="cpu",
                                     pack_fn=lambda dst: dst.copy_(packed[:mine_n])).wait()
    else:
        res = multi.gather_alc(packed, sizes, dst=0)
    if rank == 0:
        blob, all_sizes = res
        parts = multi.split_blob(blob, all_sizes)
        ok = True
        idx = 0
        for r in range(world):
            for k in multi.shard_chunks(chunks_total, r, world):
                rgb = np.random.default_rng(k).integers(0, 256, w * h * f * 3, dtype=np.uint8)
                ok &= parts[idx] == o.encode(rgb, w, h, f, 80, 1)
                ok &= np.array_equal(o.decode(parts[idx]), o.decode(o.encode(rgb, w, h, f, 80, 1)))
                idx += 1
        ret.put(bool(ok and idx == chunks_total))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("chunks_total", [4, 6])
def test_gather_alc_world2(chunks_total):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + chunks_total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, chunks_total, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert ret.get(timeout=5) is True


def _stream_worker(rank, world, port, per_rank, ret):
    """The drained variant (multi.stream_alc_to_root): no blob on the root, chunks pass through a two-slot ring per peer
    and reach a sink one by one; chunk sizes differ, one rank holds an empty chunk."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as o
    from alice_codec_amd import multi

    def blob_of(r, i):
        if r == 1 and i == 1:
            return b""                                   # a rank with nothing to send for one chunk
        w, h, f = 8 + 2 * i, 6 + r, 2 + (i & 1)
        rgb = np.random.default_rng(100 * r + i).integers(0, 256, w * h * f * 3, dtype=np.uint8)
        return o.encode(rgb, w, h, f, 80, (r + i) % 3)

    mine = [blob_of(rank, i) for i in range(per_rank)]
    bufs = [torch.from_numpy(np.frombuffer(b + bytes(32), np.uint8).copy()) for b in mine]   # views longer than the chunk
    sizes = torch.tensor([len(b) for b in mine], dtype=torch.int64)
    got = {}
    order = []

    def sink(r, i, t):
        got[(r, i)] = t.numpy().tobytes()   # the view is only valid during the call
        order.append((r, i))

    all_sizes = multi.stream_alc_to_root(lambda i: bufs[i], sizes, sink, dst=0)
    if rank == 0:
        ok = all_sizes.shape == (world, per_rank)
        for r in range(world):
            for i in range(per_rank):
                want = blob_of(r, i)
                ok &= int(all_sizes[r, i]) == len(want)
                ok &= (got.get((r, i), b"") == want)
        ok &= len(got) == sum(1 for r in range(world) for i in range(per_rank) if blob_of(r, i))
        ok &= [i for _, i in order] == sorted(i for _, i in order)   # chunk-major: what a stream writer wants
        ret.put(bool(ok))
    else:
        assert all_sizes is None and not got
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,per_rank", [(2, 5), (3, 1)])
def test_stream_alc_to_root(world, per_rank):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + 7 * world
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, per_rank, ret)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert ret.get(timeout=5) is True


def test_shard_chunks_partition():
    from alice_codec_amd import multi  # noqa
    for world in (1, 2, 3, 8):
        seen = sorted(k for r in range(world) for k in multi.shard_chunks(19, r, world))
        assert seen == list(range(19))
