"""Multi-GPU glue: independent 64-frame chunks are sharded across ranks (one process per GPU);
the only exchange is the gather of the finished `.alc` byte blobs on rank 0
(reference: chunks are self-contained bitstreams, src/pipeline.rs:461-497 has no cross-chunk state).

Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) with device tensors on the
GPU box, "gloo" with CPU tensors in the CPU test-suite."""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_chunks(n_chunks_total: int, rank: int, world: int) -> List[int]:
    """Chunk k of the job belongs to rank k mod world (round-robin keeps ranks balanced)."""
    return [k for k in range(n_chunks_total) if k % world == rank]


class PendingGather:
    """Handle of a gather started with gather_alc_start(): wait() returns what gather_alc() returns."""

    def __init__(self, reqs, result):
        self._reqs, self._result = reqs, result

    def wait(self):
        for r in self._reqs:
            r.wait()
        self._reqs = []
        return self._result


def gather_alc(packed: torch.Tensor, sizes: torch.Tensor, dst: int = 0,
               group: Optional[dist.ProcessGroup] = None) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """Blocking form of gather_alc_start()."""
    return gather_alc_start(packed, sizes, dst, group).wait()


def gather_alc_start(packed: Optional[torch.Tensor], sizes: torch.Tensor, dst: int = 0,
                     group: Optional[dist.ProcessGroup] = None, blob: Optional[torch.Tensor] = None,
                     pack_fn: Optional[Callable[[torch.Tensor], None]] = None,
                     device: Optional[torch.device] = None) -> PendingGather:
    """Variable-length gather of per-rank `.alc` blobs.

    packed: uint8 tensor holding this rank's chunks back to back (only the first sizes.sum() bytes count).
    sizes : int64 tensor [chunks_per_rank] of each chunk's byte length.
    Returns on rank `dst`: (blob, all_sizes) with blob = rank 0's bytes, rank 1's bytes, ... and
    all_sizes of shape [world, chunks_per_rank]; None elsewhere.
    One small all_gather (lengths) plus one point-to-point transfer per peer: a fan-in on the
    root's links, no ring.  The transfers are left in flight (RCCL runs them on its own stream) so the
    caller can overlap them with the decode; `blob` lets the root reuse its receive buffer across steps.
    pack_fn (root only): writes the root's own bytes straight into its slice of the blob, so that the root
    needs no `packed` staging buffer of its own (pass packed=None and `device`)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = packed.device if packed is not None else torch.device(device)
    sizes = sizes.to(dev, torch.int64)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes_t = torch.stack(all_sizes)
    totals = all_sizes_t.sum(dim=1).tolist()
    mine = int(totals[rank])
    if rank == dst:
        need = int(sum(totals))
        if blob is None or blob.numel() < need:
            blob = torch.empty(need, dtype=torch.uint8, device=dev)
        blob = blob[:need]
        offs = [0]
        for t in totals:
            offs.append(offs[-1] + int(t))
        if pack_fn is not None:
            pack_fn(blob[offs[rank]:offs[rank] + mine])
        else:
            blob[offs[rank]:offs[rank] + mine].copy_(packed[:mine])
        ops = []
        for r in range(world):
            if r != dst and totals[r] > 0:
                ops.append(dist.P2POp(dist.irecv, blob[offs[r]:offs[r + 1]], r, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return PendingGather(reqs, (blob, all_sizes_t))
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, packed[:mine], dst, group)]) if mine > 0 else []
    return PendingGather(reqs, None)


class DeviceView:
    """Zero-copy torch view of library-owned device memory (the batch's `.alc` buffers): torch.as_tensor() reads the
    CUDA array interface, which HIP tensors share."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    def tensor(self, device) -> torch.Tensor:
        return torch.as_tensor(self, device=device)


def stream_alc_to_root(get_chunk: Callable[[int], torch.Tensor], sizes: torch.Tensor,
                       sink: Callable[[int, int, torch.Tensor], None], dst: int = 0,
                       group: Optional[dist.ProcessGroup] = None, device=None, depth: int = 2) -> Optional[torch.Tensor]:
    """The gather without a blob on the root: every rank's chunks pass through a small ring of receive slots on rank
    `dst` and are handed to `sink(rank, chunk_index, bytes)` one at a time (bench.py: a checksum on the device, optionally a copy on to pinned host memory;
    a real front end: the file or socket the `.alc` stream goes to).  The root's HBM then holds `depth` slots per peer
    (a slot = the largest chunk), not world x chunks x chunk size -- gather_alc_start() with 8 GPUs x 285 chunks x 0.11 GB
    would need 250 GB on rank 0 and used to cap every rank's chunks in flight.

    get_chunk(i): uint8 tensor holding this rank's chunk i in its first sizes[i] bytes (a view, e.g. DeviceView of the
    batch's own buffer: nothing is packed or staged on the senders).  sizes: int64 [chunks_per_rank], the same count on
    every rank.  The tensor given to `sink` is only valid during the call (device work it enqueues on the current stream
    is ordered before the slot's next receive).  Chunk i of all peers travels in one grouped point-to-point batch: a
    fan-in on the root's links, no ring, no collective on the data path.  Returns all_sizes [world, chunks] on the root."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = int(sizes.numel())
    probe = get_chunk(0) if n else torch.empty(0, dtype=torch.uint8, device=device or "cpu")
    dev = probe.device
    sizes_d = sizes.to(dev, torch.int64)
    all_sizes = [torch.empty_like(sizes_d) for _ in range(world)]
    dist.all_gather(all_sizes, sizes_d, group=group)
    all_sizes_t = torch.stack(all_sizes).cpu()
    if rank != dst:
        for i in range(n):
            m = int(all_sizes_t[rank, i])
            if m > 0:
                for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, get_chunk(i)[:m], dst, group)]):
                    r.wait()
        return None
    slot_bytes = int(all_sizes_t.max()) if all_sizes_t.numel() else 0
    peers = [r for r in range(world) if r != dst]
    ring = {r: [torch.empty(max(slot_bytes, 1), dtype=torch.uint8, device=dev) for _ in range(depth)] for r in peers}
    inflight: list = []   # (chunk index, [(peer, slot view)], requests)

    def post(i: int):
        ops, views = [], []
        for r in peers:
            m = int(all_sizes_t[r, i])
            if m > 0:
                v = ring[r][i % depth][:m]
                ops.append(dist.P2POp(dist.irecv, v, r, group))
                views.append((r, v))
        inflight.append((i, views, dist.batch_isend_irecv(ops) if ops else []))

    for i in range(min(depth - 1, n)):
        post(i)
    for i in range(n):
        if i + depth - 1 < n:
            post(i + depth - 1)   # its slots were handed to the sink `depth` chunks ago
        m = int(all_sizes_t[dst, i])
        if m > 0:
            sink(dst, i, get_chunk(i)[:m])
        j, views, reqs = inflight.pop(0)
        for r in reqs:
            r.wait()
        for r, v in views:
            sink(r, j, v)
    return all_sizes_t


def split_blob(blob: torch.Tensor, all_sizes: torch.Tensor) -> List[bytes]:
    """Cuts the gathered blob back into individual `.alc` byte strings (rank-major, chunk-major)."""
    out, off = [], 0
    host = blob.cpu().numpy().tobytes()
    for s in all_sizes.reshape(-1).tolist():
        out.append(host[off:off + int(s)])
        off += int(s)
    return out
