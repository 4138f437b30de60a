"""Multi-GPU glue: independent 64-frame chunks are sharded across ranks (one process per GPU);
the only exchange is the gather of the finished `.alc` byte blobs on rank 0
(reference: chunks are self-contained bitstreams, src/pipeline.rs:461-497 has no cross-chunk state).

Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) with device tensors on the
GPU box, "gloo" with CPU tensors in the CPU test-suite."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_chunks(n_chunks_total: int, rank: int, world: int) -> List[int]:
    """Chunk k of the job belongs to rank k mod world (round-robin keeps ranks balanced)."""
    return [k for k in range(n_chunks_total) if k % world == rank]


def gather_alc(packed: torch.Tensor, sizes: torch.Tensor, dst: int = 0,
               group: Optional[dist.ProcessGroup] = None) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """Variable-length gather of per-rank `.alc` blobs.

    packed: uint8 tensor holding this rank's chunks back to back (only the first sizes.sum() bytes count).
    sizes : int64 tensor [chunks_per_rank] of each chunk's byte length.
    Returns on rank `dst`: (blob, all_sizes) with blob = rank 0's bytes, rank 1's bytes, ... and
    all_sizes of shape [world, chunks_per_rank]; None elsewhere.
    One small all_gather (lengths) plus one point-to-point transfer per peer: a fan-in on the
    root's links, no ring."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = sizes.to(packed.device, torch.int64)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes_t = torch.stack(all_sizes)
    totals = all_sizes_t.sum(dim=1).tolist()
    mine = int(totals[rank])
    if rank == dst:
        blob = torch.empty(int(sum(totals)), dtype=torch.uint8, device=packed.device)
        offs = [0]
        for t in totals:
            offs.append(offs[-1] + int(t))
        blob[offs[rank]:offs[rank] + mine].copy_(packed[:mine])
        ops = []
        for r in range(world):
            if r != dst and totals[r] > 0:
                ops.append(dist.P2POp(dist.irecv, blob[offs[r]:offs[r + 1]], r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return blob, all_sizes_t
    if mine > 0:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, packed[:mine].contiguous(), dst, group)]):
            req.wait()
    return None


def split_blob(blob: torch.Tensor, all_sizes: torch.Tensor) -> List[bytes]:
    """Cuts the gathered blob back into individual `.alc` byte strings (rank-major, chunk-major)."""
    out, off = [], 0
    host = blob.cpu().numpy().tobytes()
    for s in all_sizes.reshape(-1).tolist():
        out.append(host[off:off + int(s)])
        off += int(s)
    return out
