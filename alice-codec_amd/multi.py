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


def split_blob(blob: torch.Tensor, all_sizes: torch.Tensor) -> List[bytes]:
    """Cuts the gathered blob back into individual `.alc` byte strings (rank-major, chunk-major)."""
    out, off = [], 0
    host = blob.cpu().numpy().tobytes()
    for s in all_sizes.reshape(-1).tolist():
        out.append(host[off:off + int(s)])
        off += int(s)
    return out
