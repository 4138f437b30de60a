"""Row-slab sharding of ONE chunk across ranks (SURVEY.md §8e "C5", BASELINE.json configs[4]).

A chunk too large (or too urgent) for one GPU is cut into horizontal slabs: rank g owns the padded rows
[y0_g, y1_g) of every frame, with even boundaries so that the low/high pairing of the y lifting is kept.

encode
  1. halo exchange: every rank receives the HALO raw RGB rows above and below its slab from the ranks that
     own them (xGMI point to point).  The row lifting is recomputed on them, so nothing but raw pixels moves.
  2. each rank runs the whole forward stage (colour, x/y/t lifting, quantiser, symbols) on its local image.
     A sample after `n` lifting steps depends on the 2n+1 rows around it (n = 4 for CDF 9/7), so every owned
     row is at least HALO = 4 rows away from a local edge that is not a true frame edge and is therefore
     bit-identical to the single-GPU result; the halo outputs are dropped.
  3. the one real collective: all-reduce (sum) of the 3 x 256 histograms of the owned symbols.  Every rank can
     now build the reference's frequency tables (src/rans.rs:102-150).
  4. the single-stream rANS chain of a channel does not shard: the owned symbol rows are
     sent to the rank that runs the channel's chain (Y, Co, Cg on ranks 0, 1, 2 mod world), which puts them
     in the reference's symbol order [t][y: low | high][x] and encodes.
  5. the three streams go to `dst`, which writes the header (src/pipeline.rs:195-244) in front of them.
decode is the mirror image: header broadcast, streams to the chain ranks, symbol rows (with halo) to the
slab owners, inverse stage per rank.

The compute is supplied by a `stages` object; the product one is DeviceStages (HIP kernels through the C ABI,
tensors in HBM).  The exchange code is backend-agnostic: with "nccl" (RCCL) device tensors travel directly,
with "gloo" they are staged through host memory (used to rehearse several ranks on one GPU).
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

HALO = 4            # rows; covers the 4 lifting steps of CDF 9/7 (2 would do for CDF 5/3 / Haar)
ALC_MAGIC = b"ALCC"  # src/pipeline.rs:137
ALC_VERSION = 1
FIXED_HEADER = 18
CHANNEL_HEADER = 1040
HEADER_BYTES = FIXED_HEADER + 3 * CHANNEL_HEADER


def quality_to_step(quality: int) -> int:
    """src/pipeline.rs:456-457"""
    return max(64 - (min(int(quality), 100) * 63) // 100, 1)


def padded_dims(w: int, h: int, f: int) -> Tuple[int, int, int]:
    """src/pipeline.rs:437-439"""
    return w + (w & 1), h + (h & 1), 2 if f == 1 else f + (f & 1)


def slab_bounds(ph: int, world: int) -> List[Tuple[int, int]]:
    """Even split of the ph/2 row pairs: rank g owns padded rows [y0, y1), both even."""
    pairs = ph // 2
    return [(2 * ((pairs * g) // world), 2 * ((pairs * (g + 1)) // world)) for g in range(world)]


def chain_rank(channel: int, world: int) -> int:
    return channel % world


@dataclass
class SlabGeometry:
    w: int
    h: int
    f: int
    world: int

    def __post_init__(self):
        self.pw, self.ph, self.pf = padded_dims(self.w, self.h, self.f)
        self.bounds = slab_bounds(self.ph, self.world)

    def owned(self, r: int) -> Tuple[int, int]:
        """padded rows owned by rank r"""
        return self.bounds[r]

    def owned_real(self, r: int) -> Tuple[int, int]:
        """rows of the real image held by rank r (the pad row of an odd height exists on nobody)"""
        y0, y1 = self.bounds[r]
        return min(y0, self.h), min(y1, self.h)

    def local(self, r: int) -> Tuple[int, int]:
        """real rows [ys, ye) of rank r's local image: its slab plus HALO rows on each side, clipped to the frame"""
        y0, y1 = self.bounds[r]
        if y1 <= y0:
            return 0, 0
        return max(0, y0 - HALO), min(self.h, y1 + HALO)

    def local_padded_rows(self, r: int) -> int:
        ys, ye = self.local(r)
        return (ye - ys) + ((ye - ys) & 1)


def _overlap(a: Tuple[int, int], b: Tuple[int, int]) -> Tuple[int, int]:
    lo, hi = max(a[0], b[0]), min(a[1], b[1])
    return (lo, hi) if hi > lo else (0, 0)


class _Exchange:
    """A set of point-to-point transfers posted together (dist.batch_isend_irecv).  Messages between one pair
    of ranks are matched in posting order, so both sides enumerate them in the same order."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.host_staged = dist.get_backend(group) == "gloo"
        self.ops, self.post, self.keep = [], [], []

    def _peer(self, r: int) -> int:
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def send(self, t: torch.Tensor, dst: int):
        t = t.contiguous()
        if self.host_staged and t.is_cuda:
            t = t.cpu()
        self.keep.append(t)
        self.ops.append(dist.P2POp(dist.isend, t, self._peer(dst), self.group))

    def recv_into(self, out: torch.Tensor, src: int):
        """`out` may be any (strided) view; the data lands in a contiguous buffer first when needed."""
        direct = out.is_contiguous() and not (self.host_staged and out.is_cuda)
        buf = out if direct else torch.empty(out.shape, dtype=out.dtype, device="cpu" if self.host_staged else out.device)
        self.ops.append(dist.P2POp(dist.irecv, buf, self._peer(src), self.group))
        if not direct:
            self.post.append((out, buf))

    def run(self):
        if self.ops:
            for req in dist.batch_isend_irecv(self.ops):
                req.wait()
        for out, buf in self.post:
            out.copy_(buf)
        self.ops, self.post, self.keep = [], [], []


def _all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, group=group)
        return h.to(t.device)
    dist.all_reduce(t, group=group)
    return t


def _broadcast(t: torch.Tensor, src: int, group=None) -> torch.Tensor:
    g_src = dist.get_global_rank(group, src) if group is not None else src
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, g_src, group=group)
        return h.to(t.device)
    dist.broadcast(t, g_src, group=group)
    return t


# ---------------------------------------------------------------------------------------------
# compute provider: HIP kernels through the C ABI
# ---------------------------------------------------------------------------------------------

class DeviceStages:
    """The five device-resident stage calls of include/alice_codec.h (part 3) on torch CUDA tensors."""

    def __init__(self, device: Optional[torch.device] = None):
        from . import load_library, set_device
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceStages needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        set_device(self.device.index or 0)
        self.lib = load_library()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _check(rc: int):
        from . import _check
        _check(rc)

    def forward_symbols(self, rgb: torch.Tensor, wavelet: int, quality: int) -> torch.Tensor:
        f, h, w, _ = rgb.shape
        pw, ph, pf = padded_dims(w, h, f)
        rgb = rgb.contiguous()
        sym = torch.empty((3, pf, ph, pw), dtype=torch.uint8, device=self.device)
        self._check(self.lib.alice_codec_dev_forward_symbols(rgb.data_ptr(), w, h, f, wavelet, quality, sym.data_ptr(), None, self._stream()))
        return sym

    def inverse_symbols(self, sym: torch.Tensor, w: int, h: int, f: int, wavelet: int, steps: Sequence[int]) -> torch.Tensor:
        sym = sym.contiguous()
        rgb = torch.empty((f, h, w, 3), dtype=torch.uint8, device=self.device)
        st = (C.c_int32 * 3)(*[int(s) for s in steps])
        self._check(self.lib.alice_codec_dev_inverse_symbols(sym.data_ptr(), w, h, f, wavelet, st, rgb.data_ptr(), self._stream()))
        return rgb

    def histogram(self, sym: torch.Tensor) -> torch.Tensor:
        sym = sym.contiguous()
        hist = torch.zeros(256, dtype=torch.int32, device=self.device)
        self._check(self.lib.alice_codec_dev_histogram(sym.data_ptr() if sym.numel() else None, sym.numel(), hist.data_ptr(), self._stream()))
        return (hist.to(torch.int64) & 0xFFFFFFFF)

    def rans_encode(self, sym: torch.Tensor, hist: np.ndarray) -> torch.Tensor:
        sym = sym.contiguous()
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        hp = hist.ctypes.data_as(C.POINTER(C.c_uint32))
        cap = int(self.lib.alice_codec_rans_stream_bound(hp, sym.numel()))
        out = torch.empty(cap, dtype=torch.uint8, device=self.device)
        off, ln = C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.alice_codec_dev_rans_encode(sym.data_ptr() if sym.numel() else None, sym.numel(), hp, out.data_ptr(), cap,
                                                         C.byref(off), C.byref(ln), self._stream()))
        return out[off.value:off.value + ln.value]

    def rans_decode(self, stream: torch.Tensor, hist: np.ndarray, n: int) -> torch.Tensor:
        stream = stream.contiguous()
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        out = torch.empty(n, dtype=torch.uint8, device=self.device)
        self._check(self.lib.alice_codec_dev_rans_decode(stream.data_ptr() if stream.numel() else None, stream.numel(),
                                                         hist.ctypes.data_as(C.POINTER(C.c_uint32)), out.data_ptr(), n, self._stream()))
        return out


# ---------------------------------------------------------------------------------------------
# .alc header (host side; src/pipeline.rs:195-313)
# ---------------------------------------------------------------------------------------------

def build_header(w: int, h: int, f: int, wavelet: int, step: int, lens: Sequence[int], num_symbols: int, hists: np.ndarray) -> bytes:
    out = bytearray(ALC_MAGIC)
    out += struct.pack("<BBIII", ALC_VERSION, wavelet, w, h, f)
    for c in range(3):
        out += struct.pack("<IiiI", int(lens[c]), step, step, num_symbols)
        out += np.ascontiguousarray(hists[c], dtype="<u4").tobytes()
    assert len(out) == HEADER_BYTES
    return bytes(out)


@dataclass
class AlcHeader:
    w: int
    h: int
    f: int
    wavelet: int
    lens: List[int]
    steps: List[int]
    dead_zones: List[int]
    num_symbols: List[int]
    hists: np.ndarray


def parse_header(raw: bytes) -> AlcHeader:
    """EncodedChunk::from_bytes header checks (src/pipeline.rs:254-313); raises CodecError(InvalidBitstream)."""
    from . import CodecError
    if len(raw) < FIXED_HEADER:
        raise CodecError(4, "header too short")
    if raw[:4] != ALC_MAGIC:
        raise CodecError(4, "invalid magic bytes")
    ver, wavelet, w, h, f = struct.unpack_from("<BBIII", raw, 4)
    if ver != ALC_VERSION:
        raise CodecError(4, f"unsupported version: {ver}")
    if wavelet > 2:
        raise CodecError(4, f"unknown wavelet type: {wavelet}")
    if len(raw) < HEADER_BYTES:
        raise CodecError(4, "channel header too short")
    lens, steps, dzs, nsym = [], [], [], []
    hists = np.zeros((3, 256), dtype=np.uint32)
    for c in range(3):
        o = FIXED_HEADER + c * CHANNEL_HEADER
        ln, st, dz, ns = struct.unpack_from("<IiiI", raw, o)
        lens.append(ln); steps.append(st); dzs.append(dz); nsym.append(ns)
        hists[c] = np.frombuffer(raw, dtype="<u4", count=256, offset=o + 16)
    return AlcHeader(w, h, f, wavelet, lens, steps, dzs, nsym, hists)


# ---------------------------------------------------------------------------------------------
# the sharded encode / decode
# ---------------------------------------------------------------------------------------------

def exchange_halos(rgb_slab: torch.Tensor, geo: SlabGeometry, group=None) -> torch.Tensor:
    """rgb_slab: this rank's real rows [f, rows, w, 3].  Returns its local image [f, ye-ys, w, 3]."""
    rank = dist.get_rank(group)
    ys, ye = geo.local(rank)
    mine = geo.owned_real(rank)
    f = geo.f
    local = torch.empty((f, ye - ys, geo.w, 3), dtype=torch.uint8, device=rgb_slab.device)
    if ye > ys:
        local[:, mine[0] - ys:mine[1] - ys] = rgb_slab
    ex = _Exchange(group)
    for r in range(geo.world):
        if r == rank:
            continue
        # what r needs from me / what I need from r; both sides walk the peers in rank order
        need_r = geo.local(r)
        a, b = _overlap(need_r, mine)
        if b > a:
            ex.send(rgb_slab[:, a - mine[0]:b - mine[0]], r)
        a, b = _overlap((ys, ye), geo.owned_real(r))
        if b > a:
            ex.recv_into(local[:, a - ys:b - ys], r)
    ex.run()
    return local


def encode_sharded(rgb_slab: torch.Tensor, w: int, h: int, f: int, quality: int, wavelet: int, stages,
                   dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Encode one w x h x f chunk whose rows are spread over the ranks of `group`.

    rgb_slab: uint8 [f, rows_g, w, 3], the real rows geo.owned_real(rank) of every frame.
    Returns the `.alc` bytes (uint8 tensor) on rank `dst`, None elsewhere.  Bit-identical to
    FrameEncoder::with_wavelet(quality, wavelet).encode on the whole chunk."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    geo = SlabGeometry(w, h, f, world)
    pw, ph, pf = geo.pw, geo.ph, geo.pf
    dev = rgb_slab.device
    y0, y1 = geo.owned(rank)
    ys, _ = geo.local(rank)
    rows2 = (y1 - y0) // 2

    # 1-2. halos, local forward stage, keep the owned rows
    local = exchange_halos(rgb_slab, geo, group)
    if rows2 > 0:
        sym = stages.forward_symbols(local, wavelet, quality)                       # [3, pf, phl, pw]
        phl = sym.shape[2]
        a = (y0 - ys) // 2
        owned = sym.view(3, pf, 2, phl // 2, pw)[:, :, :, a:a + rows2, :].contiguous()   # [3, pf, 2, rows2, pw]
        del sym
    else:
        owned = torch.empty((3, pf, 2, 0, pw), dtype=torch.uint8, device=dev)

    # 3. histograms of the owned symbols, summed over the ranks
    hist = torch.stack([stages.histogram(owned[c]) for c in range(3)]).to(torch.int64)
    hist = _all_reduce_sum(hist, group).cpu().numpy().astype(np.uint32)

    # 4. owned rows -> chain ranks, reference symbol order, one chain per channel
    ex = _Exchange(group)
    channel = {}
    for c in range(3):
        root = chain_rank(c, world)
        if rank == root:
            channel[c] = torch.empty((pf, 2, ph // 2, pw), dtype=torch.uint8, device=dev)
            for r in range(world):
                b0, b1 = geo.owned(r)
                if b1 <= b0:
                    continue
                view = channel[c][:, :, b0 // 2:b1 // 2, :]
                if r == rank:
                    view.copy_(owned[c])
                else:
                    ex.recv_into(view, r)
        elif rows2 > 0:
            ex.send(owned[c], root)
    ex.run()
    del owned
    streams = {c: stages.rans_encode(channel[c].view(-1), hist[c]) for c in channel}
    channel.clear()

    # 5. stream lengths to everybody, streams to dst, header in front
    lens = torch.zeros(3, dtype=torch.int64, device=dev)
    for c, s in streams.items():
        lens[c] = s.numel()
    lens = _all_reduce_sum(lens, group).cpu().tolist()
    ex = _Exchange(group)
    alc = None
    if rank == dst:
        total = HEADER_BYTES + int(sum(lens))
        alc = torch.empty(total, dtype=torch.uint8, device=dev)
        header = build_header(w, h, f, wavelet, quality_to_step(quality), lens, pw * ph * pf, hist)
        alc[:HEADER_BYTES] = torch.frombuffer(bytearray(header), dtype=torch.uint8).to(dev)
    off = HEADER_BYTES
    for c in range(3):
        root = chain_rank(c, world)
        if lens[c] > 0:
            if rank == dst and root == rank:
                alc[off:off + lens[c]] = streams[c]
            elif rank == dst:
                ex.recv_into(alc[off:off + lens[c]], root)
            elif rank == root:
                ex.send(streams[c], dst)
        off += int(lens[c])
    ex.run()
    return alc


def decode_sharded(alc: Optional[torch.Tensor], stages, device, src: int = 0, group=None) -> Tuple[torch.Tensor, SlabGeometry]:
    """Decode one `.alc` held by rank `src`; every rank gets the rows it owns: ([f, rows_g, w, 3], geometry).
    Bit-identical to the corresponding rows of FrameDecoder::decode."""
    from . import CodecError
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device(device)
    # header to everybody
    hdr = torch.zeros(HEADER_BYTES + 8, dtype=torch.uint8, device=dev)
    if rank == src:
        alc = alc.to(dev)
        n = min(int(alc.numel()), HEADER_BYTES)
        hdr[:n] = alc[:n].to(dev)
        hdr[HEADER_BYTES:] = torch.frombuffer(bytearray(struct.pack("<Q", int(alc.numel()))), dtype=torch.uint8).to(dev)
    hdr = _broadcast(hdr, src, group).cpu().numpy().tobytes()
    total_len = struct.unpack("<Q", hdr[HEADER_BYTES:])[0]
    head = parse_header(hdr[:min(total_len, HEADER_BYTES)])
    w, h, f = head.w, head.h, head.f
    if w == 0 or h == 0 or f == 0:
        raise CodecError(2, "empty chunk: nothing to shard")
    geo = SlabGeometry(w, h, f, world)
    pw, ph, pf = geo.pw, geo.ph, geo.pf
    padded = pw * ph * pf
    off = HEADER_BYTES
    for c in range(3):                                  # src/pipeline.rs:545-579
        if head.num_symbols[c] != padded:
            raise CodecError(4, f"channel {c}: num_symbols {head.num_symbols[c]} != padded_pixels {padded}")
        if off + head.lens[c] > total_len:
            raise CodecError(4, f"channel {c}: compressed data overrun")
        off += head.lens[c]

    # streams to the chain ranks
    ex = _Exchange(group)
    streams = {}
    off = HEADER_BYTES
    for c in range(3):
        root = chain_rank(c, world)
        ln = head.lens[c]
        if rank == root:
            streams[c] = torch.empty(ln, dtype=torch.uint8, device=dev)
            if ln:
                if rank == src:
                    streams[c].copy_(alc[off:off + ln])
                else:
                    ex.recv_into(streams[c], src)
        elif rank == src and ln:
            ex.send(alc[off:off + ln], root)
        off += ln
    ex.run()
    channel = {c: stages.rans_decode(s, head.hists[c], padded).view(pf, 2, ph // 2, pw) for c, s in streams.items()}

    # symbol rows (with halo) to the slab owners
    ys, ye = geo.local(rank)
    phl = geo.local_padded_rows(rank)
    sym = torch.empty((3, pf, 2, phl // 2, pw), dtype=torch.uint8, device=dev)
    ex = _Exchange(group)
    for c in range(3):
        root = chain_rank(c, world)
        if rank == root:
            for r in range(world):
                rs, _ = geo.local(r)
                n2 = geo.local_padded_rows(r) // 2
                if n2 == 0:
                    continue
                part = channel[c][:, :, rs // 2:rs // 2 + n2, :]
                if r == rank:
                    sym[c].copy_(part)
                else:
                    ex.send(part, r)
        elif phl > 0:
            ex.recv_into(sym[c], root)
    ex.run()
    channel.clear()

    y0, _ = geo.owned(rank)
    r0, r1 = geo.owned_real(rank)
    if phl > 0:
        rgb = stages.inverse_symbols(sym.view(3, pf, phl, pw), w, ye - ys, f, head.wavelet, head.steps)
        out = rgb[:, r0 - ys:r1 - ys].contiguous()
    else:
        out = torch.empty((f, 0, w, 3), dtype=torch.uint8, device=dev)
    return out, geo


def gather_rows(slab: torch.Tensor, geo: SlabGeometry, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Reassembles the decoded slabs into [f, h, w, 3] on rank dst."""
    rank = dist.get_rank(group)
    ex = _Exchange(group)
    full = None
    if rank == dst:
        full = torch.empty((geo.f, geo.h, geo.w, 3), dtype=torch.uint8, device=slab.device)
        for r in range(geo.world):
            a, b = geo.owned_real(r)
            if b <= a:
                continue
            if r == rank:
                full[:, a:b] = slab
            else:
                ex.recv_into(full[:, a:b], r)
    elif slab.shape[1] > 0:
        ex.send(slab, dst)
    ex.run()
    return full
