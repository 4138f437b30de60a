"""One-off full-size check of the row-slab sharded path (BASELINE.json configs[3]/[4] shapes) on ONE GPU box:
`world` ranks share cuda:0 and exchange over gloo.  Compares the sharded `.alc` with the single-GPU encode of the
whole chunk (and, with --oracle, with the CPU oracle), and the sharded decode with the single-GPU decode.

  python tests/tools/slab_fullsize_check.py W H F QUALITY WAVELET WORLD [--oracle | --oracle3]

--oracle3 runs the oracle with Y, Co, Cg on three threads (the same bytes, a third of the wait).

Prints one JSON line per phase (progress) and a final summary on rank 0."""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synth_rows(dev, w, h, f, r0, r1):
    """S-smooth rows [r0, r1) of every frame, [f, r1-r0, w, 3] u8 on `dev`; integer-hash noise so that every
    rank (and any host) regenerates the same bytes."""
    out = torch.empty((f, r1 - r0, w, 3), dtype=torch.uint8, device=dev)
    if r1 <= r0:
        return out
    y = torch.arange(r0, r1, device=dev, dtype=torch.float32).view(-1, 1, 1)
    x = torch.arange(w, device=dev, dtype=torch.float32).view(1, -1, 1)
    s = torch.tensor([23.0, 31.0, 17.0], device=dev).view(1, 1, 3)
    ph = torch.tensor([0.0, 1.0, 2.0], device=dev).view(1, 1, 3)
    yi = torch.arange(r0, r1, device=dev, dtype=torch.int64).view(-1, 1, 1)
    xi = torch.arange(w, device=dev, dtype=torch.int64).view(1, -1, 1)
    ci = torch.arange(3, device=dev, dtype=torch.int64).view(1, 1, 3)
    for t in range(f):
        base = 128 + 90 * torch.sin((x + 2 * t) / s + ph) * torch.cos((y - t) / (0.7 * s))
        hsh = (xi * 73856093) ^ (yi * 19349663) ^ (ci * 83492791 + t * 2654435761)
        noise = ((hsh >> 7) % 9) - 4
        out[t] = (base + noise).round().clamp(0, 255).to(torch.uint8)
    return out


def worker(rank, world, port, args, use_oracle):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import alice_codec_amd as ac
    from alice_codec_amd import slab
    w, h, f, q, wavelet = args
    dev = torch.device("cuda", 0)
    geo = slab.SlabGeometry(w, h, f, world)
    a, b = geo.owned_real(rank)
    mine = synth_rows(dev, w, h, f, a, b)
    st = slab.DeviceStages(dev)

    def say(**kw):
        if rank == 0:
            print(json.dumps(kw), flush=True)

    dist.barrier()
    t0 = time.time()
    alc = slab.encode_sharded(mine, w, h, f, q, wavelet, st, dst=0)
    torch.cuda.synchronize()
    dist.barrier()
    t_enc = time.time() - t0
    say(phase="sharded encode", seconds=round(t_enc, 2), alc_bytes=int(alc.numel()) if rank == 0 else None)
    t0 = time.time()
    out, geo2 = slab.decode_sharded(alc, st, dev, src=0)
    torch.cuda.synchronize()
    dist.barrier()
    t_dec = time.time() - t0
    say(phase="sharded decode", seconds=round(t_dec, 2))
    same_in = bool(torch.equal(out, mine))
    whole = slab.gather_rows(out, geo2, dst=0)
    del out, mine
    res = {}
    if rank == 0:
        got = alc.cpu().numpy()
        dec = whole.cpu().numpy().reshape(-1)
        del alc, whole
        torch.cuda.empty_cache()
        full = synth_rows(dev, w, h, f, 0, h).cpu().numpy().reshape(-1)
        torch.cuda.empty_cache()
        t0 = time.time()
        chunk = ac.FrameEncoder.with_wavelet(q, ac.WaveletType(wavelet)).encode(full, w, h, f)
        want = np.frombuffer(chunk.to_bytes(), np.uint8)
        say(phase="single-GPU encode (host buffers)", seconds=round(time.time() - t0, 2))
        res["alc_equal_single_gpu"] = bool(got.size == want.size and np.array_equal(got, want))
        t0 = time.time()
        dec1 = ac.FrameDecoder().decode(chunk)
        say(phase="single-GPU decode (host buffers)", seconds=round(time.time() - t0, 2))
        res["decode_equal_single_gpu"] = bool(np.array_equal(dec, dec1))
        res["psnr_db"] = round(float(ac.psnr(full[:1 << 30], dec1[:1 << 30])), 2)
        if use_oracle:
            import oracle as o
            o.build()
            t0 = time.time()
            ref = np.frombuffer(o.encode(full, w, h, f, q, wavelet, three_threads=use_oracle == 3), np.uint8)
            say(phase="oracle encode (1 CPU thread)", seconds=round(time.time() - t0, 2))
            res["alc_equal_oracle"] = bool(ref.size == got.size and np.array_equal(ref, got))
            t0 = time.time()
            rdec = o.decode(ref, three_threads=use_oracle == 3)
            say(phase="oracle decode (1 CPU thread)", seconds=round(time.time() - t0, 2))
            res["decode_equal_oracle"] = bool(np.array_equal(rdec, dec))
        res.update(shape=[w, h, f], quality=q, wavelet=wavelet, world=world, alc_bytes=int(got.size),
                   alc_sha256=hashlib.sha256(got.tobytes()).hexdigest(), sharded_encode_s=round(t_enc, 2),
                   sharded_decode_s=round(t_dec, 2), decoded_equals_input=same_in,
                   note="ranks share one GPU and exchange over gloo with host staging: times are not xGMI times")
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    w, h, f, q, wavelet, world = (int(v) for v in a[:6])
    ctx = mp.get_context("spawn")
    port = 34000 + os.getpid() % 2000
    use_oracle = 3 if "--oracle3" in sys.argv else (1 if "--oracle" in sys.argv else 0)
    procs = [ctx.Process(target=worker, args=(r, world, port, (w, h, f, q, wavelet), use_oracle)) for r in range(world)]
    [p.start() for p in procs]
    [p.join() for p in procs]
    sys.exit(max(abs(p.exitcode or 0) for p in procs))
