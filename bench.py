#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s encode+decode, 1920x1080x64 chunks, CDF 9/7, q=80, bit-exact vs CPU.

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM: every rank encodes `--chunks` independent 1080p x 64-frame chunks (RGB -> .alc, on the device),
the finished .alc blobs stream to rank 0 (RCCL point to point, only when N > 1: through a small receive
ring and on to pinned host memory, so rank 0's HBM never holds them all), and every rank decodes its
own chunks back to RGB.  Chunks are independent bitstreams, so ranks share no data-path collective
("scaling": "weak": chunks per GPU do not change with N).  `--chunks auto` (the default) sizes the
batch from the free HBM: one rANS chain is one wavefront, and what bounds the chains in flight is memory.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = 2 * pixels / time (each pixel is encoded once and decoded
once per step), whole job.  `roofline` describes the HBM-bound kernels of the path (the forward
transform, RGB -> u8 symbols, 6 algorithmic bytes per pixel); the serial single-stream rANS chains
that dominate wall time are latency-bound and are reported separately under `entropy_chain`.
`cpu_baseline` is the CPU oracle (a scalar port of the single-threaded reference; the Rust crate
cannot be built here) timed on this host on a bounded sample of the same input."""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Nothing here touches the GPU or the process's file descriptors at import time: scripts import this module for
# W / H / F and synth_chunk(); torch and the codec are imported lazily (main() and the helpers that need them).
_RESULT_OUT = None   # main() points this at a private copy of stdout (see _claim_stdout)


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries (RCCL prints a version banner) write to file
    descriptor 1 directly, so keep a private copy of stdout for the result and point fd 1 at stderr."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _RESULT_OUT


W, H, F = 1920, 1080, 64
QUALITY = 80
WAVELET = None      # ac.WaveletType, set by _mods() / main()
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)


def _mods():
    """Import numpy / torch / the codec into this module's globals (first GPU-side use; never at import time)."""
    global np, torch, dist, ac, multi, WAVELET
    import numpy as np  # noqa: F811
    import torch  # noqa: F811
    import torch.distributed as dist  # noqa: F811
    import alice_codec_amd as ac  # noqa: F811
    from alice_codec_amd import multi  # noqa: F811
    if WAVELET is None:
        WAVELET = ac.WaveletType.Cdf97


def synth_chunk(dev, idx: int):
    """S-smooth: moving sinusoids + integer noise in [-4, 4], generated on the device."""
    _mods()
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + idx)
    t = torch.arange(F, device=dev, dtype=torch.float32).view(F, 1, 1, 1)
    y = torch.arange(H, device=dev, dtype=torch.float32).view(1, H, 1, 1)
    x = torch.arange(W, device=dev, dtype=torch.float32).view(1, 1, W, 1)
    s = torch.tensor([23.0, 31.0, 17.0], device=dev).view(1, 1, 1, 3)
    ph = torch.tensor([0.0, 1.0, 2.0], device=dev).view(1, 1, 1, 3)
    base = 128 + 90 * torch.sin((x + 2 * t + 5 * idx) / s + ph) * torch.cos((y - t) / (0.7 * s))
    noise = torch.randint(-4, 5, (F, H, W, 3), device=dev, generator=g)
    return (base + noise).round().clamp(0, 255).to(torch.uint8)


def native_oracle():
    """The oracle rebuilt -O3 -march=native on this host; falls back to the generic build (and says so)."""
    import oracle
    try:
        so = os.path.join(tempfile.mkdtemp(prefix="alice_oracle_"), "liboracle_native.so")
        flags = "-O3 -march=native -fPIC -std=c11"
        oracle.build(force=True, so_path=so, cflags=flags)
        return oracle.lib(so), "gcc " + flags
    except Exception as e:  # noqa: BLE001
        print(f"[bench] native oracle build failed ({e}); timing the generic -O3 build", file=sys.stderr)
        return oracle.lib(), "gcc -O3 -fPIC -std=c11 (generic x86-64: the -march=native build failed)"


def verify_and_baseline(first, last, frames: int) -> dict:
    """first / last = (rgb, gpu_alc_bytes, gpu_decoded) of chunk 0 and chunk B-1 of the batch the timed steps produced.
    Chunk 0: the oracle, 1 thread, timed = cpu_baseline.  Chunk B-1: the oracle with Y, Co, Cg on three threads (same
    bytes; a non-reference variant, reported as such).  Both compared byte for byte with what the GPU batch holds."""
    import oracle
    lib, build = native_oracle()
    rgb0, alc0, dec0 = first
    t0 = time.perf_counter()
    alc = oracle.encode(rgb0, W, H, frames, QUALITY, int(WAVELET), _lib=lib)
    t1 = time.perf_counter()
    dec = oracle.decode(alc, _lib=lib)
    t2 = time.perf_counter()
    ok0 = bool(alc0 == alc and np.array_equal(dec0, dec))
    del dec
    par3, ok1 = None, None
    if last is not None:
        rgb1, alc1, dec1 = last
        t3 = time.perf_counter()
        alc3 = oracle.encode(rgb1, W, H, frames, QUALITY, int(WAVELET), _lib=lib, three_threads=True)
        t4 = time.perf_counter()
        dec3 = oracle.decode(alc3, _lib=lib, three_threads=True)
        t5 = time.perf_counter()
        ok1 = bool(alc1 == alc3 and np.array_equal(dec1, dec3))
        par3 = {"value": round(2 * W * H * frames / (t5 - t3) / 1e6, 3), "unit": "Mpix/s", "cores": 3,
                "note": "NOT the reference, which is single-threaded: the same port with the three channels on three threads "
                        "(timed on the last chunk of the batch, which it also checks)"}
    px = W * H * frames
    try:
        model = subprocess.check_output("lscpu | grep 'Model name' | head -1", shell=True, text=True).split(":", 1)[1].strip()
    except Exception:  # noqa: BLE001
        model = "unknown"
    return {
        "value": round(2 * px / (t2 - t0) / 1e6, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": f"{W}x{H}x{frames} (chunk 0 of the timed batch, all {frames} frames), {WAVELET.name} q={QUALITY}, encode {t1 - t0:.2f}s + decode {t2 - t1:.2f}s",
        "encode_mpix_s": round(px / (t1 - t0) / 1e6, 3), "decode_mpix_s": round(px / (t2 - t1) / 1e6, 3),
        "oracle_build": build, "host_cpu": model, "host_threads_available": os.cpu_count(),
        "gpu_batch_chunk0_bit_exact": ok0, "gpu_batch_last_chunk_bit_exact": ok1,
        "three_thread_variant": par3,
    }


def host_api_measure(threads=(1, 8, 64), wavelet=None, quality=None) -> dict:
    """The drop-in path as a reference caller would use it (the handles are Send + Sync, src/pipeline.rs:635-644, so
    the natural move is a thread pool over chunks): T host threads share one FrameEncoder and one FrameDecoder, each
    calls alice_codec_encode64 / alice_codec_decode64 on its OWN 1920x1080x64 chunk held in pageable host memory --
    host -> device copy, transforms, the chunk's three serial chains, device -> host copy, all inside the call.
    Aggregate Mpix/s per phase (all threads encode, then all decode), PCIe included.  Never part of `value`."""
    _mods()
    import threading
    wavelet = WAVELET if wavelet is None else wavelet
    quality = QUALITY if quality is None else quality
    dev = torch.device("cuda", torch.cuda.current_device())
    enc = ac.FrameEncoder.with_wavelet(quality, wavelet)
    dec = ac.FrameDecoder()
    px = W * H * F
    tmax = max(threads)
    host = [synth_chunk(dev, 5000 + i).reshape(-1).cpu().numpy() for i in range(tmax)]   # pageable
    torch.cuda.empty_cache()
    rows = []
    warm = enc.encode(host[0], W, H, F)       # library load, pool, code objects
    dec.decode(warm)
    del warm
    for T in threads:
        chunks, outs = [None] * T, [None] * T
        gate = threading.Barrier(T + 1)

        def run(fn):
            def body(i):
                ac.set_device(dev.index or 0)
                gate.wait()
                fn(i)
            th = [threading.Thread(target=body, args=(i,)) for i in range(T)]
            [t.start() for t in th]
            gate.wait()
            t0 = time.perf_counter()
            [t.join() for t in th]
            return time.perf_counter() - t0

        def do_enc(i): chunks[i] = enc.encode(host[i], W, H, F)
        def do_dec(i): outs[i] = dec.decode(chunks[i])
        t_enc = run(do_enc)
        t_dec = run(do_dec)
        ok = all(o is not None and o.size == px * 3 for o in outs)
        rows.append({"threads": T, "encode_s": round(t_enc, 3), "decode_s": round(t_dec, 3),
                     "encode_mpix_s": round(T * px / t_enc / 1e6, 2), "decode_mpix_s": round(T * px / t_dec / 1e6, 2),
                     "encode_plus_decode_mpix_s": round(2 * T * px / (t_enc + t_dec) / 1e6, 2), "all_calls_succeeded": bool(ok),
                     "alc_bytes_chunk0": len(chunks[0].to_bytes())})
        print(f"[bench] host api, {T} thread(s): encode {rows[-1]['encode_mpix_s']} Mpix/s, decode {rows[-1]['decode_mpix_s']} Mpix/s", file=sys.stderr)
        del chunks, outs
        ac.load_library().alice_codec_trim()
    return {"what": "T host threads, one shared FrameEncoder / FrameDecoder handle, alice_codec_encode64 / alice_codec_decode64 per thread on its own "
                    f"{W}x{H}x{F} chunk in pageable host memory ({WAVELET.name if wavelet is None else ac.WaveletType(wavelet).name} q={quality}); aggregate rate per phase, "
                    "PCIe and the three serial chains of every chunk inside the call; not part of `value`",
            "rows": rows}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks", default=os.environ.get("ALICE_BENCH_CHUNKS", "auto"),
                    help="1080p x 64 chunks in flight per GPU (3 rANS chains each); auto = what the free HBM holds, at most 341 "
                         "(1023 chains: one per SIMD)")
    ap.add_argument("--wavelet", choices=["cdf97", "cdf53", "haar"], default="cdf97",
                    help="cdf97 is the BASELINE metric; cdf53 = BASELINE.json configs[1] (the roofline fields then describe that run)")
    ap.add_argument("--quality", type=int, default=80)
    ap.add_argument("--separate-output", action="store_true",
                    help="decode into a caller-owned RGB buffer instead of the batch's own storage (one more RGB-sized buffer per chunk)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the oracle check of the batch and the CPU baseline (the line then does not say bit-exact)")
    ap.add_argument("--gather-sink", choices=["device", "host"], default="device",
                    help="N > 1: what rank 0 does with the .alc bytes the gather delivers: 'device' adds them to a checksum on the "
                         "GPU (outputs stay in HBM, as the inputs are: no PCIe inside `value`); 'host' also copies every chunk "
                         "on to pinned host memory (8 GPUs x 337 chunks x 0.11 GB = 296 GB per step over rank 0's one PCIe link)")
    ap.add_argument("--no-host-api", action="store_true",
                    help="skip the drop-in host-call measurement (T host threads through alice_codec_encode64 / decode64)")
    return ap.parse_args(argv)


# ---- N ranks from one command ------------------------------------------------------------------------------------
# `python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset) starts the N ranks ITSELF, as child processes,
# before anything in this process has touched the GPU; under torchrun (WORLD_SIZE set) the process is one rank.  Any
# disagreement between --gpus and the environment is an error, never a silent single-GPU number.

def plan_ranks(n_gpus: int, argv: list, env: dict, port: int, python: str = sys.executable) -> list:
    """The child processes of a self-spawned N-rank run: [(command line, environment)], rank i on GPU i.  Pure."""
    plan = []
    for r in range(n_gpus):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_gpus), "LOCAL_WORLD_SIZE": str(n_gpus),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "ALICE_BENCH_SPAWNED": "1"})
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        plan.append(([python, os.path.abspath(__file__)] + list(argv), e))
    return plan


def check_launch(n_gpus: int, env: dict, visible_devices) -> str:
    """What this invocation is: "rank" (run as one rank of the world the environment describes), "spawn" (start
    n_gpus ranks), or an error message (then the process must exit non-zero).  Pure; visible_devices may be None
    (not counted yet)."""
    ws = env.get("WORLD_SIZE")
    if n_gpus < 1:
        return f"error: --gpus {n_gpus}"
    if ws is not None:
        if int(ws) != n_gpus:
            return (f"error: --gpus {n_gpus} but the launcher set WORLD_SIZE={ws}: refusing to report a "
                    f"{ws}-rank number as n_gpus={n_gpus}")
        return "rank"
    if n_gpus == 1:
        return "rank"
    if visible_devices is not None and visible_devices < n_gpus:
        return (f"error: --gpus {n_gpus} but only {visible_devices} GPU(s) are visible on this node: not running a "
                f"smaller world under that label")
    return "spawn"


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv) -> int:
    """Start the N ranks, relay rank 0's JSON line (the children inherit this process's stdout; only rank 0 writes to
    it), wait for all, return the first non-zero exit code (the other ranks are then terminated by PID)."""
    plan = plan_ranks(args.gpus, argv, dict(os.environ), _free_port())
    if os.environ.get("ALICE_BENCH_SPAWN_DRYRUN") == "1":   # tests: show the plan, start nothing
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")
        print(json.dumps([{"cmd": c, "env": {k: e.get(k) for k in keys}} for c, e in plan]))
        return 0
    procs = [subprocess.Popen(c, env=e) for c, e in plan]
    print(f"[bench] started {len(procs)} ranks: pids {[p.pid for p in procs]}", file=sys.stderr)
    rc = 0
    live = set(range(len(procs)))
    while live:
        for i in sorted(live):
            r = procs[i].poll()
            if r is None:
                continue
            live.discard(i)
            if r != 0 and rc == 0:
                rc = r
                print(f"[bench] rank {i} exited with {r}: stopping the other ranks", file=sys.stderr)
                for j in live:
                    procs[j].terminate()
        time.sleep(0.2)
    return rc


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    n_visible = None
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and os.environ.get("ALICE_BENCH_SPAWN_DRYRUN") != "1":
        import torch as _t
        n_visible = _t.cuda.device_count()      # counting devices does not initialise the GPU on this image
    what = check_launch(args.gpus, os.environ, n_visible)
    if what.startswith("error"):
        raise SystemExit("bench.py: " + what)
    if what == "spawn":
        raise SystemExit(spawn_ranks(args, argv))

    _claim_stdout()
    _mods()
    global QUALITY, WAVELET
    QUALITY = args.quality
    WAVELET = {"cdf97": ac.WaveletType.Cdf97, "cdf53": ac.WaveletType.Cdf53, "haar": ac.WaveletType.Haar}[args.wavelet]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ac.set_device(local)
    # ALICE_BENCH_FORCE_DIST=1 exercises the distributed code path with a single rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("ALICE_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus and not os.environ.get("ALICE_BENCH_FORCE_DIST"):
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
    # who is in the job, as the ranks themselves see it (answers "did RCCL see N ranks" from the record)
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local, "device": torch.cuda.get_device_name(dev), "pci_bus_id": getattr(props, "pci_bus_id", None),
          "hbm_gb": round(props.total_memory / 1e9, 1)}
    if use_dist:
        ranks_seen = [None] * dist.get_world_size()
        dist.all_gather_object(ranks_seen, me)
    else:
        ranks_seen = [me]

    px_chunk = W * H * F
    stream = torch.cuda.current_stream().cuda_stream
    chunk0 = synth_chunk(dev, rank * 1000)
    # ---- how many chunks: a trial batch of one chunk tells what the library holds per chunk (.alc capacities from this
    # content's histograms) and whatever the count (transform scratch); the inputs add 3 bytes per pixel
    sizing = {}
    if args.chunks == "auto":
        trial = ac.Batch(W, H, F, 1, QUALITY, WAVELET)
        trial.encode(chunk0.data_ptr(), stream)
        trial.encode_finish()
        per_chunk = px_chunk * 3 + trial.bytes_per_chunk + (px_chunk * 3 if args.separate_output else 0)
        fixed = trial.fixed_bytes
        alc_stride = trial.alc_stride
        del trial
        ac.load_library().alice_codec_trim()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        free_b, total_b = torch.cuda.mem_get_info(dev)
        if use_dist:
            fixed += 2 * max(world - 1, 1) * alc_stride                        # the root's receive ring
            # RCCL creates its point-to-point channels (buffers per peer) at the first send / receive, i.e. AFTER the batch has
            # taken its memory, and ranks cannot drop chunks one by one afterwards: keep 3 GiB back on every rank (three
            # chunks of 341) rather than lose a multi-GPU run to an allocation failure inside the library
            fixed += 3 << 30
        # (the capacities grow a little when another chunk's Y stream is longer than chunk 0's: 0.2 % of the .alc buffers kept back)
        margin = int(0.002 * 341 * alc_stride) + (64 << 20)
        B = int((free_b - fixed - margin) // per_chunk)
        B = max(1, min(B, 341))
        if use_dist:
            tb = torch.tensor([B], dtype=torch.int64, device=dev)
            dist.all_reduce(tb, op=dist.ReduceOp.MIN)
            B = int(tb.item())
        sizing = {"chunks": "auto", "free_hbm_gb_at_start": round(free_b / 1e9, 2), "bytes_per_chunk_in_flight": int(per_chunk),
                  "alc_buffer_bytes_per_chunk": int(alc_stride), "fixed_bytes": int(fixed), "margin_bytes": int(margin)}
    else:
        B = int(args.chunks)
        sizing = {"chunks": "fixed on the command line"}
    # rehearsal switch: hold this many extra GB to check a fit
    ballast = None
    if os.environ.get("ALICE_BENCH_BALLAST_GB"):
        ballast = torch.empty(int(float(os.environ["ALICE_BENCH_BALLAST_GB"]) * 1e9), dtype=torch.uint8, device=dev)
    rgb = torch.empty((B, F, H, W, 3), dtype=torch.uint8, device=dev)
    rgb[0] = chunk0
    del chunk0
    for i in range(1, B):
        rgb[i] = synth_chunk(dev, rank * 1000 + i)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()   # the generator's temporaries
    out = torch.empty_like(rgb) if args.separate_output else None
    # the batch; if the device runs out of memory after all (the first encode allocates the .alc buffers), drop chunks
    # from the end until it fits: the dropped inputs stay allocated, every chunk less frees its symbols and .alc buffer
    batch = None
    dropped = 0
    while True:
        try:
            batch = ac.Batch(W, H, F, B, QUALITY, WAVELET)
            batch.encode(rgb.data_ptr(), stream)
            batch.encode_finish()
            break
        except ac.CodecError as e:
            if e.kind != "OutOfMemory" or B <= 1 or use_dist:
                raise
            batch = None
            ac.load_library().alice_codec_trim()
            torch.cuda.synchronize()
            B -= 1
            dropped += 1
            print(f"[bench] out of device memory with {B + 1} chunks: trying {B}", file=sys.stderr)
    sizing["chunks_dropped_after_out_of_memory"] = dropped
    stage_acc = {}
    n_acc = 0
    # N > 1: the .alc bytes of every rank stream to rank 0 beside the decode, on a side stream; rank 0 hands each chunk to
    # a sink that adds its bytes to a checksum on the device and, with --gather-sink host, copies it on to pinned host
    # memory (a ring: the bench keeps nothing)
    side = torch.cuda.Stream(device=dev) if use_dist else None
    gathered = {"bytes": 0, "checksum": None, "host_ring": None, "k": 0}

    def sink(r, i, t):
        n = t.numel()
        if gathered["checksum"] is None:
            gathered["checksum"] = torch.zeros((), dtype=torch.int64, device=dev)
            if args.gather_sink == "host":
                gathered["host_ring"] = [torch.empty(batch.alc_stride, dtype=torch.uint8).pin_memory() for _ in range(4)]
        gathered["checksum"] += t.sum(dtype=torch.int64)
        if gathered["host_ring"] is not None:
            gathered["host_ring"][gathered["k"] % 4][:n].copy_(t, non_blocking=True)
        gathered["k"] += 1
        gathered["bytes"] += n

    def step(record: bool):
        nonlocal n_acc
        batch.encode(rgb.data_ptr(), stream)
        sizes = batch.encode_finish()
        batch.decode(batch.alc_ptr(0), batch.alc_stride, out.data_ptr() if out is not None else None, stream)
        if use_dist:
            # (encode_finish has synchronised: the .alc buffers are complete; the side stream need not wait for the
            # decode kernels just queued, which only read them)
            with torch.cuda.stream(side):
                stride = batch.alc_stride
                multi.stream_alc_to_root(lambda i: multi.DeviceView(batch.alc_ptr(i), stride).tensor(dev),
                                         torch.from_numpy(sizes.astype(np.int64)), sink, dst=0)
            side.synchronize()
        batch.decode_finish()
        if record:
            for k, v in batch.stage_ms().items():
                stage_acc[k] = stage_acc.get(k, 0.0) + v
            n_acc += 1
        return sizes

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    sizes = None
    for _ in range(args.warmup):
        sizes = step(False)
    fence()
    gathered["bytes"] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sizes = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    free_b, total_b = torch.cuda.mem_get_info(dev)
    print(f"[bench] rank {rank}: {B} chunks, HBM free {free_b / 1e9:.1f} GB of {total_b / 1e9:.1f} GB "
          f"({100.0 * free_b / total_b:.1f} %) after the timed steps", file=sys.stderr)

    # ---- what the timed steps left in the batch, against the oracle: first and last chunk, all 64 frames, every rank
    verdict = None
    bit_exact = None
    if not args.no_verify:
        def fetch(i):
            alc = multi.DeviceView(batch.alc_ptr(i), int(sizes[i])).tensor(dev).cpu().numpy().tobytes()
            if out is not None:
                dec = out[i].reshape(-1).cpu().numpy()
            else:
                dec = multi.DeviceView(batch.rgb_ptr(i), px_chunk * 3).tensor(dev).cpu().numpy().copy()
            return rgb[i].reshape(-1).cpu().numpy(), alc, dec
        verdict = verify_and_baseline(fetch(0), fetch(B - 1) if B > 1 else None, F)
        bit_exact = bool(verdict["gpu_batch_chunk0_bit_exact"] and verdict["gpu_batch_last_chunk_bit_exact"] is not False)
        if use_dist:
            tb = torch.tensor([1 if bit_exact else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(tb, op=dist.ReduceOp.MIN)
            bit_exact = bool(tb.item())
    # ---- the drop-in host calls, measured the way a caller of the reference would use them (rank 0 of a one-GPU run
    # only; the batch goes first: the calls need the device memory)
    host_api = None
    if rank == 0 and world == 1 and not args.no_host_api:
        chunks_per_gpu_used = B
        del batch
        del rgb
        out = None
        ac.load_library().alice_codec_trim()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        try:
            host_api = host_api_measure()
        except Exception as e:  # noqa: BLE001
            host_api = {"error": repr(e)}
    if rank == 0:
        ms = {k: v / max(n_acc, 1) for k, v in stage_acc.items()}  # per step, whole batch of this rank
        value = 2.0 * args.steps * world * B * px_chunk / elapsed / 1e6
        fwd_s = ms["forward_transform"] / 1e3 / B   # one chunk = one launch pair (fwd_xy + fwd_t)
        inv_s = ms["inverse_transform"] / 1e3 / B
        enc_chain_s = ms["rans_encode"] / 1e3
        dec_chain_s = ms["rans_decode"] / 1e3
        payload_bpp = float(sizes.sum() - 3138 * B) / (B * px_chunk)
        headline = (args.wavelet, QUALITY) == ("cdf97", 80)
        tail = "; bit-exact vs CPU" if bit_exact else ("; NOT bit-exact vs CPU" if bit_exact is False else "; not checked against the CPU in this run")
        result = {
            "metric": ("Mpixels/s encode+decode, 1080p x 64 CDF9/7 q=80" if headline
                       else f"Mpixels/s encode+decode, 1080p x 64 {args.wavelet} q={QUALITY}") + tail,
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": ("1920x1080x64 RGB chunks, CDF 9/7, q=80 (BASELINE.json configs[2]); " if headline
                                    else f"1920x1080x64 RGB chunks, {args.wavelet}, q={QUALITY} (not the headline configuration); ")
                                   + f"{B} independent chunks in flight per GPU = {3 * B} single-stream rANS chains, one wavefront each "
                                     "(a GPU has 1024 SIMDs; HBM bounds the count)",
                       "chunks_per_gpu": B, "wavelet": args.wavelet, "quality": QUALITY, "sizing": sizing,
                       "hbm_free_after_timed_steps_gb": round(free_b / 1e9, 1), "hbm_total_gb": round(total_b / 1e9, 1),
                       "parallelism": f"chunk-parallel x{world}" + (" + RCCL point-to-point stream of the .alc blobs to rank 0 (receive ring -> pinned host)" if world > 1 else "")},
            "ranks": {"world_size_seen_by_process_group": dist.get_world_size() if use_dist else 1, "devices": ranks_seen,
                      "launched_by": "bench.py itself (child processes)" if os.environ.get("ALICE_BENCH_SPAWNED") else
                                     ("a launcher (WORLD_SIZE was set)" if "WORLD_SIZE" in os.environ else "single process")},
            "batch_bit_exact": bit_exact,
            "encode_mpix_s": round(world * B * px_chunk / ((ms["forward_transform"] + ms["rans_table"] + ms["rans_encode"] + ms["assemble"]) / 1e3) / 1e6, 2),
            "decode_mpix_s": round(world * B * px_chunk / ((ms["rans_decode"] + ms["inverse_transform"]) / 1e3) / 1e6, 2),
            "payload_bytes_per_pixel": round(payload_bpp, 4),
            "stage_ms_per_step": {k: round(v, 3) for k, v in ms.items()},
            "roofline": {
                "kernel": "forward transform (fwd_xy_kernel + fwd_t_kernel, one launch pair per chunk)",
                "bound": "hbm", "achieved": round(6.0 * px_chunk / fwd_s / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(6.0 * px_chunk / fwd_s / HBM_PEAK, 4), "traffic": None,
                "algorithmic_bytes_per_launch": 6 * px_chunk, "avg_launch_ms": round(fwd_s * 1e3, 4),
            },
            "roofline_inverse": {
                "kernel": "inverse transform (inv_t_kernel + inv_xy_kernel)", "bound": "hbm",
                "achieved": round(6.0 * px_chunk / inv_s / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(6.0 * px_chunk / inv_s / HBM_PEAK, 4), "traffic": None, "avg_launch_ms": round(inv_s * 1e3, 4),
            },
            "entropy_chain": {
                "note": "single-stream rANS is one serial dependency chain per channel (latency-bound, not HBM-bound); "
                        "throughput scales with chains in flight, not with bandwidth",
                "chains_in_flight_per_gpu": 3 * B, "symbols_per_chain": px_chunk,
                "encode_msym_s_per_chain": round(px_chunk / enc_chain_s / 1e6, 2),
                "decode_msym_s_per_chain": round(px_chunk / dec_chain_s / 1e6, 2),
                "encode_ns_per_symbol": round(enc_chain_s / px_chunk * 1e9, 2),
                "decode_ns_per_symbol": round(dec_chain_s / px_chunk * 1e9, 2),
            },
        }
        if use_dist:
            result["gather"] = {"sink_on_rank0": "device checksum" if args.gather_sink == "device" else "device checksum + copy to pinned host memory",
                                "bytes_streamed_to_rank0_per_step": int(gathered["bytes"] // max(args.steps, 1)),
                                "byte_checksum_all_steps": int(gathered["checksum"].item()) if gathered["checksum"] is not None else 0}
        # PMC-derived fields describe the CDF 9/7 q=80 profile run; they are attached to that configuration only
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if headline and os.path.exists(traffic_file):
            try:
                tj = json.load(open(traffic_file))
                result["roofline"]["traffic"] = tj.get("forward_transform_hbm_bytes_per_launch")
                result["roofline_inverse"]["traffic"] = tj.get("inverse_transform_hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                pass
        if verdict is not None:
            result["cpu_baseline"] = verdict
        if host_api is not None:
            result["host_api"] = host_api
        _RESULT_OUT.write(json.dumps(result) + "\n")
        _RESULT_OUT.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if bit_exact is False:
        raise SystemExit("bench: the batch the timed steps produced differs from the CPU oracle")


if __name__ == "__main__":
    main()
