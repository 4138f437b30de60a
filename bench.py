#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s encode+decode, 1920x1080x64 chunks, CDF 9/7, q=80, bit-exact vs CPU.

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM: every rank encodes `--chunks` independent 1080p x 64-frame chunks (RGB -> .alc, on the device),
the finished .alc blobs are gathered on rank 0 (RCCL, only when N > 1), and every rank decodes its
own chunks back to RGB.  Chunks are independent bitstreams, so ranks share no data-path collective
other than that gather ("scaling": "weak": --chunks per GPU is fixed as N grows).

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

# The contract is ONE JSON line on stdout.  Native libraries (RCCL prints a version banner) write to
# file descriptor 1 directly, so keep a private copy of stdout for the result and point fd 1 at stderr.
_RESULT_OUT = os.fdopen(os.dup(1), "w")
os.dup2(2, 1)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import alice_codec_amd as ac  # noqa: E402
from alice_codec_amd import multi  # noqa: E402

W, H, F = 1920, 1080, 64
QUALITY = 80
WAVELET = ac.WaveletType.Cdf97
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)


def synth_chunk(dev, idx: int) -> torch.Tensor:
    """S-smooth: moving sinusoids + integer noise in [-4, 4], generated on the device."""
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


def cpu_baseline(sample_rgb: np.ndarray, frames: int, gpu_alc: bytes, gpu_dec: np.ndarray) -> dict:
    """Times the oracle (1 thread, -O3 -march=native built on this host) on the sample and checks the GPU
    results for the same sample byte for byte."""
    import oracle
    so = None
    try:
        so = os.path.join(tempfile.mkdtemp(prefix="alice_oracle_"), "liboracle_native.so")
        oracle.build(force=True, so_path=so, cflags="-O3 -march=native -fPIC -std=c11")
        lib = oracle.lib(so)
    except Exception:
        lib = oracle.lib()
    t0 = time.perf_counter()
    alc = oracle.encode(sample_rgb, W, H, frames, QUALITY, int(WAVELET), _lib=lib)
    t1 = time.perf_counter()
    dec = oracle.decode(alc, _lib=lib)
    t2 = time.perf_counter()
    # non-reference variant (SURVEY.md section 8d ii): Y, Co, Cg on three threads, what a rayon::join would give
    par3 = None
    try:
        t3 = time.perf_counter()
        alc3 = oracle.encode(sample_rgb, W, H, frames, QUALITY, int(WAVELET), _lib=lib, three_threads=True)
        t4 = time.perf_counter()
        dec3 = oracle.decode(alc3, _lib=lib, three_threads=True)
        t5 = time.perf_counter()
        par3 = {"value": round(2 * W * H * frames / (t5 - t3) / 1e6, 3), "unit": "Mpix/s", "cores": 3,
                "note": "NOT the reference, which is single-threaded: the same port with the three channels on three threads",
                "same_bytes": bool(alc3 == alc and np.array_equal(dec3, dec))}
    except Exception:
        pass
    px = W * H * frames
    try:
        model = subprocess.check_output("lscpu | grep 'Model name' | head -1", shell=True, text=True).split(":", 1)[1].strip()
    except Exception:
        model = "unknown"
    return {
        "value": round(2 * px / (t2 - t0) / 1e6, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": f"{W}x{H}x{frames} (first {frames} frames of chunk 0), {WAVELET.name} q={QUALITY}, encode {t1 - t0:.2f}s + decode {t2 - t1:.2f}s",
        "encode_mpix_s": round(px / (t1 - t0) / 1e6, 3), "decode_mpix_s": round(px / (t2 - t1) / 1e6, 3),
        "host_cpu": model, "host_threads_available": os.cpu_count(),
        "gpu_bit_exact_on_sample": bool(gpu_alc == alc and np.array_equal(gpu_dec, dec)),
        "three_thread_variant": par3,
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=int(os.environ.get("ALICE_BENCH_CHUNKS", "152")),
                    help="1080p x 64 chunks in flight per GPU (3 rANS chains each)")
    ap.add_argument("--wavelet", choices=["cdf97", "cdf53", "haar"], default="cdf97",
                    help="cdf97 is the BASELINE metric; cdf53 = BASELINE.json configs[1] (the roofline fields then describe that run)")
    ap.add_argument("--quality", type=int, default=80)
    ap.add_argument("--separate-output", action="store_true",
                    help="decode into a caller-owned RGB buffer instead of the batch's own storage (one more RGB-sized buffer per chunk)")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of chunk 0 in the CPU baseline sample (0 = skip)")
    args = ap.parse_args()

    global QUALITY, WAVELET
    QUALITY = args.quality
    WAVELET = {"cdf97": ac.WaveletType.Cdf97, "cdf53": ac.WaveletType.Cdf53, "haar": ac.WaveletType.Haar}[args.wavelet]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
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

    B = args.chunks
    px_chunk = W * H * F
    # rehearsal switch: hold this many extra GB, e.g. the blobs rank 0 of an 8-GPU run would receive, to check the fit
    ballast = None
    if os.environ.get("ALICE_BENCH_BALLAST_GB"):
        ballast = torch.empty(int(float(os.environ["ALICE_BENCH_BALLAST_GB"]) * 1e9), dtype=torch.uint8, device=dev)
    rgb = torch.empty((B, F, H, W, 3), dtype=torch.uint8, device=dev)
    for i in range(B):
        rgb[i] = synth_chunk(dev, rank * B + i)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()   # the generator's temporaries: rank 0 of an 8-GPU run needs the room for the gathered blobs
    out = torch.empty_like(rgb) if args.separate_output else None
    batch = ac.Batch(W, H, F, B, QUALITY, WAVELET)
    stream = torch.cuda.current_stream().cuda_stream
    packed = None
    blob = None
    stage_acc = {}
    n_acc = 0

    def step(record: bool):
        nonlocal packed, blob, n_acc
        batch.encode(rgb.data_ptr(), stream)
        sizes = batch.encode_finish()
        pending = None
        if use_dist:
            # the gather of the finished .alc blobs on rank 0 runs on RCCL's stream beside the decode
            # (the root packs its own chunks straight into its slice of the gathered blob)
            if rank == 0:
                pending = multi.gather_alc_start(None, torch.from_numpy(sizes.astype(np.int64)), blob=blob, device=dev,
                                                 pack_fn=lambda dst: batch.pack_alc(sizes, dst.data_ptr(), dst.numel(), stream))
            else:
                if packed is None:
                    packed = torch.empty(int(sizes.sum()) + 4096, dtype=torch.uint8, device=dev)
                batch.pack_alc(sizes, packed.data_ptr(), packed.numel(), stream)
                pending = multi.gather_alc_start(packed, torch.from_numpy(sizes.astype(np.int64)), blob=blob)
        batch.decode(batch.alc_ptr(0), batch.alc_stride, out.data_ptr() if out is not None else None, stream)
        batch.decode_finish()
        if pending is not None:
            res = pending.wait()
            if res is not None:
                blob = res[0]
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
    print(f"[bench] rank {rank}: HBM free {free_b / 1e9:.1f} GB of {total_b / 1e9:.1f} GB after the timed steps", file=sys.stderr)
    if rank == 0:
        ms = {k: v / max(n_acc, 1) for k, v in stage_acc.items()}  # per step, whole batch of this rank
        value = 2.0 * args.steps * world * B * px_chunk / elapsed / 1e6
        fwd_s = ms["forward_transform"] / 1e3 / B   # one chunk = one launch pair (fwd_xy + fwd_t)
        inv_s = ms["inverse_transform"] / 1e3 / B
        enc_chain_s = ms["rans_encode"] / 1e3
        dec_chain_s = ms["rans_decode"] / 1e3
        payload_bpp = float(sizes.sum() - 3138 * B) / (B * px_chunk)
        result = {
            "metric": ("Mpixels/s encode+decode, 1080p x 64 CDF9/7 q=80; bit-exact vs CPU" if (args.wavelet, QUALITY) == ("cdf97", 80)
                       else f"Mpixels/s encode+decode, 1080p x 64 {args.wavelet} q={QUALITY}; bit-exact vs CPU"),
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": ("1920x1080x64 RGB chunks, CDF 9/7, q=80 (BASELINE.json configs[2]); " if (args.wavelet, QUALITY) == ("cdf97", 80)
                                    else f"1920x1080x64 RGB chunks, {args.wavelet}, q={QUALITY} (not the headline configuration); ")
                                   + f"{B} independent chunks in flight per GPU = {3 * B} single-stream rANS chains",
                       "chunks_per_gpu": B, "wavelet": args.wavelet, "quality": QUALITY,
                       "parallelism": f"chunk-parallel x{world}" + (" + RCCL gather of .alc blobs on rank 0" if world > 1 else "")},
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
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                tj = json.load(open(traffic_file))
                result["roofline"]["traffic"] = tj.get("forward_transform_hbm_bytes_per_launch")
                result["roofline_inverse"]["traffic"] = tj.get("inverse_transform_hbm_bytes_per_launch")
            except Exception:
                pass
        # The transform kernels are bound by integer VALU issue, not by HBM: report the
        # VALU view beside the HBM one.  Lane-operations per pixel are PMC counts (SQ_INSTS_VALU x 64,
        # profiles/sq_counters.json); peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz.
        sq_file = os.path.join(ROOT, "profiles", "sq_counters.json")
        if os.path.exists(sq_file):
            try:
                pk = json.load(open(sq_file))["per_kernel"]
                peak = 256 * 4 * 16 * 2.4e9
                for key, pref, secs in (("roofline", "fwd_", fwd_s), ("roofline_inverse", "inv_", inv_s)):
                    ops = sum(v.get("valu_lane_ops_per_pixel", 0.0) for k, v in pk.items() if pref in k)
                    if ops > 0:
                        result[key]["valu"] = {"lane_ops_per_pixel": round(ops, 1), "achieved_tera_lane_ops": round(ops * px_chunk / secs / 1e12, 2),
                                               "peak_tera_lane_ops": round(peak / 1e12, 2), "frac": round(ops * px_chunk / secs / peak, 4)}
            except Exception:
                pass
        if world == 1 and args.cpu_frames > 0:
            fr = min(args.cpu_frames, F)
            sample = rgb[0, :fr].contiguous().cpu().numpy().reshape(-1)
            chunk = ac.FrameEncoder.with_wavelet(QUALITY, WAVELET).encode(sample, W, H, fr)
            result["cpu_baseline"] = cpu_baseline(sample, fr, chunk.to_bytes(), ac.FrameDecoder().decode(chunk))
        _RESULT_OUT.write(json.dumps(result) + "\n")
        _RESULT_OUT.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
