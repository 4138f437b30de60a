"""Developer soak on a GPU box (not part of the suite): for a time budget, seeded random cases through the C ABI against the
oracle -- whole chunks of random shape / wavelet / quality / content with random band plans and value-table radii, `.alc`
blobs with corrupted payloads and headers (both sides must agree on the decoded bytes or both must refuse), random call
sequences on the stateful rANS coders, batches of random size.  Prints every mismatch with the seed that reproduces it.

    python tests/tools/soak_gpu.py [seconds, default 300] [first seed, default 1] [threads, default 1]

With threads > 1 the cases run concurrently (whole chunks and coder sequences; the process-wide test tunings stay at their
defaults, batches are left out): the chain hub's merged launches under a random mix of callers."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import oracle as o  # noqa: E402
import alice_codec_amd as a  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_threads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = a.load_library()
bad = []
counts = {}


def content(rng, w, h, f):
    kind = int(rng.integers(0, 6))
    t, y, x = np.meshgrid(np.arange(f), np.arange(h), np.arange(w), indexing="ij")
    if kind == 0:
        v = rng.integers(0, 256, (f, h, w, 3))
    elif kind == 1:
        base = 128 + 100 * np.sin((x + 3 * t) / 11.0) * np.cos((y - 2 * t) / 5.0)
        v = np.stack([base, base * 0.7 + 30, 255 - base], -1) + rng.integers(-9, 10, (f, h, w, 3))
    elif kind == 2:
        s = ((x + y + t) & 1) == 1
        v = np.stack([np.where(s, 255, 0), np.full_like(x, 128), np.where(s, 0, 255)], -1)
    elif kind == 3:
        v = np.zeros((f, h, w, 3)) + int(rng.integers(0, 256))
    elif kind == 4:
        v = np.stack([(x * 255) // max(w - 1, 1), (y * 255) // max(h - 1, 1), (t * 255) // max(f - 1, 1)], -1)
    else:
        v = rng.integers(0, 2, (f, h, w, 3)) * 255
    return np.clip(v, 0, 255).astype(np.uint8).reshape(-1), kind


def note(kind, ok, what):
    counts[kind] = counts.get(kind, 0) + 1
    if not ok:
        bad.append(what)
        print("MISMATCH", what, flush=True)


def chunk_case(rng, seed):
    w = int(rng.choice([1, 2, 3, 5, 8, 17, 64, 95, 96, 97, 128, 130, 200, 255, 256, 260, 384, 511]))
    h = int(rng.choice([1, 2, 3, 6, 7, 31, 32, 33, 40, 41, 64, 75, 96, 121]))
    f = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 64, 65]))
    if w * h * f > 3_000_000:
        f = max(1, 3_000_000 // (w * h))
    k = int(rng.integers(0, 3))
    q = int(rng.choice([0, 1, 10, 30, 50, 75, 80, 85, 90, 95, 99, 100]))
    band = int(rng.choice([0, 64, 96, 200, 1024, 1024 * 1024]))
    radius = int(rng.choice([2048, 2048, 700, 64, 3, 1]))
    rgb, ck = content(rng, w, h, f)
    if n_threads == 1:          # (process-wide hooks: not while other threads are inside calls)
        lib.alice_codec_test_set_tuning(band)
        lib.alice_codec_test_set_value_table_radius(radius)
    else:
        band, radius = -1, -1
    tag = (seed, w, h, f, k, q, band, radius, ck)
    try:
        ref = o.encode(rgb, w, h, f, q, k)
    except o.OracleError:
        # the reference does not terminate on this input (a table frequency wrapped to 0, src/rans.rs:128-132): both refuse
        try:
            a.FrameEncoder.with_wavelet(q, a.WaveletType(k)).encode(rgb, w, h, f)
            refused = False
        except a.CodecError:
            refused = True
        note("diverges", refused, ("diverges",) + tag)
        return
    got = a.FrameEncoder.with_wavelet(q, a.WaveletType(k)).encode(rgb, w, h, f)
    note("encode", got.to_bytes() == ref, ("encode",) + tag)
    want = o.decode(ref)
    dec = a.FrameDecoder().decode(a.EncodedChunk.from_bytes(ref))
    note("decode", np.array_equal(dec, want), ("decode",) + tag)
    # corrupted blob: a few payload bytes, sometimes a header field
    blob = bytearray(ref)
    for _ in range(int(rng.integers(1, 5))):
        pos = int(rng.integers(18, len(blob))) if len(blob) > 18 else 0
        blob[pos] ^= int(rng.integers(1, 256))
    if rng.random() < 0.2 and len(blob) > 3200:
        c = int(rng.integers(0, 3))
        blob[18 + 1040 * c + 4:18 + 1040 * c + 8] = int(rng.choice([1, 2, 14, 300, 70000, 2 ** 31 - 1])).to_bytes(4, "little")
    try:
        want_c = o.decode(bytes(blob))
    except Exception:   # noqa: BLE001  the oracle refuses it
        want_c = None
    try:
        got_c = a.FrameDecoder().decode(a.EncodedChunk.from_bytes(bytes(blob)))
    except Exception:   # noqa: BLE001
        got_c = None
    same = (want_c is None and got_c is None) or (want_c is not None and got_c is not None and np.array_equal(want_c, got_c))
    note("corrupt", same, ("corrupt",) + tag + (want_c is None, got_c is None))


def coder_case(rng, seed):
    n_sym = int(rng.choice([2, 3, 17, 100, 256]))
    p = 1.0 / (1.0 + np.arange(n_sym)) ** float(rng.choice([0.3, 1.3, 3.0]))
    parts = [rng.choice(n_sym, size=int(rng.choice([1, 2, 63, 64, 65, 1000, 4095, 4096, 4097, 9000, 20000])), p=p / p.sum()).astype(np.uint8)
             for _ in range(int(rng.integers(1, 5)))]
    hist = np.bincount(np.concatenate(parts), minlength=n_sym).astype(np.uint32)
    tg, to = a.FrequencyTable.from_histogram(hist), o.FrequencyTable(hist)
    present = np.nonzero(hist)[0]
    if (np.asarray(to.freq)[present] == 0).any():
        # a symbol that occurs got frequency 0 (the wrapping cast of src/rans.rs:128-132): the reference divides by zero
        try:
            e = a.RansEncoder()
            for part in parts:
                e.encode_symbols(part, tg)
            refused = False
        except a.CodecError:
            refused = True
        note("coder-diverges", refused, ("coder-diverges", seed, n_sym))
        return
    eg, eo = a.RansEncoder(), o.RansEncoder()
    ok = True
    for part in parts:
        if rng.random() < 0.3 and len(part) < 200:
            for s in part[::-1]:
                eg.encode(tg.get_symbol(int(s))); eo.encode(int(to.cum_freq[s]), int(to.freq[s]))
        else:
            eg.encode_symbols(part, tg); eo.encode_symbols(part, to)
        ok &= eg.state == eo.state
    data_g, data_o = eg.finish(), eo.finish()
    note("coder-encode", ok and data_g == data_o, ("coder-encode", seed, n_sym, [len(x) for x in parts]))
    total = sum(len(x) for x in parts)
    dg, do = a.RansDecoder(data_o), o.RansDecoder(data_o)
    ok = True
    left = total + int(rng.integers(0, 50))       # a little past the end: both run dry the same way
    while left > 0:
        n = min(left, int(rng.choice([1, 2, 64, 4095, 4096, 4097, 10000])))
        ok &= np.array_equal(dg.decode_n(n, tg), do.decode_n(n, to)) and dg.state == do.state and dg.position == do.pos
        left -= n
    note("coder-decode", ok, ("coder-decode", seed, n_sym, total))


def batch_case(rng, seed):
    import ctypes as C
    import torch
    w, h, f = int(rng.choice([64, 96, 130])), int(rng.choice([32, 64, 75])), int(rng.choice([2, 8, 16]))
    B = int(rng.integers(1, 6))
    k, q = int(rng.integers(0, 3)), int(rng.choice([30, 80, 90, 100]))
    lib.alice_codec_test_set_tuning(int(rng.choice([0, 96, 1024 * 1024])))
    lib.alice_codec_test_set_value_table_radius(2048)
    chunks = [content(rng, w, h, f)[0] for _ in range(B)]
    rgb = torch.from_numpy(np.stack(chunks)).cuda()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    bt = a.Batch(w, h, f, B, q, a.WaveletType(k))
    st = torch.cuda.current_stream().cuda_stream
    ok = True
    try:
        refs = [o.encode(c, w, h, f, q, k) for c in chunks]
    except o.OracleError:      # a chunk the reference does not terminate on: the batch must refuse it too
        try:
            bt.encode(rgb.data_ptr(), st); bt.encode_finish()
            refused = False
        except a.CodecError:
            refused = True
        note("batch-diverges", refused, ("batch-diverges", seed, w, h, f, B, k, q))
        del bt
        return
    for rep in range(2):
        bt.encode(rgb.data_ptr(), st)
        sizes = bt.encode_finish()
        for i in range(B):
            t = torch.empty(int(sizes[i]), dtype=torch.uint8, device="cuda")
            hip.hipMemcpy(t.data_ptr(), bt.alc_ptr(i), int(sizes[i]), 3)
            ok &= bytes(t.cpu().numpy()) == refs[i]
        bt.decode(bt.alc_ptr(0), bt.alc_stride, None, st)
        bt.decode_finish()
        for i in range(B):
            t = torch.empty(w * h * f * 3, dtype=torch.uint8, device="cuda")
            hip.hipMemcpy(t.data_ptr(), bt.rgb_ptr(i), t.numel(), 3)
            ok &= np.array_equal(t.cpu().numpy(), o.decode(refs[i]))
    del bt
    note("batch", ok, ("batch", seed, w, h, f, B, k, q))


t0 = time.time()
seed = seed0


def one_case(seed):
    rng = np.random.default_rng(seed)
    r = seed % 10
    try:
        if r < 6:
            chunk_case(rng, seed)
        elif r < 9 or n_threads > 1:
            coder_case(rng, seed)
        else:
            batch_case(rng, seed)
    except Exception as e:   # noqa: BLE001
        note("exception", False, ("exception", seed, repr(e)[:300]))


if n_threads > 1:
    import itertools
    import threading
    ticket = itertools.count(seed0)
    lock = threading.Lock()

    def worker():
        a.set_device(0)
        while time.time() - t0 < budget:
            with lock:
                sd = next(ticket)
            one_case(sd)
    th = [threading.Thread(target=worker) for _ in range(n_threads)]
    [t.start() for t in th]
    while any(t.is_alive() for t in th):
        time.sleep(15)
        print(f"[soak] {n_threads} threads, {time.time() - t0:.0f} s, {len(bad)} mismatches, {counts}", flush=True)
    [t.join() for t in th]
    seed = next(ticket)
else:
    while time.time() - t0 < budget:
        one_case(seed)
        seed += 1
        if (seed - seed0) % 50 == 0:
            print(f"[soak] {seed - seed0} cases, {time.time() - t0:.0f} s, {len(bad)} mismatches, {counts}", flush=True)
lib.alice_codec_test_set_tuning(1024 * 1024)
lib.alice_codec_test_set_value_table_radius(2048)
print(f"[soak] done: seeds {seed0}..{seed - 1}, {counts}, mismatches: {len(bad)}")
for b in bad:
    print("  ", b)
sys.exit(1 if bad else 0)
