"""Developer check on a GPU box: HIP path vs oracle on a few shapes (not a test; see tests/)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle as o
import alice_codec_amd as a

rng = np.random.default_rng(1)
bad = 0
def check(name, cond):
    global bad
    print(("ok   " if cond else "FAIL ") + name, flush=True)
    if not cond: bad += 1

# stage: rans encode/decode vs oracle
for n in (0, 1, 5, 63, 64, 65, 1000, 1023, 1024, 1025, 5000, 70000):
    sym = (rng.integers(0, 256, n) * (rng.random(n) < 0.3)).astype(np.uint8)
    hist = np.bincount(sym, minlength=256).astype(np.uint32)
    tg = a.FrequencyTable.from_histogram(hist); to = o.FrequencyTable(hist)
    check(f"table n={n}", np.array_equal(tg.freq, to.freq) and np.array_equal(tg.cum_freq, to.cum_freq))
    enc = a.RansEncoder(); enc.encode_symbols(sym, tg); bg = enc.finish()
    bo = o.rans_encode(sym, to)
    check(f"rans enc n={n} ({len(bg)} vs {len(bo)})", bg == bo)
    dg = a.RansDecoder(bo).decode_n(n, tg); do = o.rans_decode(bo, n, to)
    check(f"rans dec n={n}", np.array_equal(dg, do))

for (w,h,f,q,k) in [(4,4,2,80,0),(4,4,2,90,1),(3,5,1,90,0),(1,1,1,100,0),(8,8,2,100,2),(64,64,8,100,2),(70,50,6,80,1),(129,67,5,75,1),(256,128,64,80,1),(200,100,33,90,0)]:
    for gen in ("grad","noise"):
        rgb = o.make_gradient(w,h,f) if gen=="grad" else rng.integers(0,256,w*h*f*3,dtype=np.uint8)
        ref = o.encode(rgb,w,h,f,q,k)
        try:
            ch = a.FrameEncoder(q, a.WaveletType(k)).encode(rgb,w,h,f); got = ch.to_bytes()
        except Exception as e:
            got = b""; print("  exception", e)
        same = got == ref
        if not same and got:
            ga = np.frombuffer(got,np.uint8); ra = np.frombuffer(ref,np.uint8)
            m = min(len(ga),len(ra)); d = np.nonzero(ga[:m]!=ra[:m])[0]
            print("   len", len(got), len(ref), "first diff", d[:5])
        check(f"encode {w}x{h}x{f} q{q} k{k} {gen}", same)
        dref = o.decode(ref)
        try:
            dgot = a.FrameDecoder().decode(a.EncodedChunk.from_bytes(ref))
        except Exception as e:
            dgot = np.zeros(0,np.uint8); print("  exception", e)
        check(f"decode {w}x{h}x{f} q{q} k{k} {gen}", np.array_equal(dgot, dref))
print("FAILED" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)
