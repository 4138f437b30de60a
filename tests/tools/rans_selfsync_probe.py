"""Probe (CPU, uses the oracle): does the reference's single-stream rANS decoder re-synchronise when it is started
in the middle of a stream with a guessed state?  That property is what "massively parallel ANS decoding by
self-synchronisation" schemes rely on.  Result recorded in DESIGN.md section 4.3: it does not (0 of 200 trials
within 4000 symbols): with a 32-bit state and byte renormalisation a perturbation of the high bits is carried
forward at constant relative size, so the chain cannot be cut without side information the `.alc` v1 format
does not carry."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as o  # noqa: E402

rng = np.random.default_rng(3)
n = 200000
p = np.array([0.45] + [0.55 * 0.8 ** k for k in range(1, 41)])
p /= p.sum()
sym = rng.choice(len(p), size=n, p=p).astype(np.uint8)
hist = np.bincount(sym, minlength=256).astype(np.uint32)
t = o.FrequencyTable(hist)
stream = np.frombuffer(o.rans_encode(sym, t), np.uint8)
freq = t.freq.astype(np.int64)
cum = t.cum_freq.astype(np.int64)
c2s = t.cum_to_sym
L = 1 << 23


def run(x, pos, steps):
    out = []
    for _ in range(steps):
        slot = x & 4095
        s = int(c2s[slot])
        x = int(freq[s]) * (x >> 12) + slot - int(cum[s])
        while x < L and pos < len(stream):
            x = ((x << 8) | int(stream[pos])) & 0xFFFFFFFF
            pos += 1
        out.append((pos, x, s))
    return out


x = 0
for i in range(4):
    x = (x << 8) | int(stream[i])
true = run(x, 4, 60000)
assert np.array_equal(np.array([s for _, _, s in true[:1000]], np.uint8), o.rans_decode(stream.tobytes(), 1000, t))
at = {}
for k, (p_, x_, _) in enumerate(true):
    at.setdefault(p_, []).append(x_)
random.seed(1)
hits = []
for _ in range(200):
    k0 = random.randrange(1000, 50000)
    tr = run((1 << 23) | random.getrandbits(23), true[k0][0], 4000)
    hits.append(next((j for j, (p_, x_, _) in enumerate(tr) if x_ in at.get(p_, ())), None))
ok = [h for h in hits if h is not None]
print(f"re-synchronised: {len(ok)} of {len(hits)} trials" + (f", median {np.median(ok)} symbols" if ok else ""))
