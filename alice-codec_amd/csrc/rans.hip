// rANS stage for gfx950: frequency-table build, single-stream encode chain, decode chain.
//
// Reference behaviour restated (reference checkout, file:line):
//   FrequencyTable::from_histogram / uniform      src/rans.rs:102-189
//   RansEncoder::encode / encode_symbols / finish  src/rans.rs:269-308
//   RansDecoder::new / decode / decode_n           src/rans.rs:330-381
//
// The single-stream format is one strict dependency chain per channel (state
// transitions are injective, so chains cannot be split or speculated), which
// leaves two levers on a GPU: (1) strip every state-independent operation out of
// the chain and (2) run many chains (3 channels x chunks in flight) side by side,
// one wavefront each.
//
// Encode chain ("ripple"): a wavefront loads 64 consecutive symbols, one per lane,
// and gathers each lane's table parameters from LDS.  The state then ripples
// through the lanes: every lane executes
//     xin  = dpp_wave_shr1(xout) + c_prev          (lane 0 keeps the carry-in)
//     y    = xin >> {0,8,16}                       (renormalise against freq<<19)
//     xout = y + (umulhi(y, rcp) >> rsh) * (4096 - freq)
// 64 times.  Lanes 0..t hold their final values after step t and are fixed points
// of the update from then on, so no masking is needed; after 64 steps every lane
// holds the exact state before and after its own symbol.  The serial chain is 9
// VALU instructions per symbol with no memory access and no cross-lane traffic
// other than the DPP operand.  Byte emission (0-2 bytes per symbol, taken from the
// low bytes of the pre-renormalisation state) is then a wave-parallel ballot/popcount
// compaction.  Bytes are written back to front so the stream comes out already
// "reversed" as RansEncoder::finish leaves it.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace alice {

// ----------------------------------------------------------------------------------
// Frequency table (device function shared by the table kernel and the decoder)
// ----------------------------------------------------------------------------------

// Computes the reference table for a histogram of n_sym <= 256 bins (FrequencyTable::from_histogram takes any slice,
// src/rans.rs:102-104; the pipeline always passes 256).  Must be called by the first 256 threads of a block (all of
// them), s = threadIdx.x; scratch: 256+ u32 in LDS; count = 0 for s >= n_sym.
// Returns freq/cum as the reference stores them (u16 truncated); symbols s >= n_sym do not exist: freq = cum = 0.
__device__ inline void freq_table_256(uint32_t count, uint32_t n_sym, uint32_t* scratch, uint32_t& freq16, uint32_t& cum16) {
    const int s = threadIdx.x;
    const bool exists = (uint32_t)s < n_sym;
    const uint32_t last = n_sym - 1u;
    __shared__ unsigned long long total_sh;
    if (s == 0) total_sh = 0ull;
    __syncthreads();
    atomicAdd(&total_sh, (unsigned long long)count);
    __syncthreads();
    const unsigned long long total = total_sh;
    uint32_t freq;
    if (!exists) {
        freq = 0u;
    } else if (total == 0ull) {
        freq = (kProbScale / n_sym) & 0xFFFFu;  // uniform(n): src/rans.rs:159-166
    } else {
        if (count == 0u) freq = 1u;  // src/rans.rs:117-118
        else {
            unsigned long long f = ((unsigned long long)count * kProbScale) / total;  // :120
            freq = (uint32_t)(f < 1ull ? 1ull : f);
        }
    }
    // exclusive scan of freq over 256 symbols
    scratch[s] = freq;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = (s >= off) ? scratch[s - off] : 0u;
        __syncthreads();
        scratch[s] += v;
        __syncthreads();
    }
    const uint32_t incl = scratch[s];
    const uint32_t nt = scratch[255];
    __syncthreads();
    uint32_t cum = incl - freq;
    if (total == 0ull) {
        // uniform: last.freq = 4096 - last.cum (src/rans.rs:169-172)
        if ((uint32_t)s == last) freq = (kProbScale - cum) & 0xFFFFu;
    } else if ((uint32_t)s == last && nt != kProbScale) {
        // src/rans.rs:128-132: wrapping cast to u16
        int32_t diff = (int32_t)kProbScale - (int32_t)nt;
        freq = (uint32_t)((int32_t)freq + diff) & 0xFFFFu;
    }
    freq16 = exists ? freq & 0xFFFFu : 0u;
    cum16 = exists ? cum & 0xFFFFu : 0u;
}

__device__ inline RansEncEntry make_enc_entry(uint32_t f, uint32_t c) {
    RansEncEntry e;
    e.freq = f;
    e.cum = c;
    // x_max = (L >> 12 << 8) * freq = freq << 19 (u64 in the reference, src/rans.rs:275);
    // states never reach 2^32-1, so saturation preserves every comparison.
    unsigned long long xm = (unsigned long long)f << 19;
    unsigned long long xm8 = (unsigned long long)f << 27;
    e.xmax = xm > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)xm;
    e.xmax8 = xm8 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)xm8;
    if (f == 0u || f > kProbScale) {
        // handled by the exact serial path in the encode kernel
        e.xmax = f == 0u ? 0xFFFFFFFFu : e.xmax;
        e.xmax8 = f == 0u ? 0xFFFFFFFFu : e.xmax8;
        e.rcp = 0u; e.rsh = 0u; e.g = 0; e.cbias = 0u;
    } else if (f == 1u) {
        // floor(y/1) = y: umulhi(y, 2^32-1) = y-1 for y >= 1, compensated in the bias:
        // y + (y-1)*4095 + c + 4095 = 4096*y + c
        e.rcp = 0xFFFFFFFFu; e.rsh = 0u; e.g = 4095; e.cbias = c + 4095u;
    } else {
        // 2 <= f <= 4096, y < f*2^19.  L = ceil(log2 f), m = ceil(2^(31+L)/f) < 2^32,
        // floor(y/f) = (y*m) >> (31+L) because y*(m*f - 2^(31+L)) < 2^(19+2L) <= 2^(31+L).
        uint32_t L = 32u - (uint32_t)__clz((int)(f - 1u));
        unsigned long long p = 1ull << (31u + L);
        e.rcp = (uint32_t)((p + f - 1u) / f);
        e.rsh = L - 1u;
        e.g = (int32_t)kProbScale - (int32_t)f;
        e.cbias = c;
    }
    return e;
}

// Decoder slot tables, filled as the reference fills cum_to_sym (src/rans.rs:135-144): zero-initialised, then symbol by
// symbol in index order over [cum, min(cum + freq, 4096)), so the LAST symbol that covers a slot owns it (ranges built
// from a histogram are disjoint; caller-supplied tables may overlap) and an uncovered slot belongs to symbol 0.
// Called by all 256 threads of the block, s = threadIdx.x holding symbol s; scratch: 768 u32 in LDS.  Returns the
// kTableDec* flags of the slots this thread filled.
__device__ inline uint32_t build_dec_slots(uint32_t f, uint32_t c, uint32_t* scratch, RansDecSlots* __restrict__ out) {
    const int s = threadIdx.x;
    uint32_t end = c + f;
    if (end > kProbScale) end = kProbScale;
    __syncthreads();
    scratch[s] = c;
    scratch[256 + s] = c < end ? end : c;   // empty range when cum >= 4096 or freq == 0
    scratch[512 + s] = f;
    out->symtab[s] = (f & 0xFFFFu) | (c << 16);
    __syncthreads();
    uint32_t fl = 0u;
    for (uint32_t slot = (uint32_t)s; slot < kProbScale; slot += 256u) {
        uint32_t owner = 0u;
        for (uint32_t k = 0; k < 256u; ++k)
            if (slot >= scratch[k] && slot < scratch[256 + k]) owner = k;
        const uint32_t fo = scratch[512 + owner], co = scratch[owner];
        // x' = freq * (x >> 12) + slot - cum = umulhi(F', x) + B' with F' = freq << 20 and
        // B' = slot - cum - ((freq * slot) >> 12): with x = 4096 h + slot, floor(freq * x / 4096) =
        // freq * h + floor(freq * slot / 4096).  freq = 4096 would overflow F'; there F' = 2^32 - 1
        // (umulhi gives x - 1 for x >= 1) and B' carries the + 1.
        uint32_t fe, be;
        if (fo >= kProbScale) {
            fe = 0xFFFFFFFFu; be = (slot - co) - slot + 1u;
            fl |= fo > kProbScale ? kTableDecExact : kTableDecBig;
        } else {
            fe = fo << 20; be = (slot - co) - ((fo * slot) >> kProbBits);
        }
        out->ftab[slot] = fe;
        out->btab[slot] = be;
        out->c2s[slot] = (uint8_t)owner;
    }
    return fl;
}

__global__ __launch_bounds__(256) void rans_table_kernel(const uint32_t* __restrict__ hist,
                                                         RansTable* __restrict__ tables, uint32_t n_sym) {
    __shared__ uint32_t scratch[768];
    const int chain = blockIdx.x;
    const int s = threadIdx.x;
    const uint32_t count = (uint32_t)s < n_sym ? hist[(size_t)chain * 256 + s] : 0u;
    uint32_t f, c;
    freq_table_256(count, n_sym, scratch, f, c);
    tables[chain].enc[s] = make_enc_entry(f, c);
    uint32_t fl = build_dec_slots(f, c, scratch, &tables[chain].dec);
    if (count > 0u && f == 0u) fl |= kTableDiverges;
    if (count > 0u && f > kProbScale) fl |= kTableNeedsGeneric;
    __shared__ uint32_t flags_sh;
    if (s == 0) flags_sh = 0u;
    __syncthreads();
    if (fl) atomicOr(&flags_sh, fl);
    __syncthreads();
    if (s == 0) tables[chain].flags = flags_sh | kTableVerified;
}

// Table from explicit (cum, freq) arrays -- stage-level API (FrequencyTable handle).
__global__ __launch_bounds__(256) void rans_table_from_arrays_kernel(const uint16_t* __restrict__ cum,
                                                                     const uint16_t* __restrict__ freq,
                                                                     RansTable* __restrict__ table) {
    __shared__ uint32_t scratch[768];
    __shared__ uint32_t flags_sh;
    const int s = threadIdx.x;
    if (s == 0) flags_sh = 0u;
    table->enc[s] = make_enc_entry(freq[s], cum[s]);
    const uint32_t fl = build_dec_slots(freq[s], cum[s], scratch, &table->dec);
    if (fl) atomicOr(&flags_sh, fl);
    __syncthreads();
    if (s == 0) table->flags = flags_sh;
}

// Two chains on one SIMD (more than 1024 chains in flight): the SIMD's issue arbiter prefers the OLDER wave, so the older
// chain runs almost undisturbed and the younger one gets what is left -- and the step waits for the slowest chain.  Taking
// turns was the proposed cure (round 2): every wave derives its priority from a bit of the shader clock
// (the same for both waves of a SIMD) XOR the low bit of its wave slot (consecutive slots: different for the two),
// re-evaluated once per tile, so at any time one of the two has priority 1 and the other 0.  Waves never wait on each
// other: this is arbitration only.  MEASURED in round 3 (960x540x64 chunks, profiles/r03_chain_probe_960x540_turns_*.log):
//   1200 chains (176 SIMDs shared): the slowest encoder of a pair 61.0 instead of 63.5 cycles/symbol (kernel 25.6 instead of
//   26.6 ns/symbol), decoders 48.6 instead of 49.3 ns/symbol -- the pairs move closer but do not finish together;
//   2046 chains (every SIMD shared): encoders unchanged (38.0 ns/symbol), decoders 86.4 instead of 79.2 ns/symbol: WORSE.
// And two chains per SIMD buy little in the first place: 2046 chains encode 53.9 Gsym/s against 50.3 with 1023 (+7 %) and
// decode 25.8 against 26.4 (-2 %): eight decoders on a CU saturate its one scalar unit.  So it is OFF unless
// ALICE_CHAIN_TURNS=1, and one chain per SIMD stays the operating point.  kExclusive instances never evaluate it.
__device__ __forceinline__ void chain_take_turns(uint32_t enabled) {
    if (!enabled) return;
    const uint32_t turn = (uint32_t)(clock64() >> 17);                                   // ~55 us at 2.4 GHz
    const uint32_t slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);           // HW_REG_HW_ID[3:0]: wave slot on the SIMD
    if ((turn ^ slot) & 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// ----------------------------------------------------------------------------------
// Encode chain
// ----------------------------------------------------------------------------------

constexpr int kEncTile = 1024;  // symbols staged per global load (16 B per lane)

// 64 ripple steps.  On entry lane 0 of xin holds the carry-in state and xout is
// don't-care; on exit every lane holds the state before (xin) and after (xout,
// without its own cum bias) its symbol.
__device__ __forceinline__ void ripple64(uint32_t& xin, uint32_t& xout, uint32_t xmax, uint32_t xmax8,
                                         uint32_t rcp, uint32_t rsh, int32_t g, uint32_t cprev) {
    uint32_t k, y, q;
    unsigned long long m1;
    // Wait states (gfx940-family): VALU-written SGPR/VCC -> VALU read needs 2 states
    // (the second compare and one SALU/nop sit between); VALU-written VGPR -> DPP read
    // needs 2 states (s_nop 1, or s_cmp + s_cbranch at the loop edge).
#define ALICE_RIPPLE_STEP(TAIL)                                                     \
    "v_add_u32_dpp %[xin], %[xout], %[cprev] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ge_u32_e64 %[m1], %[xin], %[xmax]\n\t"                                  \
    "v_cmp_ge_u32_e32 vcc, %[xin], %[xmax8]\n\t"                                   \
    "s_nop 0\n\t"                                                                    \
    "v_cndmask_b32_e64 %[k], 0, 8, %[m1]\n\t"                                      \
    "v_cndmask_b32_e64 %[k], %[k], 16, vcc\n\t"                                    \
    "v_lshrrev_b32_e32 %[y], %[k], %[xin]\n\t"                                     \
    "v_mul_hi_u32 %[q], %[y], %[rcp]\n\t"                                          \
    "v_lshrrev_b32_e32 %[q], %[rsh], %[q]\n\t"                                     \
    "v_mad_i32_i24 %[xout], %[q], %[g], %[y]\n\t" TAIL
    // fully unrolled: a taken loop branch costs ~25 cycles on a lone wave (scripts/probes/latency_probe.hip)
#define ALICE_RIPPLE_STEP4 ALICE_RIPPLE_STEP("s_nop 1\n\t") ALICE_RIPPLE_STEP("s_nop 1\n\t") ALICE_RIPPLE_STEP("s_nop 1\n\t") ALICE_RIPPLE_STEP("s_nop 1\n\t")
#define ALICE_RIPPLE_STEP16 ALICE_RIPPLE_STEP4 ALICE_RIPPLE_STEP4 ALICE_RIPPLE_STEP4 ALICE_RIPPLE_STEP4
    asm volatile(
        "s_nop 1\n\t"
        ALICE_RIPPLE_STEP16 ALICE_RIPPLE_STEP16 ALICE_RIPPLE_STEP16 ALICE_RIPPLE_STEP16
        : [xin] "+v"(xin), [xout] "+v"(xout), [k] "=&v"(k), [y] "=&v"(y), [q] "=&v"(q), [m1] "=&s"(m1)
        : [xmax] "v"(xmax), [xmax8] "v"(xmax8), [rcp] "v"(rcp), [rsh] "v"(rsh), [g] "v"(g),
          [cprev] "v"(cprev)
        : "vcc");
#undef ALICE_RIPPLE_STEP16
#undef ALICE_RIPPLE_STEP4
#undef ALICE_RIPPLE_STEP
}

// The same 64 steps with ONE compare per step, for chains whose states stay in [2^23, 2^31 + 2^17): every symbol that
// occurs has a frequency in 1..4096 (kTableVerified without the generic / diverges flags) and the chain started from
// 2^23.  Then the renormalisation shift needs only one threshold per symbol:
//   freq <= 16: freq << 19 <= 2^23 <= x, the first shift always happens:   k = 8 + 8 * (x >= freq << 27)
//   freq >= 17: freq << 27 > 2^31 + 2^17 > x, the second never happens:    k = 8 * (x >= freq << 19)
// (x < 2^31 + 2^17: x' = (q << 12) + r + cum with q < 2^19, r < freq <= 4096, cum < 2^16.)  thr / k0 / k1 are the lane's
// threshold and the two shift candidates.  9 issue slots per symbol instead of 11.
__device__ __forceinline__ void ripple64_clean(uint32_t& xin, uint32_t& xout, uint32_t thr, uint32_t k0, uint32_t k1,
                                               uint32_t rcp, uint32_t rsh, int32_t g, uint32_t cprev) {
    uint32_t k, y, q;
#define ALICE_RIPPLE_STEP                                                            \
    "v_add_u32_dpp %[xin], %[xout], %[cprev] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_ge_u32_e32 vcc, %[xin], %[thr]\n\t"                                     \
    "s_nop 1\n\t"                                                                  \
    "v_cndmask_b32_e32 %[k], %[k0], %[k1], vcc\n\t"                                \
    "v_lshrrev_b32_e32 %[y], %[k], %[xin]\n\t"                                     \
    "v_mul_hi_u32 %[q], %[y], %[rcp]\n\t"                                          \
    "v_lshrrev_b32_e32 %[q], %[rsh], %[q]\n\t"                                     \
    "v_mad_i32_i24 %[xout], %[q], %[g], %[y]\n\t"                                  \
    "s_nop 1\n\t"
#define ALICE_RIPPLE_STEP4 ALICE_RIPPLE_STEP ALICE_RIPPLE_STEP ALICE_RIPPLE_STEP ALICE_RIPPLE_STEP
#define ALICE_RIPPLE_STEP16 ALICE_RIPPLE_STEP4 ALICE_RIPPLE_STEP4 ALICE_RIPPLE_STEP4 ALICE_RIPPLE_STEP4
    asm volatile(
        "s_nop 1\n\t"
        ALICE_RIPPLE_STEP16 ALICE_RIPPLE_STEP16 ALICE_RIPPLE_STEP16 ALICE_RIPPLE_STEP16
        : [xin] "+v"(xin), [xout] "+v"(xout), [k] "=&v"(k), [y] "=&v"(y), [q] "=&v"(q)
        : [thr] "v"(thr), [k0] "v"(k0), [k1] "v"(k1), [rcp] "v"(rcp), [rsh] "v"(rsh), [g] "v"(g), [cprev] "v"(cprev)
        : "vcc");
#undef ALICE_RIPPLE_STEP16
#undef ALICE_RIPPLE_STEP4
#undef ALICE_RIPPLE_STEP
}

// kExclusive: the kernel claims more than half of the SIMD's 512 registers (an AGPR clobber; nothing uses them), so
// the dispatcher cannot put two chains on one SIMD -- see launch_rans_encode.
template <bool kExclusive>
__global__ __launch_bounds__(64) void rans_encode_kernel(const uint8_t* __restrict__ sym_base,
                                                         unsigned long long sym_stride,
                                                         unsigned long long n_first,
                                                         const RansTable* __restrict__ tables,
                                                         uint8_t* __restrict__ out_base,
                                                         unsigned long long cap0,
                                                         unsigned long long group_stride,
                                                         unsigned long long group_head,
                                                         unsigned n_split,
                                                         RansResult* __restrict__ results,
                                                         unsigned long long cap1, unsigned long long cap2,
                                                         uint32_t x_init, uint32_t keep_open, uint32_t take_turns,
                                                         const RansEncodeDesc* __restrict__ descs) {
    __shared__ uint4 tab_a[256];  // xmax, xmax8, rcp, rsh
    __shared__ uint4 tab_b[256];  // g, cbias, freq, cum
    __shared__ __attribute__((aligned(16))) uint8_t tile[kEncTile];

    if constexpr (kExclusive) asm volatile("" ::: "a255");
    const int chain = blockIdx.x;
    const int lane = threadIdx.x;
    // chains from n_split on carry one symbol less (the sub-sequences of an interleaved stream)
    unsigned long long n = n_first - (((unsigned)chain >= n_split && n_first > 0ull) ? 1ull : 0ull);
    const uint8_t* __restrict__ sym = sym_base + (size_t)chain * sym_stride;
    const RansTable* __restrict__ tbl = tables + chain;
    // group_stride == 0: regions back to back.  Otherwise the three chains of chunk g write into the chunk's own
    // .alc buffer, behind `group_head` bytes kept free for the header (the streams are compacted in place later).
    const int gch = chain % 3;
    unsigned long long cap = group_stride == 0ull ? cap0 : (gch == 0 ? cap0 : (gch == 1 ? cap1 : cap2));
    uint8_t* region = group_stride == 0ull ? out_base + (size_t)chain * cap0
                                           : out_base + (size_t)(chain / 3) * group_stride + group_head +
                                                 (gch == 0 ? 0ull : (gch == 1 ? cap0 : cap0 + cap1));
    RansResult* res = results + chain;
    if (descs) {   // a merged launch (launch_rans_encode_descs): every chain spelled out
        const RansEncodeDesc ds = descs[chain];
        sym = ds.sym; n = ds.n; tbl = ds.table; region = ds.region; cap = ds.cap; res = ds.result;
        x_init = ds.x_init; keep_open = ds.keep_open;
    }
    uint8_t* const out_end = region + cap;
    // lanes that have nothing to emit store to a private byte at the unused front of the region instead
    // of branching around the store; the capacity test keeps real bytes 64 bytes away from it
    uint8_t* const dummy = region + lane;

    for (int s = lane; s < 256; s += 64) {
        const RansEncEntry e = tbl->enc[s];
        tab_a[s] = make_uint4(e.xmax, e.xmax8, e.rcp, e.rsh);
        tab_b[s] = make_uint4((uint32_t)e.g, e.cbias, e.freq, e.cum);
    }

    // (a state carried in from an earlier call may lie outside the range the one-compare step assumes)
    const bool table_clean = x_init == kRansL && (tbl->flags & (kTableVerified | kTableNeedsGeneric | kTableDiverges)) == kTableVerified;  // uniform
    uint32_t x = x_init;  // RansEncoder::new, src/rans.rs:249-254: 2^23; a continued encoder (:288-294) brings its state
    const unsigned long long clk0 = clock64(), rt0 = wall_clock64();
    unsigned long long written = 0ull;
    uint32_t flags = 0u;

    const unsigned long long ntiles = (n + kEncTile - 1) / kEncTile;
    // tile j covers symbol indices [hi - 1024, hi), hi = n - j*1024 (clipped at 0)
    auto load_tile = [&](unsigned long long j) -> uint4 {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (j >= ntiles) return v;
        const long long hi = (long long)(n - j * kEncTile);
        const long long lo = hi - kEncTile + 16ll * lane;  // first index of this lane's 16 bytes
        if (lo >= 0) {
            const uint8_t* p = sym + lo;
            const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
            const uint32_t* p4 = (const uint32_t*)(p - mis);   // stays inside the allocation: lo >= 0
            if (mis == 0u) {
                v = make_uint4(p4[0], p4[1], p4[2], p4[3]);
            } else if (lo + 20 <= (long long)n || mis == 0u) {
                const uint32_t t0 = p4[0], t1 = p4[1], t2 = p4[2], t3 = p4[3], t4 = p4[4];
                const uint32_t sh = mis * 8u;
                v = make_uint4((t0 >> sh) | (t1 << (32u - sh)), (t1 >> sh) | (t2 << (32u - sh)),
                               (t2 >> sh) | (t3 << (32u - sh)), (t3 >> sh) | (t4 << (32u - sh)));
            } else {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                for (int b = 0; b < 16; ++b) w[b >> 2] |= (uint32_t)p[b] << (8 * (b & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        } else {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            for (int b = 0; b < 16; ++b) {
                const long long idx = lo + b;
                if (idx >= 0) w[b >> 2] |= (uint32_t)sym[idx] << (8 * (b & 3));
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return v;
    };

    struct BlockParams { uint4 ea, eb; };
    auto sym_of = [&](int b) -> uint32_t { return tile[(kEncTile - 1 - (b * 64 + lane)) & (kEncTile - 1)]; };
    auto params_of = [&](uint32_t s) -> BlockParams { BlockParams p; p.ea = tab_a[s]; p.eb = tab_b[s]; return p; };

    uint4 cur = load_tile(0);
    for (unsigned long long j = 0; j < ntiles; ++j) {
        if (!kExclusive && (j & 7ull) == 0ull) chain_take_turns(take_turns);
        __syncthreads();
        ((uint4*)tile)[lane] = cur;
        __syncthreads();
        cur = load_tile(j + 1);  // in flight while this tile's 16 blocks ripple
        const long long hi = (long long)(n - j * kEncTile);
        const int valid = hi >= kEncTile ? kEncTile : (int)hi;  // tile-local indices [1024-valid, 1024)
        const int nblocks = (valid + 63) / 64;

        // Three-stage pipeline over the blocks of a tile: while block b ripples (the asm statement touches
        // no memory), the table rows of block b+1 and the symbols of block b+2 are in flight from LDS.
        uint32_t sym1 = sym_of(1);
        BlockParams nxt = params_of(sym_of(0));
        // Full tile, table known to hold only frequencies 1..4096 for the symbols that occur, room for the worst
        // case of the whole tile: no per-block activity mask, table check or capacity test; the state is carried
        // from block to block inside the vector unit (lane 0 <- lane 63 by a wave rotate) instead of through
        // two readlanes and a scalar add.
        if (table_clean && valid == kEncTile && written + 2ull * kEncTile + 4ull + 320ull <= cap) {
            uint32_t xin = x;
#pragma unroll 2
            for (int b = 0; b < kEncTile / 64; ++b) {
                const BlockParams curp = nxt;
                nxt = params_of(sym1);
                sym1 = sym_of(b + 2);
                const uint32_t xmax = curp.ea.x, xmax8 = curp.ea.y, rcp = curp.ea.z, rsh = curp.ea.w;
                const int32_t g = (int32_t)curp.eb.x;
                const uint32_t cbias = curp.eb.y;
                uint32_t xout = 0u;
                const uint32_t cprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cbias, 0x138, 0xf, 0xf, true);
                const bool small_f = xmax <= kRansL;            // freq <= 16
                const uint32_t k0 = small_f ? 8u : 0u;
                ripple64_clean(xin, xout, small_f ? xmax8 : xmax, k0, k0 + 8u, rcp, rsh, g, cprev);
                const bool c1 = xin >= xmax;
                const bool c2 = xin >= xmax8;
                const unsigned long long b1 = __ballot(c1), b2 = __ballot(c2);
                const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
                                     __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
                // one uniform base 320 bytes ahead of the write position; real bytes at 319 - off and 318 - off,
                // lanes with nothing to emit write their own byte of the 64 at the bottom (not yet written stream space)
                uint8_t* const base = out_end - written - 320ull;
                base[c1 ? 319u - off : (uint32_t)lane] = (uint8_t)(xin & 0xFFu);
                base[c2 ? 318u - off : (uint32_t)lane] = (uint8_t)((xin >> 8) & 0xFFu);
                written += (unsigned long long)((uint32_t)__popcll(b1) + (uint32_t)__popcll(b2));
                // wave_ror:1 -- lane 0 of the next block's xin receives xout[63] + cbias[63]
                xin = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(xout + cbias), 0x13C, 0xf, 0xf, false);
            }
            x = (uint32_t)__builtin_amdgcn_readlane((int)xin, 0);
            continue;
        }
        for (int b = 0; b < nblocks; ++b) {
            const BlockParams curp = nxt;
            nxt = params_of(sym1);
            sym1 = sym_of(b + 2);
            const bool active = (kEncTile - 1 - (b * 64 + lane)) >= kEncTile - valid;  // all lanes except in the last tile
            uint4 ea = curp.ea;
            uint4 eb = curp.eb;
            if (!active) {  // identity step
                ea = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
                eb = make_uint4(0u, 0u, 1u, 0u);
            }
            const uint32_t xmax = ea.x, xmax8 = ea.y, rcp = ea.z, rsh = ea.w;
            const int32_t g = (int32_t)eb.x;
            const uint32_t cbias = eb.y, freq = eb.z, cum = eb.w;
            const bool bad = (freq - 1u) >= kProbScale;  // freq == 0 or freq > 4096 (identity lanes carry freq 1)

            uint32_t xin = x, xout = 0u;
            if (__builtin_expect(__ballot(bad) == 0ull, 1)) {
                uint32_t cprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cbias, 0x138, 0xf, 0xf, true);
                ripple64(xin, xout, xmax, xmax8, rcp, rsh, g, cprev);
                x = (uint32_t)__builtin_amdgcn_readlane((int)xout, 63) +
                    (uint32_t)__builtin_amdgcn_readlane((int)cbias, 63);
            } else {
                // exact serial path (a table entry outside 1..4096 is in use): true division
                uint32_t xs = x;
                for (int i = 0; i < 64; ++i) {
                    if (lane == i) xin = xs;
                    const uint32_t fi = (uint32_t)__builtin_amdgcn_readlane((int)freq, i);
                    const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)cum, i);
                    const uint32_t ai = (uint32_t)__builtin_amdgcn_readlane((int)active, i);
                    if (!ai) continue;
                    if (fi == 0u) { flags |= kTableDiverges; continue; }
                    const unsigned long long xm = (unsigned long long)fi << 19;  // src/rans.rs:275
                    uint32_t ys = xs;
                    while ((unsigned long long)ys >= xm) ys >>= 8;              // :276-279
                    const uint32_t qs = ys / fi, rs = ys % fi;                  // :282-283
                    xs = (qs << kProbBits) + rs + ci;                           // :284
                }
                x = xs;
            }

            // byte emission, branch-free: lane i pushes the low k_i bytes of its pre-renormalisation state
            // (identity lanes have xmax = 2^32-1 and never emit)
            const bool c1 = xin >= xmax;
            const bool c2 = xin >= xmax8;
            const unsigned long long b1 = __ballot(c1), b2 = __ballot(c2);
            const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
                                 __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
            const uint32_t total = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
            const bool room = written + total + 4ull + 64ull <= cap;  // uniform
            flags |= room ? 0u : kRansOverflow;
            uint8_t* p = out_end - 1 - (written + off);
            uint8_t* p0 = (c1 && room) ? p : dummy;
            uint8_t* p1 = (c2 && room) ? p - 1 : dummy;
            *p0 = (uint8_t)(xin & 0xFFu);
            *p1 = (uint8_t)((xin >> 8) & 0xFFu);
            written += total;
        }
    }

    // finish (src/rans.rs:298-308): push the 4 state bytes LSB first; the reversal is implicit.  keep_open: the encoder
    // object lives on (more encode / encode_symbols calls follow): the state goes back to the host instead.
    if (!keep_open) {
        if (written + 4ull + 64ull <= cap) {
            if (lane < 4) out_end[-1 - (long long)(written + lane)] = (uint8_t)((x >> (8 * lane)) & 0xFFu);
        } else {
            flags |= kRansOverflow;
        }
        written += 4ull;
    }
    if (lane == 0) {
        res->len = written;
        res->flags = flags;
        res->final_state = x;
        res->fast_tiles = 0u;
        res->slow_tiles = 0u;
        res->cycles_k = (uint32_t)((clock64() - clk0) >> 10);
        res->ticks_k = (uint32_t)((wall_clock64() - rt0) >> 10);
        res->hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        res->xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

// ----------------------------------------------------------------------------------
// Decode chain
//
// Per symbol the reference does (src/rans.rs:351-371):
//     slot = x & 4095; sym = cum_to_sym[slot]; x = freq*(x >> 12) + slot - cum;
//     while x < 2^23 && pos < len { x = x << 8 | byte }
// Every step depends on the previous state, including the table lookup.  To keep memory out of
// the chain the 4096-entry slot table (freq | (slot - cum) << 16) lives in 64 VGPRs and is read
// in VGPR-index mode (s_set_gpr_idx_idx row = slot[11:6], then an indexed v_readlane with lane = slot[5:0],
// whose lane select the hardware masks to 6 bits); the state, the byte window
// and all arithmetic stay on the scalar unit.  The stream bytes of a tile also sit in VGPRs
// (big-endian dwords) and are pulled into a 64-bit scalar window by the same indexed readlane, about once per 4 consumed bytes.  The only per-symbol output is the pre-update state
// (one v_writelane); symbols are recovered from those states 64 at a time by the vector unit.
// Tiles that cannot use the fast path (initial state below 2^23, stream nearly exhausted) run
// an exact scalar-lane loop.
// ----------------------------------------------------------------------------------

constexpr int kDecBlocks = 64;                 // 64-symbol blocks per fast tile
constexpr int kDecTile = kDecBlocks * 64;      // 4096 symbols
constexpr int kDecWinBytes = 33 * 64 * 4;       // stream window held in 33 VGPRs (big-endian dwords): 2 B/symbol + slack
constexpr int kDecWinLds = kDecWinBytes;

// The tile body is generated (gen/gen_rans_decode_asm.py -> rans_decode_tile.inc, which documents the
// register plan and the measured instruction costs behind its shape); per symbol, all on the scalar side:
//     row = x[11:6]; freq = F[row][x[5:0]]; bias = B[row][x[5:0]]      (VGPR-index mode + two v_readlane)
//     x = freq * (x >> 12) + bias
//     sh = shift_by_clz[clz(x)]                                         (s_flbit + s_movrels: 0, 8 or 16)
//     {window : x} <<= sh                                               (64-bit scalar shifts)
// and once per two symbols a test whether the 64-bit byte window needs its next dword.
#include "rans_decode_tile.inc"

// Decodes nblk*64 symbols on the fast path.  slots = the chain's RansDecSlots (F' at byte 0, B' at byte 16384, read with
// 128 global loads per lane: lane4 = 4 * lane); win_addr / rec_addr are this lane's LDS byte addresses (base + 4*lane;
// the record is 2 bytes per lane).  pos = byte offset of the next stream byte inside the window.
__device__ __forceinline__ void dec_tile_fast(uint32_t& x, uint32_t& pos, const RansDecSlots* slots, uint32_t lane4,
                                              uint32_t win_addr, uint32_t rec_addr, uint32_t nblk) {
    uint32_t xo, po;
    asm volatile(ALICE_DEC_TILE_ASM
                 : [xo] "=&s"(xo), [po] "=&s"(po), [ra] "+v"(rec_addr)
                 : [xi] "s"(x), [pi] "s"(pos), [nb] "s"(nblk), [tp] "s"(slots), [l4] "v"(lane4), [wa] "v"(win_addr)
                 : ALICE_DEC_TILE_CLOBBERS);
    x = xo;
    pos = po;
}

// The same lookup and update with the stream exhausted: nothing is shifted in any more (src/rans.rs:365-368 reads
// no byte once pos == len), the state just evolves.  A frequency-4096 entry is only right for x >= 1 (see
// build_dec_slots); the caller checks that.
__device__ __forceinline__ void dec_tile_dry(uint32_t& x, const RansDecSlots* slots, uint32_t lane4, uint32_t rec_addr, uint32_t nblk) {
    uint32_t xo;
    asm volatile(ALICE_DEC_DRY_TILE_ASM
                 : [xo] "=&s"(xo), [ra] "+v"(rec_addr)
                 : [xi] "s"(x), [nb] "s"(nblk), [tp] "s"(slots), [l4] "v"(lane4)
                 : ALICE_DEC_DRY_TILE_CLOBBERS);
    x = xo;
}

// LDS per chain: cum_to_sym 4 KB + exact-loop symbol table 1 KB + stream window 8.25 KB + state record 8 KB
// (the exact loop's output bytes share the record's space) = 21.3 KB.  The launcher adds dynamic LDS so that at most
// four (up to 1024 chains: one per SIMD) or seven chains share a CU.
template <bool kExclusive>
__global__ __launch_bounds__(64) void rans_decode_kernel(const RansDecodeDesc* __restrict__ descs,
                                                         RansResult* __restrict__ results, uint32_t take_turns) {
    if constexpr (kExclusive) asm volatile("" ::: "a63");   // 226 VGPRs + 64 AGPRs > 256: one chain per SIMD
    __shared__ __attribute__((aligned(16))) uint8_t c2s[kProbScale];                  // cum_to_sym
    __shared__ uint32_t symtab[256];                                                    // freq | cum << 16 (exact loop)
    __shared__ __attribute__((aligned(16))) uint8_t win[kDecWinLds];
    __shared__ __attribute__((aligned(16))) uint16_t rec[kDecTile];                     // low 16 bits of the pre-update states
    uint8_t* const obuf = (uint8_t*)rec;                                                // exact loop / unaligned output staging

    const RansDecodeDesc d = descs[blockIdx.x];
    const int lane = threadIdx.x;
    const RansDecSlots* const slots = &d.table->dec;
    uint32_t flags = 0u;
    const uint32_t tflags = d.table->flags;
    const bool big_freq = (tflags & kTableDecBig) != 0u;   // some slot belongs to a symbol of frequency 4096 (uniform)
    const bool exact_only = (tflags & kTableDecExact) != 0u;
    for (int i = lane; i < (int)kProbScale / 16; i += 64) ((uint4*)c2s)[i] = ((const uint4*)slots->c2s)[i];
    for (int i = lane; i < 256; i += 64) symtab[i] = slots->symtab[i];
    __syncthreads();
    bool slot0_big = false, one_owner = false;   // uniform
    if (big_freq) {
        const uint32_t o0 = c2s[0];
        slot0_big = symtab[o0] == kProbScale;     // freq 4096, cum 0
        bool same = true;
        for (int i = lane; i < (int)kProbScale; i += 64) same &= c2s[i] == o0;
        one_owner = __ballot(!same) == 0ull;
    }

    // RansDecoder::new (src/rans.rs:330-347); a decoder object that has decoded before resumes from its state
    // (decode / decode_n continue, :351-381)
    uint32_t x = 0u;
    unsigned long long pos = 0ull;
    const unsigned long long len = d.in_len;
    if (d.resume) {
        x = d.x0; pos = d.pos0;
    } else if (len >= 4ull) {
        x = ((uint32_t)d.in[0] << 24) | ((uint32_t)d.in[1] << 16) | ((uint32_t)d.in[2] << 8) | (uint32_t)d.in[3];
        pos = 4ull;
    }
    bool pending = false;  // renormalisation owed by the previous symbol
    const unsigned long long clk0 = clock64(), rt0 = wall_clock64();
    unsigned long long done = 0ull;
    uint32_t n_fast = 0u, n_slow = 0u, paths = 0u;

    while (done < d.n) {
        if (!kExclusive) chain_take_turns(take_turns);
        const unsigned long long remain = d.n - done;
        const uint32_t want = remain < (unsigned long long)kDecTile ? (uint32_t)remain : (uint32_t)kDecTile;
        // A frequency-4096 entry is wrong for x = 0 only (build_dec_slots), and x = 0 looks up slot 0: the dry tile is
        // safe unless such a symbol owns slot 0; if it does and owns every slot, x' = x - cum = x and a non-zero state
        // stays what it is.
        const bool dry_ok = !exact_only && (!slot0_big || (one_owner && x != 0u));
        if (want == (uint32_t)kDecTile && pos >= len && dry_ok) {
            // stream exhausted (a desynchronised decoder runs dry long before its last symbol): no window, no shifts
            uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
            __syncthreads();
            dec_tile_dry(xs, slots, 4u * lane, (uint32_t)(uintptr_t)rec + 2u * lane, (uint32_t)kDecBlocks);
            x = xs;
            __syncthreads();
            const bool out_al = (((uintptr_t)(d.out + done)) & 3u) == 0u;
            paths |= kDecPathDry | (out_al ? 0u : kDecPathUnaligned);
            for (int i = lane * 4; i < kDecTile; i += 256) {
                const uint2 r4 = *(const uint2*)&rec[i];
                const uint32_t packed = (uint32_t)c2s[r4.x & (kProbScale - 1u)] | ((uint32_t)c2s[(r4.x >> 16) & (kProbScale - 1u)] << 8) |
                                        ((uint32_t)c2s[r4.y & (kProbScale - 1u)] << 16) | ((uint32_t)c2s[(r4.y >> 16) & (kProbScale - 1u)] << 24);
                if (out_al) *(uint32_t*)(d.out + done + i) = packed;
                else for (int b = 0; b < 4; ++b) d.out[done + i + b] = (uint8_t)(packed >> (8 * b));
            }
            done += (unsigned long long)kDecTile;
            ++n_fast;
            continue;
        }
        // Stage the stream window [wbase, wbase + kDecWinLds) into LDS (zero beyond len).  The base is chosen so
        // that its global address is dword aligned; when the whole window lies inside the stream all 33 dword
        // loads of a lane are issued back to back (one memory latency per tile instead of 33).
        unsigned long long wbase = pos & ~3ull;
        {
            const unsigned long long mis = (unsigned long long)((uintptr_t)(d.in + pos) & 3u);
            if (pos >= mis) wbase = pos - mis;
        }
        const bool whole = (len - wbase >= (unsigned long long)kDecWinLds) && ((((uintptr_t)(d.in + wbase)) & 3u) == 0u);
        __syncthreads();
        if (whole) {
            const uint32_t* g = (const uint32_t*)(d.in + wbase) + lane;
            uint32_t wv[kDecWinLds / 256];
#pragma unroll
            for (int k = 0; k < kDecWinLds / 256; ++k) wv[k] = g[k * 64];
#pragma unroll
            for (int k = 0; k < kDecWinLds / 256; ++k) ((uint32_t*)win)[k * 64 + lane] = wv[k];
        } else {
            for (int i = lane * 4; i < kDecWinLds; i += 64 * 4) {
                const unsigned long long o = wbase + (unsigned long long)i;
                uint32_t w = 0u;
                for (int b = 0; b < 4; ++b)
                    if (o + (unsigned long long)b < len) w |= (uint32_t)d.in[o + b] << (8 * b);
                *(uint32_t*)(win + i) = w;
            }
        }
        __syncthreads();

        // the owed renormalisation first (src/rans.rs:365-368); it needs at most a few window bytes
        // unless the state is 0, which the exact loop below handles byte by byte
        uint32_t got = 0u;
        // Near the end of the stream the window is zero-padded and the fast tile runs speculatively: if it
        // turns out to have consumed padding (it ran dry part-way) its result is dropped and the exact loop redoes
        // the tile from the saved state.  A desynchronised decoder that eats a fraction of a bit per symbol spends
        // hundreds of tiles inside the last window; they all stay on the fast path this way.
        bool fast = (want == (uint32_t)kDecTile) && pos < len && !exact_only;
        const uint32_t x_save = x;
        const unsigned long long pos_save = pos;
        const bool pending_save = pending;
        if (fast && pending) {
            uint32_t xs = x;
            unsigned long long ps = pos;
            int guard = 0;
            while (xs < kRansL && ps < len && guard < 8) { xs = (xs << 8) | (uint32_t)win[ps - wbase]; ps += 1ull; ++guard; }
            if (xs < kRansL) { fast = false; paths |= kDecPathStillStarved; }  // still starved: leave x/pos untouched for the exact loop
            else { x = xs; pos = ps; pending = false; paths |= kDecPathPendingFed; }
        }
        if (fast && x < kRansL) { fast = false; paths |= kDecPathBelowL; }  // only the very first symbol of a malformed stream
        if (fast) {
            uint32_t prel = (uint32_t)(pos - wbase);
            uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
            prel = (uint32_t)__builtin_amdgcn_readfirstlane((int)prel);
            dec_tile_fast(xs, prel, slots, 4u * lane, (uint32_t)(uintptr_t)win + 4u * lane,
                          (uint32_t)(uintptr_t)rec + 2u * lane, (uint32_t)kDecBlocks);
            x = xs;
            pos = wbase + prel;
            pending = false;  // the fast path renormalises right after each update
            __syncthreads();
            if (pos > len) {  // uniform: padding was consumed
                x = x_save; pos = pos_save; pending = pending_save;
                fast = false;
                paths |= kDecPathSpecDropped;
            } else {
                paths |= whole ? kDecPathWhole : kDecPathSpecKept;
            }
        }
        if (fast) {
            // symbols from the recorded states, 4 per lane; straight to global memory when the output is aligned
            const bool out_aligned = (((uintptr_t)(d.out + done)) & 3u) == 0u;
            for (int i = lane * 4; i < kDecTile; i += 256) {
                const uint2 r4 = *(const uint2*)&rec[i];
                const uint32_t packed = (uint32_t)c2s[r4.x & (kProbScale - 1u)] | ((uint32_t)c2s[(r4.x >> 16) & (kProbScale - 1u)] << 8) |
                                        ((uint32_t)c2s[r4.y & (kProbScale - 1u)] << 16) | ((uint32_t)c2s[(r4.y >> 16) & (kProbScale - 1u)] << 24);
                if (out_aligned) *(uint32_t*)(d.out + done + i) = packed;
                else *(uint32_t*)(obuf + i) = packed;
            }
            paths |= out_aligned ? 0u : kDecPathUnaligned;
            got = out_aligned ? 0u : (uint32_t)kDecTile;   // 0: nothing left in obuf to flush
            done += out_aligned ? (unsigned long long)kDecTile : 0ull;
            ++n_fast;
        } else {
            ++n_slow;
            paths |= kDecPathExact | (want < (uint32_t)kDecTile ? kDecPathTail : 0u);
            // exact scalar-lane loop over what the window can feed
            __shared__ uint32_t sh_state, sh_cnt, sh_starved;
            __shared__ unsigned long long sh_pos;
            if (lane == 0) {
                uint32_t j = 0u, xs = x;
                unsigned long long ps = pos;
                bool pend = pending;
                const unsigned long long wend = wbase + (unsigned long long)kDecWinBytes;
                bool starved = false;
                while (j < want) {
                    if (pend) {
                        while (xs < kRansL && ps < len) {
                            if (ps >= wend) { starved = true; break; }
                            xs = (xs << 8) | (uint32_t)win[ps - wbase];
                            ps += 1ull;
                        }
                        if (starved) break;
                    }
                    const uint32_t slot = xs & (kProbScale - 1u);                 // src/rans.rs:353
                    const uint32_t sy = c2s[slot];                                // :356
                    const uint32_t fc = symtab[sy];
                    xs = (fc & 0xFFFFu) * (xs >> kProbBits) + slot - (fc >> 16);  // :361-362 (mod 2^32)
                    obuf[j] = (uint8_t)sy;
                    pend = true;
                    ++j;
                }
                sh_state = xs; sh_pos = ps; sh_cnt = j; sh_starved = starved ? 1u : 0u;
            }
            __syncthreads();
            x = sh_state; pos = sh_pos; got = sh_cnt;
            paths |= sh_starved ? kDecPathStarved : 0u;
            pending = true;
        }
        __syncthreads();
        for (uint32_t i = lane; i < got; i += 64) d.out[done + i] = obuf[i];
        done += got;
    }
    // the reference renormalises right after every symbol (src/rans.rs:365-368): settle what the exact loop still owes,
    // so that the reported state and position are the decoder object's
    if (pending) {
        while (x < kRansL && pos < len) { x = (x << 8) | (uint32_t)d.in[pos]; pos += 1ull; }
    }
    if (lane == 0) {
        RansResult* const res = d.result ? d.result : results + blockIdx.x;
        res->len = pos;
        res->flags = flags;
        res->final_state = x;
        res->fast_tiles = n_fast;
        res->slow_tiles = n_slow;
        res->paths = paths;
        res->cycles_k = (uint32_t)((clock64() - clk0) >> 10);
        res->ticks_k = (uint32_t)((wall_clock64() - rt0) >> 10);
        res->hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        res->xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
}

// ----------------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------------

void launch_rans_table(const uint32_t* d_hist, RansTable* d_tables, int n_chains, hipStream_t st, uint32_t n_symbols) {
    if (n_chains <= 0) return;
    hipLaunchKernelGGL(rans_table_kernel, dim3(n_chains), dim3(256), 0, st, d_hist, d_tables, n_symbols);
}

void launch_rans_table_from_arrays(const uint16_t* d_cum, const uint16_t* d_freq, RansTable* d_table,
                                   hipStream_t st) {
    hipLaunchKernelGGL(rans_table_from_arrays_kernel, dim3(1), dim3(256), 0, st, d_cum, d_freq, d_table);
}

// A chain is one wavefront that runs for seconds, and waves never migrate: two chains that the dispatcher puts on one
// SIMD while another SIMD idles run at 0.7 / 1.0 of their speed to the end (measured: with 900 chains, 4 single-wave
// workgroups per CU, a tenth of the SIMDs hosted two encoder waves).  Up to 1024 chains the kernels therefore claim more
// than half of a SIMD's register file, which leaves the dispatcher no choice but one chain per SIMD.  Beyond that two
// chains per SIMD are wanted (together they run at about 1.5x the rate of one); dynamic LDS that the kernels never touch
// then caps the workgroups per CU at 8 so that the surplus spreads over all CUs.
// ALICE_CHAIN_TURNS=1 switches the priority alternation of shared SIMDs on (developer A/B, scripts/chain_probe.py)
static uint32_t chain_turns_enabled() {
    static const uint32_t on = [] { const char* v = getenv("ALICE_CHAIN_TURNS"); return (v && *v == '1') ? 1u : 0u; }();
    return on;
}

static unsigned chain_lds_pad(int n_chains, unsigned static_lds) {
    if (n_chains <= 1024 || n_chains > 2048) return 0u;
    const unsigned want = 163840u / 8u - 1024u;   // a ninth workgroup no longer fits (160 KB of LDS per CU)
    return want > static_lds ? want - static_lds : 0u;
}

// The placement above rests on two things the compiler decides: that the empty AGPR clobber counts towards the kernel's
// register allocation, and how much static LDS the kernels use.  Both are read back from the runtime here (once), so a
// compiler update or a kernel edit that breaks either shows up as a failed check instead of a slower headline number.
struct ChainKernelFacts { bool ok; unsigned enc_regs, enc_lds, enc_wg_per_cu, dec_regs, dec_lds, dec_wg_per_cu, enc_lds_plain, dec_lds_plain; };
static const ChainKernelFacts& chain_kernel_facts() {
    static const ChainKernelFacts f = [] {
        ChainKernelFacts r{};
        hipFuncAttributes a{};
        int nb = 0;
        r.ok = true;
        auto q = [&](const void* fn, unsigned& regs, unsigned& lds, unsigned* wg) {
            if (hipFuncGetAttributes(&a, fn) != hipSuccess) { r.ok = false; (void)hipGetLastError(); return; }
            regs = (unsigned)a.numRegs; lds = (unsigned)a.sharedSizeBytes;
            if (wg) {
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64, 0) != hipSuccess) { r.ok = false; (void)hipGetLastError(); return; }
                *wg = (unsigned)nb;
            }
        };
        unsigned dummy = 0;
        q((const void*)rans_encode_kernel<true>, r.enc_regs, r.enc_lds, &r.enc_wg_per_cu);
        q((const void*)rans_decode_kernel<true>, r.dec_regs, r.dec_lds, &r.dec_wg_per_cu);
        q((const void*)rans_encode_kernel<false>, dummy, r.enc_lds_plain, nullptr);
        q((const void*)rans_decode_kernel<false>, dummy, r.dec_lds_plain, nullptr);
        return r;
    }();
    return f;
}
bool chain_kernel_occupancy(uint32_t out[6]) {
    const ChainKernelFacts& f = chain_kernel_facts();
    out[0] = f.enc_regs; out[1] = f.enc_lds; out[2] = f.enc_wg_per_cu;
    out[3] = f.dec_regs; out[4] = f.dec_lds; out[5] = f.dec_wg_per_cu;
    return f.ok;
}

void launch_rans_encode(const uint8_t* d_sym, uint64_t sym_stride, uint64_t n, const RansTable* d_tables,
                        uint8_t* d_out, uint64_t cap, RansResult* d_results, int n_chains, hipStream_t st,
                        uint64_t group_stride, uint64_t group_head, unsigned n_split, uint64_t cap_co, uint64_t cap_cg,
                        uint32_t x_init, bool keep_open) {
    if (n_chains <= 0) return;
    auto kern = n_chains <= 1024 ? rans_encode_kernel<true> : rans_encode_kernel<false>;
    const ChainKernelFacts& facts = chain_kernel_facts();   // real static LDS of the plain instance (9216 if the query failed)
    hipLaunchKernelGGL(kern, dim3(n_chains), dim3(64), chain_lds_pad(n_chains, facts.ok ? facts.enc_lds_plain : 9216u), st, d_sym,
                       (unsigned long long)sym_stride, (unsigned long long)n, d_tables, d_out,
                       (unsigned long long)cap, (unsigned long long)group_stride, (unsigned long long)group_head, n_split,
                       d_results, (unsigned long long)(cap_co ? cap_co : cap), (unsigned long long)(cap_cg ? cap_cg : cap),
                       x_init, keep_open ? 1u : 0u, chain_turns_enabled(), (const RansEncodeDesc*)nullptr);
}

void launch_rans_encode_descs(const RansEncodeDesc* d_descs, int n_chains, hipStream_t st) {
    if (n_chains <= 0) return;
    auto kern = n_chains <= 1024 ? rans_encode_kernel<true> : rans_encode_kernel<false>;
    const ChainKernelFacts& facts = chain_kernel_facts();
    hipLaunchKernelGGL(kern, dim3(n_chains), dim3(64), chain_lds_pad(n_chains, facts.ok ? facts.enc_lds_plain : 9216u), st,
                       (const uint8_t*)nullptr, 0ull, 0ull, (const RansTable*)nullptr, (uint8_t*)nullptr, 0ull, 0ull, 0ull, 0xFFFFFFFFu,
                       (RansResult*)nullptr, 0ull, 0ull, kRansL, 0u, chain_turns_enabled(), d_descs);
}

void launch_rans_decode(const RansDecodeDesc* d_descs, RansResult* d_results, int n_chains, hipStream_t st) {
    if (n_chains <= 0) return;
    auto kern = n_chains <= 1024 ? rans_decode_kernel<true> : rans_decode_kernel<false>;
    const ChainKernelFacts& facts = chain_kernel_facts();
    hipLaunchKernelGGL(kern, dim3(n_chains), dim3(64), chain_lds_pad(n_chains, facts.ok ? facts.dec_lds_plain : 21776u + 256u), st, d_descs, d_results,
                       chain_turns_enabled());
}

}  // namespace alice
