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
#include "common.h"
#include "kernels.h"

namespace alice {

// ----------------------------------------------------------------------------------
// Frequency table (device function shared by the table kernel and the decoder)
// ----------------------------------------------------------------------------------

// Computes the reference table for a 256-bin histogram.  Must be called by the first
// 256 threads of a block (all of them), s = threadIdx.x; scratch: 256+ u32 in LDS.
// Returns freq/cum as the reference stores them (u16 truncated).
__device__ inline void freq_table_256(uint32_t count, uint32_t* scratch, uint32_t& freq16, uint32_t& cum16) {
    const int s = threadIdx.x;
    __shared__ unsigned long long total_sh;
    if (s == 0) total_sh = 0ull;
    __syncthreads();
    atomicAdd(&total_sh, (unsigned long long)count);
    __syncthreads();
    const unsigned long long total = total_sh;
    uint32_t freq;
    if (total == 0ull) {
        freq = kProbScale / 256u;  // uniform(256): src/rans.rs:159-166
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
        if (s == 255) freq = (kProbScale - cum) & 0xFFFFu;
    } else if (s == 255 && nt != kProbScale) {
        // src/rans.rs:128-132: wrapping cast to u16
        int32_t diff = (int32_t)kProbScale - (int32_t)nt;
        freq = (uint32_t)((int32_t)freq + diff) & 0xFFFFu;
    }
    freq16 = freq & 0xFFFFu;
    cum16 = cum & 0xFFFFu;
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

__global__ __launch_bounds__(256) void rans_table_kernel(const uint32_t* __restrict__ hist,
                                                         RansTable* __restrict__ tables) {
    __shared__ uint32_t scratch[256];
    const int chain = blockIdx.x;
    const int s = threadIdx.x;
    const uint32_t count = hist[(size_t)chain * 256 + s];
    uint32_t f, c;
    freq_table_256(count, scratch, f, c);
    tables[chain].enc[s] = make_enc_entry(f, c);
    uint32_t fl = 0u;
    if (count > 0u && f == 0u) fl |= kTableDiverges;
    if (count > 0u && f > kProbScale) fl |= kTableNeedsGeneric;
    __shared__ uint32_t flags_sh;
    if (s == 0) flags_sh = 0u;
    __syncthreads();
    if (fl) atomicOr(&flags_sh, fl);
    __syncthreads();
    if (s == 0) tables[chain].flags = flags_sh;
}

// Table from explicit (cum, freq) arrays -- stage-level API (FrequencyTable handle).
__global__ __launch_bounds__(256) void rans_table_from_arrays_kernel(const uint16_t* __restrict__ cum,
                                                                     const uint16_t* __restrict__ freq,
                                                                     RansTable* __restrict__ table) {
    const int s = threadIdx.x;
    table->enc[s] = make_enc_entry(freq[s], cum[s]);
    if (s == 0) table->flags = 0u;
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
    uint32_t k, y, q, cnt;
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
    asm volatile(
        "s_mov_b32 %[cnt], 16\n\t"
        "s_nop 1\n\t"
        "1:\n\t"
        ALICE_RIPPLE_STEP("s_nop 1\n\t")
        ALICE_RIPPLE_STEP("s_nop 1\n\t")
        ALICE_RIPPLE_STEP("s_nop 1\n\t")
        ALICE_RIPPLE_STEP("s_sub_u32 %[cnt], %[cnt], 1\n\ts_cmp_lg_u32 %[cnt], 0\n\ts_cbranch_scc1 1b\n\t")
        "s_nop 1\n\t"
        : [xin] "+v"(xin), [xout] "+v"(xout), [k] "=&v"(k), [y] "=&v"(y), [q] "=&v"(q),
          [cnt] "=&s"(cnt), [m1] "=&s"(m1)
        : [xmax] "v"(xmax), [xmax8] "v"(xmax8), [rcp] "v"(rcp), [rsh] "v"(rsh), [g] "v"(g),
          [cprev] "v"(cprev)
        : "vcc", "scc");
#undef ALICE_RIPPLE_STEP
}

__global__ __launch_bounds__(64) void rans_encode_kernel(const uint8_t* __restrict__ sym_base,
                                                         unsigned long long sym_stride,
                                                         unsigned long long n,
                                                         const RansTable* __restrict__ tables,
                                                         uint8_t* __restrict__ out_base,
                                                         unsigned long long cap,
                                                         RansResult* __restrict__ results) {
    __shared__ uint4 tab_a[256];  // xmax, xmax8, rcp, rsh
    __shared__ uint4 tab_b[256];  // g, cbias, freq, cum
    __shared__ __attribute__((aligned(16))) uint8_t tile[kEncTile];

    const int chain = blockIdx.x;
    const int lane = threadIdx.x;
    const uint8_t* __restrict__ sym = sym_base + (size_t)chain * sym_stride;
    const RansTable* __restrict__ tbl = tables + chain;
    uint8_t* const out_end = out_base + (size_t)chain * cap + cap;

    for (int s = lane; s < 256; s += 64) {
        const RansEncEntry e = tbl->enc[s];
        tab_a[s] = make_uint4(e.xmax, e.xmax8, e.rcp, e.rsh);
        tab_b[s] = make_uint4((uint32_t)e.g, e.cbias, e.freq, e.cum);
    }

    uint32_t x = kRansL;  // RansEncoder::new, src/rans.rs:249-254
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
            if ((((uintptr_t)p) & 3u) == 0u) {
                const uint32_t* p4 = (const uint32_t*)p;
                v = make_uint4(p4[0], p4[1], p4[2], p4[3]);
            } else {
                uint32_t w[4];
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    w[d] = (uint32_t)p[4 * d] | ((uint32_t)p[4 * d + 1] << 8) | ((uint32_t)p[4 * d + 2] << 16) |
                           ((uint32_t)p[4 * d + 3] << 24);
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

    uint4 cur = load_tile(0);
    for (unsigned long long j = 0; j < ntiles; ++j) {
        __syncthreads();
        ((uint4*)tile)[lane] = cur;
        __syncthreads();
        cur = load_tile(j + 1);  // in flight while this tile's 16 blocks ripple
        const long long hi = (long long)(n - j * kEncTile);
        const int valid = hi >= kEncTile ? kEncTile : (int)hi;  // tile-local indices [1024-valid, 1024)

        for (int b = 0; b < kEncTile / 64; ++b) {
            const int local = kEncTile - 1 - (b * 64 + lane);  // descending symbol order
            const bool active = local >= kEncTile - valid;
            if (__ballot(active) == 0ull) break;
            const uint32_t s = tile[local];
            uint4 ea = tab_a[s];
            uint4 eb = tab_b[s];
            if (!active) {  // identity step
                ea = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
                eb = make_uint4(0u, 0u, 1u, 0u);
            }
            const uint32_t xmax = ea.x, xmax8 = ea.y, rcp = ea.z, rsh = ea.w;
            const int32_t g = (int32_t)eb.x;
            const uint32_t cbias = eb.y, freq = eb.z, cum = eb.w;
            const bool bad = active && (freq == 0u || freq > kProbScale);

            uint32_t xin = x, xout = 0u;
            if (__ballot(bad) == 0ull) {
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

            // byte emission: lane i pushes the low k_i bytes of its pre-renormalisation state
            const bool c1 = active && xin >= xmax;
            const bool c2 = active && xin >= xmax8;
            const unsigned long long b1 = __ballot(c1), b2 = __ballot(c2);
            const uint32_t off = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u)) +
                                 __builtin_amdgcn_mbcnt_hi((uint32_t)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b2, 0u));
            const uint32_t total = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
            if (written + total + 4ull > cap) {
                flags |= kRansOverflow;
            } else {
                uint8_t* p = out_end - 1 - (written + off);
                if (c1) p[0] = (uint8_t)(xin & 0xFFu);
                if (c2) p[-1] = (uint8_t)((xin >> 8) & 0xFFu);
            }
            written += total;
        }
    }

    // finish (src/rans.rs:298-308): push the 4 state bytes LSB first; the reversal is implicit
    if (written + 4ull <= cap) {
        if (lane < 4) out_end[-1 - (long long)(written + lane)] = (uint8_t)((x >> (8 * lane)) & 0xFFu);
    } else {
        flags |= kRansOverflow;
    }
    written += 4ull;
    if (lane == 0) {
        results[chain].len = written;
        results[chain].flags = flags;
        results[chain].final_state = x;
        results[chain].fast_tiles = 0u;
        results[chain].slow_tiles = 0u;
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

constexpr int kDecBlocks = 16;                 // 64-symbol blocks per fast tile
constexpr int kDecTile = kDecBlocks * 64;      // 1024 symbols
constexpr int kDecWinDwords = 9 * 64;          // window VGPR bank: 9 registers x 64 lanes
constexpr int kDecWinBytes = kDecWinDwords * 4;

// literal registers owned by the fast-tile asm statement (all listed as clobbers)
//   v[64:127]  slot table        v[128:136] stream window      v137 scratch   v138 record
//   s[60:61]   {xl : x}          s[62:63]   byte window WW     s[64:65] refill pair
//   s66 h  s67 e  s68 f  s69 b  s70 t  s71 sh  s72 av  s73 di  s74 nblk  s75 selector  s76 row
// Measured on MI355X (scripts/probes/latency_probe.hip): every SALU/VALU instruction of a lone wave
// costs ~4 cycles, a v_readlane result reaches the scalar unit ~24 cycles after issue, an untaken
// branch costs ~10 cycles and a taken one ~25.  Hence: the shift and the record sit in the readlane's
// shadow, and the refill test runs once per two symbols (a 64-bit window holds 2 x 16 bits of slack).
#define ALICE_DEC_CORE(LANE)                                                       \
    "s_bfe_u32 s76, s61, 0x60006\n\t"                                              \
    "s_set_gpr_idx_idx s76\n\t"                                                    \
    "v_readlane_b32 s67, v64, s61\n\t"                                             \
    "s_lshr_b32 s66, s61, 12\n\t"                                                  \
    "v_writelane_b32 v138, s61, " #LANE "\n\t"                                     \
    "s_and_b32 s68, s67, 0xffff\n\t"                                               \
    "s_lshr_b32 s69, s67, 16\n\t"                                                  \
    "s_mul_i32 s70, s68, s66\n\t"                                                  \
    "s_add_u32 s61, s70, s69\n\t"                                                  \
    "s_cmp_lt_u32 s61, 0x800000\n\t"                                               \
    "s_cbranch_scc0 10" #LANE "f\n\t"                                              \
    "s_cmp_lt_u32 s61, 0x8000\n\t"                                                 \
    "s_cselect_b32 s71, 16, 8\n\t"                                                 \
    "s_mov_b32 s60, s63\n\t"                                                       \
    "s_lshl_b64 s[60:61], s[60:61], s71\n\t"                                       \
    "s_lshl_b64 s[62:63], s[62:63], s71\n\t"                                       \
    "s_sub_u32 s72, s72, s71\n\t"                                                  \
    "10" #LANE ":\n\t"

#define ALICE_DEC_REFILL(LANE)                                                     \
    "s_cmp_lt_u32 s72, 33\n\t"                                                     \
    "s_cbranch_scc0 11" #LANE "f\n\t"                                              \
    "s_lshr_b32 s76, s73, 6\n\t"                                                   \
    "s_set_gpr_idx_idx s76\n\t"                                                    \
    "v_readlane_b32 s65, v128, s73\n\t"                                            \
    "s_mov_b32 s64, 0\n\t"                                                         \
    "s_add_u32 s73, s73, 1\n\t"                                                    \
    "s_lshr_b64 s[64:65], s[64:65], s72\n\t"                                       \
    "s_or_b64 s[62:63], s[62:63], s[64:65]\n\t"                                    \
    "s_add_u32 s72, s72, 32\n\t"                                                   \
    "11" #LANE ":\n\t"

#define ALICE_DEC_SYM2(A, B) ALICE_DEC_CORE(A) ALICE_DEC_CORE(B) ALICE_DEC_REFILL(B)
#define ALICE_DEC_SYM4(A, B, C_, D) ALICE_DEC_SYM2(A, B) ALICE_DEC_SYM2(C_, D)

#define ALICE_DS_ROW(R, BASE, ADDR) "ds_read_b32 v" #R ", " ADDR " offset:" #BASE "\n\t"

// Decodes nblk*64 symbols on the fast path.  tab_addr / win_addr / rec_addr are this lane's LDS
// byte addresses (base + 4*lane).  pos = byte offset of the next stream byte inside the window.
__device__ __forceinline__ void dec_tile_fast(uint32_t& x, uint32_t& pos, uint32_t tab_addr, uint32_t win_addr,
                                              uint32_t rec_addr, uint32_t nblk) {
    uint32_t xo, po;
    asm volatile(
        // ---- slot table -> v[64:127] ----
        "ds_read_b32 v64, %[ta] offset:0\n\t"     "ds_read_b32 v65, %[ta] offset:256\n\t"
        "ds_read_b32 v66, %[ta] offset:512\n\t"   "ds_read_b32 v67, %[ta] offset:768\n\t"
        "ds_read_b32 v68, %[ta] offset:1024\n\t"  "ds_read_b32 v69, %[ta] offset:1280\n\t"
        "ds_read_b32 v70, %[ta] offset:1536\n\t"  "ds_read_b32 v71, %[ta] offset:1792\n\t"
        "ds_read_b32 v72, %[ta] offset:2048\n\t"  "ds_read_b32 v73, %[ta] offset:2304\n\t"
        "ds_read_b32 v74, %[ta] offset:2560\n\t"  "ds_read_b32 v75, %[ta] offset:2816\n\t"
        "ds_read_b32 v76, %[ta] offset:3072\n\t"  "ds_read_b32 v77, %[ta] offset:3328\n\t"
        "ds_read_b32 v78, %[ta] offset:3584\n\t"  "ds_read_b32 v79, %[ta] offset:3840\n\t"
        "ds_read_b32 v80, %[ta] offset:4096\n\t"  "ds_read_b32 v81, %[ta] offset:4352\n\t"
        "ds_read_b32 v82, %[ta] offset:4608\n\t"  "ds_read_b32 v83, %[ta] offset:4864\n\t"
        "ds_read_b32 v84, %[ta] offset:5120\n\t"  "ds_read_b32 v85, %[ta] offset:5376\n\t"
        "ds_read_b32 v86, %[ta] offset:5632\n\t"  "ds_read_b32 v87, %[ta] offset:5888\n\t"
        "ds_read_b32 v88, %[ta] offset:6144\n\t"  "ds_read_b32 v89, %[ta] offset:6400\n\t"
        "ds_read_b32 v90, %[ta] offset:6656\n\t"  "ds_read_b32 v91, %[ta] offset:6912\n\t"
        "ds_read_b32 v92, %[ta] offset:7168\n\t"  "ds_read_b32 v93, %[ta] offset:7424\n\t"
        "ds_read_b32 v94, %[ta] offset:7680\n\t"  "ds_read_b32 v95, %[ta] offset:7936\n\t"
        "ds_read_b32 v96, %[ta] offset:8192\n\t"  "ds_read_b32 v97, %[ta] offset:8448\n\t"
        "ds_read_b32 v98, %[ta] offset:8704\n\t"  "ds_read_b32 v99, %[ta] offset:8960\n\t"
        "ds_read_b32 v100, %[ta] offset:9216\n\t" "ds_read_b32 v101, %[ta] offset:9472\n\t"
        "ds_read_b32 v102, %[ta] offset:9728\n\t" "ds_read_b32 v103, %[ta] offset:9984\n\t"
        "ds_read_b32 v104, %[ta] offset:10240\n\t" "ds_read_b32 v105, %[ta] offset:10496\n\t"
        "ds_read_b32 v106, %[ta] offset:10752\n\t" "ds_read_b32 v107, %[ta] offset:11008\n\t"
        "ds_read_b32 v108, %[ta] offset:11264\n\t" "ds_read_b32 v109, %[ta] offset:11520\n\t"
        "ds_read_b32 v110, %[ta] offset:11776\n\t" "ds_read_b32 v111, %[ta] offset:12032\n\t"
        "ds_read_b32 v112, %[ta] offset:12288\n\t" "ds_read_b32 v113, %[ta] offset:12544\n\t"
        "ds_read_b32 v114, %[ta] offset:12800\n\t" "ds_read_b32 v115, %[ta] offset:13056\n\t"
        "ds_read_b32 v116, %[ta] offset:13312\n\t" "ds_read_b32 v117, %[ta] offset:13568\n\t"
        "ds_read_b32 v118, %[ta] offset:13824\n\t" "ds_read_b32 v119, %[ta] offset:14080\n\t"
        "ds_read_b32 v120, %[ta] offset:14336\n\t" "ds_read_b32 v121, %[ta] offset:14592\n\t"
        "ds_read_b32 v122, %[ta] offset:14848\n\t" "ds_read_b32 v123, %[ta] offset:15104\n\t"
        "ds_read_b32 v124, %[ta] offset:15360\n\t" "ds_read_b32 v125, %[ta] offset:15616\n\t"
        "ds_read_b32 v126, %[ta] offset:15872\n\t" "ds_read_b32 v127, %[ta] offset:16128\n\t"
        // ---- stream window -> v[128:136], bytes swapped to big-endian ----
        "ds_read_b32 v128, %[wa] offset:0\n\t"    "ds_read_b32 v129, %[wa] offset:256\n\t"
        "ds_read_b32 v130, %[wa] offset:512\n\t"  "ds_read_b32 v131, %[wa] offset:768\n\t"
        "ds_read_b32 v132, %[wa] offset:1024\n\t" "ds_read_b32 v133, %[wa] offset:1280\n\t"
        "ds_read_b32 v134, %[wa] offset:1536\n\t" "ds_read_b32 v135, %[wa] offset:1792\n\t"
        "ds_read_b32 v136, %[wa] offset:2048\n\t"
        "s_mov_b32 s75, 0x00010203\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_perm_b32 v128, v128, v128, s75\n\t" "v_perm_b32 v129, v129, v129, s75\n\t"
        "v_perm_b32 v130, v130, v130, s75\n\t" "v_perm_b32 v131, v131, v131, s75\n\t"
        "v_perm_b32 v132, v132, v132, s75\n\t" "v_perm_b32 v133, v133, v133, s75\n\t"
        "v_perm_b32 v134, v134, v134, s75\n\t" "v_perm_b32 v135, v135, v135, s75\n\t"
        "v_perm_b32 v136, v136, v136, s75\n\t"
        // ---- scalar state: x, 64-bit byte window primed with two dwords ----
        "s_mov_b32 s61, %[xi]\n\t"
        "s_mov_b32 s74, %[nb]\n\t"
        "s_lshr_b32 s73, %[pi], 2\n\t"            // di = pos / 4
        "s_and_b32 s71, %[pi], 3\n\t"
        "s_lshl_b32 s71, s71, 3\n\t"              // bits to drop from the first dword
        "s_lshr_b32 s76, s73, 6\n\t"
        "s_set_gpr_idx_on s76, 0x1\n\t"          // VGPR-index mode, src0 relative, stays on for the whole tile
        "s_nop 1\n\t"
        "v_readlane_b32 s63, v128, s73\n\t"
        "s_add_u32 s73, s73, 1\n\t"
        "s_lshr_b32 s76, s73, 6\n\t"
        "s_set_gpr_idx_idx s76\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 s62, v128, s73\n\t"
        "s_add_u32 s73, s73, 1\n\t"
        "s_lshl_b64 s[62:63], s[62:63], s71\n\t"
        "s_sub_u32 s72, 64, s71\n\t"              // valid bits in the window
        "2:\n\t"
        ALICE_DEC_SYM4(0, 1, 2, 3) ALICE_DEC_SYM4(4, 5, 6, 7) ALICE_DEC_SYM4(8, 9, 10, 11) ALICE_DEC_SYM4(12, 13, 14, 15)
        ALICE_DEC_SYM4(16, 17, 18, 19) ALICE_DEC_SYM4(20, 21, 22, 23) ALICE_DEC_SYM4(24, 25, 26, 27) ALICE_DEC_SYM4(28, 29, 30, 31)
        ALICE_DEC_SYM4(32, 33, 34, 35) ALICE_DEC_SYM4(36, 37, 38, 39) ALICE_DEC_SYM4(40, 41, 42, 43) ALICE_DEC_SYM4(44, 45, 46, 47)
        ALICE_DEC_SYM4(48, 49, 50, 51) ALICE_DEC_SYM4(52, 53, 54, 55) ALICE_DEC_SYM4(56, 57, 58, 59) ALICE_DEC_SYM4(60, 61, 62, 63)
        "ds_write_b32 %[ra], v138\n\t"
        "v_add_u32_e32 %[ra], 0x100, %[ra]\n\t"
        "s_sub_u32 s74, s74, 1\n\t"
        "s_cmp_lg_u32 s74, 0\n\t"
        "s_cbranch_scc1 2b\n\t"
        "s_set_gpr_idx_off\n\t"
        // ---- results: state, consumed position = 4*di - valid_bits/8 ----
        "s_lshl_b32 s70, s73, 2\n\t"
        "s_lshr_b32 s71, s72, 3\n\t"
        "s_sub_u32 %[po], s70, s71\n\t"
        "s_mov_b32 %[xo], s61\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [xo] "=&s"(xo), [po] "=&s"(po), [ra] "+v"(rec_addr)
        : [xi] "s"(x), [pi] "s"(pos), [nb] "s"(nblk), [ta] "v"(tab_addr), [wa] "v"(win_addr)
        : "memory", "scc", "m0",
          "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76",
          "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
          "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
          "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
          "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
          "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137",
          "v138");
    x = xo;
    pos = po;
}

__global__ __launch_bounds__(64) void rans_decode_kernel(const RansDecodeDesc* __restrict__ descs,
                                                         RansResult* __restrict__ results) {
    __shared__ uint32_t slot_tab[kProbScale];                         // freq | (slot - cum) << 16
    __shared__ uint8_t c2s[kProbScale];                               // cum_to_sym
    __shared__ __attribute__((aligned(16))) uint8_t win[kDecWinBytes];
    __shared__ uint32_t rec[kDecTile];                                // pre-update states of a fast tile
    __shared__ __attribute__((aligned(16))) uint8_t obuf[kDecTile];

    const RansDecodeDesc d = descs[blockIdx.x];
    const int lane = threadIdx.x;
    uint32_t flags = 0u;

    // cum_to_sym is zero-initialised (src/rans.rs:135): default every slot to symbol 0
    {
        const uint32_t f0 = d.table->enc[0].freq, c0 = d.table->enc[0].cum;
        for (int s = lane; s < (int)kProbScale; s += 64) {
            c2s[s] = 0;
            slot_tab[s] = (f0 & 0xFFFFu) | ((((uint32_t)s - c0) & 0xFFFFu) << 16);
        }
        __syncthreads();
        // symbols in index order; their slot ranges are disjoint (cum is a running sum), src/rans.rs:136-144
        for (int k = 0; k < 4; ++k) {
            const int sym = lane * 4 + k;
            const uint32_t f = d.table->enc[sym].freq, c = d.table->enc[sym].cum;
            uint32_t end = c + f;
            if (end > kProbScale) end = kProbScale;
            for (uint32_t s = c; s < end; ++s) {
                c2s[s] = (uint8_t)sym;
                slot_tab[s] = f | ((s - c) << 16);
            }
        }
        __syncthreads();
    }

    // RansDecoder::new (src/rans.rs:330-347)
    uint32_t x = 0u;
    unsigned long long pos = 0ull;
    const unsigned long long len = d.in_len;
    if (len >= 4ull) {
        x = ((uint32_t)d.in[0] << 24) | ((uint32_t)d.in[1] << 16) | ((uint32_t)d.in[2] << 8) | (uint32_t)d.in[3];
        pos = 4ull;
    }
    bool pending = false;  // renormalisation owed by the previous symbol
    unsigned long long done = 0ull;
    uint32_t n_fast = 0u, n_slow = 0u;

    while (done < d.n) {
        const unsigned long long remain = d.n - done;
        const uint32_t want = remain < (unsigned long long)kDecTile ? (uint32_t)remain : (uint32_t)kDecTile;
        // stage the stream window [wbase, wbase + 2304) (zero beyond len)
        const unsigned long long wbase = pos & ~3ull;
        __syncthreads();
        for (int i = lane * 4; i < kDecWinBytes; i += 64 * 4) {
            const unsigned long long o = wbase + (unsigned long long)i;
            uint32_t w = 0u;
            if (o + 4ull <= len && ((((uintptr_t)(d.in + o)) & 3u) == 0u)) {
                w = *(const uint32_t*)(d.in + o);
            } else {
                for (int b = 0; b < 4; ++b)
                    if (o + (unsigned long long)b < len) w |= (uint32_t)d.in[o + b] << (8 * b);
            }
            *(uint32_t*)(win + i) = w;
        }
        __syncthreads();

        // the owed renormalisation first (src/rans.rs:365-368); it needs at most a few window bytes
        // unless the state is 0, which the exact loop below handles byte by byte
        uint32_t got = 0u;
        bool fast = (want == (uint32_t)kDecTile) && (len - wbase >= (unsigned long long)kDecWinBytes);
        if (fast && pending) {
            uint32_t xs = x;
            unsigned long long ps = pos;
            int guard = 0;
            while (xs < kRansL && ps < len && guard < 8) { xs = (xs << 8) | (uint32_t)win[ps - wbase]; ps += 1ull; ++guard; }
            if (xs < kRansL) fast = false;  // still starved: leave x/pos untouched for the exact loop
            else { x = xs; pos = ps; pending = false; }
        }
        if (fast && x < kRansL) fast = false;  // only the very first symbol of a malformed stream
        if (fast) {
            uint32_t prel = (uint32_t)(pos - wbase);
            uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
            prel = (uint32_t)__builtin_amdgcn_readfirstlane((int)prel);
            dec_tile_fast(xs, prel, (uint32_t)(uintptr_t)slot_tab + 4u * lane, (uint32_t)(uintptr_t)win + 4u * lane,
                          (uint32_t)(uintptr_t)rec + 4u * lane, (uint32_t)kDecBlocks);
            x = xs;
            pos = wbase + prel;
            pending = false;  // the fast path renormalises right after each update
            __syncthreads();
            for (int i = lane; i < kDecTile; i += 64) obuf[i] = c2s[rec[i] & (kProbScale - 1u)];
            got = (uint32_t)kDecTile;
            ++n_fast;
        } else {
            ++n_slow;
            // exact scalar-lane loop over what the window can feed
            __shared__ uint32_t sh_state, sh_cnt;
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
                    const uint32_t slot = xs & (kProbScale - 1u);
                    const uint32_t e = slot_tab[slot];
                    xs = (e & 0xFFFFu) * (xs >> kProbBits) + (e >> 16);
                    obuf[j] = c2s[slot];
                    pend = true;
                    ++j;
                }
                sh_state = xs; sh_pos = ps; sh_cnt = j;
            }
            __syncthreads();
            x = sh_state; pos = sh_pos; got = sh_cnt;
            pending = true;
        }
        __syncthreads();
        for (uint32_t i = lane; i < got; i += 64) d.out[done + i] = obuf[i];
        done += got;
    }
    if (lane == 0) {
        results[blockIdx.x].len = pos;
        results[blockIdx.x].flags = flags;
        results[blockIdx.x].final_state = x;
        results[blockIdx.x].fast_tiles = n_fast;
        results[blockIdx.x].slow_tiles = n_slow;
    }
}

// ----------------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------------

void launch_rans_table(const uint32_t* d_hist, RansTable* d_tables, int n_chains, hipStream_t st) {
    if (n_chains <= 0) return;
    hipLaunchKernelGGL(rans_table_kernel, dim3(n_chains), dim3(256), 0, st, d_hist, d_tables);
}

void launch_rans_table_from_arrays(const uint16_t* d_cum, const uint16_t* d_freq, RansTable* d_table,
                                   hipStream_t st) {
    hipLaunchKernelGGL(rans_table_from_arrays_kernel, dim3(1), dim3(256), 0, st, d_cum, d_freq, d_table);
}

void launch_rans_encode(const uint8_t* d_sym, uint64_t sym_stride, uint64_t n, const RansTable* d_tables,
                        uint8_t* d_out, uint64_t cap, RansResult* d_results, int n_chains, hipStream_t st) {
    if (n_chains <= 0) return;
    hipLaunchKernelGGL(rans_encode_kernel, dim3(n_chains), dim3(64), 0, st, d_sym,
                       (unsigned long long)sym_stride, (unsigned long long)n, d_tables, d_out,
                       (unsigned long long)cap, d_results);
}

void launch_rans_decode(const RansDecodeDesc* d_descs, RansResult* d_results, int n_chains, hipStream_t st) {
    if (n_chains <= 0) return;
    hipLaunchKernelGGL(rans_decode_kernel, dim3(n_chains), dim3(64), 0, st, d_descs, d_results);
}

}  // namespace alice
