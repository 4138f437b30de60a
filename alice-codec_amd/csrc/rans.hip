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
    }
}

// ----------------------------------------------------------------------------------
// Decode chain (v1: one active lane, slot table in LDS, stream and symbols staged
// through LDS by the whole wavefront)
// ----------------------------------------------------------------------------------

constexpr int kDecTile = 1024;   // symbols per staged output tile
constexpr int kDecWin = 8192;    // stream window bytes in LDS

__global__ __launch_bounds__(256) void rans_decode_kernel(const RansDecodeDesc* __restrict__ descs,
                                                          RansResult* __restrict__ results) {
    __shared__ uint32_t slot_tab[kProbScale];  // sym | (freq-1) << 8 | (slot-cum) << 20
    __shared__ uint32_t scratch[256];
    __shared__ __attribute__((aligned(16))) uint8_t win[kDecWin];
    __shared__ __attribute__((aligned(16))) uint8_t obuf[kDecTile];
    __shared__ uint32_t sh_state, sh_cnt, sh_flags;
    __shared__ unsigned long long sh_pos;

    const RansDecodeDesc d = descs[blockIdx.x];
    const int tid = threadIdx.x;

    // table rebuilt from the stored histogram by rans_table_kernel (src/pipeline.rs:582)
    const uint32_t f = d.table->enc[tid].freq, c = d.table->enc[tid].cum;
    if (tid == 0) sh_flags = 0u;
    // default entries: cum_to_sym is zero-initialised (src/rans.rs:135) -> symbol 0
    {
        scratch[tid] = f | (c << 16);
        __syncthreads();
        const uint32_t f0 = scratch[0] & 0xFFFFu, c0 = scratch[0] >> 16;
        for (int s = tid; s < (int)kProbScale; s += 256) {
            const uint32_t bias = (uint32_t)s - c0;
            if (f0 - 1u >= kProbScale || bias >= kProbScale) atomicOr(&sh_flags, kRansInternal);
            slot_tab[s] = 0u | (((f0 - 1u) & 0xFFFu) << 8) | ((bias & 0xFFFu) << 20);
        }
        __syncthreads();
        // symbols in index order; ranges are disjoint (cum is a running sum)
        const uint32_t start = c;
        uint32_t end = c + f;
        if (end > kProbScale) end = kProbScale;
        for (uint32_t s = start; s < end; ++s)
            slot_tab[s] = (uint32_t)tid | (((f - 1u) & 0xFFFu) << 8) | (((s - c) & 0xFFFu) << 20);
        if (start < end && (f - 1u) >= kProbScale) atomicOr(&sh_flags, kRansInternal);
    }
    __syncthreads();

    // RansDecoder::new (src/rans.rs:330-347)
    uint32_t x = 0u;
    unsigned long long pos = 0ull;
    const unsigned long long len = d.in_len;
    if (len >= 4ull) {
        x = ((uint32_t)d.in[0] << 24) | ((uint32_t)d.in[1] << 16) | ((uint32_t)d.in[2] << 8) | (uint32_t)d.in[3];
        pos = 4ull;
    }
    bool pending = false;              // renormalisation owed by the previous symbol
    unsigned long long wbase = 0ull;   // stream offset of win[0]
    bool win_valid = false;
    unsigned long long done = 0ull;

    while (done < d.n) {
        // make sure the window covers [pos, pos + 2*tile + slack) when the stream has that much
        if (!win_valid || pos - wbase + 2ull * kDecTile + 16ull > (unsigned long long)kDecWin) {
            __syncthreads();
            wbase = pos & ~15ull;
            for (int i = tid * 16; i < kDecWin; i += 256 * 16) {
                const unsigned long long o = wbase + (unsigned long long)i;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (o + 16ull <= len && ((((uintptr_t)(d.in + o)) & 15u) == 0u)) {
                    v = *(const uint4*)(d.in + o);
                } else {
                    uint32_t w[4] = {0u, 0u, 0u, 0u};
                    for (int b = 0; b < 16; ++b)
                        if (o + (unsigned long long)b < len) w[b >> 2] |= (uint32_t)d.in[o + b] << (8 * (b & 3));
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                *(uint4*)(win + i) = v;
            }
            win_valid = true;
            __syncthreads();
        }
        const unsigned long long remain = d.n - done;
        const uint32_t want = remain < (unsigned long long)kDecTile ? (uint32_t)remain : (uint32_t)kDecTile;
        if (tid == 0) {
            uint32_t j = 0u;
            uint32_t xs = x;
            unsigned long long ps = pos;
            bool pend = pending;
            const unsigned long long wend = wbase + (unsigned long long)kDecWin;
            bool starved = false;
            while (j < want) {
                if (pend) {  // src/rans.rs:365-368
                    while (xs < kRansL && ps < len) {
                        if (ps >= wend) { starved = true; break; }
                        xs = (xs << 8) | (uint32_t)win[ps - wbase];
                        ps += 1ull;
                    }
                    if (starved) break;
                }
                const uint32_t slot = xs & (kProbScale - 1u);      // :353
                const uint32_t e = slot_tab[slot];                 // :356
                const uint32_t f1 = (e >> 8) & 0xFFFu, bias = e >> 20;
                const uint32_t hq = xs >> kProbBits;
                xs = f1 * hq + hq + bias;                          // :361-362 (mod 2^32)
                obuf[j] = (uint8_t)(e & 0xFFu);
                pend = true;
                ++j;
            }
            sh_state = xs;
            sh_pos = ps;
            sh_cnt = j;
            pending = pend;
        }
        __syncthreads();
        x = sh_state;
        pos = sh_pos;
        const uint32_t got = sh_cnt;
        pending = true;
        if (got < want) win_valid = false;  // window ran dry mid-tile: reload at pos and go on
        // flush the decoded symbols
        for (uint32_t i = tid; i < got; i += 256) d.out[done + i] = obuf[i];
        done += got;
        __syncthreads();
    }
    if (tid == 0) {
        results[blockIdx.x].len = pos;
        results[blockIdx.x].flags = sh_flags;
        results[blockIdx.x].final_state = x;
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
    hipLaunchKernelGGL(rans_decode_kernel, dim3(n_chains), dim3(256), 0, st, d_descs, d_results);
}

}  // namespace alice
