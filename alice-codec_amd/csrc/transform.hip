// Pipeline-specialised transform kernels for gfx950: RGB <-> u8 symbols.
//
// Forward (FrameEncoder::encode, reference src/pipeline.rs:429-477), two roles per launch (fwd_band_kernel):
//   tile role     : per frame and 128x40 tile.  Stage A: a thread owns a 16-pixel row segment
//                   (+ one halo sample per lifting step on each side): RGB -> YCoCg-R
//                   (src/color.rs:221-228), edge-replicated padding by index clamping
//                   (src/pipeline.rs:77-114), row lifting entirely in registers
//                   (src/wavelet.rs:401-405).  The tile goes through LDS once (the transpose).
//                   Stage B: a thread owns one column of the tile (+halo rows) in registers, column
//                   lifting (:408-417), i16 store to the band slot.
//   temporal role : temporal lifting as a stream over frame pairs with a five-value window per
//                   pixel (src/wavelet.rs:421-437), 4 pixels per thread (8-byte loads, 4-byte stores),
//                   fused Quantizer::quantize (src/quant.rs:89-97), to_symbols (:555-560) and
//                   build_histogram (:594-600).  Any frame count.
// Inverse (FrameDecoder::decode, src/pipeline.rs:588-621) mirrors it (inv_band_kernel):
//   temporal role : from_symbols, dequantize, inverse temporal lifting (stream over pairs).
//   tile role     : stage A inverse column lifting (registers) -> LDS -> stage B inverse row lifting,
//                   `as i16`, ycocg_r_to_rgb_bytes.
// How the roles of consecutive bands share launches: "Band-ordered, role-fused launches" below.
//
// Lifting facts used: inside one lifting step every write depends only on samples of the other
// parity, so a step is data-parallel; an output depends on inputs within +-n_steps samples;
// mirroring happens only at the true signal ends (src/wavelet.rs:186-190,206-210).  A tile or
// segment whose halo is recomputed therefore yields the same integers as the global pass.
//
// Intermediates: the forward path stores i16 (from u8 input every value in the transform stays
// below 2^13 in magnitude, SURVEY.md section 7 hard part 5); the inverse path stores i32 because a
// desynchronised decoder feeds it arbitrary symbols (i16 when the host proves the bound).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace alice {

template <bool EXACT>
__device__ __forceinline__ int lift_delta(int a, int b, int c) {
    if (EXACT) {
        // src/wavelet.rs:193-194: wrapping i32 sum, i64 product and rounding, arithmetic shift
        const int avg = (int)((unsigned)a + (unsigned)b);
        return (int)(((long long)avg * (long long)c + 4096ll) >> 13);
    } else {
        // callers guarantee |a + b| < 2^23 and |(a + b) * c| < 2^31 (forward: u8 input; inverse: host bound
        // check), so the full-rate 24-bit multiply-add is exact.  Written as asm because the compiler, once it
        // has proven the operand ranges, rewrites __mul24 into a generic multiply and then selects the
        // quarter-rate v_mul_lo_u32 for a third of the products (seen in the ISA of every kernel here).
        int m;
        const int rnd = 4096;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(m) : "v"(a + b), "s"(c), "v"(rnd));
        return m >> 13;
    }
}
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }

struct Coeffs { int c[4]; };

// XCD-aware work order.  Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share an
// L2), while neighbouring tiles share halo rows/columns and 128-byte lines.  Launch a 1-D grid padded
// to a multiple of 8 and give XCD k the k-th contiguous eighth of the logical tile sequence, so a
// tile's neighbours hit in the same L2 (measured before: 2.0x / 2.4x HBM over-fetch in the forward /
// inverse tile kernels; placement affects speed only, never results).
__device__ __forceinline__ unsigned xcd_logical_block(unsigned total) {
    const unsigned b = blockIdx.x;
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned l = (b & 7u) * per_xcd + (b >> 3);
    return l < total ? l : 0xFFFFFFFFu;
}
static inline unsigned xcd_grid(unsigned long long total) { return (unsigned)(((total + 7) / 8) * 8); }

// Whole-sample symmetric extension of an even-length signal: x[-k] = x[k], x[n-1+k] = x[n-1-k].
// The reference mirrors one neighbour at the two ends of every lifting step (src/wavelet.rs:186-190,
// 206-210: the last odd sample predicts from its left neighbour twice, the first even sample updates from
// its right neighbour twice).  That is exactly what plain lifting computes on the symmetric extension:
// the extension is symmetric about an even sample on the left and an odd sample on the right, and every
// predict/update step maps such a signal to one with the same symmetry.  So a tile that LOADS through this
// index map needs no boundary logic in its arithmetic, and parity (low/high band) is preserved.
// One reflection is exact for every position within the halo (|distance| <= 4) of a signal of length >= 6
// (the launchers send shorter ones to the generic path); positions further out belong to tiles that overhang
// the frame, whose results are never stored: they only need a valid address, hence the clamp.
__device__ __forceinline__ int reflect_idx(int i, int n) {
    int r = i < 0 ? -i : i;
    r = r >= n ? 2 * (n - 1) - r : r;
    return min(max(r, 0), n - 1);
}

// ------------------------------------------------------------------------------------------------
// In-register 1-D lifting of N consecutive samples (v[0] has an even global index).  No boundary logic: the
// tile kernels load through the symmetric extension (reflect_idx), and the two array ends simply reuse their
// inner neighbour (that garbage stays inside the halo, which is never stored).  INVERSE applies the steps
// reversed with -coeff (src/wavelet.rs:167-174).
// ------------------------------------------------------------------------------------------------
template <int N, int NS, bool EXACT, bool INVERSE>
__device__ __forceinline__ void lift_regs(int (&v)[N], const Coeffs& cf) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k_step = INVERSE ? (NS - 1 - s) : s;
        const int c = INVERSE ? -cf.c[k_step] : cf.c[k_step];
        if ((k_step & 1) == 0) {  // predict: odd += d(even_left, even_right), src/wavelet.rs:184-196
#pragma unroll
            for (int k = 1; k < N; k += 2) v[k] = wadd(v[k], lift_delta<EXACT>(v[k - 1], (k + 1 < N) ? v[k + 1] : v[k - 1], c));
        } else {                  // update: even += d(odd_left, odd_right), src/wavelet.rs:205-216
#pragma unroll
            for (int k = 0; k < N; k += 2) {
                const int rgt = (k + 1 < N) ? v[k + 1] : v[k - 1];
                v[k] = wadd(v[k], lift_delta<EXACT>((k >= 1) ? v[k - 1] : rgt, rgt, c));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Bands
//
// Between the spatial (tile) pass and the temporal pass sits the i16 intermediate: 6 B/px, 796 MB per 1080p x 64 chunk,
// 12.7 GB per 8K x 64 chunk.  A chunk may therefore be cut into BANDS of whole tile rows (all frames of those rows): the
// tile pass and the temporal pass of one band run back to back on a band-sized slot, then the next band reuses the slot.
// A band's rows are [low-band rows | high-band rows] of the y-deinterleaved plane; inside the slot they are contiguous,
// so the temporal pass reads (writes) the slot linearly and maps a pixel to its place in the symbol plane through two
// segment bases.  The inverse band carries two halo rows of each half on either side (the temporal pass simply covers
// them: it is per pixel), so its tile pass depends on its own band only.
// Shapes that cannot be cut (a padded width that is not a multiple of 4: a segment would not be a whole number of
// 4-pixel groups) run as ONE band = the whole frame, where both mappings are the identity.
// What bands are for is MEMORY (the scratch of large frames), not speed.  Round 3 measured the alternatives on 1080p x 64
// (profiles/r03_transform_sweep.json): bands small enough to keep the slot in the 256 MiB Infinity Cache, with the two
// passes of neighbouring bands fused into one launch (temporal role || tile role), are 3-30 % SLOWER than the uncut
// chunk -- both passes turn out to be bound by vector-instruction issue (the VALU-floor probe: 0.51 / 0.43 ms with
// every global access removed, against 0.69 / 0.64 ms), so running them side by side buys nothing, and the many small
// launches cost their tails.  The default therefore cuts only when the uncut intermediate would exceed 1 GiB.
// ------------------------------------------------------------------------------------------------

// one band of a chunk as the tile role sees it
struct BandTiles {
    int nx, by0, nby;            // tiles per tile row; first tile row of the band, tile rows in it
    int ix0, ix1, iy0, iy1;      // the interior rectangle of the FRAME in tile coordinates (halo inside the frame)
};

// workgroup -> tile.  XCD-aware order (xcd_logical_block): XCD k takes the k-th contiguous eighth of the sequence
// (frame-major, then tile row, then tile), so a tile's neighbours hit in the same L2.
__device__ __forceinline__ bool band_tile_of_block(const BandTiles& bt, unsigned frames, int& bx, int& by, int& t, bool& edge) {
    const unsigned per_frame = (unsigned)(bt.nx * bt.nby);
    const unsigned l = xcd_logical_block(per_frame * frames);
    if (l == 0xFFFFFFFFu) return false;
    t = (int)(l / per_frame);
    const int i = (int)(l % per_frame);
    bx = i % bt.nx; by = bt.by0 + i / bt.nx;
    edge = !(bx >= bt.ix0 && bx < bt.ix1 && by >= bt.iy0 && by < bt.iy1);
    return true;
}

// ------------------------------------------------------------------------------------------------
// K1: forward spatial transform of one frame tile (tile role of the forward launch)
// ------------------------------------------------------------------------------------------------
constexpr int F_TW = 128, F_TH = 40, F_SEG = 16, F_NSEG = 8, F_THREADS = 384;
constexpr int F_LP = F_TW / 2 + 1;  // LDS row pitch in dwords (packed i16 pairs); odd pitch keeps stage-A stores at <= 2-way conflicts

struct FwdXy {
    const uint8_t* rgb;
    int16_t* mid;                // the band's slot: i16 [ch][t][band row][pw], band rows = [low rows | high rows]
    ChunkDims d;
    Coeffs cf;
    int aligned;
    BandTiles bt;
    int y0, rows;                // the band: first padded row (even) and number of padded rows (even)
};

// PROBE (VALU-floor measurement, alice_codec_test_transform_ms): the same instruction stream with every global load
// replaced by a register expression and every global store by an XOR into a checksum that is (never) stored at the end.
template <int NS, bool EDGE, int PROBE = 0>
__device__ __forceinline__ void fwd_xy_tile(const FwdXy& a, int bx, int by, int t, int* lds) {
    constexpr int H = NS;
    constexpr int ER = F_TH + 2 * H;
    constexpr int SE = F_SEG + 8;
    constexpr int K0 = 4 - H;
    constexpr int NL = F_SEG + 2 * H;
    const ChunkDims& d = a.d;
    const Coeffs& cf = a.cf;
    const int tid = threadIdx.x;
    const int gx0 = bx * F_TW, gy0 = by * F_TH;
    const int pw = d.pw, ph = d.ph;
    const int st = min(t, (int)d.f - 1);

    if (tid < ER * F_NSEG) {
        const int r = tid >> 3, s = tid & 7;
        const int gy = gy0 - H + r;
        {
            // border tiles read through the symmetric extension (rows and columns), then the pad row/column
            // of an odd size (src/pipeline.rs:77-114: edge replicate) clamps to the last real one
            const int sy = EDGE ? min(reflect_idx(gy, ph), (int)d.h - 1) : gy;
            const int gxs = gx0 - 4 + s * F_SEG;
            const uint8_t* row = a.rgb + ((size_t)st * d.h + sy) * d.w * 3;
            int y[SE], co[SE], cg[SE];
            if ((PROBE & 1) || (a.aligned && (!EDGE || (gxs >= 0 && gxs + SE <= (int)d.w)))) {
                const uint32_t* p4 = (const uint32_t*)(row + (ptrdiff_t)gxs * 3);
                uint32_t wd[18];
#pragma unroll
                for (int i = 0; i < 18; ++i) wd[i] = (PROBE & 1) ? ((uint32_t)tid * 0x9E3779B1u + (uint32_t)(i + t) * 0x85EBCA6Bu) : p4[i];
#pragma unroll
                for (int k = 0; k < SE; ++k) {
                    const int b0 = 3 * k;
                    const int rr = (wd[b0 >> 2] >> (8 * (b0 & 3))) & 0xFF;
                    const int gg = (wd[(b0 + 1) >> 2] >> (8 * ((b0 + 1) & 3))) & 0xFF;
                    const int bb = (wd[(b0 + 2) >> 2] >> (8 * ((b0 + 2) & 3))) & 0xFF;
                    const int c_o = rr - bb;
                    const int tt = bb + (c_o >> 1);
                    const int c_g = gg - tt;
                    y[k] = tt + (c_g >> 1); co[k] = c_o; cg[k] = c_g;
                }
            } else {
#pragma unroll
                for (int k = 0; k < SE; ++k) {
                    const int gx = gxs + k;
                    const int sx = EDGE ? min(reflect_idx(gx, pw), (int)d.w - 1) : gx;
                    const uint8_t* p = row + (size_t)sx * 3;
                    const int rr = p[0], gg = p[1], bb = p[2];
                    const int c_o = rr - bb;
                    const int tt = bb + (c_o >> 1);
                    const int c_g = gg - tt;
                    y[k] = tt + (c_g >> 1); co[k] = c_o; cg[k] = c_g;
                }
            }
            int ly[NL], lco[NL], lcg[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) { ly[k] = y[K0 + k]; lco[k] = co[K0 + k]; lcg[k] = cg[K0 + k]; }
            lift_regs<NL, NS, false, false>(ly, cf);
            lift_regs<NL, NS, false, false>(lco, cf);
            lift_regs<NL, NS, false, false>(lcg, cf);
            // interior 16 samples -> LDS row r as packed i16 pairs (values stay below 2^13 after the row pass):
            // dword q of a row holds deinterleaved positions (2q, 2q+1); evens of segment s start at
            // position s*8, odds at 64 + s*8
            int* L0 = lds + (0 * ER + r) * F_LP;
            int* L1 = lds + (1 * ER + r) * F_LP;
            int* L2 = lds + (2 * ER + r) * F_LP;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                L0[s * 4 + j] = (ly[H + 4 * j] & 0xFFFF) | (ly[H + 4 * j + 2] << 16);
                L0[32 + s * 4 + j] = (ly[H + 4 * j + 1] & 0xFFFF) | (ly[H + 4 * j + 3] << 16);
                L1[s * 4 + j] = (lco[H + 4 * j] & 0xFFFF) | (lco[H + 4 * j + 2] << 16);
                L1[32 + s * 4 + j] = (lco[H + 4 * j + 1] & 0xFFFF) | (lco[H + 4 * j + 3] << 16);
                L2[s * 4 + j] = (lcg[H + 4 * j] & 0xFFFF) | (lcg[H + 4 * j + 2] << 16);
                L2[32 + s * 4 + j] = (lcg[H + 4 * j + 1] & 0xFFFF) | (lcg[H + 4 * j + 3] << 16);
            }
        }
    }
    __syncthreads();
    {
        const int xq = tid & 127, ch = tid >> 7;
        const int par = xq >> 6, j = xq & 63;
        const int gxp = gx0 / 2 + j;
        const int hw = pw / 2, hr = a.rows / 2;
        if (!EDGE || gxp < hw) {
            int v[ER];
            const int* L = lds + (ch * ER) * F_LP + (xq >> 1);
            const int sh = (xq & 1) * 16;
#pragma unroll
            for (int r = 0; r < ER; ++r) v[r] = (int)(short)(L[r * F_LP] >> sh);
            lift_regs<ER, NS, false, false>(v, cf);
            // band slot: frame plane = rows x pw, low rows of the band first, then its high rows.  Channel and column
            // parity are the same for all 64 lanes of a wave (128 threads per channel, 64 per parity), so everything but the
            // lane's own column is WAVE-UNIFORM: the row pointers live in scalar registers and advance by scalar adds, the
            // lane adds a constant 32-bit offset (global_store saddr form) -- instead of one 64-bit vector add per stored
            // row (43 v_lshl_add_u64 per thread in the round-2 kernel)
            const int ch_u = __builtin_amdgcn_readfirstlane(ch), par_u = __builtin_amdgcn_readfirstlane(par);
            const int ly0 = (gy0 - a.y0) >> 1;   // gy0 and y0 are even
            char* const row0 = (char*)(a.mid + ((size_t)ch_u * d.pf + t) * ((size_t)a.rows * pw) + (size_t)par_u * hw + gx0 / 2 +
                                       (size_t)ly0 * pw);
            const uint32_t lane_off = (uint32_t)j * 2u;
            const size_t pitch = (size_t)pw * 2, half_off = (size_t)hr * pw * 2;
            int acc = 0;
#pragma unroll
            for (int k = 0; k < F_TH; ++k) {
                if (!EDGE || gy0 + k < ph) {
                    char* const rowp = row0 + (size_t)(k >> 1) * pitch + ((k & 1) ? half_off : 0);   // uniform
                    if (PROBE & 2) acc ^= v[H + k] + k;
                    else *(int16_t*)(rowp + lane_off) = (int16_t)v[H + k];
                }
            }
            if ((PROBE & 2) && acc == 0x5EEDF00D) a.mid[0] = (int16_t)acc;   // (keeps the checksum alive; a valid address whatever the tile)
        }
    }
}


// ------------------------------------------------------------------------------------------------
// K2: forward temporal lifting (stream over frame pairs) + quantise + symbols + histogram
//     (temporal role of the forward launch)
// ------------------------------------------------------------------------------------------------
struct I4 { int v[4]; };

template <bool EXACT>
__device__ __forceinline__ I4 lift4(const I4& base, const I4& a, const I4& b, int c) {
    I4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = wadd(base.v[i], lift_delta<EXACT>(a.v[i], b.v[i], c));
    return r;
}

// Histogram bins in LDS: 16 replicas of the 256 bins, word index = symbol * 16 + (lane & 15).  The dominant zero symbol
// (45-90 % of the data) spreads over 16 addresses instead of serialising on one; 32 replicas (one per bank) make a single
// ds_add conflict-free but cost 32 KB, and with the 4 KB value table next to them the fifth workgroup per CU: measured
// 245 us against 241 us for the kernel (profiles/r03_fwd_t_value_table_ab.txt).
#ifndef ALICE_HIST_REPLICAS
#define ALICE_HIST_REPLICAS 16
#endif
constexpr int kHistReplicas = ALICE_HIST_REPLICAS;
constexpr int kHistWords = kHistReplicas * 256;

__device__ __forceinline__ uint32_t sat_sub_u32(uint32_t a, uint32_t b) { return __builtin_elementwise_sub_sat(a, b); }

// Quantizer::quantize (src/quant.rs:89-97, dead zone = step) followed by to_symbols (:555-560), branch-free:
//   q = (|v| - step/2) / step for |v| >= step, else 0.  For |v| < step the quotient of the saturating
//   difference is already 0, so the dead-zone test disappears; symbol = 2q - (v > 0), and the one case that
//   would go negative (q = 0, v > 0; also v = 0 in the form used below) is exactly the case to_symbols maps to 0
//   (second saturating subtract).
template <bool STEP1, bool HIST>
__device__ __forceinline__ uint32_t quant_sym4(const I4& x, uint32_t hdz, uint32_t magic, uint32_t* lh, uint32_t lane_rep) {
    uint32_t s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int val = x.v[i];
        const int neg = -val;
        const uint32_t mag = (uint32_t)max(val, neg);
        const uint32_t adj = sat_sub_u32(mag, hdz);
        const uint32_t q = STEP1 ? adj : __umulhi(adj, magic);                     // exact: adj * step < 2^32
        // 2q - (v > 0) with floor 0, as 2q + (v < 0) - 1 saturating: v < 0 -> 2q, v > 0 -> 2q - 1, v = 0 -> 0
        const uint32_t t = sat_sub_u32((q << 1) + ((uint32_t)val >> 31), 1u);
        s[i] = t & 0xFFu;                                                          // `as u8`
        if (HIST) atomicAdd(&lh[s[i] * kHistReplicas + lane_rep], 1u);
    }
    return s[0] | (s[1] << 8) | (s[2] << 16) | (s[3] << 24);
}

// The same map for one value (the table below is filled with it, so table and arithmetic cannot disagree)
template <bool STEP1>
__device__ __forceinline__ uint32_t quant_sym1(int val, uint32_t hdz, uint32_t magic) {
    const uint32_t mag = (uint32_t)max(val, -val);
    const uint32_t adj = sat_sub_u32(mag, hdz);
    const uint32_t q = STEP1 ? adj : __umulhi(adj, magic);
    return sat_sub_u32((q << 1) + ((uint32_t)val >> 31), 1u) & 0xFFu;
}

// Value -> symbol table in LDS for coefficients in [-r, r), r <= kQLutR: one add and one ds_read_u8 per sample instead of
// the eight VALU operations above (one of them a quarter-rate multiply).  A wave in which any of a tick's eight values
// per lane lies outside the table takes the arithmetic path for that tick (wave-uniform branch): same symbols either way.
// 8-bit RGB cannot leave the table: the largest coefficient the three transforms give it is 2040 (Co of a 0/255
// checkerboard through CDF 5/3, whose high-pass row has absolute sum 2 per axis; 1.67 for the CDF 9/7 coefficients), so in
// the product the arithmetic path is a guard, not a path; the suite shrinks r (alice_codec_test_set_value_table_radius)
// to run it and the boundary between the two.
constexpr int kQLutR = 2048;

__device__ __forceinline__ I4 unpack4_i16(const uint2 w) {
    I4 r;
    r.v[0] = (int)(short)(w.x & 0xFFFF); r.v[1] = (int)w.x >> 16;
    r.v[2] = (int)(short)(w.y & 0xFFFF); r.v[3] = (int)w.y >> 16;
    return r;
}

// Temporal lifting as a stream over frame pairs k = 0 .. half-1 (E0[k], O0[k] = frames 2k, 2k+1 of one pixel):
//   P1: O1[k] = O0[k] + d(E0[k], E0[k+1])      (E0[half] := E0[half-1], src/wavelet.rs:186-190)
//   U1: E1[k] = E0[k] + d(O1[k-1], O1[k])      (O1[-1]   := O1[0],      :206-210)
//   P2: O2[k] = O1[k] + d(E1[k], E1[k+1])      U2: E2[k] = E1[k] + d(O2[k-1], O2[k])     (CDF 9/7 only)
// Tick k consumes pair k (and E0[k+1]) and emits E2[k-1] -> frame k-1, O2[k-1] -> frame half+k-1; the state
// between ticks is O1[k], E1[k], O2[k-1] plus the raw pairs in flight: five values per pixel.
// The tick is instantiated per position in a 4-tick block so that every register role is static: no window
// shuffling moves, no per-tick selects except the end-of-signal mirror.
template <int NS, bool STEP1, bool HIST, int PROBE = 0>
struct FwdT {
    const char* src;     // channel base inside the band slot (uniform)
    char* dst;           // channel base of the symbol volume (uniform)
    size_t plane_s;      // pixels per frame of the band slot
    size_t plane_d;      // pixels per frame of the symbol volume
    uint32_t off16, off8;  // this thread's byte offset inside an i16 slot frame / a u8 symbol frame
    int half;
    Coeffs cf;
    uint32_t hdz, magic, lane_rep;
    uint32_t* lh;
    const uint8_t* qlut; // value + qr -> symbol, for values in [-qr, qr)
    uint32_t qr;
    I4 o1p, e1p, o2pp;
    uint32_t acc;        // PROBE only

    __device__ __forceinline__ void load_pair(int pair, uint2& e, uint2& o) const {
        if (PROBE & 1) {     // small 16-bit pairs made of the pair index and the pixel offset: no memory
            const uint32_t m = (off16 + (uint32_t)pair * 0x00130017u) & 0x03FF03FFu;
            e = make_uint2(m, m ^ 0x00550033u); o = make_uint2(m + 0x00010002u, m ^ 0x000F00F0u);
            return;
        }
        const char* fe = src + (size_t)(2 * pair) * plane_s * 2;
        e = *(const uint2*)(fe + off16);
        o = *(const uint2*)(fe + plane_s * 2 + off16);
    }
    __device__ __forceinline__ void emit(int frame, const I4& lo, const I4& hi) {
        uint32_t a, b;
        uint32_t ix[8], any = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ix[i] = (uint32_t)lo.v[i] + qr; ix[4 + i] = (uint32_t)hi.v[i] + qr; }
#pragma unroll
        for (int i = 0; i < 8; ++i) any |= ix[i];
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(any >= 2u * qr) == 0ull, 1)) {
            uint32_t sy[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) sy[i] = qlut[ix[i]];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (HIST) atomicAdd(&lh[sy[i] * kHistReplicas + lane_rep], 1u);
            a = sy[0] | (sy[1] << 8) | (sy[2] << 16) | (sy[3] << 24);
            b = sy[4] | (sy[5] << 8) | (sy[6] << 16) | (sy[7] << 24);
        } else {
            a = quant_sym4<STEP1, HIST>(lo, hdz, magic, lh, lane_rep);
            b = quant_sym4<STEP1, HIST>(hi, hdz, magic, lh, lane_rep);
        }
        if (PROBE & 2) { acc ^= a + (uint32_t)frame; acc ^= b; return; }
        *(uint32_t*)(dst + (size_t)frame * plane_d + off8) = a;
        *(uint32_t*)(dst + (size_t)(half + frame) * plane_d + off8) = b;
    }
    // FIRST: k == 0, SECOND: k == 1 (the left mirrors); en = E0[k+1] or, on the last pair, E0[k] itself
    template <bool FIRST, bool SECOND>
    __device__ __forceinline__ void tick(int k, const I4& ce, const I4& co, const I4& en) {
        const I4 o1 = lift4<false>(co, ce, en, cf.c[0]);
        const I4 e1 = lift4<false>(ce, FIRST ? o1 : o1p, o1, cf.c[1]);
        if (NS == 2) {
            emit(k, e1, o1);
        } else if (!FIRST) {
            const I4 o2 = lift4<false>(o1p, e1p, e1, cf.c[2]);
            const I4 e2 = lift4<false>(e1p, SECOND ? o2 : o2pp, o2, cf.c[3]);
            emit(k - 1, e2, o2);
            o2pp = o2;
        }
        o1p = o1; e1p = e1;
    }
    // after the last pair: E1[half] := E1[half-1]
    __device__ __forceinline__ void flush() {
        if (NS == 2) return;
        const I4 o2 = lift4<false>(o1p, e1p, e1p, cf.c[2]);
        const I4 e2 = lift4<false>(e1p, half == 1 ? o2 : o2pp, o2, cf.c[3]);
        emit(half - 1, e2, o2);
    }
    // the whole stream of one 4-pixel group
    __device__ __forceinline__ void run() {
        o1p = I4{}; e1p = I4{}; o2pp = I4{};
        acc = 0u;
        uint2 re[4], ro[4];   // raw pairs in flight; slot = pair & 3, always a compile-time index below
#pragma unroll
        for (int i = 0; i < 4; ++i) { re[i] = make_uint2(0u, 0u); ro[i] = make_uint2(0u, 0u); }
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < half) load_pair(i, re[i], ro[i]);
        I4 ce = unpack4_i16(re[0]), co = unpack4_i16(ro[0]);
        // block 0: the two left mirrors are static; block ends may fall anywhere (short signals)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < half) {
                if (u + 3 < half) load_pair(u + 3, re[(u + 3) & 3], ro[(u + 3) & 3]);
                const I4 ne = unpack4_i16(re[(u + 1) & 3]);
                const bool last = u == half - 1;
                I4 en;
#pragma unroll
                for (int i = 0; i < 4; ++i) en.v[i] = last ? ce.v[i] : ne.v[i];
                if (u == 0) tick<true, false>(u, ce, co, en);
                else if (u == 1) tick<false, true>(u, ce, co, en);
                else tick<false, false>(u, ce, co, en);
                ce = ne; co = unpack4_i16(ro[(u + 1) & 3]);
            }
        }
        // steady state: whole blocks strictly before the last pair
        int kb = 4;
        for (; kb + 4 <= half - 1; kb += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u;
                if (k + 3 < half) load_pair(k + 3, re[(u + 3) & 3], ro[(u + 3) & 3]);
                const I4 ne = unpack4_i16(re[(u + 1) & 3]);
                tick<false, false>(k, ce, co, ne);
                ce = ne; co = unpack4_i16(ro[(u + 1) & 3]);
            }
        }
        // tail: up to 4 + 3 remaining pairs, the last one mirrors
        for (; kb < half; kb += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u;
                if (k < half) {
                    if (k + 3 < half) load_pair(k + 3, re[(u + 3) & 3], ro[(u + 3) & 3]);
                    const I4 ne = unpack4_i16(re[(u + 1) & 3]);
                    const bool last = k == half - 1;
                    I4 en;
#pragma unroll
                    for (int i = 0; i < 4; ++i) en.v[i] = last ? ce.v[i] : ne.v[i];
                    tick<false, false>(k, ce, co, en);
                    ce = ne; co = unpack4_i16(ro[(u + 1) & 3]);
                }
            }
        }
        flush();
        if ((PROBE & 2) && acc == 0x5EEDF00Du) *(uint32_t*)(dst + off8) = acc;
    }
};

// The temporal pass's view of a band.  A unit is one workgroup's worth of 4-pixel groups of one channel; the grid has one
// workgroup per unit (units are channel-major).
struct BandT {
    uint32_t pf;                 // padded frames
    uint32_t band_px;            // pixels of the band per frame = rows * pw (a multiple of 4)
    uint32_t seg_px;             // pixels of its low-row half; the rest are the high rows
    uint32_t lo_base, hi_base;   // where those halves start inside a frame of the chunk's (deinterleaved) plane
    uint64_t plane;              // pw * ph
    uint32_t units_per_ch;
};
// pixel index of a slot-linear pixel inside the chunk's plane (identity when the band is the whole frame)
__device__ __forceinline__ uint32_t band_plane_index(const BandT& b, uint32_t idx) {
    return idx < b.seg_px ? b.lo_base + idx : b.hi_base + (idx - b.seg_px);
}

struct FwdTm {
    const int16_t* mid;
    uint8_t* sym;
    uint32_t* hist;
    Coeffs cf;
    uint32_t hdz, magic;
    uint32_t qr;                 // radius of the value -> symbol table (1 .. kQLutR)
    BandT b;
};

template <int NS, bool STEP1, int PROBE = 0>
__global__ __launch_bounds__(256) void fwd_t_kernel(FwdTm a) {
    __shared__ uint32_t lh[kHistWords];
    __shared__ uint8_t qlut[2 * kQLutR];
    const int tid = threadIdx.x;
    for (int i = tid; i < kHistWords; i += 256) lh[i] = 0u;
    for (int i = tid; i < 2 * kQLutR / 4; i += 256) {
        uint32_t w = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) w |= quant_sym1<STEP1>(4 * i + j - (int)a.qr, a.hdz, a.magic) << (8 * j);
        ((uint32_t*)qlut)[i] = w;
    }
    __syncthreads();
    const int ch = (int)(blockIdx.x / a.b.units_per_ch);
    const uint32_t blk = blockIdx.x % a.b.units_per_ch;
    const uint32_t idx = (blk * 256u + (uint32_t)tid) * 4u;
    if (idx < a.b.band_px) {
        FwdT<NS, STEP1, true, PROBE> f;
        f.src = (const char*)(a.mid + (size_t)ch * a.b.pf * a.b.band_px);
        f.dst = (char*)(a.sym + (size_t)ch * a.b.pf * a.b.plane);
        f.plane_s = a.b.band_px; f.plane_d = a.b.plane;
        f.off16 = idx * 2u; f.off8 = band_plane_index(a.b, idx);
        f.half = (int)a.b.pf / 2; f.cf = a.cf; f.hdz = a.hdz; f.magic = a.magic; f.lane_rep = (uint32_t)tid & (kHistReplicas - 1); f.lh = lh; f.qlut = qlut; f.qr = a.qr;
        f.run();
    }
    __syncthreads();
    // fold the replicas of bin `tid` (rotated start: the 256 threads read 32 different banks)
    uint32_t cnt = 0u;
#pragma unroll 8
    for (int r = 0; r < kHistReplicas; ++r) cnt += lh[tid * kHistReplicas + ((r + tid) & (kHistReplicas - 1))];
    if (cnt) atomicAdd(&a.hist[ch * 256 + tid], cnt);
}

// (probe twins: border tiles need the clamping index map while the loads are real and the bounds tests while the stores
// are real; only with both replaced can the interior instance serve every tile)
template <int NS, int PROBE = 0>
__global__ __launch_bounds__(F_THREADS) void fwd_xy_kernel(FwdXy xa) {
    extern __shared__ int lds[];
    int bx, by, t;
    bool edge;
    if (!band_tile_of_block(xa.bt, (unsigned)xa.d.pf, bx, by, t, edge)) return;
    if (edge && PROBE != 3) fwd_xy_tile<NS, true, PROBE>(xa, bx, by, t, lds);
    else fwd_xy_tile<NS, false, PROBE>(xa, bx, by, t, lds);
}

// ------------------------------------------------------------------------------------------------
// K3: from_symbols + dequantize + inverse temporal lifting (stream over pairs); frames t < f only
//     (temporal role of the inverse launch)
// ------------------------------------------------------------------------------------------------
template <typename MidT>
__device__ __forceinline__ void store4(MidT* p, const I4& x);
template <>
__device__ __forceinline__ void store4<int32_t>(int32_t* p, const I4& x) {
    *(int4*)p = make_int4(x.v[0], x.v[1], x.v[2], x.v[3]);
}
template <>
__device__ __forceinline__ void store4<int16_t>(int16_t* p, const I4& x) {
    uint2 w;
    w.x = ((uint32_t)x.v[0] & 0xFFFFu) | ((uint32_t)x.v[1] << 16);
    w.y = ((uint32_t)x.v[2] & 0xFFFFu) | ((uint32_t)x.v[3] << 16);
    *(uint2*)p = w;
}

// Inverse stream (steps reversed with negated coefficients, src/wavelet.rs:167-174):
//   U2': E1[k] = E2[k] + d(O2[k-1], O2[k])     P2': O1[k] = O2[k] + d(E1[k], E1[k+1])
//   U1': E0[k] = E1[k] + d(O1[k-1], O1[k])     P1': O0[k] = O1[k] + d(E0[k], E0[k+1])
// Tick k consumes the symbols of frames k (low band) and half+k (high band), emits frame 2(k-1) and
// frame 2(k-2)+1.  from_symbols + dequantize (src/quant.rs:104-110,581-587) is a 256-entry table in LDS:
// one byte extract and one LDS read per sample instead of seven VALU operations.
template <int NS, bool EXACT, typename MidT, int PROBE = 0>
struct InvT {
    const char* src;      // channel base of the symbol volume
    MidT* dst;            // channel base inside the band slot + this thread's slot pixel
    size_t plane_s;       // pixels per frame of the symbol volume
    size_t plane_d;       // pixels per frame of the band slot
    uint32_t off8;
    int half, nf;
    int c0, c1, c2, c3;   // already negated
    const int* lut;
    I4 o2p, e1p, o1pp, e0pp;
    int acc;              // PROBE only

    __device__ __forceinline__ void load_pair(int pair, uint32_t& lo, uint32_t& hi) const {
        if (PROBE & 1) { lo = (off8 + (uint32_t)pair * 0x01030507u) & 0x0F070F07u; hi = lo ^ 0x01010101u; return; }
        lo = *(const uint32_t*)(src + (size_t)pair * plane_s + off8);
        hi = *(const uint32_t*)(src + (size_t)(half + pair) * plane_s + off8);
    }
    __device__ __forceinline__ I4 dq(uint32_t packed) const {
        I4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r.v[i] = lut[(packed >> (8 * i)) & 0xFFu];
        return r;
    }
    __device__ __forceinline__ void put(int frame, const I4& x) {
        if (PROBE & 2) { acc ^= (x.v[0] + frame) ^ x.v[1] ^ x.v[2] ^ x.v[3]; return; }
        if (frame < nf) store4<MidT>(dst + (size_t)frame * plane_d, x);
    }
    template <bool FIRST, bool SECOND>
    __device__ __forceinline__ void tick(int k, uint32_t lo, uint32_t hi) {
        const I4 hv = dq(hi), lv = dq(lo);
        if (NS == 4) {
            const I4 e1 = lift4<EXACT>(lv, FIRST ? hv : o2p, hv, c3);
            if (!FIRST) {
                const I4 o1 = lift4<EXACT>(o2p, e1p, e1, c2);
                const I4 e0 = lift4<EXACT>(e1p, SECOND ? o1 : o1pp, o1, c1);
                put(2 * (k - 1), e0);
                if (!SECOND) put(2 * (k - 2) + 1, lift4<EXACT>(o1pp, e0pp, e0, c0));
                o1pp = o1; e0pp = e0;
            }
            o2p = hv; e1p = e1;
        } else {
            const I4 e0 = lift4<EXACT>(lv, FIRST ? hv : o2p, hv, c1);
            put(2 * k, e0);
            if (!FIRST) put(2 * (k - 1) + 1, lift4<EXACT>(o2p, e1p, e0, c0));
            o2p = hv; e1p = e0;
        }
    }
    __device__ __forceinline__ void flush() {
        if (NS == 4) {
            const I4 o1 = lift4<EXACT>(o2p, e1p, e1p, c2);
            const I4 e0 = lift4<EXACT>(e1p, half == 1 ? o1 : o1pp, o1, c1);
            put(2 * (half - 1), e0);
            if (half >= 2) put(2 * (half - 2) + 1, lift4<EXACT>(o1pp, e0pp, e0, c0));
            put(2 * (half - 1) + 1, lift4<EXACT>(o1, e0, e0, c0));
        } else {
            put(2 * (half - 1) + 1, lift4<EXACT>(o2p, e1p, e1p, c0));
        }
    }
    __device__ __forceinline__ void run() {
        o2p = I4{}; e1p = I4{}; o1pp = I4{}; e0pp = I4{};
        acc = 0;
        uint32_t rl[4] = {0u, 0u, 0u, 0u}, rh[4] = {0u, 0u, 0u, 0u};   // symbols in flight; slot = pair & 3 (static)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < half) load_pair(i, rl[i], rh[i]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < half) {
                if (u + 3 < half) load_pair(u + 3, rl[(u + 3) & 3], rh[(u + 3) & 3]);
                if (u == 0) tick<true, false>(u, rl[u], rh[u]);
                else if (u == 1) tick<false, true>(u, rl[u], rh[u]);
                else tick<false, false>(u, rl[u], rh[u]);
            }
        }
        int kb = 4;
        for (; kb + 4 <= half; kb += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u;
                if (k + 3 < half) load_pair(k + 3, rl[(u + 3) & 3], rh[(u + 3) & 3]);
                tick<false, false>(k, rl[u], rh[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = kb + u;
            if (k < half) {
                if (k + 3 < half) load_pair(k + 3, rl[(u + 3) & 3], rh[(u + 3) & 3]);
                tick<false, false>(k, rl[u], rh[u]);
            }
        }
        flush();
        if ((PROBE & 2) && acc == 0x5EEDF00D) store4<MidT>(dst, o2p);
    }
};

struct InvTm {
    const uint8_t* sym;
    void* mid;                   // the band's slot: MidT [ch][t][slot row][pw], slot rows = [low rows L0.. | the same high rows]
    Coeffs cf;
    int step[3];
    uint32_t nf;                 // real frames (t < f are written)
    BandT b;                     // band_px = slot pixels per frame; lo_base / hi_base = L0 * pw / (hh + L0) * pw
};

template <int NS, bool EXACT, typename MidT, int PROBE = 0>
__global__ __launch_bounds__(256) void inv_t_kernel(InvTm a) {
    __shared__ int lut[256];
    const int tid = threadIdx.x;
    const int ch = (int)(blockIdx.x / a.b.units_per_ch);
    const uint32_t blk = blockIdx.x % a.b.units_per_ch;
    {
        const int s = tid;
        const int q = (s == 0) ? 0 : ((s & 1) ? (s + 1) / 2 : -(s / 2));      // src/quant.rs:581-587
        lut[s] = (int)((unsigned)q * (unsigned)a.step[ch]);                   // src/quant.rs:104-110 (wrapping)
    }
    __syncthreads();
    const uint32_t idx = (blk * 256u + (uint32_t)tid) * 4u;
    if (idx >= a.b.band_px) return;
    InvT<NS, EXACT, MidT, PROBE> f;
    f.src = (const char*)(a.sym + (size_t)ch * a.b.pf * a.b.plane);
    f.dst = (MidT*)a.mid + (size_t)ch * a.b.pf * a.b.band_px + idx;
    f.plane_s = a.b.plane; f.plane_d = a.b.band_px;
    f.off8 = band_plane_index(a.b, idx);
    f.half = (int)a.b.pf / 2; f.nf = (int)a.nf;
    f.c0 = -a.cf.c[0]; f.c1 = -a.cf.c[1]; f.c2 = -a.cf.c[2]; f.c3 = -a.cf.c[3];
    f.lut = lut;
    f.run();
}

// ------------------------------------------------------------------------------------------------
// K4: inverse spatial transform of one frame tile + colour (tile role of the inverse launch)
// ------------------------------------------------------------------------------------------------
constexpr int I_TW = 96, I_TH = 32, I_SEG = 8, I_NSEG = 12, I_THREADS = 384;
constexpr int I_HO = 56;    // LDS column of the first odd (high-band) sample of a row: the two halves sit at 0 and 56
constexpr int I_LW = 112;   // LDS row pitch in dwords.  Every 8-sample run stage B reads is two aligned ds_read_b128.  A b128
                            // read is served in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...;
                            // MI355X_MICROARCH.md, LDS), bank = dword mod 64; with 12 lanes per tile row, pitch 112 / halves at 0 and
                            // 56 is the layout below 128 dwords for which every group touches 64 different banks (pitch 108 / 52,
                            // chosen in round 2 for contiguous groups, was a 2-way conflict on every read:
                            // scripts/lds_bank_model.py); column-major stage-A stores stay at most 2-way, which costs nothing

struct InvXy {
    const void* mid;             // the band's slot (see InvTm)
    uint8_t* rgb;
    ChunkDims d;
    Coeffs cf;
    int aligned;
    BandTiles bt;
    int L0, rows_l;              // slot rows: low-band rows [L0, L0 + rows_l), then the same rows of the high band
};

// stage A of both inverse tile kernels: one thread per (extended column, channel) loads its ER rows from the band slot
// and runs the inverse column lifting in registers.  Rows and columns outside the frame are read through the symmetric
// extension (reflect_idx), in the interleaved index space, so neither stage needs boundary logic in its arithmetic;
// rows that only feed never-stored outputs of overhanging tiles are clamped into the slot.
// (Scalar row pointers plus a 32-bit lane offset here, as the forward tile's stores have them, measured no gain -- 435 vs
// 431 us -- and giving every channel 128 threads so that the channel is wave-uniform too was 6 % slower: DESIGN.md 4.0.)
template <int NS, int ER, bool EDGE, bool EXACT, typename MidT, int PROBE = 0>
__device__ __forceinline__ void inv_load_lift_column(const InvXy& a, int ch, int par, int px, int t, int gy_s, int (&v)[ER]) {
    const ChunkDims& d = a.d;
    const int pw = d.pw, ph = d.ph, hw = pw / 2;
    const MidT* src = (const MidT*)a.mid + ((size_t)ch * d.pf + t) * ((size_t)2 * a.rows_l * pw) + (size_t)par * hw + (px >> 1);
    if (PROBE & 1) {
#pragma unroll
        for (int k = 0; k < ER; ++k) v[k] = (int)(short)((px + gy_s) * 37 + k * 101 + t);
    } else if (EDGE) {
#pragma unroll
        for (int k = 0; k < ER; ++k) {
            const int gy = reflect_idx(gy_s + k, ph);
            const int yl = (gy & 1) * a.rows_l + min(max((gy >> 1) - a.L0, 0), a.rows_l - 1);
            v[k] = (int)src[(size_t)yl * pw];
        }
    } else {
        // gy_s is even: even rows of the tile are low-band rows gy_s / 2 + m, odd rows the same high-band rows
        const MidT* lo = src + (size_t)((gy_s >> 1) - a.L0) * pw;
        const MidT* hi = lo + (size_t)a.rows_l * pw;
#pragma unroll
        for (int m = 0; m < ER / 2; ++m) { v[2 * m] = (int)lo[(size_t)m * pw]; v[2 * m + 1] = (int)hi[(size_t)m * pw]; }
    }
    lift_regs<ER, NS, EXACT, true>(v, a.cf);
}

// `as i16` (src/pipeline.rs:608), wrapping i16 colour inverse (src/color.rs:266-273), 8 pixels -> 24 bytes, dword stores
template <int PROBE = 0>
__device__ __forceinline__ void store_rgb8(const InvXy& a, bool edge, const int* y, const int* co, const int* cg, int t, int gy, int gxs) {
    const ChunkDims& d = a.d;
    uint8_t out[24];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const short yv = (short)y[kk], c_o = (short)co[kk], c_g = (short)cg[kk];
        const short tt = (short)(yv - (short)(c_g >> 1));
        const short g = (short)(c_g + tt);
        const short b = (short)(tt - (short)(c_o >> 1));
        const short rr = (short)(c_o + b);
        out[3 * kk] = (uint8_t)min(max((int)rr, 0), 255);
        out[3 * kk + 1] = (uint8_t)min(max((int)g, 0), 255);
        out[3 * kk + 2] = (uint8_t)min(max((int)b, 0), 255);
    }
    uint8_t* p = a.rgb + (((size_t)t * d.h + gy) * d.w + gxs) * 3;
    if (PROBE & 2) {
        uint32_t acc = 0u;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            acc ^= ((uint32_t)out[4 * i] | ((uint32_t)out[4 * i + 1] << 8) | ((uint32_t)out[4 * i + 2] << 16) | ((uint32_t)out[4 * i + 3] << 24)) + (uint32_t)i;
        if (acc == 0x5EEDF00Du) *a.rgb = (uint8_t)acc;   // (keeps the checksum alive; a valid address whatever the tile)
    } else if (a.aligned && (!edge || gxs + 8 <= (int)d.w)) {
        uint32_t* p4 = (uint32_t*)p;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            p4[i] = (uint32_t)out[4 * i] | ((uint32_t)out[4 * i + 1] << 8) | ((uint32_t)out[4 * i + 2] << 16) |
                    ((uint32_t)out[4 * i + 3] << 24);
    } else {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            if (gxs + kk < (int)d.w) { p[3 * kk] = out[3 * kk]; p[3 * kk + 1] = out[3 * kk + 1]; p[3 * kk + 2] = out[3 * kk + 2]; }
    }
}

// Recompute variant with the packed tile: the host proved that every value after the inverse column lifting fits i16
// (InverseBounds::lds16): two tile rows share a dword, which halves the tile (26 KB: the 32-wave limit, not LDS, then
// bounds the workgroups per CU).  A stage-B thread lifts its 8 pixels plus NS halo samples on each side.
template <int NS, bool EDGE, bool EXACT, typename MidT>
__device__ __forceinline__ void inv_xy_tile_packed(const InvXy& a, int bx, int by, int t, int* lds) {
    constexpr int H = NS;
    constexpr int ER = I_TH + 2 * H;        // rows incl. halo
    constexpr int EC = I_TW + 2 * H;        // columns incl. halo (104 / 100)
    constexpr int ECh = EC / 2;
    constexpr int LR = ER / 2;              // LDS rows per channel (two tile rows per dword)
    constexpr int NL = I_SEG + 2 * H;       // samples lifted per segment
    const ChunkDims& d = a.d;
    const int tid = threadIdx.x;
    const int gx0 = bx * I_TW, gy0 = by * I_TH;
    const int pw = d.pw;
    const int gpx0 = (gx0 - H) / 2;  // first column pair of the extended tile (may be negative)

    if (tid < 3 * EC) {
        const int ch = tid / EC, xq = tid % EC;
        const int par = xq / ECh, j = xq % ECh;
        const int px = EDGE ? reflect_idx(2 * (gpx0 + j) + par, pw) : 2 * (gpx0 + j) + par;   // parity is preserved
        int v[ER];
        inv_load_lift_column<NS, ER, EDGE, EXACT, MidT>(a, ch, par, px, t, gy0 - H, v);
        int* L = lds + (ch * LR) * I_LW + par * I_HO + j;
#pragma unroll
        for (int m = 0; m < ER / 2; ++m) L[m * I_LW] = (v[2 * m] & 0xFFFF) | (v[2 * m + 1] << 16);
    }
    __syncthreads();

    // stage B: one thread per (interior row, 8-pixel segment): inverse row lifting + colour
    if (tid < I_TH * I_NSEG) {
        const int r = tid / I_NSEG, s = tid % I_NSEG;  // 384 = 32 * 12
        const int gy = gy0 + r;
        const int gxs = gx0 + s * I_SEG;               // first interior pixel of the segment
        if (!EDGE || (gy < (int)d.h && gxs < (int)d.w)) {
            int y[NL], co[NL], cg[NL];
            // sample k of the segment is extended-tile column s*8 + k: even k in the low half at dword s*4 + k/2,
            // odd k in the high half at I_HO + s*4 + k/2
            const int k = r + H;
            const int lrow = k >> 1;
            const bool upper = k & 1;
            auto fetch = [&](int ch, int (&dst)[NL]) {
                const int* Lc = lds + (ch * LR + lrow) * I_LW + s * 4;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int4 va = *(const int4*)(Lc + half * I_HO);
                    const int4 vb = *(const int4*)(Lc + half * I_HO + 4);
                    const int w8[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
                    for (int q = 0; q < NL / 2; ++q) {
                        const int w = w8[q];
                        dst[2 * q + half] = upper ? (w >> 16) : (int)(short)w;
                    }
                }
            };
            fetch(0, y); fetch(1, co); fetch(2, cg);
            lift_regs<NL, NS, EXACT, true>(y, a.cf);
            lift_regs<NL, NS, EXACT, true>(co, a.cf);
            lift_regs<NL, NS, EXACT, true>(cg, a.cf);
            store_rgb8(a, EDGE, y + H, co + H, cg + H, t, gy, gxs);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4x: the same inverse tile with LANE-EXCHANGE row lifting.  In the recompute variant a stage-B thread lifts its 8 pixels
// plus NS halo samples on each side (16 samples to keep 8) because it cannot see its neighbours' intermediate values.  Here
// the 16 lanes of a DPP row hold one tile row -- a left halo segment, twelve 8-pixel segments, a right halo segment --
// and every lifting step fetches the one neighbour sample it needs with a row_shr / row_shl DPP move: a lane lifts exactly
// its own 8 samples.  An output depends on inputs at most NS samples away and the halo lanes hold NS true samples next to
// the tile (their far halves are never-read garbage), so lanes 1..12 end up with the same integers as the global pass.
// Used whenever the packed-i16 tile is not provably safe (e.g. q = 80).
// ------------------------------------------------------------------------------------------------
constexpr int X_TH = 32, X_THREADS = X_TH * 16;   // stage B: X_TH rows x 16 lanes
constexpr int X_HO = 56, X_LW = 112;    // tile row: even samples at columns 2 .. 53 (+2 never-read columns each side), odd at 56 + ...

// one inverse/forward lifting pass over the 8 samples of a lane (v[0] even), neighbours by DPP within the 16-lane row
template <int NS, bool EXACT, bool INVERSE>
__device__ __forceinline__ void lift_seg8_dpp(int (&v)[8], const Coeffs& cf) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k_step = INVERSE ? (NS - 1 - s) : s;
        const int c = INVERSE ? -cf.c[k_step] : cf.c[k_step];
        if ((k_step & 1) == 0) {  // predict: odd += d(even_left, even_right); v[7] needs the right neighbour's v[0]
            const int right0 = __builtin_amdgcn_update_dpp(0, v[0], 0x101 /* row_shl:1 */, 0xf, 0xf, false);
#pragma unroll
            for (int k = 1; k < 8; k += 2) v[k] = wadd(v[k], lift_delta<EXACT>(v[k - 1], k + 1 < 8 ? v[k + 1] : right0, c));
        } else {                  // update: even += d(odd_left, odd_right); v[0] needs the left neighbour's v[7]
            const int left7 = __builtin_amdgcn_update_dpp(0, v[7], 0x111 /* row_shr:1 */, 0xf, 0xf, false);
#pragma unroll
            for (int k = 0; k < 8; k += 2) v[k] = wadd(v[k], lift_delta<EXACT>(k >= 1 ? v[k - 1] : left7, v[k + 1], c));
        }
    }
}

template <int NS, bool EDGE, bool EXACT, typename MidT, int PROBE = 0>
__device__ __forceinline__ void inv_xy_tile_dpp(const InvXy& a, int bx, int by, int t, int* lds) {
    constexpr int H = NS;
    constexpr int ER = X_TH + 2 * H;
    constexpr int EC = I_TW + 2 * H;
    constexpr int ECh = EC / 2;
    const ChunkDims& d = a.d;
    const int tid = threadIdx.x;
    const int gx0 = bx * I_TW, gy0 = by * X_TH;
    const int pw = d.pw;
    const int gpx0 = (gx0 - H) / 2;

    // stage A: extended column e = 2 j + par goes to tile column par * X_HO + (8 - H) / 2 + j (the H halo samples of a
    // side end right at the first / last real segment: the row starts 8 samples left of the tile)
    if (tid < 3 * EC) {
        const int ch = tid / EC, xq = tid % EC;
        const int par = xq / ECh, j = xq % ECh;
        const int px = EDGE ? reflect_idx(2 * (gpx0 + j) + par, pw) : 2 * (gpx0 + j) + par;
        int v[ER];
        inv_load_lift_column<NS, ER, EDGE, EXACT, MidT, PROBE>(a, ch, par, px, t, gy0 - H, v);
        int* L = lds + (ch * ER) * X_LW + par * X_HO + (8 - H) / 2 + j;
#pragma unroll
        for (int k = 0; k < ER; ++k) L[k * X_LW] = v[k];
    }
    __syncthreads();

    // stage B: lane s of a 16-lane row holds segment s - 1 (8 samples = 4 even + 4 odd: one ds_read_b128 per half).
    // Which tile rows share a wave matters: a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31} (and the same + 32), so lanes of two DPP rows meet in one group.  With the two rows 4 tile rows apart their
    // bases differ by 4 * 112 = 7 * 64 dwords, the same bank phase, and each group covers 64 different banks (consecutive
    // rows, 112 = 48 mod 64 apart, were a 2-way conflict on every read: 0.40 of all LDS cycles, scripts/lds_bank_model.py).
    {
        const int s = tid & 15, q = (tid >> 4) & 3, pr = 2 * (tid >> 6) + (q >> 1);
        const int r = 8 * (pr >> 2) + (pr & 3) + 4 * (q & 1);
        const int gy = gy0 + r;
        const int gxs = gx0 + (s - 1) * 8;
        int y[8], co[8], cg[8];
        auto fetch = [&](int ch, int (&dst)[8]) {
            const int* Lc = lds + (ch * ER + r + H) * X_LW + s * 4;
            const int4 va = s < 14 ? *(const int4*)Lc : make_int4(0, 0, 0, 0);
            const int4 vb = s < 14 ? *(const int4*)(Lc + X_HO) : make_int4(0, 0, 0, 0);
            dst[0] = va.x; dst[2] = va.y; dst[4] = va.z; dst[6] = va.w;
            dst[1] = vb.x; dst[3] = vb.y; dst[5] = vb.z; dst[7] = vb.w;
        };
        fetch(0, y); fetch(1, co); fetch(2, cg);
        lift_seg8_dpp<NS, EXACT, true>(y, a.cf);
        lift_seg8_dpp<NS, EXACT, true>(co, a.cf);
        lift_seg8_dpp<NS, EXACT, true>(cg, a.cf);
        if (s >= 1 && s <= 12 && (!EDGE || (gy < (int)d.h && gxs < (int)d.w))) store_rgb8<PROBE>(a, EDGE, y, co, cg, t, gy, gxs);
    }
}

// PACKED = the packed-i16 recompute tile (384 threads); otherwise the lane-exchange tile (512 threads).
template <bool PACKED> struct InvTileShape { static constexpr int kThreads = PACKED ? I_THREADS : X_THREADS;
                                             static constexpr int kLdsInts = PACKED ? 3 * ((I_TH + 8) / 2) * I_LW : 3 * (X_TH + 8) * X_LW; };

template <int NS, bool EXACT, typename MidT, bool PACKED, int PROBE = 0>
__global__ __launch_bounds__(InvTileShape<PACKED>::kThreads) void inv_xy_kernel(InvXy xa) {
    __shared__ __attribute__((aligned(16))) int lds[InvTileShape<PACKED>::kLdsInts];
    int bx, by, t;
    bool edge;
    if (!band_tile_of_block(xa.bt, xa.d.f, bx, by, t, edge)) return;
    if (PACKED) {
        if (edge) inv_xy_tile_packed<NS, true, EXACT, MidT>(xa, bx, by, t, lds);
        else inv_xy_tile_packed<NS, false, EXACT, MidT>(xa, bx, by, t, lds);
    } else {
        if (edge && PROBE != 3) inv_xy_tile_dpp<NS, true, EXACT, MidT, PROBE>(xa, bx, by, t, lds);
        else inv_xy_tile_dpp<NS, false, EXACT, MidT, PROBE>(xa, bx, by, t, lds);
    }
}

// ------------------------------------------------------------------------------------------------
// Stage-level Wavelet2D / Wavelet3D on caller-shaped i32 data (src/wavelet.rs:292-340, 392-484): the same tile
// structure as the pipeline kernels with the reference's exact arithmetic (wrapping i32 sums, 64-bit products), so any
// i32 values are allowed.  Three planes per workgroup play the part the three colour channels play above.
// Shapes: even width and height of at least 6 (one reflection covers the halo), even depth; everything else (odd
// lengths with the reference's dropped tail sample, tiny sizes) stays on generic.hip.
// ------------------------------------------------------------------------------------------------
constexpr int S_LP = F_TW + 1;   // LDS row pitch in dwords (one i32 sample each)

template <int NS>
__global__ __launch_bounds__(F_THREADS) void stage_fwd_xy_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                                        int w, int h, unsigned planes, unsigned tiles_x,
                                                                        unsigned tiles_per_plane, Coeffs cf) {
    constexpr int H = NS;
    constexpr int ER = F_TH + 2 * H;
    constexpr int NL = F_SEG + 2 * H;
    extern __shared__ int lds[];   // [3][ER][S_LP]
    const int tid = threadIdx.x;
    const unsigned groups = (planes + 2u) / 3u;
    const unsigned l = xcd_logical_block(tiles_per_plane * groups);
    if (l == 0xFFFFFFFFu) return;
    const unsigned g = l / tiles_per_plane, ti = l % tiles_per_plane;
    const int gx0 = (int)(ti % tiles_x) * F_TW, gy0 = (int)(ti / tiles_x) * F_TH;
    const size_t plane = (size_t)w * h;
    // stage A: rows (src/wavelet.rs:401-405), a 16-sample segment + halo per thread, one plane after the other
    if (tid < ER * F_NSEG) {
        const int r = tid >> 3, s = tid & 7;
        const int gy = reflect_idx(gy0 - H + r, h);
        for (int pl = 0; pl < 3; ++pl) {
            const unsigned p = g * 3u + (unsigned)pl;
            if (p >= planes) break;
            const int32_t* row = in + (size_t)p * plane + (size_t)gy * w;
            int v[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) v[k] = row[reflect_idx(gx0 - H + s * F_SEG + k, w)];
            lift_regs<NL, NS, true, false>(v, cf);
            int* L = lds + (pl * ER + r) * S_LP;
#pragma unroll
            for (int j = 0; j < F_SEG / 2; ++j) { L[s * 8 + j] = v[H + 2 * j]; L[64 + s * 8 + j] = v[H + 2 * j + 1]; }
        }
    }
    __syncthreads();
    // stage B: columns (:408-417), one deinterleaved column of one plane per thread
    {
        const int xq = tid & 127, pl = tid >> 7;
        const unsigned p = g * 3u + (unsigned)pl;
        const int par = xq >> 6, j = xq & 63;
        const int gxp = gx0 / 2 + j;
        const int hw = w / 2, hh = h / 2;
        if (p < planes && gxp < hw) {
            int v[ER];
            const int* L = lds + (pl * ER) * S_LP + xq;
#pragma unroll
            for (int r = 0; r < ER; ++r) v[r] = L[r * S_LP];
            lift_regs<ER, NS, true, false>(v, cf);
            int32_t* o = out + (size_t)p * plane + (size_t)par * hw + gxp;
#pragma unroll
            for (int k = 0; k < F_TH; ++k) {
                const int gy = gy0 + k;
                if (gy < h) o[(size_t)((gy & 1) * hh + (gy >> 1)) * w] = v[H + k];
            }
        }
    }
}

template <int NS>
__global__ __launch_bounds__(I_THREADS) void stage_inv_xy_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                                        int w, int h, unsigned planes, unsigned tiles_x,
                                                                        unsigned tiles_per_plane, Coeffs cf) {
    constexpr int H = NS;
    constexpr int ER = I_TH + 2 * H;
    constexpr int EC = I_TW + 2 * H;
    constexpr int ECh = EC / 2;
    constexpr int NL = I_SEG + 2 * H;
    __shared__ __attribute__((aligned(16))) int lds[3 * ER * I_LW];
    const int tid = threadIdx.x;
    const unsigned groups = (planes + 2u) / 3u;
    const unsigned l = xcd_logical_block(tiles_per_plane * groups);
    if (l == 0xFFFFFFFFu) return;
    const unsigned g = l / tiles_per_plane, ti = l % tiles_per_plane;
    const int gx0 = (int)(ti % tiles_x) * I_TW, gy0 = (int)(ti / tiles_x) * I_TH;
    const size_t plane = (size_t)w * h;
    const int hw = w / 2, hh = h / 2;
    const int gpx0 = (gx0 - H) / 2;
    // stage A: inverse columns (src/wavelet.rs:319-329 order: columns first), one extended column of one plane per thread
    if (tid < 3 * EC) {
        const int pl = tid / EC, xq = tid % EC;
        const unsigned p = g * 3u + (unsigned)pl;
        if (p < planes) {
            const int par = xq / ECh, j = xq % ECh;
            const int px = reflect_idx(2 * (gpx0 + j) + par, w);
            const int32_t* src = in + (size_t)p * plane + (size_t)par * hw + (px >> 1);
            int v[ER];
#pragma unroll
            for (int k = 0; k < ER; ++k) {
                const int gy = reflect_idx(gy0 - H + k, h);
                v[k] = src[(size_t)((gy & 1) * hh + (gy >> 1)) * w];
            }
            lift_regs<ER, NS, true, true>(v, cf);
            int* L = lds + (pl * ER) * I_LW + par * I_HO + j;
#pragma unroll
            for (int k = 0; k < ER; ++k) L[k * I_LW] = v[k];
        }
    }
    __syncthreads();
    // stage B: inverse rows, 8 interleaved output samples per thread and plane
    {
        const int r = tid / I_NSEG, s = tid % I_NSEG;
        const int gy = gy0 + r, gxs = gx0 + s * I_SEG;
        if (gy < h && gxs < w) {
            for (int pl = 0; pl < 3; ++pl) {
                const unsigned p = g * 3u + (unsigned)pl;
                if (p >= planes) break;
                int v[NL];
                const int* Lc = lds + (pl * ER + r + H) * I_LW + s * 4;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int4 a = *(const int4*)(Lc + half * I_HO);
                    const int4 b = *(const int4*)(Lc + half * I_HO + 4);
                    const int w8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int q = 0; q < NL / 2; ++q) v[2 * q + half] = w8[q];
                }
                lift_regs<NL, NS, true, true>(v, cf);
                int32_t* o = out + (size_t)p * plane + (size_t)gy * w + gxs;
#pragma unroll
                for (int k = 0; k < I_SEG; ++k)
                    if (gxs + k < w) o[k] = v[H + k];
            }
        }
    }
}

// Temporal pass of Wavelet3D (src/wavelet.rs:421-437 forward, :447-463 inverse) as a stream over frame pairs, one
// pixel per thread, exact arithmetic; `in` and `out` are different buffers (the transform deinterleaves along t).
template <int NS, bool INVERSE>
__global__ __launch_bounds__(256) void stage_t_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, size_t plane,
                                                      int depth, Coeffs cf) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= plane) return;
    const int half = depth / 2;
    const int32_t* s = in + idx;
    int32_t* o = out + idx;
    auto d = [](int a, int b, int c) { return lift_delta<true>(a, b, c); };
    if (!INVERSE) {
        int o1p = 0, e1p = 0, o2pp = 0;
        int ce = s[0], co = s[plane];
        for (int k = 0; k < half; ++k) {
            const int ne = (k + 1 < half) ? s[(size_t)(2 * k + 2) * plane] : ce;
            const int no = (k + 1 < half) ? s[(size_t)(2 * k + 3) * plane] : 0;
            const int o1 = wadd(co, d(ce, ne, cf.c[0]));
            const int e1 = wadd(ce, d(k == 0 ? o1 : o1p, o1, cf.c[1]));
            if (NS == 2) {
                o[(size_t)k * plane] = e1; o[(size_t)(half + k) * plane] = o1;
            } else if (k >= 1) {
                const int o2 = wadd(o1p, d(e1p, e1, cf.c[2]));
                const int e2 = wadd(e1p, d(k == 1 ? o2 : o2pp, o2, cf.c[3]));
                o[(size_t)(k - 1) * plane] = e2; o[(size_t)(half + k - 1) * plane] = o2;
                o2pp = o2;
            }
            o1p = o1; e1p = e1; ce = ne; co = no;
        }
        if (NS == 4) {
            const int o2 = wadd(o1p, d(e1p, e1p, cf.c[2]));
            const int e2 = wadd(e1p, d(half == 1 ? o2 : o2pp, o2, cf.c[3]));
            o[(size_t)(half - 1) * plane] = e2; o[(size_t)(2 * half - 1) * plane] = o2;
        }
    } else {
        const int c0 = -cf.c[0], c1 = -cf.c[1], c2 = -cf.c[2], c3 = -cf.c[3];
        int o2p = 0, e1p = 0, o1pp = 0, e0pp = 0;
        for (int k = 0; k < half; ++k) {
            const int lv = s[(size_t)k * plane], hv = s[(size_t)(half + k) * plane];
            if (NS == 4) {
                const int e1 = wadd(lv, d(k == 0 ? hv : o2p, hv, c3));
                if (k >= 1) {
                    const int o1 = wadd(o2p, d(e1p, e1, c2));
                    const int e0 = wadd(e1p, d(k == 1 ? o1 : o1pp, o1, c1));
                    o[(size_t)(2 * (k - 1)) * plane] = e0;
                    if (k >= 2) o[(size_t)(2 * (k - 2) + 1) * plane] = wadd(o1pp, d(e0pp, e0, c0));
                    o1pp = o1; e0pp = e0;
                }
                o2p = hv; e1p = e1;
            } else {
                const int e0 = wadd(lv, d(k == 0 ? hv : o2p, hv, c1));
                o[(size_t)(2 * k) * plane] = e0;
                if (k >= 1) o[(size_t)(2 * (k - 1) + 1) * plane] = wadd(o2p, d(e1p, e0, c0));
                o2p = hv; e1p = e0;
            }
        }
        if (NS == 4) {
            const int o1 = wadd(o2p, d(e1p, e1p, c2));
            const int e0 = wadd(e1p, d(half == 1 ? o1 : o1pp, o1, c1));
            o[(size_t)(2 * (half - 1)) * plane] = e0;
            if (half >= 2) o[(size_t)(2 * (half - 2) + 1) * plane] = wadd(o1pp, d(e0pp, e0, c0));
            o[(size_t)(2 * (half - 1) + 1) * plane] = wadd(o1, d(e0, e0, c0));
        } else {
            o[(size_t)(2 * (half - 1) + 1) * plane] = wadd(o2p, d(e1p, e1p, c0));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

static Coeffs to_coeffs(const LiftSteps& s) {
    Coeffs c{};
    for (int i = 0; i < 4; ++i) c.c[i] = i < s.n ? s.coeff[i] : 0;
    return c;
}

template <typename K>
static bool set_dyn_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

// Target size of a band slot in KiB (0 = never cut a chunk into bands).  Default 1 GiB: 1080p x 64 chunks stay whole,
// 4K x 64 chunks run in 4 bands, 8K x 64 chunks in 13 (see "Bands").  Settable by alice_codec_test_set_tuning (the suite
// runs small shapes through many bands) and, for developer runs, by ALICE_BAND_KB at first use.
static long& band_target_kb() {
    static long kb = [] { const char* v = getenv("ALICE_BAND_KB"); return (v && *v) ? atol(v) : 1024L * 1024L; }();
    return kb;
}
void set_transform_tuning(long band_kb) { if (band_kb >= 0) band_target_kb() = band_kb; }
static int g_value_table_radius = kQLutR;
void set_value_table_radius(int r) { g_value_table_radius = r < 1 ? 1 : (r > kQLutR ? kQLutR : r); }
// alice_codec_test_transform_ms(probe != 0): the CDF 9/7, step > 1, i16 lane-exchange instances run their probe twins
static thread_local int tl_valu_probe = 0;
void set_transform_probe(int mode) { tl_valu_probe = mode; }

bool transform_tiles_eligible(const ChunkDims& d) {
    if ((unsigned long long)d.pw * d.ph > (1ull << 30)) return false;   // 32-bit byte offsets inside one frame
    if (d.pw < 6 || d.ph < 6) return false;                             // reflect_idx: one reflection must cover the halo
    if ((unsigned long long)((d.pw + I_TW - 1) / I_TW) * ((d.ph + I_TH - 1) / I_TH) * d.pf > 0x7FFFFFF0ull) return false;
    return true;
}

// Band plan of a chunk for tiles of tile_h rows: bands of `tpb` tile rows (the last one takes what is left).  Cut only
// when the padded width is a multiple of 4 (see "Bands") and there are at least two bands' worth of rows.
BandPlan plan_bands(const ChunkDims& d, int tile_h, size_t bytes_per_sample, int halo_rows) {
    BandPlan p{};
    p.tile_h = tile_h;
    p.tiles_y = (int)((d.ph + tile_h - 1) / tile_h);
    const size_t row_bytes = (size_t)d.pw * d.pf * 3 * bytes_per_sample;     // one padded row of all frames and channels
    long tpb = p.tiles_y;
    const long target = band_target_kb();
    if (target > 0 && d.pw % 4 == 0) {
        tpb = (long)(((size_t)target << 10) / (row_bytes * (size_t)tile_h));
        if (tpb < 1) tpb = 1;
        if (tpb * 2 > p.tiles_y) tpb = p.tiles_y;                            // fewer than two bands: do not cut
    }
    p.n_bands = (int)((p.tiles_y + tpb - 1) / tpb);
    p.tpb = (p.tiles_y + p.n_bands - 1) / p.n_bands;                         // bands of equal height (no sliver at the end)
    p.n_bands = (p.tiles_y + p.tpb - 1) / p.tpb;
    // rows a slot holds: the band's own rows (+ the halo rows of the inverse band on either side), or the whole frame
    const size_t slot_rows = p.n_bands > 1 ? std::min<size_t>((size_t)p.tpb * tile_h, d.ph) + (size_t)2 * halo_rows : d.ph;
    p.slot_bytes = ((slot_rows * row_bytes + 255) / 256) * 256 + 256;
    return p;
}

size_t forward_scratch_bytes(const ChunkDims& d) { return plan_bands(d, F_TH, sizeof(int16_t), 0).slot_bytes; }
size_t inverse_scratch_bytes(const ChunkDims& d, bool mid16) {
    return plan_bands(d, I_TH, mid16 ? sizeof(int16_t) : sizeof(int32_t), 4).slot_bytes;
}

// whether the inverse launches of a chunk of this shape work band by band (either sample width)
bool inverse_cuts_chunk(const ChunkDims& d) {
    return plan_bands(d, I_TH, sizeof(int16_t), 4).n_bands > 1 || plan_bands(d, I_TH, sizeof(int32_t), 4).n_bands > 1;
}

static BandT make_band_t(const ChunkDims& d, uint32_t band_px, uint32_t seg_px, uint32_t lo_base, uint32_t hi_base) {
    BandT b{};
    b.pf = (uint32_t)d.pf; b.band_px = band_px; b.seg_px = seg_px; b.lo_base = lo_base; b.hi_base = hi_base;
    b.plane = d.pw * d.ph;
    b.units_per_ch = (band_px + 1023u) / 1024u;      // 256 threads x 4 pixels
    return b;
}

// ---- forward ----

template <int NS, int PROBE = 0>
static void fwd_xy_launch(const FwdXy& xa, hipStream_t st) {
    const size_t lds = (size_t)3 * (F_TH + 2 * NS) * F_LP * sizeof(int);
    static const bool ok = set_dyn_lds(fwd_xy_kernel<NS, PROBE>, lds);
    (void)ok;
    const unsigned long long tiles = (unsigned long long)xa.bt.nx * xa.bt.nby * xa.d.pf;
    hipLaunchKernelGGL((fwd_xy_kernel<NS, PROBE>), dim3(xcd_grid(tiles)), dim3(F_THREADS), lds, st, xa);
}
template <int NS, bool STEP1, int PROBE = 0>
static void fwd_t_launch(const FwdTm& ta, hipStream_t st) {
    hipLaunchKernelGGL((fwd_t_kernel<NS, STEP1, PROBE>), dim3(3u * ta.b.units_per_ch), dim3(256), 0, st, ta);
}

bool launch_forward_transform(const uint8_t* d_rgb, const ChunkDims& d, int wavelet, int32_t step,
                              void* d_scratch, uint8_t* d_sym, uint32_t* d_hist, hipStream_t st) {
    if (step < 1 || step > 64) return false;
    if (!transform_tiles_eligible(d)) return false;
    const LiftSteps ls = lift_steps(wavelet);
    const BandPlan bp = plan_bands(d, F_TH, sizeof(int16_t), 0);
    const Coeffs cf = to_coeffs(ls);
    const unsigned nx = (unsigned)((d.pw + F_TW - 1) / F_TW), ny = (unsigned)bp.tiles_y;
    // a tile is interior when its loaded range [gx0-4, gx0+128+4) x [gy0-H, gy0+40+H) lies inside w x h
    auto interior_x = [&](unsigned bx) { return bx >= 1 && (bx * F_TW + F_TW + 4) <= d.w; };
    auto interior_y = [&](unsigned by) { return by >= 1 && (by * F_TH + F_TH + 4) <= d.h; };
    unsigned ix1 = 1, iy1 = 1;
    while (ix1 < nx && interior_x(ix1)) ++ix1;
    while (iy1 < ny && interior_y(iy1)) ++iy1;
    const uint32_t magic = step == 1 ? 0u : (uint32_t)(((1ull << 32) + (uint32_t)step - 1u) / (uint32_t)step);
    const uint32_t hh = (uint32_t)(d.ph / 2), pw = (uint32_t)d.pw;
    const int probe = (ls.n == 4 && step != 1) ? tl_valu_probe : 0;
    for (int band = 0; band < bp.n_bands; ++band) {
        const int by0 = band * bp.tpb, nby = std::min(bp.tpb, bp.tiles_y - by0);
        const int y0 = by0 * F_TH, rows = (int)std::min<uint64_t>((uint64_t)(by0 + nby) * F_TH, d.ph) - y0;
        FwdXy xa{};
        xa.rgb = d_rgb; xa.mid = (int16_t*)d_scratch; xa.d = d; xa.cf = cf;
        xa.aligned = (d.w % 4 == 0) && ((((uintptr_t)d_rgb) & 3u) == 0u);
        xa.bt = BandTiles{(int)nx, by0, nby, 1, (int)ix1, 1, (int)iy1};
        xa.y0 = y0; xa.rows = rows;
        FwdTm ta{};
        ta.mid = (const int16_t*)d_scratch; ta.sym = d_sym; ta.hist = d_hist; ta.cf = cf; ta.hdz = (uint32_t)step / 2u; ta.magic = magic; ta.qr = (uint32_t)g_value_table_radius;
        ta.b = make_band_t(d, (uint32_t)rows * pw, (uint32_t)(rows / 2) * pw, (uint32_t)(y0 / 2) * pw, (hh + (uint32_t)(y0 / 2)) * pw);
        if (probe) {   // loads and stores replaced / loads only / stores only
            if (probe == 1) { fwd_xy_launch<4, 3>(xa, st); fwd_t_launch<4, false, 3>(ta, st); }
            else if (probe == 2) { fwd_xy_launch<4, 1>(xa, st); fwd_t_launch<4, false, 1>(ta, st); }
            else { fwd_xy_launch<4, 2>(xa, st); fwd_t_launch<4, false, 2>(ta, st); }
        } else if (ls.n == 4) {
            fwd_xy_launch<4>(xa, st);
            if (step == 1) fwd_t_launch<4, true>(ta, st); else fwd_t_launch<4, false>(ta, st);
        } else {
            fwd_xy_launch<2>(xa, st);
            if (step == 1) fwd_t_launch<2, true>(ta, st); else fwd_t_launch<2, false>(ta, st);
        }
    }
    return true;
}

// ---- inverse ----

template <int NS, bool EXACT, typename MidT, bool PACKED, int PROBE = 0>
static void inv_band_launch(const InvTm& ta, const InvXy& xa, hipStream_t st) {
    hipLaunchKernelGGL((inv_t_kernel<NS, EXACT, MidT, PROBE>), dim3(3u * ta.b.units_per_ch), dim3(256), 0, st, ta);
    const unsigned long long tiles = (unsigned long long)xa.bt.nx * xa.bt.nby * xa.d.f;
    hipLaunchKernelGGL((inv_xy_kernel<NS, EXACT, MidT, PACKED, PROBE>), dim3(xcd_grid(tiles)), dim3(InvTileShape<PACKED>::kThreads), 0, st, xa);
}

bool launch_inverse_transform(const uint8_t* d_sym, const ChunkDims& d, int wavelet, const int32_t step[3],
                              bool exact, bool mid16, bool lds16, void* d_scratch, uint8_t* d_rgb, hipStream_t st) {
    if (!transform_tiles_eligible(d)) return false;
    const LiftSteps ls = lift_steps(wavelet);
    // mid16: the host proved every value after the inverse temporal pass fits i16 (then exact is false too);
    // lds16: also after the inverse column pass (packed tile).
    // variant: 0 exact (i32), 1 fast i32, 2 fast i16 lane-exchange tile, 3 fast i16 packed tile
    const int variant = exact ? 0 : (mid16 ? (lds16 ? 3 : 2) : 1);
    const BandPlan bp = plan_bands(d, I_TH, variant >= 2 ? sizeof(int16_t) : sizeof(int32_t), 4);
    const Coeffs cf = to_coeffs(ls);
    const unsigned nx = (d.w + I_TW - 1) / I_TW, ny = (d.h + I_TH - 1) / I_TH;   // tiles over the REAL frame (every pixel written exists)
    // a tile is interior when the range it reads, [gx0 - 4, gx0 + 96 + 4) x [gy0 - 4, gy0 + 32 + 4), lies inside w x h
    // (then inside the padded frame too)
    auto interior_x = [&](unsigned bx) { return bx >= 1 && (bx * I_TW + I_TW + 4) <= d.w; };
    auto interior_y = [&](unsigned by) { return by >= 1 && (by * I_TH + I_TH + 4) <= d.h; };
    unsigned ix1 = 1, iy1 = 1;
    while (ix1 < nx && interior_x(ix1)) ++ix1;
    while (iy1 < ny && interior_y(iy1)) ++iy1;
    const int hh = (int)(d.ph / 2);
    const uint32_t pw = (uint32_t)d.pw;
    const int n_bands = bp.n_bands > 1 ? ((int)ny + bp.tpb - 1) / bp.tpb : 1;
    const int probe = (ls.n == 4 && variant == 2) ? tl_valu_probe : 0;
    for (int band = 0; band < n_bands; ++band) {
        const int by0 = n_bands > 1 ? band * bp.tpb : 0, nby = n_bands > 1 ? std::min(bp.tpb, (int)ny - by0) : (int)ny;
        // slot rows: the band's own rows plus two halo rows of each half on either side, inside the frame
        int L0 = 0, L1 = hh;
        if (n_bands > 1) {
            const int Y0 = by0 * I_TH, Y1 = (int)std::min<uint64_t>((uint64_t)(by0 + nby) * I_TH, d.ph);
            L0 = std::max(0, Y0 / 2 - 2); L1 = std::min(hh, Y1 / 2 + 2);
        }
        const int rows_l = L1 - L0;
        InvTm ta{};
        ta.sym = d_sym; ta.mid = d_scratch; ta.cf = cf; ta.step[0] = step[0]; ta.step[1] = step[1]; ta.step[2] = step[2]; ta.nf = d.f;
        ta.b = make_band_t(d, 2u * (uint32_t)rows_l * pw, (uint32_t)rows_l * pw, (uint32_t)L0 * pw, (uint32_t)(hh + L0) * pw);
        InvXy xa{};
        xa.mid = d_scratch; xa.rgb = d_rgb; xa.d = d; xa.cf = cf;
        xa.aligned = (d.w % 4 == 0) && ((((uintptr_t)d_rgb) & 3u) == 0u);
        xa.bt = BandTiles{(int)nx, by0, nby, 1, (int)ix1, 1, (int)iy1};
        xa.L0 = L0; xa.rows_l = rows_l;
        if (probe) {
            if (probe == 1) inv_band_launch<4, false, int16_t, false, 3>(ta, xa, st);
            else if (probe == 2) inv_band_launch<4, false, int16_t, false, 1>(ta, xa, st);
            else inv_band_launch<4, false, int16_t, false, 2>(ta, xa, st);
            continue;
        }
#define ALICE_INV(NS_) \
        switch (variant) { \
        case 0: inv_band_launch<NS_, true, int32_t, false>(ta, xa, st); break; \
        case 1: inv_band_launch<NS_, false, int32_t, false>(ta, xa, st); break; \
        case 2: inv_band_launch<NS_, false, int16_t, false>(ta, xa, st); break; \
        default: inv_band_launch<NS_, false, int16_t, true>(ta, xa, st); break; \
        }
        if (ls.n == 4) { ALICE_INV(4) } else { ALICE_INV(2) }
#undef ALICE_INV
    }
    return true;
}


bool stage_tiles_eligible(uint64_t w, uint64_t h, uint64_t depth, int ndim) {
    if (ndim < 2 || (w & 1) || (h & 1) || w < 6 || h < 6) return false;
    if (w * h > (1ull << 30) || depth > 0x7FFFFFFFull || depth == 0) return false;
    if (ndim >= 3 && depth > 1 && (depth & 1)) return false;      // an odd temporal length drops its tail sample: generic path
    const uint64_t groups = (depth + 2) / 3;
    if (((w + I_TW - 1) / I_TW) * ((h + I_TH - 1) / I_TH) * groups > 0x7FFFFFF0ull) return false;
    return true;
}

// data -> result in `data`; tmp: same size.  ndim 2: every one of the `depth` planes is transformed on its own.
void launch_stage_wavelet(int32_t* d_data, int32_t* d_tmp, uint64_t w, uint64_t h, uint64_t depth, int ndim, int wavelet,
                          bool inverse, hipStream_t st) {
    const LiftSteps ls = lift_steps(wavelet);
    const Coeffs cf = to_coeffs(ls);
    const size_t plane = (size_t)w * h;
    const unsigned planes = (unsigned)depth, groups = (planes + 2) / 3;
    const bool temporal = ndim >= 3 && depth >= 2;
    const size_t ldsf = (size_t)3 * (F_TH + 2 * ls.n) * S_LP * sizeof(int);
    auto fwd_xy = [&](const int32_t* a, int32_t* b) {
        const unsigned tx = (unsigned)((w + F_TW - 1) / F_TW), ty = (unsigned)((h + F_TH - 1) / F_TH);
        dim3 grid(xcd_grid((unsigned long long)tx * ty * groups));
        if (ls.n == 4) {
            static const bool ok = set_dyn_lds(stage_fwd_xy_kernel<4>, (size_t)3 * (F_TH + 8) * S_LP * sizeof(int)); (void)ok;
            hipLaunchKernelGGL((stage_fwd_xy_kernel<4>), grid, dim3(F_THREADS), ldsf, st, a, b, (int)w, (int)h, planes, tx, tx * ty, cf);
        } else {
            static const bool ok = set_dyn_lds(stage_fwd_xy_kernel<2>, (size_t)3 * (F_TH + 4) * S_LP * sizeof(int)); (void)ok;
            hipLaunchKernelGGL((stage_fwd_xy_kernel<2>), grid, dim3(F_THREADS), ldsf, st, a, b, (int)w, (int)h, planes, tx, tx * ty, cf);
        }
    };
    auto inv_xy = [&](const int32_t* a, int32_t* b) {
        const unsigned tx = (unsigned)((w + I_TW - 1) / I_TW), ty = (unsigned)((h + I_TH - 1) / I_TH);
        dim3 grid(xcd_grid((unsigned long long)tx * ty * groups));
        if (ls.n == 4) hipLaunchKernelGGL((stage_inv_xy_kernel<4>), grid, dim3(I_THREADS), 0, st, a, b, (int)w, (int)h, planes, tx, tx * ty, cf);
        else hipLaunchKernelGGL((stage_inv_xy_kernel<2>), grid, dim3(I_THREADS), 0, st, a, b, (int)w, (int)h, planes, tx, tx * ty, cf);
    };
    auto t_pass = [&](const int32_t* a, int32_t* b, bool inv) {
        dim3 grid((unsigned)((plane + 255) / 256));
        if (ls.n == 4) { if (inv) hipLaunchKernelGGL((stage_t_kernel<4, true>), grid, dim3(256), 0, st, a, b, plane, (int)depth, cf);
                         else hipLaunchKernelGGL((stage_t_kernel<4, false>), grid, dim3(256), 0, st, a, b, plane, (int)depth, cf); }
        else { if (inv) hipLaunchKernelGGL((stage_t_kernel<2, true>), grid, dim3(256), 0, st, a, b, plane, (int)depth, cf);
               else hipLaunchKernelGGL((stage_t_kernel<2, false>), grid, dim3(256), 0, st, a, b, plane, (int)depth, cf); }
    };
    const size_t bytes = plane * depth * sizeof(int32_t);
    if (!inverse) {
        fwd_xy(d_data, d_tmp);
        if (temporal) t_pass(d_tmp, d_data, false);
        else (void)hipMemcpyAsync(d_data, d_tmp, bytes, hipMemcpyDeviceToDevice, st);
    } else {
        if (temporal) { t_pass(d_data, d_tmp, true); inv_xy(d_tmp, d_data); }
        else { inv_xy(d_data, d_tmp); (void)hipMemcpyAsync(d_data, d_tmp, bytes, hipMemcpyDeviceToDevice, st); }
    }
}

}  // namespace alice
