// Pipeline-specialised transform kernels for gfx950: RGB <-> u8 symbols.
//
// Forward (FrameEncoder::encode, reference src/pipeline.rs:429-477):
//   fwd_xy_kernel : per frame and 128x40 tile.  Stage A: a thread owns a 16-pixel row segment
//                   (+ one halo sample per lifting step on each side): RGB -> YCoCg-R
//                   (src/color.rs:221-228), edge-replicated padding by index clamping
//                   (src/pipeline.rs:77-114), row lifting entirely in registers
//                   (src/wavelet.rs:401-405).  The tile goes through LDS once (the transpose).
//                   Stage B: a thread owns one column of the tile (+halo rows) in registers, column
//                   lifting (:408-417), i16 store to the [L|H]-deinterleaved plane.
//   fwd_t_kernel  : temporal lifting as a stream over frame pairs with a five-value window per
//                   pixel (src/wavelet.rs:421-437), 4 pixels per thread (8-byte loads, 4-byte stores),
//                   fused Quantizer::quantize (src/quant.rs:89-97), to_symbols (:555-560) and
//                   build_histogram (:594-600).  Any frame count.
// Inverse (FrameDecoder::decode, src/pipeline.rs:588-621) mirrors it:
//   inv_t_kernel  : from_symbols, dequantize, inverse temporal lifting (stream over pairs).
//   inv_xy_kernel : stage A inverse column lifting (registers) -> LDS -> stage B inverse row lifting,
//                   `as i16`, ycocg_r_to_rgb_bytes.
//
// Lifting facts used: inside one lifting step every write depends only on samples of the other
// parity, so a step is data-parallel; an output depends on inputs within +-n_steps samples;
// mirroring happens only at the true signal ends (src/wavelet.rs:186-190,206-210).  A tile or
// segment whose halo is recomputed therefore yields the same integers as the global pass.
//
// Intermediates: the forward path stores i16 (from u8 input every value in the transform stays
// below 2^13 in magnitude, SURVEY.md section 7 hard part 5); the inverse path stores i32 because a
// desynchronised decoder feeds it arbitrary symbols (i16 when the host proves the bound).
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
        // check), so the full-rate 24-bit multiply-add is exact
        return (__mul24(a + b, c) + 4096) >> 13;
    }
}
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }

struct Coeffs { int c[4]; };
struct TileMap { int nx, ny, ix0, ix1, iy0, iy1; };

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

// ------------------------------------------------------------------------------------------------
// In-register 1-D lifting of N consecutive samples (v[0] has even global index g0).
// EDGE: g0 + k may fall outside [0, n); mirrors follow the reference at the true ends, array ends
// clamp (their garbage stays inside the halo).  INVERSE applies the steps reversed with -coeff.
// ------------------------------------------------------------------------------------------------
template <int N, int NS, bool EDGE, bool EXACT, bool INVERSE>
__device__ __forceinline__ void lift_regs(int (&v)[N], const Coeffs& cf, int g0, int n) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k_step = INVERSE ? (NS - 1 - s) : s;
        const int c = INVERSE ? -cf.c[k_step] : cf.c[k_step];
        if ((k_step & 1) == 0) {  // predict: odd += d(even_left, even_right), src/wavelet.rs:184-196
#pragma unroll
            for (int k = 1; k < N; k += 2) {
                int right;
                if (k + 1 < N) right = (!EDGE || (g0 + k + 1 < n)) ? v[k + 1] : v[k - 1];
                else right = v[k - 1];
                v[k] = wadd(v[k], lift_delta<EXACT>(v[k - 1], right, c));
            }
        } else {                  // update: even += d(odd_left, odd_right), src/wavelet.rs:205-216
#pragma unroll
            for (int k = 0; k < N; k += 2) {
                const int rgt = (k + 1 < N) ? v[k + 1] : v[k - 1];
                int left;
                if (k >= 1) left = (!EDGE || (g0 + k > 0)) ? v[k - 1] : rgt;
                else left = rgt;
                v[k] = wadd(v[k], lift_delta<EXACT>(left, rgt, c));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: forward spatial transform of one frame tile
// ------------------------------------------------------------------------------------------------
constexpr int F_TW = 128, F_TH = 40, F_SEG = 16, F_NSEG = 8, F_THREADS = 384;
constexpr int F_LP = F_TW / 2 + 1;  // LDS row pitch in dwords (packed i16 pairs); odd pitch keeps stage-A stores at <= 2-way conflicts

template <int NS, bool EDGE>
__global__ __launch_bounds__(F_THREADS) void fwd_xy_kernel(const uint8_t* __restrict__ rgb, int16_t* __restrict__ mid,
                                                                  ChunkDims d, Coeffs cf, int aligned, TileMap tm) {
    // interior launch: a rectangle of tiles at (ix0, iy0); border launch: one linear grid over the
    // four border strips (top, bottom, left, right of the interior rectangle)
    constexpr int H = NS;
    constexpr int ER = F_TH + 2 * H;
    constexpr int SE = F_SEG + 8;
    constexpr int K0 = 4 - H;
    constexpr int NL = F_SEG + 2 * H;
    extern __shared__ int lds[];
    const int tid = threadIdx.x;
    int bx, by, t;
    {
        const int n_top = tm.nx * tm.iy0, n_bot = tm.nx * (tm.ny - tm.iy1), n_left = tm.ix0 * (tm.iy1 - tm.iy0);
        const int n_right = (tm.nx - tm.ix1) * (tm.iy1 - tm.iy0);
        const int iw = tm.ix1 - tm.ix0, ih = tm.iy1 - tm.iy0;
        const unsigned per_frame = EDGE ? (unsigned)(n_top + n_bot + n_left + n_right) : (unsigned)(iw * ih);
        const unsigned l = xcd_logical_block(per_frame * d.pf);
        if (l == 0xFFFFFFFFu) return;
        t = (int)(l / per_frame);
        int i = (int)(l % per_frame);
        if (!EDGE) { bx = tm.ix0 + i % iw; by = tm.iy0 + i / iw; }
        else if (i < n_top) { bx = i % tm.nx; by = i / tm.nx; }
        else if ((i -= n_top) < n_bot) { bx = i % tm.nx; by = tm.iy1 + i / tm.nx; }
        else if ((i -= n_bot) < n_left) { bx = i % tm.ix0; by = tm.iy0 + i / tm.ix0; }
        else { i -= n_left; const int wr = tm.nx - tm.ix1; bx = tm.ix1 + i % wr; by = tm.iy0 + i / wr; }
    }
    const int gx0 = bx * F_TW, gy0 = by * F_TH;
    const int pw = d.pw, ph = d.ph;
    const int st = min(t, (int)d.f - 1);

    if (tid < ER * F_NSEG) {
        const int r = tid >> 3, s = tid & 7;
        const int gy = gy0 - H + r;
        if (!EDGE || (gy >= 0 && gy < ph)) {
            const int sy = EDGE ? min(gy, (int)d.h - 1) : gy;
            const int gxs = gx0 - 4 + s * F_SEG;
            const uint8_t* row = rgb + ((size_t)st * d.h + sy) * d.w * 3;
            int y[SE], co[SE], cg[SE];
            if (aligned && (!EDGE || (gxs >= 0 && gxs + SE <= (int)d.w))) {
                const uint32_t* p4 = (const uint32_t*)(row + (ptrdiff_t)gxs * 3);
                uint32_t wd[18];
#pragma unroll
                for (int i = 0; i < 18; ++i) wd[i] = p4[i];
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
                    const int sx = min(max(gx, 0), (int)d.w - 1);
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
            const int g0 = gxs + K0;
            lift_regs<NL, NS, EDGE, false, false>(ly, cf, g0, pw);
            lift_regs<NL, NS, EDGE, false, false>(lco, cf, g0, pw);
            lift_regs<NL, NS, EDGE, false, false>(lcg, cf, g0, pw);
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
        const int hw = pw / 2, hh = ph / 2;
        if (!EDGE || gxp < hw) {
            int v[ER];
            const int* L = lds + (ch * ER) * F_LP + (xq >> 1);
            const int sh = (xq & 1) * 16;
#pragma unroll
            for (int r = 0; r < ER; ++r) v[r] = (int)(short)(L[r * F_LP] >> sh);
            lift_regs<ER, NS, EDGE, false, false>(v, cf, gy0 - H, ph);
            int16_t* out = mid + ((size_t)ch * d.pf + t) * ph * pw + (size_t)par * hw + gxp;
#pragma unroll
            for (int k = 0; k < F_TH; ++k) {
                const int gy = gy0 + k;
                if (!EDGE || gy < ph) {
                    const int yy = (gy & 1) * hh + (gy >> 1);
                    out[(size_t)yy * pw] = (int16_t)v[H + k];
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// K2: forward temporal lifting (stream over frame pairs) + quantise + symbols + histogram
// ------------------------------------------------------------------------------------------------
struct I4 { int v[4]; };

__device__ __forceinline__ I4 load4_i16(const int16_t* p) {
    const uint2 w = *(const uint2*)p;
    I4 r;
    r.v[0] = (int)(short)(w.x & 0xFFFF); r.v[1] = (int)(short)(w.x >> 16);
    r.v[2] = (int)(short)(w.y & 0xFFFF); r.v[3] = (int)(short)(w.y >> 16);
    return r;
}
template <bool EXACT>
__device__ __forceinline__ I4 lift4(const I4& base, const I4& a, const I4& b, int c) {
    I4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = wadd(base.v[i], lift_delta<EXACT>(a.v[i], b.v[i], c));
    return r;
}

// Histogram bins in LDS: 8 replicas of the 256 bins (replica = lane & 7) cut same-address atomics
// eightfold; a zero symbol (45-90 % of the data) goes to a slot private to its thread instead, so the
// add is unconditional (no exec-mask branch) and conflict-free.  Zeros are recovered as total - nonzero.
constexpr int kHistReplicas = 8;
constexpr int kHistWords = kHistReplicas * 256 + 256;

// Quantizer::quantize (src/quant.rs:89-97, dead zone = step) followed by to_symbols (:555-560), branch-free:
//   q = (|v| - step/2) / step for |v| >= step, else 0.  For |v| < step the quotient of the saturating
//   difference is already 0, so the dead-zone test disappears; symbol = 2q - (v > 0), and the one case that
//   would go negative (q = 0, v > 0) is exactly the case to_symbols maps to 0.
template <bool STEP1, bool HIST>
__device__ __forceinline__ uint32_t quant_sym4(const I4& x, int hdz, uint32_t magic, uint32_t* lh, int rep_base, int dummy) {
    uint32_t packed = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int val = x.v[i];
        const int neg = -val;
        const uint32_t mag = (uint32_t)max(val, neg);
        const uint32_t adj = mag > (uint32_t)hdz ? mag - (uint32_t)hdz : 0u;      // saturating
        const uint32_t q = STEP1 ? adj : __umulhi(adj, magic);                     // exact: adj * step < 2^32
        const int t = (int)(q << 1) + (neg >> 31);                                 // 2q - (v > 0)
        const uint32_t s = (uint32_t)max(t, 0) & 0xFFu;                            // `as u8`
        if (HIST) atomicAdd(&lh[s ? (rep_base + (int)s) : dummy], 1u);
        packed |= s << (8 * i);
    }
    return packed;
}

template <int NS, bool STEP1, bool HIST>
__global__ __launch_bounds__(256) void fwd_t_kernel(const int16_t* __restrict__ mid, uint8_t* __restrict__ sym,
                                                    uint32_t* __restrict__ hist, ChunkDims d, Coeffs cf, int step,
                                                    uint32_t magic) {
    __shared__ uint32_t lh[kHistWords];
    const int tid = threadIdx.x;
    for (int i = tid; i < kHistWords; i += 256) lh[i] = 0u;
    __syncthreads();
    const size_t plane = (size_t)d.pw * d.ph;
    const size_t idx = ((size_t)blockIdx.x * 256 + tid) * 4;
    const int ch = blockIdx.y;
    const int pf = d.pf, half = pf / 2;
    const int rep_base = (tid & (kHistReplicas - 1)) * 256, dummy = kHistReplicas * 256 + tid;
    const int hdz = step / 2;
    uint32_t emitted = 0u;
    if (idx < plane) {
        emitted = 4u * (uint32_t)pf;
        const int16_t* src = mid + (size_t)ch * pf * plane + idx;
        uint32_t* dst = (uint32_t*)(sym + (size_t)ch * pf * plane + idx);
        const size_t plane4 = plane / 4;
        // window: raw pair j-1, O1[j-2], E1[j-2], O2[j-3] (9/7); see the derivation in DESIGN.md
        I4 e0p{}, o0p{}, o1pp{}, e1pp{}, o2ppp{};
        I4 e0c = load4_i16(src), o0c = load4_i16(src + plane);
        I4 e0n{}, o0n{};
        if (1 < half) { e0n = load4_i16(src + (size_t)2 * plane); o0n = load4_i16(src + (size_t)3 * plane); }
        const int last = (NS == 4) ? half + 1 : half;
#pragma unroll 2
        for (int j = 0; j <= last; ++j) {
            I4 e0nn{}, o0nn{};
            if (j + 2 < half) {  // prefetch pair j+2
                e0nn = load4_i16(src + (size_t)(2 * j + 4) * plane);
                o0nn = load4_i16(src + (size_t)(2 * j + 5) * plane);
            }
            if (j >= 1) {
                I4 o1p, e1p;
                bool have1 = j <= half;
                if (have1) {
                    // P1 for pair j-1: right neighbour is E0[j] or (at the end) E0[j-1] itself
                    o1p = lift4<false>(o0p, e0p, (j < half) ? e0c : e0p, cf.c[0]);
                    // U1 for pair j-1: left neighbour is O1[j-2] or (at the start) O1[0]
                    e1p = lift4<false>(e0p, (j >= 2) ? o1pp : o1p, o1p, cf.c[1]);
                }
                if (NS == 2) {
                    const uint32_t lo = quant_sym4<STEP1, HIST>(e1p, hdz, magic, lh, rep_base, dummy);
                    const uint32_t hi = quant_sym4<STEP1, HIST>(o1p, hdz, magic, lh, rep_base, dummy);
                    dst[(size_t)(j - 1) * plane4] = lo;
                    dst[(size_t)(half + j - 1) * plane4] = hi;
                } else {
                    if (j >= 2) {
                        // P2 for pair j-2: right neighbour E1[j-1] or mirror E1[j-2]
                        const I4 o2pp = lift4<false>(o1pp, e1pp, have1 ? e1p : e1pp, cf.c[2]);
                        // U2 for pair j-2: left neighbour O2[j-3] or mirror O2[0]
                        const I4 e2pp = lift4<false>(e1pp, (j >= 3) ? o2ppp : o2pp, o2pp, cf.c[3]);
                        const uint32_t lo = quant_sym4<STEP1, HIST>(e2pp, hdz, magic, lh, rep_base, dummy);
                        const uint32_t hi = quant_sym4<STEP1, HIST>(o2pp, hdz, magic, lh, rep_base, dummy);
                        dst[(size_t)(j - 2) * plane4] = lo;
                        dst[(size_t)(half + j - 2) * plane4] = hi;
                        o2ppp = o2pp;
                    }
                }
                if (have1) { o1pp = o1p; e1pp = e1p; }
            }
            e0p = e0c; o0p = o0c;
            e0c = e0n; o0c = o0n;
            e0n = e0nn; o0n = o0nn;
        }
    }
    if (!HIST) return;
    __syncthreads();
    // fold the replicas; bin 0 = symbols emitted by this block - nonzero symbols
    __shared__ uint32_t tot_sh, nz_sh;
    if (tid == 0) { tot_sh = 0u; nz_sh = 0u; }
    __syncthreads();
    uint32_t cnt = 0u;
#pragma unroll
    for (int r = 0; r < kHistReplicas; ++r) cnt += lh[r * 256 + tid];
    if (emitted) atomicAdd(&tot_sh, emitted);
    if (tid != 0 && cnt) { atomicAdd(&nz_sh, cnt); atomicAdd(&hist[ch * 256 + tid], cnt); }
    __syncthreads();
    if (tid == 0 && tot_sh > nz_sh) atomicAdd(&hist[ch * 256], tot_sh - nz_sh);
}

// ------------------------------------------------------------------------------------------------
// K3: from_symbols + dequantize + inverse temporal lifting (stream over pairs); frames t < f only
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ I4 dequant4(uint32_t packed, int step) {
    I4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = (packed >> (8 * i)) & 0xFF;
        const int q = (s == 0) ? 0 : ((s & 1) ? (s + 1) / 2 : -(s / 2));  // src/quant.rs:581-587
        r.v[i] = (int)((unsigned)q * (unsigned)step);                      // src/quant.rs:104-110
    }
    return r;
}
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

template <int NS, bool EXACT, typename MidT>
__global__ __launch_bounds__(256) void inv_t_kernel(const uint8_t* __restrict__ sym, MidT* __restrict__ mid, ChunkDims d,
                                                    Coeffs cf, int step0, int step1, int step2) {
    const size_t plane = (size_t)d.pw * d.ph;
    const size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int ch = blockIdx.y;
    const int pf = d.pf, half = pf / 2, nf = d.f;
    if (idx >= plane) return;
    const int step = ch == 0 ? step0 : (ch == 1 ? step1 : step2);
    const uint32_t* src = (const uint32_t*)(sym + (size_t)ch * pf * plane + idx);
    MidT* dst = mid + (size_t)ch * pf * plane + idx;
    const size_t plane4 = plane / 4;
    // inverse step order (src/wavelet.rs:167-174): U2', P2', U1', P1' with negated coefficients
    const int c0 = -cf.c[0], c1 = -cf.c[1], c2 = -cf.c[2], c3 = -cf.c[3];
    I4 lo_c = dequant4(src[0], step), hi_c = dequant4(src[(size_t)half * plane4], step);  // E2[0], O2[0]
    I4 o2p{}, e1p{}, o1pp{}, e0pp{};
    const int last = (NS == 4) ? half + 1 : half;
    for (int j = 0; j <= last; ++j) {
        I4 lo_n{}, hi_n{};
        if (j + 1 < half) {
            lo_n = dequant4(src[(size_t)(j + 1) * plane4], step);
            hi_n = dequant4(src[(size_t)(half + j + 1) * plane4], step);
        }
        if (NS == 4) {
            I4 e1c{};
            const bool havec = j < half;
            if (havec) e1c = lift4<EXACT>(lo_c, (j >= 1) ? o2p : hi_c, hi_c, c3);        // U2': E1[j]
            if (j >= 1) {
                I4 o1p, e0p;
                const bool have1 = j <= half;
                if (have1) {
                    o1p = lift4<EXACT>(o2p, e1p, havec ? e1c : e1p, c2);                 // P2': O1[j-1]
                    e0p = lift4<EXACT>(e1p, (j >= 2) ? o1pp : o1p, o1p, c1);             // U1': E0[j-1]
                    if (2 * (j - 1) < nf) store4<MidT>(dst + (size_t)(2 * (j - 1)) * plane, e0p);
                }
                if (j >= 2) {
                    const I4 o0pp = lift4<EXACT>(o1pp, e0pp, have1 ? e0p : e0pp, c0);    // P1': O0[j-2]
                    if (2 * (j - 2) + 1 < nf) store4<MidT>(dst + (size_t)(2 * (j - 2) + 1) * plane, o0pp);
                }
                if (have1) { o1pp = o1p; e0pp = e0p; }
            }
            if (havec) { o2p = hi_c; e1p = e1c; }
        } else {
            I4 e0c{};
            const bool havec = j < half;
            if (havec) {
                e0c = lift4<EXACT>(lo_c, (j >= 1) ? o2p : hi_c, hi_c, c1);                // U1': E0[j]
                if (2 * j < nf) store4<MidT>(dst + (size_t)(2 * j) * plane, e0c);
            }
            if (j >= 1) {
                const I4 o0p = lift4<EXACT>(o2p, e1p, havec ? e0c : e1p, c0);            // P1': O0[j-1]
                if (2 * (j - 1) + 1 < nf) store4<MidT>(dst + (size_t)(2 * (j - 1) + 1) * plane, o0p);
            }
            if (havec) { o2p = hi_c; e1p = e0c; }
        }
        lo_c = lo_n; hi_c = hi_n;
    }
}

// ------------------------------------------------------------------------------------------------
// K4: inverse spatial transform of one frame tile + colour
// ------------------------------------------------------------------------------------------------
constexpr int I_TW = 96, I_TH = 32, I_SEG = 8, I_NSEG = 12, I_THREADS = 384;

template <int NS, bool EDGE, bool EXACT, typename MidT>
__global__ __launch_bounds__(I_THREADS) void inv_xy_kernel(const MidT* __restrict__ mid, uint8_t* __restrict__ rgb,
                                                           ChunkDims d, Coeffs cf, int aligned) {
    constexpr int H = NS;
    constexpr int ER = I_TH + 2 * H;        // rows incl. halo
    constexpr int EC = I_TW + 2 * H;        // columns incl. halo (104 / 100)
    constexpr int ECh = EC / 2;
    constexpr int LW = 105;                 // LDS row pitch (odd: stage-B reads stay at <= 2-way conflicts)
    constexpr int NL = I_SEG + 2 * H;       // samples lifted per segment
    __shared__ int lds[3 * ER * LW];
    const int tid = threadIdx.x;
    const unsigned ntx = (d.w + I_TW - 1) / I_TW, nty = (d.h + I_TH - 1) / I_TH;
    const unsigned lb = xcd_logical_block(ntx * nty * d.f);
    if (lb == 0xFFFFFFFFu) return;
    const int t = (int)(lb / (ntx * nty));
    const int gx0 = (int)((lb % (ntx * nty)) % ntx) * I_TW, gy0 = (int)((lb % (ntx * nty)) / ntx) * I_TH;
    const int pw = d.pw, ph = d.ph, hw = pw / 2, hh = ph / 2;
    const int gpx0 = (gx0 - H) / 2;  // first column pair of the extended tile (may be negative)

    // stage A: one thread per (extended column, channel): inverse column lifting in registers
    if (tid < 3 * EC) {
        const int ch = tid / EC, xq = tid % EC;
        const int par = xq / ECh, j = xq % ECh;
        const int gxp = gpx0 + j;
        if (!EDGE || (gxp >= 0 && gxp < hw)) {
            int v[ER];
            const MidT* src = mid + ((size_t)ch * d.pf + t) * ph * pw + (size_t)par * hw + gxp;
            const int gy_s = gy0 - H;
#pragma unroll
            for (int k = 0; k < ER; ++k) {
                const int gy = gy_s + k;
                int val = 0;
                if (!EDGE || (gy >= 0 && gy < ph)) {
                    const int yy = (gy & 1) * hh + (gy >> 1);
                    val = (int)src[(size_t)yy * pw];
                }
                v[k] = val;
            }
            lift_regs<ER, NS, EDGE, EXACT, true>(v, cf, gy_s, ph);
            int* L = lds + (ch * ER) * LW + xq;
#pragma unroll
            for (int k = 0; k < ER; ++k) L[k * LW] = v[k];
        }
    }
    __syncthreads();

    // stage B: one thread per (interior row, 8-pixel segment): inverse row lifting + colour
    {
        const int r = tid / I_NSEG, s = tid % I_NSEG;  // 384 = 32 * 12
        const int gy = gy0 + r;
        const int gxs = gx0 + s * I_SEG;               // first interior pixel of the segment
        if (gy < (int)d.h && gxs < (int)d.w) {
            int y[NL], co[NL], cg[NL];
            const int lx0 = s * I_SEG;                 // offset of sample 0 inside the extended tile (even)
            const int* L0 = lds + (0 * ER + r + H) * LW;
            const int* L1 = lds + (1 * ER + r + H) * LW;
            const int* L2 = lds + (2 * ER + r + H) * LW;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int lx = lx0 + k;
                const int q = (lx & 1) * ECh + (lx >> 1);
                y[k] = L0[q]; co[k] = L1[q]; cg[k] = L2[q];
            }
            const int g0 = gxs - H;
            lift_regs<NL, NS, EDGE, EXACT, true>(y, cf, g0, pw);
            lift_regs<NL, NS, EDGE, EXACT, true>(co, cf, g0, pw);
            lift_regs<NL, NS, EDGE, EXACT, true>(cg, cf, g0, pw);
            uint8_t out[I_SEG * 3];
#pragma unroll
            for (int k = 0; k < I_SEG; ++k) {
                // `as i16` (src/pipeline.rs:608) then wrapping i16 arithmetic (src/color.rs:266-273)
                const short yv = (short)y[H + k], c_o = (short)co[H + k], c_g = (short)cg[H + k];
                const short tt = (short)(yv - (short)(c_g >> 1));
                const short g = (short)(c_g + tt);
                const short b = (short)(tt - (short)(c_o >> 1));
                const short rr = (short)(c_o + b);
                out[3 * k] = (uint8_t)min(max((int)rr, 0), 255);
                out[3 * k + 1] = (uint8_t)min(max((int)g, 0), 255);
                out[3 * k + 2] = (uint8_t)min(max((int)b, 0), 255);
            }
            uint8_t* p = rgb + (((size_t)t * d.h + gy) * d.w + gxs) * 3;
            if (aligned && gxs + I_SEG <= (int)d.w) {
                uint32_t* p4 = (uint32_t*)p;
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    p4[i] = (uint32_t)out[4 * i] | ((uint32_t)out[4 * i + 1] << 8) | ((uint32_t)out[4 * i + 2] << 16) |
                            ((uint32_t)out[4 * i + 3] << 24);
            } else {
#pragma unroll
                for (int k = 0; k < I_SEG; ++k)
                    if (gxs + k < (int)d.w) { p[3 * k] = out[3 * k]; p[3 * k + 1] = out[3 * k + 1]; p[3 * k + 2] = out[3 * k + 2]; }
            }
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

// Tiles strictly inside the frame (no clamping, no mirroring) run the EDGE=false instance, launched as
// one interior rectangle of tiles; the four border strips run EDGE=true.
bool launch_forward_transform(const uint8_t* d_rgb, const ChunkDims& d, int wavelet, int32_t step,
                              int32_t* d_mid, uint8_t* d_sym, uint32_t* d_hist, hipStream_t st) {
    if (step < 1 || step > 64) return false;
    if ((unsigned long long)((d.pw + F_TW - 1) / F_TW) * ((d.ph + F_TH - 1) / F_TH) * d.pf > 0x7FFFFFF0ull) return false;
    const LiftSteps ls = lift_steps(wavelet);
    const Coeffs cf = to_coeffs(ls);
    int16_t* mid = (int16_t*)d_mid;
    const unsigned nx = (d.pw + F_TW - 1) / F_TW, ny = (d.ph + F_TH - 1) / F_TH;
    const int aligned = (d.w % 4 == 0) && ((((uintptr_t)d_rgb) & 3u) == 0u);
    // a tile is interior when its loaded range [gx0-4, gx0+128+4) x [gy0-H, gy0+40+H) lies inside w x h
    auto interior_x = [&](unsigned bx) { return bx >= 1 && (bx * F_TW + F_TW + 4) <= d.w; };
    auto interior_y = [&](unsigned by) { return by >= 1 && (by * F_TH + F_TH + 4) <= d.h; };
    unsigned ix0 = 1, ix1 = 1, iy0 = 1, iy1 = 1;
    while (ix1 < nx && interior_x(ix1)) ++ix1;
    while (iy1 < ny && interior_y(iy1)) ++iy1;
    const bool has_interior = ix1 > ix0 && iy1 > iy0;
    const size_t lds4 = (size_t)3 * (F_TH + 8) * F_LP * sizeof(int), lds2 = (size_t)3 * (F_TH + 4) * F_LP * sizeof(int);
    dim3 block(F_THREADS);
    TileMap tm{(int)nx, (int)ny, 0, 0, 0, 0};
    if (has_interior) { tm.ix0 = (int)ix0; tm.ix1 = (int)ix1; tm.iy0 = (int)iy0; tm.iy1 = (int)iy1; }
    else { tm.iy0 = (int)ny; tm.iy1 = (int)ny; }  // everything is "top strip"
    const unsigned n_interior_x = has_interior ? ix1 - ix0 : 0, n_interior_y = has_interior ? iy1 - iy0 : 0;
    const unsigned n_edge = nx * ny - n_interior_x * n_interior_y;
    if (ls.n == 4) {
        static const bool ok = set_dyn_lds(fwd_xy_kernel<4, true>, lds4) && set_dyn_lds(fwd_xy_kernel<4, false>, lds4);
        (void)ok;
        if (has_interior) hipLaunchKernelGGL((fwd_xy_kernel<4, false>), dim3(xcd_grid((unsigned long long)n_interior_x * n_interior_y * d.pf)), block, lds4, st, d_rgb, mid, d, cf, aligned, tm);
        if (n_edge) hipLaunchKernelGGL((fwd_xy_kernel<4, true>), dim3(xcd_grid((unsigned long long)n_edge * d.pf)), block, lds4, st, d_rgb, mid, d, cf, aligned, tm);
    } else {
        static const bool ok = set_dyn_lds(fwd_xy_kernel<2, true>, lds2) && set_dyn_lds(fwd_xy_kernel<2, false>, lds2);
        (void)ok;
        if (has_interior) hipLaunchKernelGGL((fwd_xy_kernel<2, false>), dim3(xcd_grid((unsigned long long)n_interior_x * n_interior_y * d.pf)), block, lds2, st, d_rgb, mid, d, cf, aligned, tm);
        if (n_edge) hipLaunchKernelGGL((fwd_xy_kernel<2, true>), dim3(xcd_grid((unsigned long long)n_edge * d.pf)), block, lds2, st, d_rgb, mid, d, cf, aligned, tm);
    }
    const size_t plane = (size_t)d.pw * d.ph;
    const uint32_t magic = step == 1 ? 0u : (uint32_t)(((1ull << 32) + (uint32_t)step - 1u) / (uint32_t)step);
    dim3 gt((unsigned)((plane / 4 + 255) / 256), 3);
    // ALICE_CODEC_SPLIT_HIST=1: histogram in a separate pass over the symbols instead of fused LDS atomics (A/B switch)
    static const bool split_hist = getenv("ALICE_CODEC_SPLIT_HIST") != nullptr;
#define ALICE_FWD_T(NS_, S1_, H_) hipLaunchKernelGGL((fwd_t_kernel<NS_, S1_, H_>), gt, dim3(256), 0, st, mid, d_sym, d_hist, d, cf, step, magic)
    if (ls.n == 4) {
        if (split_hist) { if (step == 1) ALICE_FWD_T(4, true, false); else ALICE_FWD_T(4, false, false); }
        else { if (step == 1) ALICE_FWD_T(4, true, true); else ALICE_FWD_T(4, false, true); }
    } else {
        if (split_hist) { if (step == 1) ALICE_FWD_T(2, true, false); else ALICE_FWD_T(2, false, false); }
        else { if (step == 1) ALICE_FWD_T(2, true, true); else ALICE_FWD_T(2, false, true); }
    }
#undef ALICE_FWD_T
    if (split_hist)
        for (int c = 0; c < 3; ++c) launch_histogram(d_sym + (size_t)c * d.padded, d.padded, d_hist + c * 256, st);
    return true;
}

template <int NS, bool EXACT, typename MidT>
static void inv_launch(const uint8_t* sym, MidT* mid, uint8_t* rgb, const ChunkDims& d, Coeffs cf, const int32_t step[3],
                       hipStream_t st) {
    const size_t plane = (size_t)d.pw * d.ph;
    dim3 gt((unsigned)((plane / 4 + 255) / 256), 3);
    hipLaunchKernelGGL((inv_t_kernel<NS, EXACT, MidT>), gt, dim3(256), 0, st, sym, mid, d, cf, step[0], step[1], step[2]);
    const unsigned nx = (d.w + I_TW - 1) / I_TW, ny = (d.h + I_TH - 1) / I_TH;
    const int aligned = (d.w % 4 == 0) && ((((uintptr_t)rgb) & 3u) == 0u);
    dim3 grid(xcd_grid((unsigned long long)nx * ny * d.f));
    // the inverse tiles are few enough per frame that one EDGE=true instance serves all of them
    hipLaunchKernelGGL((inv_xy_kernel<NS, true, EXACT, MidT>), grid, dim3(I_THREADS), 0, st, mid, rgb, d, cf, aligned);
}

bool launch_inverse_transform(const uint8_t* d_sym, const ChunkDims& d, int wavelet, const int32_t step[3],
                              bool exact, bool mid16, int32_t* d_mid, uint8_t* d_rgb, hipStream_t st) {
    const LiftSteps ls = lift_steps(wavelet);
    const Coeffs cf = to_coeffs(ls);
    if ((unsigned long long)((d.w + I_TW - 1) / I_TW) * ((d.h + I_TH - 1) / I_TH) * d.f > 0x7FFFFFF0ull) return false;
    // mid16: the host proved every value after the inverse temporal pass fits i16 (then exact is false too)
    if (ls.n == 4) {
        if (exact) inv_launch<4, true, int32_t>(d_sym, d_mid, d_rgb, d, cf, step, st);
        else if (mid16) inv_launch<4, false, int16_t>(d_sym, (int16_t*)d_mid, d_rgb, d, cf, step, st);
        else inv_launch<4, false, int32_t>(d_sym, d_mid, d_rgb, d, cf, step, st);
    } else {
        if (exact) inv_launch<2, true, int32_t>(d_sym, d_mid, d_rgb, d, cf, step, st);
        else if (mid16) inv_launch<2, false, int16_t>(d_sym, (int16_t*)d_mid, d_rgb, d, cf, step, st);
        else inv_launch<2, false, int32_t>(d_sym, d_mid, d_rgb, d, cf, step, st);
    }
    return true;
}

}  // namespace alice
