// Pipeline-specialised transform kernels for gfx950: RGB <-> u8 symbols.
//
// Forward (FrameEncoder::encode, reference src/pipeline.rs:429-477):
//   fwd_xy_kernel : RGB -> YCoCg-R (src/color.rs:221-228) -> edge-replicated padding
//                   (src/pipeline.rs:77-114, done by index clamping) -> row lifting -> column
//                   lifting per frame (src/wavelet.rs:401-417), all inside one LDS tile with a
//                   halo of one sample per lifting step; writes [L|H]-deinterleaved planes.
//   fwd_t_kernel  : temporal lifting with the whole temporal vector of a pixel in registers
//                   (src/wavelet.rs:421-437), fused Quantizer::quantize (src/quant.rs:89-97),
//                   to_symbols (src/quant.rs:555-560) and build_histogram (:594-600).
// Inverse (FrameDecoder::decode, src/pipeline.rs:588-621) mirrors it:
//   inv_t_kernel  : from_symbols, dequantize, inverse temporal lifting.
//   inv_xy_kernel : inverse column then row lifting, `as i16`, ycocg_r_to_rgb_bytes.
//
// Lifting facts used: inside one lifting step every write depends only on samples of the
// other parity, so a step is data-parallel; an output depends on inputs within +-n_steps
// samples; mirroring happens only at the true signal ends (src/wavelet.rs:186-190,206-210).
// A tile whose halo is recomputed therefore yields the same integers as the global pass.
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
        return ((a + b) * c + 4096) >> 13;
    }
}
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }

struct Coeffs { int c[4]; };

constexpr int TW = 64;  // tile interior width  (even)
constexpr int TH = 32;  // tile interior height (even)

// One in-place lifting step over all lines of a deinterleaved LDS tile.
// AXIS 0: along x (pairs indexed by j in [0, EWh)); AXIS 1: along y.
// Storage: L[ch][(ly&1)*EHh + ly/2][(lx&1)*EWh + lx/2].
template <int EW, int EH, bool EXACT, bool PREDICT, int AXIS>
__device__ __forceinline__ void tile_lift_step(int (*L)[EH][EW], int coeff, int gpair0, int n_axis, int tid) {
    constexpr int EWh = EW / 2, EHh = EH / 2;
    if (AXIS == 0) {
        constexpr int items = 3 * EH * EWh;
        for (int it = tid; it < items; it += 256) {
            const int j = it % EWh;
            const int row = (it / EWh) % EH;
            const int ch = it / (EWh * EH);
            const int gi = gpair0 + j;
            int* E = &L[ch][row][0];
            int* O = &L[ch][row][EWh];
            if (PREDICT) {
                const int jn = (j + 1 < EWh && 2 * gi + 2 < n_axis) ? j + 1 : j;
                O[j] = wadd(O[j], lift_delta<EXACT>(E[j], E[jn], coeff));
            } else {
                const int jl = (j > 0 && gi > 0) ? j - 1 : j;
                E[j] = wadd(E[j], lift_delta<EXACT>(O[jl], O[j], coeff));
            }
        }
    } else {
        constexpr int items = 3 * EHh * EW;
        for (int it = tid; it < items; it += 256) {
            const int col = it % EW;
            const int i = (it / EW) % EHh;
            const int ch = it / (EW * EHh);
            const int gi = gpair0 + i;
            if (PREDICT) {
                const int in = (i + 1 < EHh && 2 * gi + 2 < n_axis) ? i + 1 : i;
                L[ch][EHh + i][col] = wadd(L[ch][EHh + i][col], lift_delta<EXACT>(L[ch][i][col], L[ch][in][col], coeff));
            } else {
                const int il = (i > 0 && gi > 0) ? i - 1 : i;
                L[ch][i][col] = wadd(L[ch][i][col], lift_delta<EXACT>(L[ch][EHh + il][col], L[ch][EHh + i][col], coeff));
            }
        }
    }
    __syncthreads();
}

template <int NS>
__global__ __launch_bounds__(256) void fwd_xy_kernel(const uint8_t* __restrict__ rgb, int32_t* __restrict__ mid,
                                                     ChunkDims d, Coeffs cf) {
    constexpr int H = NS;  // halo per side (even)
    constexpr int EW = TW + 2 * H, EH = TH + 2 * H, EWh = EW / 2, EHh = EH / 2;
    __shared__ int L[3][EH][EW];
    const int tid = threadIdx.x;
    const int gx0 = blockIdx.x * TW, gy0 = blockIdx.y * TH, t = blockIdx.z;
    const int pw = d.pw, ph = d.ph;
    const int st = min(t, (int)d.f - 1);

    for (int it = tid; it < EH * EW; it += 256) {
        const int lx = it % EW, ly = it / EW;
        const int gx = gx0 - H + lx, gy = gy0 - H + ly;
        int y = 0, co = 0, cg = 0;
        if (gx >= 0 && gx < pw && gy >= 0 && gy < ph) {
            const int sx = min(gx, (int)d.w - 1), sy = min(gy, (int)d.h - 1);
            const uint8_t* p = rgb + (((size_t)st * d.h + sy) * d.w + sx) * 3;
            const int r = p[0], g = p[1], b = p[2];
            co = r - b;
            const int tt = b + (co >> 1);
            cg = g - tt;
            y = tt + (cg >> 1);
        }
        const int rr = (ly & 1) * EHh + (ly >> 1), cc = (lx & 1) * EWh + (lx >> 1);
        L[0][rr][cc] = y;
        L[1][rr][cc] = co;
        L[2][rr][cc] = cg;
    }
    __syncthreads();

    const int gpx0 = (gx0 - H) / 2, gpy0 = (gy0 - H) / 2;  // H even, gx0 even -> exact (also when negative)
    tile_lift_step<EW, EH, false, true, 0>(L, cf.c[0], gpx0, pw, tid);
    tile_lift_step<EW, EH, false, false, 0>(L, cf.c[1], gpx0, pw, tid);
    if (NS == 4) {
        tile_lift_step<EW, EH, false, true, 0>(L, cf.c[2], gpx0, pw, tid);
        tile_lift_step<EW, EH, false, false, 0>(L, cf.c[3], gpx0, pw, tid);
    }
    tile_lift_step<EW, EH, false, true, 1>(L, cf.c[0], gpy0, ph, tid);
    tile_lift_step<EW, EH, false, false, 1>(L, cf.c[1], gpy0, ph, tid);
    if (NS == 4) {
        tile_lift_step<EW, EH, false, true, 1>(L, cf.c[2], gpy0, ph, tid);
        tile_lift_step<EW, EH, false, false, 1>(L, cf.c[3], gpy0, ph, tid);
    }

    // interior -> mid[ch][t][yy][xx], already deinterleaved ([L|H] along x and y)
    const int hw = pw / 2, hh = ph / 2;
    for (int it = tid; it < 3 * TH * TW; it += 256) {
        const int c = it % TW;             // 0..31 even half, 32..63 odd half
        const int r = (it / TW) % TH;
        const int ch = it / (TW * TH);
        const int px = c / (TW / 2), jx = c % (TW / 2);
        const int py = r / (TH / 2), jy = r % (TH / 2);
        const int gxp = gx0 / 2 + jx, gyp = gy0 / 2 + jy;  // global pair indices
        if (gxp < hw && gyp < hh) {
            const int v = L[ch][py * EHh + H / 2 + jy][px * EWh + H / 2 + jx];
            mid[(((size_t)ch * d.pf + t) * ph + (py * hh + gyp)) * pw + (px * hw + gxp)] = v;
        }
    }
}

// Temporal lifting + quantise + symbol + histogram.  One thread per (channel, y, x).
template <int NS, int MAXF>
__global__ __launch_bounds__(256) void fwd_t_kernel(const int32_t* __restrict__ mid, uint8_t* __restrict__ sym,
                                                    uint32_t* __restrict__ hist, ChunkDims d, Coeffs cf,
                                                    int step, uint32_t magic) {
    __shared__ uint32_t lh[256];
    const int tid = threadIdx.x;
    lh[tid] = 0u;
    __syncthreads();
    const size_t plane = (size_t)d.pw * d.ph;
    const size_t idx = (size_t)blockIdx.x * 256 + tid;
    const int ch = blockIdx.y;
    const int pf = d.pf, half = pf / 2;
    const bool live = idx < plane;
    uint32_t zeros = 0u;
    if (live) {
        int v[MAXF];
        const int32_t* src = mid + (size_t)ch * pf * plane + idx;
#pragma unroll
        for (int t = 0; t < MAXF; ++t) v[t] = (t < pf) ? src[(size_t)t * plane] : 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int c = cf.c[k];
            if ((k & 1) == 0) {
#pragma unroll
                for (int i = 0; i < MAXF / 2; ++i) {
                    if (i < half) {
                        const int right = (2 * i + 2 < MAXF && 2 * i + 2 < pf) ? v[(2 * i + 2) % MAXF] : v[2 * i];
                        v[2 * i + 1] += lift_delta<false>(v[2 * i], right, c);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < MAXF / 2; ++i) {
                    if (i < half) {
                        const int left = (i > 0) ? v[(2 * i - 1 + MAXF) % MAXF] : v[1];
                        v[2 * i] += lift_delta<false>(left, v[2 * i + 1], c);
                    }
                }
            }
        }
        uint8_t* dst = sym + (size_t)ch * pf * plane + idx;
        const int hdz = step / 2;
#pragma unroll
        for (int t = 0; t < MAXF; ++t) {
            if (t < pf) {
                const int val = v[t];
                const int mag = val < 0 ? -val : val;
                uint32_t s = 0u;
                if (mag >= step) {  // dead zone = step (Quantizer::new)
                    const uint32_t adj = (uint32_t)(mag - hdz);
                    const uint32_t q = (step == 1) ? adj : __umulhi(adj, magic);
                    // q can be 0 just above the dead zone ((mag - step/2) / step); to_symbols maps 0 -> 0
                    s = (q == 0u) ? 0u : ((val > 0) ? (2u * q - 1u) : (2u * q));
                    s &= 0xFFu;  // `as u8`
                }
                const int tt = (t & 1) * half + (t >> 1);
                dst[(size_t)tt * plane] = (uint8_t)s;
                if (s == 0u) ++zeros;
                else atomicAdd(&lh[s], 1u);
            }
        }
    }
    if (zeros) atomicAdd(&lh[0], zeros);
    __syncthreads();
    const uint32_t cnt = lh[tid];
    if (cnt) atomicAdd(&hist[ch * 256 + tid], cnt);
}

// from_symbols + dequantize + inverse temporal lifting.  Writes frames t < f only.
template <int NS, int MAXF, bool EXACT>
__global__ __launch_bounds__(256) void inv_t_kernel(const uint8_t* __restrict__ sym, int32_t* __restrict__ mid,
                                                    ChunkDims d, Coeffs cf, int step0, int step1, int step2) {
    const size_t plane = (size_t)d.pw * d.ph;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y;
    const int pf = d.pf, half = pf / 2;
    if (idx >= plane) return;
    const int step = ch == 0 ? step0 : (ch == 1 ? step1 : step2);
    int v[MAXF];
    const uint8_t* src = sym + (size_t)ch * pf * plane + idx;
#pragma unroll
    for (int t = 0; t < MAXF; ++t) {
        int val = 0;
        if (t < pf) {
            const int tt = (t & 1) * half + (t >> 1);  // interleave: v[2i] = low[i], v[2i+1] = high[i]
            const int s = src[(size_t)tt * plane];
            const int q = (s == 0) ? 0 : ((s & 1) ? (s + 1) / 2 : -(s / 2));  // src/quant.rs:581-587
            val = (int)((unsigned)q * (unsigned)step);                       // src/quant.rs:104-110
        }
        v[t] = val;
    }
#pragma unroll
    for (int k = NS - 1; k >= 0; --k) {
        const int c = -cf.c[k];  // src/wavelet.rs:167-174
        if ((k & 1) == 0) {
#pragma unroll
            for (int i = 0; i < MAXF / 2; ++i) {
                if (i < half) {
                    const int right = (2 * i + 2 < MAXF && 2 * i + 2 < pf) ? v[(2 * i + 2) % MAXF] : v[2 * i];
                    v[2 * i + 1] = wadd(v[2 * i + 1], lift_delta<EXACT>(v[2 * i], right, c));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MAXF / 2; ++i) {
                if (i < half) {
                    const int left = (i > 0) ? v[(2 * i - 1 + MAXF) % MAXF] : v[1];
                    v[2 * i] = wadd(v[2 * i], lift_delta<EXACT>(left, v[2 * i + 1], c));
                }
            }
        }
    }
    int32_t* dst = mid + (size_t)ch * pf * plane + idx;
#pragma unroll
    for (int t = 0; t < MAXF; ++t)
        if (t < (int)d.f) dst[(size_t)t * plane] = v[t];
}

template <int NS, bool EXACT>
__global__ __launch_bounds__(256) void inv_xy_kernel(const int32_t* __restrict__ mid, uint8_t* __restrict__ rgb,
                                                     ChunkDims d, Coeffs cf) {
    constexpr int H = NS;
    constexpr int EW = TW + 2 * H, EH = TH + 2 * H, EWh = EW / 2, EHh = EH / 2;
    __shared__ int L[3][EH][EW];
    const int tid = threadIdx.x;
    const int gx0 = blockIdx.x * TW, gy0 = blockIdx.y * TH, t = blockIdx.z;
    const int pw = d.pw, ph = d.ph, hw = pw / 2, hh = ph / 2;
    const int gpx0 = (gx0 - H) / 2, gpy0 = (gy0 - H) / 2;

    for (int it = tid; it < 3 * EH * EW; it += 256) {
        const int cc = it % EW, rr = (it / EW) % EH, ch = it / (EW * EH);
        const int px = cc / EWh, jx = cc % EWh, py = rr / EHh, jy = rr % EHh;
        const int gxp = gpx0 + jx, gyp = gpy0 + jy;
        int v = 0;
        if (gxp >= 0 && gxp < hw && gyp >= 0 && gyp < hh)
            v = mid[(((size_t)ch * d.pf + t) * ph + (py * hh + gyp)) * pw + (px * hw + gxp)];
        L[ch][rr][cc] = v;
    }
    __syncthreads();

    // columns first, then rows (src/wavelet.rs:465-482); steps reversed with negated coefficients
    if (NS == 4) {
        tile_lift_step<EW, EH, EXACT, false, 1>(L, -cf.c[3], gpy0, ph, tid);
        tile_lift_step<EW, EH, EXACT, true, 1>(L, -cf.c[2], gpy0, ph, tid);
    }
    tile_lift_step<EW, EH, EXACT, false, 1>(L, -cf.c[1], gpy0, ph, tid);
    tile_lift_step<EW, EH, EXACT, true, 1>(L, -cf.c[0], gpy0, ph, tid);
    if (NS == 4) {
        tile_lift_step<EW, EH, EXACT, false, 0>(L, -cf.c[3], gpx0, pw, tid);
        tile_lift_step<EW, EH, EXACT, true, 0>(L, -cf.c[2], gpx0, pw, tid);
    }
    tile_lift_step<EW, EH, EXACT, false, 0>(L, -cf.c[1], gpx0, pw, tid);
    tile_lift_step<EW, EH, EXACT, true, 0>(L, -cf.c[0], gpx0, pw, tid);

    for (int it = tid; it < TH * TW; it += 256) {
        const int lx = it % TW, ly = it / TW;
        const int gx = gx0 + lx, gy = gy0 + ly;
        if (gx < (int)d.w && gy < (int)d.h) {
            const int rr = ((ly + H) & 1) * EHh + ((ly + H) >> 1), cc = ((lx + H) & 1) * EWh + ((lx + H) >> 1);
            // `as i16` (src/pipeline.rs:608) then wrapping i16 arithmetic (src/color.rs:266-273)
            const short yv = (short)L[0][rr][cc], co = (short)L[1][rr][cc], cg = (short)L[2][rr][cc];
            const short tt = (short)(yv - (short)(cg >> 1));
            const short g = (short)(cg + tt);
            const short b = (short)(tt - (short)(co >> 1));
            const short r = (short)(co + b);
            uint8_t* p = rgb + (((size_t)t * d.h + gy) * d.w + gx) * 3;
            p[0] = (uint8_t)min(max((int)r, 0), 255);
            p[1] = (uint8_t)min(max((int)g, 0), 255);
            p[2] = (uint8_t)min(max((int)b, 0), 255);
        }
    }
}

// ----------------------------------------------------------------------------------

static Coeffs to_coeffs(const LiftSteps& s) {
    Coeffs c{};
    for (int i = 0; i < 4; ++i) c.c[i] = i < s.n ? s.coeff[i] : 0;
    return c;
}

template <int NS>
static void fwd_t_dispatch(const int32_t* mid, uint8_t* sym, uint32_t* hist, const ChunkDims& d, Coeffs cf,
                           int step, uint32_t magic, hipStream_t st) {
    const size_t plane = (size_t)d.pw * d.ph;
    dim3 grid((unsigned)((plane + 255) / 256), 3), block(256);
    if (d.pf <= 8) hipLaunchKernelGGL((fwd_t_kernel<NS, 8>), grid, block, 0, st, mid, sym, hist, d, cf, step, magic);
    else if (d.pf <= 16) hipLaunchKernelGGL((fwd_t_kernel<NS, 16>), grid, block, 0, st, mid, sym, hist, d, cf, step, magic);
    else if (d.pf <= 32) hipLaunchKernelGGL((fwd_t_kernel<NS, 32>), grid, block, 0, st, mid, sym, hist, d, cf, step, magic);
    else hipLaunchKernelGGL((fwd_t_kernel<NS, 64>), grid, block, 0, st, mid, sym, hist, d, cf, step, magic);
}

bool launch_forward_transform(const uint8_t* d_rgb, const ChunkDims& d, int wavelet, int32_t step,
                              int32_t* d_mid, uint8_t* d_sym, uint32_t* d_hist, hipStream_t st) {
    if (d.pf > 64 || step < 1 || step > 64) return false;
    const LiftSteps ls = lift_steps(wavelet);
    const Coeffs cf = to_coeffs(ls);
    dim3 grid((d.pw + TW - 1) / TW, (d.ph + TH - 1) / TH, d.pf), block(256);
    if (grid.y > 65535u || grid.z > 65535u) return false;
    // adj * step < 2^32 (|coefficient| < 2^15 from u8 input, step <= 64) makes umulhi(adj, ceil(2^32/step)) exact
    const uint32_t magic = step == 1 ? 0u : (uint32_t)(((1ull << 32) + (uint32_t)step - 1u) / (uint32_t)step);
    if (ls.n == 4) {
        hipLaunchKernelGGL((fwd_xy_kernel<4>), grid, block, 0, st, d_rgb, d_mid, d, cf);
        fwd_t_dispatch<4>(d_mid, d_sym, d_hist, d, cf, step, magic, st);
    } else {
        hipLaunchKernelGGL((fwd_xy_kernel<2>), grid, block, 0, st, d_rgb, d_mid, d, cf);
        fwd_t_dispatch<2>(d_mid, d_sym, d_hist, d, cf, step, magic, st);
    }
    return true;
}

template <int NS, bool EXACT>
static void inv_dispatch(const uint8_t* sym, int32_t* mid, uint8_t* rgb, const ChunkDims& d, Coeffs cf,
                         const int32_t step[3], hipStream_t st) {
    const size_t plane = (size_t)d.pw * d.ph;
    dim3 gt((unsigned)((plane + 255) / 256), 3), block(256);
    if (d.pf <= 8) hipLaunchKernelGGL((inv_t_kernel<NS, 8, EXACT>), gt, block, 0, st, sym, mid, d, cf, step[0], step[1], step[2]);
    else if (d.pf <= 16) hipLaunchKernelGGL((inv_t_kernel<NS, 16, EXACT>), gt, block, 0, st, sym, mid, d, cf, step[0], step[1], step[2]);
    else if (d.pf <= 32) hipLaunchKernelGGL((inv_t_kernel<NS, 32, EXACT>), gt, block, 0, st, sym, mid, d, cf, step[0], step[1], step[2]);
    else hipLaunchKernelGGL((inv_t_kernel<NS, 64, EXACT>), gt, block, 0, st, sym, mid, d, cf, step[0], step[1], step[2]);
    dim3 gxy((d.pw + TW - 1) / TW, (d.ph + TH - 1) / TH, d.f);
    hipLaunchKernelGGL((inv_xy_kernel<NS, EXACT>), gxy, block, 0, st, mid, rgb, d, cf);
}

bool launch_inverse_transform(const uint8_t* d_sym, const ChunkDims& d, int wavelet, const int32_t step[3],
                              bool exact, int32_t* d_mid, uint8_t* d_rgb, hipStream_t st) {
    if (d.pf > 64) return false;
    const LiftSteps ls = lift_steps(wavelet);
    const Coeffs cf = to_coeffs(ls);
    if ((d.ph + TH - 1) / TH > 65535u || d.f > 65535u) return false;
    if (ls.n == 4) {
        if (exact) inv_dispatch<4, true>(d_sym, d_mid, d_rgb, d, cf, step, st);
        else inv_dispatch<4, false>(d_sym, d_mid, d_rgb, d, cf, step, st);
    } else {
        if (exact) inv_dispatch<2, true>(d_sym, d_mid, d_rgb, d, cf, step, st);
        else inv_dispatch<2, false>(d_sym, d_mid, d_rgb, d, cf, step, st);
    }
    return true;
}

}  // namespace alice
