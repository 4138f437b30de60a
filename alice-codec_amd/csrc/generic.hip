// Stage-level kernels on caller-shaped data (exact reference arithmetic, any length,
// including odd lengths and arbitrary i32 magnitudes), plus the small byte-moving kernels
// that assemble `.alc` buffers on the device.  These back the public Wavelet1D/2D/3D,
// Quantizer/FastQuantizer, to_symbols/from_symbols/build_histogram and colour entry points,
// and the pipeline when a chunk has more than 64 (padded) frames.
#include "common.h"
#include "kernels.h"

namespace alice {

__device__ __forceinline__ int g_delta(int a, int b, int c) {
    // reference src/wavelet.rs:193-194
    const int avg = (int)((unsigned)a + (unsigned)b);
    return (int)(((long long)avg * (long long)c + 4096ll) >> 13);
}

// One lifting step over every line.  work item = (line, pair); lines are (a, b) pairs.
template <bool PREDICT>
__global__ __launch_bounds__(256) void axis_lift_kernel(int32_t* __restrict__ data, unsigned long long n,
                                                        unsigned long long stride_k, unsigned long long n_a,
                                                        unsigned long long stride_a, unsigned long long n_b,
                                                        unsigned long long stride_b, int coeff, int line_fast) {
    const unsigned long long half = n / 2, lines = n_a * n_b, total = half * lines;
    for (unsigned long long it = (unsigned long long)blockIdx.x * 256 + threadIdx.x; it < total;
         it += (unsigned long long)gridDim.x * 256) {
        unsigned long long i, line;
        if (line_fast) { line = it % lines; i = it / lines; }   // adjacent threads on adjacent lines
        else { i = it % half; line = it / half; }               // adjacent threads along the line
        const unsigned long long a = line / n_b, b = line % n_b;
        int32_t* s = data + a * stride_a + b * stride_b;
        if (PREDICT) {  // src/wavelet.rs:184-196
            const int el = s[(2 * i) * stride_k];
            const int er = (2 * i + 2 < n) ? s[(2 * i + 2) * stride_k] : el;
            int32_t* o = s + (2 * i + 1) * stride_k;
            *o = (int)((unsigned)*o + (unsigned)g_delta(el, er, coeff));
        } else {        // src/wavelet.rs:205-216
            const int orr = s[(2 * i + 1) * stride_k];
            const int ol = (i > 0) ? s[(2 * i - 1) * stride_k] : s[stride_k];
            int32_t* e = s + (2 * i) * stride_k;
            *e = (int)((unsigned)*e + (unsigned)g_delta(ol, orr, coeff));
        }
    }
}

// deinterleave (src/wavelet.rs:220-233) / interleave (:236-248) into tmp; an odd tail becomes 0.
template <bool INTERLEAVE>
__global__ __launch_bounds__(256) void axis_shuffle_kernel(const int32_t* __restrict__ data, int32_t* __restrict__ tmp,
                                                           unsigned long long n, unsigned long long stride_k,
                                                           unsigned long long n_a, unsigned long long stride_a,
                                                           unsigned long long n_b, unsigned long long stride_b,
                                                           int line_fast) {
    const unsigned long long half = n / 2, lines = n_a * n_b, total = n * lines;
    for (unsigned long long it = (unsigned long long)blockIdx.x * 256 + threadIdx.x; it < total;
         it += (unsigned long long)gridDim.x * 256) {
        unsigned long long k, line;
        if (line_fast) { line = it % lines; k = it / lines; }
        else { k = it % n; line = it / n; }
        const unsigned long long a = line / n_b, b = line % n_b;
        const unsigned long long base = a * stride_a + b * stride_b;
        int v = 0;  // temp is zero-initialised in the reference
        if (k < 2 * half) {
            unsigned long long src;
            if (INTERLEAVE) src = (k & 1ull) ? half + (k >> 1) : (k >> 1);   // tmp[2i]=s[i], tmp[2i+1]=s[half+i]
            else src = (k < half) ? 2 * k : 2 * (k - half) + 1;              // tmp[i]=s[2i], tmp[half+i]=s[2i+1]
            v = data[base + src * stride_k];
        }
        tmp[base + k * stride_k] = v;
    }
}

__global__ __launch_bounds__(256) void axis_copy_kernel(const int32_t* __restrict__ tmp, int32_t* __restrict__ data,
                                                        unsigned long long n, unsigned long long stride_k,
                                                        unsigned long long n_a, unsigned long long stride_a,
                                                        unsigned long long n_b, unsigned long long stride_b,
                                                        int line_fast) {
    const unsigned long long lines = n_a * n_b, total = n * lines;
    for (unsigned long long it = (unsigned long long)blockIdx.x * 256 + threadIdx.x; it < total;
         it += (unsigned long long)gridDim.x * 256) {
        unsigned long long k, line;
        if (line_fast) { line = it % lines; k = it / lines; }
        else { k = it % n; line = it / n; }
        const unsigned long long a = line / n_b, b = line % n_b;
        const unsigned long long off = a * stride_a + b * stride_b + k * stride_k;
        data[off] = tmp[off];
    }
}

static unsigned grid_for(unsigned long long items) {
    unsigned long long g = (items + 255) / 256;
    if (g < 1) g = 1;
    if (g > 65535ull * 4) g = 65535ull * 4;
    return (unsigned)g;
}

void launch_wavelet_axis(int32_t* d_data, int32_t* d_tmp, uint64_t n, uint64_t stride_k, uint64_t n_a,
                         uint64_t stride_a, uint64_t n_b, uint64_t stride_b, int wavelet, bool inverse,
                         hipStream_t st) {
    if (n < 2 || n_a == 0 || n_b == 0) return;  // src/wavelet.rs:135,159
    const LiftSteps ls = lift_steps(wavelet);
    const int line_fast = stride_k != 1 ? 1 : 0;
    const unsigned long long lines = n_a * n_b;
    const unsigned gl = grid_for((n / 2) * lines), gs = grid_for(n * lines);
    dim3 block(256);
#define AX_ARGS (unsigned long long)n, (unsigned long long)stride_k, (unsigned long long)n_a, \
                (unsigned long long)stride_a, (unsigned long long)n_b, (unsigned long long)stride_b
    if (!inverse) {
        for (int k = 0; k < ls.n; ++k) {
            if ((k & 1) == 0) hipLaunchKernelGGL(axis_lift_kernel<true>, dim3(gl), block, 0, st, d_data, AX_ARGS, ls.coeff[k], line_fast);
            else hipLaunchKernelGGL(axis_lift_kernel<false>, dim3(gl), block, 0, st, d_data, AX_ARGS, ls.coeff[k], line_fast);
        }
        hipLaunchKernelGGL(axis_shuffle_kernel<false>, dim3(gs), block, 0, st, d_data, d_tmp, AX_ARGS, line_fast);
        hipLaunchKernelGGL(axis_copy_kernel, dim3(gs), block, 0, st, d_tmp, d_data, AX_ARGS, line_fast);
    } else {
        hipLaunchKernelGGL(axis_shuffle_kernel<true>, dim3(gs), block, 0, st, d_data, d_tmp, AX_ARGS, line_fast);
        hipLaunchKernelGGL(axis_copy_kernel, dim3(gs), block, 0, st, d_tmp, d_data, AX_ARGS, line_fast);
        for (int k = ls.n - 1; k >= 0; --k) {
            if ((k & 1) == 0) hipLaunchKernelGGL(axis_lift_kernel<true>, dim3(gl), block, 0, st, d_data, AX_ARGS, -ls.coeff[k], line_fast);
            else hipLaunchKernelGGL(axis_lift_kernel<false>, dim3(gl), block, 0, st, d_data, AX_ARGS, -ls.coeff[k], line_fast);
        }
    }
#undef AX_ARGS
}

// ---- colour ----------------------------------------------------------------------------

__global__ __launch_bounds__(256) void rgb_to_ycocg_kernel(const uint8_t* __restrict__ rgb, unsigned long long n,
                                                           int16_t* __restrict__ y, int16_t* __restrict__ co,
                                                           int16_t* __restrict__ cg) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        const int r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];  // src/color.rs:221-228
        const int c_o = r - b;
        const int t = b + (c_o >> 1);
        const int c_g = g - t;
        y[i] = (int16_t)(t + (c_g >> 1));
        co[i] = (int16_t)c_o;
        cg[i] = (int16_t)c_g;
    }
}

__global__ __launch_bounds__(256) void ycocg_to_rgb_kernel(const int16_t* __restrict__ y, const int16_t* __restrict__ co,
                                                           const int16_t* __restrict__ cg, unsigned long long n,
                                                           uint8_t* __restrict__ rgb) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        const short yv = y[i], c_o = co[i], c_g = cg[i];  // src/color.rs:266-273, i16 wrapping
        const short t = (short)(yv - (short)(c_g >> 1));
        const short g = (short)(c_g + t);
        const short b = (short)(t - (short)(c_o >> 1));
        const short r = (short)(c_o + b);
        rgb[3 * i] = (uint8_t)min(max((int)r, 0), 255);
        rgb[3 * i + 1] = (uint8_t)min(max((int)g, 0), 255);
        rgb[3 * i + 2] = (uint8_t)min(max((int)b, 0), 255);
    }
}

void launch_rgb_to_ycocg(const uint8_t* d_rgb, uint64_t n_pixels, int16_t* y, int16_t* co, int16_t* cg, hipStream_t st) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(rgb_to_ycocg_kernel, dim3(grid_for(n_pixels)), dim3(256), 0, st, d_rgb,
                       (unsigned long long)n_pixels, y, co, cg);
}
void launch_ycocg_to_rgb(const int16_t* y, const int16_t* co, const int16_t* cg, uint64_t n_pixels, uint8_t* d_rgb, hipStream_t st) {
    if (!n_pixels) return;
    hipLaunchKernelGGL(ycocg_to_rgb_kernel, dim3(grid_for(n_pixels)), dim3(256), 0, st, y, co, cg,
                       (unsigned long long)n_pixels, d_rgb);
}

// ---- pad / strip (src/pipeline.rs:77-114, 603-611) ---------------------------------------

__global__ __launch_bounds__(256) void pad_channel_kernel(const int16_t* __restrict__ ch, ChunkDims d, int32_t* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < d.padded;
         i += (unsigned long long)gridDim.x * 256) {
        const unsigned x = (unsigned)(i % d.pw), yy = (unsigned)((i / d.pw) % d.ph), t = (unsigned)(i / ((unsigned long long)d.pw * d.ph));
        const unsigned sx = min(x, d.w - 1), sy = min(yy, d.h - 1), stt = min(t, d.f - 1);
        out[i] = ch[((unsigned long long)stt * d.h + sy) * d.w + sx];
    }
}
__global__ __launch_bounds__(256) void strip_channel_kernel(const int32_t* __restrict__ in, ChunkDims d, int16_t* __restrict__ ch) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < d.n_pixels;
         i += (unsigned long long)gridDim.x * 256) {
        const unsigned x = (unsigned)(i % d.w), yy = (unsigned)((i / d.w) % d.h), t = (unsigned)(i / ((unsigned long long)d.w * d.h));
        ch[i] = (int16_t)in[((unsigned long long)t * d.ph + yy) * d.pw + x];
    }
}
void launch_pad_channel(const int16_t* ch, const ChunkDims& d, int32_t* out, hipStream_t st) {
    if (!d.padded) return;
    hipLaunchKernelGGL(pad_channel_kernel, dim3(grid_for(d.padded)), dim3(256), 0, st, ch, d, out);
}
void launch_strip_channel(const int32_t* in, const ChunkDims& d, int16_t* ch, hipStream_t st) {
    if (!d.n_pixels) return;
    hipLaunchKernelGGL(strip_channel_kernel, dim3(grid_for(d.n_pixels)), dim3(256), 0, st, in, d, ch);
}

// ---- quantisers ----------------------------------------------------------------------------

__device__ __forceinline__ int wabs(int v) { return v < 0 ? (int)(0u - (unsigned)v) : v; }

__global__ __launch_bounds__(256) void quantize_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                       unsigned long long n, int step, int dead_zone) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        const int v = in[i];  // src/quant.rs:89-97 (true division, truncating)
        int q = 0;
        if (!(wabs(v) < dead_zone)) {
            const int hdz = dead_zone / 2;
            const int adj = v >= 0 ? (int)((unsigned)v - (unsigned)hdz) : (int)((unsigned)v + (unsigned)hdz);
            q = (step == -1) ? (int)(0u - (unsigned)adj) : adj / step;
        }
        out[i] = q;
    }
}

__global__ __launch_bounds__(256) void fast_quantize_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                            unsigned long long n, unsigned long long reciprocal,
                                                            unsigned shift, int dead_zone) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        const int v = in[i];  // src/quant.rs:243-264
        const int a = wabs(v);
        int q = 0;
        if (!(a < dead_zone)) {
            const unsigned adj = (unsigned)a - (unsigned)(dead_zone >> 1);
            const unsigned long long prod = (unsigned long long)adj * reciprocal;  // wraps like u64
            const int qa = (int)(unsigned)(shift >= 64 ? 0ull : (prod >> shift));
            q = v < 0 ? (int)(0u - (unsigned)qa) : qa;
        }
        out[i] = q;
    }
}

__global__ __launch_bounds__(256) void dequantize_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                         unsigned long long n, int step) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256)
        out[i] = (int)((unsigned)in[i] * (unsigned)step);  // src/quant.rs:104-110
}

__global__ __launch_bounds__(256) void to_symbols_kernel(const int32_t* __restrict__ in, uint8_t* __restrict__ out,
                                                         unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        const int c = in[i];  // src/quant.rs:555-560
        unsigned s = 0u;
        if (c > 0) s = (unsigned)c * 2u - 1u;
        else if (c < 0) s = (0u - (unsigned)c) * 2u;
        out[i] = (uint8_t)s;
    }
}

__global__ __launch_bounds__(256) void from_symbols_kernel(const uint8_t* __restrict__ in, int32_t* __restrict__ out,
                                                           unsigned long long n) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        const int s = in[i];  // src/quant.rs:580-588
        out[i] = (s == 0) ? 0 : ((s & 1) ? (s + 1) / 2 : -(s / 2));
    }
}

// build_histogram (src/quant.rs:594-600).  16 symbols per thread per step; the dominant zero symbol is
// counted in a register (byte-wise zero test), the others go to 8 LDS replicas of the bins.
__global__ __launch_bounds__(256) void histogram_kernel(const uint8_t* __restrict__ sym, unsigned long long n,
                                                        uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[8 * 256];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8 * 256; i += 256) lh[i] = 0u;
    __syncthreads();
    uint32_t* my = lh + (tid & 7) * 256;
    unsigned long long zeros = 0ull;
    const unsigned long long nvec = (((uintptr_t)sym & 15u) == 0u) ? n / 16 : 0ull;
    for (unsigned long long v = (unsigned long long)blockIdx.x * 256 + tid; v < nvec; v += (unsigned long long)gridDim.x * 256) {
        const uint4 q = ((const uint4*)sym)[v];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t s = (w[k] >> (8 * b)) & 0xFFu;
                if (s) atomicAdd(&my[s], 1u); else ++zeros;
            }
        }
    }
    for (unsigned long long i = nvec * 16 + (unsigned long long)blockIdx.x * 256 + tid; i < n; i += (unsigned long long)gridDim.x * 256) {
        const uint32_t s = sym[i];
        if (s) atomicAdd(&my[s], 1u); else ++zeros;
    }
    if (zeros) atomicAdd(&lh[0], (uint32_t)zeros);
    __syncthreads();
    uint32_t cnt = 0u;
#pragma unroll
    for (int r = 0; r < 8; ++r) cnt += lh[r * 256 + tid];
    if (cnt) atomicAdd(&hist[tid], cnt);
}

__global__ __launch_bounds__(256) void sq_diff_sum_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                          unsigned long long n, unsigned long long* __restrict__ sum) {
    unsigned long long acc = 0ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * 256) {
        const int dd = (int)a[i] - (int)b[i];
        acc += (unsigned long long)(dd * dd);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);
}

// ---- AnalyticalRDO::estimate_variance (src/quant.rs:414-435) ------------------------------------------------
// The i64 sum is exact in any order.  The sum of squared deviations is an f64 fold in element order in the
// reference, and f64 addition is not associative, so it is reproduced as such: all threads compute the terms
// (x - mean)^2 of a 4096-element tile (each term rounds the same way wherever it is computed), then ONE lane adds
// them in order.  8 cycles per element on a path that runs once per sub-band, not per pixel.
__global__ __launch_bounds__(256) void sum_i32_kernel(const int32_t* __restrict__ x, unsigned long long n,
                                                      unsigned long long* __restrict__ sum) {
    long long acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256)
        acc += (long long)x[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, (unsigned long long)acc);   // two's complement: wraps to the signed sum
}
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void ordered_sqdev_sum_kernel(const int32_t* __restrict__ x, unsigned long long n, double mean,
                                                                double* __restrict__ out) {
    constexpr int T = 4096;
    __shared__ double term[T];
    double acc = 0.0;
    for (unsigned long long base = 0; base < n; base += T) {
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += 256) {
            const unsigned long long k = base + (unsigned long long)i;
            double sq = 0.0;
            if (k < n) { const double diff = (double)x[k] - mean; sq = diff * diff; }
            term[i] = sq;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int m = (n - base) < (unsigned long long)T ? (int)(n - base) : T;
            for (int i = 0; i < m; ++i) acc = acc + term[i];
        }
    }
    if (threadIdx.x == 0) *out = acc;
}
// ---- ssim (src/ssim.rs:18-115): one thread per 8x8 block.  All block sums are sums of multiples of 1/4096 below
// 2^22, hence exact in f64 in any order; the scalar tail uses fma exactly where the reference uses mul_add; the
// mean over the blocks is an f64 fold in raster order, done by one lane (ordered_sum_f64_kernel).
__global__ __launch_bounds__(256) void ssim_blocks_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                          unsigned long long width, unsigned long long bw, unsigned long long nblocks,
                                                          double* __restrict__ out) {
    const unsigned long long blk = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (blk >= nblocks) return;
    const unsigned long long by = blk / bw, bx = blk % bw;
    const uint8_t* pa = a + (by * 8) * width + bx * 8;
    const uint8_t* pb = b + (by * 8) * width + bx * 8;
    double sa = 0.0, sb = 0.0;
    for (int dy = 0; dy < 8; ++dy)
        for (int dx = 0; dx < 8; ++dx) { sa += (double)pa[dy * width + dx]; sb += (double)pb[dy * width + dx]; }
    const double n = 64.0;
    const double mu_a = sa / n, mu_b = sb / n;
    double va = 0.0, vb = 0.0, vab = 0.0;
    for (int dy = 0; dy < 8; ++dy)
        for (int dx = 0; dx < 8; ++dx) {
            const double da = (double)pa[dy * width + dx] - mu_a, db = (double)pb[dy * width + dx] - mu_b;
            va += da * da; vb += db * db; vab += da * db;
        }
    const double denom = n - 1.0;
    va /= denom; vb /= denom; vab /= denom;
    const double C1 = 6.5025, C2 = 58.5225;
    const double numerator = fma(2.0 * mu_a, mu_b, C1) * fma(2.0, vab, C2);
    const double denominator = (fma(mu_a, mu_a, mu_b * mu_b) + C1) * (va + vb + C2);
    out[blk] = numerator / denominator;
}
__global__ __launch_bounds__(256) void ordered_sum_f64_kernel(const double* __restrict__ v, unsigned long long n, double* __restrict__ out) {
    constexpr int T = 4096;
    __shared__ double term[T];
    double acc = 0.0;
    for (unsigned long long base = 0; base < n; base += T) {
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += 256) term[i] = (base + (unsigned long long)i < n) ? v[base + i] : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int m = (n - base) < (unsigned long long)T ? (int)(n - base) : T;
            for (int i = 0; i < m; ++i) acc = acc + term[i];
        }
    }
    if (threadIdx.x == 0) *out = acc;
}
// downsample_2x (src/ssim.rs:181-200): truncating mean of each 2x2 cell
__global__ __launch_bounds__(256) void downsample2_kernel(const uint8_t* __restrict__ in, unsigned long long width, unsigned long long nw,
                                                          unsigned long long n_out, uint8_t* __restrict__ out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const unsigned long long y = i / nw, x = i % nw;
    const uint8_t* p = in + (2 * y) * width + 2 * x;
    out[i] = (uint8_t)(((unsigned)p[0] + p[1] + p[width] + p[width + 1]) / 4u);
}
#pragma clang fp contract(fast)
void launch_ssim_blocks(const uint8_t* d_a, const uint8_t* d_b, uint64_t width, uint64_t bw, uint64_t nblocks, double* d_out, hipStream_t st) {
    if (!nblocks) return;
    hipLaunchKernelGGL(ssim_blocks_kernel, dim3(grid_for(nblocks)), dim3(256), 0, st, d_a, d_b, (unsigned long long)width,
                       (unsigned long long)bw, (unsigned long long)nblocks, d_out);
}
void launch_ordered_sum_f64(const double* d_v, uint64_t n, double* d_out, hipStream_t st) {
    hipLaunchKernelGGL(ordered_sum_f64_kernel, dim3(1), dim3(256), 0, st, d_v, (unsigned long long)n, d_out);
}
void launch_downsample2(const uint8_t* d_in, uint64_t width, uint64_t height, uint8_t* d_out, hipStream_t st) {
    const uint64_t nw = width / 2, nh = height / 2;
    if (!(nw * nh)) return;
    hipLaunchKernelGGL(downsample2_kernel, dim3(grid_for(nw * nh)), dim3(256), 0, st, d_in, (unsigned long long)width,
                       (unsigned long long)nw, (unsigned long long)(nw * nh), d_out);
}
void launch_sum_i32(const int32_t* d_x, uint64_t n, unsigned long long* d_sum /*zeroed*/, hipStream_t st) {
    if (!n) return;
    unsigned g = grid_for(n);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(sum_i32_kernel, dim3(g), dim3(256), 0, st, d_x, (unsigned long long)n, d_sum);
}
void launch_ordered_sqdev_sum(const int32_t* d_x, uint64_t n, double mean, double* d_out, hipStream_t st) {
    hipLaunchKernelGGL(ordered_sqdev_sum_kernel, dim3(1), dim3(256), 0, st, d_x, (unsigned long long)n, mean, d_out);
}

void launch_quantize(const int32_t* in, int32_t* out, uint64_t n, int32_t step, int32_t dead_zone, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(quantize_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, out, (unsigned long long)n, step, dead_zone);
}
void launch_fast_quantize(const int32_t* in, int32_t* out, uint64_t n, uint64_t reciprocal, uint32_t shift,
                          int32_t dead_zone, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(fast_quantize_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, out, (unsigned long long)n,
                       (unsigned long long)reciprocal, shift, dead_zone);
}
void launch_dequantize(const int32_t* in, int32_t* out, uint64_t n, int32_t step, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(dequantize_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, out, (unsigned long long)n, step);
}
void launch_to_symbols(const int32_t* in, uint8_t* out, uint64_t n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(to_symbols_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, out, (unsigned long long)n);
}
void launch_from_symbols(const uint8_t* in, int32_t* out, uint64_t n, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(from_symbols_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, out, (unsigned long long)n);
}
void launch_histogram(const uint8_t* sym, uint64_t n, uint32_t* hist, hipStream_t st) {
    if (!n) return;
    unsigned g = grid_for(n);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(histogram_kernel, dim3(g), dim3(256), 0, st, sym, (unsigned long long)n, hist);
}
void launch_sq_diff_sum(const uint8_t* a, const uint8_t* b, uint64_t n, unsigned long long* d_sum, hipStream_t st) {
    if (!n) return;
    unsigned g = grid_for(n);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(sq_diff_sum_kernel, dim3(g), dim3(256), 0, st, a, b, (unsigned long long)n, d_sum);
}

// ---- .alc assembly on the device -------------------------------------------------------------

__device__ __forceinline__ void put_u32le(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

// EncodedChunk::to_bytes header part, reference src/pipeline.rs:200-221.  One block per chunk.
__global__ __launch_bounds__(256) void write_headers_kernel(uint8_t* __restrict__ alc, unsigned long long alc_stride,
                                                            ChunkDims d, int wavelet, int step,
                                                            const uint32_t* __restrict__ hist,
                                                            const RansResult* __restrict__ res,
                                                            unsigned long long* __restrict__ sizes) {
    const int chunk = blockIdx.x, tid = threadIdx.x;
    uint8_t* p = alc + (size_t)chunk * alc_stride;
    if (tid == 0) {
        p[0] = 'A'; p[1] = 'L'; p[2] = 'C'; p[3] = 'C';
        p[4] = 1;
        p[5] = (uint8_t)wavelet;
        put_u32le(p + 6, d.w);
        put_u32le(p + 10, d.h);
        put_u32le(p + 14, d.f);
        unsigned long long total = kAlcHeaderBytes;
        for (int c = 0; c < 3; ++c) {
            uint8_t* h = p + kFixedHeaderBytes + c * kChannelHeaderBytes;
            // a chain that overflowed its region produced no stream: the host sees the flag and encodes again
            const unsigned long long len = (res[chunk * 3 + c].flags & kRansOverflow) ? 0ull : res[chunk * 3 + c].len;
            put_u32le(h, (uint32_t)len);            // compressed_len as u32 (src/pipeline.rs:489)
            put_u32le(h + 4, (uint32_t)step);
            put_u32le(h + 8, (uint32_t)step);       // dead_zone = step (Quantizer::new)
            put_u32le(h + 12, (uint32_t)d.padded);  // num_symbols as u32 (:492)
            total += len;
        }
        sizes[chunk] = total;
    }
    for (int c = 0; c < 3; ++c)
        put_u32le(p + kFixedHeaderBytes + c * kChannelHeaderBytes + 16 + 4 * tid, hist[((size_t)chunk * 3 + c) * 256 + tid]);
}

// Moves the three streams of a chunk (each sits at the tail of its cap-sized region inside the chunk's own
// .alc buffer) to directly behind the header, IN PLACE.  Every byte moves towards lower addresses, so a
// front-to-back copy is safe as long as a block of the source has been read completely before anything at
// or beyond its destination is written: one workgroup per chunk walks the stream in 64 KB blocks,
// load (all threads) -> barrier -> store.  The destination of block k ends where the source of block k
// began or earlier, so it can only overlap sources that have already been read.
constexpr int kCompactThreads = 1024, kCompactVecs = 4;
__global__ __launch_bounds__(kCompactThreads) void compact_streams_kernel(uint8_t* __restrict__ alc, unsigned long long alc_stride,
                                                                         unsigned long long head_bytes, unsigned long long cap0,
                                                                         unsigned long long cap1, unsigned long long cap2,
                                                                         const RansResult* __restrict__ res) {
    const int chunk = blockIdx.x;
    const unsigned tid = threadIdx.x;
    uint8_t* const base = alc + (size_t)chunk * alc_stride;
    uint8_t* dst = base + kAlcHeaderBytes;
    for (int c = 0; c < 3; ++c) {
        // never touch memory for a chain that overflowed its region (its length exceeds the capacity)
        const RansResult rr = res[chunk * 3 + c];
        const unsigned long long cap = c == 0 ? cap0 : (c == 1 ? cap1 : cap2);
        const unsigned long long region_end = c == 0 ? cap0 : (c == 1 ? cap0 + cap1 : cap0 + cap1 + cap2);
        const unsigned long long len = ((rr.flags & kRansOverflow) || rr.len > cap) ? 0ull : rr.len;
        const uint8_t* src = base + head_bytes + (size_t)region_end - len;
        // bytes up to the first 16-byte boundary of dst, then 16 B stores fed by (possibly unaligned) 4 B loads
        unsigned long long head = (16u - (unsigned)((uintptr_t)dst & 15u)) & 15u;
        if (head > len) head = len;
        const unsigned long long nvec = (len - head) / 16;
        const unsigned long long tail0 = head + nvec * 16;
        // head and tail bytes: read now, written after the first barrier (a later block may overwrite their source)
        uint8_t hb = 0, tb = 0;
        if (tid < head) hb = src[tid];
        if (tail0 + tid < len) tb = src[tail0 + tid];
        const unsigned mis = (unsigned)((uintptr_t)(src + head) & 3u);
        const unsigned sh = mis * 8;
        const unsigned long long per_block = (unsigned long long)kCompactThreads * kCompactVecs;
        bool first = true;
        for (unsigned long long v0 = 0; v0 < nvec || first; v0 += per_block) {
            uint4 w[kCompactVecs];
#pragma unroll
            for (int u = 0; u < kCompactVecs; ++u) {
                const unsigned long long v = v0 + (unsigned long long)u * kCompactThreads + tid;
                if (v < nvec) {
                    const uint32_t* s4 = (const uint32_t*)(src + head + v * 16 - mis);
                    if (mis == 0) {
                        w[u] = make_uint4(s4[0], s4[1], s4[2], s4[3]);
                    } else {
                        const uint32_t t0 = s4[0], t1 = s4[1], t2 = s4[2], t3 = s4[3], t4 = s4[4];
                        w[u] = make_uint4((t0 >> sh) | (t1 << (32 - sh)), (t1 >> sh) | (t2 << (32 - sh)),
                                          (t2 >> sh) | (t3 << (32 - sh)), (t3 >> sh) | (t4 << (32 - sh)));
                    }
                }
            }
            __syncthreads();
            if (first) {
                if (tid < head) dst[tid] = hb;
                first = false;
            }
#pragma unroll
            for (int u = 0; u < kCompactVecs; ++u) {
                const unsigned long long v = v0 + (unsigned long long)u * kCompactThreads + tid;
                if (v < nvec) *(uint4*)(dst + head + v * 16) = w[u];
            }
        }
        if (tail0 + tid < len) dst[tail0 + tid] = tb;
        __syncthreads();   // the next stream's destination may cover this stream's source
        dst += len;
    }
}

__global__ __launch_bounds__(256) void split4_kernel(const uint8_t* __restrict__ in, unsigned long long n,
                                                     uint8_t* __restrict__ out, unsigned long long stride) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[(i & 3ull) * stride + (i >> 2)] = in[i];
}
struct Count4 { unsigned long long c[4]; };
__global__ __launch_bounds__(256) void merge4_kernel(const uint8_t* __restrict__ in, unsigned long long stride, Count4 have,
                                                     Count4 cnt, uint8_t* __restrict__ out, unsigned long long n_out) {
    const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (k >= have.c[j]) return;
    unsigned long long pos = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pos += cnt.c[i] < k ? cnt.c[i] : k;
        if (i < j && cnt.c[i] > k) ++pos;
    }
    if (pos < n_out) out[pos] = in[(size_t)j * stride + k];
}
void launch_split4(const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t stride, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(split4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, (unsigned long long)n, d_out,
                       (unsigned long long)stride);
}
void launch_merge4(const uint8_t* d_in, uint64_t stride, const uint64_t have[4], const uint64_t count[4], uint8_t* d_out,
                   uint64_t n_out, hipStream_t st) {
    Count4 h{{have[0], have[1], have[2], have[3]}}, c{{count[0], count[1], count[2], count[3]}};
    uint64_t mx = 0;
    for (int i = 0; i < 4; ++i) mx = have[i] > mx ? have[i] : mx;
    if (!mx || !n_out) return;
    hipLaunchKernelGGL(merge4_kernel, dim3((unsigned)((mx + 255) / 256), 4), dim3(256), 0, st, d_in, (unsigned long long)stride, h, c,
                       d_out, (unsigned long long)n_out);
}

void launch_write_headers(uint8_t* d_alc, uint64_t alc_stride, const ChunkDims& d, int wavelet, int32_t step,
                          const uint32_t* d_hist, const RansResult* d_results, unsigned long long* d_sizes,
                          int n_chunks, hipStream_t st) {
    if (n_chunks <= 0) return;
    hipLaunchKernelGGL(write_headers_kernel, dim3(n_chunks), dim3(256), 0, st, d_alc, (unsigned long long)alc_stride, d,
                       wavelet, step, d_hist, d_results, d_sizes);
}
void launch_compact_streams(uint8_t* d_alc, uint64_t alc_stride, uint64_t head, const uint64_t cap[3],
                            const RansResult* d_results, int n_chunks, hipStream_t st) {
    if (n_chunks <= 0) return;
    hipLaunchKernelGGL(compact_streams_kernel, dim3(n_chunks), dim3(kCompactThreads), 0, st, d_alc,
                       (unsigned long long)alc_stride, (unsigned long long)head, (unsigned long long)cap[0],
                       (unsigned long long)cap[1], (unsigned long long)cap[2], d_results);
}

}  // namespace alice
