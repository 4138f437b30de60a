#define _GNU_SOURCE
/*
 * alice_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE)
 *
 * Scalar C restatement of the reference hot path; see alice_oracle.h for the
 * pinning status ("whole-bitstream parity unpinned") and the rules on who may
 * call this.  Loop structure deliberately follows the reference (gathered
 * columns / temporal vectors, a temporary per 1-D call, true division in the
 * quantizer and in rANS) so that the timed cpu_baseline is an honest stand-in
 * for the single-threaded Rust code.
 *
 * Every function cites the reference file:line it restates.
 */
#include "alice_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- Rust integer semantics helpers ------------------------------------- */

static inline int32_t wrap_add_i32(int32_t a, int32_t b) {
    return (int32_t)((uint32_t)a + (uint32_t)b);
}
static inline int32_t wrap_sub_i32(int32_t a, int32_t b) {
    return (int32_t)((uint32_t)a - (uint32_t)b);
}
static inline int32_t wrap_mul_i32(int32_t a, int32_t b) {
    return (int32_t)((uint32_t)a * (uint32_t)b);
}
static inline int32_t wrap_neg_i32(int32_t a) { return (int32_t)(0u - (uint32_t)a); }
/* i32::abs in release: abs(i32::MIN) == i32::MIN */
static inline int32_t wrap_abs_i32(int32_t a) { return a < 0 ? wrap_neg_i32(a) : a; }
static inline int16_t wrap_add_i16(int16_t a, int16_t b) {
    return (int16_t)(uint16_t)((uint16_t)a + (uint16_t)b);
}
static inline int16_t wrap_sub_i16(int16_t a, int16_t b) {
    return (int16_t)(uint16_t)((uint16_t)a - (uint16_t)b);
}
/* arithmetic (flooring) shift right on i64 */
static inline int64_t asr64(int64_t v, unsigned s) {
    return v >= 0 ? (v >> s) : -((-(v + 1)) >> s) - 1;
}
static inline int16_t asr16_1(int16_t v) { return (int16_t)(v >= 0 ? v >> 1 : -((-(v + 1)) >> 1) - 1); }
/* i32 / i32 truncating toward zero, wrapping on MIN / -1 like release Rust would
 * panic there -- never reached with step >= 1 */
static inline int32_t div_trunc_i32(int32_t a, int32_t b) { return a / b; }

void ao_free(void *p) { free(p); }

/* ---- colour -------------------------------------------------------------- */

/* src/color.rs:199-235 rgb_bytes_to_ycocg_r */
int ao_rgb_bytes_to_ycocg_r(const uint8_t *rgb, size_t rgb_len, int16_t *y, int16_t *co,
                            int16_t *cg, size_t out_len) {
    if (rgb_len % 3 != 0) return AO_ERR_INVALID_BUFFER_SIZE; /* :205-210 */
    size_t n = rgb_len / 3;
    if (out_len < n) return AO_ERR_INVALID_BUFFER_SIZE;      /* :212-218 */
    for (size_t i = 0; i < n; ++i) {                          /* :220-233 */
        int16_t r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
        int16_t co_v = wrap_sub_i16(r, b);
        int16_t t = wrap_add_i16(b, asr16_1(co_v));
        int16_t cg_v = wrap_sub_i16(g, t);
        int16_t y_v = wrap_add_i16(t, asr16_1(cg_v));
        y[i] = y_v;
        co[i] = co_v;
        cg[i] = cg_v;
    }
    return AO_OK;
}

static inline uint8_t clamp_u8_i16(int16_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* src/color.rs:245-276 ycocg_r_to_rgb_bytes (i16 arithmetic wraps in release) */
int ao_ycocg_r_to_rgb_bytes(const int16_t *y, const int16_t *co, const int16_t *cg, size_t n,
                            uint8_t *rgb, size_t rgb_len) {
    if (rgb_len < n * 3) return AO_ERR_INVALID_BUFFER_SIZE; /* :258-263 */
    for (size_t i = 0; i < n; ++i) {                         /* :265-274 */
        int16_t t = wrap_sub_i16(y[i], asr16_1(cg[i]));
        int16_t g = wrap_add_i16(cg[i], t);
        int16_t b = wrap_sub_i16(t, asr16_1(co[i]));
        int16_t r = wrap_add_i16(co[i], b);
        rgb[i * 3] = clamp_u8_i16(r);
        rgb[i * 3 + 1] = clamp_u8_i16(g);
        rgb[i * 3 + 2] = clamp_u8_i16(b);
    }
    return AO_OK;
}

/* ---- wavelet ------------------------------------------------------------- */

typedef struct {
    int32_t coeff;
    int predict;
} lifting_step;

/* src/wavelet.rs:66-127 coefficient tables (scale 2^12) */
static const lifting_step STEPS_CDF97[4] = {{-6497, 1}, {-217, 0}, {3616, 1}, {1817, 0}};
static const lifting_step STEPS_HAAR[2] = {{-4096, 1}, {2048, 0}};
static const lifting_step STEPS_CDF53[2] = {{-4096, 1}, {1024, 0}};

static const lifting_step *steps_for(int type, int *count) {
    switch (type) {
    case AO_WAVELET_CDF97: *count = 4; return STEPS_CDF97;
    case AO_WAVELET_HAAR: *count = 2; return STEPS_HAAR;
    default: *count = 2; return STEPS_CDF53;
    }
}

/* src/wavelet.rs:193-195 / 213-215: avg wraps in i32, product+round in i64,
 * arithmetic shift, truncate to i32, wrapping += */
static inline int32_t lift_delta(int32_t a, int32_t b, int32_t coeff) {
    int32_t avg = wrap_add_i32(a, b);
    return (int32_t)asr64((int64_t)avg * (int64_t)coeff + 4096, 13);
}

/* src/wavelet.rs:180-197 lift_predict */
static void lift_predict(int32_t *s, size_t n, int32_t coeff) {
    size_t half = n / 2;
    for (size_t i = 0; i < half; ++i) {
        int32_t el = s[i * 2];
        int32_t er = (i * 2 + 2 < n) ? s[i * 2 + 2] : s[i * 2]; /* mirror */
        s[i * 2 + 1] = wrap_add_i32(s[i * 2 + 1], lift_delta(el, er, coeff));
    }
}

/* src/wavelet.rs:201-217 lift_update */
static void lift_update(int32_t *s, size_t n, int32_t coeff) {
    size_t half = n / 2;
    for (size_t i = 0; i < half; ++i) {
        int32_t ol = (i > 0) ? s[i * 2 - 1] : s[1]; /* mirror */
        int32_t orr = s[i * 2 + 1];
        s[i * 2] = wrap_add_i32(s[i * 2], lift_delta(ol, orr, coeff));
    }
}

/* src/wavelet.rs:220-233 deinterleave (temp zero-filled: an odd tail is dropped) */
static void deinterleave(int32_t *s, size_t n) {
    size_t half = n / 2;
    int32_t *tmp = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    for (size_t i = 0; i < half; ++i) {
        tmp[i] = s[i * 2];
        tmp[half + i] = s[i * 2 + 1];
    }
    memcpy(s, tmp, n * sizeof(int32_t));
    free(tmp);
}

/* src/wavelet.rs:236-248 interleave */
static void interleave(int32_t *s, size_t n) {
    size_t half = n / 2;
    int32_t *tmp = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    for (size_t i = 0; i < half; ++i) {
        tmp[i * 2] = s[i];
        tmp[i * 2 + 1] = s[half + i];
    }
    memcpy(s, tmp, n * sizeof(int32_t));
    free(tmp);
}

/* src/wavelet.rs:133-152 Wavelet1D::forward */
void ao_wavelet1d_forward(int type, int32_t *signal, size_t n) {
    if (n < 2) return;
    int cnt;
    const lifting_step *st = steps_for(type, &cnt);
    for (int k = 0; k < cnt; ++k) {
        if (st[k].predict) lift_predict(signal, n, st[k].coeff);
        else lift_update(signal, n, st[k].coeff);
    }
    deinterleave(signal, n);
}

/* src/wavelet.rs:157-176 Wavelet1D::inverse (steps reversed with negated coeff) */
void ao_wavelet1d_inverse(int type, int32_t *signal, size_t n) {
    if (n < 2) return;
    int cnt;
    const lifting_step *st = steps_for(type, &cnt);
    interleave(signal, n);
    for (int k = cnt - 1; k >= 0; --k) {
        if (st[k].predict) lift_predict(signal, n, -st[k].coeff);
        else lift_update(signal, n, -st[k].coeff);
    }
}

/* src/wavelet.rs:292-316 Wavelet2D::forward: rows then gathered columns */
void ao_wavelet2d_forward(int type, int32_t *image, size_t width, size_t height) {
    for (size_t y = 0; y < height; ++y) ao_wavelet1d_forward(type, image + y * width, width);
    int32_t *col = (int32_t *)malloc((height ? height : 1) * sizeof(int32_t));
    for (size_t x = 0; x < width; ++x) {
        for (size_t y = 0; y < height; ++y) col[y] = image[y * width + x];
        ao_wavelet1d_forward(type, col, height);
        for (size_t y = 0; y < height; ++y) image[y * width + x] = col[y];
    }
    free(col);
}

/* src/wavelet.rs:319-340 Wavelet2D::inverse: columns then rows */
void ao_wavelet2d_inverse(int type, int32_t *image, size_t width, size_t height) {
    int32_t *col = (int32_t *)malloc((height ? height : 1) * sizeof(int32_t));
    for (size_t x = 0; x < width; ++x) {
        for (size_t y = 0; y < height; ++y) col[y] = image[y * width + x];
        ao_wavelet1d_inverse(type, col, height);
        for (size_t y = 0; y < height; ++y) image[y * width + x] = col[y];
    }
    free(col);
    for (size_t y = 0; y < height; ++y) ao_wavelet1d_inverse(type, image + y * width, width);
}

/* src/wavelet.rs:392-438 Wavelet3D::forward */
void ao_wavelet3d_forward(int type, int32_t *volume, size_t width, size_t height, size_t depth) {
    size_t frame_size = width * height;
    for (size_t t = 0; t < depth; ++t) ao_wavelet2d_forward(type, volume + t * frame_size, width, height);
    int32_t *temporal = (int32_t *)malloc((depth ? depth : 1) * sizeof(int32_t));
    for (size_t y = 0; y < height; ++y) {
        for (size_t x = 0; x < width; ++x) {
            for (size_t t = 0; t < depth; ++t) temporal[t] = volume[t * frame_size + y * width + x];
            ao_wavelet1d_forward(type, temporal, depth);
            for (size_t t = 0; t < depth; ++t) volume[t * frame_size + y * width + x] = temporal[t];
        }
    }
    free(temporal);
}

/* src/wavelet.rs:441-484 Wavelet3D::inverse: temporal first, then per frame cols, rows */
void ao_wavelet3d_inverse(int type, int32_t *volume, size_t width, size_t height, size_t depth) {
    size_t frame_size = width * height;
    int32_t *temporal = (int32_t *)malloc((depth ? depth : 1) * sizeof(int32_t));
    for (size_t y = 0; y < height; ++y) {
        for (size_t x = 0; x < width; ++x) {
            for (size_t t = 0; t < depth; ++t) temporal[t] = volume[t * frame_size + y * width + x];
            ao_wavelet1d_inverse(type, temporal, depth);
            for (size_t t = 0; t < depth; ++t) volume[t * frame_size + y * width + x] = temporal[t];
        }
    }
    free(temporal);
    for (size_t t = 0; t < depth; ++t) ao_wavelet2d_inverse(type, volume + t * frame_size, width, height);
}

/* ---- quantizer ------------------------------------------------------------ */

/* src/quant.rs:89-97 Quantizer::quantize */
int32_t ao_quantize(int32_t step, int32_t dead_zone, int32_t value) {
    if (wrap_abs_i32(value) < dead_zone) return 0;
    if (value >= 0) return div_trunc_i32(wrap_sub_i32(value, dead_zone / 2), step);
    return div_trunc_i32(wrap_add_i32(value, dead_zone / 2), step);
}

/* src/quant.rs:104-110 Quantizer::dequantize */
int32_t ao_dequantize(int32_t step, int32_t qvalue) {
    return qvalue == 0 ? 0 : wrap_mul_i32(qvalue, step);
}

/* src/quant.rs:117-128 */
void ao_quantize_buffer(int32_t step, int32_t dead_zone, const int32_t *in, int32_t *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = ao_quantize(step, dead_zone, in[i]);
}
/* src/quant.rs:135-146 */
void ao_dequantize_buffer(int32_t step, const int32_t *in, int32_t *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = ao_dequantize(step, in[i]);
}

/* src/quant.rs:190-217 FastQuantizer::new */
int ao_fast_quantizer_new(int32_t step, ao_fast_quantizer *q) {
    if (step <= 0) return AO_ERR_INVALID_QUANT_STEP;
    uint32_t step_u = (uint32_t)step;
    uint32_t extra_bits = 32u - (uint32_t)__builtin_clz(step_u);
    uint32_t shift = 32u + extra_bits;
    unsigned __int128 power = (unsigned __int128)1 << shift;
    unsigned __int128 rec = (power + step_u - 1) / step_u; /* div_ceil */
    q->reciprocal = (uint64_t)rec;
    q->shift = shift;
    q->step = step;
    q->dead_zone = step;
    return AO_OK;
}
/* src/quant.rs:224-228 */
int ao_fast_quantizer_with_dead_zone(int32_t step, int32_t dead_zone, ao_fast_quantizer *q) {
    int rc = ao_fast_quantizer_new(step, q);
    if (rc) return rc;
    q->dead_zone = dead_zone;
    return AO_OK;
}
/* src/quant.rs:232-264 fast_div + quantize (u64 product wraps) */
int32_t ao_fast_quantize(const ao_fast_quantizer *q, int32_t value) {
    int32_t abs_val = wrap_abs_i32(value);
    if (abs_val < q->dead_zone) return 0;
    int32_t offset = q->dead_zone >> 1; /* arithmetic on i32 */
    uint32_t adjusted = (uint32_t)wrap_sub_i32(abs_val, offset);
    uint64_t product = (uint64_t)adjusted * q->reciprocal; /* wrapping u64 */
    int32_t q_abs = (int32_t)(uint32_t)(product >> q->shift);
    return value < 0 ? wrap_neg_i32(q_abs) : q_abs;
}
void ao_fast_quantize_buffer(const ao_fast_quantizer *q, const int32_t *in, int32_t *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = ao_fast_quantize(q, in[i]);
}

/* ---- symbols / histogram -------------------------------------------------- */

/* src/quant.rs:555-560 to_symbols: low 8 bits of (2c-1) / (-2c), wrapping */
void ao_to_symbols(const int32_t *coeffs, uint8_t *symbols, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        int32_t c = coeffs[i];
        if (c == 0) symbols[i] = 0;
        else if (c > 0) symbols[i] = (uint8_t)(uint32_t)wrap_sub_i32(wrap_mul_i32(c, 2), 1);
        else symbols[i] = (uint8_t)(uint32_t)wrap_mul_i32(wrap_neg_i32(c), 2);
    }
}
/* src/quant.rs:580-588 from_symbols */
void ao_from_symbols(const uint8_t *symbols, int32_t *coeffs, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        uint8_t s = symbols[i];
        if (s == 0) coeffs[i] = 0;
        else if (s % 2 == 1) coeffs[i] = ((int32_t)s + 1) / 2;
        else coeffs[i] = -((int32_t)s / 2);
    }
}
/* src/quant.rs:594-600 build_histogram */
void ao_build_histogram(const uint8_t *symbols, size_t n, uint32_t hist[256]) {
    memset(hist, 0, 256 * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) hist[symbols[i]] += 1;
}

/* ---- rANS ------------------------------------------------------------------ */

static int table_alloc(size_t n_symbols, ao_freq_table *t) {
    t->n_symbols = n_symbols;
    t->cum_freq = (uint16_t *)calloc(n_symbols ? n_symbols : 1, sizeof(uint16_t));
    t->freq = (uint16_t *)calloc(n_symbols ? n_symbols : 1, sizeof(uint16_t));
    if (!t->cum_freq || !t->freq) return AO_ERR_NOMEM;
    memset(t->cum_to_sym, 0, sizeof(t->cum_to_sym));
    return AO_OK;
}

void ao_freq_table_free(ao_freq_table *t) {
    free(t->cum_freq);
    free(t->freq);
    t->cum_freq = t->freq = NULL;
}

/* src/rans.rs:134-144 / 174-183 cum_to_sym fill: [cum, min(cum+freq, 4096)) */
static void fill_cum_to_sym(ao_freq_table *t) {
    for (size_t sym = 0; sym < t->n_symbols; ++sym) {
        size_t start = t->cum_freq[sym];
        size_t end = start + t->freq[sym];
        if (end > AO_PROB_SCALE) end = AO_PROB_SCALE;
        for (size_t s = start; s < end; ++s) t->cum_to_sym[s] = (uint8_t)sym;
    }
}

/* src/rans.rs:158-189 FrequencyTable::uniform (n_symbols == 0 panics in the reference) */
int ao_freq_table_uniform(size_t n_symbols, ao_freq_table *t) {
    if (n_symbols == 0) return AO_ERR_REFERENCE_DIVERGES;
    int rc = table_alloc(n_symbols, t);
    if (rc) return rc;
    uint16_t fps = (uint16_t)(AO_PROB_SCALE / n_symbols);
    uint16_t cum = 0;
    for (size_t i = 0; i < n_symbols; ++i) {
        t->cum_freq[i] = cum;
        t->freq[i] = fps;
        cum = (uint16_t)(cum + fps);
    }
    t->freq[n_symbols - 1] = (uint16_t)((uint16_t)AO_PROB_SCALE - t->cum_freq[n_symbols - 1]);
    fill_cum_to_sym(t);
    return AO_OK;
}

/* src/rans.rs:102-150 FrequencyTable::from_histogram */
int ao_freq_table_from_histogram(const uint32_t *hist, size_t n_symbols, ao_freq_table *t) {
    uint64_t total = 0;
    for (size_t i = 0; i < n_symbols; ++i) total += hist[i];
    if (total == 0) return ao_freq_table_uniform(n_symbols, t); /* :106-109 */
    int rc = table_alloc(n_symbols, t);
    if (rc) return rc;
    uint32_t cum = 0, normalized_total = 0;
    for (size_t i = 0; i < n_symbols; ++i) { /* :116-125 */
        uint32_t freq;
        if (hist[i] == 0) freq = 1;
        else {
            uint64_t f = ((uint64_t)hist[i] * AO_PROB_SCALE) / total;
            freq = (uint32_t)(f < 1 ? 1 : f);
        }
        normalized_total += freq;
        t->cum_freq[i] = (uint16_t)cum;
        t->freq[i] = (uint16_t)freq;
        cum += freq;
    }
    if (n_symbols && normalized_total != AO_PROB_SCALE) { /* :128-132, wrapping cast */
        int32_t diff = (int32_t)AO_PROB_SCALE - (int32_t)normalized_total;
        t->freq[n_symbols - 1] = (uint16_t)(uint32_t)((int32_t)t->freq[n_symbols - 1] + diff);
    }
    fill_cum_to_sym(t);
    return AO_OK;
}

typedef struct {
    uint32_t state;
    uint8_t *buf;
    size_t len, cap;
} rans_enc;

static int enc_push(rans_enc *e, uint8_t b) {
    if (e->len == e->cap) {
        size_t nc = e->cap ? e->cap * 2 : 64;
        uint8_t *nb = (uint8_t *)realloc(e->buf, nc);
        if (!nb) return AO_ERR_NOMEM;
        e->buf = nb;
        e->cap = nc;
    }
    e->buf[e->len++] = b;
    return AO_OK;
}

/* src/rans.rs:269-285 RansEncoder::encode */
static int enc_put(rans_enc *e, uint16_t cum_freq, uint16_t freq16) {
    uint32_t freq = freq16;
    if (freq == 0) return AO_ERR_REFERENCE_DIVERGES; /* reference: endless loop / div by zero */
    uint64_t x_max = (((uint64_t)(AO_RANS32_L >> AO_PROB_BITS)) << 8) * (uint64_t)freq;
    while ((uint64_t)e->state >= x_max) {
        int rc = enc_push(e, (uint8_t)(e->state & 0xFF));
        if (rc) return rc;
        e->state >>= 8;
    }
    uint32_t q = e->state / freq;
    uint32_t r = e->state % freq;
    e->state = (q << AO_PROB_BITS) + r + (uint32_t)cum_freq; /* u32 wrapping */
    return AO_OK;
}

/* src/rans.rs:298-308 RansEncoder::finish */
static int enc_finish(rans_enc *e, uint8_t **out, size_t *out_len) {
    for (int k = 0; k < 4; ++k) {
        int rc = enc_push(e, (uint8_t)((e->state >> (8 * k)) & 0xFF));
        if (rc) return rc;
    }
    for (size_t i = 0, j = e->len - 1; i < j; ++i, --j) {
        uint8_t tmp = e->buf[i];
        e->buf[i] = e->buf[j];
        e->buf[j] = tmp;
    }
    *out = e->buf;
    *out_len = e->len;
    return AO_OK;
}

/* src/rans.rs:288-294 encode_symbols (reverse order) + finish */
int ao_rans_encode(const uint8_t *symbols, size_t n, const ao_freq_table *t, uint8_t **out,
                   size_t *out_len) {
    rans_enc e = {AO_RANS32_L, NULL, 0, 0};
    for (size_t i = n; i-- > 0;) {
        uint8_t s = symbols[i];
        if ((size_t)s >= t->n_symbols) { free(e.buf); return AO_ERR_REFERENCE_DIVERGES; }
        int rc = enc_put(&e, t->cum_freq[s], t->freq[s]);
        if (rc) { free(e.buf); return rc; }
    }
    int rc = enc_finish(&e, out, out_len);
    if (rc) free(e.buf);
    return rc;
}

typedef struct {
    uint32_t state;
    const uint8_t *in;
    size_t len, pos;
} rans_dec;

/* src/rans.rs:330-347 RansDecoder::new / init_state */
static void dec_init(rans_dec *d, const uint8_t *in, size_t len) {
    d->state = 0;
    d->in = in;
    d->len = len;
    d->pos = 0;
    if (len >= 4) {
        d->state = ((uint32_t)in[0] << 24) | ((uint32_t)in[1] << 16) | ((uint32_t)in[2] << 8) | in[3];
        d->pos = 4;
    }
}

/* src/rans.rs:351-371 RansDecoder::decode */
static uint8_t dec_get(rans_dec *d, const ao_freq_table *t) {
    uint32_t slot = d->state & (AO_PROB_SCALE - 1);
    uint8_t sym = t->cum_to_sym[slot];
    uint64_t freq = t->freq[sym];
    d->state = (uint32_t)(freq * (uint64_t)(d->state >> AO_PROB_BITS) + (uint64_t)slot -
                          (uint64_t)t->cum_freq[sym]);
    while (d->state < AO_RANS32_L && d->pos < d->len) {
        d->state = (d->state << 8) | d->in[d->pos];
        d->pos += 1;
    }
    return sym;
}

/* src/rans.rs:375-381 decode_n */
void ao_rans_decode(const uint8_t *in, size_t in_len, size_t n, const ao_freq_table *t,
                    uint8_t *symbols) {
    rans_dec d;
    dec_init(&d, in, in_len);
    for (size_t i = 0; i < n; ++i) symbols[i] = dec_get(&d, t);
}

/* ---- the encoder / decoder as OBJECTS that live across calls (src/rans.rs:238-309, 321-389): repeated encode_symbols
 * and single-symbol encode calls continue one state and one output vector; decode / decode_n continue from the current
 * position.  The GPU path's stateful handles are checked against these. ---- */
struct ao_rans_encoder { rans_enc e; };
struct ao_rans_decoder { rans_dec d; uint8_t *own; };

ao_rans_encoder *ao_rans_encoder_new(void) { /* RansEncoder::new / with_capacity, :249-264 */
    ao_rans_encoder *h = (ao_rans_encoder *)calloc(1, sizeof(*h));
    if (h) h->e.state = AO_RANS32_L;
    return h;
}
void ao_rans_encoder_free(ao_rans_encoder *h) {
    if (h) { free(h->e.buf); free(h); }
}
int ao_rans_encoder_encode(ao_rans_encoder *h, uint16_t cum_freq, uint16_t freq) { /* encode(&RansSymbol), :269-285 */
    return enc_put(&h->e, cum_freq, freq);
}
int ao_rans_encoder_encode_symbols(ao_rans_encoder *h, const uint8_t *symbols, size_t n, const ao_freq_table *t) { /* :288-294 */
    for (size_t i = n; i-- > 0;) {
        uint8_t s = symbols[i];
        if ((size_t)s >= t->n_symbols) return AO_ERR_REFERENCE_DIVERGES; /* reference: index out of bounds */
        int rc = enc_put(&h->e, t->cum_freq[s], t->freq[s]);
        if (rc) return rc;
    }
    return AO_OK;
}
uint32_t ao_rans_encoder_state(const ao_rans_encoder *h) { return h->e.state; }
int ao_rans_encoder_finish(ao_rans_encoder *h, uint8_t **out, size_t *out_len) { /* finish(self), :298-308: consumes the encoder */
    int rc = enc_finish(&h->e, out, out_len);
    if (rc == AO_OK) h->e.buf = NULL;
    ao_rans_encoder_free(h);
    return rc;
}

ao_rans_decoder *ao_rans_decoder_new(const uint8_t *in, size_t len) { /* RansDecoder::new, :330-347 (the input is copied) */
    ao_rans_decoder *h = (ao_rans_decoder *)calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->own = (uint8_t *)malloc(len ? len : 1);
    if (!h->own) { free(h); return NULL; }
    if (len) memcpy(h->own, in, len);
    dec_init(&h->d, h->own, len);
    return h;
}
void ao_rans_decoder_free(ao_rans_decoder *h) {
    if (h) { free(h->own); free(h); }
}
void ao_rans_decoder_decode_n(ao_rans_decoder *h, size_t n, const ao_freq_table *t, uint8_t *symbols) { /* :351-381 */
    for (size_t i = 0; i < n; ++i) symbols[i] = dec_get(&h->d, t);
}
int ao_rans_decoder_is_empty(const ao_rans_decoder *h) { /* :385-389 */
    return h->d.pos >= h->d.len && h->d.state < AO_RANS32_L;
}
uint32_t ao_rans_decoder_state(const ao_rans_decoder *h) { return h->d.state; }
size_t ao_rans_decoder_pos(const ao_rans_decoder *h) { return h->d.pos; }

static void put_u32le(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
static uint32_t get_u32le(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* src/rans.rs:414-454 InterleavedRansEncoder::encode + finish */
int ao_rans_encode_interleaved(const uint8_t *symbols, size_t n, const ao_freq_table *t,
                               uint8_t **out, size_t *out_len) {
    rans_enc e[4];
    size_t count[4];
    for (int k = 0; k < 4; ++k) {
        e[k].state = AO_RANS32_L; e[k].buf = NULL; e[k].len = 0; e[k].cap = 0;
        count[k] = (n + 3 - (size_t)k) / 4;
    }
    int rc = AO_OK;
    for (size_t i = n; i-- > 0 && rc == AO_OK;) {
        uint8_t s = symbols[i];
        if ((size_t)s >= t->n_symbols) { rc = AO_ERR_REFERENCE_DIVERGES; break; }
        rc = enc_put(&e[i % 4], t->cum_freq[s], t->freq[s]);
    }
    uint8_t *bufs[4] = {0, 0, 0, 0};
    size_t lens[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4 && rc == AO_OK; ++k) rc = enc_finish(&e[k], &bufs[k], &lens[k]);
    if (rc) { for (int k = 0; k < 4; ++k) free(e[k].buf); return rc; }
    size_t total = 32 + lens[0] + lens[1] + lens[2] + lens[3];
    uint8_t *res = (uint8_t *)malloc(total);
    if (!res) { for (int k = 0; k < 4; ++k) free(bufs[k]); return AO_ERR_NOMEM; }
    for (int k = 0; k < 4; ++k) put_u32le(res + 4 * k, (uint32_t)lens[k]);
    for (int k = 0; k < 4; ++k) put_u32le(res + 16 + 4 * k, (uint32_t)count[k]);
    size_t off = 32;
    for (int k = 0; k < 4; ++k) { memcpy(res + off, bufs[k], lens[k]); off += lens[k]; free(bufs[k]); }
    *out = res;
    *out_len = total;
    return AO_OK;
}

/* src/rans.rs:474-523 InterleavedRansDecoder::new + decode_n
 * (the reference indexes out of bounds -> panic on short input; reported as an error here) */
int ao_rans_decode_interleaved(const uint8_t *in, size_t in_len, size_t n, const ao_freq_table *t,
                               uint8_t *symbols) {
    if (in_len < 32) return AO_ERR_INVALID_BITSTREAM;
    size_t len[4], rem[4];
    for (int k = 0; k < 4; ++k) { len[k] = get_u32le(in + 4 * k); rem[k] = get_u32le(in + 16 + 4 * k); }
    size_t off = 32;
    rans_dec d[4];
    for (int k = 0; k < 4; ++k) {
        if (off + len[k] > in_len) return AO_ERR_INVALID_BITSTREAM;
        dec_init(&d[k], in + off, len[k]);
        off += len[k];
    }
    if (rem[0] + rem[1] + rem[2] + rem[3] < n) return AO_ERR_INVALID_BITSTREAM; /* reference spins */
    size_t idx = 0;
    for (size_t i = 0; i < n; ++i) {
        while (rem[idx] == 0) idx = (idx + 1) % 4;
        symbols[i] = dec_get(&d[idx], t);
        rem[idx] -= 1;
        idx = (idx + 1) % 4;
    }
    return AO_OK;
}

/* ---- pipeline --------------------------------------------------------------- */

#define FIXED_HEADER_BYTES 18u    /* src/pipeline.rs:148 */
#define CHANNEL_HEADER_BYTES 1040u /* src/pipeline.rs:137 */

/* src/pipeline.rs:456-457 */
int32_t ao_quality_to_step(uint8_t quality) {
    int32_t q = quality > 100 ? 100 : quality;
    int32_t step = 64 - (q * 63) / 100;
    return step < 1 ? 1 : step;
}

/* src/pipeline.rs:67-71 checked_pixel_count */
static int checked_pixel_count(size_t w, size_t h, size_t f, size_t *out) {
    size_t wh, whf;
    if (__builtin_mul_overflow(w, h, &wh)) return AO_ERR_DIMENSION_OVERFLOW;
    if (__builtin_mul_overflow(wh, f, &whf)) return AO_ERR_DIMENSION_OVERFLOW;
    *out = whf;
    return AO_OK;
}

/* src/pipeline.rs:77-114 pad_channel_to_i32 */
static int32_t *pad_channel_to_i32(const int16_t *ch, size_t w, size_t h, size_t f, size_t pw,
                                   size_t ph, size_t pf) {
    size_t padded = pw * ph * pf;
    int32_t *buf = (int32_t *)calloc(padded ? padded : 1, sizeof(int32_t));
    if (!buf) return NULL;
    for (size_t t = 0; t < f; ++t) {
        for (size_t row = 0; row < h; ++row) {
            for (size_t col = 0; col < w; ++col)
                buf[t * pw * ph + row * pw + col] = ch[t * w * h + row * w + col];
            if (pw > w) buf[t * pw * ph + row * pw + w] = ch[t * w * h + row * w + (w - 1)];
        }
        if (ph > h)
            for (size_t col = 0; col < pw; ++col)
                buf[t * pw * ph + h * pw + col] = buf[t * pw * ph + (h - 1) * pw + col];
    }
    for (size_t t = f; t < pf; ++t)
        for (size_t idx = 0; idx < pw * ph; ++idx)
            buf[t * pw * ph + idx] = buf[(f - 1) * pw * ph + idx];
    return buf;
}

typedef struct {
    uint32_t compressed_len;
    int32_t quant_step;
    int32_t quant_dead_zone;
    uint32_t num_symbols;
    uint32_t histogram[256];
} channel_header;

static void write_header(uint8_t *p, int wavelet, uint32_t w, uint32_t h, uint32_t f,
                         const channel_header hdr[3]) {
    /* src/pipeline.rs:200-221 */
    memcpy(p, "ALCC", 4);
    p[4] = 1;
    p[5] = (uint8_t)wavelet;
    put_u32le(p + 6, w);
    put_u32le(p + 10, h);
    put_u32le(p + 14, f);
    size_t off = FIXED_HEADER_BYTES;
    for (int c = 0; c < 3; ++c) {
        put_u32le(p + off, hdr[c].compressed_len); off += 4;
        put_u32le(p + off, (uint32_t)hdr[c].quant_step); off += 4;
        put_u32le(p + off, (uint32_t)hdr[c].quant_dead_zone); off += 4;
        put_u32le(p + off, hdr[c].num_symbols); off += 4;
        for (int k = 0; k < 256; ++k) { put_u32le(p + off, hdr[c].histogram[k]); off += 4; }
    }
}

static void padded_dims(size_t w, size_t h, size_t f, size_t *pw, size_t *ph, size_t *pf) {
    /* src/pipeline.rs:437-439 / 547-549 */
    *pf = (f == 1) ? 2 : f + (f & 1);
    *pw = w + (w & 1);
    *ph = h + (h & 1);
}

/* shared front half of FrameEncoder::encode, src/pipeline.rs:377-477.
 * On success with n_pixels > 0: *symbols = 3 * padded u8 (channel-major). */
static int encode_front(const uint8_t *rgb, size_t rgb_len, uint32_t width, uint32_t height,
                        uint32_t frames, uint8_t quality, int wavelet, uint8_t **symbols,
                        size_t *padded_out, size_t *n_pixels_out) {
    size_t w = width, h = height, f = frames, n_pixels;
    int rc = checked_pixel_count(w, h, f, &n_pixels); /* :388 */
    if (rc) return rc;
    *n_pixels_out = n_pixels;
    *symbols = NULL;
    *padded_out = 0;
    if (n_pixels == 0) return rgb_len ? AO_ERR_INVALID_BUFFER_SIZE : AO_OK; /* :391-412 */
    if (w == 0 || h == 0) return AO_ERR_INVALID_DIMENSIONS;                   /* :415-417 */
    size_t expected;
    if (__builtin_mul_overflow(n_pixels, (size_t)3, &expected)) return AO_ERR_DIMENSION_OVERFLOW;
    if (rgb_len != expected) return AO_ERR_INVALID_BUFFER_SIZE;              /* :422-427 */

    int16_t *ych = (int16_t *)malloc(n_pixels * sizeof(int16_t));
    int16_t *coch = (int16_t *)malloc(n_pixels * sizeof(int16_t));
    int16_t *cgch = (int16_t *)malloc(n_pixels * sizeof(int16_t));
    if (!ych || !coch || !cgch) { free(ych); free(coch); free(cgch); return AO_ERR_NOMEM; }
    ao_rgb_bytes_to_ycocg_r(rgb, rgb_len, ych, coch, cgch, n_pixels); /* :434 */

    size_t pw, ph, pf;
    padded_dims(w, h, f, &pw, &ph, &pf);
    size_t padded = pw * ph * pf;
    int32_t step = ao_quality_to_step(quality);
    uint8_t *sym = (uint8_t *)malloc(3 * padded);
    int32_t *qbuf = (int32_t *)malloc(padded * sizeof(int32_t));
    if (!sym || !qbuf) { free(sym); free(qbuf); free(ych); free(coch); free(cgch); return AO_ERR_NOMEM; }
    const int16_t *chs[3] = {ych, coch, cgch};
    for (int c = 0; c < 3; ++c) { /* :461-477 */
        int32_t *buf = pad_channel_to_i32(chs[c], w, h, f, pw, ph, pf);
        if (!buf) { free(sym); free(qbuf); free(ych); free(coch); free(cgch); return AO_ERR_NOMEM; }
        ao_wavelet3d_forward(wavelet, buf, pw, ph, pf);
        ao_quantize_buffer(step, step, buf, qbuf, padded); /* Quantizer::new(step): dead_zone = step */
        ao_to_symbols(qbuf, sym + (size_t)c * padded, padded);
        free(buf);
    }
    free(qbuf); free(ych); free(coch); free(cgch);
    *symbols = sym;
    *padded_out = padded;
    return AO_OK;
}

int ao_encode_symbols(const uint8_t *rgb, size_t rgb_len, uint32_t width, uint32_t height,
                      uint32_t frames, uint8_t quality, int wavelet, uint8_t **symbols,
                      size_t *padded_pixels) {
    size_t n_pixels;
    return encode_front(rgb, rgb_len, width, height, frames, quality, wavelet, symbols,
                        padded_pixels, &n_pixels);
}

/* FrameEncoder::encode (src/pipeline.rs:377-507) followed by EncodedChunk::to_bytes (:200-226) */
int ao_encode(const uint8_t *rgb, size_t rgb_len, uint32_t width, uint32_t height, uint32_t frames,
              uint8_t quality, int wavelet, uint8_t **out, size_t *out_len) {
    uint8_t *sym = NULL;
    size_t padded = 0, n_pixels = 0;
    int rc = encode_front(rgb, rgb_len, width, height, frames, quality, wavelet, &sym, &padded,
                          &n_pixels);
    if (rc) return rc;
    channel_header hdr[3];
    for (int c = 0; c < 3; ++c) { /* defaults, :445-451 */
        hdr[c].compressed_len = 0; hdr[c].quant_step = 1; hdr[c].quant_dead_zone = 1;
        hdr[c].num_symbols = 0; memset(hdr[c].histogram, 0, sizeof(hdr[c].histogram));
    }
    uint8_t *streams[3] = {0, 0, 0};
    size_t lens[3] = {0, 0, 0};
    if (n_pixels != 0) {
        int32_t step = ao_quality_to_step(quality);
        for (int c = 0; c < 3; ++c) { /* :480-494 */
            ao_build_histogram(sym + (size_t)c * padded, padded, hdr[c].histogram);
            ao_freq_table tbl;
            rc = ao_freq_table_from_histogram(hdr[c].histogram, 256, &tbl);
            if (rc == AO_OK) rc = ao_rans_encode(sym + (size_t)c * padded, padded, &tbl, &streams[c], &lens[c]);
            ao_freq_table_free(&tbl);
            if (rc) { free(sym); for (int k = 0; k < 3; ++k) free(streams[k]); return rc; }
            hdr[c].compressed_len = (uint32_t)lens[c];
            hdr[c].quant_step = step;
            hdr[c].quant_dead_zone = step;
            hdr[c].num_symbols = (uint32_t)padded;
        }
    }
    free(sym);
    size_t total = FIXED_HEADER_BYTES + 3 * CHANNEL_HEADER_BYTES + lens[0] + lens[1] + lens[2];
    uint8_t *buf = (uint8_t *)malloc(total);
    if (!buf) { for (int k = 0; k < 3; ++k) free(streams[k]); return AO_ERR_NOMEM; }
    write_header(buf, wavelet, width, height, frames, hdr);
    size_t off = FIXED_HEADER_BYTES + 3 * CHANNEL_HEADER_BYTES;
    for (int c = 0; c < 3; ++c) {
        if (lens[c]) memcpy(buf + off, streams[c], lens[c]);
        off += lens[c];
        free(streams[c]);
    }
    *out = buf;
    *out_len = total;
    return AO_OK;
}

/* EncodedChunk::from_bytes (src/pipeline.rs:235-313) then FrameDecoder::decode (:537-624) */
int ao_decode(const uint8_t *alc, size_t alc_len, uint8_t **rgb, size_t *rgb_len) {
    size_t min_len = FIXED_HEADER_BYTES + 3 * CHANNEL_HEADER_BYTES;
    if (alc_len < min_len) return AO_ERR_INVALID_BITSTREAM;       /* :236-243 */
    if (memcmp(alc, "ALCC", 4) != 0) return AO_ERR_INVALID_BITSTREAM; /* :246-250 */
    if (alc[4] != 1) return AO_ERR_INVALID_BITSTREAM;             /* :252-257 */
    int wavelet = alc[5];
    if (wavelet > 2) return AO_ERR_INVALID_BITSTREAM;             /* :52-61 */
    uint32_t width = get_u32le(alc + 6), height = get_u32le(alc + 10), frames = get_u32le(alc + 14);
    channel_header hdr[3];
    size_t off = FIXED_HEADER_BYTES, total_compressed = 0;
    for (int c = 0; c < 3; ++c) { /* :275-294 */
        hdr[c].compressed_len = get_u32le(alc + off); off += 4;
        hdr[c].quant_step = (int32_t)get_u32le(alc + off); off += 4;
        hdr[c].quant_dead_zone = (int32_t)get_u32le(alc + off); off += 4;
        hdr[c].num_symbols = get_u32le(alc + off); off += 4;
        for (int k = 0; k < 256; ++k) { hdr[c].histogram[k] = get_u32le(alc + off); off += 4; }
        total_compressed += hdr[c].compressed_len;
    }
    if (alc_len < off + total_compressed) return AO_ERR_INVALID_BITSTREAM; /* :296-301 */
    const uint8_t *payload = alc + off;

    size_t w = width, h = height, f = frames, n_pixels;
    int rc = checked_pixel_count(w, h, f, &n_pixels); /* :541 */
    if (rc) return rc;
    if (n_pixels == 0) { *rgb = (uint8_t *)malloc(1); *rgb_len = 0; return AO_OK; } /* :543-545 */
    size_t pw, ph, pf;
    padded_dims(w, h, f, &pw, &ph, &pf);
    size_t padded = pw * ph * pf;

    /* validate before allocating (the reference allocates first; same results on success) */
    {
        size_t data_offset = 0;
        for (int c = 0; c < 3; ++c) {
            if ((size_t)hdr[c].num_symbols != padded) return AO_ERR_INVALID_BITSTREAM; /* :566-570 */
            if (data_offset + hdr[c].compressed_len > total_compressed) return AO_ERR_INVALID_BITSTREAM;
            data_offset += hdr[c].compressed_len;
        }
    }
    int16_t *chan[3];
    for (int c = 0; c < 3; ++c) chan[c] = (int16_t *)calloc(n_pixels, sizeof(int16_t));
    uint8_t *symbols = (uint8_t *)malloc(padded);
    int32_t *qbuf = (int32_t *)malloc(padded * sizeof(int32_t));
    int32_t *buf = (int32_t *)malloc(padded * sizeof(int32_t));
    uint8_t *out = (uint8_t *)malloc(n_pixels * 3);
    if (!chan[0] || !chan[1] || !chan[2] || !symbols || !qbuf || !buf || !out) {
        for (int c = 0; c < 3; ++c) free(chan[c]);
        free(symbols); free(qbuf); free(buf); free(out);
        return AO_ERR_NOMEM;
    }
    size_t data_offset = 0;
    for (int c = 0; c < 3; ++c) {
        const uint8_t *comp = payload + data_offset;
        size_t clen = hdr[c].compressed_len;
        data_offset += clen;
        ao_freq_table tbl;
        rc = ao_freq_table_from_histogram(hdr[c].histogram, 256, &tbl); /* :582 */
        if (rc) break;
        ao_rans_decode(comp, clen, padded, &tbl, symbols);               /* :585-586 */
        ao_freq_table_free(&tbl);
        ao_from_symbols(symbols, qbuf, padded);                          /* :589-590 */
        ao_dequantize_buffer(hdr[c].quant_step, qbuf, buf, padded);      /* :593-595 */
        ao_wavelet3d_inverse(wavelet, buf, pw, ph, pf);                  /* :598-599 */
        for (size_t t = 0; t < f; ++t)                                   /* :603-611 */
            for (size_t row = 0; row < h; ++row)
                for (size_t col = 0; col < w; ++col)
                    chan[c][t * w * h + row * w + col] =
                        (int16_t)(uint16_t)(uint32_t)buf[t * pw * ph + row * pw + col];
    }
    if (rc == AO_OK) rc = ao_ycocg_r_to_rgb_bytes(chan[0], chan[1], chan[2], n_pixels, out, n_pixels * 3);
    for (int c = 0; c < 3; ++c) free(chan[c]);
    free(symbols); free(qbuf); free(buf);
    if (rc) { free(out); return rc; }
    *rgb = out;
    *rgb_len = n_pixels * 3;
    return AO_OK;
}

/* ---- SSIM / MS-SSIM (src/ssim.rs:12-200) ---------------------------------------------------------------- */
static double ssim_raw(const double *a, const double *b, size_t len) { /* :18-49 */
    double n = (double)len;
    if (n < 1.0) return 1.0;
    double sa = 0.0, sb = 0.0;
    for (size_t i = 0; i < len; ++i) sa += a[i];
    for (size_t i = 0; i < len; ++i) sb += b[i];
    double mu_a = sa / n, mu_b = sb / n;
    double va = 0.0, vb = 0.0, vab = 0.0;
    for (size_t i = 0; i < len; ++i) {
        double da = a[i] - mu_a, db = b[i] - mu_b;
        va += da * da; vb += db * db; vab += da * db;
    }
    double denom = (n - 1.0) > 1.0 ? (n - 1.0) : 1.0;
    va /= denom; vb /= denom; vab /= denom;
    const double C1 = 6.5025, C2 = 58.5225;
    double numerator = fma(2.0 * mu_a, mu_b, C1) * fma(2.0, vab, C2);
    double denominator = (fma(mu_a, mu_a, mu_b * mu_b) + C1) * (va + vb + C2);
    return numerator / denominator;
}
int ao_ssim(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, size_t width, size_t height, double *out) {
    if (a_len != b_len) return AO_ERR_INVALID_BUFFER_SIZE;     /* :64-69 */
    if (a_len != width * height) return AO_ERR_INVALID_BUFFER_SIZE; /* :70-75 */
    if (a_len == 0) { *out = 1.0; return AO_OK; }
    double total = 0.0;
    uint64_t count = 0;
    double ba[64], bb[64];
    size_t bh = height / 8, bw = width / 8;
    for (size_t by = 0; by < bh; ++by)
        for (size_t bx = 0; bx < bw; ++bx) {
            size_t k = 0;
            for (size_t dy = 0; dy < 8; ++dy)
                for (size_t dx = 0; dx < 8; ++dx) {
                    size_t idx = (by * 8 + dy) * width + bx * 8 + dx;
                    ba[k] = (double)a[idx]; bb[k] = (double)b[idx]; ++k;
                }
            total += ssim_raw(ba, bb, 64);
            count += 1;
        }
    *out = count == 0 ? 1.0 : total / (double)count;
    return AO_OK;
}
static uint8_t *downsample_2x(const uint8_t *buf, size_t width, size_t height) { /* :181-200 */
    size_t nw = width / 2, nh = height / 2;
    uint8_t *out = (uint8_t *)malloc(nw * nh ? nw * nh : 1);
    if (!out) return NULL;
    for (size_t y = 0; y < nh; ++y)
        for (size_t x = 0; x < nw; ++x) {
            size_t sy = y * 2, sx = x * 2;
            unsigned avg = ((unsigned)buf[sy * width + sx] + buf[sy * width + sx + 1] + buf[(sy + 1) * width + sx] +
                            buf[(sy + 1) * width + sx + 1]) / 4;
            out[y * nw + x] = (uint8_t)avg;
        }
    return out;
}
int ao_ms_ssim(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, size_t width, size_t height, double *out) {
    if (a_len != b_len) return AO_ERR_INVALID_BUFFER_SIZE;
    if (a_len != width * height) return AO_ERR_INVALID_BUFFER_SIZE;
    if (a_len == 0) { *out = 1.0; return AO_OK; }
    const double weights[3] = {0.3333, 0.3333, 0.3334};
    uint8_t *ca = (uint8_t *)malloc(a_len), *cb = (uint8_t *)malloc(a_len);
    if (!ca || !cb) { free(ca); free(cb); return AO_ERR_NOMEM; }
    memcpy(ca, a, a_len); memcpy(cb, b, a_len);
    size_t cw = width, ch = height;
    double result = 0.0;
    for (int wi = 0; wi < 3; ++wi) {
        double weight = weights[wi], s = 0.0;
        int rc = ao_ssim(ca, cw * ch, cb, cw * ch, cw, ch, &s);
        if (rc) { free(ca); free(cb); return rc; }
        double l = log(s > 0.0 ? s : 0.0);              /* s.max(0.0).ln().max(-10.0) */
        if (!(l > -10.0)) l = -10.0;
        result += weight * l;
        size_t nw = cw / 2, nh = ch / 2;
        if (nw < 8 || nh < 8) {
            /* the reference looks the current weight up BY VALUE (first match), :153-163: on the second scale that is
             * index 0 again, so weights[1] is counted once more */
            int pos = 0;
            for (int k = 0; k < 3; ++k) if (fabs(weights[k] - weight) < 1e-10) { pos = k; break; }
            for (int k = pos + 1; k < 3; ++k) result += weights[k] * l;
            break;
        }
        uint8_t *na = downsample_2x(ca, cw, ch), *nb = downsample_2x(cb, cw, ch);
        free(ca); free(cb);
        ca = na; cb = nb;
        if (!ca || !cb) { free(ca); free(cb); return AO_ERR_NOMEM; }
        cw = nw; ch = nh;
    }
    free(ca); free(cb);
    *out = exp(result);
    return AO_OK;
}

/* ---- AnalyticalRDO (src/quant.rs:377-505) and SubBand3D::quant_strength (src/lib.rs:149-158) ---------------- */
double ao_rdo_target_bpp(uint8_t quality) { /* with_quality, :398-411 */
    const double RCP_100 = 1.0 / 100.0;
    unsigned qq = quality > 100 ? 100u : quality;
    double q = (double)qq * RCP_100;
    return fma(q * q, 23.9, 0.1); /* (q * q).mul_add(23.9, 0.1) */
}
int ao_subband_quant_strength(int subband) {
    switch (subband) {
    case 0: return 1;
    case 1: case 2: case 4: return 2;
    case 3: case 5: case 6: return 4;
    default: return 8;
    }
}
/* estimate_variance (:414-435): exact i64 sum, then a sequential f64 sum of squared deviations */
double ao_rdo_estimate_variance(const int32_t *coeffs, size_t n) {
    if (n == 0) return 1.0;
    double nn = (double)n, inv_n = 1.0 / nn;
    int64_t sum = 0;
    for (size_t i = 0; i < n; ++i) sum += (int64_t)coeffs[i];
    double mean = (double)sum * inv_n;
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double diff = (double)coeffs[i] - mean;
        double sq = diff * diff;
        acc = acc + sq;
    }
    double variance = acc * inv_n;
    return variance > 1.0 ? variance : 1.0; /* f64::max */
}
/* compute_quantizer (:455-470): step and dead zone of the Quantizer it returns */
void ao_rdo_compute_quantizer(double target_bpp, const int32_t *coeffs, size_t n, int subband, int32_t *step,
                              int32_t *dead_zone) {
    double variance = ao_rdo_estimate_variance(coeffs, n);
    double lambda = (6.0 * M_LN2 * variance) / target_bpp;     /* :440-443 */
    double st = sqrt(12.0 * lambda);                          /* :448-451 */
    double r = round(st);
    int32_t base = (r >= 2147483647.0) ? INT32_MAX : (r <= -2147483648.0 ? INT32_MIN : (int32_t)r); /* `as i32` saturates */
    if (r != r) base = 0;
    if (base < 1) base = 1;
    int32_t s = (int32_t)((uint32_t)base * (uint32_t)ao_subband_quant_strength(subband));
    if (s < 1) s = 1;
    *step = s;
    *dead_zone = (int32_t)((uint32_t)s + (uint32_t)(s / 2));
}

/* ---- NOT the reference: the same encode/decode with the three channels on three threads --------------------
 * The reference is single-threaded (it has no rayon path).  These two functions are what
 * a `rayon::join` over Y / Co / Cg would give and exist only so that bench.py can time that hypothetical
 * variant beside the faithful one (variant ii).  Same stage functions, same bytes. */
#include <pthread.h>

typedef struct {
    /* encode */
    const int16_t *plane; size_t w, h, f, pw, ph, pf, padded; int wavelet; int32_t step;
    channel_header *hdr; uint8_t *stream; size_t len;
    /* decode */
    const uint8_t *comp; size_t clen; int16_t *out_plane;
    int rc;
} chan_job;

static void *enc_channel(void *arg) {
    chan_job *j = (chan_job *)arg;
    j->rc = AO_ERR_NOMEM;
    int32_t *buf = pad_channel_to_i32(j->plane, j->w, j->h, j->f, j->pw, j->ph, j->pf);
    int32_t *qbuf = (int32_t *)malloc(j->padded * sizeof(int32_t));
    uint8_t *sym = (uint8_t *)malloc(j->padded);
    if (buf && qbuf && sym) {
        ao_wavelet3d_forward(j->wavelet, buf, j->pw, j->ph, j->pf);
        ao_quantize_buffer(j->step, j->step, buf, qbuf, j->padded);
        ao_to_symbols(qbuf, sym, j->padded);
        ao_build_histogram(sym, j->padded, j->hdr->histogram);
        ao_freq_table tbl;
        j->rc = ao_freq_table_from_histogram(j->hdr->histogram, 256, &tbl);
        if (j->rc == AO_OK) j->rc = ao_rans_encode(sym, j->padded, &tbl, &j->stream, &j->len);
        ao_freq_table_free(&tbl);
        j->hdr->compressed_len = (uint32_t)j->len;
        j->hdr->quant_step = j->step; j->hdr->quant_dead_zone = j->step; j->hdr->num_symbols = (uint32_t)j->padded;
    }
    free(buf); free(qbuf); free(sym);
    return NULL;
}

int ao_encode_par3(const uint8_t *rgb, size_t rgb_len, uint32_t width, uint32_t height, uint32_t frames,
                   uint8_t quality, int wavelet, uint8_t **out, size_t *out_len) {
    size_t w = width, h = height, f = frames, n_pixels;
    int rc = checked_pixel_count(w, h, f, &n_pixels);
    if (rc) return rc;
    if (n_pixels == 0 || w == 0 || h == 0 || rgb_len != n_pixels * 3)
        return ao_encode(rgb, rgb_len, width, height, frames, quality, wavelet, out, out_len);
    int16_t *pl[3];
    for (int c = 0; c < 3; ++c) pl[c] = (int16_t *)malloc(n_pixels * sizeof(int16_t));
    if (!pl[0] || !pl[1] || !pl[2]) { for (int c = 0; c < 3; ++c) free(pl[c]); return AO_ERR_NOMEM; }
    ao_rgb_bytes_to_ycocg_r(rgb, rgb_len, pl[0], pl[1], pl[2], n_pixels);
    size_t pw, ph, pf;
    padded_dims(w, h, f, &pw, &ph, &pf);
    channel_header hdr[3];
    chan_job job[3];
    pthread_t th[3];
    for (int c = 0; c < 3; ++c) {
        memset(&hdr[c], 0, sizeof(hdr[c]));
        memset(&job[c], 0, sizeof(job[c]));
        job[c].plane = pl[c]; job[c].w = w; job[c].h = h; job[c].f = f; job[c].pw = pw; job[c].ph = ph; job[c].pf = pf;
        job[c].padded = pw * ph * pf; job[c].wavelet = wavelet; job[c].step = ao_quality_to_step(quality); job[c].hdr = &hdr[c];
        pthread_create(&th[c], NULL, enc_channel, &job[c]);
    }
    for (int c = 0; c < 3; ++c) { pthread_join(th[c], NULL); free(pl[c]); if (job[c].rc) rc = job[c].rc; }
    if (rc) { for (int c = 0; c < 3; ++c) free(job[c].stream); return rc; }
    size_t total = FIXED_HEADER_BYTES + 3 * CHANNEL_HEADER_BYTES + job[0].len + job[1].len + job[2].len;
    uint8_t *buf = (uint8_t *)malloc(total);
    if (!buf) { for (int c = 0; c < 3; ++c) free(job[c].stream); return AO_ERR_NOMEM; }
    write_header(buf, wavelet, width, height, frames, hdr);
    size_t off = FIXED_HEADER_BYTES + 3 * CHANNEL_HEADER_BYTES;
    for (int c = 0; c < 3; ++c) { if (job[c].len) memcpy(buf + off, job[c].stream, job[c].len); off += job[c].len; free(job[c].stream); }
    *out = buf; *out_len = total;
    return AO_OK;
}

static void *dec_channel(void *arg) {
    chan_job *j = (chan_job *)arg;
    j->rc = AO_ERR_NOMEM;
    uint8_t *symbols = (uint8_t *)malloc(j->padded);
    int32_t *qbuf = (int32_t *)malloc(j->padded * sizeof(int32_t));
    int32_t *buf = (int32_t *)malloc(j->padded * sizeof(int32_t));
    if (symbols && qbuf && buf) {
        ao_freq_table tbl;
        j->rc = ao_freq_table_from_histogram(j->hdr->histogram, 256, &tbl);
        if (j->rc == AO_OK) {
            ao_rans_decode(j->comp, j->clen, j->padded, &tbl, symbols);
            ao_freq_table_free(&tbl);
            ao_from_symbols(symbols, qbuf, j->padded);
            ao_dequantize_buffer(j->hdr->quant_step, qbuf, buf, j->padded);
            ao_wavelet3d_inverse(j->wavelet, buf, j->pw, j->ph, j->pf);
            for (size_t t = 0; t < j->f; ++t)
                for (size_t row = 0; row < j->h; ++row)
                    for (size_t col = 0; col < j->w; ++col)
                        j->out_plane[t * j->w * j->h + row * j->w + col] =
                            (int16_t)(uint16_t)(uint32_t)buf[t * j->pw * j->ph + row * j->pw + col];
        }
    }
    free(symbols); free(qbuf); free(buf);
    return NULL;
}

/* valid, non-empty chunks only (everything else goes through ao_decode, which carries the validation) */
int ao_decode_par3(const uint8_t *alc, size_t alc_len, uint8_t **rgb, size_t *rgb_len) {
    size_t hdr_len = FIXED_HEADER_BYTES + 3 * CHANNEL_HEADER_BYTES;
    if (alc_len < hdr_len || memcmp(alc, "ALCC", 4) != 0 || alc[4] != 1 || alc[5] > 2) return ao_decode(alc, alc_len, rgb, rgb_len);
    int wavelet = alc[5];
    size_t w = get_u32le(alc + 6), h = get_u32le(alc + 10), f = get_u32le(alc + 14), n_pixels;
    if (checked_pixel_count(w, h, f, &n_pixels) || n_pixels == 0) return ao_decode(alc, alc_len, rgb, rgb_len);
    size_t pw, ph, pf;
    padded_dims(w, h, f, &pw, &ph, &pf);
    channel_header hdr[3];
    size_t off = FIXED_HEADER_BYTES, total = 0;
    for (int c = 0; c < 3; ++c) {
        hdr[c].compressed_len = get_u32le(alc + off); off += 4;
        hdr[c].quant_step = (int32_t)get_u32le(alc + off); off += 4;
        hdr[c].quant_dead_zone = (int32_t)get_u32le(alc + off); off += 4;
        hdr[c].num_symbols = get_u32le(alc + off); off += 4;
        for (int k = 0; k < 256; ++k) { hdr[c].histogram[k] = get_u32le(alc + off); off += 4; }
        total += hdr[c].compressed_len;
        if ((size_t)hdr[c].num_symbols != pw * ph * pf) return ao_decode(alc, alc_len, rgb, rgb_len);
    }
    if (alc_len < off + total) return ao_decode(alc, alc_len, rgb, rgb_len);
    int16_t *pl[3];
    uint8_t *out = (uint8_t *)malloc(n_pixels * 3);
    for (int c = 0; c < 3; ++c) pl[c] = (int16_t *)calloc(n_pixels, sizeof(int16_t));
    if (!out || !pl[0] || !pl[1] || !pl[2]) { free(out); for (int c = 0; c < 3; ++c) free(pl[c]); return AO_ERR_NOMEM; }
    chan_job job[3];
    pthread_t th[3];
    size_t doff = 0;
    for (int c = 0; c < 3; ++c) {
        memset(&job[c], 0, sizeof(job[c]));
        job[c].w = w; job[c].h = h; job[c].f = f; job[c].pw = pw; job[c].ph = ph; job[c].pf = pf; job[c].padded = pw * ph * pf;
        job[c].wavelet = wavelet; job[c].hdr = &hdr[c]; job[c].comp = alc + off + doff; job[c].clen = hdr[c].compressed_len;
        job[c].out_plane = pl[c];
        doff += hdr[c].compressed_len;
        pthread_create(&th[c], NULL, dec_channel, &job[c]);
    }
    int rc = AO_OK;
    for (int c = 0; c < 3; ++c) { pthread_join(th[c], NULL); if (job[c].rc) rc = job[c].rc; }
    if (rc == AO_OK) rc = ao_ycocg_r_to_rgb_bytes(pl[0], pl[1], pl[2], n_pixels, out, n_pixels * 3);
    for (int c = 0; c < 3; ++c) free(pl[c]);
    if (rc) { free(out); return rc; }
    *rgb = out; *rgb_len = n_pixels * 3;
    return AO_OK;
}

/* src/metrics.rs:16-63 mse + psnr (sequential f64 sum; libm log10) */
double ao_psnr(const uint8_t *a, const uint8_t *b, size_t len) {
    if (len == 0) return INFINITY;
    double sum = 0.0;
    for (size_t i = 0; i < len; ++i) {
        double d = (double)a[i] - (double)b[i];
        sum += d * d;
    }
    double mse = sum / (double)len;
    if (mse == 0.0) return INFINITY;
    return 10.0 * log10(255.0 * 255.0 / mse);
}
