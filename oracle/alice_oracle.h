/*
 * alice_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE)
 *
 * Scalar C restatement of the ALICE-Codec encode/decode hot path, written from
 * a reading of the reference Rust sources (cited per function as file:line,
 * relative to the reference checkout).  It exists only so that tests/, the
 * smoke check in __graft_entry__.py and the cpu_baseline leg of bench.py have
 * something to compare the HIP path against.  Nothing under alice-codec_amd/
 * may include, link or call it.
 *
 * PINNING STATUS
 *   - stage level: pinned by every exact-value assertion the reference's own
 *     inline tests make for this path (tests/test_oracle_reference_kats.py
 *     lists each with its file:line).
 *   - whole-bitstream (.alc byte pattern): PARITY UNPINNED.  The reference
 *     holds no golden .alc file and cannot be built here (Rust toolchain
 *     absent), so the byte pattern rests on source reading, cross-checked by an
 *     independently structured numpy restatement (oracle/alice_oracle_np.py).
 *
 * Rust semantics restated explicitly: signed >> is arithmetic, / truncates
 * toward zero, `as` casts truncate, and (release profile, Cargo.toml:46-51)
 * integer overflow wraps.
 */
#ifndef ALICE_ORACLE_H
#define ALICE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes: src/error.rs:12-23 (CodecError variants) + one divergence */
enum {
    AO_OK = 0,
    AO_ERR_INVALID_BUFFER_SIZE = 1,
    AO_ERR_INVALID_DIMENSIONS = 2,
    AO_ERR_DIMENSION_OVERFLOW = 3,
    AO_ERR_INVALID_BITSTREAM = 4,
    AO_ERR_INVALID_QUANT_STEP = 5,
    /* The reference would loop forever / divide by zero (src/rans.rs:275-283)
     * when a symbol whose table frequency wrapped to 0 is encoded.  The one
     * intentional divergence: report it instead of hanging. */
    AO_ERR_REFERENCE_DIVERGES = 6,
    AO_ERR_NOMEM = 7
};

/* WaveletType: src/pipeline.rs:34-41 */
enum { AO_WAVELET_CDF53 = 0, AO_WAVELET_CDF97 = 1, AO_WAVELET_HAAR = 2 };

/* ---- colour: src/color.rs:199-235, 245-276 ---- */
int ao_rgb_bytes_to_ycocg_r(const uint8_t *rgb, size_t rgb_len, int16_t *y, int16_t *co,
                            int16_t *cg, size_t out_len);
int ao_ycocg_r_to_rgb_bytes(const int16_t *y, const int16_t *co, const int16_t *cg, size_t n,
                            uint8_t *rgb, size_t rgb_len);

/* ---- wavelet: src/wavelet.rs:66-248, 292-340, 392-484 ---- */
void ao_wavelet1d_forward(int type, int32_t *signal, size_t n);
void ao_wavelet1d_inverse(int type, int32_t *signal, size_t n);
void ao_wavelet2d_forward(int type, int32_t *image, size_t width, size_t height);
void ao_wavelet2d_inverse(int type, int32_t *image, size_t width, size_t height);
void ao_wavelet3d_forward(int type, int32_t *volume, size_t width, size_t height, size_t depth);
void ao_wavelet3d_inverse(int type, int32_t *volume, size_t width, size_t height, size_t depth);

/* ---- quantizer: src/quant.rs:70-146 (Quantizer), 190-311 (FastQuantizer) ---- */
int32_t ao_quantize(int32_t step, int32_t dead_zone, int32_t value);
int32_t ao_dequantize(int32_t step, int32_t qvalue);
void ao_quantize_buffer(int32_t step, int32_t dead_zone, const int32_t *in, int32_t *out, size_t n);
void ao_dequantize_buffer(int32_t step, const int32_t *in, int32_t *out, size_t n);

typedef struct {
    uint64_t reciprocal;
    uint32_t shift;
    int32_t step;
    int32_t dead_zone;
} ao_fast_quantizer;
int ao_fast_quantizer_new(int32_t step, ao_fast_quantizer *q);
int ao_fast_quantizer_with_dead_zone(int32_t step, int32_t dead_zone, ao_fast_quantizer *q);
int32_t ao_fast_quantize(const ao_fast_quantizer *q, int32_t value);
void ao_fast_quantize_buffer(const ao_fast_quantizer *q, const int32_t *in, int32_t *out, size_t n);

/* ---- symbol mapping / histogram: src/quant.rs:547-600 ---- */
void ao_to_symbols(const int32_t *coeffs, uint8_t *symbols, size_t n);
void ao_from_symbols(const uint8_t *symbols, int32_t *coeffs, size_t n);
void ao_build_histogram(const uint8_t *symbols, size_t n, uint32_t hist[256]);

/* ---- rANS: src/rans.rs:50-72, 102-219, 244-389, 393-524 ---- */
#define AO_PROB_BITS 12u
#define AO_PROB_SCALE 4096u
#define AO_RANS32_L (1u << 23)

typedef struct {
    size_t n_symbols;
    uint16_t *cum_freq; /* [n_symbols] */
    uint16_t *freq;     /* [n_symbols] */
    uint8_t cum_to_sym[AO_PROB_SCALE];
} ao_freq_table;

int ao_freq_table_from_histogram(const uint32_t *hist, size_t n_symbols, ao_freq_table *t);
int ao_freq_table_uniform(size_t n_symbols, ao_freq_table *t);
void ao_freq_table_free(ao_freq_table *t);

/* encode_symbols + finish.  *out is malloc'ed; caller frees with ao_free. */
int ao_rans_encode(const uint8_t *symbols, size_t n, const ao_freq_table *t, uint8_t **out,
                   size_t *out_len);
/* RansDecoder::new + decode_n */
void ao_rans_decode(const uint8_t *in, size_t in_len, size_t n, const ao_freq_table *t,
                    uint8_t *symbols);
/* the same coders as objects that live across calls: RansEncoder::{new, encode, encode_symbols, finish} :249-308,
 * RansDecoder::{new, decode_n, is_empty} :330-389 */
typedef struct ao_rans_encoder ao_rans_encoder;
typedef struct ao_rans_decoder ao_rans_decoder;
ao_rans_encoder *ao_rans_encoder_new(void);
void ao_rans_encoder_free(ao_rans_encoder *h);
int ao_rans_encoder_encode(ao_rans_encoder *h, uint16_t cum_freq, uint16_t freq);
int ao_rans_encoder_encode_symbols(ao_rans_encoder *h, const uint8_t *symbols, size_t n, const ao_freq_table *t);
uint32_t ao_rans_encoder_state(const ao_rans_encoder *h);
int ao_rans_encoder_finish(ao_rans_encoder *h, uint8_t **out, size_t *out_len); /* frees h */
ao_rans_decoder *ao_rans_decoder_new(const uint8_t *in, size_t len);
void ao_rans_decoder_free(ao_rans_decoder *h);
void ao_rans_decoder_decode_n(ao_rans_decoder *h, size_t n, const ao_freq_table *t, uint8_t *symbols);
int ao_rans_decoder_is_empty(const ao_rans_decoder *h);
uint32_t ao_rans_decoder_state(const ao_rans_decoder *h);
size_t ao_rans_decoder_pos(const ao_rans_decoder *h);
int ao_rans_encode_interleaved(const uint8_t *symbols, size_t n, const ao_freq_table *t,
                               uint8_t **out, size_t *out_len);
int ao_rans_decode_interleaved(const uint8_t *in, size_t in_len, size_t n, const ao_freq_table *t,
                               uint8_t *symbols);

/* ---- pipeline: src/pipeline.rs:67-114, 200-313, 377-507, 537-624 ---- */
int32_t ao_quality_to_step(uint8_t quality); /* pipeline.rs:456-457 */
/* FrameEncoder::with_wavelet(q, w).encode(..).to_bytes(); *out malloc'ed */
int ao_encode(const uint8_t *rgb, size_t rgb_len, uint32_t width, uint32_t height, uint32_t frames,
              uint8_t quality, int wavelet, uint8_t **out, size_t *out_len);
/* FrameDecoder::new().decode(EncodedChunk::from_bytes(..)); *rgb malloc'ed */
int ao_decode(const uint8_t *alc, size_t alc_len, uint8_t **rgb, size_t *rgb_len);
/* ssim / ms_ssim (src/ssim.rs:63-176): mean SSIM over 8x8 blocks; 3-scale variant */
int ao_ssim(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, size_t width, size_t height, double *out);
int ao_ms_ssim(const uint8_t *a, size_t a_len, const uint8_t *b, size_t b_len, size_t width, size_t height, double *out);
/* AnalyticalRDO (src/quant.rs:377-505), SubBand3D::quant_strength (src/lib.rs:149-158) */
double ao_rdo_target_bpp(uint8_t quality);
int ao_subband_quant_strength(int subband);
double ao_rdo_estimate_variance(const int32_t *coeffs, size_t n);
void ao_rdo_compute_quantizer(double target_bpp, const int32_t *coeffs, size_t n, int subband, int32_t *step,
                              int32_t *dead_zone);
/* NOT the reference (it is single-threaded): the same two calls with Y, Co, Cg on three threads -- the hypothetical
 * "rayon::join" variant that bench.py times beside the faithful one.  Byte-identical results. */
int ao_encode_par3(const uint8_t *rgb, size_t rgb_len, uint32_t width, uint32_t height, uint32_t frames,
                   uint8_t quality, int wavelet, uint8_t **out, size_t *out_len);
int ao_decode_par3(const uint8_t *alc, size_t alc_len, uint8_t **rgb, size_t *rgb_len);
/* encode front half only: per-channel u8 symbols (3 * padded_pixels, channel-major) */
int ao_encode_symbols(const uint8_t *rgb, size_t rgb_len, uint32_t width, uint32_t height,
                      uint32_t frames, uint8_t quality, int wavelet, uint8_t **symbols,
                      size_t *padded_pixels);

/* metrics::psnr, src/metrics.rs:16-63 (returns -1.0 on length mismatch via caller) */
double ao_psnr(const uint8_t *a, const uint8_t *b, size_t len);

void ao_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
