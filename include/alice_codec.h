/*
 * alice_codec.h -- C ABI of libalice_codec.so, the MI355X (gfx950) encode/decode path.
 *
 * PART 1 is the drop-in boundary: the 20 functions the reference `cdylib` exports from
 * src/ffi.rs (C statement of the ABI: bindings/ue5/AliceCodec.h:14-68 in the reference),
 * with the same names, signatures, ownership and failure conventions (NULL / -1.0 / 0,
 * `*out_len` written only on success).  Every call runs on the GPU; there is no CPU
 * fallback -- without a usable HIP device the calls fail (NULL) and
 * alice_codec_last_error() reports ALICE_ERR_DEVICE.
 *
 * PART 2 are extension entry points the reference ABI cannot express: wavelet selection
 * (the reference FFI can only create CDF 5/3 encoders, src/ffi.rs:92-94), 64-bit lengths
 * (src/ffi.rs:119 carries u32), device-resident batches of chunks (many rANS chains in
 * flight is the only parallelism the single-stream format offers), and stage-level calls
 * for the Rust API surface named by the task (Wavelet2D/3D, Quantizer/FastQuantizer,
 * to_symbols/from_symbols/build_histogram, FrequencyTable, RansEncoder/RansDecoder, colour).
 *
 * Handles are opaque and immutable after creation; encode/decode may be called from many
 * threads on the same handle (reference: Send + Sync, src/pipeline.rs:635-644).  Concurrent
 * encode / decode calls scale: the serial entropy chains of all calls in flight leave in merged
 * kernel launches (chain hub), so a thread pool over chunks keeps as many
 * chunks' chains running as the device holds, not one per hardware queue.
 */
#ifndef ALICE_CODEC_H
#define ALICE_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Wavelet1D Wavelet1D;
typedef struct FrameEncoder FrameEncoder;
typedef struct EncodedChunk EncodedChunk;

/* ===================== PART 1: reference ABI (src/ffi.rs) ===================== */

Wavelet1D *alice_codec_wavelet1d_haar(void);                 /* src/ffi.rs:16  */
Wavelet1D *alice_codec_wavelet1d_cdf53(void);                /* src/ffi.rs:22  */
Wavelet1D *alice_codec_wavelet1d_cdf97(void);                /* src/ffi.rs:28  */
void alice_codec_wavelet1d_destroy(Wavelet1D *ptr);          /* src/ffi.rs:38  null-ok */
/* no-op if wavelet/data is NULL or len < 2 */
void alice_codec_wavelet1d_forward(const Wavelet1D *wavelet, int32_t *data, uint32_t len); /* src/ffi.rs:52 */
void alice_codec_wavelet1d_inverse(const Wavelet1D *wavelet, int32_t *data, uint32_t len); /* src/ffi.rs:73 */

FrameEncoder *alice_codec_encoder_create(uint8_t quality);   /* src/ffi.rs:92  always CDF 5/3 */
void alice_codec_encoder_destroy(FrameEncoder *ptr);         /* src/ffi.rs:102 null-ok */
/* NULL on any error or NULL argument */
EncodedChunk *alice_codec_encode(const FrameEncoder *encoder, const uint8_t *rgb, uint32_t rgb_len,
                                 uint32_t width, uint32_t height, uint32_t frames); /* src/ffi.rs:116 */
/* NULL on error; caller frees with alice_codec_data_free(ptr, *out_len) */
uint8_t *alice_codec_decode(const EncodedChunk *chunk, uint32_t *out_len);          /* src/ffi.rs:145 */

void alice_codec_chunk_destroy(EncodedChunk *ptr);                                  /* src/ffi.rs:171 */
uint8_t *alice_codec_chunk_to_bytes(const EncodedChunk *chunk, uint32_t *out_len);  /* src/ffi.rs:185 */
EncodedChunk *alice_codec_chunk_from_bytes(const uint8_t *data, uint32_t len);      /* src/ffi.rs:207 */
uint32_t alice_codec_chunk_width(const EncodedChunk *chunk);                        /* src/ffi.rs:226 */
uint32_t alice_codec_chunk_height(const EncodedChunk *chunk);                       /* src/ffi.rs:240 */
uint32_t alice_codec_chunk_frames(const EncodedChunk *chunk);                       /* src/ffi.rs:254 */

/* -1.0 on NULL; +inf when identical or empty */
double alice_codec_psnr(const uint8_t *a, const uint8_t *b, uint32_t len);          /* src/ffi.rs:270 */

void alice_codec_data_free(uint8_t *ptr, uint32_t len);                             /* src/ffi.rs:288 */
void alice_codec_string_free(char *s);                                              /* src/ffi.rs:302 */
char *alice_codec_version(void);                                                    /* src/ffi.rs:311 */

/* ===================== PART 2: extensions ===================== */

/* CodecError (src/error.rs:12-23) as integers, plus library-side conditions */
enum {
    ALICE_OK = 0,
    ALICE_ERR_INVALID_BUFFER_SIZE = 1,
    ALICE_ERR_INVALID_DIMENSIONS = 2,
    ALICE_ERR_DIMENSION_OVERFLOW = 3,
    ALICE_ERR_INVALID_BITSTREAM = 4,
    ALICE_ERR_INVALID_QUANT_STEP = 5,
    ALICE_ERR_REFERENCE_DIVERGES = 6, /* the reference would hang/divide by zero (src/rans.rs:275-283) */
    ALICE_ERR_OUT_OF_MEMORY = 7,
    ALICE_ERR_DEVICE = 8,
    ALICE_ERR_NULL_ARGUMENT = 9,
    ALICE_ERR_INTERNAL = 10
};
/* WaveletType (src/pipeline.rs:34-41) */
enum { ALICE_WAVELET_CDF53 = 0, ALICE_WAVELET_CDF97 = 1, ALICE_WAVELET_HAAR = 2 };

int alice_codec_last_error(void);                 /* code of the last failing call on this thread */
const char *alice_codec_last_error_message(void); /* thread-local, valid until the next call */
int alice_codec_device_count(void);
int alice_codec_set_device(int device);           /* device used by this thread's later calls */
void alice_codec_trim(void);                      /* release cached device memory */
/* (test and measurement hooks live in alice_codec_test.h, which this header does not include) */

/* FrameEncoder::with_wavelet (src/pipeline.rs:356) */
FrameEncoder *alice_codec_encoder_create_ex(uint8_t quality, uint8_t wavelet_type);
uint8_t alice_codec_encoder_quality(const FrameEncoder *encoder);
uint8_t alice_codec_encoder_wavelet(const FrameEncoder *encoder);
/* EncodedChunk public fields / compressed_size (src/pipeline.rs:172-192) */
uint8_t alice_codec_chunk_wavelet(const EncodedChunk *chunk);
uint64_t alice_codec_chunk_compressed_size(const EncodedChunk *chunk);

/* 64-bit length variants of #9, #10, #12, #13, #18 */
EncodedChunk *alice_codec_encode64(const FrameEncoder *encoder, const uint8_t *rgb, uint64_t rgb_len,
                                   uint32_t width, uint32_t height, uint32_t frames);
uint8_t *alice_codec_decode64(const EncodedChunk *chunk, uint64_t *out_len);
uint8_t *alice_codec_chunk_to_bytes64(const EncodedChunk *chunk, uint64_t *out_len);
EncodedChunk *alice_codec_chunk_from_bytes64(const uint8_t *data, uint64_t len);
void alice_codec_data_free64(uint8_t *ptr, uint64_t len);

/* ---- many equal-shaped chunks from host memory in one call (what a 64-frame chunk driver wants: the serial
 * entropy chains of all chunks run side by side).  rgb = n_chunks chunks back to back; out_chunks[n_chunks] receives
 * handles to free with alice_codec_chunk_destroy.  Same results as n_chunks calls of alice_codec_encode64. ---- */
int alice_codec_encode_many(const FrameEncoder *encoder, const uint8_t *rgb, uint64_t rgb_len, uint32_t width,
                            uint32_t height, uint32_t frames, uint32_t n_chunks, EncodedChunk **out_chunks);
int alice_codec_decode_many(const EncodedChunk *const *chunks, uint32_t n_chunks, uint8_t *rgb_out, uint64_t rgb_out_len);
/* (Both take as many chunks at a time as the device's memory holds and loop over the rest.)
 *
 * The same over several GPUs of the node, for hosts that are not Python (the chunk driver of src/pipeline.rs:461-497:
 * 64-frame chunks are independent bitstreams).  Chunk k runs on devices[k mod n_devices]; the library starts one host
 * thread per entry of the list, every device copies its own chunks in and its own results out (its
 * own PCIe link; nothing is staged on another GPU), and results arrive in chunk order.  A device may be listed more than
 * once (two host threads sharing it).  Byte-identical to alice_codec_encode_many / decode_many for every device list.
 * alice_codec_many_devices_plan fills device_of_chunk[n_chunks] with that assignment (no device is touched). */
int alice_codec_many_devices_plan(uint32_t n_chunks, const int *devices, uint32_t n_devices, int *device_of_chunk);
int alice_codec_encode_many_devices(const FrameEncoder *encoder, const uint8_t *rgb, uint64_t rgb_len, uint32_t width,
                                    uint32_t height, uint32_t frames, uint32_t n_chunks, const int *devices,
                                    uint32_t n_devices, EncodedChunk **out_chunks);
int alice_codec_decode_many_devices(const EncodedChunk *const *chunks, uint32_t n_chunks, const int *devices,
                                    uint32_t n_devices, uint8_t *rgb_out, uint64_t rgb_out_len);

/* ---- device-resident batches: n_chunks equal-shaped chunks, inputs and outputs in HBM ---- */
typedef struct AliceBatch AliceBatch;
AliceBatch *alice_codec_batch_create(uint32_t width, uint32_t height, uint32_t frames, uint32_t n_chunks,
                                     uint8_t quality, uint8_t wavelet_type);
void alice_codec_batch_destroy(AliceBatch *batch);
/* d_rgb: device pointer to n_chunks * width*height*frames*3 bytes.  hip_stream: hipStream_t (NULL = default).
 * Runs the transforms, waits for them once to size the stream regions from the histograms, then queues the
 * entropy coding and the .alc assembly asynchronously; the .alc buffers stay on the device.
 * d_rgb must stay valid until alice_codec_batch_encode_finish. Returns an ALICE_* code. */
int alice_codec_batch_encode(AliceBatch *batch, const void *d_rgb, void *hip_stream);
/* waits for the encode, checks per-chain flags, writes the .alc size of each chunk */
int alice_codec_batch_encode_finish(AliceBatch *batch, uint64_t *sizes /* n_chunks */);
const void *alice_codec_batch_alc_ptr(const AliceBatch *batch, uint32_t chunk); /* device pointer */
uint64_t alice_codec_batch_alc_stride(const AliceBatch *batch); /* valid after alice_codec_batch_encode */
/* copies the n_chunks finished .alc buffers back to back into d_dst (device), in chunk order:
 * the contiguous byte blob a rank contributes to the multi-GPU gather. Asynchronous. */
int alice_codec_batch_pack_alc(AliceBatch *batch, const uint64_t *sizes, void *d_dst, uint64_t dst_capacity,
                               void *hip_stream);
/* d_alc: device pointer, chunk i at d_alc + i*alc_stride (whole .alc, header first).
 * d_rgb_out: device pointer to n_chunks * width*height*frames*3 bytes, or NULL: the pixels of chunk i are then
 * written into the batch's own storage (over the chunk's symbols, which the decode has consumed by then) and
 * are read at alice_codec_batch_rgb_ptr(batch, i) until the next encode/decode on the batch -- saves one
 * RGB-sized buffer per chunk in flight.  Synchronises once to read headers. */
int alice_codec_batch_decode(AliceBatch *batch, const void *d_alc, uint64_t alc_stride, void *d_rgb_out,
                             void *hip_stream);
const void *alice_codec_batch_rgb_ptr(const AliceBatch *batch, uint32_t chunk); /* device pointer, see above */
int alice_codec_batch_decode_finish(AliceBatch *batch);
/* per-stage device times of the last encode+finish / decode+finish, measured with HIP events on the
 * batch's stream: [0] forward transform, [1] table, [2] rANS encode, [3] assemble,
 * [4] rANS decode, [5] inverse transform.  Milliseconds. */
int alice_codec_batch_stage_ms(const AliceBatch *batch, float out[6]);
/* device pointer to the batch's u8 symbols (3 * padded per chunk, channel-major) -- for parity tests */
const void *alice_codec_batch_symbols_ptr(const AliceBatch *batch);
/* Device memory the batch holds per chunk (symbols = decoded pixels, the .alc buffer at its current capacities, tables)
 * and independent of the chunk count (transform scratch): a trial batch of one chunk tells how many chunks the free HBM
 * holds. */
uint64_t alice_codec_batch_bytes_per_chunk(const AliceBatch *batch);
uint64_t alice_codec_batch_fixed_bytes(const AliceBatch *batch);
uint64_t alice_codec_batch_padded_pixels(const AliceBatch *batch);

/* ---- stage level (host pointers; data is staged through the GPU) ---- */
/* Wavelet2D / Wavelet3D forward/inverse (src/wavelet.rs:292-340, 392-484), in place */
int alice_codec_wavelet2d_forward(uint8_t wavelet_type, int32_t *image, uint64_t width, uint64_t height);
int alice_codec_wavelet2d_inverse(uint8_t wavelet_type, int32_t *image, uint64_t width, uint64_t height);
int alice_codec_wavelet3d_forward(uint8_t wavelet_type, int32_t *volume, uint64_t width, uint64_t height, uint64_t depth);
int alice_codec_wavelet3d_inverse(uint8_t wavelet_type, int32_t *volume, uint64_t width, uint64_t height, uint64_t depth);
/* Quantizer::{quantize_buffer,dequantize_buffer} (src/quant.rs:117-146); error if n_out < n_in */
int alice_codec_quantize_buffer(int32_t step, int32_t dead_zone, const int32_t *in, uint64_t n_in, int32_t *out, uint64_t n_out);
int alice_codec_dequantize_buffer(int32_t step, const int32_t *in, uint64_t n_in, int32_t *out, uint64_t n_out);
/* FastQuantizer (src/quant.rs:171-359) */
typedef struct FastQuantizer FastQuantizer;
FastQuantizer *alice_codec_fastquant_new(int32_t step);                              /* NULL if step <= 0 */
FastQuantizer *alice_codec_fastquant_with_dead_zone(int32_t step, int32_t dead_zone);
void alice_codec_fastquant_destroy(FastQuantizer *q);
int32_t alice_codec_fastquant_step(const FastQuantizer *q);
int32_t alice_codec_fastquant_dead_zone(const FastQuantizer *q);
int alice_codec_fastquant_quantize_buffer(const FastQuantizer *q, const int32_t *in, uint64_t n_in, int32_t *out, uint64_t n_out);
int alice_codec_fastquant_dequantize_buffer(const FastQuantizer *q, const int32_t *in, uint64_t n_in, int32_t *out, uint64_t n_out);
/* to_symbols / from_symbols / build_histogram (src/quant.rs:547-600) */
int alice_codec_to_symbols(const int32_t *coeffs, uint64_t n, uint8_t *symbols, uint64_t n_out);
int alice_codec_from_symbols(const uint8_t *symbols, uint64_t n, int32_t *coeffs, uint64_t n_out);
int alice_codec_build_histogram(const uint8_t *symbols, uint64_t n, uint32_t hist[256]);
/* FrequencyTable::from_histogram over 256 bins (src/rans.rs:102-150): writes cum_freq[256], freq[256] */
int alice_codec_freq_table_from_histogram(const uint32_t hist[256], uint16_t cum_freq[256], uint16_t freq[256]);
/* RansEncoder::new().encode_symbols(symbols, table).finish() (src/rans.rs:288-308); table given as arrays.
 * Returns a buffer to free with alice_codec_data_free64, NULL on error. */
uint8_t *alice_codec_rans_encode(const uint8_t *symbols, uint64_t n, const uint16_t cum_freq[256],
                                 const uint16_t freq[256], uint64_t *out_len);
/* RansDecoder::new(bytes).decode_n(n, table) (src/rans.rs:330-381); cum_to_sym is rebuilt from the arrays */
int alice_codec_rans_decode(const uint8_t *bytes, uint64_t len, const uint16_t cum_freq[256],
                            const uint16_t freq[256], uint64_t n, uint8_t *symbols);
/* FrequencyTable::from_histogram(&[u32]) for a slice of n_symbols bins, 1 <= n_symbols <= 256 (src/rans.rs:102-150; an
 * all-zero slice gives FrequencyTable::uniform(n_symbols), :106-109,158-189).  Entries from n_symbols on come back as
 * (0, 0): those symbols do not exist, and encoding one is an error (the reference indexes out of bounds).  n_symbols = 0:
 * ALICE_ERR_REFERENCE_DIVERGES (the reference divides by zero). */
int alice_codec_freq_table_from_histogram_n(const uint32_t *hist, uint32_t n_symbols, uint16_t cum_freq[256], uint16_t freq[256]);
/* RansEncoder as an object that lives across calls (src/rans.rs:238-309): new / with_capacity, encode(&RansSymbol),
 * encode_symbols (any number of calls: each continues the state, so s1 then s2 leaves the stream of one call on
 * s2 || s1), finish (consumes the encoder; buffer to free with alice_codec_data_free64). */
typedef struct AliceRansEncoder AliceRansEncoder;
AliceRansEncoder *alice_codec_rans_encoder_new(void);
void alice_codec_rans_encoder_destroy(AliceRansEncoder *e);
int alice_codec_rans_encoder_encode(AliceRansEncoder *e, uint16_t cum_freq, uint16_t freq);
int alice_codec_rans_encoder_encode_symbols(AliceRansEncoder *e, const uint8_t *symbols, uint64_t n,
                                            const uint16_t cum_freq[256], const uint16_t freq[256]);
uint32_t alice_codec_rans_encoder_state(const AliceRansEncoder *e);
uint8_t *alice_codec_rans_encoder_finish(AliceRansEncoder *e, uint64_t *out_len);
/* RansDecoder as an object (src/rans.rs:321-389): new (copies the input), decode_n (continues from the current state and
 * position; decode() is decode_n(1)), is_empty. */
typedef struct AliceRansDecoder AliceRansDecoder;
AliceRansDecoder *alice_codec_rans_decoder_new(const uint8_t *data, uint64_t len);
void alice_codec_rans_decoder_destroy(AliceRansDecoder *d);
int alice_codec_rans_decoder_decode_n(AliceRansDecoder *d, uint64_t n, const uint16_t cum_freq[256], const uint16_t freq[256],
                                      uint8_t *symbols);
int alice_codec_rans_decoder_is_empty(const AliceRansDecoder *d);
uint32_t alice_codec_rans_decoder_state(const AliceRansDecoder *d);
uint64_t alice_codec_rans_decoder_position(const AliceRansDecoder *d);
/* quantize_subband / dequantize_subband (src/quant.rs:518-545): a sub-band's coefficients through a Quantizer */
int alice_codec_quantize_subband(int32_t step, int32_t dead_zone, const int32_t *coeffs, uint64_t n, int32_t *out, uint64_t n_out);
int alice_codec_dequantize_subband(int32_t step, const int32_t *coeffs, uint64_t n, int32_t *out, uint64_t n_out);
/* ssim / ms_ssim (src/ssim.rs:63-176): mean SSIM over 8x8 blocks of two single-plane images, and the 3-scale
 * variant.  Bit-identical f64 results (block sums are exact, the mean over blocks is folded in raster order).
 * Returns -1.0 on the reference's Err cases (length mismatch), with alice_codec_last_error() set. */
double alice_codec_ssim(const uint8_t *a, uint64_t a_len, const uint8_t *b, uint64_t b_len, uint64_t width, uint64_t height);
double alice_codec_ms_ssim(const uint8_t *a, uint64_t a_len, const uint8_t *b, uint64_t b_len, uint64_t width, uint64_t height);
/* AnalyticalRDO (src/quant.rs:377-505): with_quality's target bits per pixel, and compute_quantizer for one
 * sub-band (0 = LLL .. 7 = HHH, src/lib.rs:115-132) -> step and dead zone of the Quantizer it returns.  The f64
 * sum of squared deviations is accumulated in element order, as the reference does, so the step is identical. */
double alice_codec_rdo_target_bpp(uint8_t quality);
uint8_t alice_codec_subband_quant_strength(uint8_t subband);     /* SubBand3D::quant_strength, src/lib.rs:149-158 */
int alice_codec_rdo_compute_quantizer(double target_bpp, const int32_t *coeffs, uint64_t n, uint8_t subband,
                                      int32_t *step, int32_t *dead_zone);
/* InterleavedRansEncoder::new().encode(symbols, table).finish() (src/rans.rs:393-456): 32-byte header + four
 * independent streams over the sub-sequences i = j mod 4 (an opt-in format, not used by .alc v1).
 * Returns a buffer to free with alice_codec_data_free64, NULL on error. */
uint8_t *alice_codec_rans_encode_interleaved(const uint8_t *symbols, uint64_t n, const uint16_t cum_freq[256],
                                             const uint16_t freq[256], uint64_t *out_len);
/* InterleavedRansDecoder::new(bytes).decode_n(n, table) (src/rans.rs:468-519; SimdRansDecoder, :531-666, reads
 * the same format).  Inputs on which the reference indexes out of bounds or never terminates give an error. */
int alice_codec_rans_decode_interleaved(const uint8_t *bytes, uint64_t len, const uint16_t cum_freq[256],
                                        const uint16_t freq[256], uint64_t n, uint8_t *symbols);
/* rgb_bytes_to_ycocg_r / ycocg_r_to_rgb_bytes (src/color.rs:199-276) */
int alice_codec_rgb_to_ycocg_r(const uint8_t *rgb, uint64_t rgb_len, int16_t *y, int16_t *co, int16_t *cg, uint64_t n_out);
int alice_codec_ycocg_r_to_rgb(const int16_t *y, const int16_t *co, const int16_t *cg, uint64_t n, uint8_t *rgb, uint64_t rgb_len);

/* ---- device-resident stage calls: the pieces of FrameEncoder::encode / FrameDecoder::decode
 * (src/pipeline.rs:429-497, 581-623) on device pointers, for callers that split ONE chunk across GPUs
 * (row slabs) and run the exchanges between the pieces themselves.
 * d_* are device pointers; work runs on hip_stream (hipStream_t, NULL = default) and has finished on return. ---- */
/* colour + pad + Wavelet3D::forward + Quantizer(step(quality), dead zone = step) + to_symbols
 * (src/pipeline.rs:429-470) of a width x height x frames RGB volume -> 3 * padded u8 symbols (Y, Co, Cg; each
 * [pf][ph][pw], low|high halves along every axis).  d_hist: 3*256 u32 or NULL. */
int alice_codec_dev_forward_symbols(const void *d_rgb, uint32_t width, uint32_t height, uint32_t frames,
                                    uint8_t wavelet_type, uint8_t quality, void *d_symbols, void *d_hist,
                                    void *hip_stream);
/* from_symbols + dequantise (steps as stored in the chunk header) + Wavelet3D::inverse + strip + colour
 * (src/pipeline.rs:597-621) */
int alice_codec_dev_inverse_symbols(const void *d_symbols, uint32_t width, uint32_t height, uint32_t frames,
                                    uint8_t wavelet_type, const int32_t step[3], void *d_rgb, void *hip_stream);
/* Wavelet3D::forward / inverse (src/wavelet.rs:392-484) of a device-resident i32 volume [depth][height][width], in place
 * for the caller; d_tmp: scratch of the same size.  Any i32 values (wrapping sums, 64-bit products).  Even width and
 * height >= 6 with an even depth run two tiled passes over the data; other shapes the per-axis kernels. */
int alice_codec_dev_wavelet3d_forward(uint8_t wavelet_type, void *d_volume, void *d_tmp, uint64_t width, uint64_t height,
                                      uint64_t depth, void *hip_stream);
int alice_codec_dev_wavelet3d_inverse(uint8_t wavelet_type, void *d_volume, void *d_tmp, uint64_t width, uint64_t height,
                                      uint64_t depth, void *hip_stream);
/* build_histogram (src/quant.rs:587-600) of n device symbols -> d_hist[256] (u32, device) */
int alice_codec_dev_histogram(const void *d_symbols, uint64_t n, void *d_hist, void *hip_stream);
/* capacity that alice_codec_dev_rans_encode needs for n symbols with this histogram (NULL: worst case) */
uint64_t alice_codec_rans_stream_bound(const uint32_t hist[256], uint64_t n);
/* FrequencyTable::from_histogram(hist) + RansEncoder over n device symbols (src/pipeline.rs:479-484): the
 * stream is written at the END of [d_out, d_out + cap): bytes [*out_offset, *out_offset + *out_len). */
int alice_codec_dev_rans_encode(const void *d_symbols, uint64_t n, const uint32_t hist[256], void *d_out,
                                uint64_t cap, uint64_t *out_offset, uint64_t *out_len, void *hip_stream);
/* RansDecoder::new(stream).decode_n(n, from_histogram(hist)) (src/pipeline.rs:585-594) into device memory */
int alice_codec_dev_rans_decode(const void *d_stream, uint64_t len, const uint32_t hist[256], void *d_symbols,
                                uint64_t n, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* ALICE_CODEC_H */
