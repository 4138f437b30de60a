/* TEST AND MEASUREMENT HOOKS of libalice_codec.so.  Not part of the API: nothing here is needed by a caller, nothing
 * here is mirrored by the language bindings, and include/alice_codec.h does not include this file.  The test-suite and
 * the developer probes under scripts/ bind these symbols directly. */
#ifndef ALICE_CODEC_TEST_H
#define ALICE_CODEC_TEST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The next encodes OF THE CALLING THREAD size their stream regions with this capacity instead of the histogram-derived
 * one (0 = off), so that the suite can drive the overflow-and-retry path.  Thread-local: encodes on other threads are
 * not affected, and a test that dies before resetting it leaves no process-wide state behind. */
void alice_codec_test_force_first_cap(uint64_t cap);

/* The last alice_codec_rans_decode / alice_codec_dev_rans_decode of the calling thread: tiles taken by the fast path,
 * tiles taken by the exact loop, the mask of tile-loop branches that ran (kDecPath* in csrc/kernels.h), stream bytes
 * consumed. */
void alice_codec_test_last_decode_stats(uint32_t out[4]);

/* Times the transform launches alone (no chains) with HIP events on `hip_stream`: `reps` passes over `n_chunks` chunks
 * of w x h x f pixels, forward (RGB -> symbols + histograms) and inverse (symbols -> RGB), through the same pipes the
 * encode / decode of a batch use.  Device buffers: d_rgb and d_rgb_out hold n_buffers chunks of RGB, d_sym n_buffers
 * chunks of 3 * padded symbols; chunk c uses buffer c mod n_buffers.  out_ms[0] / out_ms[1] = milliseconds per chunk
 * forward / inverse.  probe: 0 = the real kernels; 1 = the VALU-floor probe (same instruction streams, same registers
 * and LDS, global loads and stores replaced by register moves; the outputs are NOT produced); 2 = only the loads
 * replaced; 3 = only the stores replaced (CDF 9/7 with a step > 1 and the i16 lane-exchange inverse only). */
int alice_codec_test_transform_ms(const void *d_rgb, void *d_sym, void *d_rgb_out, uint32_t n_buffers, uint32_t width,
                                  uint32_t height, uint32_t frames, uint8_t wavelet_type, uint8_t quality, uint32_t n_chunks,
                                  uint32_t reps, int probe, float out_ms[2], void *hip_stream);

/* Band plan of the transform launches (csrc/transform.hip, "Bands"), process-wide: target size of a band slot in KiB
 * (0 = never cut a chunk into bands; negative = keep; default 1048576).  The suite uses it to run small shapes through
 * many bands; results never depend on it. */
void alice_codec_test_set_tuning(long band_kb);

/* Radius r of the value -> symbol table of the forward temporal kernel (csrc/transform.hip, kQLutR), process-wide, clamped
 * to 1 .. 2048 (the default): coefficients in [-r, r) are quantised through the table, a wavefront holding any other value
 * through the arithmetic.  8-bit RGB never leaves the default table, so the suite shrinks it to run the arithmetic path
 * and the boundary; results never depend on it. */
void alice_codec_test_set_value_table_radius(int r);

/* Admission budget of the calling thread's device (csrc/codec.hip, ChainHub::admit): whole-chunk host calls state the
 * device memory they are about to allocate and wait while the calls already in flight hold more than the budget allows
 * (a call alone always enters).  Default: 90 % of what is free (device + the library's cache) whenever a call enters an
 * idle hub; 0 restores that.  The suite
 * shrinks it so that a handful of small calls already queue; results never depend on it. */
int alice_codec_test_set_admission_budget(uint64_t bytes);

/* Resident chain kernels: what the runtime reports for the one-chain-per-SIMD instances of the rANS kernels.
 * out[0..2] = encoder: registers per lane (VGPR + AGPR, as allocated), static LDS bytes, workgroups per CU the runtime
 * would co-schedule; out[3..5] = the same for the decoder.  The exclusive instances must report at most 4 workgroups
 * per CU (one wave per SIMD). */
int alice_codec_test_chain_occupancy(uint32_t out[6]);

#ifdef __cplusplus
}
#endif
#endif
