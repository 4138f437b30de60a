// Kernel launch interface (host side) for the gfx950 ALICE-Codec path.
#pragma once

#include "common.h"

namespace alice {

constexpr uint32_t kRansOverflow = 4u;   // output region too small (host retries with 2N+4)
constexpr uint32_t kRansInternal = 8u;   // invariant violated (never expected)

struct RansResult {
    unsigned long long len;   // encode: stream bytes; decode: stream bytes consumed
    uint32_t flags;
    uint32_t final_state;
    uint32_t fast_tiles;      // decode: tiles taken by the scalar fast path / by the exact lane loop
    uint32_t slow_tiles;
    // diagnostics (ALICE_CODEC_DEBUG): shader cycles and 100 MHz ticks the chain took (kibi-units), where it ran
    uint32_t cycles_k, ticks_k;
    uint32_t hw_id;           // HW_REG_HW_ID: wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13]
    uint32_t xcc_id;          // HW_REG_XCC_ID [3:0]
    uint32_t paths;           // decode: which branches of the tile loop ran (kDecPath*), for the test-suite's coverage check
    uint32_t pad_;
};

constexpr uint32_t kDecPathDry = 1u;          // stream exhausted: no-window tile
constexpr uint32_t kDecPathWhole = 2u;        // fast tile on a window that lies wholly inside the stream
constexpr uint32_t kDecPathSpecKept = 4u;     // fast tile on a zero-padded window, no padding consumed
constexpr uint32_t kDecPathSpecDropped = 8u;  // ... padding consumed: result dropped, exact loop redid the tile
constexpr uint32_t kDecPathPendingFed = 16u;  // renormalisation owed by an exact-loop symbol settled before a fast tile
constexpr uint32_t kDecPathExact = 32u;       // exact scalar-lane loop
constexpr uint32_t kDecPathStarved = 64u;     // exact loop stopped at the end of its window with bytes still owed
constexpr uint32_t kDecPathTail = 128u;       // last tile of fewer than 4096 symbols
constexpr uint32_t kDecPathUnaligned = 256u;  // output not dword aligned
constexpr uint32_t kDecPathBelowL = 512u;     // fast tile refused: state below 2^23 with no renormalisation owed
constexpr uint32_t kDecPathStillStarved = 1024u;  // fast tile refused: the owed renormalisation could not reach 2^23

struct RansDecodeDesc {
    const uint8_t* in;        // channel stream
    unsigned long long in_len;
    uint8_t* out;             // n symbols
    unsigned long long n;
    const RansTable* table;   // built by rans_table_kernel from the stored channel histogram
    // a RansDecoder object that has decoded before continues from where it stopped (src/rans.rs:351-381): state and
    // position of the next stream byte instead of the four head bytes
    uint32_t resume, x0;
    unsigned long long pos0;
    // where the chain's result goes; null: results[chain] of the launch.  (Chains of several callers merged into one
    // launch -- codec.hip, ChainHub -- report into their callers' own buffers.)
    RansResult* result;
};

// One encode chain of a merged launch (launch_rans_encode_descs): what launch_rans_encode derives from a base, strides and
// the chain number, spelled out per chain.
struct RansEncodeDesc {
    const uint8_t* sym;       // n symbols
    unsigned long long n;
    const RansTable* table;
    uint8_t* region;          // the stream is written back to front into [region, region + cap)
    unsigned long long cap;
    RansResult* result;
    uint32_t x_init;          // kRansL for a fresh encoder; a RansEncoder object between calls brings its state
    uint32_t keep_open;       // 1: leave the four state bytes of finish() unwritten (the object lives on)
};

// ---- rans.hip ----
// n_symbols <= 256: length of the histogram slice (FrequencyTable::from_histogram(&[u32]), src/rans.rs:102-104); bins from
// n_symbols on are ignored and the symbols do not exist in the table (freq 0)
void launch_rans_table(const uint32_t* d_hist, RansTable* d_tables, int n_chains, hipStream_t st, uint32_t n_symbols = 256);
void launch_rans_table_from_arrays(const uint16_t* d_cum, const uint16_t* d_freq, RansTable* d_table,
                                   hipStream_t st);
// chain c reads sym + c*sym_stride (n symbols) and writes its stream back-to-front into a cap-sized region;
// the stream is the last results[c].len bytes of that region.  group_stride == 0: region c starts at
// out + c*cap.  Otherwise chains 3g, 3g+1, 3g+2 write into chunk g's .alc buffer: region start =
// out + g*group_stride + group_head + (c % 3)*cap.  Chains c >= n_split encode n - 1 symbols.
// cap_co / cap_cg (grouped mode only, 0 = same as cap): the second and third chain of a group get regions of their own
// sizes, laid out back to back behind the first (Y streams are about twice as long as Co / Cg streams).
void launch_rans_encode(const uint8_t* d_sym, uint64_t sym_stride, uint64_t n, const RansTable* d_tables,
                        uint8_t* d_out, uint64_t cap, RansResult* d_results, int n_chains, hipStream_t st,
                        uint64_t group_stride = 0, uint64_t group_head = 0, unsigned n_split = 0xFFFFFFFFu,
                        uint64_t cap_co = 0, uint64_t cap_cg = 0, uint32_t x_init = kRansL, bool keep_open = false);
// x_init / keep_open: a RansEncoder object between calls (src/rans.rs:269-294): the chain starts from the object's state
// and leaves the four state bytes of finish() unwritten (results[c].final_state carries the state on)
// out_j[k] = in[4k + j] for the four sub-sequences of an interleaved stream (out_j = out + j*stride)
void launch_split4(const uint8_t* d_in, uint64_t n, uint8_t* d_out, uint64_t stride, hipStream_t st);
// InterleavedRansDecoder::decode_n order (src/rans.rs:501-519): symbol k of stream j lands at
// sum_i min(count_i, k) + #{i < j : count_i > k}; positions >= n_out are dropped.  have[j] symbols of stream j
// are present in d_in (the ones that land below n_out), count[j] is the header's count.
void launch_merge4(const uint8_t* d_in, uint64_t stride, const uint64_t have[4], const uint64_t count[4], uint8_t* d_out,
                   uint64_t n_out, hipStream_t st);
void launch_rans_decode(const RansDecodeDesc* d_descs, RansResult* d_results, int n_chains, hipStream_t st);
// chains described one by one; every desc names its result, its start state and whether the stream is finished
void launch_rans_encode_descs(const RansEncodeDesc* d_descs, int n_chains, hipStream_t st);
// what the runtime reports for the one-chain-per-SIMD instances: out[0..2] = encoder registers per lane (VGPR + AGPR),
// static LDS bytes, workgroups per CU it would co-schedule; out[3..5] = decoder.  False when a query failed.
bool chain_kernel_occupancy(uint32_t out[6]);

// ---- transform.hip (pipeline-specialised: RGB <-> u8 symbols) ----
// A chunk is processed in BANDS of whole tile rows so that a band's intermediate (i16 after the spatial pass, i16 / i32
// after the inverse temporal pass) stays in the Infinity Cache between the two passes, and every launch runs the
// temporal role of one band beside the tile role of its neighbour (transform.hip, "Band-ordered, role-fused launches").
struct BandPlan {
    int tile_h, tiles_y;   // tile height of the tile pass, tile rows of the frame
    int tpb, n_bands;      // tile rows per band, bands (1 = the frame is not cut)
    size_t slot_bytes;     // scratch the launches of one chunk need
};
// false: the shape needs the generic path -- a padded width or height below 6 (the tile kernels read their halo
// through one reflection), a frame of more than 2^30 samples (32-bit offsets inside a frame), or more tiles than a
// 1-D grid holds.
bool transform_tiles_eligible(const ChunkDims& d);
// scratch (one band slot) the forward / inverse launches of a chunk of this shape need; the caller owns the buffer
size_t forward_scratch_bytes(const ChunkDims& d);
size_t inverse_scratch_bytes(const ChunkDims& d, bool mid16);
bool inverse_cuts_chunk(const ChunkDims& d);   // the inverse launches of this shape work band by band (see launch_inverse_transform)
// measurement only (alice_codec_test_transform_ms): the calling thread's next launches of the CDF 9/7 instances run their
// probe twins: 0 off, 1 loads and stores replaced by register moves (the VALU floor), 2 loads only, 3 stores only
void set_transform_probe(int mode);
// target size of a band slot in KiB (0 = never cut; negative = keep).  Process-wide; meant for tests and probes.
void set_transform_tuning(long band_kb);
// radius of the forward temporal kernel's value -> symbol table (clamped to 1 .. 2048, the default).  Process-wide; tests only.
void set_value_table_radius(int r);
// Launches of one chunk on `st`, band after band (tile pass then temporal pass; the inverse the other way round).
// hist: uint32 [3][256], zeroed by the caller.  Returns false (nothing launched) when the shape needs the generic path.
bool launch_forward_transform(const uint8_t* d_rgb, const ChunkDims& d, int wavelet, int32_t step,
                              void* d_scratch, uint8_t* d_sym, uint32_t* d_hist, hipStream_t st);
// steps per channel come from the chunk header.  exact = 64-bit lifting products.  mid16 = the intermediate after the
// temporal pass provably fits i16 (halves its traffic); lds16 = so does everything after the column pass (packed tile).
// When the chunk is cut into bands the pixels of the first band are written while the symbols of later bands are still
// unread: d_rgb must not overlap d_sym then (an uncut chunk may decode over its own symbols).
bool launch_inverse_transform(const uint8_t* d_sym, const ChunkDims& d, int wavelet, const int32_t step[3],
                              bool exact, bool mid16, bool lds16, void* d_scratch, uint8_t* d_rgb, hipStream_t st);

// ---- transform.hip, stage level: Wavelet2D / Wavelet3D of caller-shaped i32 data on the tile kernels' exact instances ----
// eligible: even width and height >= 6, even depth (or depth 1); otherwise the caller uses launch_wavelet_axis.
bool stage_tiles_eligible(uint64_t w, uint64_t h, uint64_t depth, int ndim);
// in place for the caller (result in d_data); d_tmp: same size.  ndim 2: `depth` independent planes.
void launch_stage_wavelet(int32_t* d_data, int32_t* d_tmp, uint64_t w, uint64_t h, uint64_t depth, int ndim, int wavelet,
                          bool inverse, hipStream_t st);

// ---- generic.hip (stage-level API on arbitrary i32 data; exact reference arithmetic) ----
// 1-D transform of n_lines lines: element k of line (a, b) is at data[a*stride_a + b*stride_b + k*stride_k],
// a in [0, n_a), b in [0, n_b).  tmp: same size as data.
void launch_wavelet_axis(int32_t* d_data, int32_t* d_tmp, uint64_t n, uint64_t stride_k, uint64_t n_a,
                         uint64_t stride_a, uint64_t n_b, uint64_t stride_b, int wavelet, bool inverse,
                         hipStream_t st);
void launch_rgb_to_ycocg(const uint8_t* d_rgb, uint64_t n_pixels, int16_t* y, int16_t* co, int16_t* cg, hipStream_t st);
void launch_ycocg_to_rgb(const int16_t* y, const int16_t* co, const int16_t* cg, uint64_t n_pixels, uint8_t* d_rgb, hipStream_t st);
void launch_pad_channel(const int16_t* ch, const ChunkDims& d, int32_t* out, hipStream_t st);
void launch_strip_channel(const int32_t* in, const ChunkDims& d, int16_t* ch, hipStream_t st);
void launch_quantize(const int32_t* in, int32_t* out, uint64_t n, int32_t step, int32_t dead_zone, hipStream_t st);
void launch_fast_quantize(const int32_t* in, int32_t* out, uint64_t n, uint64_t reciprocal, uint32_t shift,
                          int32_t dead_zone, hipStream_t st);
void launch_dequantize(const int32_t* in, int32_t* out, uint64_t n, int32_t step, hipStream_t st);
void launch_to_symbols(const int32_t* in, uint8_t* out, uint64_t n, hipStream_t st);
void launch_from_symbols(const uint8_t* in, int32_t* out, uint64_t n, hipStream_t st);
void launch_histogram(const uint8_t* sym, uint64_t n, uint32_t* hist /*zeroed*/, hipStream_t st);
// ssim (src/ssim.rs): per-8x8-block values in raster order; f64 fold in element order; 2x2 truncating mean
void launch_ssim_blocks(const uint8_t* d_a, const uint8_t* d_b, uint64_t width, uint64_t bw, uint64_t nblocks, double* d_out, hipStream_t st);
void launch_ordered_sum_f64(const double* d_v, uint64_t n, double* d_out, hipStream_t st);
void launch_downsample2(const uint8_t* d_in, uint64_t width, uint64_t height, uint8_t* d_out, hipStream_t st);
// AnalyticalRDO::estimate_variance pieces: exact i64 sum; f64 sum of (x - mean)^2 in element order
void launch_sum_i32(const int32_t* d_x, uint64_t n, unsigned long long* d_sum /*zeroed*/, hipStream_t st);
void launch_ordered_sqdev_sum(const int32_t* d_x, uint64_t n, double mean, double* d_out, hipStream_t st);
void launch_sq_diff_sum(const uint8_t* a, const uint8_t* b, uint64_t n, unsigned long long* d_sum /*zeroed*/, hipStream_t st);
// Each chunk's .alc buffer holds, behind `head` bytes, three cap-sized regions with a stream at the tail of
// each; moves the streams, in place, to directly behind the 3138-byte header slot.
constexpr uint64_t kStreamHead = 3328;   // >= kAlcHeaderBytes, multiple of 256
void launch_compact_streams(uint8_t* d_alc, uint64_t alc_stride, uint64_t head, const uint64_t cap[3],
                            const RansResult* d_results, int n_chunks, hipStream_t st);
// fills the 3138-byte headers on the device (magic, dims, per-channel fields, histograms)
void launch_write_headers(uint8_t* d_alc, uint64_t alc_stride, const ChunkDims& d, int wavelet, int32_t step,
                          const uint32_t* d_hist, const RansResult* d_results, unsigned long long* d_sizes,
                          int n_chunks, hipStream_t st);

}  // namespace alice
