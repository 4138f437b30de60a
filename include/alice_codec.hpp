// alice_codec.hpp -- header-only C++17 mirror of the reference's Rust API over the C ABI (alice_codec.h).
//
// The reference is a Rust crate; no Rust toolchain exists in the build image, so the host side above the
// C ABI is C++ with the reference's names, argument meaning and error behaviour (Result<T, CodecError>
// becomes a thrown alice_codec::CodecError carrying the same variant):
//   FrameEncoder::{new_, with_wavelet, encode}      src/pipeline.rs:335-507
//   FrameDecoder::{decode}                          src/pipeline.rs:519-631
//   EncodedChunk::{to_bytes, from_bytes, ...}       src/pipeline.rs:172-313
//   Wavelet1D / Wavelet2D / Wavelet3D               src/wavelet.rs:47-485
//   Quantizer / FastQuantizer                       src/quant.rs:57-359
//   to_symbols / from_symbols / build_histogram     src/quant.rs:547-600
//   FrequencyTable / RansEncoder / RansDecoder      src/rans.rs:85-389
// Everything executes on the GPU through libalice_codec.so; there is no CPU fallback.
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "alice_codec.h"

namespace alice_codec {

enum class WaveletType : uint8_t { Cdf53 = 0, Cdf97 = 1, Haar = 2 };  // src/pipeline.rs:34-41

// src/error.rs:12-23 (+ library-side conditions)
struct CodecError : std::runtime_error {
    enum Kind { InvalidBufferSize = 1, InvalidDimensions, DimensionOverflow, InvalidBitstream, InvalidQuantStep,
                ReferenceDiverges, OutOfMemory, DeviceError, NullArgument, Internal };
    Kind kind;
    CodecError(int code, const std::string& msg) : std::runtime_error(msg), kind(static_cast<Kind>(code)) {}
};

namespace detail {
[[noreturn]] inline void raise(int fallback = ALICE_ERR_INTERNAL) {
    int code = alice_codec_last_error();
    const char* m = alice_codec_last_error_message();
    throw CodecError(code ? code : fallback, m ? m : "");
}
inline void check(int rc) { if (rc != ALICE_OK) raise(rc); }
inline std::vector<uint8_t> take(uint8_t* p, uint64_t n) {
    std::vector<uint8_t> v(p, p + n);
    alice_codec_data_free64(p, n);
    return v;
}
}  // namespace detail

class EncodedChunk {
public:
    explicit EncodedChunk(::EncodedChunk* h) : h_(h) {}
    EncodedChunk(EncodedChunk&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    EncodedChunk& operator=(EncodedChunk&& o) noexcept { reset(); h_ = std::exchange(o.h_, nullptr); return *this; }
    EncodedChunk(const EncodedChunk&) = delete;
    EncodedChunk& operator=(const EncodedChunk&) = delete;
    ~EncodedChunk() { reset(); }
    uint32_t width() const { return alice_codec_chunk_width(h_); }
    uint32_t height() const { return alice_codec_chunk_height(h_); }
    uint32_t frames() const { return alice_codec_chunk_frames(h_); }
    WaveletType wavelet_type() const { return static_cast<WaveletType>(alice_codec_chunk_wavelet(h_)); }
    size_t compressed_size() const { return alice_codec_chunk_compressed_size(h_); }
    std::vector<uint8_t> to_bytes() const {
        uint64_t n = 0;
        uint8_t* p = alice_codec_chunk_to_bytes64(h_, &n);
        if (!p) detail::raise();
        return detail::take(p, n);
    }
    static EncodedChunk from_bytes(const uint8_t* data, size_t len) {
        static const uint8_t empty = 0;
        ::EncodedChunk* h = alice_codec_chunk_from_bytes64(data ? data : &empty, len);
        if (!h) detail::raise(ALICE_ERR_INVALID_BITSTREAM);
        return EncodedChunk(h);
    }
    static EncodedChunk from_bytes(const std::vector<uint8_t>& v) { return from_bytes(v.data(), v.size()); }
    const ::EncodedChunk* handle() const { return h_; }
    static EncodedChunk adopt(::EncodedChunk* h) { return EncodedChunk(h); }   // takes ownership of a handle from the C ABI
private:
    void reset() { if (h_) alice_codec_chunk_destroy(h_); h_ = nullptr; }
    ::EncodedChunk* h_;
};

class FrameEncoder {
public:
    static FrameEncoder new_(uint8_t quality) { return FrameEncoder(quality, WaveletType::Cdf53); }
    static FrameEncoder with_wavelet(uint8_t quality, WaveletType w) { return FrameEncoder(quality, w); }
    FrameEncoder(FrameEncoder&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    FrameEncoder(const FrameEncoder&) = delete;
    ~FrameEncoder() { if (h_) alice_codec_encoder_destroy(h_); }
    EncodedChunk encode(const uint8_t* rgb, size_t len, uint32_t width, uint32_t height, uint32_t frames) const {
        static const uint8_t empty = 0;
        ::EncodedChunk* c = alice_codec_encode64(h_, rgb ? rgb : &empty, len, width, height, frames);
        if (!c) detail::raise();
        return EncodedChunk(c);
    }
    EncodedChunk encode(const std::vector<uint8_t>& rgb, uint32_t w, uint32_t h, uint32_t f) const {
        return encode(rgb.data(), rgb.size(), w, h, f);
    }
    const ::FrameEncoder* handle() const { return h_; }
private:
    FrameEncoder(uint8_t q, WaveletType w) : h_(alice_codec_encoder_create_ex(q, static_cast<uint8_t>(w))) { if (!h_) detail::raise(); }
    ::FrameEncoder* h_;
};

struct FrameDecoder {
    static FrameDecoder new_() { return {}; }
    std::vector<uint8_t> decode(const EncodedChunk& chunk) const {
        uint64_t n = 0;
        uint8_t* p = alice_codec_decode64(chunk.handle(), &n);
        if (!p) detail::raise();
        return detail::take(p, n);
    }
};

// Many equal-shaped chunks in one call (the 64-frame chunk driver, src/pipeline.rs:461-497).  devices empty: the calling
// thread's device; otherwise chunk k runs on devices[k mod n], one host thread per entry of the list.
inline std::vector<EncodedChunk> encode_many(const FrameEncoder& enc, const std::vector<uint8_t>& rgb, uint32_t w, uint32_t h, uint32_t f,
                                             uint32_t n_chunks, const std::vector<int>& devices = {}) {
    std::vector<::EncodedChunk*> raw(n_chunks, nullptr);
    static const uint8_t empty = 0;
    const uint8_t* p = rgb.empty() ? &empty : rgb.data();
    if (devices.empty()) detail::check(alice_codec_encode_many(enc.handle(), p, rgb.size(), w, h, f, n_chunks, raw.data()));
    else detail::check(alice_codec_encode_many_devices(enc.handle(), p, rgb.size(), w, h, f, n_chunks, devices.data(),
                                                       static_cast<uint32_t>(devices.size()), raw.data()));
    std::vector<EncodedChunk> out;
    out.reserve(n_chunks);
    for (auto* c : raw) out.push_back(EncodedChunk::adopt(c));
    return out;
}
inline std::vector<uint8_t> decode_many(const std::vector<EncodedChunk>& chunks, const std::vector<int>& devices = {}) {
    if (chunks.empty()) return {};
    std::vector<const ::EncodedChunk*> raw;
    for (const auto& c : chunks) raw.push_back(c.handle());
    std::vector<uint8_t> out((size_t)chunks[0].width() * chunks[0].height() * chunks[0].frames() * 3 * chunks.size());
    static uint8_t sink = 0;
    uint8_t* p = out.empty() ? &sink : out.data();
    if (devices.empty()) detail::check(alice_codec_decode_many(raw.data(), static_cast<uint32_t>(raw.size()), p, out.size()));
    else detail::check(alice_codec_decode_many_devices(raw.data(), static_cast<uint32_t>(raw.size()), devices.data(),
                                                       static_cast<uint32_t>(devices.size()), p, out.size()));
    return out;
}

class Wavelet1D {
public:
    static Wavelet1D cdf97() { return Wavelet1D(alice_codec_wavelet1d_cdf97()); }
    static Wavelet1D cdf53() { return Wavelet1D(alice_codec_wavelet1d_cdf53()); }
    static Wavelet1D haar() { return Wavelet1D(alice_codec_wavelet1d_haar()); }
    Wavelet1D(Wavelet1D&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    Wavelet1D(const Wavelet1D&) = delete;
    ~Wavelet1D() { if (h_) alice_codec_wavelet1d_destroy(h_); }
    void forward(std::vector<int32_t>& s) const { run(s, alice_codec_wavelet1d_forward); }
    void inverse(std::vector<int32_t>& s) const { run(s, alice_codec_wavelet1d_inverse); }
private:
    explicit Wavelet1D(::Wavelet1D* h) : h_(h) {}
    template <typename F> void run(std::vector<int32_t>& s, F fn) const {
        if (s.empty()) return;
        fn(h_, s.data(), static_cast<uint32_t>(s.size()));
        if (alice_codec_last_error()) detail::raise();
    }
    ::Wavelet1D* h_;
};

struct Wavelet2D {
    WaveletType kind = WaveletType::Cdf53;
    static Wavelet2D cdf97() { return {WaveletType::Cdf97}; }
    static Wavelet2D cdf53() { return {WaveletType::Cdf53}; }
    void forward(std::vector<int32_t>& img, size_t w, size_t h) const { detail::check(alice_codec_wavelet2d_forward((uint8_t)kind, img.data(), w, h)); }
    void inverse(std::vector<int32_t>& img, size_t w, size_t h) const { detail::check(alice_codec_wavelet2d_inverse((uint8_t)kind, img.data(), w, h)); }
};

struct Wavelet3D {
    WaveletType kind = WaveletType::Cdf53;
    static Wavelet3D cdf97() { return {WaveletType::Cdf97}; }
    static Wavelet3D cdf53() { return {WaveletType::Cdf53}; }
    void forward(std::vector<int32_t>& v, size_t w, size_t h, size_t d) const { detail::check(alice_codec_wavelet3d_forward((uint8_t)kind, v.data(), w, h, d)); }
    void inverse(std::vector<int32_t>& v, size_t w, size_t h, size_t d) const { detail::check(alice_codec_wavelet3d_inverse((uint8_t)kind, v.data(), w, h, d)); }
};

struct Quantizer {  // src/quant.rs:57-153
    int32_t step, dead_zone;
    static Quantizer new_(int32_t step) { return {step, step}; }
    static Quantizer with_dead_zone(int32_t step, int32_t dz) { return {step, dz}; }
    void quantize_buffer(const std::vector<int32_t>& in, std::vector<int32_t>& out) const {
        detail::check(alice_codec_quantize_buffer(step, dead_zone, in.data(), in.size(), out.data(), out.size()));
    }
    void dequantize_buffer(const std::vector<int32_t>& in, std::vector<int32_t>& out) const {
        detail::check(alice_codec_dequantize_buffer(step, in.data(), in.size(), out.data(), out.size()));
    }
};

enum class SubBand3D : uint8_t { LLL = 0, LLH, LHL, LHH, HLL, HLH, HHL, HHH };  // src/lib.rs:115-132
inline bool is_temporal_high(SubBand3D s) { return (static_cast<uint8_t>(s) & 1u) != 0; }          // LLH, LHH, HLH, HHH
inline bool is_dc(SubBand3D s) { return s == SubBand3D::LLL; }
inline uint8_t quant_strength(SubBand3D s) { return alice_codec_subband_quant_strength(static_cast<uint8_t>(s)); }

class AnalyticalRDO {  // src/quant.rs:377-505
public:
    static AnalyticalRDO new_(double target_bpp) { return AnalyticalRDO(target_bpp, 75); }
    static AnalyticalRDO with_quality(uint8_t quality) {
        const uint8_t q = quality > 100 ? 100 : quality;
        return AnalyticalRDO(alice_codec_rdo_target_bpp(q), q);
    }
    Quantizer compute_quantizer(const std::vector<int32_t>& coeffs, SubBand3D subband) const {
        int32_t step = 1, dz = 1;
        static const int32_t empty = 0;
        detail::check(alice_codec_rdo_compute_quantizer(target_bpp_, coeffs.empty() ? &empty : coeffs.data(), coeffs.size(),
                                                        static_cast<uint8_t>(subband), &step, &dz));
        return Quantizer::with_dead_zone(step, dz);
    }
    std::array<Quantizer, 8> compute_all_quantizers(const std::array<std::vector<int32_t>, 8>& subbands) const {
        std::array<Quantizer, 8> q{};
        for (size_t i = 0; i < 8; ++i) q[i] = compute_quantizer(subbands[i], static_cast<SubBand3D>(i));
        return q;
    }
    uint8_t quality() const { return quality_; }
    double target_bpp() const { return target_bpp_; }
private:
    AnalyticalRDO(double bpp, uint8_t q) : target_bpp_(bpp), quality_(q) {}
    double target_bpp_;
    uint8_t quality_;
};

class FastQuantizer {  // src/quant.rs:171-359
public:
    static FastQuantizer new_(int32_t step) { return FastQuantizer(alice_codec_fastquant_new(step)); }
    static FastQuantizer with_dead_zone(int32_t step, int32_t dz) { return FastQuantizer(alice_codec_fastquant_with_dead_zone(step, dz)); }
    static FastQuantizer from(const Quantizer& q) { return with_dead_zone(q.step, q.dead_zone); }
    FastQuantizer(FastQuantizer&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    FastQuantizer(const FastQuantizer&) = delete;
    ~FastQuantizer() { if (h_) alice_codec_fastquant_destroy(h_); }
    int32_t step() const { return alice_codec_fastquant_step(h_); }
    int32_t dead_zone() const { return alice_codec_fastquant_dead_zone(h_); }
    void quantize_buffer(const std::vector<int32_t>& in, std::vector<int32_t>& out) const {
        detail::check(alice_codec_fastquant_quantize_buffer(h_, in.data(), in.size(), out.data(), out.size()));
    }
    void quantize_buffer_simd(const std::vector<int32_t>& in, std::vector<int32_t>& out) const { quantize_buffer(in, out); }
    void dequantize_buffer(const std::vector<int32_t>& in, std::vector<int32_t>& out) const {
        detail::check(alice_codec_fastquant_dequantize_buffer(h_, in.data(), in.size(), out.data(), out.size()));
    }
private:
    explicit FastQuantizer(::FastQuantizer* h) : h_(h) { if (!h_) detail::raise(ALICE_ERR_INVALID_QUANT_STEP); }
    ::FastQuantizer* h_;
};

// quantize_subband / dequantize_subband (src/quant.rs:518-545)
inline void quantize_subband(const std::vector<int32_t>& coeffs, const Quantizer& q, std::vector<int32_t>& out) {
    detail::check(alice_codec_quantize_subband(q.step, q.dead_zone, coeffs.data(), coeffs.size(), out.data(), out.size()));
}
inline void dequantize_subband(const std::vector<int32_t>& coeffs, const Quantizer& q, std::vector<int32_t>& out) {
    detail::check(alice_codec_dequantize_subband(q.step, coeffs.data(), coeffs.size(), out.data(), out.size()));
}

inline void to_symbols(const std::vector<int32_t>& coeffs, std::vector<uint8_t>& symbols) {
    detail::check(alice_codec_to_symbols(coeffs.data(), coeffs.size(), symbols.data(), symbols.size()));
}
inline void from_symbols(const std::vector<uint8_t>& symbols, std::vector<int32_t>& coeffs) {
    detail::check(alice_codec_from_symbols(symbols.data(), symbols.size(), coeffs.data(), coeffs.size()));
}
inline std::array<uint32_t, 256> build_histogram(const std::vector<uint8_t>& symbols) {
    std::array<uint32_t, 256> h{};
    static const uint8_t empty = 0;
    detail::check(alice_codec_build_histogram(symbols.empty() ? &empty : symbols.data(), symbols.size(), h.data()));
    return h;
}

struct RansSymbol {  // src/rans.rs:59-72
    uint16_t cum_freq = 0, freq = 0;
};

struct FrequencyTable {  // src/rans.rs:85-219; n symbols, 1 <= n <= 256 (entries from n on are (0, 0))
    std::array<uint16_t, 256> cum_freq{}, freq{};
    size_t n_symbols = 256;
    static FrequencyTable from_histogram(const std::vector<uint32_t>& hist) {      // any slice length (src/rans.rs:102-104)
        FrequencyTable t;
        static const uint32_t empty = 0;
        detail::check(alice_codec_freq_table_from_histogram_n(hist.empty() ? &empty : hist.data(), static_cast<uint32_t>(hist.size()),
                                                              t.cum_freq.data(), t.freq.data()));
        t.n_symbols = hist.size();
        return t;
    }
    static FrequencyTable from_histogram(const std::array<uint32_t, 256>& hist) {
        FrequencyTable t;
        detail::check(alice_codec_freq_table_from_histogram(hist.data(), t.cum_freq.data(), t.freq.data()));
        return t;
    }
    static FrequencyTable uniform(size_t n = 256) { return from_histogram(std::vector<uint32_t>(n, 0u)); }   // :158-189
    RansSymbol get_symbol(uint8_t sym) const {                                                               // :194
        if (sym >= n_symbols) detail::raise(ALICE_ERR_INVALID_DIMENSIONS);   // the reference panics
        return RansSymbol{cum_freq[sym], freq[sym]};
    }
    size_t len() const { return n_symbols; }
    bool is_empty() const { return n_symbols == 0; }
};

class RansEncoder {  // src/rans.rs:238-309: lives across calls; encode / encode_symbols continue one state
public:
    static RansEncoder new_() { return RansEncoder(); }
    static RansEncoder with_capacity(size_t) { return RansEncoder(); }   // a hint in the reference too
    RansEncoder() : h_(alice_codec_rans_encoder_new()) { if (!h_) detail::raise(); }
    RansEncoder(RansEncoder&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    RansEncoder(const RansEncoder&) = delete;
    ~RansEncoder() { if (h_) alice_codec_rans_encoder_destroy(h_); }
    void encode(const RansSymbol& sym) { detail::check(alice_codec_rans_encoder_encode(h_, sym.cum_freq, sym.freq)); }          // :269-285
    void encode_symbols(const std::vector<uint8_t>& symbols, const FrequencyTable& table) {                                    // :288-294
        if (!symbols.empty())
            detail::check(alice_codec_rans_encoder_encode_symbols(h_, symbols.data(), symbols.size(), table.cum_freq.data(), table.freq.data()));
    }
    std::vector<uint8_t> finish() {   // :298-308, consumes the encoder
        uint64_t n = 0;
        uint8_t* p = alice_codec_rans_encoder_finish(std::exchange(h_, nullptr), &n);
        if (!p) detail::raise();
        return detail::take(p, n);
    }
private:
    AliceRansEncoder* h_;
};

class RansDecoder {  // src/rans.rs:321-389: decode / decode_n continue from the current position
public:
    explicit RansDecoder(const std::vector<uint8_t>& input) {
        static const uint8_t empty = 0;
        h_ = alice_codec_rans_decoder_new(input.empty() ? &empty : input.data(), input.size());
        if (!h_) detail::raise();
    }
    RansDecoder(RansDecoder&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    RansDecoder(const RansDecoder&) = delete;
    ~RansDecoder() { if (h_) alice_codec_rans_decoder_destroy(h_); }
    std::vector<uint8_t> decode_n(size_t n, const FrequencyTable& table) {                                                    // :375-381
        std::vector<uint8_t> out(n);
        uint8_t sink = 0;
        detail::check(alice_codec_rans_decoder_decode_n(h_, n, table.cum_freq.data(), table.freq.data(), n ? out.data() : &sink));
        return out;
    }
    uint8_t decode(const FrequencyTable& table) { return decode_n(1, table)[0]; }                                             // :351-371
    bool is_empty() const { return alice_codec_rans_decoder_is_empty(h_) != 0; }                                             // :385-389
private:
    AliceRansDecoder* h_ = nullptr;
};

class InterleavedRansEncoder {  // src/rans.rs:393-456 (opt-in 4-stream format)
public:
    static InterleavedRansEncoder new_() { return {}; }
    void encode(const std::vector<uint8_t>& symbols, const FrequencyTable& table) { sym_ = symbols; table_ = table; }
    std::vector<uint8_t> finish() {
        uint64_t n = 0;
        static const uint8_t empty = 0;
        uint8_t* p = alice_codec_rans_encode_interleaved(sym_.empty() ? &empty : sym_.data(), sym_.size(), table_.cum_freq.data(),
                                                         table_.freq.data(), &n);
        if (!p) detail::raise();
        return detail::take(p, n);
    }
private:
    std::vector<uint8_t> sym_;
    FrequencyTable table_ = FrequencyTable{};
};

class InterleavedRansDecoder {  // src/rans.rs:468-519
public:
    explicit InterleavedRansDecoder(std::vector<uint8_t> input) : in_(std::move(input)) {}
    std::vector<uint8_t> decode_n(size_t n, const FrequencyTable& table) const {
        std::vector<uint8_t> out(n);
        static const uint8_t empty = 0;
        uint8_t sink = 0;
        detail::check(alice_codec_rans_decode_interleaved(in_.empty() ? &empty : in_.data(), in_.size(), table.cum_freq.data(),
                                                          table.freq.data(), n, n ? out.data() : &sink));
        return out;
    }
private:
    std::vector<uint8_t> in_;
};
using SimdRansDecoder = InterleavedRansDecoder;  // src/rans.rs:531-666: same format, same symbols

inline double psnr(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b) {
    if (a.size() != b.size()) return -1.0;
    static const uint8_t empty = 0;
    return alice_codec_psnr(a.empty() ? &empty : a.data(), b.empty() ? &empty : b.data(), static_cast<uint32_t>(a.size()));
}

inline double ssim(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b, size_t width, size_t height) {  // src/ssim.rs:63
    static const uint8_t empty = 0;
    const double v = alice_codec_ssim(a.empty() ? &empty : a.data(), a.size(), b.empty() ? &empty : b.data(), b.size(), width, height);
    if (v == -1.0 && alice_codec_last_error() != 0) detail::raise();
    return v;
}
inline double ms_ssim(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b, size_t width, size_t height) {  // src/ssim.rs:125
    static const uint8_t empty = 0;
    const double v = alice_codec_ms_ssim(a.empty() ? &empty : a.data(), a.size(), b.empty() ? &empty : b.data(), b.size(), width, height);
    if (v == -1.0 && alice_codec_last_error() != 0) detail::raise();
    return v;
}

}  // namespace alice_codec
