// C++ host-side mirror (include/alice_codec.hpp) exercised the way the reference's own tests use its API:
// src/pipeline.rs:686-693 (roundtrip), :748-761 (serialisation), :786-797 (errors), src/rans.rs:225-236 (doc-test),
// src/wavelet.rs:37-45 (doc-test).  Prints "CPP MIRROR OK" on success.  Links against libalice_codec.so only.
#include <cstdio>
#include <cstdlib>
#include "alice_codec.hpp"
namespace ac = alice_codec;
using ac::CodecError; using ac::WaveletType;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static std::vector<uint8_t> make_gradient(unsigned w, unsigned h, unsigned f) {  // src/pipeline.rs:673-683
    size_t n = (size_t)w * h * f;
    std::vector<uint8_t> rgb(n * 3);
    for (size_t i = 0; i < n; ++i) { uint8_t v = (uint8_t)((i * 7) % 256); rgb[3 * i] = v; rgb[3 * i + 1] = (uint8_t)(v + 30); rgb[3 * i + 2] = (uint8_t)(v + 60); }
    return rgb;
}

int main(int argc, char** argv) {
    const bool have_gpu = alice_codec_device_count() > 0;
    // host-only behaviour first (no device needed)
    try { ac::FrameEncoder::new_(50).encode(std::vector<uint8_t>(10), 4, 4, 2); CHECK(false); }
    catch (const CodecError& e) { CHECK(e.kind == CodecError::InvalidBufferSize); }
    try { ac::FrameEncoder::new_(50).encode(std::vector<uint8_t>(), 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu); CHECK(false); }
    catch (const CodecError& e) { CHECK(e.kind == CodecError::DimensionOverflow); }
    try { ac::EncodedChunk::from_bytes(std::vector<uint8_t>{'A', 'L', 'C', 'C'}); CHECK(false); }
    catch (const CodecError& e) { CHECK(e.kind == CodecError::InvalidBitstream); }
    try { ac::FastQuantizer::new_(0); CHECK(false); }
    catch (const CodecError& e) { CHECK(e.kind == CodecError::InvalidQuantStep); }
    {
        ac::EncodedChunk empty = ac::FrameEncoder::new_(50).encode(std::vector<uint8_t>(), 0, 0, 0);
        CHECK(empty.compressed_size() == 0 && empty.to_bytes().size() == 3138);
        CHECK(ac::FrameDecoder::new_().decode(empty).empty());
    }
    if (!have_gpu) {
        try { ac::FrameEncoder::new_(80).encode(std::vector<uint8_t>(96, 128), 4, 4, 2); CHECK(false); }
        catch (const CodecError& e) { CHECK(e.kind == CodecError::DeviceError); }  // loud failure, no CPU fallback
        std::puts("CPP MIRROR OK (host-only checks; no GPU present)");
        return 0;
    }
    auto rgb = make_gradient(4, 4, 2);
    ac::EncodedChunk chunk = ac::FrameEncoder::new_(90).encode(rgb, 4, 4, 2);
    auto bytes = chunk.to_bytes();
    CHECK(bytes.size() == 3167);                       // SURVEY.md section 8c hand-derived length
    auto decoded = ac::FrameDecoder::new_().decode(chunk);
    CHECK(decoded.size() == rgb.size() && ac::psnr(rgb, decoded) > 15.0);
    ac::EncodedChunk restored = ac::EncodedChunk::from_bytes(bytes);
    CHECK(restored.width() == 4 && restored.height() == 4 && restored.frames() == 2 && restored.wavelet_type() == WaveletType::Cdf53);
    CHECK(ac::FrameDecoder::new_().decode(restored) == decoded);
    ac::EncodedChunk c97 = ac::FrameEncoder::with_wavelet(90, WaveletType::Cdf97).encode(rgb, 4, 4, 2);
    CHECK(c97.wavelet_type() == WaveletType::Cdf97 && c97.to_bytes().size() == 3180);
    // rANS doc-test
    ac::FrequencyTable table = ac::FrequencyTable::uniform();
    ac::RansEncoder enc = ac::RansEncoder::new_();
    enc.encode_symbols({42, 100, 200}, table);
    auto stream = enc.finish();
    CHECK((ac::RansDecoder(stream).decode_n(3, table) == std::vector<uint8_t>{42, 100, 200}));
    // the coders are objects that live across calls (src/rans.rs:269-294, 351-381): s1 then s2 == one call on s2 || s1,
    // single symbols through encode(&RansSymbol), the decoder continues from its position
    {
        ac::RansEncoder one = ac::RansEncoder::new_(), two = ac::RansEncoder::with_capacity(64), three = ac::RansEncoder::new_();
        one.encode_symbols({4, 5, 1, 2, 3}, table);
        two.encode_symbols({1, 2, 3}, table);
        two.encode_symbols({4, 5}, table);
        three.encode_symbols({2, 3}, table);
        three.encode(table.get_symbol(1));
        three.encode_symbols({4, 5}, table);
        auto s1 = one.finish();
        CHECK(s1 == two.finish() && s1 == three.finish());
        ac::RansDecoder d(s1);
        CHECK(!d.is_empty());
        CHECK((d.decode_n(2, table) == std::vector<uint8_t>{4, 5}));
        CHECK(d.decode(table) == 1);
        CHECK((d.decode_n(2, table) == std::vector<uint8_t>{2, 3}));
        ac::FrequencyTable small = ac::FrequencyTable::uniform(2);      // src/rans.rs:934-944
        CHECK(small.len() == 2 && small.freq[0] + small.freq[1] == 4096 && small.freq[2] == 0);
        ac::RansEncoder e2 = ac::RansEncoder::new_();
        e2.encode_symbols({0, 1, 1, 0, 1}, small);
        CHECK((ac::RansDecoder(e2.finish()).decode_n(5, small) == std::vector<uint8_t>{0, 1, 1, 0, 1}));
    }
    // chunk drivers: one device, and the same device listed twice (two host threads)
    {
        std::vector<uint8_t> two = make_gradient(8, 8, 4);
        auto more = make_gradient(8, 8, 4);
        for (auto& v : more) v = (uint8_t)(v * 3 + 1);
        two.insert(two.end(), more.begin(), more.end());
        ac::FrameEncoder e = ac::FrameEncoder::with_wavelet(80, WaveletType::Cdf97);
        auto a1 = ac::encode_many(e, two, 8, 8, 4, 2);
        auto a2 = ac::encode_many(e, two, 8, 8, 4, 2, {0, 0});
        CHECK(a1.size() == 2 && a2.size() == 2 && a1[0].to_bytes() == a2[0].to_bytes() && a1[1].to_bytes() == a2[1].to_bytes());
        CHECK(ac::decode_many(a1) == ac::decode_many(a2, {0}));
        std::vector<int32_t> coeffs = {-100, -7, 0, 7, 100}, qs(5), qb(5), back(5);
        ac::quantize_subband(coeffs, ac::Quantizer::new_(8), qs);
        ac::Quantizer::new_(8).quantize_buffer(coeffs, qb);
        CHECK(qs == qb);
        ac::dequantize_subband(qs, ac::Quantizer::new_(8), back);
        CHECK(back[0] == qs[0] * 8 && back[4] == qs[4] * 8);
    }
    // wavelet doc-test
    std::vector<int32_t> sig = {10, 20, 30, 40, 50, 60, 70, 80};
    ac::Wavelet1D w = ac::Wavelet1D::cdf53();
    w.forward(sig);
    CHECK((sig == std::vector<int32_t>{10, 30, 50, 71, 0, 0, 0, 10}));
    w.inverse(sig);
    CHECK((sig == std::vector<int32_t>{10, 20, 30, 40, 50, 60, 70, 80}));
    // quantiser + symbols
    std::vector<int32_t> vals = {0, 1, -1, 2, -2, 3, -3}, q(7);
    std::vector<uint8_t> sym(7);
    ac::to_symbols(vals, sym);
    CHECK((sym == std::vector<uint8_t>{0, 1, 2, 3, 4, 5, 6}));
    ac::FastQuantizer fq = ac::FastQuantizer::from(ac::Quantizer::with_dead_zone(32, 48));
    CHECK(fq.step() == 32 && fq.dead_zone() == 48);
    std::vector<int32_t> in = {-200, -100, 0, 100, 200}, a(5), b(5);
    fq.quantize_buffer(in, a);
    ac::Quantizer::with_dead_zone(32, 48).quantize_buffer(in, b);
    CHECK(a == b);
    std::puts("CPP MIRROR OK");
    return 0;
}
