// Shared definitions for the MI355X (gfx950) ALICE-Codec path.
// Host + device.  No CUDA compatibility layer: HIP for CDNA4 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace alice {

// CodecError, reference src/error.rs:12-23 (+ library-side conditions)
enum ErrorCode : int {
    kOk = 0,
    kInvalidBufferSize = 1,
    kInvalidDimensions = 2,
    kDimensionOverflow = 3,
    kInvalidBitstream = 4,
    kInvalidQuantStep = 5,
    // Reference would spin forever / divide by zero (src/rans.rs:275-283): a symbol
    // whose table frequency wrapped to 0 is present in the data.
    kReferenceDiverges = 6,
    kOutOfMemory = 7,
    kDeviceError = 8,     // HIP runtime failure / no gfx950 device
    kNullArgument = 9,
    kInternal = 10,
};

// WaveletType, reference src/pipeline.rs:34-41
enum WaveletKind : int { kCdf53 = 0, kCdf97 = 1, kHaar = 2 };

// Lifting coefficient lists, reference src/wavelet.rs:66-127 (scale 2^12).  Every
// filter alternates predict, update, (predict, update).
struct LiftSteps {
    int n;         // 2 or 4
    int coeff[4];
};

inline LiftSteps lift_steps(int kind) {
    LiftSteps s{};
    switch (kind) {
    case kCdf97: s.n = 4; s.coeff[0] = -6497; s.coeff[1] = -217; s.coeff[2] = 3616; s.coeff[3] = 1817; break;
    case kHaar: s.n = 2; s.coeff[0] = -4096; s.coeff[1] = 2048; break;
    default: s.n = 2; s.coeff[0] = -4096; s.coeff[1] = 1024; break;
    }
    return s;
}

// .alc layout constants, reference src/pipeline.rs:137,148
constexpr uint32_t kFixedHeaderBytes = 18;
constexpr uint32_t kChannelHeaderBytes = 1040;
constexpr uint32_t kAlcHeaderBytes = kFixedHeaderBytes + 3 * kChannelHeaderBytes;  // 3138

constexpr uint32_t kProbBits = 12;
constexpr uint32_t kProbScale = 1u << kProbBits;
constexpr uint32_t kRansL = 1u << 23;

// Per-symbol encoder parameters consumed by the rANS encode kernels.
// Built on the device by rans_table_kernel from the channel histogram.
struct RansEncEntry {
    uint32_t xmax;    // freq << 19 (saturated to 2^32-1)
    uint32_t xmax8;   // freq << 27 (saturated)
    uint32_t rcp;     // magic reciprocal: floor(y / freq) = umulhi(y, rcp) >> rsh  (y < xmax)
    uint32_t rsh;
    int32_t g;        // 4096 - freq            (x' = y + q*g + cbias)
    uint32_t cbias;   // cum_freq (+4095 when freq == 1, see rans_table_kernel)
    uint32_t freq;    // as stored in the reference table (u16)
    uint32_t cum;     // as stored in the reference table (u16)
};
static_assert(sizeof(RansEncEntry) == 32, "RansEncEntry must be 32 bytes");

// flags produced by rans_table_kernel
constexpr uint32_t kTableNeedsGeneric = 1u;   // a symbol present in the data has freq > 4096
constexpr uint32_t kTableDiverges = 2u;       // a symbol present in the data has freq == 0
constexpr uint32_t kTableVerified = 16u;      // built from the data's own histogram: the two flags above are authoritative
constexpr uint32_t kTableDecBig = 32u;        // a symbol of frequency 4096 owns slots (its packed decode entry needs x >= 1)
constexpr uint32_t kTableDecExact = 64u;      // a symbol of frequency > 4096 owns slots: the packed decode entries cannot
                                              // express it, the decoder takes its exact loop for every symbol

// Decoder view of a table, one entry per slot (x & 4095), laid out for the chain kernel: the decode update
// x' = freq * (x >> 12) + slot - cum is evaluated as umulhi(F', x) + B' (rans.hip).  The chain wave copies F' and B'
// into 128 VGPRs at the start of every 4096-symbol tile straight from here (L2-resident, 32 KB per chain), so they
// never occupy LDS.
struct RansDecSlots {
    uint32_t ftab[kProbScale];    // F'[slot] = freq << 20         (0xFFFFFFFF for freq 4096)
    uint32_t btab[kProbScale];    // B'[slot] = slot - cum - ((freq * slot) >> 12)
    uint32_t symtab[256];         // freq | cum << 16 per symbol (exact loop)
    uint8_t c2s[kProbScale];      // cum_to_sym, src/rans.rs:135-144
};

struct RansTable {
    RansEncEntry enc[256];
    uint32_t flags;
    uint32_t pad[7];
    RansDecSlots dec;
};
static_assert(sizeof(RansTable) % 32 == 0 && offsetof(RansTable, dec) % 32 == 0, "RansTable layout");

struct ChunkDims {
    uint32_t w, h, f;     // as given
    uint64_t pw, ph, pf;  // padded to even (f == 1 -> 2), reference src/pipeline.rs:437-439 (usize there: 0xFFFFFFFF pads to 2^32)
    uint64_t n_pixels;    // w*h*f
    uint64_t padded;      // pw*ph*pf, saturated at 2^64 - 1 (callers reject anything above the header's u32 num_symbols)
};

inline ChunkDims make_dims(uint32_t w, uint32_t h, uint32_t f) {
    ChunkDims d{};
    d.w = w; d.h = h; d.f = f;
    d.pw = (uint64_t)w + (w & 1u);
    d.ph = (uint64_t)h + (h & 1u);
    d.pf = (f == 1u) ? 2ull : (uint64_t)f + (f & 1u);
    d.n_pixels = (uint64_t)w * h * f;
    const unsigned __int128 p = (unsigned __int128)d.pw * d.ph * d.pf;   // < 2^99
    d.padded = p > (unsigned __int128)UINT64_MAX ? UINT64_MAX : (uint64_t)p;
    return d;
}

// reference src/pipeline.rs:456-457
inline int32_t quality_to_step(uint8_t quality) {
    int32_t q = quality > 100 ? 100 : quality;
    int32_t step = 64 - (q * 63) / 100;
    return step < 1 ? 1 : step;
}

}  // namespace alice
