//! MI355X back end for ALICE-Codec's hot path, as the reference's own Rust API.
//!
//! NOT COMPILED, NOT TESTED: the image this repository is built in has no Rust toolchain (no `cargo`, no `rustc`),
//! and neither has the GPU box.  What is verified is the C ABI underneath (`include/alice_codec.h`, exercised through
//! ctypes by `tests/` and through the C++ mirror `include/alice_codec.hpp` by `tests/cpp/test_cpp_mirror.cpp`).  This
//! file is the binding a maintainer of the reference crate would add -- `src/gpu_backend.rs` behind a cargo feature --
//! and states, type by type, which reference item each wrapper replaces.  Names, argument order, argument meaning and
//! error variants are the reference's:
//!
//!   FrameEncoder::{new, with_wavelet, encode}          src/pipeline.rs:347, 356, 377
//!   FrameDecoder::{new, decode}                        src/pipeline.rs:524, 537
//!   EncodedChunk::{compressed_size, to_bytes, from_bytes}   src/pipeline.rs:190, 200, 235
//!   Wavelet3D::{new, cdf97, cdf53, forward, inverse}   src/wavelet.rs:366, 372, 378, 392, 441
//!   Wavelet2D::{cdf97, cdf53, forward, inverse}        src/wavelet.rs:278, 284, 292, 319
//!   FastQuantizer::{new, with_dead_zone, quantize, dequantize, quantize_buffer, dequantize_buffer,
//!                   quantize_buffer_simd, step, dead_zone}   src/quant.rs:190-347
//!   to_symbols / from_symbols / build_histogram        src/quant.rs:547, 572, 594
//!   FrequencyTable::{from_histogram, uniform, get_symbol, len, is_empty}   src/rans.rs:102, 158, 194, 209, 216 (any n <= 256)
//!   RansEncoder::{new, with_capacity, encode, encode_symbols, finish}       src/rans.rs:249, 258, 269, 288, 298
//!   RansDecoder::{new, decode, decode_n, is_empty}     src/rans.rs:330, 351, 375, 385
//!   quantize_subband / dequantize_subband              src/quant.rs:518, 531
//!   encode_many / decode_many (chunk driver, optionally over several GPUs)   src/pipeline.rs:461-497
//!
//! Every call computes on the GPU; without an MI355X the library returns an error (`CodecError::Device`), it never
//! falls back to the CPU.  Results are byte-identical to the reference's (this repository's parity tests), with one
//! documented divergence: encoding a symbol whose table frequency wrapped to 0 returns `CodecError::ReferenceDiverges`
//! where the reference never returns (src/rans.rs:275-283).
#![allow(dead_code)]

use std::os::raw::{c_char, c_int};

// ---------------------------------------------------------------------------------------------------------------
// the C ABI (include/alice_codec.h); opaque handles
// ---------------------------------------------------------------------------------------------------------------

#[repr(C)] pub struct RawEncoder { _p: [u8; 0] }
#[repr(C)] pub struct RawChunk { _p: [u8; 0] }
#[repr(C)] pub struct RawFastQuantizer { _p: [u8; 0] }
#[repr(C)] pub struct RawRansEncoder { _p: [u8; 0] }
#[repr(C)] pub struct RawRansDecoder { _p: [u8; 0] }

#[link(name = "alice_codec")]
extern "C" {
    fn alice_codec_last_error() -> c_int;
    fn alice_codec_last_error_message() -> *const c_char;

    fn alice_codec_encoder_create_ex(quality: u8, wavelet_type: u8) -> *mut RawEncoder;
    fn alice_codec_encoder_destroy(p: *mut RawEncoder);
    fn alice_codec_encode64(e: *const RawEncoder, rgb: *const u8, rgb_len: u64, w: u32, h: u32, f: u32) -> *mut RawChunk;
    fn alice_codec_decode64(c: *const RawChunk, out_len: *mut u64) -> *mut u8;
    fn alice_codec_chunk_destroy(c: *mut RawChunk);
    fn alice_codec_chunk_to_bytes64(c: *const RawChunk, out_len: *mut u64) -> *mut u8;
    fn alice_codec_chunk_from_bytes64(data: *const u8, len: u64) -> *mut RawChunk;
    fn alice_codec_chunk_width(c: *const RawChunk) -> u32;
    fn alice_codec_chunk_height(c: *const RawChunk) -> u32;
    fn alice_codec_chunk_frames(c: *const RawChunk) -> u32;
    fn alice_codec_chunk_wavelet(c: *const RawChunk) -> u8;
    fn alice_codec_chunk_compressed_size(c: *const RawChunk) -> u64;
    fn alice_codec_data_free64(p: *mut u8, len: u64);

    fn alice_codec_wavelet2d_forward(wavelet: u8, image: *mut i32, w: u64, h: u64) -> c_int;
    fn alice_codec_wavelet2d_inverse(wavelet: u8, image: *mut i32, w: u64, h: u64) -> c_int;
    fn alice_codec_wavelet3d_forward(wavelet: u8, volume: *mut i32, w: u64, h: u64, d: u64) -> c_int;
    fn alice_codec_wavelet3d_inverse(wavelet: u8, volume: *mut i32, w: u64, h: u64, d: u64) -> c_int;

    fn alice_codec_fastquant_new(step: i32) -> *mut RawFastQuantizer;
    fn alice_codec_fastquant_with_dead_zone(step: i32, dead_zone: i32) -> *mut RawFastQuantizer;
    fn alice_codec_fastquant_destroy(q: *mut RawFastQuantizer);
    fn alice_codec_fastquant_quantize_buffer(q: *const RawFastQuantizer, input: *const i32, n_in: u64, out: *mut i32, n_out: u64) -> c_int;
    fn alice_codec_fastquant_dequantize_buffer(q: *const RawFastQuantizer, input: *const i32, n_in: u64, out: *mut i32, n_out: u64) -> c_int;

    fn alice_codec_to_symbols(coeffs: *const i32, n: u64, symbols: *mut u8, n_out: u64) -> c_int;
    fn alice_codec_from_symbols(symbols: *const u8, n: u64, coeffs: *mut i32, n_out: u64) -> c_int;
    fn alice_codec_build_histogram(symbols: *const u8, n: u64, hist: *mut u32) -> c_int;
    fn alice_codec_freq_table_from_histogram(hist: *const u32, cum_freq: *mut u16, freq: *mut u16) -> c_int;
    fn alice_codec_freq_table_from_histogram_n(hist: *const u32, n_symbols: u32, cum_freq: *mut u16, freq: *mut u16) -> c_int;
    fn alice_codec_rans_encode(symbols: *const u8, n: u64, cum_freq: *const u16, freq: *const u16, out_len: *mut u64) -> *mut u8;
    fn alice_codec_rans_decode(bytes: *const u8, len: u64, cum_freq: *const u16, freq: *const u16, n: u64, symbols: *mut u8) -> c_int;
    fn alice_codec_rans_encoder_new() -> *mut RawRansEncoder;
    fn alice_codec_rans_encoder_destroy(e: *mut RawRansEncoder);
    fn alice_codec_rans_encoder_encode(e: *mut RawRansEncoder, cum_freq: u16, freq: u16) -> c_int;
    fn alice_codec_rans_encoder_encode_symbols(e: *mut RawRansEncoder, symbols: *const u8, n: u64, cum_freq: *const u16, freq: *const u16) -> c_int;
    fn alice_codec_rans_encoder_finish(e: *mut RawRansEncoder, out_len: *mut u64) -> *mut u8;
    fn alice_codec_rans_decoder_new(data: *const u8, len: u64) -> *mut RawRansDecoder;
    fn alice_codec_rans_decoder_destroy(d: *mut RawRansDecoder);
    fn alice_codec_rans_decoder_decode_n(d: *mut RawRansDecoder, n: u64, cum_freq: *const u16, freq: *const u16, symbols: *mut u8) -> c_int;
    fn alice_codec_rans_decoder_is_empty(d: *const RawRansDecoder) -> c_int;
    fn alice_codec_quantize_subband(step: i32, dead_zone: i32, coeffs: *const i32, n: u64, out: *mut i32, n_out: u64) -> c_int;
    fn alice_codec_dequantize_subband(step: i32, coeffs: *const i32, n: u64, out: *mut i32, n_out: u64) -> c_int;
    fn alice_codec_encode_many(e: *const RawEncoder, rgb: *const u8, rgb_len: u64, w: u32, h: u32, f: u32, n_chunks: u32, out: *mut *mut RawChunk) -> c_int;
    fn alice_codec_decode_many(chunks: *const *const RawChunk, n_chunks: u32, rgb_out: *mut u8, rgb_out_len: u64) -> c_int;
    fn alice_codec_encode_many_devices(e: *const RawEncoder, rgb: *const u8, rgb_len: u64, w: u32, h: u32, f: u32, n_chunks: u32,
                                       devices: *const c_int, n_devices: u32, out: *mut *mut RawChunk) -> c_int;
    fn alice_codec_decode_many_devices(chunks: *const *const RawChunk, n_chunks: u32, devices: *const c_int, n_devices: u32,
                                       rgb_out: *mut u8, rgb_out_len: u64) -> c_int;
}

// ---------------------------------------------------------------------------------------------------------------
// errors: CodecError (src/error.rs:12-23) + the library-side conditions
// ---------------------------------------------------------------------------------------------------------------

#[derive(Debug, Clone, PartialEq, Eq)]
pub enum CodecError {
    InvalidBufferSize { expected: usize, got: usize },
    InvalidDimensions { width: u32, height: u32 },
    DimensionOverflow,
    InvalidBitstream(String),
    InvalidQuantStep(i32),
    /// the reference would spin forever / divide by zero on this input (src/rans.rs:275-283)
    ReferenceDiverges,
    /// no usable MI355X, out of device memory, or a HIP runtime failure; the message is the library's
    Device(String),
}

fn last_error(expected: usize, got: usize, width: u32, height: u32, step: i32) -> CodecError {
    // ALICE_ERR_* (include/alice_codec.h): 1..=5 are the reference's variants in declaration order
    let msg = unsafe {
        let p = alice_codec_last_error_message();
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    match unsafe { alice_codec_last_error() } {
        1 => CodecError::InvalidBufferSize { expected, got },
        2 => CodecError::InvalidDimensions { width, height },
        3 => CodecError::DimensionOverflow,
        4 => CodecError::InvalidBitstream(msg),
        5 => CodecError::InvalidQuantStep(step),
        6 => CodecError::ReferenceDiverges,
        _ => CodecError::Device(msg),
    }
}

/// 0 -> Ok, anything else -> the calling thread's last error (expected / got fill InvalidBufferSize)
fn check(rc: c_int, expected: usize, got: usize) -> Result<(), CodecError> {
    if rc == 0 { Ok(()) } else { Err(last_error(expected, got, 0, 0, 0)) }
}

unsafe fn take(p: *mut u8, n: u64) -> Vec<u8> {
    let v = std::slice::from_raw_parts(p, n as usize).to_vec();
    alice_codec_data_free64(p, n);
    v
}

// ---------------------------------------------------------------------------------------------------------------
// pipeline: src/pipeline.rs
// ---------------------------------------------------------------------------------------------------------------

/// src/pipeline.rs:34-41 (the discriminants are the `.alc` wavelet byte, :46-62)
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
#[repr(u8)]
pub enum WaveletType { Cdf53 = 0, Cdf97 = 1, Haar = 2 }

/// src/pipeline.rs:172-185.  Owns the library's chunk handle; `to_bytes` is the `.alc` serialisation.
pub struct EncodedChunk { raw: *mut RawChunk }
unsafe impl Send for EncodedChunk {}
unsafe impl Sync for EncodedChunk {}   // immutable after creation, as src/pipeline.rs:635-644 asserts

impl EncodedChunk {
    pub fn width(&self) -> u32 { unsafe { alice_codec_chunk_width(self.raw) } }
    pub fn height(&self) -> u32 { unsafe { alice_codec_chunk_height(self.raw) } }
    pub fn frames(&self) -> u32 { unsafe { alice_codec_chunk_frames(self.raw) } }
    pub fn wavelet_type(&self) -> WaveletType {
        match unsafe { alice_codec_chunk_wavelet(self.raw) } { 1 => WaveletType::Cdf97, 2 => WaveletType::Haar, _ => WaveletType::Cdf53 }
    }
    /// src/pipeline.rs:190
    pub fn compressed_size(&self) -> usize { unsafe { alice_codec_chunk_compressed_size(self.raw) as usize } }
    /// src/pipeline.rs:200
    pub fn to_bytes(&self) -> Vec<u8> {
        let mut n = 0u64;
        unsafe { let p = alice_codec_chunk_to_bytes64(self.raw, &mut n); if p.is_null() { Vec::new() } else { take(p, n) } }
    }
    /// src/pipeline.rs:235 (validation and messages: codec.hip parse_alc_header / chunk_from_bytes)
    pub fn from_bytes(data: &[u8]) -> Result<Self, CodecError> {
        let raw = unsafe { alice_codec_chunk_from_bytes64(data.as_ptr(), data.len() as u64) };
        if raw.is_null() { Err(last_error(0, data.len(), 0, 0, 0)) } else { Ok(Self { raw }) }
    }
}
impl Drop for EncodedChunk { fn drop(&mut self) { unsafe { alice_codec_chunk_destroy(self.raw) } } }

/// src/pipeline.rs:335-340
pub struct FrameEncoder { quality: u8, wavelet_type: WaveletType }

impl FrameEncoder {
    /// src/pipeline.rs:347: CDF 5/3
    pub const fn new(quality: u8) -> Self { Self { quality, wavelet_type: WaveletType::Cdf53 } }
    /// src/pipeline.rs:356
    pub const fn with_wavelet(quality: u8, wavelet_type: WaveletType) -> Self { Self { quality, wavelet_type } }
    /// src/pipeline.rs:377.  `rgb_frames`: interleaved RGB, frame-major, `width * height * frames * 3` bytes.
    pub fn encode(&self, rgb_frames: &[u8], width: u32, height: u32, frames: u32) -> Result<EncodedChunk, CodecError> {
        unsafe {
            let e = alice_codec_encoder_create_ex(self.quality, self.wavelet_type as u8);
            let raw = alice_codec_encode64(e, rgb_frames.as_ptr(), rgb_frames.len() as u64, width, height, frames);
            alice_codec_encoder_destroy(e);
            if raw.is_null() {
                let expected = (width as usize).saturating_mul(height as usize).saturating_mul(frames as usize).saturating_mul(3);
                return Err(last_error(expected, rgb_frames.len(), width, height, 0));
            }
            Ok(EncodedChunk { raw })
        }
    }
}

/// src/pipeline.rs:519
pub struct FrameDecoder;

impl FrameDecoder {
    /// src/pipeline.rs:524
    pub const fn new() -> Self { Self }
    /// src/pipeline.rs:537
    pub fn decode(&self, chunk: &EncodedChunk) -> Result<Vec<u8>, CodecError> {
        let mut n = 0u64;
        unsafe {
            let p = alice_codec_decode64(chunk.raw, &mut n);
            if p.is_null() { Err(last_error(0, 0, chunk.width(), chunk.height(), 0)) } else { Ok(take(p, n)) }
        }
    }
}
impl Default for FrameDecoder { fn default() -> Self { Self::new() } }

// ---------------------------------------------------------------------------------------------------------------
// wavelets: src/wavelet.rs.  The filter is carried as the wavelet byte; the lifting itself runs on the GPU.
// ---------------------------------------------------------------------------------------------------------------

/// src/wavelet.rs:47 -- here only the choice of filter (coefficient lists: csrc/common.h lift_steps)
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct Wavelet1D { kind: WaveletType }
impl Wavelet1D {
    pub fn cdf97() -> Self { Self { kind: WaveletType::Cdf97 } }   // src/wavelet.rs:66
    pub fn haar() -> Self { Self { kind: WaveletType::Haar } }     // :96
    pub fn cdf53() -> Self { Self { kind: WaveletType::Cdf53 } }   // :113
    /// src/wavelet.rs:133 (no-op below two samples; an odd tail sample is dropped, :225-232)
    pub fn forward(&self, signal: &mut [i32]) { Wavelet3D::new(*self).forward(signal, signal.len(), 1, 1) }
    /// src/wavelet.rs:157
    pub fn inverse(&self, signal: &mut [i32]) { Wavelet3D::new(*self).inverse(signal, signal.len(), 1, 1) }
}

/// src/wavelet.rs:265
pub struct Wavelet2D { wavelet_1d: Wavelet1D }
impl Wavelet2D {
    pub const fn new(wavelet_1d: Wavelet1D) -> Self { Self { wavelet_1d } }       // :272
    pub fn cdf97() -> Self { Self::new(Wavelet1D::cdf97()) }                        // :278
    pub fn cdf53() -> Self { Self::new(Wavelet1D::cdf53()) }                        // :284
    /// src/wavelet.rs:292 (rows, then columns)
    pub fn forward(&self, image: &mut [i32], width: usize, height: usize) {
        assert!(image.len() >= width * height);
        unsafe { alice_codec_wavelet2d_forward(self.wavelet_1d.kind as u8, image.as_mut_ptr(), width as u64, height as u64); }
    }
    /// src/wavelet.rs:319 (columns, then rows)
    pub fn inverse(&self, image: &mut [i32], width: usize, height: usize) {
        assert!(image.len() >= width * height);
        unsafe { alice_codec_wavelet2d_inverse(self.wavelet_1d.kind as u8, image.as_mut_ptr(), width as u64, height as u64); }
    }
}

/// src/wavelet.rs:359
pub struct Wavelet3D { wavelet_1d: Wavelet1D }
impl Wavelet3D {
    pub const fn new(wavelet_1d: Wavelet1D) -> Self { Self { wavelet_1d } }       // :366
    pub fn cdf97() -> Self { Self::new(Wavelet1D::cdf97()) }                        // :372
    pub fn cdf53() -> Self { Self::new(Wavelet1D::cdf53()) }                        // :378
    /// src/wavelet.rs:392: rows and columns of every frame, then the temporal vector of every pixel; one level;
    /// `volume[t * width * height + y * width + x]`, in place, any i32 values (wrapping sums, 64-bit products).
    pub fn forward(&self, volume: &mut [i32], width: usize, height: usize, depth: usize) {
        assert!(volume.len() >= width * height * depth);
        unsafe { alice_codec_wavelet3d_forward(self.wavelet_1d.kind as u8, volume.as_mut_ptr(), width as u64, height as u64, depth as u64); }
    }
    /// src/wavelet.rs:441: temporal, then columns, then rows (negated coefficients, rounding not mirrored, :167-174)
    pub fn inverse(&self, volume: &mut [i32], width: usize, height: usize, depth: usize) {
        assert!(volume.len() >= width * height * depth);
        unsafe { alice_codec_wavelet3d_inverse(self.wavelet_1d.kind as u8, volume.as_mut_ptr(), width as u64, height as u64, depth as u64); }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// quantiser and symbols: src/quant.rs
// ---------------------------------------------------------------------------------------------------------------

/// src/quant.rs:171.  The scalar `quantize` / `dequantize` are the reference's own arithmetic (they are two lines and
/// gain nothing from a device round trip); the buffer forms run on the GPU.
pub struct FastQuantizer { raw: *mut RawFastQuantizer, step: i32, dead_zone: i32, reciprocal: u64, shift: u32 }
unsafe impl Send for FastQuantizer {}
unsafe impl Sync for FastQuantizer {}

impl FastQuantizer {
    /// src/quant.rs:190: `step <= 0` is `InvalidQuantStep`
    pub fn new(step: i32) -> Result<Self, CodecError> { Self::with_dead_zone(step, step) }
    /// src/quant.rs:224
    pub fn with_dead_zone(step: i32, dead_zone: i32) -> Result<Self, CodecError> {
        if step <= 0 { return Err(CodecError::InvalidQuantStep(step)); }
        let raw = unsafe { alice_codec_fastquant_with_dead_zone(step, dead_zone) };
        if raw.is_null() { return Err(last_error(0, 0, 0, 0, step)); }
        // src/quant.rs:200-216: reciprocal = ceil(2^(32 + bits(step)) / step)
        let extra = 32 - (step as u32).leading_zeros();
        let shift = 32 + extra;
        let reciprocal = (((1u128 << shift) + step as u128 - 1) / step as u128) as u64;
        Ok(Self { raw, step, dead_zone, reciprocal, shift })
    }
    /// src/quant.rs:243
    pub const fn quantize(&self, value: i32) -> i32 {
        let abs_val = value.unsigned_abs();
        if abs_val < self.dead_zone as u32 { return 0; }
        let adjusted = abs_val - (self.dead_zone as u32) / 2;
        let q = ((adjusted as u64).wrapping_mul(self.reciprocal) >> self.shift) as i32;   // fast_div, :232-236
        if value < 0 { -q } else { q }
    }
    /// src/quant.rs:269
    pub const fn dequantize(&self, qvalue: i32) -> i32 { qvalue.wrapping_mul(self.step) }
    /// src/quant.rs:282: `output.len() < input.len()` is `InvalidBufferSize`
    pub fn quantize_buffer(&self, input: &[i32], output: &mut [i32]) -> Result<(), CodecError> {
        let rc = unsafe { alice_codec_fastquant_quantize_buffer(self.raw, input.as_ptr(), input.len() as u64, output.as_mut_ptr(), output.len() as u64) };
        if rc == 0 { Ok(()) } else { Err(last_error(input.len(), output.len(), 0, 0, self.step)) }
    }
    /// src/quant.rs:300
    pub fn dequantize_buffer(&self, input: &[i32], output: &mut [i32]) -> Result<(), CodecError> {
        let rc = unsafe { alice_codec_fastquant_dequantize_buffer(self.raw, input.as_ptr(), input.len() as u64, output.as_mut_ptr(), output.len() as u64) };
        if rc == 0 { Ok(()) } else { Err(last_error(input.len(), output.len(), 0, 0, self.step)) }
    }
    /// src/quant.rs:322: the reference's AVX2 entry point; same results as `quantize_buffer`, panics on a short output
    pub fn quantize_buffer_simd(&self, input: &[i32], output: &mut [i32]) {
        self.quantize_buffer(input, output).expect("output buffer too small");
    }
    pub const fn step(&self) -> i32 { self.step }             // src/quant.rs:337
    pub const fn dead_zone(&self) -> i32 { self.dead_zone }   // src/quant.rs:344
}
impl Drop for FastQuantizer { fn drop(&mut self) { unsafe { alice_codec_fastquant_destroy(self.raw) } } }

/// src/quant.rs:547
pub fn to_symbols(coeffs: &[i32], symbols: &mut [u8]) -> Result<(), CodecError> {
    let rc = unsafe { alice_codec_to_symbols(coeffs.as_ptr(), coeffs.len() as u64, symbols.as_mut_ptr(), symbols.len() as u64) };
    if rc == 0 { Ok(()) } else { Err(last_error(coeffs.len(), symbols.len(), 0, 0, 0)) }
}
/// src/quant.rs:572
pub fn from_symbols(symbols: &[u8], coeffs: &mut [i32]) -> Result<(), CodecError> {
    let rc = unsafe { alice_codec_from_symbols(symbols.as_ptr(), symbols.len() as u64, coeffs.as_mut_ptr(), coeffs.len() as u64) };
    if rc == 0 { Ok(()) } else { Err(last_error(symbols.len(), coeffs.len(), 0, 0, 0)) }
}
/// src/quant.rs:594
pub fn build_histogram(symbols: &[u8]) -> [u32; 256] {
    let mut h = [0u32; 256];
    unsafe { alice_codec_build_histogram(symbols.as_ptr(), symbols.len() as u64, h.as_mut_ptr()); }
    h
}

// ---------------------------------------------------------------------------------------------------------------
// entropy coder: src/rans.rs
// ---------------------------------------------------------------------------------------------------------------

/// src/rans.rs:59
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct RansSymbol { pub cum_freq: u16, pub freq: u16 }
impl RansSymbol { pub const fn new(cum_freq: u16, freq: u16) -> Self { Self { cum_freq, freq } } }   // :69

/// src/rans.rs:85.  n symbols, 1 <= n <= 256 (the coders address symbols as u8); the arrays always hold 256 entries,
/// those from n on are (0, 0).
pub struct FrequencyTable { cum_freq: [u16; 256], freq: [u16; 256], n_symbols: usize }
impl FrequencyTable {
    /// src/rans.rs:102 (freq 1 for unused symbols, wrapping fix on the last symbol, uniform fallback when empty); any
    /// slice length the u8 symbol API can address.  Panics where the reference panics (an empty slice divides by zero in
    /// `uniform(0)`, :159) and for more than 256 symbols, which the GPU path does not carry.
    pub fn from_histogram(histogram: &[u32]) -> Self {
        let mut t = Self { cum_freq: [0; 256], freq: [0; 256], n_symbols: histogram.len() };
        let rc = unsafe { alice_codec_freq_table_from_histogram_n(histogram.as_ptr(), histogram.len() as u32, t.cum_freq.as_mut_ptr(), t.freq.as_mut_ptr()) };
        assert!(rc == 0, "FrequencyTable::from_histogram: {:?}", last_error(0, 0, 0, 0, 0));
        t
    }
    /// src/rans.rs:158
    pub fn uniform(n_symbols: usize) -> Self { Self::from_histogram(&vec![0u32; n_symbols]) }   // total == 0 -> uniform(n) (:106-109)
    pub fn get_symbol(&self, sym: u8) -> RansSymbol {                                             // :194 (panics out of range, as the Vec index does)
        assert!((sym as usize) < self.n_symbols);
        RansSymbol::new(self.cum_freq[sym as usize], self.freq[sym as usize])
    }
    pub const fn len(&self) -> usize { self.n_symbols }          // :209
    pub const fn is_empty(&self) -> bool { self.n_symbols == 0 } // :216
}

/// src/rans.rs:238: an encoder object that lives across calls.  Every call runs one chain on the device from the object's
/// current state (`encode_symbols` s1 then s2 leaves the stream of one call on s2 || s1, :288-294; `encode` takes the
/// (cum_freq, freq) pair itself, :269-285); the bytes a call emits come back to the host, `finish` prepends the state.
/// The methods return Results because of the one documented divergence (a frequency-0 symbol, :275-283: the reference
/// never returns).
pub struct RansEncoder { raw: *mut RawRansEncoder }
impl RansEncoder {
    pub fn new() -> Self { Self { raw: unsafe { alice_codec_rans_encoder_new() } } }             // :249
    pub fn with_capacity(_capacity: usize) -> Self { Self::new() }                               // :258 (a hint there too)
    /// src/rans.rs:269
    pub fn encode(&mut self, sym: &RansSymbol) -> Result<(), CodecError> {
        check(unsafe { alice_codec_rans_encoder_encode(self.raw, sym.cum_freq, sym.freq) }, 0, 0)
    }
    /// src/rans.rs:288
    pub fn encode_symbols(&mut self, symbols: &[u8], table: &FrequencyTable) -> Result<(), CodecError> {
        if symbols.is_empty() { return Ok(()); }
        check(unsafe { alice_codec_rans_encoder_encode_symbols(self.raw, symbols.as_ptr(), symbols.len() as u64, table.cum_freq.as_ptr(), table.freq.as_ptr()) }, 0, 0)
    }
    /// src/rans.rs:298: the stream starts with the final state, big-endian
    pub fn finish(mut self) -> Result<Vec<u8>, CodecError> {
        let mut n = 0u64;
        let raw = std::mem::replace(&mut self.raw, std::ptr::null_mut());
        unsafe {
            let p = alice_codec_rans_encoder_finish(raw, &mut n);   // consumes the handle
            if p.is_null() { Err(last_error(0, 0, 0, 0, 0)) } else { Ok(take(p, n)) }
        }
    }
}
impl Drop for RansEncoder { fn drop(&mut self) { if !self.raw.is_null() { unsafe { alice_codec_rans_encoder_destroy(self.raw) } } } }
impl Default for RansEncoder { fn default() -> Self { Self::new() } }

/// src/rans.rs:321: a decoder object; `decode` / `decode_n` continue from the current state and position (the library
/// keeps a copy of the input, so the lifetime parameter of the reference type only documents the borrow)
pub struct RansDecoder<'a> { raw: *mut RawRansDecoder, _input: std::marker::PhantomData<&'a [u8]> }
impl<'a> RansDecoder<'a> {
    pub fn new(input: &'a [u8]) -> Self {   // :330
        Self { raw: unsafe { alice_codec_rans_decoder_new(input.as_ptr(), input.len() as u64) }, _input: std::marker::PhantomData }
    }
    /// src/rans.rs:351
    pub fn decode(&mut self, table: &FrequencyTable) -> u8 { self.decode_n(1, table)[0] }
    /// src/rans.rs:375, including the behaviour on short, truncated and desynchronised streams (:341-347, 365-368)
    pub fn decode_n(&mut self, n: usize, table: &FrequencyTable) -> Vec<u8> {
        let mut out = vec![0u8; n];
        unsafe { alice_codec_rans_decoder_decode_n(self.raw, n as u64, table.cum_freq.as_ptr(), table.freq.as_ptr(), out.as_mut_ptr()); }
        out
    }
    /// src/rans.rs:385
    pub fn is_empty(&self) -> bool { unsafe { alice_codec_rans_decoder_is_empty(self.raw) != 0 } }
}
impl<'a> Drop for RansDecoder<'a> { fn drop(&mut self) { unsafe { alice_codec_rans_decoder_destroy(self.raw) } } }

/// src/quant.rs:518 / :531: a sub-band's coefficients through a Quantizer (step, dead zone)
pub fn quantize_subband(coeffs: &[i32], step: i32, dead_zone: i32, output: &mut [i32]) -> Result<(), CodecError> {
    check(unsafe { alice_codec_quantize_subband(step, dead_zone, coeffs.as_ptr(), coeffs.len() as u64, output.as_mut_ptr(), output.len() as u64) }, coeffs.len(), output.len())
}
pub fn dequantize_subband(coeffs: &[i32], step: i32, output: &mut [i32]) -> Result<(), CodecError> {
    check(unsafe { alice_codec_dequantize_subband(step, coeffs.as_ptr(), coeffs.len() as u64, output.as_mut_ptr(), output.len() as u64) }, coeffs.len(), output.len())
}

/// The chunk driver over several GPUs of the node (src/pipeline.rs:461-497: 64-frame chunks are independent): chunk k
/// runs on devices[k % devices.len()], one host thread per entry inside the library; an empty list = the calling
/// thread's device.  Byte-identical to `chunks.len()` calls of `FrameEncoder::encode`.
pub fn encode_many(encoder: &FrameEncoder, rgb_chunks: &[u8], width: u32, height: u32, frames: u32, n_chunks: u32, devices: &[i32])
                   -> Result<Vec<EncodedChunk>, CodecError> {
    let raw_enc = unsafe { alice_codec_encoder_create_ex(encoder.quality, encoder.wavelet_type as u8) };
    let mut raw: Vec<*mut RawChunk> = vec![std::ptr::null_mut(); n_chunks as usize];
    let rc = unsafe {
        if devices.is_empty() { alice_codec_encode_many(raw_enc, rgb_chunks.as_ptr(), rgb_chunks.len() as u64, width, height, frames, n_chunks, raw.as_mut_ptr()) }
        else { alice_codec_encode_many_devices(raw_enc, rgb_chunks.as_ptr(), rgb_chunks.len() as u64, width, height, frames, n_chunks,
                                               devices.as_ptr(), devices.len() as u32, raw.as_mut_ptr()) }
    };
    unsafe { alice_codec_encoder_destroy(raw_enc) };
    check(rc, 0, 0)?;
    Ok(raw.into_iter().map(|r| EncodedChunk { raw: r }).collect())
}
pub fn decode_many(chunks: &[EncodedChunk], devices: &[i32]) -> Result<Vec<u8>, CodecError> {
    if chunks.is_empty() { return Ok(Vec::new()); }
    let raw: Vec<*const RawChunk> = chunks.iter().map(|c| c.raw as *const RawChunk).collect();
    let per = chunks[0].width() as usize * chunks[0].height() as usize * chunks[0].frames() as usize * 3;
    let mut out = vec![0u8; per * chunks.len()];
    let rc = unsafe {
        if devices.is_empty() { alice_codec_decode_many(raw.as_ptr(), raw.len() as u32, out.as_mut_ptr(), out.len() as u64) }
        else { alice_codec_decode_many_devices(raw.as_ptr(), raw.len() as u32, devices.as_ptr(), devices.len() as u32, out.as_mut_ptr(), out.len() as u64) }
    };
    check(rc, 0, 0)?;
    Ok(out)
}
