//! # JPEG encoder — MI355X drop-in
//!
//! The public API of `jpeg-encoder` 0.7.0 (`Encoder<W: JfifWrite>`, `ColorType`, `SamplingFactor`,
//! `QuantizationTableType`, `PixelDensity`, `EncodingError`, `ImageBuffer`, `rgb_to_ycbcr`, `cmyk_to_ycck`;
//! reference: `src/lib.rs:45-49`, `src/encoder.rs:213-515`) over `libjpegenc_mi355x.so`: colour conversion,
//! subsampling, the forward DCT, quantisation and the Huffman coding of every scan run on the GPU, the emitted
//! bytes are the reference's.
//!
//! ```no_run
//! use jpeg_encoder::{Encoder, ColorType};
//! # fn main() -> Result<(), jpeg_encoder::EncodingError> {
//! let data = [255u8, 0, 0, 0, 255, 0, 0, 0, 255, 255, 255, 255];
//! let encoder = Encoder::new_file("some.jpeg", 100)?;
//! encoder.encode(&data, 2, 2, ColorType::Rgb)?;
//! # Ok(()) }
//! ```
//!
//! Differences a caller can observe: there is no CPU fallback (without a gfx950 device `encode` returns
//! `EncodingError::Write("...no HIP device...")`), and three extension methods exist that the reference lacks
//! (`set_device`, `encode_batch`, `encode_batch_multi`).
#![cfg_attr(not(feature = "std"), no_std)]

extern crate alloc;

pub mod sys;

use alloc::boxed::Box;
use alloc::string::String;
use alloc::vec::Vec;
use core::ffi::{c_int, c_void};
use core::fmt;

// ---- error.rs:5-28 ------------------------------------------------------------------------------------------

/// # The error type for encoding
#[derive(Debug)]
pub enum EncodingError {
    /// An invalid app segment number has been used
    InvalidAppSegment(u8),
    /// App segment exceeds maximum allowed data length
    AppSegmentTooLarge(usize),
    /// Color profile exceeds maximum allowed data length
    IccTooLarge(usize),
    /// Image data is too short
    BadImageData { length: usize, required: usize },
    /// Width or height is zero
    ZeroImageDimensions { width: u16, height: u16 },
    /// An io error occurred during writing
    #[cfg(feature = "std")]
    IoError(std::io::Error),
    /// An io error occurred during writing (no_std), or a device / argument error of the MI355X library
    Write(String),
}

#[cfg(feature = "std")]
impl From<std::io::Error> for EncodingError {
    fn from(err: std::io::Error) -> EncodingError {
        EncodingError::IoError(err)
    }
}

impl fmt::Display for EncodingError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        use EncodingError::*;
        match self {
            InvalidAppSegment(nr) => write!(f, "Invalid app segment number: {}", nr),
            AppSegmentTooLarge(length) => write!(f, "App segment exceeds maximum allowed data length of 65533: {}", length),
            IccTooLarge(length) => write!(f, "ICC profile exceeds maximum allowed data length: {}", length),
            BadImageData { length, required } => {
                write!(f, "Image data too small for dimensions and color_type: {} need at least {}", length, required)
            }
            ZeroImageDimensions { width, height } => write!(f, "Image dimensions must be non zero: {}x{}", width, height),
            #[cfg(feature = "std")]
            IoError(err) => err.fmt(f),
            Write(err) => write!(f, "{}", err),
        }
    }
}

#[cfg(feature = "std")]
impl std::error::Error for EncodingError {
    fn source(&self) -> Option<&(dyn std::error::Error + 'static)> {
        match self {
            EncodingError::IoError(err) => Some(err),
            _ => None,
        }
    }
}

// ---- writer.rs:17-82 ----------------------------------------------------------------------------------------

/// Represents the pixel density of an image
#[derive(Clone, Copy, Debug, Eq, PartialEq)]
pub struct PixelDensity {
    /// A couple of values for (Xdensity, Ydensity)
    pub density: (u16, u16),
    /// The unit in which the density is measured
    pub unit: PixelDensityUnit,
}

impl PixelDensity {
    /// Horizontal and vertical density equal, in pixels per inch
    #[must_use]
    pub fn dpi(density: u16) -> Self {
        PixelDensity { density: (density, density), unit: PixelDensityUnit::Inches }
    }
}

impl Default for PixelDensity {
    fn default() -> Self {
        PixelDensity { density: (1, 1), unit: PixelDensityUnit::PixelAspectRatio }
    }
}

/// Represents a unit in which the density of an image is measured
#[derive(Clone, Copy, Debug, Eq, PartialEq)]
pub enum PixelDensityUnit {
    PixelAspectRatio,
    Inches,
    Centimeters,
}

/// A no_std alternative for `std::io::Write` (writer.rs:76-82)
pub trait JfifWrite {
    /// Writes the whole buffer. The behavior must be identical to std::io::Write::write_all
    fn write_all(&mut self, buf: &[u8]) -> Result<(), EncodingError>;
}

#[cfg(not(feature = "std"))]
impl<W: JfifWrite + ?Sized> JfifWrite for &mut W {
    fn write_all(&mut self, buf: &[u8]) -> Result<(), EncodingError> {
        (**self).write_all(buf)
    }
}

#[cfg(not(feature = "std"))]
impl JfifWrite for Vec<u8> {
    fn write_all(&mut self, buf: &[u8]) -> Result<(), EncodingError> {
        self.extend_from_slice(buf);
        Ok(())
    }
}

#[cfg(feature = "std")]
impl<W: std::io::Write + ?Sized> JfifWrite for W {
    #[inline(always)]
    fn write_all(&mut self, buf: &[u8]) -> Result<(), EncodingError> {
        std::io::Write::write_all(self, buf)?;
        Ok(())
    }
}

// ---- encoder.rs:23-188, quantization.rs:8-58 ------------------------------------------------------------------

/// # Color types used in encoding
#[derive(Copy, Clone, Debug, Eq, PartialEq)]
pub enum JpegColorType {
    Luma,
    Ycbcr,
    Cmyk,
    Ycck,
}

/// # Color types for input images (same order as `jpegenc_color_type`)
#[derive(Copy, Clone, Debug, Eq, PartialEq)]
pub enum ColorType {
    Luma,
    Rgb,
    Rgba,
    Bgr,
    Bgra,
    Ycbcr,
    Cmyk,
    CmykAsYcck,
    Ycck,
}

impl ColorType {
    fn get_bytes_per_pixel(self) -> usize {
        use ColorType::*;
        match self {
            Luma => 1,
            Rgb | Bgr | Ycbcr => 3,
            Rgba | Bgra | Cmyk | CmykAsYcck | Ycck => 4,
        }
    }
}

/// # Sampling factors for chroma subsampling (same discriminants as `jpegenc_sampling_factor`)
#[repr(u8)]
#[derive(Copy, Clone, Debug, Eq, PartialEq)]
#[allow(non_camel_case_types)]
pub enum SamplingFactor {
    F_1_1 = 1 << 4 | 1,
    F_2_1 = 2 << 4 | 1,
    F_1_2 = 1 << 4 | 2,
    F_2_2 = 2 << 4 | 2,
    F_4_1 = 4 << 4 | 1,
    F_4_2 = 4 << 4 | 2,
    F_1_4 = 1 << 4 | 4,
    F_2_4 = 2 << 4 | 4,
    R_4_4_4 = 0x80 | 1 << 4 | 1,
    R_4_4_0 = 0x80 | 1 << 4 | 2,
    R_4_4_1 = 0x80 | 1 << 4 | 4,
    R_4_2_2 = 0x80 | 2 << 4 | 1,
    R_4_2_0 = 0x80 | 2 << 4 | 2,
    R_4_2_1 = 0x80 | 2 << 4 | 4,
    R_4_1_1 = 0x80 | 4 << 4 | 1,
    R_4_1_0 = 0x80 | 4 << 4 | 2,
}

impl SamplingFactor {
    /// Get variant for supplied factors or None if not supported
    pub fn from_factors(horizontal: u8, vertical: u8) -> Option<SamplingFactor> {
        use SamplingFactor::*;
        match (horizontal, vertical) {
            (1, 1) => Some(F_1_1),
            (1, 2) => Some(F_1_2),
            (1, 4) => Some(F_1_4),
            (2, 1) => Some(F_2_1),
            (2, 2) => Some(F_2_2),
            (2, 4) => Some(F_2_4),
            (4, 1) => Some(F_4_1),
            (4, 2) => Some(F_4_2),
            _ => None,
        }
    }

    fn from_discriminant(v: c_int) -> SamplingFactor {
        use SamplingFactor::*;
        const ALL: [SamplingFactor; 16] = [
            F_1_1, F_2_1, F_1_2, F_2_2, F_4_1, F_4_2, F_1_4, F_2_4, R_4_4_4, R_4_4_0, R_4_4_1, R_4_2_2, R_4_2_0, R_4_2_1,
            R_4_1_1, R_4_1_0,
        ];
        for s in ALL {
            if s as u8 as c_int == v {
                return s;
            }
        }
        F_1_1
    }
}

/// # Quantization table used for encoding (index() == `jpegenc_qtable_type`)
#[derive(Debug, Clone)]
pub enum QuantizationTableType {
    Default,
    Flat,
    CustomMsSsim,
    CustomPsnrHvs,
    ImageMagick,
    KleinSilversteinCarney,
    DentalXRays,
    VisualDetectionModel,
    ImprovedDetectionModel,
    /// A user supplied quantization table
    Custom(Box<[u16; 64]>),
}

impl QuantizationTableType {
    fn abi(&self) -> (c_int, *const u16) {
        use QuantizationTableType::*;
        match self {
            Default => (0, core::ptr::null()),
            Flat => (1, core::ptr::null()),
            CustomMsSsim => (2, core::ptr::null()),
            CustomPsnrHvs => (3, core::ptr::null()),
            ImageMagick => (4, core::ptr::null()),
            KleinSilversteinCarney => (5, core::ptr::null()),
            DentalXRays => (6, core::ptr::null()),
            VisualDetectionModel => (7, core::ptr::null()),
            ImprovedDetectionModel => (8, core::ptr::null()),
            Custom(t) => (sys::JPEGENC_Q_CUSTOM, t.as_ptr()),
        }
    }
}

// ---- image_buffer.rs:9-98 -----------------------------------------------------------------------------------

/// Conversion from RGB to YCbCr (image_buffer.rs:9-31)
#[inline]
pub fn rgb_to_ycbcr(r: u8, g: u8, b: u8) -> (u8, u8, u8) {
    let mut out = [0u8; 3];
    unsafe { sys::jpegenc_rgb_to_ycbcr(r, g, b, out.as_mut_ptr()) };
    (out[0], out[1], out[2])
}

/// Conversion from CMYK to YCCK (YCbCrK) (image_buffer.rs:33-38)
#[inline]
pub fn cmyk_to_ycck(c: u8, m: u8, y: u8, k: u8) -> (u8, u8, u8, u8) {
    let mut out = [0u8; 4];
    unsafe { sys::jpegenc_cmyk_to_ycck(c, m, y, k, out.as_mut_ptr()) };
    (out[0], out[1], out[2], out[3])
}

/// # Buffer used as input value for image encoding (image_buffer.rs:86-98)
pub trait ImageBuffer {
    /// The color type used in the image encoding
    fn get_jpeg_color_type(&self) -> JpegColorType;
    /// Width of the image
    fn width(&self) -> u16;
    /// Height of the image
    fn height(&self) -> u16;
    /// Add color values for the row to color component buffers
    fn fill_buffers(&self, y: u16, buffers: &mut [Vec<u8>; 4]);
}

// ---- glue -----------------------------------------------------------------------------------------------------

fn last_error() -> String {
    unsafe {
        let p = sys::jpegenc_last_error();
        if p.is_null() {
            return String::new();
        }
        let mut n = 0usize;
        while *p.add(n) != 0 {
            n += 1;
        }
        String::from_utf8_lossy(core::slice::from_raw_parts(p as *const u8, n)).into_owned()
    }
}

/// Where the sink trampoline writes and where it parks the writer's own error (an `io::Error` must reach the
/// caller as `EncodingError::IoError`, not as a status code).
struct SinkState<'a, W: JfifWrite> {
    w: &'a mut W,
    err: Option<EncodingError>,
    panic: Option<Box<dyn core::any::Any + Send + 'static>>,
}

/// A panic of the caller's writer / `ImageBuffer` must not unwind through the library's C++ frames: with `std` it is
/// caught here, the call fails with a non-zero status and the payload is re-raised once the FFI call has returned
/// (`Encoder::finish`); without `std` there is no unwinding to contain (panic = abort).
#[cfg(feature = "std")]
fn contain<R>(f: impl FnOnce() -> R) -> Result<R, Box<dyn core::any::Any + Send + 'static>> {
    std::panic::catch_unwind(std::panic::AssertUnwindSafe(f))
}
#[cfg(not(feature = "std"))]
fn contain<R>(f: impl FnOnce() -> R) -> Result<R, Box<dyn core::any::Any + Send + 'static>> {
    Ok(f())
}
/// Re-raises a panic caught inside a callback, now that no C++ frame is on the stack.
fn resume(payload: Option<Box<dyn core::any::Any + Send + 'static>>) {
    #[cfg(feature = "std")]
    if let Some(p) = payload {
        std::panic::resume_unwind(p);
    }
    #[cfg(not(feature = "std"))]
    let _ = payload;
}

unsafe extern "C" fn sink_trampoline<W: JfifWrite>(user: *mut c_void, data: *const u8, len: usize) -> c_int {
    let st = &mut *(user as *mut SinkState<'_, W>);
    if st.err.is_some() || st.panic.is_some() {
        return 1;
    }
    let w = &mut *st.w;
    match contain(|| w.write_all(core::slice::from_raw_parts(data, len))) {
        Ok(Ok(())) => 0,
        Ok(Err(e)) => {
            st.err = Some(e);
            1
        }
        Err(payload) => {
            st.panic = Some(payload);
            1
        }
    }
}

struct FillState<'a, I: ImageBuffer> {
    image: &'a I,
    bufs: [Vec<u8>; 4],
    planes: usize,
    width: usize,
    panic: Option<Box<dyn core::any::Any + Send + 'static>>,
}

unsafe extern "C" fn fill_trampoline<I: ImageBuffer>(user: *mut c_void, y: u16, planes: *const *mut u8) {
    let st = &mut *(user as *mut FillState<'_, I>);
    if st.panic.is_some() {
        return;                                                              // (the rows after a panic stay as they are; the panic is resumed after the call)
    }
    for b in st.bufs.iter_mut() {
        b.clear();
    }
    let (image, bufs) = (st.image, &mut st.bufs);
    if let Err(payload) = contain(|| image.fill_buffers(y, bufs)) {
        st.panic = Some(payload);
        return;
    }
    for i in 0..st.planes {
        let n = core::cmp::min(st.width, st.bufs[i].len());
        core::ptr::copy_nonoverlapping(st.bufs[i].as_ptr(), *planes.add(i), n);
    }
}

/// # The JPEG encoder
pub struct Encoder<W: JfifWrite> {
    h: *mut sys::jpegenc_encoder,
    w: W,
    quantization_tables: [QuantizationTableType; 2],
}

// The handle owns device buffers and streams but no thread affinity: it may move to another thread with its writer.
unsafe impl<W: JfifWrite + Send> Send for Encoder<W> {}

impl<W: JfifWrite> Drop for Encoder<W> {
    fn drop(&mut self) {
        unsafe { sys::jpegenc_encoder_free(self.h) }
    }
}

impl<W: JfifWrite> Encoder<W> {
    /// Create a new encoder with the given quality (encoder.rs:239-275).
    ///
    /// Quality settings below 90 use a chroma subsampling of 2x2 (4:2:0) by default.
    pub fn new(w: W, quality: u8) -> Encoder<W> {
        let h = unsafe { sys::jpegenc_encoder_new(quality as c_int) };
        assert!(!h.is_null(), "jpegenc_encoder_new failed");
        // encoder.rs:456-488: a `simd` build encodes with fdct_avx2 where AVX2 is detected at run time; its
        // coefficients differ from the scalar fdct (two rows floored instead of rounded), so follow the same test
        #[cfg(all(feature = "simd", any(target_arch = "x86", target_arch = "x86_64")))]
        {
            if std::is_x86_feature_detected!("avx2") {
                unsafe { sys::jpegenc_encoder_set_fdct_variant(h, sys::JPEGENC_FDCT_SIMD) };
            }
        }
        Encoder { h, w, quantization_tables: [QuantizationTableType::Default, QuantizationTableType::Default] }
    }

    /// Set pixel density for the image
    pub fn set_density(&mut self, density: PixelDensity) {
        let unit = match density.unit {
            PixelDensityUnit::PixelAspectRatio => 0,
            PixelDensityUnit::Inches => 1,
            PixelDensityUnit::Centimeters => 2,
        };
        unsafe { sys::jpegenc_encoder_set_density(self.h, unit, density.density.0, density.density.1) };
    }

    /// Return pixel density
    pub fn density(&self) -> PixelDensity {
        let (mut unit, mut x, mut y) = (0 as c_int, 0u16, 0u16);
        unsafe { sys::jpegenc_encoder_density(self.h, &mut unit, &mut x, &mut y) };
        let unit = match unit {
            1 => PixelDensityUnit::Inches,
            2 => PixelDensityUnit::Centimeters,
            _ => PixelDensityUnit::PixelAspectRatio,
        };
        PixelDensity { density: (x, y), unit }
    }

    /// Set chroma subsampling factor
    pub fn set_sampling_factor(&mut self, sampling: SamplingFactor) {
        unsafe { sys::jpegenc_encoder_set_sampling_factor(self.h, sampling as u8 as c_int) };
    }

    /// Get chroma subsampling factor
    pub fn sampling_factor(&self) -> SamplingFactor {
        SamplingFactor::from_discriminant(unsafe { sys::jpegenc_encoder_sampling_factor(self.h) })
    }

    /// Set quantization tables for luma and chroma components
    pub fn set_quantization_tables(&mut self, luma: QuantizationTableType, chroma: QuantizationTableType) {
        {
            let (lt, lp) = luma.abi();
            let (ct, cp) = chroma.abi();
            unsafe { sys::jpegenc_encoder_set_quantization_tables(self.h, lt, lp, ct, cp) };
        }
        self.quantization_tables = [luma, chroma];
    }

    /// Get configured quantization tables
    pub fn quantization_tables(&self) -> &[QuantizationTableType; 2] {
        &self.quantization_tables
    }

    /// Controls if progressive encoding is used (4 scans by default)
    pub fn set_progressive(&mut self, progressive: bool) {
        unsafe { sys::jpegenc_encoder_set_progressive(self.h, progressive as c_int) };
    }

    /// Set number of scans per component for progressive encoding
    ///
    /// # Panics
    /// If number of scans is not within 2..=64 (encoder.rs:328-335)
    pub fn set_progressive_scans(&mut self, scans: u8) {
        assert!((2..=64).contains(&scans), "Invalid number of scans: {}", scans);
        unsafe { sys::jpegenc_encoder_set_progressive_scans(self.h, scans as c_int) };
    }

    /// Return number of progressive scans if progressive encoding is enabled
    pub fn progressive_scans(&self) -> Option<u8> {
        match unsafe { sys::jpegenc_encoder_progressive_scans(self.h) } {
            n if n > 0 => Some(n as u8),
            _ => None,
        }
    }

    /// Set restart interval: numbers of MCUs between restart markers
    pub fn set_restart_interval(&mut self, interval: u16) {
        unsafe { sys::jpegenc_encoder_set_restart_interval(self.h, interval) };
    }

    /// Return the restart interval
    pub fn restart_interval(&self) -> Option<u16> {
        match unsafe { sys::jpegenc_encoder_restart_interval(self.h) } {
            n if n > 0 => Some(n as u16),
            _ => None,
        }
    }

    /// Set if optimized huffman table should be created
    pub fn set_optimized_huffman_tables(&mut self, optimize_huffman_table: bool) {
        unsafe { sys::jpegenc_encoder_set_optimized_huffman_tables(self.h, optimize_huffman_table as c_int) };
    }

    /// Returns if optimized huffman table should be generated
    pub fn optimized_huffman_tables(&self) -> bool {
        unsafe { sys::jpegenc_encoder_optimized_huffman_tables(self.h) > 0 }
    }

    /// Appends a custom app segment to the JFIF file (encoder.rs:374-383)
    pub fn add_app_segment(&mut self, segment_nr: u8, data: Vec<u8>) -> Result<(), EncodingError> {
        match unsafe { sys::jpegenc_encoder_add_app_segment(self.h, segment_nr as c_int, data.as_ptr(), data.len()) } {
            sys::JPEGENC_OK => Ok(()),
            sys::JPEGENC_ERR_INVALID_APP_SEGMENT => Err(EncodingError::InvalidAppSegment(segment_nr)),
            sys::JPEGENC_ERR_APP_SEGMENT_TOO_LARGE => Err(EncodingError::AppSegmentTooLarge(data.len())),
            _ => Err(EncodingError::Write(last_error())),
        }
    }

    /// Add an ICC profile (encoder.rs:392-417)
    pub fn add_icc_profile(&mut self, data: &[u8]) -> Result<(), EncodingError> {
        match unsafe { sys::jpegenc_encoder_add_icc_profile(self.h, data.as_ptr(), data.len()) } {
            sys::JPEGENC_OK => Ok(()),
            sys::JPEGENC_ERR_ICC_TOO_LARGE => Err(EncodingError::IccTooLarge(data.len())),
            sys::JPEGENC_ERR_APP_SEGMENT_TOO_LARGE => Err(EncodingError::AppSegmentTooLarge(data.len())),
            _ => Err(EncodingError::Write(last_error())),
        }
    }

    /// Embeds Exif metadata into the image (encoder.rs:426-435)
    pub fn add_exif_metadata(&mut self, data: &[u8]) -> Result<(), EncodingError> {
        match unsafe { sys::jpegenc_encoder_add_exif_metadata(self.h, data.as_ptr(), data.len()) } {
            sys::JPEGENC_OK => Ok(()),
            sys::JPEGENC_ERR_APP_SEGMENT_TOO_LARGE => Err(EncodingError::AppSegmentTooLarge(data.len() + 6)),
            _ => Err(EncodingError::Write(last_error())),
        }
    }

    fn finish(status: c_int, sink_error: Option<EncodingError>, length: usize, required: usize, width: u16, height: u16) -> Result<(), EncodingError> {
        match status {
            sys::JPEGENC_OK => Ok(()),
            sys::JPEGENC_ERR_BAD_IMAGE_DATA => Err(EncodingError::BadImageData { length, required }),
            sys::JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS => Err(EncodingError::ZeroImageDimensions { width, height }),
            sys::JPEGENC_ERR_WRITE => Err(sink_error.unwrap_or_else(|| EncodingError::Write(last_error()))),
            _ => Err(EncodingError::Write(last_error())),
        }
    }

    /// Encode an image (encoder.rs:440-503).  Data format and length must conform to width, height and color type.
    pub fn encode(mut self, data: &[u8], width: u16, height: u16, color_type: ColorType) -> Result<(), EncodingError> {
        let required = width as usize * height as usize * color_type.get_bytes_per_pixel();
        let h = self.h;
        let mut st = SinkState { w: &mut self.w, err: None, panic: None };
        let status = unsafe {
            sys::jpegenc_encoder_encode(h, data.as_ptr(), data.len(), width as c_int, height as c_int, color_type as c_int,
                                        sink_trampoline::<W>, &mut st as *mut SinkState<'_, W> as *mut c_void)
        };
        resume(st.panic.take());
        let err = st.err.take();
        Self::finish(status, err, data.len(), required, width, height)
    }

    /// Encode an image from a user `ImageBuffer` (encoder.rs:505-515): its rows are produced on the host, one
    /// `fill_buffers` call per row, into pinned staging memory; everything after that runs on the GPU.
    pub fn encode_image<I: ImageBuffer>(mut self, image: I) -> Result<(), EncodingError> {
        let (width, height) = (image.width(), image.height());
        let (jct, planes) = match image.get_jpeg_color_type() {
            JpegColorType::Luma => (0, 1),
            JpegColorType::Ycbcr => (1, 3),
            JpegColorType::Cmyk => (2, 4),
            JpegColorType::Ycck => (3, 4),
        };
        let h = self.h;
        let mut fill = FillState { image: &image, bufs: [Vec::new(), Vec::new(), Vec::new(), Vec::new()], planes, width: width as usize, panic: None };
        let mut st = SinkState { w: &mut self.w, err: None, panic: None };
        let status = unsafe {
            sys::jpegenc_encoder_encode_image(h, jct, width as c_int, height as c_int, fill_trampoline::<I>,
                                              &mut fill as *mut FillState<'_, I> as *mut c_void, sink_trampoline::<W>,
                                              &mut st as *mut SinkState<'_, W> as *mut c_void)
        };
        resume(fill.panic.take());
        resume(st.panic.take());
        let err = st.err.take();
        Self::finish(status, err, 0, 0, width, height)
    }

    // ---- extensions (no counterpart in the reference) ---------------------------------------------------------

    /// HIP device this encoder drives (default 0).
    pub fn set_device(&mut self, device: i32) {
        unsafe { sys::jpegenc_encoder_set_device(self.h, device as c_int) };
    }

    /// `false`: coefficients come back over PCIe and are Huffman-coded on the host (same bytes).
    pub fn set_device_entropy(&mut self, enable: bool) {
        unsafe { sys::jpegenc_encoder_set_device_entropy(self.h, enable as c_int) };
    }

    /// `true`: the host threads of the batch calls run on the NUMA node of the encoder's GPU (off by default).
    pub fn set_numa_bind(&mut self, enable: bool) {
        unsafe { sys::jpegenc_encoder_set_numa_bind(self.h, enable as c_int) };
    }

    /// `true`: one thread of the handle page-locks the pageable frames of a batch a few ahead of the workers, which then upload
    /// them where they lie (one DRAM move per byte instead of three); `false` (default): every worker stages its frame.
    pub fn set_register_ahead_uploads(&mut self, enable: bool) {
        unsafe { sys::jpegenc_encoder_set_batch_upload(self.h, enable as c_int) };
    }

    /// Upper bound on the host threads this encoder's batch calls keep busy at once, the calling thread included
    /// (0 = sized by the library: at most 4 where the GPU codes the scans).  A process that shares its CPU quota with
    /// other ranks - one process per GPU - passes its share.  The reference is single-threaded: 1 reproduces that.
    pub fn set_batch_workers(&mut self, threads: u32) {
        unsafe { sys::jpegenc_encoder_set_batch_workers(self.h, threads as c_int) };
    }

    /// The setting of `set_batch_workers` (0 = automatic).
    pub fn batch_workers(&self) -> u32 {
        (unsafe { sys::jpegenc_encoder_batch_workers(self.h) }).max(0) as u32
    }

    /// Upper bound on the frames of a device-resident batch in flight together (0 = sized by device memory footprint).
    pub fn set_batch_round_frames(&mut self, frames: u32) {
        unsafe { sys::jpegenc_encoder_set_batch_round_frames(self.h, frames as c_int) };
    }

    /// Number of usable MI355X devices (0: `encode` will fail, keep the CPU crate as the fallback).
    pub fn device_count() -> i32 {
        unsafe { sys::jpegenc_device_count() as i32 }
    }

    /// A batch of same-geometry frames with this encoder's settings -> one JPEG file per frame, in order.
    /// `devices`: HIP devices to shard over (frame k -> devices[k % n], `jpegenc_shard_frames`); empty = this
    /// encoder's device.  The writer passed to `new` is not used.
    pub fn encode_batch_multi(&mut self, devices: &[i32], frames: &[&[u8]], width: u16, height: u16, color_type: ColorType) -> Result<Vec<Vec<u8>>, EncodingError> {
        let n = frames.len();
        if n == 0 {
            return Ok(Vec::new());
        }
        let frame_len = frames.iter().map(|f| f.len()).min().unwrap_or(0);
        let required = width as usize * height as usize * color_type.get_bytes_per_pixel();
        let ptrs: Vec<*const u8> = frames.iter().map(|f| f.as_ptr()).collect();
        let mut cap = required / 2 + (1 << 16);
        let mut retried = false;
        loop {
            let mut outs: Vec<Vec<u8>> = (0..n).map(|_| Vec::with_capacity(cap)).collect();
            let out_ptrs: Vec<*mut u8> = outs.iter_mut().map(|o| o.as_mut_ptr()).collect();
            let caps: Vec<usize> = outs.iter().map(|o| o.capacity()).collect();
            let mut lens = alloc::vec![0usize; n];
            let devs: Vec<c_int> = devices.iter().map(|d| *d as c_int).collect();
            let status = unsafe {
                if devs.is_empty() {
                    sys::jpegenc_encoder_encode_batch_to_buffers(self.h, ptrs.as_ptr(), frame_len, n as c_int, width as c_int, height as c_int,
                                                                 color_type as c_int, out_ptrs.as_ptr(), caps.as_ptr(), lens.as_mut_ptr())
                } else {
                    sys::jpegenc_encoder_encode_batch_multi_to_buffers(self.h, devs.as_ptr(), devs.len() as c_int, ptrs.as_ptr(), frame_len,
                                                                       n as c_int, width as c_int, height as c_int, color_type as c_int,
                                                                       out_ptrs.as_ptr(), caps.as_ptr(), lens.as_mut_ptr())
                }
            };
            if status == sys::JPEGENC_ERR_BUFFER_TOO_SMALL && !retried && lens.iter().zip(caps.iter()).any(|(l, c)| l > c) {
                // some output buffer was too small and the library reported the size it needs: once more with room for the
                // largest.  (The same status also means "scan workspace too small" - then no length exceeds its capacity
                // and a retry could not help: it is reported, never looped on.)
                cap = lens.iter().copied().max().unwrap_or(cap) + 4096;
                retried = true;
                continue;
            }
            Self::finish(status, None, frame_len, required, width, height)?;
            for (o, l) in outs.iter_mut().zip(lens.iter()) {
                unsafe { o.set_len(*l) };                                    // the library wrote exactly that many bytes
            }
            return Ok(outs);
        }
    }

    /// `encode_batch_multi` on this encoder's own device.
    pub fn encode_batch(&mut self, frames: &[&[u8]], width: u16, height: u16, color_type: ColorType) -> Result<Vec<Vec<u8>>, EncodingError> {
        self.encode_batch_multi(&[], frames, width, height, color_type)
    }

    /// A batch of same-geometry frames that already lie in this encoder's device memory (a decoder's or a camera pipeline's output),
    /// `frame_stride` bytes apart from `d_frames` on -> one JPEG file per frame, in order.  The call runs as a pipeline of rounds (the
    /// GPU codes round r + 1 while the link carries round r and the library's background threads assemble the files of the rounds
    /// before), so frames per call are worth having.  No counterpart in the crate.
    ///
    /// # Safety
    /// `d_frames` must point to `num_frames` frames of `width * height * bytes-per-pixel` bytes each in memory of the device this
    /// encoder drives (`set_device`), valid and unmodified until the call returns.
    pub unsafe fn encode_batch_device(&mut self, d_frames: *const c_void, frame_stride: usize, num_frames: usize, width: u16, height: u16,
                                      color_type: ColorType) -> Result<Vec<Vec<u8>>, EncodingError> {
        let n = num_frames;
        if n == 0 {
            return Ok(Vec::new());
        }
        let required = width as usize * height as usize * color_type.get_bytes_per_pixel();
        let mut cap = required / 2 + (1 << 16);
        let mut retried = false;
        loop {
            let mut outs: Vec<Vec<u8>> = (0..n).map(|_| Vec::with_capacity(cap)).collect();
            let out_ptrs: Vec<*mut u8> = outs.iter_mut().map(|o| o.as_mut_ptr()).collect();
            let caps: Vec<usize> = outs.iter().map(|o| o.capacity()).collect();
            let mut lens = alloc::vec![0usize; n];
            let status = sys::jpegenc_encoder_encode_batch_device_to_buffers(self.h, d_frames, frame_stride, n as c_int, width as c_int, height as c_int,
                                                                             color_type as c_int, out_ptrs.as_ptr(), caps.as_ptr(), lens.as_mut_ptr());
            if status == sys::JPEGENC_ERR_BUFFER_TOO_SMALL && !retried && lens.iter().zip(caps.iter()).any(|(l, c)| l > c) {
                cap = lens.iter().copied().max().unwrap_or(cap) + 4096;       // (as in encode_batch_multi: one retry, only for an output buffer)
                retried = true;
                continue;
            }
            Self::finish(status, None, required, required, width, height)?;
            for (o, l) in outs.iter_mut().zip(lens.iter()) {
                o.set_len(*l);                                                // the library wrote exactly that many bytes
            }
            return Ok(outs);
        }
    }
}

/// A frame buffer in page-locked host memory (`jpegenc_host_alloc`): batches built from such frames are uploaded in place
/// by the DMA engine, without the workers' staging copy.  Derefs to `[u8]`, so `&*frame` goes where `&[u8]` frames go
/// (`encode_batch`, `encode_batch_multi`).  No counterpart in the crate.
pub struct PinnedFrame {
    ptr: *mut u8,
    len: usize,
}

// the buffer is plain bytes owned by this value
unsafe impl Send for PinnedFrame {}
unsafe impl Sync for PinnedFrame {}

impl PinnedFrame {
    /// `len` zero-initialised page-locked bytes.
    pub fn new(len: usize) -> Result<PinnedFrame, EncodingError> {
        let mut p: *mut core::ffi::c_void = core::ptr::null_mut();
        let status = unsafe { sys::jpegenc_host_alloc(len, &mut p) };
        if status != sys::JPEGENC_OK || (p.is_null() && len != 0) {
            return Err(EncodingError::Write(alloc::string::String::from("page-locked allocation failed")));
        }
        if len != 0 {
            unsafe { core::ptr::write_bytes(p as *mut u8, 0, len) };
        }
        Ok(PinnedFrame { ptr: p as *mut u8, len })
    }

    /// A page-locked copy of `pixels`.
    pub fn from_slice(pixels: &[u8]) -> Result<PinnedFrame, EncodingError> {
        let mut f = PinnedFrame::new(pixels.len())?;
        f.fill_from(pixels);
        Ok(f)
    }

    /// Overwrites the frame with `pixels` (same length) through the library's staging copy (`jpegenc_host_copy`: streaming
    /// stores - the destination is about to be read by the DMA engine, not by this core).
    pub fn fill_from(&mut self, pixels: &[u8]) {
        assert_eq!(pixels.len(), self.len);
        if self.len != 0 {
            unsafe { sys::jpegenc_host_copy(self.ptr as *mut core::ffi::c_void, pixels.as_ptr() as *const core::ffi::c_void, self.len) };
        }
    }
}

impl core::ops::Deref for PinnedFrame {
    type Target = [u8];
    fn deref(&self) -> &[u8] {
        if self.len == 0 { &[] } else { unsafe { core::slice::from_raw_parts(self.ptr, self.len) } }
    }
}

impl core::ops::DerefMut for PinnedFrame {
    fn deref_mut(&mut self) -> &mut [u8] {
        if self.len == 0 { &mut [] } else { unsafe { core::slice::from_raw_parts_mut(self.ptr, self.len) } }
    }
}

impl Drop for PinnedFrame {
    fn drop(&mut self) {
        if !self.ptr.is_null() {
            unsafe { sys::jpegenc_host_free(self.ptr as *mut core::ffi::c_void) };
        }
    }
}

/// Page-locks a buffer the caller keeps (a capture ring, a reused `Vec<u8>`) for as long as the guard lives: frames inside
/// it are uploaded in place.  Registering costs about one copy of the range - it pays for buffers that are reused.
pub struct PinnedRegistration<'a> {
    range: &'a mut [u8],
}

impl<'a> PinnedRegistration<'a> {
    pub fn new(range: &'a mut [u8]) -> Result<PinnedRegistration<'a>, EncodingError> {
        let status = unsafe { sys::jpegenc_host_register(range.as_mut_ptr() as *mut core::ffi::c_void, range.len()) };
        if status != sys::JPEGENC_OK {
            return Err(EncodingError::Write(alloc::string::String::from("page-locking the range failed")));
        }
        Ok(PinnedRegistration { range })
    }
    pub fn as_slice(&self) -> &[u8] {
        self.range
    }
    pub fn as_mut_slice(&mut self) -> &mut [u8] {
        self.range
    }
}

impl<'a> Drop for PinnedRegistration<'a> {
    fn drop(&mut self) {
        unsafe { sys::jpegenc_host_unregister(self.range.as_mut_ptr() as *mut core::ffi::c_void) };
    }
}

#[cfg(feature = "std")]
impl Encoder<std::io::BufWriter<std::fs::File>> {
    /// Create a new encoder that writes into a file (encoder.rs:1204-1219)
    pub fn new_file<P: AsRef<std::path::Path>>(path: P, quality: u8) -> Result<Encoder<std::io::BufWriter<std::fs::File>>, EncodingError> {
        let file = std::fs::File::create(path)?;
        let buf = std::io::BufWriter::new(file);
        Ok(Self::new(buf, quality))
    }
}

#[cfg(all(test, feature = "std"))]
mod tests {
    //! The reference's own API-level tests that need no decoder (src/lib.rs:484-553); they need an MI355X.
    use super::*;

    #[test]
    fn test_quantization_and_settings_round_trip() {
        let mut e = Encoder::new(Vec::new(), 85);
        assert_eq!(e.sampling_factor(), SamplingFactor::F_2_2);
        e.set_sampling_factor(SamplingFactor::R_4_2_2);
        assert_eq!(e.sampling_factor(), SamplingFactor::R_4_2_2);
        e.set_progressive(true);
        assert_eq!(e.progressive_scans(), Some(4));
        e.set_restart_interval(32);
        assert_eq!(e.restart_interval(), Some(32));
        e.set_density(PixelDensity::dpi(300));
        assert_eq!(e.density(), PixelDensity { density: (300, 300), unit: PixelDensityUnit::Inches });
    }

    #[test]
    fn test_app_segment_errors() {
        let mut e = Encoder::new(Vec::new(), 100);
        assert!(matches!(e.add_app_segment(0, Vec::new()), Err(EncodingError::InvalidAppSegment(0))));
        assert!(matches!(e.add_app_segment(16, Vec::new()), Err(EncodingError::InvalidAppSegment(16))));
        assert!(matches!(e.add_app_segment(1, alloc::vec![0u8; 65534]), Err(EncodingError::AppSegmentTooLarge(65534))));
    }

    #[test]
    fn test_bad_image_data_and_zero_dimensions() {
        let e = Encoder::new(Vec::new(), 100);
        assert!(matches!(e.encode(&[0u8; 11], 2, 2, ColorType::Rgb), Err(EncodingError::BadImageData { length: 11, required: 12 })));
        let e = Encoder::new(Vec::new(), 100);
        assert!(matches!(e.encode(&[], 0, 0, ColorType::Rgb), Err(EncodingError::ZeroImageDimensions { width: 0, height: 0 })));
    }

    #[test]
    fn test_encode_one_pixel() {
        if Encoder::<Vec<u8>>::device_count() == 0 {
            return;
        }
        let mut out = Vec::new();
        Encoder::new(&mut out, 100).encode(&[0xfb, 0x15, 0x15], 1, 1, ColorType::Rgb).unwrap();
        assert_eq!(&out[..2], &[0xFF, 0xD8]);
        assert_eq!(&out[out.len() - 2..], &[0xFF, 0xD9]);
    }
}
