//! MI355X block coder behind `redux::compress` / `redux::decompress`.
//!
//! Drop this file into the reference crate as `src/hip.rs` and declare it from `src/lib.rs` with
//! `pub mod hip;`.  It binds the C ABI of `include/redux_hip.h` (libredux_hip.so) and contains no
//! coding logic: only argument marshalling and the mapping of status codes onto `redux::Error`
//! (src/lib.rs:57-64).  Rust 2015 edition, like the crate (`try!`, bare trait objects).
//!
//! Acceleration is selected BY API: `Box<Model>` is an open trait object the GPU cannot call back
//! into, so these functions take `&Parameters` and always code with the semantics of
//! `AdaptiveTreeModel::new(params.clone())` (src/model/adaptive_tree.rs:36).  The existing
//! `compress` / `decompress` / `Codec` / `Model` surface is untouched.
//!
//! `tests/test_rust_shim_cpu.py` in the redux_amd repository parses the `extern "C"` block below and
//! checks every declaration against `include/redux_hip.h` (name, arity, pointer-vs-integer and
//! integer width of each argument and of the return type).
use std::io;
use std::os::raw::c_int;
use std::ptr;

use super::model::Parameters;
use super::{Error, Result};

/// `redux_params` of include/redux_hip.h: the three arguments of `Parameters::new`
/// (src/model/mod.rs:63); the library derives the other eight fields itself.
#[repr(C)]
pub struct ReduxParams {
    symbol_bits: u32,
    freq_bits: u32,
    code_bits: u32,
}

extern "C" {
    fn redux_params_check(symbol_bits: u32, freq_bits: u32, code_bits: u32) -> c_int;
    fn redux_device_supports(p: *const ReduxParams) -> c_int;
    fn redux_block_count(in_len: u64, block_size: u32) -> u64;
    fn redux_encode_bound(p: *const ReduxParams, in_len: u64, block_size: u32) -> u64;
    fn redux_encode_blocks(p: *const ReduxParams, input: *const u8, in_len: u64, block_size: u32,
                           out: *mut u8, out_cap: u64, out_offsets: *mut u64,
                           block_status: *mut i32) -> c_int;
    fn redux_decode_blocks(p: *const ReduxParams, input: *const u8, in_offsets: *const u64,
                           nblocks: u64, block_size: u32, out: *mut u8, out_cap: u64,
                           out_sizes: *mut u32, block_status: *mut i32) -> c_int;
    fn redux_block_count_v(in_len: *const u64, ninputs: u64, block_size: u32) -> u64;
    fn redux_encode_slot_bytes(p: *const ReduxParams, block_size: u32) -> u64;
    fn redux_encode_blocks_v(p: *const ReduxParams, input: *const u8, in_off: *const u64, in_len: *const u64,
                             ninputs: u64, block_size: u32, out: *mut u8, out_cap: u64,
                             out_offsets: *mut u64, block_status: *mut i32) -> c_int;
    fn redux_decode_blocks_v(p: *const ReduxParams, input: *const u8, in_offsets: *const u64, out: *mut u8,
                             out_off: *const u64, out_len: *const u64, ninputs: u64, block_size: u32,
                             out_sizes: *mut u32, block_status: *mut i32) -> c_int;
    fn redux_compress(p: *const ReduxParams, input: *const u8, in_len: u64, out: *mut u8,
                      out_cap: u64, bytes_in: *mut u64, bytes_out: *mut u64) -> c_int;
    fn redux_decompress(p: *const ReduxParams, input: *const u8, in_len: u64, out: *mut u8,
                        out_cap: u64, bytes_in: *mut u64, bytes_out: *mut u64) -> c_int;
    fn redux_host_release() -> c_int;
    fn redux_host_set_devices(device_ids: *const i32, n: u32) -> c_int;
}

/// Status codes of include/redux_hip.h -> `redux::Error` (src/lib.rs:57-64).
fn status(st: c_int) -> Result<()> {
    match st {
        0 => Ok(()),
        1 => Err(Error::Eof),
        2 => Err(Error::InvalidInput),
        4 => Err(Error::IoError(io::Error::new(io::ErrorKind::Other, "redux_hip: output too small"))),
        5 => Err(Error::IoError(io::Error::new(io::ErrorKind::Other, "redux_hip: parameters unsupported on the device"))),
        _ => Err(Error::IoError(io::Error::new(io::ErrorKind::Other, "redux_hip: HIP runtime error"))),
    }
}

fn c_params(p: &Parameters) -> ReduxParams {
    ReduxParams { symbol_bits: p.symbol_bits as u32, freq_bits: p.freq_bits as u32, code_bits: p.code_bits as u32 }
}

/// `true` when the device implements these parameters (`symbol_bits <= 16`); a caller may route
/// everything else to the pure-Rust `redux::compress`.  The library itself has no CPU fallback.
pub fn supports(p: &Parameters) -> bool {
    let cp = c_params(p);
    unsafe {
        redux_params_check(cp.symbol_bits, cp.freq_bits, cp.code_bits) == 0 && redux_device_supports(&cp) == 0
    }
}

/// The library keeps one lazily built, mutex-guarded context per GPU for these calls (chunk slots in HBM, pinned
/// staging, streams), grown on demand and reused: calls from several threads are safe and run one after the other.
/// `release` frees it (it is rebuilt by the next call); a long-lived process that is done coding may call it.
pub fn release() {
    unsafe {
        redux_host_release();
    }
}

/// Several GPUs behind `compress_blocks` / `decompress_blocks`: every later call deals its chunks round-robin over one
/// context per entry of `device_ids` (each fed over its own PCIe link; the devices exchange nothing).  An empty slice
/// goes back to the default, HIP's current device.
pub fn set_devices(device_ids: &[i32]) -> Result<()> {
    unsafe { status(redux_host_set_devices(if device_ids.is_empty() { ptr::null() } else { device_ids.as_ptr() }, device_ids.len() as u32)) }
}

/// One `redux::compress` per block of `block_size` bytes, all blocks coded in parallel on the GPU.
/// Returns the dense streams and `nblocks + 1` offsets; block `b` is
/// `out[offsets[b] as usize..offsets[b + 1] as usize]` and is byte-identical to
/// `redux::compress(&mut &data[b * block_size..][..len_b], .., AdaptiveTreeModel::new(p.clone()))`.
pub fn compress_blocks(data: &[u8], block_size: u32, p: &Parameters) -> Result<(Vec<u8>, Vec<u64>)> {
    if block_size == 0 {
        return Err(Error::InvalidInput);
    }
    let cp = c_params(p);
    unsafe {
        try!(status(redux_device_supports(&cp)));
        let nb = redux_block_count(data.len() as u64, block_size) as usize;
        let cap = redux_encode_bound(&cp, data.len() as u64, block_size) as usize;
        let mut out = vec![0u8; cap];
        let mut offs = vec![0u64; nb + 1];
        try!(status(redux_encode_blocks(&cp, data.as_ptr(), data.len() as u64, block_size,
                                        out.as_mut_ptr(), cap as u64, offs.as_mut_ptr(), ptr::null_mut())));
        out.truncate(offs[nb] as usize);
        Ok((out, offs))
    }
}

/// Inverse of `compress_blocks`: block `b` of the result is `out[b * block_size..][..sizes[b]]`.
pub fn decompress_blocks(streams: &[u8], offsets: &[u64], block_size: u32, p: &Parameters) -> Result<(Vec<u8>, Vec<u32>)> {
    if block_size == 0 || offsets.is_empty() || offsets[offsets.len() - 1] as usize > streams.len() {
        return Err(Error::InvalidInput);
    }
    let cp = c_params(p);
    let nb = offsets.len() - 1;
    unsafe {
        try!(status(redux_device_supports(&cp)));
        let mut out = vec![0u8; nb * block_size as usize];
        let mut sizes = vec![0u32; nb];
        try!(status(redux_decode_blocks(&cp, streams.as_ptr(), offsets.as_ptr(), nb as u64, block_size,
                                        out.as_mut_ptr(), out.len() as u64, sizes.as_mut_ptr(), ptr::null_mut())));
        Ok((out, sizes))
    }
}

/// Many independent inputs in ONE launch -- what the reference's corpus harness does file by file
/// (tests/corpora.rs:32-85).  Every input is cut into blocks of `block_size` on its own (ragged tail per
/// input, an empty input is one empty block); blocks are numbered input by input.  Returns the dense
/// streams, `nblocks + 1` offsets and, per input, the number of its first block (`inputs.len() + 1`
/// entries).  Block streams equal those of `compress_blocks` called once per input.
pub fn compress_blocks_v(inputs: &[&[u8]], block_size: u32, p: &Parameters) -> Result<(Vec<u8>, Vec<u64>, Vec<u64>)> {
    if block_size == 0 || inputs.is_empty() {
        return Err(Error::InvalidInput);
    }
    let cp = c_params(p);
    let lens: Vec<u64> = inputs.iter().map(|x| x.len() as u64).collect();
    let mut offs_in = vec![0u64; inputs.len()];
    let mut flat = Vec::with_capacity(lens.iter().sum::<u64>() as usize);
    let mut first = vec![0u64; inputs.len() + 1];
    for (i, x) in inputs.iter().enumerate() {
        offs_in[i] = flat.len() as u64;
        flat.extend_from_slice(x);
        first[i + 1] = first[i] + unsafe { redux_block_count(lens[i], block_size) };
    }
    unsafe {
        try!(status(redux_device_supports(&cp)));
        let nb = redux_block_count_v(lens.as_ptr(), lens.len() as u64, block_size) as usize;
        let cap = nb * redux_encode_slot_bytes(&cp, block_size) as usize;
        let mut out = vec![0u8; cap];
        let mut offs = vec![0u64; nb + 1];
        try!(status(redux_encode_blocks_v(&cp, flat.as_ptr(), offs_in.as_ptr(), lens.as_ptr(), lens.len() as u64, block_size,
                                          out.as_mut_ptr(), cap as u64, offs.as_mut_ptr(), ptr::null_mut())));
        out.truncate(offs[nb] as usize);
        Ok((out, offs, first))
    }
}

/// Inverse of `compress_blocks_v`: `lengths[i]` is the decoded size of input `i`; returns the inputs back to back
/// (input `i` at the sum of the lengths before it).
pub fn decompress_blocks_v(streams: &[u8], offsets: &[u64], lengths: &[u64], block_size: u32, p: &Parameters) -> Result<Vec<u8>> {
    if block_size == 0 || lengths.is_empty() || offsets.is_empty() || offsets[offsets.len() - 1] as usize > streams.len() {
        return Err(Error::InvalidInput);
    }
    let cp = c_params(p);
    let mut out_off = vec![0u64; lengths.len()];
    for i in 1..lengths.len() {
        out_off[i] = out_off[i - 1] + lengths[i - 1];
    }
    let total = (out_off[lengths.len() - 1] + lengths[lengths.len() - 1]) as usize;
    unsafe {
        try!(status(redux_device_supports(&cp)));
        let nb = redux_block_count_v(lengths.as_ptr(), lengths.len() as u64, block_size) as usize;
        if nb + 1 != offsets.len() {
            return Err(Error::InvalidInput);
        }
        let mut out = vec![0u8; std::cmp::max(total, 1)];
        let mut sizes = vec![0u32; nb];
        try!(status(redux_decode_blocks_v(&cp, streams.as_ptr(), offsets.as_ptr(), out.as_mut_ptr(), out_off.as_ptr(),
                                          lengths.as_ptr(), lengths.len() as u64, block_size, sizes.as_mut_ptr(), ptr::null_mut())));
        out.truncate(total);
        Ok(out)
    }
}

/// Same signature shape and same bytes as `redux::compress` (src/lib.rs:102-109) with an
/// `AdaptiveTreeModel`: the whole stream is ONE block, coded by one GPU lane.  Correct, serial;
/// `compress_blocks` is the accelerated path.
pub fn compress(istream: &mut io::Read, ostream: &mut io::Write, p: &Parameters) -> Result<(u64, u64)> {
    let mut data = Vec::new();
    try!(istream.read_to_end(&mut data).map_err(Error::IoError));
    let cp = c_params(p);
    unsafe {
        try!(status(redux_device_supports(&cp)));
        let bs = if data.is_empty() { 1 } else { data.len() as u32 };
        let cap = redux_encode_bound(&cp, data.len() as u64, bs) as usize;
        let mut out = vec![0u8; cap];
        let (mut bi, mut bo) = (0u64, 0u64);
        try!(status(redux_compress(&cp, data.as_ptr(), data.len() as u64, out.as_mut_ptr(), cap as u64, &mut bi, &mut bo)));
        try!(ostream.write_all(&out[..bo as usize]).map_err(Error::IoError));
        Ok((bi, bo))
    }
}

/// `redux::decompress` (src/lib.rs:113-120).  The reference writes to an unbounded `io::Write`;
/// the C ABI wants a capacity, so the buffer grows (x8) until the stream fits or the ABI's
/// one-block limit is reached.  Returns (compressed bytes the reader fetched, bytes written).
pub fn decompress(istream: &mut io::Read, ostream: &mut io::Write, p: &Parameters) -> Result<(u64, u64)> {
    const LIMIT: usize = 0xFFFF_FF00;
    let mut data = Vec::new();
    try!(istream.read_to_end(&mut data).map_err(Error::IoError));
    let cp = c_params(p);
    let mut cap = std::cmp::max(64 * data.len(), 1 << 20);
    unsafe {
        try!(status(redux_device_supports(&cp)));
        loop {
            cap = std::cmp::min(cap, LIMIT);
            let mut out = vec![0u8; cap];
            let (mut bi, mut bo) = (0u64, 0u64);
            let st = redux_decompress(&cp, data.as_ptr(), data.len() as u64, out.as_mut_ptr(), cap as u64, &mut bi, &mut bo);
            if st == 4 && cap < LIMIT {
                cap *= 8;
                continue;
            }
            try!(status(st));
            try!(ostream.write_all(&out[..bo as usize]).map_err(Error::IoError));
            return Ok((bi, bo));
        }
    }
}
