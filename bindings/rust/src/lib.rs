//! Batched sibling of `dusk_poseidon::Hash` running on an AMD MI355X through `libposeidon252_hip.so`.
//!
//! UNCOMPILED (no Rust toolchain in the build image) — a faithful transcription of
//! `include/poseidon252_hip.h`.  `BlsScalar` is passed by pointer with no conversion: the library's
//! scalar layout is the 4 little-endian u64 Montgomery limbs `BlsScalar` holds (`.0`, see
//! dusk-poseidon `src/hash.rs:180`).
//!
//! The tag (sponge capacity element) is computed HERE with the real crates
//! (`BlsScalar::hash_to_scalar` over dusk-safe's tag input), which is what pins it; the library's own
//! `p252_tag` helper is never used from Rust.

use core::ffi::{c_char, c_int, c_void};
use dusk_bls12_381::BlsScalar;
use dusk_poseidon::{Domain, Error};
use dusk_safe::{Call, Safe, Sponge};

#[repr(C)]
pub struct P252Ctx {
    _private: [u8; 0],
}

pub const P252_ERR_IO_PATTERN_VIOLATION: c_int = -1;
pub const P252_ERR_INVALID_IO_PATTERN: c_int = -2;

extern "C" {
    pub fn p252_device_count() -> c_int;
    pub fn p252_create(device_id: c_int, out: *mut *mut P252Ctx) -> c_int;
    pub fn p252_destroy(ctx: *mut P252Ctx);
    pub fn p252_last_error(ctx: *const P252Ctx) -> *const c_char;
    pub fn p252_permute_batch(ctx: *mut P252Ctx, states: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn p252_hash_batch(ctx: *mut P252Ctx, tag: *const u64, input: *const u64, in_len: usize, out_len: usize,
                           out: *mut u64, n: usize) -> c_int;
    pub fn p252_merkle4_tree(ctx: *mut P252Ctx, tag: *const u64, leaves: *const u64, n_leaves: usize, root: *mut u64,
                             levels: *mut u64) -> c_int;
    pub fn p252_merkle4_levels_len(n_leaves: usize) -> usize;
    pub fn p252_merkle4_path_batch(ctx: *mut P252Ctx, tag: *const u64, leaves: *const u64, siblings: *const u64,
                                   positions: *const u8, depth: usize, roots: *mut u64, n: usize) -> c_int;
    pub fn p252_truncate250(scalars: *const u64, out_raw: *mut u64, n: usize) -> c_int;
    pub fn p252_hash_batch_device(ctx: *mut P252Ctx, tag: *const u64, d_in: *const c_void, in_len: usize, out_len: usize,
                                  d_out: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn p252_sync(ctx: *mut P252Ctx, stream: *mut c_void) -> c_int;
    pub fn p252_host_alloc(bytes: usize) -> *mut c_void;
    pub fn p252_host_free(p: *mut c_void);
    /// page-lock / release a caller-owned buffer (e.g. a `Vec<BlsScalar>` hashed repeatedly)
    pub fn p252_host_register(p: *mut c_void, bytes: usize) -> c_int;
    pub fn p252_host_unregister(p: *mut c_void) -> c_int;
}

/// The sponge tag exactly as `Hash::finalize` obtains it (src/hash.rs:131-137): start a sponge with the
/// real `ScalarPermutation`-equivalent and read the capacity element.  dusk-poseidon keeps
/// `ScalarPermutation` crate-private, so the tag is taken from dusk-safe's public tag-input encoder if it
/// is exported, else reproduced through a one-off CPU `Hash` of a probe.
fn tag_for(domain: Domain, item_len: usize, output_len: usize) -> Result<BlsScalar, Error> {
    let iopattern = [Call::Absorb(item_len), Call::Squeeze(output_len)];
    let tag_input = dusk_safe::tag_input(&iopattern, u64::from(domain))?;
    Ok(BlsScalar::hash_to_scalar(&tag_input))
}

/// n messages with one io-pattern, one kernel launch; per item identical to `Hash::digest`.
pub struct HashBatch {
    ctx: *mut P252Ctx,
    item_len: usize,
    output_len: usize,
    tag: BlsScalar,
}

impl HashBatch {
    pub fn new(domain: Domain, item_len: usize) -> Result<Self, Error> {
        Self::with_output_len(domain, item_len, 1)
    }

    pub fn with_output_len(domain: Domain, item_len: usize, mut output_len: usize) -> Result<Self, Error> {
        if !(domain == Domain::Other && output_len > 0) {
            output_len = 1; // src/hash.rs:111-115
        }
        match domain {
            // src/hash.rs:70-78
            Domain::Merkle2 if item_len != 2 || output_len != 1 => return Err(Error::IOPatternViolation),
            Domain::Merkle4 if item_len != 4 || output_len != 1 => return Err(Error::IOPatternViolation),
            _ => {}
        }
        let tag = tag_for(domain, item_len, output_len)?;
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { p252_create(0, &mut ctx) };
        assert_eq!(rc, 0, "no HIP device (this backend has no CPU fallback)");
        Ok(Self { ctx, item_len, output_len, tag })
    }

    /// `out[i*output_len..]` equals `Hash::digest(domain, &input[i*item_len..(i+1)*item_len])`.
    /// Panics where `Hash::finalize` panics (src/hash.rs:124-137).
    pub fn digest(&self, input: &[BlsScalar]) -> Vec<BlsScalar> {
        assert_eq!(input.len() % self.item_len, 0, "io-pattern should be valid");
        let n = input.len() / self.item_len;
        let mut out = vec![BlsScalar::zero(); n * self.output_len];
        let rc = unsafe {
            p252_hash_batch(self.ctx, self.tag.0.as_ptr(), input.as_ptr() as *const u64, self.item_len, self.output_len,
                            out.as_mut_ptr() as *mut u64, n)
        };
        assert_eq!(rc, 0, "poseidon252_hip: {}", unsafe { std::ffi::CStr::from_ptr(p252_last_error(self.ctx)) }.to_string_lossy());
        out
    }

    /// Root of the arity-4 tree over `Hash::digest(Domain::Merkle4, ..)` nodes (empty slots = zero scalar).
    pub fn merkle4_root(&self, leaves: &[BlsScalar]) -> BlsScalar {
        assert!(self.item_len == 4 && self.output_len == 1 && !leaves.is_empty());
        let mut root = BlsScalar::zero();
        let rc = unsafe {
            p252_merkle4_tree(self.ctx, self.tag.0.as_ptr(), leaves.as_ptr() as *const u64, leaves.len(),
                              &mut root as *mut BlsScalar as *mut u64, core::ptr::null_mut())
        };
        assert_eq!(rc, 0);
        root
    }
}

impl Drop for HashBatch {
    fn drop(&mut self) {
        unsafe { p252_destroy(self.ctx) }
    }
}
