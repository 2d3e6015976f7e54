//! Batched sibling of `dusk_poseidon::Hash` / `encrypt` / `decrypt` running on AMD MI355X GPUs through
//! `libposeidon252_hip.so` (include/poseidon252_hip.h; raw declarations in `sys`, generated from that header).
//!
//! NOT COMPILED in the repository's build image (no Rust toolchain, no crates.io): written against the public APIs the
//! reference itself uses — `dusk_safe::{Call, Safe, Sponge, Encryption, encrypt}` exactly as in
//! dusk-poseidon `src/hades.rs:63-125`, `src/hash.rs:128-155`, `src/encryption.rs:62-95`.  `RUN.md` has the commands.
//!
//! `BlsScalar` is passed by pointer with no conversion: the library's scalar layout is the 4 little-endian u64
//! Montgomery limbs `BlsScalar` holds (`.0`, dusk-poseidon `src/hash.rs:180`).
//!
//! The TAG (sponge capacity element) is obtained HERE from the real crates, never from the library's `p252_tag`:
//! `TagProbe` implements `Safe<BlsScalar, 5>` with `tag() = BlsScalar::hash_to_scalar(input)` — the body of
//! `ScalarPermutation::tag` (src/hades/permutation/scalar.rs:29-31) — and records both the tag input dusk-safe hands it
//! and the result; `Sponge::start` (or `dusk_safe::encrypt`) is then run once on the probe.  That is the reference's own
//! `Test` pattern (src/hades.rs:63-92) with the tag kept instead of zeroed, and it uses nothing crate-private.

pub mod sys;

use core::ffi::c_int;
use std::cell::RefCell;
use std::rc::Rc;

use dusk_bls12_381::BlsScalar;
use dusk_poseidon::{Domain, Error};
use dusk_safe::{Call, Safe, Sponge};

use sys::*;

/// What a probe run saw: the bytes dusk-safe hashed into the tag, the tag, and how often `permute` was called.
#[derive(Default, Debug, Clone)]
pub struct ProbeLog {
    pub tag_input: Vec<u8>,
    pub tag: Option<BlsScalar>,
    pub permutations: usize,
}

/// `Safe` implementation that records instead of permuting (see the crate docs).
#[derive(Clone)]
pub struct TagProbe(pub Rc<RefCell<ProbeLog>>);

impl TagProbe {
    pub fn new() -> (Self, Rc<RefCell<ProbeLog>>) {
        let log = Rc::new(RefCell::new(ProbeLog::default()));
        (Self(log.clone()), log)
    }
}

impl Safe<BlsScalar, 5> for TagProbe {
    fn permute(&mut self, _state: &mut [BlsScalar; 5]) {
        self.0.borrow_mut().permutations += 1;
    }
    fn tag(&mut self, input: &[u8]) -> BlsScalar {
        let t = BlsScalar::hash_to_scalar(input); // src/hades/permutation/scalar.rs:29-31
        let mut log = self.0.borrow_mut();
        log.tag_input = input.to_vec();
        log.tag = Some(t);
        t
    }
    fn add(&mut self, right: &BlsScalar, left: &BlsScalar) -> BlsScalar {
        right + left
    }
}

#[cfg(feature = "encryption")]
impl dusk_safe::Encryption<BlsScalar, 5> for TagProbe {
    fn subtract(&mut self, minuend: &BlsScalar, subtrahend: &BlsScalar) -> BlsScalar {
        minuend - subtrahend
    }
    fn is_equal(&mut self, lhs: &BlsScalar, rhs: &BlsScalar) -> bool {
        lhs == rhs
    }
}

/// The tag `Hash::finalize` would use for `domain` with one absorb call per entry of `absorb_lens` and `output_len`
/// squeezed elements (src/hash.rs:62-85 builds exactly this io-pattern; src/hash.rs:131-137 starts the sponge).
pub fn hash_tag(domain: Domain, absorb_lens: &[usize], output_len: usize) -> Result<(BlsScalar, ProbeLog), Error> {
    let total: usize = absorb_lens.iter().sum();
    match domain {
        // src/hash.rs:70-78
        Domain::Merkle2 if total != 2 || output_len != 1 => return Err(Error::IOPatternViolation),
        Domain::Merkle4 if total != 4 || output_len != 1 => return Err(Error::IOPatternViolation),
        _ => {}
    }
    let mut iopattern: Vec<Call> = absorb_lens.iter().map(|l| Call::Absorb(*l)).collect();
    iopattern.push(Call::Squeeze(output_len));
    let (probe, log) = TagProbe::new();
    let _sponge = Sponge::start(probe, iopattern, u64::from(domain))?;
    let log = log.borrow().clone();
    Ok((log.tag.expect("Sponge::start derives the tag"), log))
}

/// The tag `dusk_poseidon::encrypt` uses for a message of `message_len` scalars: dusk_safe::encrypt run once on the
/// probe (identity permutation, so the cipher it returns is meaningless; the recorded tag is not).  The log's
/// `tag_input` spells out the io-pattern dusk-safe really uses — which of the library's two variants it is.
#[cfg(feature = "encryption")]
pub fn encryption_tag(message_len: usize) -> Result<(BlsScalar, ProbeLog), Error> {
    let (probe, log) = TagProbe::new();
    let message = vec![BlsScalar::zero(); message_len];
    let _ = dusk_safe::encrypt(probe, Domain::Encryption, &message, &[BlsScalar::zero(), BlsScalar::zero()], &BlsScalar::zero())?;
    let log = log.borrow().clone();
    Ok((log.tag.expect("dusk_safe::encrypt derives the tag"), log))
}

fn last_error(ctx: *const P252Ctx) -> String {
    unsafe { std::ffi::CStr::from_ptr(p252_last_error(ctx)) }.to_string_lossy().into_owned()
}

/// One GPU context (`p252_ctx`): bound to one device, used by one thread at a time.
pub struct Context(*mut P252Ctx);

impl Context {
    pub fn new(device_id: i32) -> Self {
        // argument lists changed between library versions under unchanged names: never call into another interface
        let abi = unsafe { p252_abi_version() };
        assert_eq!(abi, P252_ABI_VERSION, "libposeidon252_hip.so implements ABI version {abi}, this crate was generated for {P252_ABI_VERSION}");
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { p252_create(device_id as c_int, &mut ctx) };
        assert_eq!(rc, P252_OK, "poseidon252_hip: {} (this backend has no CPU fallback)", last_error(core::ptr::null()));
        Self(ctx)
    }
    pub fn device_count() -> usize {
        unsafe { p252_device_count() as usize }
    }
    fn check(&self, rc: c_int) {
        assert_eq!(rc, P252_OK, "poseidon252_hip: {}", last_error(self.0));
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { p252_destroy(self.0) }
    }
}

/// n messages with one io-pattern, one kernel launch; per item identical to `Hash::digest` / `Hash::finalize`.
pub struct HashBatch {
    ctxs: Vec<Context>,
    item_len: usize,
    output_len: usize,
    tag: BlsScalar,
}

impl HashBatch {
    pub fn new(domain: Domain, item_len: usize) -> Result<Self, Error> {
        Self::with_output_len(domain, item_len, 1)
    }

    /// `output_len` is honoured only for `Domain::Other` and `> 0`, like `Hash::output_len` (src/hash.rs:111-115).
    pub fn with_output_len(domain: Domain, item_len: usize, mut output_len: usize) -> Result<Self, Error> {
        if !(domain == Domain::Other && output_len > 0) {
            output_len = 1;
        }
        let (tag, _) = hash_tag(domain, &[item_len], output_len)?;
        Ok(Self { ctxs: vec![Context::new(0)], item_len, output_len, tag })
    }

    /// One context per visible GPU: `digest` and `merkle4_root` then shard across all of them inside the library.
    pub fn on_all_devices(mut self) -> Self {
        self.ctxs = (0..Context::device_count()).map(|d| Context::new(d as i32)).collect();
        self
    }

    pub fn tag(&self) -> BlsScalar {
        self.tag
    }

    /// `out[i*output_len..]` equals `Hash::digest(domain, &input[i*item_len..(i+1)*item_len])`.
    /// Panics where `Hash::finalize` panics (src/hash.rs:124-137).
    pub fn digest(&self, input: &[BlsScalar]) -> Vec<BlsScalar> {
        assert_eq!(input.len() % self.item_len, 0, "io-pattern should be valid");
        let n = input.len() / self.item_len;
        let mut out = vec![BlsScalar::zero(); n * self.output_len];
        let ptrs: Vec<*mut P252Ctx> = self.ctxs.iter().map(|c| c.0).collect();
        let rc = unsafe {
            p252_hash_batch_multi(ptrs.as_ptr(), ptrs.len(), self.tag.0.as_ptr(), input.as_ptr() as *const u64, self.item_len,
                                  self.output_len, out.as_mut_ptr() as *mut u64, n)
        };
        self.ctxs[0].check(rc);
        out
    }

    /// `Hash::digest_truncated` per item (src/hash.rs:203-210) in ONE kernel launch: the raw limbs the reference hands to
    /// `JubJubScalar::from_raw` (src/hash.rs:180) — canonical value & (2^250 - 1), truncated by the digest kernel's output stage.
    pub fn digest_truncated_raw(&self, input: &[BlsScalar]) -> Vec<[u64; 4]> {
        assert_eq!(input.len() % self.item_len, 0, "io-pattern should be valid");
        let n = input.len() / self.item_len;
        let mut out = vec![[0u64; 4]; n * self.output_len];
        let rc = unsafe {
            p252_hash_batch_truncated(self.ctxs[0].0, self.tag.0.as_ptr(), input.as_ptr() as *const u64, self.item_len, self.output_len,
                                      out.as_mut_ptr() as *mut u64, n)
        };
        self.ctxs[0].check(rc);
        out
    }

    /// Root of the arity-4 tree over `Hash::digest(Domain::Merkle4, ..)` nodes (empty slots = zero scalar).  With
    /// several devices `leaves.len()` must be `devices * 4^k` (one complete subtree per GPU, roots gathered on the host).
    pub fn merkle4_root(&self, leaves: &[BlsScalar]) -> BlsScalar {
        assert!(self.item_len == 4 && self.output_len == 1 && !leaves.is_empty());
        let mut root = BlsScalar::zero();
        let rc = if self.ctxs.len() == 1 {
            unsafe {
                p252_merkle4_tree(self.ctxs[0].0, self.tag.0.as_ptr(), leaves.as_ptr() as *const u64, leaves.len(),
                                  &mut root as *mut BlsScalar as *mut u64, core::ptr::null_mut())
            }
        } else {
            let ptrs: Vec<*mut P252Ctx> = self.ctxs.iter().map(|c| c.0).collect();
            unsafe {
                p252_merkle4_tree_multi(ptrs.as_ptr(), ptrs.len(), self.tag.0.as_ptr(), leaves.as_ptr() as *const u64, leaves.len(),
                                        &mut root as *mut BlsScalar as *mut u64)
            }
        };
        self.ctxs[0].check(rc);
        root
    }

    /// Root of the arity-2 tree over `Hash::digest(Domain::Merkle2, ..)` nodes (a ragged level's missing sibling is the zero
    /// scalar, src/hash.rs:27-31); `self` must be a `HashBatch::new(Domain::Merkle2, 2)`.
    pub fn merkle2_root(&self, leaves: &[BlsScalar]) -> BlsScalar {
        assert!(self.item_len == 2 && self.output_len == 1 && !leaves.is_empty());
        let mut root = BlsScalar::zero();
        let rc = unsafe {
            p252_merkle2_tree(self.ctxs[0].0, self.tag.0.as_ptr(), leaves.as_ptr() as *const u64, leaves.len(),
                              &mut root as *mut BlsScalar as *mut u64, core::ptr::null_mut())
        };
        self.ctxs[0].check(rc);
        root
    }

    /// The root each of n arity-4 openings re-hashes to (`poseidon-merkle`'s `Opening::verify` compares it with the tree's root):
    /// `siblings` holds n x depth x 3 scalars, `positions` n x depth bytes in 0..3 (the slot of the value coming from below).
    pub fn merkle4_path_roots(&self, leaves: &[BlsScalar], siblings: &[BlsScalar], positions: &[u8], depth: usize) -> Vec<BlsScalar> {
        assert!(self.item_len == 4 && self.output_len == 1);
        let n = leaves.len();
        assert!(siblings.len() == n * depth * 3 && positions.len() == n * depth);
        let mut roots = vec![BlsScalar::zero(); n];
        let rc = unsafe {
            p252_merkle4_path_batch(self.ctxs[0].0, self.tag.0.as_ptr(), leaves.as_ptr() as *const u64, siblings.as_ptr() as *const u64,
                                    positions.as_ptr(), depth, roots.as_mut_ptr() as *mut u64, n)
        };
        self.ctxs[0].check(rc);
        roots
    }
}

/// Batched `dusk_poseidon::encrypt` (src/encryption.rs:62-76): `messages` holds n messages of `message_len` scalars,
/// `secrets` n pairs `[shared.get_u(), shared.get_v()]`, `nonces` n scalars; returns n ciphers of `message_len + 1`.
/// `variant`: `sys::P252_CRYPT_STREAM` or `sys::P252_CRYPT_DUPLEX` — tests/parity.rs establishes which one is dusk-safe's.
#[cfg(feature = "encryption")]
pub fn encrypt_batch(ctx: &Context, variant: c_int, messages: &[BlsScalar], message_len: usize, secrets: &[[BlsScalar; 2]],
                     nonces: &[BlsScalar]) -> Result<Vec<BlsScalar>, Error> {
    let n = nonces.len();
    assert!(messages.len() == n * message_len && secrets.len() == n);
    let (tag, _) = encryption_tag(message_len)?;
    let mut out = vec![BlsScalar::zero(); n * (message_len + 1)];
    let rc = unsafe {
        p252_encrypt_batch(ctx.0, variant, tag.0.as_ptr(), messages.as_ptr() as *const u64, secrets.as_ptr() as *const u64,
                           nonces.as_ptr() as *const u64, message_len, out.as_mut_ptr() as *mut u64, n)
    };
    ctx.check(rc);
    Ok(out)
}

/// Batched `dusk_poseidon::decrypt` (src/encryption.rs:81-95): item i is `Err(Error::DecryptionFailed)` when its MAC
/// does not verify.
#[cfg(feature = "encryption")]
pub fn decrypt_batch(ctx: &Context, variant: c_int, ciphers: &[BlsScalar], message_len: usize, secrets: &[[BlsScalar; 2]],
                     nonces: &[BlsScalar]) -> Result<Vec<Result<Vec<BlsScalar>, Error>>, Error> {
    let n = nonces.len();
    assert!(ciphers.len() == n * (message_len + 1) && secrets.len() == n);
    let (tag, _) = encryption_tag(message_len)?;
    let mut msgs = vec![BlsScalar::zero(); n * message_len];
    let mut ok = vec![0u8; n];
    let rc = unsafe {
        p252_decrypt_batch(ctx.0, variant, tag.0.as_ptr(), ciphers.as_ptr() as *const u64, secrets.as_ptr() as *const u64,
                           nonces.as_ptr() as *const u64, message_len, msgs.as_mut_ptr() as *mut u64, ok.as_mut_ptr(), n)
    };
    ctx.check(rc);
    Ok((0..n)
        .map(|i| if ok[i] == 1 { Ok(msgs[i * message_len..(i + 1) * message_len].to_vec()) } else { Err(Error::DecryptionFailed) })
        .collect())
}

/// `BlsScalar::to_bytes` for a slice: the 32 little-endian bytes of each canonical value (host-side twin of
/// `p252_to_bytes_device`; the reference uses the pair at src/hades/round_constants.rs:66-67).
pub fn to_bytes_batch(scalars: &[BlsScalar]) -> Vec<[u8; 32]> {
    let mut out = vec![[0u8; 32]; scalars.len()];
    let rc = unsafe { p252_to_bytes(scalars.as_ptr() as *const u64, out.as_mut_ptr() as *mut u8, scalars.len()) };
    assert_eq!(rc, P252_OK);
    out
}

/// `BlsScalar::from_bytes` for a slice of 32-byte records: `None` where the value is not below the modulus
/// (`from_bytes` returns an error there).
pub fn from_bytes_batch(bytes: &[[u8; 32]]) -> Vec<Option<BlsScalar>> {
    let mut limbs = vec![BlsScalar::zero(); bytes.len()];
    let mut ok = vec![0u8; bytes.len()];
    let rc = unsafe { p252_from_bytes(bytes.as_ptr() as *const u8, limbs.as_mut_ptr() as *mut u64, ok.as_mut_ptr(), bytes.len()) };
    assert_eq!(rc, P252_OK);
    limbs.into_iter().zip(ok).map(|(s, o)| if o == 1 { Some(s) } else { None }).collect()
}
