/*
 * p252_oracle.h — CPU ORACLE for the Poseidon252 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's native (non-gadget) path
 *   dusk_poseidon::Hash / Domain  ->  dusk-safe Sponge  ->  Hades permutation  ->  BlsScalar
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (poseidon252_amd/csrc) never links, calls or falls back to it.
 *
 * Parity pinning:
 *   - field arithmetic, constant encoding, round schedule, MDS orientation and the sponge
 *     absorb/squeeze positions are PINNED by the reference's only known-answer test,
 *     src/hades.rs:94-162 (6 digests; tests/golden/hades_kat.json).
 *   - the TAG (dusk-safe 0.3 tag-input encoding + dusk-bls12_381 0.14 BlsScalar::hash_to_scalar,
 *     called at src/hades/permutation/scalar.rs:29-31) lives in un-vendored crates and is NOT
 *     covered by any reference test: "parity unpinned".  Every hashing entry point therefore takes
 *     the tag as an explicit input; p252o_tag() is a convenience flagged UNPINNED.
 *     Equally UNPINNED: the encryption call sequence (dusk_safe::encrypt) and the `&` of
 *     finalize_truncated (BitAnd of dusk-bls12_381).
 *   - how these three get pinned: bindings/rust/refgen (a crate that needs only cargo and the dusk
 *     crates — no GPU, no library of this repository) writes tests/golden/reference_fixtures.json from
 *     the reference itself; tests/test_reference_fixtures.py then compares THIS oracle with it (and the
 *     HIP library with it on the GPU).  The file cannot be made in the build image (no Rust toolchain);
 *     until it is committed those tests skip and the labels above stand.
 *
 * A scalar ("BlsScalar") is 4 little-endian u64 limbs holding the Montgomery residue a*2^256 mod p,
 * fully reduced in [0,p) — byte-identical to dusk_bls12_381::BlsScalar memory.
 */
#ifndef P252_ORACLE_H
#define P252_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P252O_WIDTH 5            /* src/hades.rs:34 */
#define P252O_FULL_ROUNDS 8      /* src/hades.rs:29 */
#define P252O_PARTIAL_ROUNDS 60  /* src/hades.rs:31 */

/* ---- field helpers (dusk-bls12_381 BlsScalar semantics) ---- */
/* from_raw: 4 LE u64 holding an integer v (any 256-bit value) -> Montgomery limbs of v mod p */
void p252o_from_raw(const uint64_t raw[4], uint64_t out[4]);
/* Montgomery limbs -> canonical integer limbs (what to_bytes() serialises, little endian) */
void p252o_to_canonical(const uint64_t mont[4], uint64_t out[4]);
void p252o_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void p252o_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
/* 1 if limbs are a fully reduced residue (< p) */
int p252o_is_reduced(const uint64_t a[4]);

/* ---- constants, as the reference builds them (round_constants.rs:26-54, mds_matrix.rs:17-39) ---- */
/* Montgomery limbs of ROUND_CONSTANTS[round][i] / MDS_MATRIX[row][col] */
void p252o_round_constant(int round, int i, uint64_t out[4]);
void p252o_mds(int row, int col, uint64_t out[4]);

/* ---- Hades permutation, reference schedule (src/hades/permutation.rs:105-123) ---- */
void p252o_permute(uint64_t state[P252O_WIDTH * 4]);
void p252o_permute_batch(const uint64_t *states, uint64_t *out, size_t n);

/* ---- SAFE sponge driven like Hash::finalize (src/hash.rs:128-155); tag explicit ---- */
/* one message: absorb `in_len` scalars (as ONE or several absorb calls — position rules are
 * identical), squeeze `out_len`.  Returns 0, or -1 on in_len==0 / out_len==0
 * (dusk-safe rejects such io-patterns -> the reference panics, hash.rs:134-137). */
int p252o_sponge(const uint64_t tag[4], const uint64_t *in, size_t in_len, uint64_t *out,
                 size_t out_len);
int p252o_hash_batch(const uint64_t tag[4], const uint64_t *in, size_t in_len, size_t out_len,
                     uint64_t *out, size_t n);
/* multithreaded variant for the cpu_baseline leg (one pthread per chunk) */
int p252o_hash_batch_mt(const uint64_t tag[4], const uint64_t *in, size_t in_len, size_t out_len,
                        uint64_t *out, size_t n, int threads);

/* the KAT's sponge usage (src/hades.rs:107-125): tag=0, absorb(n) then absorb(1)=[one], squeeze 1.
 * inputs are canonical little-endian 32-byte strings (from_hex_str), output canonical LE bytes. */
void p252o_kat_hash(const uint8_t *inputs_le32, size_t n, uint8_t out_le32[32]);

/* ---- arity-4 Merkle tree (no in-repo reference builder since 0.29.0, CHANGELOG.md:164-168;
 * composition defined in SURVEY §8a: node = digest(Merkle4,[c0..c3]); empty slots = zero) ---- */
/* n_leaves >= 1.  Levels are built WHILE more than one node remains (a single leaf is its own root,
 * so a 4^k-leaf tree costs exactly k levels); a level whose size is not a multiple of 4 is
 * zero-padded (hash.rs:22-26).  If levels != NULL it receives every computed level above
 * the leaves, concatenated bottom-up.  Returns number of permutations, or -1. */
long long p252o_merkle4_tree(const uint64_t tag[4], const uint64_t *leaves, size_t n_leaves,
                             uint64_t root[4], uint64_t *levels);

/* arity-2 tree over Domain::Merkle2 digests (hash.rs:27-31): odd levels zero-padded, single leaf = root */
long long p252o_merkle2_tree(const uint64_t tag[4], const uint64_t *leaves, size_t n_leaves,
                             uint64_t root[4], uint64_t *levels);

/* Merkle opening re-hash: leaves[n], siblings[n][depth][3], positions[n][depth] in 0..3 -> roots[n].
 * -1 on a position outside 0..3. */
int p252o_merkle4_path_batch(const uint64_t tag[4], const uint64_t *leaves, const uint64_t *siblings,
                             const uint8_t *positions, size_t depth, uint64_t *roots, size_t n);

/* ---- Domain / io-pattern (src/hash.rs:38-85) ---- */
/* domain: 0=Merkle4 1=Merkle2 2=Encryption 3=Other */
uint64_t p252o_domain_separator(int domain);
/* 0 ok, -1 IOPatternViolation (arity mismatch), -2 InvalidIOPattern (empty/zero-length call) */
int p252o_check_io(int domain, const size_t *absorb_lens, size_t n_absorbs, size_t out_len);

/* ---- UNPINNED: tag = hash_to_scalar(tag_input(io_pattern, domain_sep)) ----
 * tag_input (dusk-safe 0.3, recollection): contiguous same-kind calls aggregated; each word
 * big-endian u32 (absorb: 0x80000000|len, squeeze: len); then domain_sep as big-endian u64.
 * hash_to_scalar (dusk-bls12_381 0.14, recollection): BLAKE2b-512(bytes) read as a 512-bit
 * little-endian integer, reduced mod p. */
int p252o_tag(int domain, const size_t *absorb_lens, size_t n_absorbs, size_t out_len,
              uint64_t tag_out[4]);
void p252o_blake2b512(const uint8_t *msg, size_t len, uint8_t out[64]);

/* ---- UNPINNED: encryption (src/encryption.rs:62-95 -> dusk_safe::encrypt/decrypt, un-vendored).
 * The literal sponge-call sequence, two candidates behind `variant`:
 *   P252O_CRYPT_STREAM (default): [Absorb(2), Absorb(1), Squeeze(len), Absorb(len), Squeeze(1)]
 *   P252O_CRYPT_DUPLEX:           [Absorb(2), Absorb(1), {Squeeze(c), Absorb(c)}*, Squeeze(1)], c = min(4, remaining)
 * cipher[i] = message[i] + mask[i] followed by the last squeezed element (the MAC).  The secret is the two
 * coordinates of the JubJub shared point (encryption.rs:62-76) as BlsScalars.  The un-suffixed functions are
 * variant P252O_CRYPT_STREAM. ---- */
#define P252O_CRYPT_STREAM 0
#define P252O_CRYPT_DUPLEX 1
int p252o_encryption_tag_v(int variant, size_t message_len, uint64_t tag_out[4]);
int p252o_encrypt_v(int variant, const uint64_t tag[4], const uint64_t *message, size_t len, const uint64_t secret[8],
                    const uint64_t nonce[4], uint64_t *cipher /* len + 1 */);
/* 0 ok, -1 DecryptionFailed */
int p252o_decrypt_v(int variant, const uint64_t tag[4], const uint64_t *cipher /* len + 1 */, size_t len,
                    const uint64_t secret[8], const uint64_t nonce[4], uint64_t *message);
int p252o_encryption_tag(size_t message_len, uint64_t tag_out[4]);
int p252o_encrypt(const uint64_t tag[4], const uint64_t *message, size_t len, const uint64_t secret[8],
                  const uint64_t nonce[4], uint64_t *cipher /* len + 1 */);
int p252o_decrypt(const uint64_t tag[4], const uint64_t *cipher /* len + 1 */, size_t len, const uint64_t secret[8],
                  const uint64_t nonce[4], uint64_t *message);

/* ---- finalize_truncated post-processing (src/hash.rs:164-183): canonical value & (2^250-1),
 * returned as raw limbs (JubJubScalar::from_raw input).  Off the hot path; unpinned. ---- */
void p252o_truncate250(const uint64_t mont[4], uint64_t out_raw[4]);

/* deterministic input generator shared by tests and bench: splitmix64 stream,
 * top bit cleared, rejection-sampled < p, used directly as Montgomery-form memory. */
void p252o_fill_random(uint64_t seed, uint64_t *out, size_t n_scalars);

#ifdef __cplusplus
}
#endif
#endif
