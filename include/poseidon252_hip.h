/*
 * poseidon252_hip.h — C ABI of libposeidon252_hip.so: batched Poseidon252 (Hades width-5 + SAFE
 * sponge over the BLS12-381 scalar field) on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for the reference's native hash path.  The reference
 * (dusk-poseidon 0.42) has no FFI of its own; the seam this library replaces is
 *     dusk_poseidon::Hash::{new, update, output_len, finalize, digest}   src/hash.rs:98-195
 *     dusk_poseidon::Domain + From<Domain> for u64                        src/hash.rs:21-56
 *     dusk_safe::Safe<BlsScalar, 5>::permute for ScalarPermutation        src/hades/permutation/scalar.rs:24-27
 * A per-call replacement of Safe::permute would be one permutation per kernel launch, so the
 * boundary is the BATCHED sibling of Hash with identical per-item semantics.  INTEGRATION.md shows
 * the Rust `extern "C"` block and the `HashBatch` wrapper a maintainer would add.
 *
 * Scalars: every `uint64_t*` scalar buffer is an array of BlsScalar in the reference's memory
 * layout — 4 little-endian u64 limbs of the Montgomery residue a*2^256 mod p, fully reduced — so a
 * Rust `&[BlsScalar]` is passed as a pointer with zero conversion.  Outputs are fully reduced too
 * (bit-exact equality with the reference is limb equality).
 *
 * Tag: the sponge capacity element state[0] = Safe::tag(...) = BlsScalar::hash_to_scalar(tag_input)
 * (scalar.rs:29-31) is an explicit INPUT of every hashing call.  A Rust caller passes the value
 * computed by the real dusk crates; p252_tag() is a host convenience whose byte-level recipe is
 * not covered by any reference test ("parity unpinned").
 *
 * Device-resident (`*_device`) scalar arrays must be 16-byte aligned (any hipMalloc'ed BlsScalar array, or a
 * whole-scalar offset into one, is); a misaligned pointer returns P252_ERR_INVALID_ARGUMENT.  Input and output
 * arrays of one call must not overlap unless a function says otherwise (inputs are only read, hash.rs:94,118-120).
 * Errors: functions return 0 on success or a negative P252_ERR_*; nothing unwinds across the
 * boundary.  Where the reference panics (Hash::finalize on an invalid io-pattern, hash.rs:124-137)
 * this library returns P252_ERR_IO_PATTERN_VIOLATION / P252_ERR_INVALID_IO_PATTERN and the host
 * wrapper raises.  There is NO CPU fallback: without a usable HIP device every compute entry point
 * fails with P252_ERR_NO_DEVICE / P252_ERR_HIP.
 *
 * Batch size: any n >= 0.  A batch of at most 8,192 items (16,384 Merkle digests) cannot fill the chip and is bound by
 * one wave's latency; such batches run kernels that spread each state over a group of lanes (0.12 ms per permutation
 * instead of 0.17).  Results are the same bytes whichever kernel runs.
 *
 * Threading and streams: a context is bound to one device and its functions are CALLED by one thread at a time; distinct
 * contexts are independent.  The `*_device` entry points only enqueue on the stream they are given, and calls on DIFFERENT
 * streams of one context may overlap on the device: every root-only tree / forest build (d_levels == NULL) ping-pongs its
 * levels in scratch owned by the context PER STREAM (up to four streams at once; a fifth takes over the least recently used
 * scratch behind an event — never concurrently), and a communicator's buffers are handed from one stream to the next behind
 * an event as well.  Memory: that scratch is grow-only and sized by the largest build of its stream — 5/16 of the leaves' bytes
 * (160 MiB for 2^24 leaves, 40 GiB for 2^32) PER stream in use; pass d_levels to keep a build's memory entirely the caller's.  The one context-wide piece of state is the encryption call table: p252_{encrypt,decrypt}_batch_device
 * with another (variant, len) than the previous call drains the device before replacing it.
 * HIP graphs: the `*_device` hashing and tree entry points do nothing but enqueue kernels (and one 32-byte copy) on `hip_stream`, so
 * they can be stream-captured into a hipGraph and replayed on new data in the same buffers; call once outside the capture first
 * (context-owned scratch and the encryption call table are allocated / uploaded on first use, which a capture cannot do).
 * A tree / forest / verify build that is CAPTURED should pass caller-owned scratch (d_levels; for verify: re-hash into a buffer of
 * the caller's with p252_merkle{4,2}_path_batch_device): with d_levels = NULL the graph holds the addresses of the context's
 * per-stream pair and the pair's hand-over event becomes a graph node, so a replay is ordered against nothing the context does
 * later — a fifth stream taking the pair over, p252_trim, a growing build — and would race with it.  The hashing
 * entry points (digests, sponges, permutations, truncated forms) use no context scratch and capture without conditions.
 * Multi-GPU = one context per GPU: either one process (or thread) per GPU driving its own context,
 * or the p252_*_multi entry points below, which take the array of contexts and shard inside the library; batches
 * shard with no inter-GPU dependence.
 */
#ifndef POSEIDON252_HIP_H
#define POSEIDON252_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this interface.  Bindings compare it with p252_abi_version() of the library they loaded: a function that
 * changes its argument list keeps its name only together with a new number here.
 *   4  p252_encryption_tag / p252_{encrypt,decrypt}_batch[_device] take a leading `int variant`; the default construction
 *      of every wrapper is P252_CRYPT_STREAM (version 0.2 of the library had no selector and computed P252_CRYPT_DUPLEX:
 *      ciphertexts of messages longer than 4 scalars made then decrypt only with variant = P252_CRYPT_DUPLEX)
 *   5  + p252_abi_version, p252_merkle4_update_checked_device, p252_clock_probe_device, p252_staging_lanes; out-of-range
 *      indices of p252_merkle4_update_device are skipped (were undefined behaviour)
 *   6  + the RCCL communicator (p252_comm_*), p252_merkle4_tree_sharded_device, p252_merkle4_tree_multi_device_resident,
 *      p252_merkle4_forest[_device], p252_merkle2_forest_device, p252_merkle{4,2}_openings_device, p252_merkle{4,2}_depth,
 *      p252_merkle2_path_batch_device, P252_ERR_COMM; p252_merkle4_tree_multi_device exchanges the subtree roots with one
 *      ncclAllGather whenever its contexts sit on distinct devices (versions 6 and 7 linked librccl; version 8 resolves it at run time)
 *   7  + p252_hash_batch_truncated[_device] (finalize_truncated fused into the digest kernels' output stage: one launch),
 *      p252_wipe, p252_scratch_residue, p252_merkle{4,2}_verify_batch_device; the host-buffer p252_{encrypt,decrypt}_batch and p252_destroy clear the library-owned
 *      copies of what they were handed; root-only tree / forest builds on DIFFERENT streams of one context no longer share
 *      scratch; p252_merkle4_tree_multi_device accepts any array of contexts again (as ABI 5 did) and makes no communicator for
 *      a single context
 *   8  RCCL is no longer a load-time dependency: the library resolves it on the first p252_comm_* / RCCL-path call (search
 *      order: $P252_RCCL_PATH, a copy the process already holds, the loader's path, $ROCM_PATH/lib), so a single-GPU hashing
 *      deployment needs no RCCL and a process gets ONE copy; + p252_comm_backend (which copy), p252_comm_check (a peer's failed
 *      local build in a sharded tree: the healthy ranks' root becomes all-ones and p252_comm_check / p252_sync return
 *      P252_ERR_COMM — they used to return a garbage root as P252_OK), p252_trim (gives the grow-only scratch back);
 *      p252_scratch_residue no longer counts the encryption call table (it holds nothing of the caller's) */
#define P252_ABI_VERSION 8

#define P252_OK 0
#define P252_ERR_IO_PATTERN_VIOLATION (-1) /* dusk_poseidon::Error::IOPatternViolation, src/error.rs:12-14 */
#define P252_ERR_INVALID_IO_PATTERN (-2)   /* dusk_poseidon::Error::InvalidIOPattern,   src/error.rs:16-17 */
#define P252_ERR_INVALID_ARGUMENT (-3)
#define P252_ERR_HIP (-4)
#define P252_ERR_NO_DEVICE (-5)
#define P252_ERR_COMM (-6) /* an RCCL call failed (communicator creation, broadcast of the constants, all-gather of the roots) */

/* Domain discriminants, in the declaration order of `enum Domain` (src/hash.rs:21-36) */
#define P252_DOMAIN_MERKLE4 0
#define P252_DOMAIN_MERKLE2 1
#define P252_DOMAIN_ENCRYPTION 2
#define P252_DOMAIN_OTHER 3

#define P252_HADES_WIDTH 5 /* dusk_poseidon::HADES_WIDTH, src/lib.rs:17 */

typedef struct p252_ctx p252_ctx;
typedef struct p252_comm p252_comm; /* one rank of an RCCL communicator, bound to one context (multi-GPU section below) */
#define P252_COMM_ID_BYTES 128      /* size of the opaque id p252_comm_unique_id produces (= RCCL's ncclUniqueId) */

/* ---- lifecycle ---- */
int p252_device_count(void);
/* Binds to HIP device `device_id`, derives the constant tables from the embedded arc.bin / mds.bin
 * (round_constants.rs:26-54, mds_matrix.rs:17-39) and uploads them. */
int p252_create(int device_id, p252_ctx** out);
void p252_destroy(p252_ctx* ctx);
/* message for the last error on this context (ctx == NULL: last p252_create failure) */
const char* p252_last_error(const p252_ctx* ctx);

/* ---- batched compute, HOST buffers (synchronous; H2D + kernel + D2H) ---- */
/* n independent Hades permutations: replaces Safe::permute / Hades::perm
 * (scalar.rs:25-27, permutation.rs:105-123).  states/out: n x 5 scalars. */
int p252_permute_batch(p252_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n);
/* n independent sponge hashes with the same io-pattern: replaces Hash::finalize (hash.rs:128-155)
 * for n messages of in_len scalars each (contiguous, message-major), out_len outputs each in
 * squeeze order.  Merkle4 digest = (in_len 4, out_len 1).  in_len == 0 or out_len == 0 ->
 * P252_ERR_INVALID_IO_PATTERN. */
int p252_hash_batch(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* in, size_t in_len,
                    size_t out_len, uint64_t* out, size_t n);
/* the same with Hash::finalize_truncated's outputs (hash.rs:164-183): out_raw = n x out_len x 4 raw limbs, canonical value
 * & (2^250 - 1), truncated inside the digest kernel (see p252_hash_batch_truncated_device) */
int p252_hash_batch_truncated(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* in, size_t in_len,
                              size_t out_len, uint64_t* out_raw, size_t n);
/* Arity-4 Merkle tree over Hash::digest(Domain::Merkle4, [c0,c1,c2,c3]) nodes; empty child slots
 * are the zero scalar (hash.rs:22-26).  Levels are built while more than one node remains (a single
 * leaf is its own root; a 4^k-leaf tree costs exactly k levels).  `levels`
 * (optional) receives every level above the leaves, bottom-up, p252_merkle4_levels_len(n_leaves)
 * scalars.  n_leaves == 0 -> P252_ERR_INVALID_ARGUMENT. */
int p252_merkle4_tree(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* leaves, size_t n_leaves,
                      uint64_t root[4], uint64_t* levels);
size_t p252_merkle4_levels_len(size_t n_leaves);
/* The same for arity 2: nodes are Hash::digest(Domain::Merkle2, [c0, c1]) (hash.rs:27-31), an odd level is
 * padded with the zero scalar; `tag` must be the Merkle2 tag ([Absorb(2), Squeeze(1)], separator 0x3). */
int p252_merkle2_tree(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* leaves, size_t n_leaves,
                      uint64_t root[4], uint64_t* levels);
size_t p252_merkle2_levels_len(size_t n_leaves);

/* Page-locked host memory (hipHostMalloc).  p252_hash_batch pipelines H2D / kernel / D2H for large batches.  With
 * ordinary (pageable) caller memory — a Rust Vec<BlsScalar> — nothing the caller owns is touched: chunks go through
 * library-owned page-locked staging lanes (a worker thread and stream per lane, two chunks in flight per lane;
 * P252_HOST_LANES, default 3; P252_HOST_CHUNK_MB, default 8) so that the host copy of one chunk overlaps the DMA and kernel
 * of the others: 3.5-3.8e8 Merkle4 digests/s host-to-host.  With buffers from p252_host_alloc / p252_host_register on BOTH
 * sides the copies are zero-copy DMA over 3 streams: 4.0e8.  NULL on failure. */
void* p252_host_alloc(size_t bytes);
void p252_host_free(void* p);
/* Page-lock / release a buffer the caller already owns (hipHostRegister / hipHostUnregister) — e.g. a Rust
 * Vec<BlsScalar> that is hashed repeatedly: registered once, every later host-buffer call on it runs at the
 * p252_host_alloc rate instead of paying the per-call page-locking.  The buffer must stay allocated until it is
 * unregistered.  0 or P252_ERR_HIP / P252_ERR_INVALID_ARGUMENT. */
int p252_host_register(void* p, size_t bytes);
int p252_host_unregister(void* p);

/* ---- batched compute, DEVICE buffers (asynchronous on `hip_stream`, a hipStream_t; NULL = the
 * default stream).  Pointers are device addresses with the same layouts as above.  This is the
 * path bench.py times: inputs already resident in HBM. ---- */
int p252_permute_batch_device(p252_ctx* ctx, const void* d_states, void* d_out, size_t n, void* hip_stream);
int p252_hash_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_in, size_t in_len,
                           size_t out_len, void* d_out, size_t n, void* hip_stream);
/* Hash::finalize_truncated / digest_truncated (hash.rs:164-183, 203-210) for the whole batch in ONE launch: the same kernels
 * with a truncating output stage — every squeezed scalar is canonicalised (Montgomery form dropped), masked to its low 250 bits
 * and stored as the raw limbs JubJubScalar::from_raw receives; identical to p252_hash_batch_device followed by
 * p252_truncate250_device without the second launch and the 64 B / scalar round trip through HBM. */
int p252_hash_batch_truncated_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_in, size_t in_len,
                                     size_t out_len, void* d_out_raw, size_t n, void* hip_stream);
/* d_levels may be NULL: the levels then ping-pong in context-owned scratch — one pair per caller stream, so builds queued on
 * different streams of one context never share it (see "Threading and streams"); d_root receives 1 scalar. */
int p252_merkle4_tree_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, size_t n_leaves,
                             void* d_root, void* d_levels, void* hip_stream);
int p252_merkle2_tree_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, size_t n_leaves,
                             void* d_root, void* d_levels, void* hip_stream);
/* waits for hip_stream.  P252_ERR_COMM when a sharded tree build of this context's communicator found a failed peer since the
 * last call (see p252_comm_check). */
int p252_sync(p252_ctx* ctx, void* hip_stream);
/* Secret hygiene (the reference builds with `zeroize`, Cargo.toml:14; dusk-safe zeroizes a finished sponge).  The kernels keep
 * sponge states in registers only.  What a HOST-buffer call leaves behind is the library's copy of the caller's arrays — the
 * context's device scratch and the page-locked staging lanes: p252_encrypt_batch / p252_decrypt_batch clear what they used
 * before returning, p252_destroy clears everything before freeing it, and p252_wipe clears every buffer the context owns on
 * demand (it waits for the device first; hashing calls do not wipe — their inputs are public).  `_device` callers own their
 * buffers; the only thing such a call leaves in the context is the (variant, len) call table, which holds no secret. */
int p252_wipe(p252_ctx* ctx);
/* diagnostics: the number of non-zero bytes in every buffer of the context that receives caller data (device scratch, level
 * scratch, staging lanes on both sides; not the encryption call table) — 0 right after p252_wipe, p252_trim or a host-buffer
 * encrypt / decrypt call on a context that has done nothing else since its last wipe */
int p252_scratch_residue(p252_ctx* ctx, uint64_t* nonzero_bytes);
/* The context's scratch is grow-only (device scratch of the host-buffer entry points, up to four per-stream level pairs of
 * 5/16 of a root-only build's leaves each, staging lanes, the call table): p252_trim waits for the device, clears it as p252_wipe
 * does and FREES it — the reference holds no state at all (hash.rs:92-96).  The next call allocates what it needs again.  The
 * constant table and a communicator stay.  Not to be called while a hipGraph captured from this context's root-only builds is
 * still to be replayed: captured builds that pass d_levels = NULL bake the scratch addresses in (pass caller-owned d_levels to
 * builds that are captured, which is also what keeps a replay from racing with another stream's build on the same pair). */
int p252_trim(p252_ctx* ctx);

/* ---- SURVEY §8(f) "next" rows ---- */
/* finalize_truncated's post-processing (hash.rs:164-183) on n device-resident BlsScalars: canonical
 * value & (2^250 - 1), written as the raw limbs JubJubScalar::from_raw receives.  d_out_raw may alias
 * d_scalars. */
int p252_truncate250_device(p252_ctx* ctx, const void* d_scalars, void* d_out_raw, size_t n, void* hip_stream);
/* Incremental update of a stored arity-4 tree (the poseidon-merkle consumer keeps its tree and changes leaves): writes
 * d_new_leaves[i] to d_leaves[d_indices[i]] (k distinct uint32 positions < n_leaves) and re-hashes every ancestor, level by
 * level, in place in d_levels (the layout p252_merkle4_tree_device fills: all levels above the leaves, bottom-up,
 * p252_merkle4_levels_len(n_leaves) scalars).  Cost: log4(n_leaves) launches of k digests; afterwards leaves and levels equal
 * those of a fresh build.  d_root (optional) receives the new root.  An index >= n_leaves is skipped (nothing is written for
 * it).  The positions must be distinct: the same position twice with different values leaves one of them — or, the two
 * 16-byte halves of a scalar being stored separately, a mix — in the leaf, and the levels above consistent with whatever
 * was stored.  The _checked variant additionally counts the skipped indices: *d_n_bad (a device uint32 the caller has
 * zeroed) is incremented once per out-of-range index. */
int p252_merkle4_update_device(p252_ctx* ctx, const uint64_t tag[4], void* d_leaves, size_t n_leaves, void* d_levels,
                               const void* d_indices, const void* d_new_leaves, size_t k, void* d_root, void* hip_stream);
int p252_merkle4_update_checked_device(p252_ctx* ctx, const uint64_t tag[4], void* d_leaves, size_t n_leaves, void* d_levels,
                                       const void* d_indices, const void* d_new_leaves, size_t k, void* d_root, void* d_n_bad,
                                       void* hip_stream);
/* The canonical byte format on either side of the path: BlsScalar::to_bytes / from_bytes (dusk-bls12_381; the reference
 * round-trips its round constants through the pair, src/hades/round_constants.rs:66-67, and reads its KAT inputs with
 * from_hex_str, src/hades.rs:131).  bytes = n records of 32 little-endian bytes of the canonical value (16-byte aligned
 * on the device).  from_bytes: ok[i] = 1 iff the value is < p — BlsScalar::from_bytes fails otherwise; the limbs written
 * are then those of the value mod p; ok may be NULL.  Buffers may alias (in place). */
int p252_to_bytes_device(p252_ctx* ctx, const void* d_scalars, void* d_bytes, size_t n, void* hip_stream);
int p252_from_bytes_device(p252_ctx* ctx, const void* d_bytes, void* d_scalars, void* d_ok, size_t n, void* hip_stream);
/* Batched Merkle openings (arity 4): recompute the root from a leaf and its sibling path — the branch
 * re-hash a `poseidon-merkle` verifier performs (AGENTS.md:62-66 names the downstream crate).  Per
 * level l, node = Hash::digest(Domain::Merkle4, children) where children[positions[l]] is the value
 * coming from below and the 3 siblings fill the other slots in order.  Layouts: leaves[n],
 * siblings[n][depth][3] scalars, positions[n][depth] bytes in 0..3, roots[n].  depth == 0 copies the
 * leaves.  positions outside 0..3 -> P252_ERR_INVALID_ARGUMENT (host variant; the device variant masks). */
int p252_merkle4_path_batch(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* leaves, const uint64_t* siblings,
                            const uint8_t* positions, size_t depth, uint64_t* roots, size_t n);
int p252_merkle4_path_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, const void* d_siblings,
                                   const void* d_positions, size_t depth, void* d_roots, size_t n, void* hip_stream);

/* Batched encryption / decryption: replaces dusk_poseidon::encrypt / decrypt (src/encryption.rs:62-95),
 * thin wrappers over dusk_safe::encrypt / decrypt with ScalarPermutation and Domain::Encryption.  Per
 * item: message of `len` scalars, shared secret = the 2 coordinates (u, v) of the JubJub shared point as
 * BlsScalars (encryption.rs:66-69), nonce = 1 scalar; cipher = len + 1 scalars (masked message + MAC).
 * Layouts: messages[n][len], secrets[n][2], nonces[n], ciphers[n][len+1], ok[n] bytes (1 = MAC verified;
 * 0 = dusk_poseidon::Error::DecryptionFailed, src/error.rs:27-29, and that item's message output is
 * unspecified).
 *
 * `variant` selects the sponge-call sequence.  dusk-safe 0.3 is not vendored in the reference and the reference's
 * tests (tests/encryption.rs:30-115, message lengths 42 and 21) are round trips and failures only, so NO reference
 * value pins the construction; both candidates run on the KAT-pinned sponge state machine, and the library, its
 * p252_encryption_tag() and the test oracle all derive from one call table per variant:
 *   P252_CRYPT_STREAM (0, the default of every wrapper — dusk-safe's encrypt as published, to the best of three
 *       independent recollections): io-pattern [Absorb(2), Absorb(1), Squeeze(len), Absorb(len), Squeeze(1)]:
 *       absorb secret, absorb nonce, squeeze ALL len masks, absorb the whole message, squeeze the MAC;
 *       cipher[i] = message[i] + mask[i], cipher[len] = MAC.
 *   P252_CRYPT_DUPLEX (1, what version 0.2 of this library shipped): [Absorb(2), Absorb(1), {Squeeze(c), Absorb(c)}*,
 *       Squeeze(1)] with c = min(4, remaining): squeeze / absorb chunk by chunk.
 * The two are identical for len <= 4 and differ beyond.  bindings/rust/tests/parity.rs decides between them on a
 * machine with the dusk crates.  `tag` as everywhere: pass the real crate's value, or p252_encryption_tag() (UNPINNED). */
#define P252_CRYPT_STREAM 0
#define P252_CRYPT_DUPLEX 1
int p252_encryption_tag(int variant, size_t message_len, uint64_t tag_out[4]);
int p252_encrypt_batch(p252_ctx* ctx, int variant, const uint64_t tag[4], const uint64_t* messages, const uint64_t* secrets,
                       const uint64_t* nonces, size_t len, uint64_t* ciphers, size_t n);
int p252_decrypt_batch(p252_ctx* ctx, int variant, const uint64_t tag[4], const uint64_t* ciphers, const uint64_t* secrets,
                       const uint64_t* nonces, size_t len, uint64_t* messages, uint8_t* ok, size_t n);
int p252_encrypt_batch_device(p252_ctx* ctx, int variant, const uint64_t tag[4], const void* d_messages, const void* d_secrets,
                              const void* d_nonces, size_t len, void* d_ciphers, size_t n, void* hip_stream);
int p252_decrypt_batch_device(p252_ctx* ctx, int variant, const uint64_t tag[4], const void* d_ciphers, const void* d_secrets,
                              const void* d_nonces, size_t len, void* d_messages, void* d_ok, size_t n, void* hip_stream);

/* Openings of a STORED tree, extracted on the device (no hashing; HBM-bound data movement): d_leaves[n_leaves] and d_levels as
 * p252_merkle4_tree_device filled them (all levels, bottom-up); d_indices = k leaf positions (uint32, device).  Writes, in exactly
 * the layout p252_merkle4_path_batch_device reads: d_leaves_out[k] (the leaves at those positions), d_siblings[k][depth][3],
 * d_positions[k][depth] with depth = p252_merkle4_depth(n_leaves); missing siblings of a ragged level are the zero scalar
 * (hash.rs:22-26).  A position >= n_leaves yields an all-zero opening and is counted in *d_n_bad (uint32, device; may be NULL).
 * Re-hashing the result with p252_merkle4_path_batch_device gives the tree's root for every valid position.  Asynchronous. */
size_t p252_merkle4_depth(size_t n_leaves);
int p252_merkle4_openings_device(p252_ctx* ctx, const void* d_leaves, size_t n_leaves, const void* d_levels, const void* d_indices, size_t k,
                                 void* d_leaves_out, void* d_siblings, void* d_positions, void* d_n_bad, void* hip_stream);
/* The arity-2 twins (Domain::Merkle2 trees as p252_merkle2_tree_device builds them; pass the Merkle2 tag): ONE sibling per level —
 * d_siblings[k][depth], d_positions[k][depth] in 0..1 (1 = the path's node is the right child), depth = p252_merkle2_depth(n_leaves);
 * p252_merkle2_path_batch_device re-hashes n such openings (depth sequential Hash::digest(Domain::Merkle2, [left, right]) per lane). */
size_t p252_merkle2_depth(size_t n_leaves);
int p252_merkle2_openings_device(p252_ctx* ctx, const void* d_leaves, size_t n_leaves, const void* d_levels, const void* d_indices, size_t k,
                                 void* d_leaves_out, void* d_siblings, void* d_positions, void* d_n_bad, void* hip_stream);
int p252_merkle2_path_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, const void* d_siblings,
                                   const void* d_positions, size_t depth, void* d_roots, size_t n, void* hip_stream);

/* Opening::verify in bulk (the downstream poseidon-merkle verifier, AGENTS.md:62-66): the n openings are re-hashed exactly as
 * p252_merkle{4,2}_path_batch_device does and each result is compared with the ONE expected root at d_root (32 bytes, device):
 * d_ok[i] = 1 iff opening i leads to that root — n bytes leave the device instead of n x 32.  The recomputed roots live in the
 * context's scratch of `hip_stream`.  Asynchronous. */
int p252_merkle4_verify_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, const void* d_siblings,
                                     const void* d_positions, size_t depth, const void* d_root, void* d_ok, size_t n, void* hip_stream);
int p252_merkle2_verify_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, const void* d_siblings,
                                     const void* d_positions, size_t depth, const void* d_root, void* d_ok, size_t n, void* hip_stream);

/* A FOREST of n_trees independent complete arity-4 trees of leaves_per_tree = 4^k leaves each (the downstream poseidon-merkle
 * shape, AGENTS.md:62-66: many small trees), tree-major in d_leaves.  Level l of all trees is one array of
 * n_trees * 4^(k-l) nodes, tree-major, and each level is ONE launch across all trees — the narrow upper levels of many small
 * trees fill the chip together instead of each tree paying one wave's latency per level (k launches instead of
 * n_trees * k).  d_roots[n_trees] (device) receives tree t's root at index t: identical to p252_merkle4_tree_device of
 * that tree alone.  d_levels (device, may be NULL): all levels bottom-up, level-major —
 * n_trees * p252_merkle4_levels_len(leaves_per_tree) scalars, level l at offset n_trees * (4^(k-1) + ... + 4^(k-l+1)),
 * tree t's nodes of that level at t * 4^(k-l) within it.  Asynchronous on hip_stream. */
int p252_merkle4_forest_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, size_t n_trees, size_t leaves_per_tree,
                               void* d_roots, void* d_levels, void* hip_stream);
/* the same from HOST leaves (pageable memory is fine), roots to host memory: the first level — three quarters of the work — is hashed
 * chunk by chunk while the leaves stream in through the context's staging lanes (the leaves are never resident as a whole), the
 * upper levels then run once, one launch per level across all trees; synchronous */
int p252_merkle4_forest(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* leaves, size_t n_trees, size_t leaves_per_tree,
                        uint64_t* roots);
/* the same for arity 2 (Domain::Merkle2 nodes; pass the Merkle2 tag): leaves_per_tree = 2^k, p252_merkle2_levels_len per tree */
int p252_merkle2_forest_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, size_t n_trees, size_t leaves_per_tree,
                               void* d_roots, void* d_levels, void* hip_stream);

/* ---- multi-device: an array of contexts, one per GPU.  Shards are contiguous and independent: no
 * inter-GPU dependence and no collective on the data path.  The calls are synchronous; inside, one host thread drives
 * each context.  A context may appear only once, and when the node has at least n_ctx devices no two contexts may be bound
 * to the same one (P252_MULTI_ALLOW_SHARED_DEVICE=1 lifts that).  The contexts share the process's CPU budget: each uses
 * p252_staging_lanes(n_ctx) staging lanes for pageable buffers.  On failure the error text is on ctxs[0] (p252_last_error). ---- */
/* n digests / sponges, item i of the batch on device floor-split: device t hashes a contiguous n/n_ctx slice (sizes
 * differ by at most one); per-item results identical to p252_hash_batch. */
int p252_hash_batch_multi(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const uint64_t* in, size_t in_len,
                          size_t out_len, uint64_t* out, size_t n);
/* the same on device-resident shards: d_in[t] / d_out[t] live on ctxs[t]'s device and hold n_per_ctx[t] items;
 * asynchronous on hip_streams[t] (hip_streams == NULL: every device's default stream); p252_sync each context. */
int p252_hash_batch_multi_device(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const void* const* d_in, size_t in_len,
                                 size_t out_len, void* const* d_out, const size_t* n_per_ctx, void* const* hip_streams);
/* Sharded arity-4 tree (BASELINE configs[4]: 2^27 leaves = 8 x 4^12): device t reduces the t-th contiguous run of
 * n_leaves / n_ctx leaves — which must be a complete subtree, 4^k leaves — to its root with zero communication; the
 * n_ctx roots (32 bytes each) are gathered on the host and the <= log4(n_ctx) + 1 top levels (zero-padded per
 * hash.rs:22-26) are hashed on ctxs[0].  root == p252_merkle4_tree over the concatenation.  Anything else than
 * n_leaves = n_ctx * 4^k -> P252_ERR_INVALID_ARGUMENT. */
int p252_merkle4_tree_multi(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const uint64_t* leaves, size_t n_leaves,
                            uint64_t root[4]);
/* the same with every device's leaves_per_ctx = 4^k leaves already resident at d_leaves[t]; root is written to host memory.
 * Contexts on distinct devices: the roots are exchanged by ONE ncclAllGather on the devices' streams (the RCCL communicator
 * over the contexts is created on the first such call — or beforehand with p252_comm_create_all — and kept; it is destroyed
 * with its contexts) and every device hashes the top levels, so nothing but the final 32 bytes crosses PCIe.  Contexts
 * that share a device (RCCL wants one device per rank: the single-GPU test configuration), or P252_MULTI_HOST_GATHER=1:
 * the roots are gathered through the host and the top levels run on ctxs[0].
 * ANY array of contexts is accepted, call after call (ctxs[0..8), then ctxs[0..4), a permutation ...): a communicator this
 * entry point made itself over another array is torn down and re-made for the array at hand (communicator creation costs
 * ~0.1-1 s: keep the array stable, or make the communicator once with p252_comm_create_all); contexts that sit in a
 * communicator the CALLER made over another array, and a failing lazy creation, fall back to the host gather.  n_ctx == 1
 * makes no communicator at all. */
int p252_merkle4_tree_multi_device(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const void* const* d_leaves,
                                   size_t leaves_per_ctx, uint64_t root[4]);
/* the RCCL path only, fully asynchronous and device-resident: on return the work is queued on hip_streams[t] (NULL: the
 * default streams) and d_root_out[t] (32 bytes on ctxs[t]'s device; entries, or the array, may be NULL) will hold the
 * root on EVERY device — no host round trip.  P252_ERR_COMM when the contexts cannot form a communicator (shared device,
 * P252_MULTI_HOST_GATHER=1, contexts of a caller-made communicator over another array, RCCL refusing): there is no host path here. */
int p252_merkle4_tree_multi_device_resident(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const void* const* d_leaves,
                                            size_t leaves_per_ctx, void* const* d_root_out, void* const* hip_streams);

/* ---- RCCL communicator (BASELINE north_star: "RCCL broadcast of constants", "RCCL gather of roots").  The path
 * has no data-path collective; a communicator exists for (i) the constant table: on creation rank 0's table is broadcast
 * (ncclBroadcast) and every rank REFUSES it unless it equals the table it derives from its own arc.bin / mds.bin — the rule
 * of p252_tables_import — and (ii) the 32-byte subtree roots of a sharded tree build (ncclAllGather on the rank's stream).
 * One process per GPU: rank 0 calls p252_comm_unique_id and passes the bytes to the other ranks by any host means; every
 * rank calls p252_comm_create_rank with its own context (collective: returns when all `world` ranks have called).
 * One process, several GPUs: p252_comm_create_all over contexts on DISTINCT devices.  A context belongs to at most one
 * communicator; destroy the communicator before its context.  xGMI note: these messages are world x 32 bytes — latency
 * only.  Creation and the sharded build are COLLECTIVE: every rank calls them, in the same order.  Every device allocation
 * of a rank happens before its first collective, so a rank that cannot allocate fails before its peers wait for it; a rank
 * whose subtree build fails later still enters the all-gather (contributing an all-ones root, which is no BlsScalar) and
 * returns its error — peers are never left blocked on the stream.  The healthy ranks DETECT the sentinel on the device: their
 * d_root is overwritten with all-ones as well, and the next p252_comm_check / p252_sync on that rank returns P252_ERR_COMM
 * naming the failed rank (the build itself is asynchronous and has long returned P252_OK by then).  Builds of one communicator
 * queued on different streams run one after the other (event-ordered on the device), never concurrently.
 * RCCL is resolved at run time (ABI 8): without it these entry points return P252_ERR_COMM and p252_last_error lists what was
 * tried; nothing else in this header needs it. ---- */
int p252_comm_unique_id(void* id_out, size_t len /* = P252_COMM_ID_BYTES */);
int p252_comm_create_rank(p252_ctx* ctx, const void* id, size_t len, int rank, int world, p252_comm** out);
int p252_comm_create_all(p252_ctx* const* ctxs, size_t n_ctx, p252_comm** comms_out /* [n_ctx] */);
void p252_comm_destroy(p252_comm* comm);
int p252_comm_rank(const p252_comm* comm);
int p252_comm_size(const p252_comm* comm);
/* waits for hip_stream, then reports whether a sharded build of this communicator met a failed peer since the last check:
 * P252_ERR_COMM (p252_last_error names the rank) once, P252_OK otherwise */
int p252_comm_check(p252_comm* comm, void* hip_stream);
/* which RCCL this process's library calls go to: resolves it if no call has yet, writes the path of the object the symbols
 * came from (NUL-terminated, truncated to len; path_out may be NULL).  P252_ERR_COMM: none found (p252_last_error(NULL)) */
int p252_comm_backend(char* path_out, size_t len);
/* BASELINE configs[4], one rank's part (one process per GPU): this rank's n_leaves_local = 4^k device-resident leaves are
 * reduced to their root with zero communication, the `world` roots are all-gathered (the path's only exchange step, on
 * hip_stream) and the <= log4(world) + 1 top levels (zero-padded per hash.rs:22-26) are hashed on every rank: d_root
 * (32 bytes, device) = the root of the tree over the concatenation of all ranks' leaves, in rank order, on every rank.
 * Asynchronous on hip_stream; collective: every rank of the communicator must call it. */
int p252_merkle4_tree_sharded_device(p252_comm* comm, const uint64_t tag[4], const void* d_leaves, size_t n_leaves_local, void* d_root,
                                     void* hip_stream);

/* ---- constant-table exchange (multi-GPU: rank 0 broadcasts its derived table over RCCL, every
 * rank imports it; byte-identical to what p252_create derives locally) ---- */
size_t p252_tables_size(void);
int p252_tables_export(p252_ctx* ctx, void* host_buf, size_t len);
int p252_tables_import(p252_ctx* ctx, const void* host_buf, size_t len);

/* ---- host helpers mirroring src/hash.rs ---- */
/* From<Domain> for u64 (hash.rs:38-56) */
int p252_domain_separator(int domain, uint64_t* sep_out);
/* io_pattern() validation (hash.rs:62-85) + dusk-safe's pattern rules: 0, or
 * P252_ERR_IO_PATTERN_VIOLATION (Merkle arity mismatch), P252_ERR_INVALID_IO_PATTERN (no absorb,
 * zero-length call, out_len 0). */
int p252_check_io_pattern(int domain, const size_t* absorb_lens, size_t n_absorbs, size_t out_len);
/* UNPINNED convenience: tag = hash_to_scalar(tag_input(io_pattern, domain_sep)). */
int p252_tag(int domain, const size_t* absorb_lens, size_t n_absorbs, size_t out_len, uint64_t tag_out[4]);
/* finalize_truncated post-processing (hash.rs:164-183): canonical value & (2^250 - 1), as the raw
 * limbs handed to JubJubScalar::from_raw.  Host-side, n scalars. */
int p252_truncate250(const uint64_t* scalars, uint64_t* out_raw, size_t n);

/* host-side twins of p252_to_bytes_device / p252_from_bytes_device (no context, no GPU: one multiplication per scalar) */
int p252_to_bytes(const uint64_t* scalars, uint8_t* bytes, size_t n);
int p252_from_bytes(const uint8_t* bytes, uint64_t* scalars, uint8_t* ok, size_t n);

/* ---- measurement aids (bench.py; not on the hashing path) ---- */
/* Launches ONE wave on `hip_stream` that samples the shader-clock counter (s_memtime) and the constant 100 MHz real-time
 * counter (s_memrealtime) before and after sleeping for about spin_us microseconds, and writes
 * d_out6[6] (uint64, device) = {memtime0, realtime0, memtime after a chain of 1,024 dependent v_add_u32, memtime1, realtime1,
 * chain result}: (memtime1 - memtime0) / (realtime1 - realtime0) x 100 MHz = the shader clock over the interval.  On a stream
 * of its own while hashing kernels run it measures the clock under that load. */
int p252_clock_probe_device(p252_ctx* ctx, void* d_out6, unsigned spin_us, void* hip_stream);
/* staging lanes (worker threads) each context uses for pageable host buffers when n_ctx contexts are driven at once by a
 * p252_*_multi call: clamp(floor(usable CPUs / n_ctx), 1, 3) — the driver thread of a context is its first lane, so the
 * call never runs more workers than the process has CPUs (min(affinity, cgroup quota)); n_ctx <= 1: the single-context
 * default (3; 2 below four CPUs).  P252_HOST_LANES overrides both. */
int p252_staging_lanes(size_t n_ctx);

/* library/version introspection */
int p252_abi_version(void); /* P252_ABI_VERSION the library was built from */
const char* p252_version(void);

#ifdef __cplusplus
}
#endif
#endif
