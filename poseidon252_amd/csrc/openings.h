// openings.h — launch interface of openings.hip (extraction of Merkle openings from a stored tree: pure data movement, no hashing).
// Kept apart from kernels.h / kernels.hip: those are the HASHING kernels, whose source digest the committed counter passes and ISA
// counts under profiles/ are keyed to (bench.py KERNEL_SOURCES).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace p252 {

// k openings out of a tree stored as leaves[n_leaves] + levels (bottom-up concatenation, as the tree builder writes them):
// leaves_out[k], siblings[k][depth][3], positions[k][depth] — the arrays launch_merkle4_path takes.  index[i] >= n_leaves: the
// opening is all zeros and *n_bad (may be null) is incremented.
hipError_t launch_merkle4_openings(const void* leaves, size_t n_leaves, const void* levels, const void* index, size_t k, unsigned depth,
                                   void* leaves_out, void* siblings, void* positions, void* n_bad, hipStream_t st);

// the same for an arity-2 tree (Domain::Merkle2): siblings[k][depth][1], positions in 0..1
hipError_t launch_merkle2_openings(const void* leaves, size_t n_leaves, const void* levels, const void* index, size_t k, unsigned depth,
                                   void* leaves_out, void* siblings, void* positions, void* n_bad, hipStream_t st);

// merkle2.hip: re-hash of n arity-2 openings (depth sequential Merkle2 digests per lane): siblings[n][depth], positions[n][depth] in 0..1
struct TagArg;
hipError_t launch_merkle2_path(const int32_t* tab, const TagArg& tag, const void* leaves, const void* siblings, const void* positions, unsigned depth,
                               void* roots, size_t n, hipStream_t st);

// merkle2.hip: ok[i] = (roots[i] == *expected), one byte per opening (Opening::verify in bulk, behind either arity's re-hash)
hipError_t launch_compare_roots(const void* roots, const void* expected, void* ok, size_t n, hipStream_t st);

// merkle2.hip: sharded builds (comm.cpp) — a gathered root of all-ones is a failed peer's sentinel: poison root_out (32 B, all-ones) and
// store 1 + that rank in *fail_flag (device pointer of a host-mapped word)
hipError_t launch_poison_if_peer_failed(const void* roots, unsigned world, void* root_out, unsigned* fail_flag, hipStream_t st);

}  // namespace p252
