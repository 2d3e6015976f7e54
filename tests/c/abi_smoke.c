/* abi_smoke.c — include/poseidon252_hip.h consumed from plain C (gcc -std=c11 -pedantic -Werror): proves the header is
 * valid C, links against libposeidon252_hip.so and drives the hot path through the C ABI exactly as a cgo / Rust-FFI /
 * C caller would.  Without a HIP device it checks the host helpers and that compute fails loudly (exit 0, "no device");
 * with one it hashes, permutes, builds trees (single and multi-context) and cross-checks them against each other. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "poseidon252_hip.h"

/* the four HIP runtime calls a C caller with device-resident data needs (libamdhip64; declared here so that this file stays
 * plain C11 without the HIP headers): all return 0 on success; kind 1 = host to device, 2 = device to host */
extern int hipMalloc(void** ptr, size_t bytes);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t bytes, int kind);
extern int hipDeviceSynchronize(void);

#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

static uint64_t sm64(uint64_t* s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

int main(void) {
    uint64_t sep = 0, tag[4], tag2[4];
    size_t four = 4, lens2[2] = {1, 3};
    printf("%s\n", p252_version());
    CHECK(p252_abi_version() == P252_ABI_VERSION); /* the library this binary loads implements the header it was compiled against */
    CHECK(p252_domain_separator(P252_DOMAIN_MERKLE4, &sep) == P252_OK && sep == 0xf);
    CHECK(p252_domain_separator(P252_DOMAIN_ENCRYPTION, &sep) == P252_OK && sep == 0x100000000ULL);
    CHECK(p252_check_io_pattern(P252_DOMAIN_MERKLE4, &four, 1, 1) == P252_OK);
    CHECK(p252_check_io_pattern(P252_DOMAIN_MERKLE4, &four, 1, 2) == P252_ERR_IO_PATTERN_VIOLATION);
    CHECK(p252_check_io_pattern(P252_DOMAIN_OTHER, NULL, 0, 1) == P252_ERR_INVALID_IO_PATTERN);
    CHECK(p252_tag(P252_DOMAIN_MERKLE4, &four, 1, 1, tag) == P252_OK);
    CHECK(p252_tag(P252_DOMAIN_MERKLE4, lens2, 2, 1, tag2) == P252_OK && memcmp(tag, tag2, 32) == 0); /* chunks aggregate */
    CHECK(p252_merkle4_levels_len(16) == 5 && p252_merkle2_levels_len(8) == 7 && p252_tables_size() > 0);
    CHECK(p252_encryption_tag(P252_CRYPT_STREAM, 42, tag2) == P252_OK && p252_encryption_tag(7, 42, tag2) == P252_ERR_INVALID_ARGUMENT);
    {   /* the canonical byte format (host side): 1 <-> bytes 01 00 .. 00; all-ones is not below the modulus */
        uint8_t one_bytes[32] = {1}, ff[32], okf[2] = {9, 9}, back[32];
        uint64_t s2[8];
        memset(ff, 0xff, 32);
        uint8_t two[64];
        memcpy(two, one_bytes, 32);
        memcpy(two + 32, ff, 32);
        CHECK(p252_from_bytes(two, s2, okf, 2) == P252_OK && okf[0] == 1 && okf[1] == 0);
        CHECK(s2[0] == 0x00000001fffffffeULL && s2[3] == 0x1824b159acc5056fULL); /* R = the Montgomery form of 1 */
        CHECK(p252_to_bytes(s2, back, 1) == P252_OK && memcmp(back, one_bytes, 32) == 0);
    }

    p252_ctx* ctx = NULL;
    int rc = p252_create(0, &ctx);
    if (rc == P252_ERR_NO_DEVICE) {
        CHECK(ctx == NULL && strlen(p252_last_error(NULL)) > 0);
        printf("no device: host helpers OK, compute refuses to run (no CPU fallback)\n");
        return 0;
    }
    CHECK(rc == P252_OK && ctx != NULL);

    enum { N = 4096 };
    uint64_t seed = 0xc10d;
    uint64_t* in = (uint64_t*)malloc(sizeof(uint64_t) * 4 * 4 * N);
    uint64_t* out = (uint64_t*)malloc(sizeof(uint64_t) * 4 * N);
    uint64_t* st = (uint64_t*)malloc(sizeof(uint64_t) * 4 * 5 * N);
    uint64_t* st_out = (uint64_t*)malloc(sizeof(uint64_t) * 4 * 5 * N);
    CHECK(in && out && st && st_out);
    for (size_t i = 0; i < 16 * (size_t)N; ++i) in[i] = sm64(&seed) >> 2; /* limbs < 2^62: every scalar < p */
    /* Hash::digest(Merkle4, x)[0] == perm([tag, x0..x3])[1]  (hash.rs:128-155 with [Absorb(4), Squeeze(1)]) */
    CHECK(p252_hash_batch(ctx, tag, in, 4, 1, out, N) == P252_OK);
    for (size_t i = 0; i < N; ++i) {
        memcpy(st + 20 * i, tag, 32);
        memcpy(st + 20 * i + 4, in + 16 * i, 128);
    }
    CHECK(p252_permute_batch(ctx, st, st_out, N) == P252_OK);
    for (size_t i = 0; i < N; ++i) CHECK(memcmp(out + 4 * i, st_out + 20 * i + 4, 32) == 0);
    /* tree over the 4N inputs as leaves: level 1 is exactly the digests above */
    uint64_t root[4], root2[4];
    uint64_t* levels = (uint64_t*)malloc(32 * p252_merkle4_levels_len(4 * N));
    CHECK(levels && p252_merkle4_tree(ctx, tag, in, 4 * N, root, levels) == P252_OK);
    CHECK(memcmp(levels, out, 32 * N) == 0);
    /* multi-context: 4 contexts (here all on device 0), one complete subtree each */
    p252_ctx* ctxs[4] = {NULL, NULL, NULL, NULL};
    for (int t = 0; t < 4; ++t) CHECK(p252_create(0, &ctxs[t]) == P252_OK);
    CHECK(p252_merkle4_tree_multi(ctxs, 4, tag, in, 4 * N, root2) == P252_OK && memcmp(root, root2, 32) == 0);
    uint64_t* out2 = (uint64_t*)malloc(sizeof(uint64_t) * 4 * N);
    CHECK(out2 && p252_hash_batch_multi(ctxs, 3, tag, in, 4, 1, out2, N) == P252_OK && memcmp(out, out2, 32 * N) == 0);
    CHECK(p252_merkle4_tree_multi(ctxs, 3, tag, in, 4 * N, root2) == P252_ERR_INVALID_ARGUMENT); /* 4N/3 is not 4^k */
    /* device-resident entry points + RCCL inside the library, at one rank on the real backend */
    {
        void *d_leaves = NULL, *d_root = NULL, *d_forest = NULL;
        uint64_t root3[4];
        CHECK(hipMalloc(&d_leaves, 32 * 4 * (size_t)N) == 0 && hipMalloc(&d_root, 32) == 0 && hipMalloc(&d_forest, 32 * (size_t)N / 4) == 0);
        CHECK(hipMemcpy(d_leaves, in, 32 * 4 * (size_t)N, 1) == 0);
        /* one process, an array of contexts on distinct devices: ncclCommInitAll, constants broadcast + validated */
        p252_comm *comm = NULL, *dup = NULL;
        CHECK(p252_comm_create_all(&ctx, 1, &comm) == P252_OK && comm != NULL && p252_comm_rank(comm) == 0 && p252_comm_size(comm) == 1);
        CHECK(p252_comm_create_all(&ctx, 1, &dup) == P252_ERR_INVALID_ARGUMENT && dup == NULL); /* a context belongs to one communicator */
        /* subtree -> ncclAllGather of the roots on the stream -> top levels: at one rank the tree's own root */
        CHECK(p252_merkle4_tree_sharded_device(comm, tag, d_leaves, 4 * (size_t)N, d_root, NULL) == P252_OK);
        CHECK(hipDeviceSynchronize() == 0 && hipMemcpy(root3, d_root, 32, 2) == 0 && memcmp(root3, root, 32) == 0);
        CHECK(p252_merkle4_tree_sharded_device(comm, tag, d_leaves, 3, d_root, NULL) == P252_ERR_INVALID_ARGUMENT); /* not 4^k */
        /* ABI 8: no peer failed (p252_comm_check waits for the stream itself; p252_sync reports the same); RCCL was resolved at run
         * time — a C program has one copy, the one the loader finds — and p252_comm_backend says which */
        CHECK(p252_comm_check(comm, NULL) == P252_OK && p252_sync(ctx, NULL) == P252_OK && p252_comm_check(NULL, NULL) == P252_ERR_INVALID_ARGUMENT);
        {
            char where[1024];
            where[0] = 0;
            CHECK(p252_comm_backend(where, sizeof where) == P252_OK && strstr(where, "rccl") != NULL && p252_comm_backend(NULL, 0) == P252_OK);
        }
        p252_comm_destroy(comm);
        /* one process per GPU: rank 0 makes the id, every rank joins with it (here: world = 1) */
        {
            unsigned char id[P252_COMM_ID_BYTES];
            p252_comm* cr = NULL;
            p252_ctx* c2 = NULL;
            CHECK(p252_create(0, &c2) == P252_OK);
            CHECK(p252_comm_unique_id(id, sizeof id) == P252_OK && p252_comm_unique_id(id, 5) == P252_ERR_INVALID_ARGUMENT);
            CHECK(p252_comm_create_rank(c2, id, sizeof id, 0, 1, &cr) == P252_OK && p252_comm_size(cr) == 1);
            memset(root3, 0, 32);
            CHECK(p252_merkle4_tree_sharded_device(cr, tag, d_leaves, 4 * (size_t)N, d_root, NULL) == P252_OK);
            CHECK(hipDeviceSynchronize() == 0 && hipMemcpy(root3, d_root, 32, 2) == 0 && memcmp(root3, root, 32) == 0);
            CHECK(p252_comm_create_rank(c2, id, sizeof id, 1, 1, &cr) == P252_ERR_INVALID_ARGUMENT); /* rank >= world */
            p252_destroy(c2); /* context first: the communicator stays a husk to destroy */
            p252_comm_destroy(cr);
        }
        /* the multi-device entry point: ONE context has nothing to exchange (no communicator is made; ABI 7) */
        {
            const void* dl[1];
            void* dr[1];
            p252_ctx* c3 = NULL;
            CHECK(p252_create(0, &c3) == P252_OK);
            dl[0] = d_leaves;
            dr[0] = d_root;
            memset(root3, 0, 32);
            CHECK(p252_merkle4_tree_multi_device(&c3, 1, tag, dl, 4 * (size_t)N, root3) == P252_OK && memcmp(root3, root, 32) == 0);
            memset(root3, 0, 32);
            CHECK(p252_merkle4_tree_multi_device_resident(&c3, 1, tag, dl, 4 * (size_t)N, dr, NULL) == P252_OK);
            CHECK(hipDeviceSynchronize() == 0 && hipMemcpy(root3, d_root, 32, 2) == 0 && memcmp(root3, root, 32) == 0);
            /* two contexts on ONE device cannot form an RCCL communicator: the resident variant says so ... */
            {
                p252_ctx* two[2];
                const void* dl2[2];
                two[0] = ctxs[0];
                two[1] = ctxs[1];
                dl2[0] = d_leaves;
                dl2[1] = (const char*)d_leaves + 32 * 2 * (size_t)N;
                if (p252_device_count() < 2) {
                    CHECK(p252_merkle4_tree_multi_device_resident(two, 2, tag, dl2, 1024, NULL, NULL) == P252_ERR_COMM);
                    /* ... and the synchronous one gathers the two roots through the host */
                    CHECK(p252_merkle4_tree_multi_device(two, 2, tag, dl2, 1024, root3) == P252_OK);
                }
            }
            {   /* ... so the context is free to join the caller's communicator afterwards, and the same calls then run through it */
                p252_comm* own = NULL;
                CHECK(p252_comm_create_all(&c3, 1, &own) == P252_OK && own != NULL);
                memset(root3, 0, 32);
                CHECK(p252_merkle4_tree_multi_device(&c3, 1, tag, dl, 4 * (size_t)N, root3) == P252_OK && memcmp(root3, root, 32) == 0);
                p252_comm_destroy(own);
            }
            p252_destroy(c3);
        }
        /* forest: 1,024 trees of 16 leaves, one launch per level across all trees = level 2 of the big tree above */
        CHECK(p252_merkle4_forest_device(ctx, tag, d_leaves, (size_t)N / 4, 16, d_forest, NULL, NULL) == P252_OK);
        CHECK(hipDeviceSynchronize() == 0 && hipMemcpy(out2, d_forest, 32 * (size_t)N / 4, 2) == 0);
        CHECK(memcmp(out2, levels + 4 * (size_t)N, 32 * (size_t)N / 4) == 0);
        CHECK(p252_merkle4_forest_device(ctx, tag, d_leaves, 5, 12, d_forest, NULL, NULL) == P252_ERR_INVALID_ARGUMENT);
        memset(out2, 0, 32 * (size_t)N / 4); /* the host-buffer twin: leaves in, roots out */
        CHECK(p252_merkle4_forest(ctx, tag, in, (size_t)N / 4, 16, out2) == P252_OK && memcmp(out2, levels + 4 * (size_t)N, 32 * (size_t)N / 4) == 0);
        CHECK(hipFree(d_leaves) == 0 && hipFree(d_root) == 0 && hipFree(d_forest) == 0);
    }
    /* ABI 7: Hash::digest_truncated for the batch in one launch (hash.rs:164-183, 203-210) == the host rule on the full digests;
     * secret hygiene: the library-owned scratch counts non-zero bytes until it is wiped */
    {
        uint64_t residue = 1;
        CHECK(p252_hash_batch_truncated(ctx, tag, in, 4, 1, out2, N) == P252_OK);
        CHECK(p252_truncate250(out, st_out, N) == P252_OK && memcmp(out2, st_out, 32 * (size_t)N) == 0);
        for (size_t i = 0; i < N; ++i) CHECK((out2[4 * i + 3] >> 58) == 0); /* below 2^250 */
        CHECK(p252_scratch_residue(ctx, &residue) == P252_OK && residue > 0);
        CHECK(p252_wipe(ctx) == P252_OK && p252_scratch_residue(ctx, &residue) == P252_OK && residue == 0);
        CHECK(p252_hash_batch(ctx, tag, in, 4, 1, out2, N) == P252_OK && memcmp(out, out2, 32 * (size_t)N) == 0); /* works as before */
        CHECK(p252_wipe(NULL) == P252_ERR_INVALID_ARGUMENT && p252_scratch_residue(ctx, NULL) == P252_ERR_INVALID_ARGUMENT);
        /* ABI 8: p252_trim frees the grow-only scratch (the next call allocates again) */
        CHECK(p252_trim(ctx) == P252_OK && p252_scratch_residue(ctx, &residue) == P252_OK && residue == 0 && p252_trim(NULL) == P252_ERR_INVALID_ARGUMENT);
        CHECK(p252_hash_batch(ctx, tag, in, 4, 1, out2, N) == P252_OK && memcmp(out, out2, 32 * (size_t)N) == 0);
        CHECK(p252_merkle4_tree(ctx, tag, in, 4 * (size_t)N, root2, NULL) == P252_OK && memcmp(root2, root, 32) == 0 && p252_trim(ctx) == P252_OK);
    }
    /* error paths return codes, nothing unwinds */
    CHECK(p252_hash_batch(ctx, tag, in, 0, 1, out, N) == P252_ERR_INVALID_IO_PATTERN && strlen(p252_last_error(ctx)) > 0);
    CHECK(p252_hash_batch(ctx, tag, NULL, 4, 1, out, N) == P252_ERR_INVALID_ARGUMENT);
    for (int t = 0; t < 4; ++t) p252_destroy(ctxs[t]);
    p252_destroy(ctx);
    free(in); free(out); free(out2); free(st); free(st_out); free(levels);
    printf("ABI SMOKE PASSED\n");
    return 0;
}
