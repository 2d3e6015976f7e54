/* abi_smoke.c — include/poseidon252_hip.h consumed from plain C (gcc -std=c11 -pedantic -Werror): proves the header is
 * valid C, links against libposeidon252_hip.so and drives the hot path through the C ABI exactly as a cgo / Rust-FFI /
 * C caller would.  Without a HIP device it checks the host helpers and that compute fails loudly (exit 0, "no device");
 * with one it hashes, permutes, builds trees (single and multi-context) and cross-checks them against each other. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "poseidon252_hip.h"

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
    /* error paths return codes, nothing unwinds */
    CHECK(p252_hash_batch(ctx, tag, in, 0, 1, out, N) == P252_ERR_INVALID_IO_PATTERN && strlen(p252_last_error(ctx)) > 0);
    CHECK(p252_hash_batch(ctx, tag, NULL, 4, 1, out, N) == P252_ERR_INVALID_ARGUMENT);
    for (int t = 0; t < 4; ++t) p252_destroy(ctxs[t]);
    p252_destroy(ctx);
    free(in); free(out); free(out2); free(st); free(st_out); free(levels);
    printf("ABI SMOKE PASSED\n");
    return 0;
}
