// sbox_variants.hip — VERDICT r3 item 4, candidate (a), priced by MEASUREMENT on the library's own arithmetic (fr29.hpp):
// "balanced carried digits (|d| <= 2^28) for x, so that x^4 can leave as a wide digit into x^4 * x".
//
// The body timed is one partial round's generic work: x <- carried( x^5 * G ), i.e. sbox_w<true>(x) then one product by a
// constant — 4 products, 4 reductions, 521 + 187 VALU instructions of the 826 a partial round issues (the rest, the integer
// recurrence, is the same in both variants).
//   cur   the shipped code: x carried in [0, 2^29) (8 digits x {64-bit add, and, 64-bit shift}: 24 instructions, 80 issue
//         cycles), x^2 and x^4 carried, x^5 and W wide
//   bal   the candidate: the producing chain emits BALANCED digits (the +2^28 bias of every digit rides in the accumulator's
//         initial value — free, it would come from the additive constants — and each digit pays one v_sub: 32 instructions,
//         96 cycles), x^4 leaves as WIDE digits (16 instructions, 48 cycles instead of 24 / 80): x^4 * x then fills a column
//         to 9 x 2^31 x 2^28 + 2^60.4 = 2^62.55 < 2^63.
// Predicted by the 4- / 2-cycle issue model: (80 - 48) - (96 - 80) = 16 cycles of 2,880 per partial round, 0.55 %.
// The two variants compute the same field element (checked: both chains end in the same canonical scalar).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -I poseidon252_amd/csrc -o bench_tools/sbox_variants bench_tools/sbox_variants.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fr29.hpp"

using namespace p252;

#define CHECK(x)                                                                                          \
    do {                                                                                                  \
        hipError_t e_ = (x);                                                                              \
        if (e_ != hipSuccess) {                                                                           \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            exit(1);                                                                                      \
        }                                                                                                 \
    } while (0)

// G: some 9-digit constant with balanced digits (|g| <= 2^28), as the tables hold them (SGPR operands in the kernels)
__constant__ int32_t GC[NL] = {-112233445, 98765432, -55555555, 123456789, -87654321, 33333333, -101010101, 77777777, 1234567};

// nine wide steps, then the carry chain with BALANCED digits: the high columns were initialised with a bias of 2^28 each
// (acc_zero_bal), digit = (v & mask) - 2^28 in [-2^28, 2^28)
__device__ __forceinline__ E29 redc_w_bal(A29& t, const RK& K) {
    P252_WSTEP(t.c, 0, 2 * NL, K)
    P252_WSTEP(t.c, 1, 2 * NL, K)
    P252_WSTEP(t.c, 2, 2 * NL, K)
    P252_WSTEP(t.c, 3, 2 * NL, K)
    P252_WSTEP(t.c, 4, 2 * NL, K)
    P252_WSTEP(t.c, 5, 2 * NL, K)
    P252_WSTEP(t.c, 6, 2 * NL, K)
    P252_WSTEP(t.c, 7, 2 * NL, K)
    P252_WSTEP(t.c, 8, 2 * NL, K)
    E29 r;
    int64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NL - 1; ++k) {
        const int64_t v = t.c[NL + k] + carry;
        r.d[k] = opaque_digit((int32_t)((uint32_t)v & DMASK) - (1 << 28));
        carry = v >> WB;
    }
    r.d[NL - 1] = opaque_digit((int32_t)(t.c[2 * NL - 1] + carry));
    return r;
}
__device__ __forceinline__ void acc_zero_bal(A29& t, const RK& K, int64_t bias28) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        t.c[k] = K.bias;
        t.c[NL + k] = k < NL - 1 ? bias28 : (int64_t)0;
    }
}

template <bool BAL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_chain(uint32_t* out, int iters) {
    const RK K = make_rk();
    int64_t bias28 = (int64_t)1 << 28;
    asm("" : "+v"(bias28));  // a register pair that lives for the whole kernel, like K.bias
    E29 x;
#pragma unroll
    for (int k = 0; k < NL; ++k) x.d[k] = (int32_t)(((threadIdx.x + 256u * blockIdx.x) * 2654435761u + 12345u * (k + 1)) & (DMASK >> (k == NL - 1 ? 6 : 1)));
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        A29 t;
        acc_zero_w(t, K);
        acc_sqr(t, x);
        const E29 x2 = redc_w(t, K);
        acc_zero_w<BAL>(t, K);  // (BAL: the wide-output bias in the high columns)
        acc_sqr(t, x2);
        const E29 x4 = redc_w<BAL>(t, K);
        acc_zero_w<true>(t, K);
        acc_mul(t, x4, x.d);
        const E29 x5 = redc_w<true>(t, K);
        // W = x^5 * G, leaving as the next round's S-box input (stands for the recurrence's carry chain)
        if (BAL) {
            acc_zero_bal(t, K, bias28);
            acc_mul(t, x5, GC);
            x = redc_w_bal(t, K);
        } else {
            acc_zero_w(t, K);
            acc_mul(t, x5, GC);
            x = redc_w(t, K);
        }
    }
    // canonical form of the final value (tight reduction of x * 1, then to_mont4) for the cross-check of the variants
    E29 one = e29_zero();
    one.d[0] = 1;
    A29 t;
    acc_zero(t);
    acc_mul(t, x, one.d);
    const E29 r = redc(t);
    uint32_t w[8];
    to_mont4<5>(r, w);
    const unsigned gid = threadIdx.x + 256u * blockIdx.x;
    if (gid < 64)
#pragma unroll
        for (int k = 0; k < 8; ++k) out[gid * 8 + k] = w[k];
}

template <bool BAL>
static double run(int blocks, int iters, uint32_t* out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_chain<BAL>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_chain<BAL>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    uint32_t* out;
    CHECK(hipMalloc(&out, 64 * 8 * 4));
    uint32_t a[512], b[512];
    hipLaunchKernelGGL(k_chain<false>, dim3(1), dim3(256), 0, 0, out, 37);
    CHECK(hipMemcpy(a, out, sizeof a, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_chain<true>, dim3(1), dim3(256), 0, 0, out, 37);
    CHECK(hipMemcpy(b, out, sizeof b, hipMemcpyDeviceToHost));
    printf("# the two variants end in the same canonical scalars after 37 rounds: %s\n", memcmp(a, b, sizeof a) == 0 ? "yes" : "NO");
    const int iters = 2000;
    printf("# bench_tools/sbox_variants: %d x (S-box + product by a constant) per lane, 256-thread blocks, 3 waves per SIMD held by attribute\n", iters);
    printf("# columns: waves per SIMD | cur: ms, ns per round body | bal: ms, ns | cur / bal\n");
    for (int rep = 0; rep < 3; ++rep)
        for (int wps : {1, 3, 6}) {  // lone waves, the kernels' occupancy, two full rounds of it
            const int blocks = 256 * wps;  // 1024 SIMDs x wps waves / 4 waves per block
            const double ta = run<false>(blocks, iters, out), tb = run<true>(blocks, iters, out);
            printf("%d waves/SIMD   cur %8.3f ms %7.1f ns   bal %8.3f ms %7.1f ns   ratio %.4f\n", wps, ta, ta * 1e6 / iters, tb, tb * 1e6 / iters, ta / tb);
        }
    return 0;
}
