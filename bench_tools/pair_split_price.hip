// pair_split_price.hip — what would splitting ONE field product over a DPP lane pair buy a lone wave?  
//
// The lane-group kernels (coop29.hpp) are bound by one wave's instruction count: a lone wave issues one VALU instruction per
// 5.1-6.8 cycles whatever its class, and a digest is a chain of 209 products with their reductions.  A pair split of a
// product is limited by lockstep: both lanes of a pair execute ONE instruction stream, so only the product itself divides
// (each lane a parallelogram of 54 of the 81 digit products, over rotated operand digits); the digit-serial reduction is
// issued in full by both lanes, plus one DPP move per exchanged digit.  This program times exactly those two instruction
// streams with the library's own arithmetic (fr29.hpp) on lone waves — one wave per SIMD, a dependent chain of
// product + reduction — so that the price is measured, not estimated:
//   single   x <- redc_w(x * y)                                  81 multiply-adds + 114
//   pair     the same reduction, a 54-multiply-add product, 9 + 9 DPP moves (operand exchange before, digit exchange after)
// The pair stream does not compute a product (the operand rotation is not built); its instruction mix and dependency
// structure are the split's.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I poseidon252_amd/csrc -o pair_split_price bench_tools/pair_split_price.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

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

__device__ __forceinline__ E29 swap1(const E29& v) {
    E29 r;
#pragma unroll
    for (int k = 0; k < NL; ++k) r.d[k] = __builtin_amdgcn_mov_dpp(v.d[k], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
    return r;
}

template <bool PAIR>
__global__ void __launch_bounds__(64) k_chain(int32_t* sink, int iters) {
    const RK K = make_rk();
    E29 x, y;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        x.d[k] = (int32_t)((threadIdx.x * 2654435761u + 12345u * (k + 1)) & DMASK);
        y.d[k] = (int32_t)((threadIdx.x * 40503u + 977u * (k + 3)) & DMASK);
    }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        A29 t;
        acc_zero_w(t, K);
        if (!PAIR) {
            acc_mul(t, x, y.d);
        } else {
            const E29 xo = swap1(x);  // the partner's digits of the operand (9 DPP moves)
            // a parallelogram of 54 digit products: six of the nine multiplier digits against all nine multiplicand digits
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int64_t bj = y.d[j];
#pragma unroll
                for (int i = 0; i < NL; ++i) t.c[i + j] += (int64_t)(j & 1 ? xo.d[i] : x.d[i]) * bj;
            }
        }
        E29 r = redc_w(t, K);
        if (PAIR) {
            const E29 ro = swap1(r);  // the partner's half of the result digits (9 DPP moves)
#pragma unroll
            for (int k = 0; k < NL; ++k) r.d[k] = (k & 1) ? ro.d[k] : r.d[k];
        }
        x = r;
    }
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) s ^= x.d[k];
    if (s == 0x7fffffff) sink[threadIdx.x] = s;
}

template <bool PAIR>
static double run(int blocks, int iters, int32_t* sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_chain<PAIR>, dim3(blocks), dim3(64), 0, 0, sink, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_chain<PAIR>, dim3(blocks), dim3(64), 0, 0, sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    int32_t* sink;
    CHECK(hipMalloc(&sink, 4096));
    const int iters = 20000;
    printf("# bench_tools/pair_split_price: a dependent chain of %d (product + wide reduction) per lane, 64-thread blocks\n", iters);
    printf("# columns: waves on the chip | single: ms, ns per product+reduction | pair-split stream: ms, ns | single / pair\n");
    for (int blocks : {256, 1024, 2048}) {  // a quarter of the SIMDs, one wave per SIMD, two per SIMD
        const double a = run<false>(blocks, iters, sink), b = run<true>(blocks, iters, sink);
        printf("%5d waves   single %8.3f ms %7.1f ns   pair %8.3f ms %7.1f ns   ratio %.3f\n", blocks, a, a * 1e6 / iters, b, b * 1e6 / iters, a / b);
    }
    return 0;
}
