// mad_banks.hip — does v_mad_i64_i32 lose issue cycles to VGPR bank conflicts on gfx950?
// The multiply-add reads src0 (32 b), src1 (32 b) and src2 (64 b = a register pair).  Every stream below is 8 independent
// accumulator chains; what differs is WHICH registers hold the multiplicand(s) relative to the accumulator pairs
// (bank = register number mod 4).  Rates after clock ramp, >= 100 ms per point, like valu_rates.hip.
// build: hipcc --offload-arch=gfx950 -O3 -o mad_banks mad_banks.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

// accumulators: A0..A7 = the register pairs named by the variant; M = multiplicand vgpr(s)
#define KERNEL_(NAME, A0, A1, A2, A3, A4, A5, A6, A7, M0, M1SRC, M1CONS)                                           \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned* sink, unsigned sarg, int niter) {                   \
        unsigned long long a0 = threadIdx.x * 0x9e3779b97f4a7c15ull + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7,    \
                           a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;                                \
        unsigned m0 = threadIdx.x + 9u, m1 = threadIdx.x + 11u;                                                   \
        for (int it = 0; it < niter; ++it) {                                                                      \
            asm volatile(                                                                                         \
                "v_mad_i64_i32 %0, vcc, %8, " M1SRC ", %0\nv_mad_i64_i32 %1, vcc, %8, " M1SRC ", %1\n"              \
                "v_mad_i64_i32 %2, vcc, %8, " M1SRC ", %2\nv_mad_i64_i32 %3, vcc, %8, " M1SRC ", %3\n"              \
                "v_mad_i64_i32 %4, vcc, %8, " M1SRC ", %4\nv_mad_i64_i32 %5, vcc, %8, " M1SRC ", %5\n"              \
                "v_mad_i64_i32 %6, vcc, %8, " M1SRC ", %6\nv_mad_i64_i32 %7, vcc, %8, " M1SRC ", %7\n"              \
                "v_mad_i64_i32 %0, vcc, %8, " M1SRC ", %0\nv_mad_i64_i32 %1, vcc, %8, " M1SRC ", %1\n"              \
                "v_mad_i64_i32 %2, vcc, %8, " M1SRC ", %2\nv_mad_i64_i32 %3, vcc, %8, " M1SRC ", %3\n"              \
                "v_mad_i64_i32 %4, vcc, %8, " M1SRC ", %4\nv_mad_i64_i32 %5, vcc, %8, " M1SRC ", %5\n"              \
                "v_mad_i64_i32 %6, vcc, %8, " M1SRC ", %6\nv_mad_i64_i32 %7, vcc, %8, " M1SRC ", %7\n"              \
                : "+{" A0 "}"(a0), "+{" A1 "}"(a1), "+{" A2 "}"(a2), "+{" A3 "}"(a3), "+{" A4 "}"(a4), "+{" A5 "}"(a5), \
                  "+{" A6 "}"(a6), "+{" A7 "}"(a7)                                                                \
                : "{" M0 "}"(m0), M1CONS(m1), "s"(sarg)                                                           \
                : "vcc");                                                                                         \
        }                                                                                                         \
        unsigned long long s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                             \
        if (s == 12345ull) sink[threadIdx.x] = (unsigned)s;                                                       \
    }

#define KERNEL(...) KERNEL_(__VA_ARGS__)
// accumulator pairs in banks (0,1): v[8:9], v[12:13], ...; in banks (2,3): v[10:11], v[14:15], ...
#define ACC01 "v[8:9]", "v[12:13]", "v[16:17]", "v[20:21]", "v[24:25]", "v[28:29]", "v[32:33]", "v[36:37]"
#define ACC23 "v[10:11]", "v[14:15]", "v[18:19]", "v[22:23]", "v[26:27]", "v[30:31]", "v[34:35]", "v[38:39]"
// accumulators spread over all four alignments (what a register allocator produces): pairs starting at banks 0,2 only
// (64-bit pairs are even-aligned), alternating
#define ACCMIX "v[8:9]", "v[10:11]", "v[12:13]", "v[14:15]", "v[16:17]", "v[18:19]", "v[20:21]", "v[22:23]"
// %8 = vgpr multiplicand, %9 = second multiplicand (vgpr) or unused, %10 = sgpr
KERNEL(s_nc, ACC01, "v2", "%10", "v")      // vgpr (bank 2) x sgpr + acc (banks 0,1): no conflict possible
KERNEL(s_c0, ACC01, "v4", "%10", "v")      // vgpr (bank 0) x sgpr + acc (banks 0,1): multiplicand conflicts with acc low
KERNEL(s_c1, ACC01, "v5", "%10", "v")      // vgpr (bank 1): conflicts with acc high
KERNEL(s_mix, ACCMIX, "v2", "%10", "v")    // accumulators alternate between banks (0,1) and (2,3), multiplicand bank 2
KERNEL(vv_nc, ACC01, "v2", "%9", "{v3}")   // two vgpr multiplicands in banks 2,3 + acc (0,1): all four banks distinct
KERNEL(vv_c, ACC01, "v2", "%9", "{v6}")    // both multiplicands in bank 2
KERNEL(vv_cc, ACC01, "v4", "%9", "{v5}")   // multiplicands in banks 0,1 = the accumulator's banks

struct T {
    const char* name;
    void (*k)(unsigned*, unsigned, int);
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned* sink;
    CHECK(hipMalloc(&sink, 4096));
    T tests[] = {{"vgpr(bank2) x sgpr + acc(banks 0,1)       [no conflict]", k_s_nc},
                 {"vgpr(bank0) x sgpr + acc(banks 0,1)       [conflict with acc.lo]", k_s_c0},
                 {"vgpr(bank1) x sgpr + acc(banks 0,1)       [conflict with acc.hi]", k_s_c1},
                 {"vgpr(bank2) x sgpr + acc alternating      [half conflict]", k_s_mix},
                 {"vgpr(b2) x vgpr(b3) + acc(banks 0,1)      [no conflict]", k_vv_nc},
                 {"vgpr(b2) x vgpr(b2) + acc(banks 0,1)      [multiplicands conflict]", k_vv_c},
                 {"vgpr(b0) x vgpr(b1) + acc(banks 0,1)      [both conflict with acc]", k_vv_cc}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int niter = 8000;
    auto run = [&](void (*k)(unsigned*, unsigned, int), int grid, double ms_target, long* launches_out) {
        float ms = 0;
        long launches = 0;
        CHECK(hipEventRecord(e0));
        do {
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, sink, 12345679u, niter);
            launches += 10;
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        } while (ms < ms_target);
        *launches_out = launches;
        return (double)ms;
    };
    long l;
    run(k_s_nc, cus * 4, 300.0, &l);  // clock ramp
    printf("%-70s %6s %14s %18s\n", "stream of v_mad_i64_i32", "w/SIMD", "Gwave-inst/s", "cycles/inst @2.4GHz");
    for (auto& t : tests)
        for (int kk : {1, 3, 4, 8}) {
            run(k_s_nc, cus * 4, 30.0, &l);
            const double ms = run(t.k, cus * kk, 100.0, &l);
            const double rate = (double)niter * 16 * 4.0 * cus * kk * l / (ms * 1e-3);
            printf("%-70s %6d %14.1f %18.3f\n", t.name, kk, rate / 1e9, 1024.0 * 2.4e9 / rate);
        }
    return 0;
}
