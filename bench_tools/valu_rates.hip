// valu_rates.hip — gfx950 VALU issue-rate microbenchmark.
//
// Why: the Hades permutation is bound by multi-precision multiply throughput; AMD
// publishes no integer-multiply rate for MI355X, so the limb representation (8x32-bit limbs on
// v_mad_u64_u32, vs 24-bit limbs held exactly in FP64 on v_fma_f64) is chosen from these numbers.
// For every instruction: 8 independent dependency chains per lane, NITER x 8 x UNROLL instructions
// per wave, k waves per SIMD (k = 1,2,4).  Reports shader cycles (s_memtime) per wave-instruction
// per SIMD and G-instr/s for the chip.
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

constexpr int NITER = 2000;
constexpr int REP = 4;  // asm block repeated REP times per loop iteration; each block = 8 instrs

#define REP4(x) x x x x

// ---- one kernel per instruction; ASM8 expands to 8 independent instructions ----
#define DEFINE_KERNEL(NAME, TYPE, INIT, ASM8, CLOB)                                         \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned long long* cyc, TYPE* sink,    \
                                                    unsigned sarg) {                        \
        TYPE a0 = INIT(0), a1 = INIT(1), a2 = INIT(2), a3 = INIT(3), a4 = INIT(4),          \
             a5 = INIT(5), a6 = INIT(6), a7 = INIT(7);                                      \
        TYPE b = INIT(9), c = INIT(11);                                                     \
        (void)c; (void)sarg;                                                                \
        unsigned long long t0 = __builtin_readcyclecounter();                               \
        for (int it = 0; it < NITER; ++it) {                                                \
            REP4(asm volatile(ASM8                                                          \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                                "+v"(a6), "+v"(a7)                                          \
                              : "v"(b), "v"(c), "s"(sarg)                                   \
                              : CLOB);)                                                     \
        }                                                                                   \
        unsigned long long t1 = __builtin_readcyclecounter();                               \
        TYPE s = a0;                                                                        \
        s += a1; s += a2; s += a3; s += a4; s += a5; s += a6; s += a7;                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                          \
        if (s == (TYPE)12345) sink[threadIdx.x] = s;                                        \
    }

#define INIT_U32(i) (unsigned)(threadIdx.x * 2654435761u + i * 40503u + 7u)
#define INIT_U64(i) ((unsigned long long)(threadIdx.x * 2654435761u + i * 40503u + 7u) * 0x9e3779b97f4a7c15ull)
#define INIT_F64(i) (double)(threadIdx.x * 3 + i + 1)
#define INIT_F32(i) (float)(threadIdx.x * 3 + i + 1)

// operands: %0..%7 chains, %8 = b (vgpr), %9 = c (vgpr), %10 = sarg (sgpr)
#define A8(fmt_pre, fmt_post)                                                   \
    fmt_pre "%0" fmt_post "\n" fmt_pre "%1" fmt_post "\n" fmt_pre "%2" fmt_post "\n" \
    fmt_pre "%3" fmt_post "\n" fmt_pre "%4" fmt_post "\n" fmt_pre "%5" fmt_post "\n" \
    fmt_pre "%6" fmt_post "\n" fmt_pre "%7" fmt_post "\n"

// 32-bit integer ops: d = op(d, b)
DEFINE_KERNEL(add_u32, unsigned, INIT_U32,
              "v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
              "v_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n", "memory")
DEFINE_KERNEL(add3_u32, unsigned, INIT_U32,
              "v_add3_u32 %0, %0, %8, %9\nv_add3_u32 %1, %1, %8, %9\nv_add3_u32 %2, %2, %8, %9\nv_add3_u32 %3, %3, %8, %9\n"
              "v_add3_u32 %4, %4, %8, %9\nv_add3_u32 %5, %5, %8, %9\nv_add3_u32 %6, %6, %8, %9\nv_add3_u32 %7, %7, %8, %9\n", "memory")
DEFINE_KERNEL(and_b32, unsigned, INIT_U32,
              "v_and_b32 %0, %0, %8\nv_and_b32 %1, %1, %8\nv_and_b32 %2, %2, %8\nv_and_b32 %3, %3, %8\n"
              "v_and_b32 %4, %4, %8\nv_and_b32 %5, %5, %8\nv_and_b32 %6, %6, %8\nv_and_b32 %7, %7, %8\n", "memory")
DEFINE_KERNEL(add_co_u32, unsigned, INIT_U32,
              "v_add_co_u32 %0, vcc, %0, %8\nv_add_co_u32 %1, vcc, %1, %8\nv_add_co_u32 %2, vcc, %2, %8\nv_add_co_u32 %3, vcc, %3, %8\n"
              "v_add_co_u32 %4, vcc, %4, %8\nv_add_co_u32 %5, vcc, %5, %8\nv_add_co_u32 %6, vcc, %6, %8\nv_add_co_u32 %7, vcc, %7, %8\n", "vcc")
DEFINE_KERNEL(addc_co_u32, unsigned, INIT_U32,
              "v_addc_co_u32 %0, vcc, %0, %8, vcc\nv_addc_co_u32 %1, vcc, %1, %8, vcc\nv_addc_co_u32 %2, vcc, %2, %8, vcc\nv_addc_co_u32 %3, vcc, %3, %8, vcc\n"
              "v_addc_co_u32 %4, vcc, %4, %8, vcc\nv_addc_co_u32 %5, vcc, %5, %8, vcc\nv_addc_co_u32 %6, vcc, %6, %8, vcc\nv_addc_co_u32 %7, vcc, %7, %8, vcc\n", "vcc")
DEFINE_KERNEL(mul_lo_u32, unsigned, INIT_U32,
              "v_mul_lo_u32 %0, %0, %8\nv_mul_lo_u32 %1, %1, %8\nv_mul_lo_u32 %2, %2, %8\nv_mul_lo_u32 %3, %3, %8\n"
              "v_mul_lo_u32 %4, %4, %8\nv_mul_lo_u32 %5, %5, %8\nv_mul_lo_u32 %6, %6, %8\nv_mul_lo_u32 %7, %7, %8\n", "memory")
DEFINE_KERNEL(mul_hi_u32, unsigned, INIT_U32,
              "v_mul_hi_u32 %0, %0, %8\nv_mul_hi_u32 %1, %1, %8\nv_mul_hi_u32 %2, %2, %8\nv_mul_hi_u32 %3, %3, %8\n"
              "v_mul_hi_u32 %4, %4, %8\nv_mul_hi_u32 %5, %5, %8\nv_mul_hi_u32 %6, %6, %8\nv_mul_hi_u32 %7, %7, %8\n", "memory")
DEFINE_KERNEL(mad_u32_u24, unsigned, INIT_U32,
              "v_mad_u32_u24 %0, %0, %8, %9\nv_mad_u32_u24 %1, %1, %8, %9\nv_mad_u32_u24 %2, %2, %8, %9\nv_mad_u32_u24 %3, %3, %8, %9\n"
              "v_mad_u32_u24 %4, %4, %8, %9\nv_mad_u32_u24 %5, %5, %8, %9\nv_mad_u32_u24 %6, %6, %8, %9\nv_mad_u32_u24 %7, %7, %8, %9\n", "memory")
DEFINE_KERNEL(mul_hi_u32_u24, unsigned, INIT_U32,
              "v_mul_hi_u32_u24 %0, %0, %8\nv_mul_hi_u32_u24 %1, %1, %8\nv_mul_hi_u32_u24 %2, %2, %8\nv_mul_hi_u32_u24 %3, %3, %8\n"
              "v_mul_hi_u32_u24 %4, %4, %8\nv_mul_hi_u32_u24 %5, %5, %8\nv_mul_hi_u32_u24 %6, %6, %8\nv_mul_hi_u32_u24 %7, %7, %8\n", "memory")
DEFINE_KERNEL(alignbit_b32, unsigned, INIT_U32,
              "v_alignbit_b32 %0, %0, %8, 24\nv_alignbit_b32 %1, %1, %8, 24\nv_alignbit_b32 %2, %2, %8, 24\nv_alignbit_b32 %3, %3, %8, 24\n"
              "v_alignbit_b32 %4, %4, %8, 24\nv_alignbit_b32 %5, %5, %8, 24\nv_alignbit_b32 %6, %6, %8, 24\nv_alignbit_b32 %7, %7, %8, 24\n", "memory")
DEFINE_KERNEL(cndmask_b32, unsigned, INIT_U32,
              "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
              "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n", "memory")

// 64-bit integer: 64-bit chains, b/c 32-bit inputs taken from the low half of a 64-bit VGPR pair
#define DEFINE_KERNEL64(NAME, ASM8, CLOB)                                                   \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned long long* cyc,                \
                                                    unsigned long long* sink, unsigned sarg) { \
        unsigned long long a0 = INIT_U64(0), a1 = INIT_U64(1), a2 = INIT_U64(2),            \
                           a3 = INIT_U64(3), a4 = INIT_U64(4), a5 = INIT_U64(5),            \
                           a6 = INIT_U64(6), a7 = INIT_U64(7);                              \
        unsigned b = INIT_U32(9), c = INIT_U32(11);                                         \
        unsigned long long b64 = INIT_U64(13);                                              \
        unsigned long long t0 = __builtin_readcyclecounter();                               \
        for (int it = 0; it < NITER; ++it) {                                                \
            REP4(asm volatile(ASM8                                                          \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                                "+v"(a6), "+v"(a7)                                          \
                              : "v"(b), "v"(c), "s"(sarg), "v"(b64)                         \
                              : CLOB);)                                                     \
        }                                                                                   \
        unsigned long long t1 = __builtin_readcyclecounter();                               \
        unsigned long long s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                       \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                          \
        if (s == 12345ull) sink[threadIdx.x] = s;                                           \
    }

DEFINE_KERNEL64(mad_u64_u32,
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\nv_mad_u64_u32 %1, vcc, %8, %9, %1\nv_mad_u64_u32 %2, vcc, %8, %9, %2\nv_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                "v_mad_u64_u32 %4, vcc, %8, %9, %4\nv_mad_u64_u32 %5, vcc, %8, %9, %5\nv_mad_u64_u32 %6, vcc, %8, %9, %6\nv_mad_u64_u32 %7, vcc, %8, %9, %7\n", "vcc")
DEFINE_KERNEL64(mad_u64_u32_sgpr,
                "v_mad_u64_u32 %0, vcc, %8, %10, %0\nv_mad_u64_u32 %1, vcc, %8, %10, %1\nv_mad_u64_u32 %2, vcc, %8, %10, %2\nv_mad_u64_u32 %3, vcc, %8, %10, %3\n"
                "v_mad_u64_u32 %4, vcc, %8, %10, %4\nv_mad_u64_u32 %5, vcc, %8, %10, %5\nv_mad_u64_u32 %6, vcc, %8, %10, %6\nv_mad_u64_u32 %7, vcc, %8, %10, %7\n", "vcc")
DEFINE_KERNEL64(lshl_add_u64,
                "v_lshl_add_u64 %0, %0, 0, %11\nv_lshl_add_u64 %1, %1, 0, %11\nv_lshl_add_u64 %2, %2, 0, %11\nv_lshl_add_u64 %3, %3, 0, %11\n"
                "v_lshl_add_u64 %4, %4, 0, %11\nv_lshl_add_u64 %5, %5, 0, %11\nv_lshl_add_u64 %6, %6, 0, %11\nv_lshl_add_u64 %7, %7, 0, %11\n", "memory")
DEFINE_KERNEL64(lshrrev_b64,
                "v_lshrrev_b64 %0, 1, %0\nv_lshrrev_b64 %1, 1, %1\nv_lshrrev_b64 %2, 1, %2\nv_lshrrev_b64 %3, 1, %3\n"
                "v_lshrrev_b64 %4, 1, %4\nv_lshrrev_b64 %5, 1, %5\nv_lshrrev_b64 %6, 1, %6\nv_lshrrev_b64 %7, 1, %7\n", "memory")

// FP64
DEFINE_KERNEL(fma_f64, double, INIT_F64,
              "v_fma_f64 %0, %8, %9, %0\nv_fma_f64 %1, %8, %9, %1\nv_fma_f64 %2, %8, %9, %2\nv_fma_f64 %3, %8, %9, %3\n"
              "v_fma_f64 %4, %8, %9, %4\nv_fma_f64 %5, %8, %9, %5\nv_fma_f64 %6, %8, %9, %6\nv_fma_f64 %7, %8, %9, %7\n", "memory")
DEFINE_KERNEL(add_f64, double, INIT_F64,
              "v_add_f64 %0, %0, %8\nv_add_f64 %1, %1, %8\nv_add_f64 %2, %2, %8\nv_add_f64 %3, %3, %8\n"
              "v_add_f64 %4, %4, %8\nv_add_f64 %5, %5, %8\nv_add_f64 %6, %6, %8\nv_add_f64 %7, %7, %8\n", "memory")
DEFINE_KERNEL(mul_f64, double, INIT_F64,
              "v_mul_f64 %0, %0, %8\nv_mul_f64 %1, %1, %8\nv_mul_f64 %2, %2, %8\nv_mul_f64 %3, %3, %8\n"
              "v_mul_f64 %4, %4, %8\nv_mul_f64 %5, %5, %8\nv_mul_f64 %6, %6, %8\nv_mul_f64 %7, %7, %8\n", "memory")
// FP32
DEFINE_KERNEL(fma_f32, float, INIT_F32,
              "v_fma_f32 %0, %8, %9, %0\nv_fma_f32 %1, %8, %9, %1\nv_fma_f32 %2, %8, %9, %2\nv_fma_f32 %3, %8, %9, %3\n"
              "v_fma_f32 %4, %8, %9, %4\nv_fma_f32 %5, %8, %9, %5\nv_fma_f32 %6, %8, %9, %6\nv_fma_f32 %7, %8, %9, %7\n", "memory")
DEFINE_KERNEL(pk_fma_f32, double, INIT_F64,
              "v_pk_fma_f32 %0, %8, %9, %0\nv_pk_fma_f32 %1, %8, %9, %1\nv_pk_fma_f32 %2, %8, %9, %2\nv_pk_fma_f32 %3, %8, %9, %3\n"
              "v_pk_fma_f32 %4, %8, %9, %4\nv_pk_fma_f32 %5, %8, %9, %5\nv_pk_fma_f32 %6, %8, %9, %6\nv_pk_fma_f32 %7, %8, %9, %7\n", "memory")

// conversions (double dst, u32 src b): chains are broken by construction (dst only) — throughput only
#define DEFINE_KERNEL_CVT(NAME, DT, DINIT, ST, SINIT, ASM8)                                 \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned long long* cyc, DT* sink,      \
                                                    unsigned sarg) {                        \
        DT a0 = DINIT(0), a1 = DINIT(1), a2 = DINIT(2), a3 = DINIT(3), a4 = DINIT(4),       \
           a5 = DINIT(5), a6 = DINIT(6), a7 = DINIT(7);                                     \
        ST b = SINIT(9);                                                                    \
        (void)sarg;                                                                         \
        unsigned long long t0 = __builtin_readcyclecounter();                               \
        for (int it = 0; it < NITER; ++it) {                                                \
            REP4(asm volatile(ASM8                                                          \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                                "+v"(a6), "+v"(a7)                                          \
                              : "v"(b)                                                      \
                              : "memory");)                                                 \
        }                                                                                   \
        unsigned long long t1 = __builtin_readcyclecounter();                               \
        DT s = a0;                                                                          \
        s += a1; s += a2; s += a3; s += a4; s += a5; s += a6; s += a7;                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                          \
        if (s == (DT)12345) sink[threadIdx.x] = s;                                          \
    }
DEFINE_KERNEL_CVT(cvt_f64_u32, double, INIT_F64, unsigned, INIT_U32,
                  "v_cvt_f64_u32 %0, %8\nv_cvt_f64_u32 %1, %8\nv_cvt_f64_u32 %2, %8\nv_cvt_f64_u32 %3, %8\n"
                  "v_cvt_f64_u32 %4, %8\nv_cvt_f64_u32 %5, %8\nv_cvt_f64_u32 %6, %8\nv_cvt_f64_u32 %7, %8\n")
DEFINE_KERNEL_CVT(cvt_i32_f64, unsigned, INIT_U32, double, INIT_F64,
                  "v_cvt_i32_f64 %0, %8\nv_cvt_i32_f64 %1, %8\nv_cvt_i32_f64 %2, %8\nv_cvt_i32_f64 %3, %8\n"
                  "v_cvt_i32_f64 %4, %8\nv_cvt_i32_f64 %5, %8\nv_cvt_i32_f64 %6, %8\nv_cvt_i32_f64 %7, %8\n")

struct Entry {
    const char* name;
    void (*launch)(int grid, unsigned long long* cyc, void* sink);
};

#define LAUNCHER(NAME, TYPE)                                                        \
    static void l_##NAME(int grid, unsigned long long* cyc, void* sink) {           \
        hipLaunchKernelGGL(k_##NAME, dim3(grid), dim3(256), 0, 0, cyc, (TYPE*)sink, 12345679u); \
    }
LAUNCHER(add_u32, unsigned) LAUNCHER(add3_u32, unsigned) LAUNCHER(and_b32, unsigned)
LAUNCHER(add_co_u32, unsigned) LAUNCHER(addc_co_u32, unsigned) LAUNCHER(mul_lo_u32, unsigned)
LAUNCHER(mul_hi_u32, unsigned) LAUNCHER(mad_u32_u24, unsigned) LAUNCHER(mul_hi_u32_u24, unsigned)
LAUNCHER(alignbit_b32, unsigned) LAUNCHER(cndmask_b32, unsigned)
LAUNCHER(mad_u64_u32, unsigned long long) LAUNCHER(mad_u64_u32_sgpr, unsigned long long)
LAUNCHER(lshl_add_u64, unsigned long long) LAUNCHER(lshrrev_b64, unsigned long long)
LAUNCHER(fma_f64, double) LAUNCHER(add_f64, double) LAUNCHER(mul_f64, double)
LAUNCHER(fma_f32, float) LAUNCHER(pk_fma_f32, double)
LAUNCHER(cvt_f64_u32, double) LAUNCHER(cvt_i32_f64, unsigned)

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    unsigned long long* cyc;
    void* sink;
    CHECK(hipMalloc(&cyc, 8));
    CHECK(hipMalloc(&sink, 256 * 8));
    std::vector<Entry> tests = {
        {"v_add_u32", l_add_u32},           {"v_add3_u32", l_add3_u32},
        {"v_and_b32", l_and_b32},           {"v_add_co_u32", l_add_co_u32},
        {"v_addc_co_u32", l_addc_co_u32},   {"v_cndmask_b32", l_cndmask_b32},
        {"v_alignbit_b32", l_alignbit_b32}, {"v_mul_lo_u32", l_mul_lo_u32},
        {"v_mul_hi_u32", l_mul_hi_u32},     {"v_mad_u32_u24", l_mad_u32_u24},
        {"v_mul_hi_u32_u24", l_mul_hi_u32_u24},
        {"v_mad_u64_u32", l_mad_u64_u32},   {"v_mad_u64_u32(sgpr)", l_mad_u64_u32_sgpr},
        {"v_lshl_add_u64", l_lshl_add_u64}, {"v_lshrrev_b64", l_lshrrev_b64},
        {"v_fma_f64", l_fma_f64},           {"v_add_f64", l_add_f64},
        {"v_mul_f64", l_mul_f64},           {"v_fma_f32", l_fma_f32},
        {"v_pk_fma_f32", l_pk_fma_f32},     {"v_cvt_f64_u32", l_cvt_f64_u32},
        {"v_cvt_i32_f64", l_cvt_i32_f64},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double n_inst_per_wave = (double)NITER * REP * 8;
    printf("%-22s %6s %14s %14s %12s\n", "instr", "w/SIMD", "cyc/inst/SIMD", "Ginst/s(chip)", "eff.GHz");
    for (auto& t : tests) {
        for (int k : {1, 2, 4}) {
            int grid = cus * k;  // 256 threads = 4 waves = one wave per SIMD per block
            t.launch(grid, cyc, sink);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            t.launch(grid, cyc, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c;
            CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            double cyc_per_inst_simd = (double)c / (n_inst_per_wave * k);
            double total_wave_inst = n_inst_per_wave * 4.0 * grid;
            double ginst = total_wave_inst / (ms * 1e-3) / 1e9;
            double ghz = (double)c / (ms * 1e-3) / 1e9;  // s_memtime ticks per second (kernel ≈ loop)
            printf("%-22s %6d %14.2f %14.2f %12.3f\n", t.name, k, cyc_per_inst_simd, ginst, ghz);
        }
    }
    return 0;
}
