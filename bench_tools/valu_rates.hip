// valu_rates.hip — gfx950 VALU issue-rate microbenchmark (the `peak` of bench.py's roofline comes from here).
//
// Why: the Hades permutation is bound by multi-precision multiply throughput; AMD publishes no
// integer-multiply rate for MI355X.  For every instruction: 8 independent dependency chains per lane,
// NITER x REP x 8 instructions per wave, k waves per SIMD (k = 1, 2, 4).
//
// Round-2 rework: the round-1 run timed ONE launch of ~64 k instructions per test, i.e. before the
// clocks had ramped — the kernel it was meant to bound sustained a higher rate than the "peak".  Now every test is
// preceded by a ramp (>= 150 ms of continuous multiply-add issue at the start, the same kernel again before each
// measurement) and is measured over >= 100 ms of back-to-back launches.  Reported per test: wave-instructions/s for
// the chip, and that rate divided by (1024 SIMDs x 2.4 GHz) = instructions per SIMD-cycle at the nominal clock.
// `--latency` adds ONE dependent chain per wave at 1 wave/SIMD: the issue-to-issue latency a lone wave sees.
// Run under `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` to get the real clock (cycles / duration) beside it.
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

constexpr int REP = 4;  // asm block repeated REP times per loop iteration; each block = 8 instrs
#define REP4(x) x x x x

// 32-bit chains: %0..%7 chains, %8 = b (vgpr), %9 = c (vgpr), %10 = sarg (sgpr)
#define DEFINE_KERNEL(NAME, ASM8, CLOB)                                                     \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned* sink, unsigned sarg, int niter) { \
        unsigned a0 = threadIdx.x * 2654435761u + 7u, a1 = a0 + 40503u, a2 = a1 + 40503u, a3 = a2 + 40503u, \
                 a4 = a3 + 40503u, a5 = a4 + 40503u, a6 = a5 + 40503u, a7 = a6 + 40503u;   \
        unsigned b = a7 + 9u, c = a7 + 11u;                                                 \
        for (int it = 0; it < niter; ++it) {                                                \
            REP4(asm volatile(ASM8                                                          \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                                "+v"(a6), "+v"(a7)                                          \
                              : "v"(b), "v"(c), "s"(sarg)                                   \
                              : CLOB);)                                                     \
        }                                                                                   \
        unsigned s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                 \
        if (s == 12345u) sink[threadIdx.x] = s;                                             \
    }
// 64-bit chains (VGPR pairs): %0..%7 chains, %8 = b, %9 = c (32-bit vgprs), %10 = sarg, %11 = 64-bit vgpr pair
#define DEFINE_KERNEL64(NAME, ASM8, ...)                                                  \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned* sink, unsigned sarg, int niter) { \
        unsigned long long a0 = (threadIdx.x * 2654435761u + 7u) * 0x9e3779b97f4a7c15ull, a1 = a0 * 3, a2 = a0 * 5, \
                           a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15; \
        unsigned b = (unsigned)a0 + 9u, c = (unsigned)a0 + 11u;                             \
        unsigned long long b64 = a0 * 17;                                                   \
        for (int it = 0; it < niter; ++it) {                                                \
            REP4(asm volatile(ASM8                                                          \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                                "+v"(a6), "+v"(a7)                                          \
                              : "v"(b), "v"(c), "s"(sarg), "v"(b64)                         \
                              : __VA_ARGS__);)                                                     \
        }                                                                                   \
        unsigned long long s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                       \
        if (s == 12345ull) sink[threadIdx.x] = (unsigned)s;                                 \
    }

#define X8(pre, post) pre "%0" post "\n" pre "%1" post "\n" pre "%2" post "\n" pre "%3" post "\n" \
                      pre "%4" post "\n" pre "%5" post "\n" pre "%6" post "\n" pre "%7" post "\n"
// d = op(d, b)   /  d = op(d, b, c)  / d = op(b, c, d)
#define OP2(NAME, MN) DEFINE_KERNEL(NAME, \
    MN " %0, %0, %8\n" MN " %1, %1, %8\n" MN " %2, %2, %8\n" MN " %3, %3, %8\n" \
    MN " %4, %4, %8\n" MN " %5, %5, %8\n" MN " %6, %6, %8\n" MN " %7, %7, %8\n", "memory")
#define OP3(NAME, MN) DEFINE_KERNEL(NAME, \
    MN " %0, %0, %8, %9\n" MN " %1, %1, %8, %9\n" MN " %2, %2, %8, %9\n" MN " %3, %3, %8, %9\n" \
    MN " %4, %4, %8, %9\n" MN " %5, %5, %8, %9\n" MN " %6, %6, %8, %9\n" MN " %7, %7, %8, %9\n", "memory")
#define OP3ACC(NAME, MN) DEFINE_KERNEL(NAME, \
    MN " %0, %8, %9, %0\n" MN " %1, %8, %9, %1\n" MN " %2, %8, %9, %2\n" MN " %3, %8, %9, %3\n" \
    MN " %4, %8, %9, %4\n" MN " %5, %8, %9, %5\n" MN " %6, %8, %9, %6\n" MN " %7, %8, %9, %7\n", "memory")
// shifts with the "rev" operand order: d = d shifted by an inline constant
#define OPSH(NAME, MN) DEFINE_KERNEL(NAME, \
    MN " %0, 3, %0\n" MN " %1, 3, %1\n" MN " %2, 3, %2\n" MN " %3, 3, %3\n" \
    MN " %4, 3, %4\n" MN " %5, 3, %5\n" MN " %6, 3, %6\n" MN " %7, 3, %7\n", "memory")

OP2(add_u32, "v_add_u32")
OP2(and_b32, "v_and_b32")
OP2(xor_b32, "v_xor_b32")
OP2(mul_lo_u32, "v_mul_lo_u32")
OP2(mul_hi_u32, "v_mul_hi_u32")
OP2(mul_i32_i24, "v_mul_i32_i24")
OPSH(lshlrev_b32, "v_lshlrev_b32")
OPSH(lshrrev_b32, "v_lshrrev_b32")
OPSH(ashrrev_i32, "v_ashrrev_i32")
OP3(add3_u32, "v_add3_u32")
OP3(lshl_add_u32, "v_lshl_add_u32")
OP3(and_or_b32, "v_and_or_b32")
OP3(mad_u32_u24, "v_mad_u32_u24")
OP3(mad_i32_i24, "v_mad_i32_i24")
OP3ACC(fma_f32, "v_fma_f32")
DEFINE_KERNEL(mov_b32, X8("v_mov_b32 ", ", %8"), "memory")
DEFINE_KERNEL(bfe_i32, X8("v_bfe_i32 ", ", %8, 0, 29"), "memory")
DEFINE_KERNEL(bfe_u32, X8("v_bfe_u32 ", ", %8, 3, 29"), "memory")
DEFINE_KERNEL(alignbit_b32,
              "v_alignbit_b32 %0, %0, %8, 29\nv_alignbit_b32 %1, %1, %8, 29\nv_alignbit_b32 %2, %2, %8, 29\nv_alignbit_b32 %3, %3, %8, 29\n"
              "v_alignbit_b32 %4, %4, %8, 29\nv_alignbit_b32 %5, %5, %8, 29\nv_alignbit_b32 %6, %6, %8, 29\nv_alignbit_b32 %7, %7, %8, 29\n", "memory")
DEFINE_KERNEL(add_co_u32,
              "v_add_co_u32 %0, vcc, %0, %8\nv_add_co_u32 %1, vcc, %1, %8\nv_add_co_u32 %2, vcc, %2, %8\nv_add_co_u32 %3, vcc, %3, %8\n"
              "v_add_co_u32 %4, vcc, %4, %8\nv_add_co_u32 %5, vcc, %5, %8\nv_add_co_u32 %6, vcc, %6, %8\nv_add_co_u32 %7, vcc, %7, %8\n", "vcc")
DEFINE_KERNEL(mov_dpp_quad,
              "v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
              "v_mov_b32_dpp %2, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
              "v_mov_b32_dpp %4, %4 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
              "v_mov_b32_dpp %6, %6 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n", "memory")

// (the X8 helper cannot place the accumulator last; spell the 64-bit ones out)
#define ACC64(NAME, MN, SRC1) DEFINE_KERNEL64(NAME, \
    MN " %0, vcc, %8, " SRC1 ", %0\n" MN " %1, vcc, %8, " SRC1 ", %1\n" MN " %2, vcc, %8, " SRC1 ", %2\n" MN " %3, vcc, %8, " SRC1 ", %3\n" \
    MN " %4, vcc, %8, " SRC1 ", %4\n" MN " %5, vcc, %8, " SRC1 ", %5\n" MN " %6, vcc, %8, " SRC1 ", %6\n" MN " %7, vcc, %8, " SRC1 ", %7\n", "vcc")
ACC64(mad_u64_u32_v, "v_mad_u64_u32", "%9")
ACC64(mad_i64_i32_v, "v_mad_i64_i32", "%9")
ACC64(mad_i64_i32_s, "v_mad_i64_i32", "%10")
DEFINE_KERNEL64(lshl_add_u64,
                "v_lshl_add_u64 %0, %0, 0, %11\nv_lshl_add_u64 %1, %1, 0, %11\nv_lshl_add_u64 %2, %2, 0, %11\nv_lshl_add_u64 %3, %3, 0, %11\n"
                "v_lshl_add_u64 %4, %4, 0, %11\nv_lshl_add_u64 %5, %5, 0, %11\nv_lshl_add_u64 %6, %6, 0, %11\nv_lshl_add_u64 %7, %7, 0, %11\n", "memory")
DEFINE_KERNEL64(ashrrev_i64,
                "v_ashrrev_i64 %0, 1, %0\nv_ashrrev_i64 %1, 1, %1\nv_ashrrev_i64 %2, 1, %2\nv_ashrrev_i64 %3, 1, %3\n"
                "v_ashrrev_i64 %4, 1, %4\nv_ashrrev_i64 %5, 1, %5\nv_ashrrev_i64 %6, 1, %6\nv_ashrrev_i64 %7, 1, %7\n", "memory")
DEFINE_KERNEL64(fma_f64,
                "v_fma_f64 %0, %11, %11, %0\nv_fma_f64 %1, %11, %11, %1\nv_fma_f64 %2, %11, %11, %2\nv_fma_f64 %3, %11, %11, %3\n"
                "v_fma_f64 %4, %11, %11, %4\nv_fma_f64 %5, %11, %11, %5\nv_fma_f64 %6, %11, %11, %6\nv_fma_f64 %7, %11, %11, %7\n", "memory")
// the two Montgomery digit steps of fr29.hpp as instruction mixes (8 chains = the 8 columns a step touches):
//   tight (round 1): v_and + 8 MAD + v_ashrrev_i64 + v_lshl_add_u64      wide (round 2): v_xor + 9 MAD
DEFINE_KERNEL64(mix_step_tight,
                "v_and_b32 %8, 0x1fffffff, %8\n"
                "v_mad_i64_i32 %0, vcc, %8, %10, %0\nv_mad_i64_i32 %1, vcc, %8, %10, %1\nv_mad_i64_i32 %2, vcc, %8, %10, %2\nv_mad_i64_i32 %3, vcc, %8, %10, %3\n"
                "v_mad_i64_i32 %4, vcc, %8, %10, %4\nv_mad_i64_i32 %5, vcc, %8, %10, %5\nv_mad_i64_i32 %6, vcc, %8, %10, %6\nv_mad_i64_i32 %7, vcc, %8, %10, %7\n"
                "v_ashrrev_i64 %11, 29, %11\nv_lshl_add_u64 %0, %0, 0, %11\n", "vcc")
DEFINE_KERNEL64(mix_step_wide,
                "v_xor_b32 %8, 0x80000000, %8\n"
                "v_mad_i64_i32 %0, vcc, %8, %10, %0\nv_mad_i64_i32 %1, vcc, %8, %10, %1\nv_mad_i64_i32 %2, vcc, %8, %10, %2\nv_mad_i64_i32 %3, vcc, %8, %10, %3\n"
                "v_mad_i64_i32 %4, vcc, %8, %10, %4\nv_mad_i64_i32 %5, vcc, %8, %10, %5\nv_mad_i64_i32 %6, vcc, %8, %10, %6\nv_mad_i64_i32 %7, vcc, %8, %10, %7\n"
                "v_mad_i64_i32 %0, vcc, %9, %10, %0\n", "vcc")

// Does the carry-out register matter?  v_mad_i64_i32 / v_add_co_u32 (VOP3b) write an SGPR pair; every stream above names
// vcc.  Same streams with the carry-outs rotating over 8 / 2 different SGPR pairs (--sdst).
#define MAD_SD(A, SD) "v_mad_i64_i32 " A ", " SD ", %8, %10, " A "\n"
DEFINE_KERNEL64(mad_sdst_rot8,
                MAD_SD("%0", "s[36:37]") MAD_SD("%1", "s[38:39]") MAD_SD("%2", "s[40:41]") MAD_SD("%3", "s[42:43]")
                MAD_SD("%4", "s[44:45]") MAD_SD("%5", "s[46:47]") MAD_SD("%6", "s[48:49]") MAD_SD("%7", "s[50:51]"),
                "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51")
DEFINE_KERNEL64(mad_sdst_rot2,
                MAD_SD("%0", "s[36:37]") MAD_SD("%1", "s[38:39]") MAD_SD("%2", "s[36:37]") MAD_SD("%3", "s[38:39]")
                MAD_SD("%4", "s[36:37]") MAD_SD("%5", "s[38:39]") MAD_SD("%6", "s[36:37]") MAD_SD("%7", "s[38:39]"),
                "s36", "s37", "s38", "s39")
DEFINE_KERNEL64(mad_sdst_one,
                MAD_SD("%0", "s[36:37]") MAD_SD("%1", "s[36:37]") MAD_SD("%2", "s[36:37]") MAD_SD("%3", "s[36:37]")
                MAD_SD("%4", "s[36:37]") MAD_SD("%5", "s[36:37]") MAD_SD("%6", "s[36:37]") MAD_SD("%7", "s[36:37]"),
                "s36", "s37")
DEFINE_KERNEL64(mix_step_wide_rot,
                "v_xor_b32 %8, 0x80000000, %8\n"
                MAD_SD("%0", "s[36:37]") MAD_SD("%1", "s[38:39]") MAD_SD("%2", "s[40:41]") MAD_SD("%3", "s[42:43]")
                MAD_SD("%4", "s[44:45]") MAD_SD("%5", "s[46:47]") MAD_SD("%6", "s[48:49]") MAD_SD("%7", "s[50:51]")
                "v_mad_i64_i32 %0, s[52:53], %9, %10, %0\n",
                "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53")
// interleaved with 2-cycle work: does a plain op between two multiply-adds hide the carry-out's extra cycle?
DEFINE_KERNEL64(mad_and_interleaved,
                "v_mad_i64_i32 %0, vcc, %8, %10, %0\nv_and_b32 %9, %9, %8\nv_mad_i64_i32 %1, vcc, %8, %10, %1\nv_and_b32 %9, %9, %8\n"
                "v_mad_i64_i32 %2, vcc, %8, %10, %2\nv_and_b32 %9, %9, %8\nv_mad_i64_i32 %3, vcc, %8, %10, %3\nv_and_b32 %9, %9, %8\n", "vcc")

// one dependent chain per wave: what a lone wave pays per instruction (the tree's narrow levels, small batches)
#define DEFINE_LAT(NAME, TYPE, ASM1, CLOB)                                                  \
    __global__ void __launch_bounds__(256) k_lat_##NAME(unsigned* sink, unsigned sarg, int niter) { \
        TYPE a0 = (TYPE)(threadIdx.x * 2654435761u + 7u);                                   \
        unsigned b = threadIdx.x + 9u, c = threadIdx.x + 11u;                               \
        for (int it = 0; it < niter; ++it) {                                                \
            REP4(asm volatile(ASM1 ASM1 ASM1 ASM1 ASM1 ASM1 ASM1 ASM1                       \
                              : "+v"(a0) : "v"(b), "v"(c), "s"(sarg) : CLOB);)              \
        }                                                                                   \
        if (a0 == (TYPE)12345) sink[threadIdx.x] = (unsigned)a0;                            \
    }
DEFINE_LAT(mad_i64_i32, unsigned long long, "v_mad_i64_i32 %0, vcc, %1, %3, %0\n", "vcc")
DEFINE_LAT(and_b32, unsigned, "v_and_b32 %0, %0, %1\n", "memory")
DEFINE_LAT(lshl_add_u64, unsigned long long, "v_lshl_add_u64 %0, %0, 0, %0\n", "memory")

struct Entry {
    const char* name;
    void (*kern)(unsigned*, unsigned, int);
    double per_block;  // instructions per asm block (8 for pure streams, 11 / 10 for the mixes)
    bool latency;
};

int main(int argc, char** argv) {
    bool with_latency = false, clock_mode = false, sdst_mode = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--latency")) with_latency = true;
        if (!strcmp(argv[i], "--sdst")) sdst_mode = true;    // only the carry-out-register experiment
        if (!strcmp(argv[i], "--clock")) clock_mode = true;  // short run for `rocprofv3 --pmc GRBM_GUI_ACTIVE`: few, long launches
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate %d kHz (nominal)\n", prop.gcnArchName, cus, prop.clockRate);
    unsigned* sink;
    CHECK(hipMalloc(&sink, 256 * 8));
    std::vector<Entry> tests = {
        {"v_mad_i64_i32 (sgpr)", k_mad_i64_i32_s, 8, false}, {"v_mad_i64_i32 (vgpr)", k_mad_i64_i32_v, 8, false},
        {"v_mad_u64_u32 (vgpr)", k_mad_u64_u32_v, 8, false}, {"v_lshl_add_u64", k_lshl_add_u64, 8, false},
        {"v_ashrrev_i64", k_ashrrev_i64, 8, false},           {"v_fma_f64", k_fma_f64, 8, false},
        {"v_mul_lo_u32", k_mul_lo_u32, 8, false},             {"v_mul_hi_u32", k_mul_hi_u32, 8, false},
        {"v_mad_u32_u24", k_mad_u32_u24, 8, false},           {"v_mad_i32_i24", k_mad_i32_i24, 8, false},
        {"v_mul_i32_i24", k_mul_i32_i24, 8, false},           {"v_alignbit_b32", k_alignbit_b32, 8, false},
        {"v_add3_u32", k_add3_u32, 8, false},                 {"v_lshl_add_u32", k_lshl_add_u32, 8, false},
        {"v_and_or_b32", k_and_or_b32, 8, false},             {"v_bfe_i32", k_bfe_i32, 8, false},
        {"v_bfe_u32", k_bfe_u32, 8, false},                   {"v_add_co_u32", k_add_co_u32, 8, false},
        {"v_add_u32", k_add_u32, 8, false},                   {"v_and_b32", k_and_b32, 8, false},
        {"v_xor_b32", k_xor_b32, 8, false},                   {"v_lshlrev_b32", k_lshlrev_b32, 8, false},
        {"v_lshrrev_b32", k_lshrrev_b32, 8, false},           {"v_ashrrev_i32", k_ashrrev_i32, 8, false},
        {"v_mov_b32", k_mov_b32, 8, false},                   {"v_mov_b32 dpp quad_perm", k_mov_dpp_quad, 8, false},
        {"v_fma_f32", k_fma_f32, 8, false},
        {"mix: tight step (and+8mad+ashr64+add64)", k_mix_step_tight, 11, false},
        {"mix: wide step (xor+9mad)", k_mix_step_wide, 10, false},
    };
    if (sdst_mode) {
        tests = {{"v_mad_i64_i32 sdst = vcc", k_mad_i64_i32_s, 8, false},
                 {"v_mad_i64_i32 sdst = one SGPR pair", k_mad_sdst_one, 8, false},
                 {"v_mad_i64_i32 sdst rotating over 2 pairs", k_mad_sdst_rot2, 8, false},
                 {"v_mad_i64_i32 sdst rotating over 8 pairs", k_mad_sdst_rot8, 8, false},
                 {"mix: wide step (xor+9mad), sdst = vcc", k_mix_step_wide, 10, false},
                 {"mix: wide step (xor+9mad), sdst rotating", k_mix_step_wide_rot, 10, false},
                 {"mix: mad / v_and alternating (8 per block)", k_mad_and_interleaved, 8, false},
                 {"v_lshl_add_u64", k_lshl_add_u64, 8, false}};
    }
    if (with_latency) {
        tests.push_back({"lone wave, 1 chain: v_mad_i64_i32", k_lat_mad_i64_i32, 8, true});
        tests.push_back({"lone wave, 1 chain: v_and_b32", k_lat_and_b32, 8, true});
        tests.push_back({"lone wave, 1 chain: v_lshl_add_u64", k_lat_lshl_add_u64, 8, true});
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int niter = 4000;  // x REP x 8 = 128 k instructions per wave per launch (~0.25 ms at 4 waves/SIMD)
    auto ramp = [&](double ms_target) {  // continuous multiply-add issue on the whole chip
        float ms = 0;
        CHECK(hipEventRecord(e0));
        int launches = 0;
        do {
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_mad_i64_i32_s, dim3(cus * 4), dim3(256), 0, 0, sink, 12345679u, niter);
            launches += 20;
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        } while (ms < ms_target);
        return launches;
    };
    ramp(300.0);
    if (clock_mode) {
        // 4 streams x 12 launches of ~2.5 ms at 4 waves/SIMD, back to back after the ramp: GRBM_GUI_ACTIVE / 8 / duration of
        // each dispatch in the profiler's CSV is the shader clock this instruction mix runs at
        struct { const char* name; void (*kern)(unsigned*, unsigned, int); } sel[] = {
            {"v_mad_i64_i32 (sgpr)", k_mad_i64_i32_s}, {"v_and_b32", k_and_b32},
            {"mix: wide step", k_mix_step_wide}, {"mix: tight step", k_mix_step_tight}};
        for (auto& t : sel) {
            for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(t.kern, dim3(cus * 4), dim3(256), 0, 0, sink, 12345679u, 40000);
            CHECK(hipDeviceSynchronize());
            printf("clock mode: 12 launches of %s\n", t.name);
        }
        return 0;
    }
    const double nominal_simd_cycles = 1024.0 * 2.4e9;
    printf("%-44s %6s %14s %16s %18s\n", "instruction", "w/SIMD", "Gwave-inst/s", "inst/SIMD-cycle", "cycles/inst/SIMD");
    for (auto& t : tests) {
        const bool is_mad = sdst_mode || strstr(t.name, "v_mad_i64_i32 (sgpr)") || strstr(t.name, "mix:");
        for (int k : {1, 2, 3, 4, 6, 8}) {
            if (t.latency && k != 1) continue;
            if (!is_mad && (k == 3 || k == 6 || k == 8)) continue;  // the occupancy sweep only where it matters
            const int grid = cus * k;  // 256 threads = 4 waves = one wave per SIMD per block
            ramp(40.0);
            // >= 100 ms of back-to-back launches
            float ms = 0;
            long launches = 0;
            CHECK(hipEventRecord(e0));
            do {
                for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(t.kern, dim3(grid), dim3(256), 0, 0, sink, 12345679u, niter);
                launches += 10;
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            } while (ms < 100.0);
            const double inst_per_wave = (double)niter * REP * t.per_block;
            const double total_wave_inst = inst_per_wave * 4.0 * grid * launches;
            const double rate = total_wave_inst / (ms * 1e-3);
            printf("%-44s %6d %14.1f %16.4f %18.3f\n", t.name, k, rate / 1e9, rate / nominal_simd_cycles, nominal_simd_cycles / rate);
        }
    }
    return 0;
}
