// copy_rate.hip — the HBM yardstick of the one data-movement kernel (csrc/openings.hip), measured on the box the kernel runs on:
// a hand-written 16-byte-per-lane copy kernel (grid-stride, 8 waves per SIMD; MI355X_MICROARCH.md quotes ~6.3 TB/s achievable for it)
// next to hipMemcpy device-to-device, both moving 1 GiB in + 1 GiB out.  Build: hipcc --offload-arch=gfx950 -O3 copy_rate.hip -o copy_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// the same bytes, every lane reading a pseudo-random 128-byte line's 16-byte piece (six lanes per 96 bytes of a line, as the
// extraction's gather does) and storing contiguously: the ceiling of a gather-in / stream-out kernel with no index arithmetic
__global__ void __launch_bounds__(256) k_gather16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, size_t lines) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const size_t rec = t / 6, piece = t - rec * 6;
    const size_t line = (rec * 0x9e3779b97f4a7c15ull >> 20) % lines;
    out[t] = in[line * 8 + piece];
}

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    uint4 *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, auto&& f, double moved) {
        for (int i = 0; i < 5; ++i) f();
        std::vector<float> ms;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0, nullptr);
            for (int i = 0; i < 10; ++i) f();
            (void)hipEventRecord(e1, nullptr);
            (void)hipEventSynchronize(e1);
            float t = 0;
            (void)hipEventElapsedTime(&t, e0, e1);
            ms.push_back(t / 10);
        }
        float best = ms[0], sum = 0;
        for (float v : ms) {
            best = v < best ? v : best;
            sum += v;
        }
        std::printf("%-44s %8.4f ms mean  %8.4f ms best   %6.2f TB/s mean (read + written)\n", name, sum / ms.size(), best, moved / (sum / ms.size() * 1e-3) / 1e12);
    };
    for (unsigned blocks : {2048u, 8192u, 65536u})
        time_it(blocks == 2048 ? "k_copy16, 2,048 blocks (grid-stride)" : blocks == 8192 ? "k_copy16, 8,192 blocks (grid-stride)" : "k_copy16, 65,536 blocks (grid-stride)",
                [&] { hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, nullptr, a, b, n); }, 2.0 * bytes);
    time_it("k_copy16, one 16-byte word per lane", [&] { hipLaunchKernelGGL(k_copy16, dim3((unsigned)(n / 256)), dim3(256), 0, nullptr, a, b, n); }, 2.0 * bytes);
    time_it("hipMemcpy device to device", [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, nullptr); }, 2.0 * bytes);
    // gather: 96 of every 128-byte line fetched are used -> bytes fetched = 4/3 x bytes written; reported on fetched lines + written bytes
    time_it("k_gather16 (random lines, 6 lanes per line)", [&] { hipLaunchKernelGGL(k_gather16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, a, b, n, bytes / 128); },
            bytes * (1.0 + 4.0 / 3.0));
    CK(hipDeviceSynchronize());
    return 0;
}
