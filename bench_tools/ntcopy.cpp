#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static void copy_nt(void* d, const void* s, size_t n) {
    char* dp = (char*)d; const char* sp = (const char*)s;
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        __m256i a = _mm256_loadu_si256((const __m256i*)(sp + i)), b = _mm256_loadu_si256((const __m256i*)(sp + i + 32));
        __m256i c = _mm256_loadu_si256((const __m256i*)(sp + i + 64)), e = _mm256_loadu_si256((const __m256i*)(sp + i + 96));
        _mm256_stream_si256((__m256i*)(dp + i), a); _mm256_stream_si256((__m256i*)(dp + i + 32), b);
        _mm256_stream_si256((__m256i*)(dp + i + 64), c); _mm256_stream_si256((__m256i*)(dp + i + 96), e);
    }
    _mm_sfence();
    if (i < n) memcpy(dp + i, sp + i, n - i);
}
int main(int argc, char** argv) {
    int T = argc > 1 ? atoi(argv[1]) : 8;
    size_t chunk = 16u << 20, total = (size_t)512 << 20;
    char* src = (char*)aligned_alloc(4096, total); char* dst = (char*)aligned_alloc(4096, (size_t)T * chunk);
    memset(src, 1, total); memset(dst, 2, (size_t)T * chunk);
    for (int mode = 0; mode < 2; ++mode) {
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                for (size_t off = (size_t)t * chunk; off + chunk <= total; off += (size_t)T * chunk)
                    mode ? copy_nt(dst + (size_t)t * chunk, src + off, chunk) : (void)memcpy(dst + (size_t)t * chunk, src + off, chunk);
            });
            for (auto& x : th) x.join();
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s < best) best = s;
        }
        printf("%s, %d threads: %.1f GB/s\n", mode ? "nt-store copy" : "memcpy", T, total / best / 1e9);
    }
}
