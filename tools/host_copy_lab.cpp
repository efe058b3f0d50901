// Lab: how fast can one host thread move a block that the GPU has just written into pinned host memory (fr_store_kernel's staging
// buffer) into an ordinary malloc'd block -- the last step of a refined Create.  memcpy against explicit vector copies (streaming
// stores, prefetch), rep movsb, and the same cut over helper threads; staging memory allocated coherent, non-coherent and by default.
//   hipcc -O3 -mavx2 --offload-arch=gfx950 tools/host_copy_lab.cpp -o /tmp/host_copy_lab -lpthread && /tmp/host_copy_lab
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

__global__ void fill(uint64_t* p, size_t n, uint64_t salt) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i * 0x9E3779B97F4A7C15ull + salt;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void copy_memcpy(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
__attribute__((target("avx2"))) static void copy_nt(void* d, const void* s, size_t n) {
    auto* dd = (__m256i*)d; auto* ss = (const __m256i*)s;
    size_t k = n / 32;
    for (size_t i = 0; i < k; i += 4) {
        __m256i a = _mm256_loadu_si256(ss + i), b = _mm256_loadu_si256(ss + i + 1), c = _mm256_loadu_si256(ss + i + 2), e = _mm256_loadu_si256(ss + i + 3);
        _mm256_stream_si256(dd + i, a); _mm256_stream_si256(dd + i + 1, b); _mm256_stream_si256(dd + i + 2, c); _mm256_stream_si256(dd + i + 3, e);
    }
    _mm_sfence();
}
__attribute__((target("avx2"))) static void copy_nt_pf(void* d, const void* s, size_t n) {
    auto* dd = (__m256i*)d; auto* ss = (const __m256i*)s;
    size_t k = n / 32;
    for (size_t i = 0; i < k; i += 4) {
        _mm_prefetch((const char*)(ss + i + 64), _MM_HINT_NTA);
        _mm_prefetch((const char*)(ss + i + 66), _MM_HINT_NTA);
        __m256i a = _mm256_loadu_si256(ss + i), b = _mm256_loadu_si256(ss + i + 1), c = _mm256_loadu_si256(ss + i + 2), e = _mm256_loadu_si256(ss + i + 3);
        _mm256_stream_si256(dd + i, a); _mm256_stream_si256(dd + i + 1, b); _mm256_stream_si256(dd + i + 2, c); _mm256_stream_si256(dd + i + 3, e);
    }
    _mm_sfence();
}
__attribute__((target("avx2"))) static void copy_pf_plain(void* d, const void* s, size_t n) {
    auto* dd = (__m256i*)d; auto* ss = (const __m256i*)s;
    size_t k = n / 32;
    for (size_t i = 0; i < k; i += 4) {
        _mm_prefetch((const char*)(ss + i + 64), _MM_HINT_T0);
        _mm_prefetch((const char*)(ss + i + 66), _MM_HINT_T0);
        __m256i a = _mm256_loadu_si256(ss + i), b = _mm256_loadu_si256(ss + i + 1), c = _mm256_loadu_si256(ss + i + 2), e = _mm256_loadu_si256(ss + i + 3);
        _mm256_storeu_si256(dd + i, a); _mm256_storeu_si256(dd + i + 1, b); _mm256_storeu_si256(dd + i + 2, c); _mm256_storeu_si256(dd + i + 3, e);
    }
}
static void copy_movsb(void* d, const void* s, size_t n) { asm volatile("rep movsb" : "+D"(d), "+S"(s), "+c"(n) : : "memory"); }

struct Helpers {  // helper threads that sleep on a futex-like spin-then-yield flag (lab only: they spin)
    std::vector<std::thread> th;
    std::atomic<int> go{0}, done{0};
    std::atomic<bool> quit{false};
    const char* src = nullptr; char* dst = nullptr; size_t n = 0; int parts = 1;
    void start(int k) {
        for (int t = 1; t <= k; ++t)
            th.emplace_back([this, t] {
                int seen = 0;
                while (!quit.load(std::memory_order_relaxed)) {
                    if (go.load(std::memory_order_acquire) == seen) { _mm_pause(); continue; }
                    seen = go.load(std::memory_order_acquire);
                    if (t < parts) { size_t lo = n * t / parts & ~(size_t)63, hi = t + 1 == parts ? n : (n * (t + 1) / parts & ~(size_t)63); std::memcpy(dst + lo, src + lo, hi - lo); }
                    done.fetch_add(1, std::memory_order_release);
                }
            });
    }
    void run(char* d, const char* s, size_t bytes, int p) {
        src = s, dst = d, n = bytes, parts = p; done.store(0);
        go.fetch_add(1, std::memory_order_release);
        size_t hi = p == 1 ? n : (n / p & ~(size_t)63);
        std::memcpy(d, s, hi);
        while (done.load(std::memory_order_acquire) < (int)th.size()) _mm_pause();
    }
    void stop() { quit = true; go.fetch_add(1); for (auto& t : th) t.join(); }
};

int main() {
    const size_t bytes = 1900000 & ~(size_t)127;  // a 12 000-node tree's block
    hipStream_t st; hipStreamCreate(&st);
    Helpers H; H.start(3);
    struct { const char* name; unsigned flags; } kinds[] = {{"coherent|mapped (what the frontier uses)", hipHostMallocCoherent | hipHostMallocMapped},
                                                            {"non-coherent|mapped", hipHostMallocNonCoherent | hipHostMallocMapped},
                                                            {"default", hipHostMallocDefault}};
    for (auto& kd : kinds) {
        char* pin = nullptr; uint64_t* pinDev = nullptr;
        if (hipHostMalloc((void**)&pin, 4u << 20, kd.flags) != hipSuccess) { std::printf("%s: allocation failed\n", kd.name); continue; }
        hipHostGetDevicePointer((void**)&pinDev, pin, 0);
        char* dst = (char*)std::malloc(bytes + 4096);
        std::memset(dst, 1, bytes + 4096);
        char* dal = (char*)(((uintptr_t)dst + 63) & ~(uintptr_t)63);
        std::printf("staging memory %s, %.2f MB per copy (median / min of 15, us -> GB/s)\n", kd.name, bytes / 1e6);
        struct V { const char* name; int id; } vs[] = {{"memcpy", 0}, {"avx2 loads + streaming stores", 1}, {"... + prefetchnta 2 KB ahead", 2}, {"avx2 plain stores + prefetch", 3},
                                                      {"rep movsb", 4}, {"memcpy on 2 threads", 5}, {"memcpy on 4 threads", 6}, {"memcpy again (source now cached)", 7}};
        for (auto& v : vs) {
            std::vector<double> ts;
            for (int rep = 0; rep < 15; ++rep) {
                if (v.id != 7) { hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, st, pinDev, bytes / 8, (uint64_t)rep * 77 + v.id); hipStreamSynchronize(st); }
                const double t0 = now();
                switch (v.id) {
                    case 0: case 7: copy_memcpy(dal, pin, bytes); break;
                    case 1: copy_nt(dal, pin, bytes); break;
                    case 2: copy_nt_pf(dal, pin, bytes); break;
                    case 3: copy_pf_plain(dal, pin, bytes); break;
                    case 4: copy_movsb(dal, pin, bytes); break;
                    case 5: H.run(dal, pin, bytes, 2); break;
                    case 6: H.run(dal, pin, bytes, 4); break;
                }
                ts.push_back(now() - t0);
                if (std::memcmp(dal, pin, bytes)) { std::printf("  %s: WRONG COPY\n", v.name); return 1; }
            }
            std::sort(ts.begin(), ts.end());
            std::printf("  %-36s %7.1f / %7.1f us   %5.1f GB/s\n", v.name, ts[7], ts[0], bytes / ts[7] / 1e3);
        }
        std::free(dst); hipHostFree(pin);
    }
    H.stop();
    return 0;
}
