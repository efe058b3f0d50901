// Lab (round 6, measured and NOT adopted): divide3 -- a gradient's three components divided by one denominator with the reciprocal formed once, written out
// as the compiler's own expansion of a double division minus the three instructions that only act at the ends of the exponent range --
// against the `/` operator on the device, bit for bit.  Operands: random mantissas and the awkward ones (1.0, all ones, one bit, the
// neighbours of 1.0 and 2.0), exponents over the whole range the fast path accepts (2^-200 .. 2^200) and beyond it, zeros, denormals,
// infinities and NaN (those must take the ordinary division inside divide3 and so agree trivially -- the lab checks that they do).
// Outcome on an MI355X: 1.29e10 quotients, 3.76e9 triples on the shared-reciprocal path, none differs from the `/` operator -- and
// QueryWithGradient on the refined tree 372 -> 373 us, on the headline tree 223.5 -> 219 us: the six divisions' cost in those kernels is
// not their reciprocals (profiles/r06_query_general_floor.txt takes all six and the square root out for -21 %), so kernels.hip keeps `/`.
//   hipcc -O3 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 tools/div3_lab.hip -o /tmp/div3_lab && /tmp/div3_lab [iterations per thread]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__device__ __forceinline__ uint32_t expField(double v) { return ((uint32_t)__double2hiint(v) >> 20) & 0x7FFu; }
__device__ __forceinline__ void divide3(double (&g)[3], double d) {
    const uint32_t e0 = expField(g[0]), e1 = expField(g[1]), e2 = expField(g[2]), ed = expField(d);
    const uint32_t lo = min(min(e0, e1), min(e2, ed)), hi = max(max(e0, e1), max(e2, ed));
    if (lo >= 823u && hi <= 1223u) {
        double r = __builtin_amdgcn_rcp(d);
        double e = __builtin_fma(-d, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-d, r, 1.0);
        r = __builtin_fma(r, e, r);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double q0 = g[k] * r;
            const double res = __builtin_fma(-d, q0, g[k]);
            g[k] = __builtin_fma(res, r, q0);
        }
    } else {
        g[0] = g[0] / d, g[1] = g[1] / d, g[2] = g[2] / d;
    }
}

__device__ __forceinline__ uint64_t mix(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull, z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double operand(uint64_t& s, bool wide) {
    const uint64_t a = mix(s), b = mix(s);
    uint64_t mant = a & 0xFFFFFFFFFFFFFull;
    switch (b & 15u) {  // awkward mantissas now and then
        case 0: mant = 0; break;
        case 1: mant = 0xFFFFFFFFFFFFFull; break;
        case 2: mant = 1ull << (a % 52); break;
        case 3: mant = 0xFFFFFFFFFFFFFull ^ (1ull << (a % 52)); break;
        case 4: mant = a & 0xFFull; break;
        case 5: mant = 0xFFFFFFFFFFFFFull - (a & 0xFFull); break;
        default: break;
    }
    int64_t e;
    if (!wide) {
        e = 1023 + (int64_t)((b >> 8) % 401) - 200;  // the fast path's range
    } else {
        e = (int64_t)((b >> 8) % 2048);  // anything: zeros, denormals, infinities, NaN among them
    }
    return __longlong_as_double((long long)(((b >> 63) << 63) | ((uint64_t)e << 52) | mant));
}

__global__ void lab(uint64_t seed, int iters, unsigned long long* out) {
    uint64_t s = seed ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0xD1B54A32D192ED03ull);
    unsigned long long bad = 0, fastTaken = 0;
    for (int it = 0; it < iters; ++it) {
        const bool wide = (it & 7) == 7;
        double g[3] = {operand(s, wide), operand(s, wide), operand(s, wide)};
        double d = operand(s, wide);
        if ((it & 3) == 1) d = 2.0 * 0.0001;  // the constant denominator of Octree.cpp:956-968
        if ((it & 3) == 2) d = sqrt(fabs(d));  // a square root, as the norm is
        double want[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double num = g[k], den = d;
            asm volatile("" : "+v"(num), "+v"(den));  // (three independent divisions, as the compiler writes each)
            want[k] = num / den;
        }
        const uint32_t e0 = expField(g[0]), e1 = expField(g[1]), e2 = expField(g[2]), ed = expField(d);
        if (min(min(e0, e1), min(e2, ed)) >= 823u && max(max(e0, e1), max(e2, ed)) <= 1223u) ++fastTaken;
        divide3(g, d);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool same = __double_as_longlong(g[k]) == __double_as_longlong(want[k]) || (g[k] != g[k] && want[k] != want[k]);
            bad += same ? 0ull : 1ull;
        }
    }
    atomicAdd(&out[0], bad);
    atomicAdd(&out[1], fastTaken);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2048;
    unsigned long long* d;
    hipMalloc(&d, 16);
    hipMemset(d, 0, 16);
    const int blocks = 16384, threads = 256;
    hipLaunchKernelGGL(lab, dim3(blocks), dim3(threads), 0, 0, 0x1234567ull, iters, d);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 2; }
    unsigned long long h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double triples = (double)blocks * threads * iters;
    std::printf("%.3g triples (%.3g quotients), %.3g of them on the shared-reciprocal path: %llu quotients differ from the `/` operator\n", triples, 3 * triples, (double)h[1], h[0]);
    return h[0] != 0;
}
