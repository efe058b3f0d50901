// Lab: can the FP64 matrix pipe add a chain of doubles IN ORDER?  v_mfma_f64_4x4x4_4b_f64 computes D = A B + C with k = 4; with B = 1 every
// product is exact, so D = C + a0 + a1 + a2 + a3 -- and if the hardware accumulates the four terms one after the other, each through an
// IEEE fused multiply-add, that is four dependent additions of the running total of Octree.cpp:253-290 in ONE instruction of 4 passes
// (16 cycles), against ~9 cycles an addition on the vector pipe (tools/chain_lab.hip).  This lab finds (1) which lanes feed which output
// (A = 2^lane), (2) whether the result equals the sequential sum in k order, in reverse order, or the exactly rounded sum, on random
// operands of mixed magnitudes, and (3) what a dependent chain of such instructions costs per addition.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/mfma_chain_lab.hip -o tools/_bin/mfma_chain_lab && tools/_bin/mfma_chain_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void layout(double* out) {
    const int l = threadIdx.x;
    const double a = ldexp(1.0, l);
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, 1.0, 0.0, 0, 0, 0);
}
// one instruction per test: lane l holds its own a (tests are laid out on the host with the layout found above) and c
__global__ void sums(const double* __restrict__ a, const double* __restrict__ c, double* __restrict__ out, int n) {
    const int t = blockIdx.x, l = threadIdx.x;
    if (t >= n) return;
    out[t * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t * 64 + l], 1.0, c[t * 64 + l], 0, 0, 0);
}
// a dependent chain: iters instructions, each adding four operands (per output) to the running total
__global__ void chain(const double* __restrict__ a, double* __restrict__ out, unsigned long long* ticks, int iters) {
    const int l = threadIdx.x;
    double tot = 0.0;
    double x = a[l];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += 32) {
#pragma unroll
        for (int u = 0; u < 32; ++u) tot = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, tot, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[l] = tot;
    if (l == 0) ticks[0] = t1 - t0;
}
__global__ void chainAdd(const double* __restrict__ a, double* __restrict__ out, unsigned long long* ticks, int iters) {
    const int l = threadIdx.x;
    double tot = 0.0;
    const double x = a[l];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i += 32) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            tot = tot + x;
            asm volatile("" : "+v"(tot));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[l] = tot;
    if (l == 0) ticks[0] = t1 - t0;
}

static uint64_t rng = 88172645463325252ull;
static uint64_t next() { rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17; return rng; }
static double rnd() {  // mixed magnitudes and signs
    const double m = 1.0 + (double)(next() >> 11) * 0x1.0p-53;
    const int e = (int)(next() % 60) - 30;
    return ((next() & 1) ? -1.0 : 1.0) * ldexp(m, e);
}

int main() {
    double *dOut, *dA, *dC;
    unsigned long long* dT;
    const int n = 4096;
    hipMalloc(&dOut, n * 64 * 8), hipMalloc(&dA, n * 64 * 8), hipMalloc(&dC, n * 64 * 8), hipMalloc(&dT, 8);
    std::vector<double> out(64);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dOut);
    hipMemcpy(out.data(), dOut, 64 * 8, hipMemcpyDeviceToHost);
    int src[64][4];
    std::printf("layout (output lane <- the lanes whose A it sums):\n");
    for (int l = 0; l < 64; ++l) {
        int k = 0;
        double v = out[l];
        for (int s = 63; s >= 0; --s)
            if (v >= ldexp(1.0, s)) {
                v -= ldexp(1.0, s);
                if (k < 4) src[l][3 - k] = s;
                ++k;
            }
        if (l < 20 || l % 16 == 0) std::printf("  lane %2d <- %d %d %d %d%s\n", l, src[l][0], src[l][1], src[l][2], src[l][3], k == 4 ? "" : "  (!! not four terms)");
    }
    // random tests: every output lane of every instruction is one test; the terms in ascending lane order = (a0, a1, a2, a3)
    std::vector<double> a((size_t)n * 64), c((size_t)n * 64), got((size_t)n * 64);
    for (auto& v : a) v = rnd();
    for (auto& v : c) v = rnd();
    // a few adversarial ones: half-ulp terms that a sequential sum drops and an exact sum keeps
    for (int l = 0; l < 64; ++l) a[l] = 0x1.0p-53, c[l] = 1.0;
    // signed zeros, subnormals (operands and totals), values that overflow: instructions 1..63
    {
        const double sp[] = {0.0, -0.0, 4.9406564584124654e-324, -4.9406564584124654e-324, 2.2250738585072009e-308, -2.2250738585072014e-308, 1.0e-310, -3.0e-315,
                             1.7976931348623157e308, -1.7976931348623157e308, 8.9e307, 1.0, -1.0, 409600.0, 1.0e-300, -1.0e-300};
        for (int t = 1; t < 64; ++t)
            for (int l = 0; l < 64; ++l) a[(size_t)t * 64 + l] = sp[next() % 16], c[(size_t)t * 64 + l] = sp[next() % 16];
    }
    hipMemcpy(dA, a.data(), a.size() * 8, hipMemcpyHostToDevice), hipMemcpy(dC, c.data(), c.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sums, dim3(n), dim3(64), 0, 0, dA, dC, dOut, n);
    hipMemcpy(got.data(), dOut, got.size() * 8, hipMemcpyDeviceToHost);
    long fwd = 0, rev = 0, exact = 0, total = 0, special = 0, specialOk = 0;
    for (int t = 0; t < n; ++t)
        for (int l = 0; l < 64; ++l) {
            const double* at = &a[(size_t)t * 64];
            const double x0 = at[src[l][0]], x1 = at[src[l][1]], x2 = at[src[l][2]], x3 = at[src[l][3]], cc = c[(size_t)t * 64 + l];
            volatile double f = cc; f = f + x0; f = f + x1; f = f + x2; f = f + x3;
            volatile double r = cc; r = r + x3; r = r + x2; r = r + x1; r = r + x0;
            const long double e = (long double)cc + x0 + x1 + x2 + x3;  // (80-bit: not exact, an indication only)
            const double g = got[(size_t)t * 64 + l];
            if (t >= 1 && t < 64) special += 1, specialOk += std::memcmp(&g, (const void*)&f, 8) == 0 || (g != g && f != f);
            ++total, fwd += std::memcmp(&g, (const void*)&f, 8) == 0 || (g != g && f != f), rev += std::memcmp(&g, (const void*)&r, 8) == 0, exact += g == (double)e;
        }
    std::printf("%ld sums: equal to c + a0 + a1 + a2 + a3 in lane order %ld, in reverse order %ld, to the (80-bit) sum rounded once %ld\n", total, fwd, rev, exact);
    std::printf("of them %ld with signed zeros, subnormal and near-overflow operands and totals: %ld equal to the sequential sum (NaN = NaN)\n", special, specialOk);
    std::printf("adversarial (c = 1, four terms of 2^-53): got %.17g  (sequential: 1, exact: 1.0000000000000004)\n", got[0]);
    for (int iters : {1024, 4096}) {
        unsigned long long tk = 0;
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dOut, dT, iters);
        hipMemcpy(&tk, dT, 8, hipMemcpyDeviceToHost);
        std::printf("chain of %d matrix instructions (= %d additions per output): %llu ticks = %.2f per instruction, %.2f per addition (x the counter's rate)\n", iters, 4 * iters, tk,
                    (double)tk / iters, (double)tk / iters / 4);
        hipLaunchKernelGGL(chainAdd, dim3(1), dim3(64), 0, 0, dA, dOut, dT, 4 * iters);
        hipMemcpy(&tk, dT, 8, hipMemcpyDeviceToHost);
        std::printf("chain of %d v_add_f64: %llu ticks = %.2f per addition\n", 4 * iters, tk, (double)tk / (4 * iters));
    }
    return 0;
}
