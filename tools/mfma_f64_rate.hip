// What does the FP64 matrix pipe of this chip sustain?  Back-to-back independent v_mfma_f64_16x16x4_f64 (8 accumulator
// sets per wave), one to eight waves per SIMD, all CUs; and the same for v_fma_f64.  Prints cycles per instruction (s_memtime
// is a 100 MHz counter: wall time is what counts) and TFLOP/s.   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o tools/_bin/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int N>
__global__ void mfma_loop(double* out, int iters) {
    double4_t acc[N];
    for (int t = 0; t < N; ++t) acc[t] = double4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < N; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < N; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void fma_loop(double* out, int iters) {
    double x[16];
    for (int t = 0; t < 16; ++t) x[t] = threadIdx.x * 1e-3 + t;
    const double a = 1.0000001, b = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] = __builtin_fma(x[t], a, b);
    }
    double s = 0;
    for (int t = 0; t < 16; ++t) s += x[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d;
    hipMalloc(&d, sizeof(double) * 256 * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 20000;
    for (int wavesPerSimd : {1, 2, 4}) {
        const int blocks = 256 * wavesPerSimd;  // 256 threads = 4 waves = one per SIMD of a CU
        hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, 0, d, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 8 * blocks * 4;  // MFMA instructions
        std::printf("v_mfma_f64_16x16x4_f64, %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s, %.1f ns per instruction per SIMD\n", wavesPerSimd, ms,
                    n * 2048 / ms / 1e9, ms * 1e6 / ((double)iters * 8 * wavesPerSimd));
    }
    for (int wavesPerSimd : {1, 2, 4}) {
        const int blocks = 256 * wavesPerSimd;
        hipLaunchKernelGGL(fma_loop, dim3(blocks), dim3(256), 0, 0, d, 100);
        hipEventRecord(e0);
        hipLaunchKernelGGL(fma_loop, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 16 * blocks * 4;
        std::printf("v_fma_f64, %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s, %.2f ns per instruction per SIMD\n", wavesPerSimd, ms, n * 128 / ms / 1e9,
                    ms * 1e6 / ((double)iters * 16 * wavesPerSimd));
    }
    return 0;
}
