// Continuity post-process, the solve (Octree::PerformContinuityPostProcess, Octree.cpp:1751-1756) on the device:
// the Jacobi-preconditioned conjugate-gradient loop of continuity.cpp, a kernel per region, with the SAME arithmetic,
// so the block it returns is bit-identical to the host solve's (tests/test_gpu_parity.py):
//   * a matrix row is summed left to right in CSR order (stored here as sliced ELL, 64 rows per slice, so that a wave
//     reads its 64 rows' k-th entries as one contiguous run);
//   * a dot product is summed in the order cgChunkSum (launch.hpp) fixes: chunks of 256 elements -- one workgroup's
//     rows -- each as 64 lane sums of 4 and a shuffle tree, then the chunks left to right.
// The loop's scalars (alpha, beta, |r|^2, the iteration count, the stop flag) live in HBM; the host launches batches
// of iterations and looks at the flag in between; the kernels of iterations past the stop return at once.
#include <hip/hip_runtime.h>

#include "launch.hpp"

namespace hpsdf {

// cgChunkSum of the workgroup's 256 values (one per thread; rows past n hold 0.0); the result is valid in thread 0
__device__ __forceinline__ double blockChunkSum(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        s = ((sh[l] + sh[64 + l]) + sh[128 + l]) + sh[192 + l];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_down(s, off, 64);
    }
    __syncthreads();
    return s;
}

// CSR (as assembled on the host) -> sliced ELL, one thread per row; padding slots get column 0 / value 0
__global__ __launch_bounds__(256) void cg_ell_kernel(uint64_t n, const uint64_t* __restrict__ rowPtr, const uint32_t* __restrict__ csrCol,
                                                     const double* __restrict__ csrVal, const uint64_t* __restrict__ sliceOff,
                                                     uint32_t* __restrict__ rowLen, uint32_t* __restrict__ col, double* __restrict__ val) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // the grid covers whole slices
    const uint64_t slice = i >> 6, base = sliceOff[slice] + (i & 63);
    const uint32_t width = (uint32_t)((sliceOff[slice + 1] - sliceOff[slice]) >> 6);
    const uint64_t first = i < n ? rowPtr[i] : 0;
    const uint32_t len = i < n ? (uint32_t)(rowPtr[i + 1] - first) : 0u;
    if (i < n) rowLen[i] = len;
    for (uint32_t k = 0; k < width; ++k) {
        col[base + (uint64_t)k * 64] = k < len ? csrCol[first + k] : 0u;
        val[base + (uint64_t)k * 64] = k < len ? csrVal[first + k] : 0.0;
    }
}

// region 1 of an iteration: tmp = (M + lambda I) p and the chunk sums of p . tmp
__global__ __launch_bounds__(256) void cg_spmv_kernel(CgDev d) {
    __shared__ double sh[256];
    if (d.s->done) return;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < d.n;
    const uint64_t slice = i >> 6;
    const uint64_t base = d.sliceOff[slice] + (threadIdx.x & 63);
    const uint32_t width = (uint32_t)((d.sliceOff[slice + 1] - d.sliceOff[slice]) >> 6);  // wave-uniform
    const uint32_t len = live ? d.rowLen[i] : 0u;
    const double pi = live ? d.p[i] : 0.0;
    double acc = d.s->lambda * pi;
    // sixteen entries' gathers in flight at a time, the next sixteen columns already on their way (padding slots hold
    // column 0, value 0 and are never added); the additions stay strictly left to right
    constexpr int B = 16;
    uint32_t cn[B];
#pragma unroll
    for (int j = 0; j < B; ++j) cn[j] = d.col[base + (uint64_t)((uint32_t)j < width ? j : (width ? width - 1 : 0)) * 64];
    for (uint32_t k0 = 0; k0 < width; k0 += B) {
        double v[B], pv[B];
#pragma unroll
        for (int j = 0; j < B; ++j) pv[j] = d.p[cn[j]];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const uint32_t k = k0 + j < width ? k0 + j : width - 1;
            v[j] = d.val[base + (uint64_t)k * 64];
            const uint32_t kn = k0 + B + j < width ? k0 + B + j : width - 1;
            cn[j] = d.col[base + (uint64_t)kn * 64];
        }
#pragma unroll
        for (int j = 0; j < B; ++j)
            if (k0 + j < len) acc += v[j] * pv[j];
    }
    if (live) d.tmp[i] = acc;
    const double cs = blockChunkSum(live ? pi * acc : 0.0, sh);
    if (threadIdx.x == 0) d.partA[blockIdx.x] = cs;
}

// the chunk sums left to right, by thread 0 out of LDS (tiles of 1024)
template <bool TWO>
__device__ __forceinline__ void sumChunks(const CgDev& d, double& sumA, double& sumB) {
    __shared__ double ta[1024], tb[1024];
    sumA = sumB = 0.0;
    for (uint64_t c0 = 0; c0 < d.nChunks; c0 += 1024) {
        const uint64_t c = c0 + threadIdx.x;
        ta[threadIdx.x] = c < d.nChunks ? d.partA[c] : 0.0;
        if (TWO) tb[threadIdx.x] = c < d.nChunks ? d.partB[c] : 0.0;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int m = (int)(d.nChunks - c0 < 1024 ? d.nChunks - c0 : 1024);
            for (int j = 0; j < m; ++j) {
                sumA += ta[j];
                if (TWO) sumB += tb[j];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void cg_alpha_kernel(CgDev d) {
    if (d.s->done) return;
    double pAp, unused;
    sumChunks<false>(d, pAp, unused);
    if (threadIdx.x == 0) d.s->alpha = d.s->absNew / pAp;
}

// region 2: x, r, z and the chunk sums of r . r and r . z
__global__ __launch_bounds__(256) void cg_update_kernel(CgDev d) {
    __shared__ double sh[256];
    if (d.s->done) return;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < d.n;
    double rr = 0.0, rz = 0.0;
    if (live) {
        const double alpha = d.s->alpha;
        d.x[i] += alpha * d.p[i];
        const double r = d.r[i] - alpha * d.tmp[i];
        d.r[i] = r;
        const double z = d.dinv[i] * r;
        d.z[i] = z;
        rr = r * r;
        rz = r * z;
    }
    const double a = blockChunkSum(rr, sh);
    const double b = blockChunkSum(rz, sh);
    if (threadIdx.x == 0) d.partA[blockIdx.x] = a, d.partB[blockIdx.x] = b;
}

__global__ __launch_bounds__(1024) void cg_beta_kernel(CgDev d) {
    if (d.s->done) return;
    double rr, rz;
    sumChunks<true>(d, rr, rz);
    if (threadIdx.x == 0) {
        CgScalars& s = *d.s;
        s.resNorm2 = rr;
        if (rr < s.threshold) {
            s.done = 1;
        } else {
            const double absOld = s.absNew;
            s.absNew = rz;
            s.beta = rz / absOld;
            s.it += 1;
            if (s.it >= s.maxIter) s.done = 2;  // the p update of the last iteration feeds nothing
        }
    }
}

// region 3
__global__ __launch_bounds__(256) void cg_direction_kernel(CgDev d) {
    if (d.s->done) return;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n) return;
    d.p[i] = d.z[i] + d.s->beta * d.p[i];
}

hipError_t launchCgLayout(hipStream_t stream, uint64_t n, const uint64_t* dRowPtr, const uint32_t* dCsrCol, const double* dCsrVal,
                          const CgDev& d) {
    if (d.n == 0) return hipSuccess;
    hipLaunchKernelGGL(cg_ell_kernel, dim3((unsigned)d.nChunks), dim3(256), 0, stream, n, dRowPtr, dCsrCol, dCsrVal, d.sliceOff,
                       const_cast<uint32_t*>(d.rowLen), const_cast<uint32_t*>(d.col), const_cast<double*>(d.val));
    return hipGetLastError();
}

hipError_t launchCgIterations(hipStream_t stream, const CgDev& d, int iterations) {
    if (d.n == 0 || iterations <= 0) return hipSuccess;
    if (d.nChunks != (d.n + kCgChunk - 1) / kCgChunk) return hipErrorInvalidValue;
    const dim3 wide((unsigned)d.nChunks), one(1);
    for (int k = 0; k < iterations; ++k) {
        hipLaunchKernelGGL(cg_spmv_kernel, wide, dim3(256), 0, stream, d);
        hipLaunchKernelGGL(cg_alpha_kernel, one, dim3(1024), 0, stream, d);
        hipLaunchKernelGGL(cg_update_kernel, wide, dim3(256), 0, stream, d);
        hipLaunchKernelGGL(cg_beta_kernel, one, dim3(1024), 0, stream, d);
        hipLaunchKernelGGL(cg_direction_kernel, wide, dim3(256), 0, stream, d);
    }
    return hipGetLastError();
}

}  // namespace hpsdf
