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

// region 1 of an iteration: tmp = (M + lambda I) p and the chunk sums of p . tmp.  One workgroup of 1024 threads per
// chunk of 256 rows: four lanes share a row -- lane q of them fetches entries q, q + 4, ... of the current tile of 32 and
// parks the products in LDS -- and the row's own lane (q = 0) then adds them strictly left to right.  The loads of a tile
// are independent of every addition, so they are all in flight at once (a thread per row ran one dependent
// load -> gather -> add chain per entry batch: 30 us per SpMV against 8 for the bytes alone).
constexpr int kCgTile = 32;
// out = (M + shift I) in and the chunk sums of in . out -> part
__device__ __forceinline__ void spmvChunk(const CgDev& d, const double* __restrict__ in, double shift, double* __restrict__ out,
                                          double* __restrict__ part) {
    __shared__ double sProd[4][kCgTile][64];  // [slice of the chunk][entry in tile][row in slice]; reused for the chunk sum
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = wave >> 2, q = wave & 3;
    const uint64_t row = (uint64_t)blockIdx.x * 256 + sl * 64 + lane;
    const bool live = row < d.n;
    const uint64_t slice = (uint64_t)blockIdx.x * 4 + sl;
    const uint64_t base = d.sliceOff[slice] + lane;
    const uint32_t width = (uint32_t)((d.sliceOff[slice + 1] - d.sliceOff[slice]) >> 6);  // wave-uniform
    uint32_t maxWidth = width;  // the workgroup's tile loop must be uniform over its four slices
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const uint64_t so = (uint64_t)blockIdx.x * 4 + o;
        const uint32_t w = (uint32_t)((d.sliceOff[so + 1] - d.sliceOff[so]) >> 6);
        maxWidth = w > maxWidth ? w : maxWidth;
    }
    const uint32_t len = live ? d.rowLen[row] : 0u;
    const double pi = live ? in[row] : 0.0;
    double acc = shift * pi;
    for (uint32_t k0 = 0; k0 < maxWidth; k0 += kCgTile) {
        double v[kCgTile / 4], pv[kCgTile / 4];
#pragma unroll
        for (int m = 0; m < kCgTile / 4; ++m) {
            const uint32_t k = k0 + q + 4 * m;
            const uint64_t idx = base + (uint64_t)(k < width ? k : (width ? width - 1 : 0)) * 64;
            v[m] = width ? d.val[idx] : 0.0;
            pv[m] = width ? in[d.col[idx]] : 0.0;
        }
#pragma unroll
        for (int m = 0; m < kCgTile / 4; ++m) sProd[sl][q + 4 * m][lane] = v[m] * pv[m];
        __syncthreads();
        if (q == 0) {
            const uint32_t end = len < k0 + kCgTile ? len : k0 + kCgTile;
            for (uint32_t k = k0; k < end; ++k) acc += sProd[sl][k - k0][lane];
        }
        __syncthreads();
    }
    if (q == 0 && live) out[row] = acc;
    // cgChunkSum over the chunk's 256 values in . out (rows past n hold 0.0): element e of the chunk = slice e / 64, lane e % 64
    double* sh = &sProd[0][0][0];
    if (q == 0) sh[sl * 64 + lane] = live ? pi * acc : 0.0;
    __syncthreads();
    if (threadIdx.x < 64) {
        double s = ((sh[lane] + sh[64 + lane]) + sh[128 + lane]) + sh[192 + lane];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_down(s, off, 64);
        if (lane == 0) part[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(1024) void cg_spmv_kernel(CgDev d) {  // region 1 of an iteration
    if (d.s->done) return;
    spmvChunk(d, d.p, d.s->lambda, d.tmp, d.partA);
}

// ---- the steps around the loop (continuity.cpp runs them on the host when it has no device): same arithmetic
// which: 0  tmp = M c, chunk sums of c . tmp (jump energy before);  1  tmp = (M + lambda I) x (for the first residual);
//        2  tmp = M x, chunk sums of x . tmp (jump energy after)
__global__ __launch_bounds__(1024) void cg_spmv_aux_kernel(CgDev d, int which) {
    spmvChunk(d, which == 0 ? d.c : d.x, which == 1 ? d.s->lambda : 0.0, d.tmp, d.partA);
}

// rhs = lambda c, x = rhs (solveWithGuess(old, old)), dinv = 1 / (lambda + diagonal entries of the row, in row order)
__global__ __launch_bounds__(256) void cg_setup_kernel(CgDev d) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n) return;
    const double lambda = d.s->lambda;
    const double rhs = d.c[i] * lambda;
    d.rhs[i] = rhs;
    d.x[i] = rhs;
    const uint64_t base = d.sliceOff[i >> 6] + (i & 63);
    double diag = lambda;
    for (uint32_t k = 0, len = d.rowLen[i]; k < len; ++k)
        if (d.col[base + (uint64_t)k * 64] == (uint32_t)i) diag += d.val[base + (uint64_t)k * 64];
    d.dinv[i] = 1.0 / diag;
}

// r = rhs - tmp, p = dinv r, and the chunk sums of rhs . rhs, r . r, r . p
__global__ __launch_bounds__(256) void cg_residual_kernel(CgDev d) {
    __shared__ double sh[256];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < d.n;
    double bb = 0.0, rr = 0.0, rp = 0.0;
    if (live) {
        const double rhs = d.rhs[i], r = rhs - d.tmp[i], p = d.dinv[i] * r;
        d.r[i] = r;
        d.p[i] = p;
        bb = rhs * rhs, rr = r * r, rp = r * p;
    }
    const double a = blockChunkSum(bb, sh);
    const double b = blockChunkSum(rr, sh);
    const double c = blockChunkSum(rp, sh);
    if (threadIdx.x == 0) d.partA[blockIdx.x] = a, d.partB[blockIdx.x] = b, d.partC[blockIdx.x] = c;
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

// which: 0  jumpBefore = sum of partA;  2  jumpAfter = sum of partA;
//        1  |b|^2, |r|^2, r . p from partA / partB / partC -> threshold, absNew, and whether there is anything to iterate
__global__ __launch_bounds__(1024) void cg_scalar_aux_kernel(CgDev d, int which) {
    double a, b;
    sumChunks<true>(d, a, b);
    __shared__ double tc[1024];
    double c = 0.0;
    if (which == 1) {
        for (uint64_t c0 = 0; c0 < d.nChunks; c0 += 1024) {
            const uint64_t ci = c0 + threadIdx.x;
            tc[threadIdx.x] = ci < d.nChunks ? d.partC[ci] : 0.0;
            __syncthreads();
            if (threadIdx.x == 0) {
                const int m = (int)(d.nChunks - c0 < 1024 ? d.nChunks - c0 : 1024);
                for (int j = 0; j < m; ++j) c += tc[j];
            }
            __syncthreads();
        }
    }
    if (threadIdx.x != 0) return;
    CgScalars& s = *d.s;
    if (which == 0) {
        s.jumpBefore = a;
    } else if (which == 2) {
        s.jumpAfter = a;
    } else {
        s.rhsNorm2 = a;
        s.resNorm2 = b;
        s.absNew = c;
        if (a == 0.0) {  // Eigen: a zero right-hand side has the zero solution
            s.resNorm2 = 0.0;
            s.done = 3;
        } else {
            const double t = s.tol * s.tol * a;
            s.threshold = t > 2.2250738585072014e-308 ? t : 2.2250738585072014e-308;  // max(tol^2 |b|^2, DBL_MIN)
            if (b < s.threshold) s.done = 1;
        }
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

// everything before the loop: setup, jump energy of c, first residual, threshold
hipError_t launchCgStart(hipStream_t stream, const CgDev& d) {
    if (d.n == 0) return hipSuccess;
    const dim3 wide((unsigned)d.nChunks), one(1);
    hipLaunchKernelGGL(cg_setup_kernel, wide, dim3(256), 0, stream, d);
    hipLaunchKernelGGL(cg_spmv_aux_kernel, wide, dim3(1024), 0, stream, d, 0);
    hipLaunchKernelGGL(cg_scalar_aux_kernel, one, dim3(1024), 0, stream, d, 0);
    hipLaunchKernelGGL(cg_spmv_aux_kernel, wide, dim3(1024), 0, stream, d, 1);
    hipLaunchKernelGGL(cg_residual_kernel, wide, dim3(256), 0, stream, d);
    hipLaunchKernelGGL(cg_scalar_aux_kernel, one, dim3(1024), 0, stream, d, 1);
    return hipGetLastError();
}

// after the loop: the jump energy of x
hipError_t launchCgFinish(hipStream_t stream, const CgDev& d) {
    if (d.n == 0) return hipSuccess;
    hipLaunchKernelGGL(cg_spmv_aux_kernel, dim3((unsigned)d.nChunks), dim3(1024), 0, stream, d, 2);
    hipLaunchKernelGGL(cg_scalar_aux_kernel, dim3(1), dim3(1024), 0, stream, d, 2);
    return hipGetLastError();
}

hipError_t launchCgIterations(hipStream_t stream, const CgDev& d, int iterations) {
    if (d.n == 0 || iterations <= 0) return hipSuccess;
    if (d.nChunks != (d.n + kCgChunk - 1) / kCgChunk) return hipErrorInvalidValue;
    const dim3 wide((unsigned)d.nChunks), one(1);
    for (int k = 0; k < iterations; ++k) {
        hipLaunchKernelGGL(cg_spmv_kernel, wide, dim3(1024), 0, stream, d);
        hipLaunchKernelGGL(cg_alpha_kernel, one, dim3(1024), 0, stream, d);
        hipLaunchKernelGGL(cg_update_kernel, wide, dim3(256), 0, stream, d);
        hipLaunchKernelGGL(cg_beta_kernel, one, dim3(1024), 0, stream, d);
        hipLaunchKernelGGL(cg_direction_kernel, wide, dim3(256), 0, stream, d);
    }
    return hipGetLastError();
}

}  // namespace hpsdf
