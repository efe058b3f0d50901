// Continuity post-process, the solve (Octree::PerformContinuityPostProcess, Octree.cpp:1751-1756) on the device:
// the Jacobi-preconditioned conjugate-gradient loop of continuity.cpp, a kernel per region, with the SAME arithmetic,
// so the block it returns is bit-identical to the host solve's (tests/test_gpu_parity.py):
//   * a matrix row is summed left to right in CSR order (read from the CSR arrays as they are: 20 MB for the 83 k-unknown
//     benchmark system, which stays in the L2s between iterations; the sliced-ELL copy of the first versions was 35 MB --
//     three quarters padding, neighbouring rows differ a lot in length -- and streamed from HBM every iteration);
//   * a dot product is summed in the order cgChunkSum (launch.hpp) fixes: chunks of 256 elements -- one workgroup's
//     rows -- each as 64 lane sums of 4 and a shuffle tree, then the chunk sums the same way again (cgCombine).
// The loop's scalars (|r|^2, r . z, the iteration count, the stop flag) live in HBM; the host launches batches of
// iterations and looks at the flag in between; the kernels of iterations past the stop return at once.
// An iteration is THREE kernels (five in the first version: 42 us per iteration of which 21 were launch gaps):
//   step 1 (k): every workgroup sums the chunk sums of |r|^2 and r . z that step 2 (k - 1) left -- the same fixed order,
//               so all of them get the same beta and the same verdict on the stop rule -- and then runs
//               tmp = (M + lambda I) p with the new direction p = z + beta p_old formed on the fly, entry by entry,
//               from z and the previous direction (two buffers, alternating; the products are the same bits).  The work
//               is cut by ENTRIES of M, not rows (spmvEntries), so the chunk sums of p . tmp are formed by
//   dot:        cg_dot_kernel, a workgroup per chunk of 256 rows;
//   step 2 (k): every workgroup sums the chunk sums of p . tmp, alpha = (r . z) / (p . tmp), then x, r, z and the new
//               chunk sums.
// Only workgroup 0 writes scalars, and only into slots no workgroup of the same kernel reads.
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

// tmp = (M + shift I) in and the chunk sums of in . tmp.  One workgroup of 1024 threads per chunk of 256 rows, FOUR ADJACENT
// LANES PER ROW: of every run of 32 entries lane q fetches entries 8 q .. 8 q + 7 (all loads of the run in flight at once)
// and forms the products; the row's sum then runs strictly left to right in all four lanes at once -- step k adds the
// product held by lane k / 8, handed round inside the quad by DPP (no LDS, no barrier; the additions are the only
// dependent chain).  A wave owns 16 rows and never waits for another.  (First version: a thread per row, 30 us per SpMV;
// second: four waves per 64 rows of a sliced-ELL copy, products parked in LDS, a barrier per 32 entries: 21-25 us.)
// FUSED: in = z + beta * pOld, formed per entry (and written to pNew for the rows of this chunk); otherwise in is read
// as it is.  sh: 256 doubles of LDS.
constexpr int kCgBatch = 8;  // entries per lane in flight: 32 per row
template <int J>
__device__ __forceinline__ double quadBroadcast(double v) {  // the value lane J of every quad holds, in all four lanes
    constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);  // quad_perm: [J, J, J, J]
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, ctrl, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), ctrl, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
template <int J>
__device__ __forceinline__ double quadRun(double acc, const double (&prod)[kCgBatch], uint32_t k0, uint32_t len) {
#pragma unroll
    for (int j = 0; j < kCgBatch; ++j) {  // entries k0 + 8 J + j of the row, in order; past the row's length nothing is added
        const double t = acc + quadBroadcast<J>(prod[j]);
        acc = k0 + (uint32_t)(kCgBatch * J + j) < len ? t : acc;
    }
    return acc;
}
template <bool FUSED>
__device__ __forceinline__ void spmvChunk(const CgDev& d, const double* __restrict__ in, const double* __restrict__ pOld, double beta,
                                          double* __restrict__ pNew, double shift, double* __restrict__ out, double* __restrict__ part,
                                          double* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane & 3;
    const int rowInChunk = wave * 16 + (lane >> 2);
    const uint64_t row = (uint64_t)blockIdx.x * 256 + rowInChunk;
    const bool live = row < d.n;
    const uint64_t first = live ? d.rowPtr[row] : 0;
    const uint32_t len = live ? (uint32_t)(d.rowPtr[row + 1] - first) : 0u;
    auto value = [&](uint64_t j) { return FUSED ? in[j] + beta * pOld[j] : in[j]; };
    const double pi = live ? value(row) : 0.0;
    if (FUSED && q == 0 && live) pNew[row] = pi;
    double acc = shift * pi;
    for (uint32_t k0 = 0; __any(k0 < len); k0 += 4 * kCgBatch) {
        double prod[kCgBatch], v[kCgBatch];
        uint32_t col[kCgBatch];
#pragma unroll
        for (int j = 0; j < kCgBatch; ++j) {
            const uint32_t k = k0 + (uint32_t)(kCgBatch * q + j);
            const bool on = k < len;
            v[j] = on ? d.val[first + k] : 0.0;
            col[j] = on ? d.col[first + k] : (uint32_t)(live ? row : 0);
        }
#pragma unroll
        for (int j = 0; j < kCgBatch; ++j) prod[j] = v[j] * value(col[j]);
        acc = quadRun<0>(acc, prod, k0, len);
        acc = quadRun<1>(acc, prod, k0, len);
        acc = quadRun<2>(acc, prod, k0, len);
        acc = quadRun<3>(acc, prod, k0, len);
    }
    if (q == 0 && live) out[row] = acc;
    // cgChunkSum over the chunk's 256 values in . out (rows past n hold 0.0)
    if (q == 0) sh[rowInChunk] = live ? pi * acc : 0.0;
    __syncthreads();
    if (threadIdx.x < 64) {
        double s = ((sh[lane] + sh[64 + lane]) + sh[128 + lane]) + sh[192 + lane];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_down(s, off, 64);
        if (lane == 0) part[blockIdx.x] = s;
    }
}

// The same product with the work cut by ENTRIES instead of rows (what the loop runs; the row-owned version above stays for
// matrices with a row longer than kCgRowTail).  Row lengths run from 1 to ~100 here and the row-owned kernel lasts as long
// as its longest rows: every run of 32 entries is two more dependent fetches (entries, then the gathered vector) for that
// wave, 21-30 us per launch when the bytes alone are 3.  Here workgroup w multiplies entries [w, w + 1) * 4096 (plus the
// tail of its last row): every thread fetches its four or five entries at once and gathers -- two fetches deep whatever
// the rows look like -- parks the products in LDS, and the rows whose first entry lies in that range are then summed from
// LDS, a thread per row, strictly left to right: the same sums, bit for bit.  The chunk sums of in . out need all 256 rows
// of a chunk, which no longer sit in one workgroup: cg_dot_kernel forms them afterwards.
constexpr int kCgPer = (int)((kCgEntriesPerWg + kCgRowTail) / 1024);  // entries a thread of the entry-cut SpMV fetches
struct SpmvFetched {  // a thread's entries and the vector's values at their columns: everything the products need but beta
    double v[kCgPer], a[kCgPer], b[kCgPer];
    uint32_t rs, re;
};
// the fetches of the entry-cut SpMV: they do not depend on beta, so step 1 asks for them BEFORE it sums the chunk sums that give beta
// (two dependent trips to memory that used to follow the reduction's two)
template <bool FUSED>
__device__ __forceinline__ SpmvFetched spmvFetch(const CgDev& d, const double* __restrict__ in, const double* __restrict__ pOld, uint32_t w) {
    const uint32_t tid = threadIdx.x;
    const uint64_t e0 = (uint64_t)w * kCgEntriesPerWg, nnz = d.rowPtr[d.n];
    SpmvFetched f;
    uint32_t c[kCgPer];
#pragma unroll
    for (int k = 0; k < kCgPer; ++k) {
        const uint64_t e = e0 + tid + (uint64_t)k * 1024;
        const bool on = e < nnz;
        f.v[k] = on ? d.val[e] : 0.0;
        c[k] = on ? d.col[e] : 0u;
    }
    f.rs = d.wgRow[w], f.re = d.wgRow[w + 1];
#pragma unroll
    for (int k = 0; k < kCgPer; ++k) {
        f.a[k] = in[c[k]];
        f.b[k] = FUSED ? pOld[c[k]] : 0.0;
    }
    return f;
}
template <bool FUSED>
__device__ __forceinline__ void spmvEntries(const CgDev& d, const SpmvFetched& f, const double* __restrict__ in, const double* __restrict__ pOld, double beta,
                                            double* __restrict__ pNew, double shift, double* __restrict__ out, double* sProd /* kCgEntriesPerWg + kCgRowTail */, uint32_t w) {
    const uint32_t tid = threadIdx.x;
    const uint64_t e0 = (uint64_t)w * kCgEntriesPerWg;
    auto value = [&](uint64_t j) { return FUSED ? in[j] + beta * pOld[j] : in[j]; };
#pragma unroll
    for (int k = 0; k < kCgPer; ++k) sProd[tid + k * 1024] = f.v[k] * (FUSED ? f.a[k] + beta * f.b[k] : f.a[k]);
    __syncthreads();
    for (uint32_t r = f.rs + tid; r < f.re; r += 1024u) {
        const uint32_t a = (uint32_t)(d.rowPtr[r] - e0), b = (uint32_t)(d.rowPtr[r + 1] - e0);
        const double pi = value(r);
        if (FUSED) pNew[r] = pi;
        double acc = shift * pi;
        for (uint32_t k = a; k < b; ++k) acc += sProd[k];
        out[r] = acc;
    }
}

// partA[chunk] = cgChunkSum over the chunk's rows of a . b
__global__ __launch_bounds__(256) void cg_dot_kernel(CgDev d, const double* __restrict__ a, const double* __restrict__ b, int checkDone) {
    __shared__ double sh[256];
    if (checkDone && d.s->done) return;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const double s = blockChunkSum(i < d.n ? a[i] * b[i] : 0.0, sh);
    if (threadIdx.x == 0) d.partA[blockIdx.x] = s;
}

// workgroup w of the entry-cut SpMV owns the rows whose first entry lies in [w, w + 1) * kCgEntriesPerWg
__global__ __launch_bounds__(256) void cg_rows_kernel(CgDev d) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w <= d.nWg) {
        uint32_t lo = 0, hi = (uint32_t)d.n;  // first row whose first entry is >= w * kCgEntriesPerWg (n if none)
        const uint64_t key = (uint64_t)w * kCgEntriesPerWg;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (d.rowPtr[mid] < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        d.wgRow[w] = w == d.nWg ? (uint32_t)d.n : lo;
    }
}

// cgChunkSum (launch.hpp) of e[0 .. cnt), cnt <= 256, by one wave; the result is valid in lane 0
__device__ __forceinline__ double waveChunkSum(const double* e, uint64_t cnt, int lane) {
    auto at = [&](uint64_t i) { return i < cnt ? e[i] : 0.0; };
    double s = ((at((uint64_t)lane) + at(64 + (uint64_t)lane)) + at(128 + (uint64_t)lane)) + at(192 + (uint64_t)lane);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_down(s, off, 64);
    return s;
}

// cgCombine (launch.hpp) of the chunk sums of one or two dot products, with the totals handed to every thread of the
// workgroup: the waves share the groups of 256 chunk sums, wave 0 combines the group sums.  nChunks <= kCgMaxChunksOnDevice.
// sh: 514 doubles of LDS.
template <bool TWO>
__device__ __forceinline__ void sumChunksAll(uint64_t nChunks, const double* __restrict__ pa, const double* __restrict__ pb, double* sh,
                                             double& sumA, double& sumB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const uint64_t m = (nChunks + 255) / 256;
    for (uint64_t g = (uint64_t)wave; g < m; g += (uint64_t)nWaves) {
        const uint64_t cnt = nChunks - 256 * g < 256 ? nChunks - 256 * g : 256;
        const double a = waveChunkSum(pa + 256 * g, cnt, lane);
        if (lane == 0) sh[g] = a;
        if (TWO) {
            const double b = waveChunkSum(pb + 256 * g, cnt, lane);
            if (lane == 0) sh[256 + g] = b;
        }
    }
    __syncthreads();
    if (m > 1) {
        if (wave == 0) {
            const double a = waveChunkSum(sh, m, lane);
            const double b = TWO ? waveChunkSum(sh + 256, m, lane) : 0.0;
            if (lane == 0) sh[512] = a, sh[513] = b;
        }
        __syncthreads();
        sumA = sh[512], sumB = TWO ? sh[513] : 0.0;
    } else {
        sumA = sh[0], sumB = TWO ? sh[256] : 0.0;
    }
    __syncthreads();
}

// What ends iteration k - 1 and opens iteration k (the reference's order: |r|^2 against the threshold, then beta, then the
// iteration count against the cap).  Every workgroup computes it for itself; `writer` also records it.  Returns false
// when the loop is over.
__device__ __forceinline__ bool cgOpenIteration(const CgDev& d, int k, double* tile, bool writer, double& beta) {
    double rr, rz;
    sumChunksAll<true>(d.nChunks, d.partB, d.partC, tile, rr, rz);
    CgScalars& s = *d.s;
    const bool converged = rr < s.threshold;
    const bool capped = !converged && k >= s.maxIter;
    beta = converged ? 0.0 : rz / s.absRing[(k - 1) & 1];
    if (writer && threadIdx.x == 0) {
        s.resNorm2 = rr;
        if (converged) {
            s.done = 1;
        } else {
            s.absRing[k & 1] = rz;
            s.absNew = rz;
            s.beta = beta;
            s.it = k;
            if (capped) s.done = 2;  // the direction of an iteration past the cap feeds nothing
        }
    }
    return !(converged || capped);
}

// step 1 of iteration k
__global__ __launch_bounds__(1024, 8) void cg_step1_kernel(CgDev d, int k) {
    __shared__ double sh[514];
    if (d.s->done) return;
    double* const P[2] = {d.p, d.rhs};  // the direction of iteration k lives in P[k & 1] (rhs is free once the loop runs)
    if (k == 0) {
        spmvChunk<false>(d, d.p, nullptr, 0.0, nullptr, d.s->lambda, d.tmp, d.partA, sh);
        return;
    }
    double beta;
    if (!cgOpenIteration(d, k, sh, blockIdx.x == 0, beta)) return;
    spmvChunk<true>(d, d.z, P[(k - 1) & 1], beta, P[k & 1], d.s->lambda, d.tmp, d.partA, sh);
}

// step 1 of iteration k with the SpMV cut by entries (grid = d.nWg); cg_dot_kernel follows
__global__ __launch_bounds__(1024, 8) void cg_step1e_kernel(CgDev d, int k) {
    __shared__ double sProd[kCgEntriesPerWg + kCgRowTail];
    if (d.s->done) return;
    double* const P[2] = {d.p, d.rhs};
    if (k == 0) {
        spmvEntries<false>(d, spmvFetch<false>(d, d.p, nullptr, blockIdx.x), d.p, nullptr, 0.0, nullptr, d.s->lambda, d.tmp, sProd, blockIdx.x);
        return;
    }
    const SpmvFetched f = spmvFetch<true>(d, d.z, P[(k - 1) & 1], blockIdx.x);  // (in flight while the chunk sums are added up)
    double beta;
    if (!cgOpenIteration(d, k, sProd, blockIdx.x == 0, beta)) return;
    spmvEntries<true>(d, f, d.z, P[(k - 1) & 1], beta, P[k & 1], d.s->lambda, d.tmp, sProd, blockIdx.x);
}
__global__ __launch_bounds__(1024, 8) void cg_spmv_aux_e_kernel(CgDev d, int which) {
    __shared__ double sProd[kCgEntriesPerWg + kCgRowTail];
    const double* in = which == 0 ? d.c : d.x;
    spmvEntries<false>(d, spmvFetch<false>(d, in, nullptr, blockIdx.x), in, nullptr, 0.0, nullptr, which == 1 ? d.s->lambda : 0.0, d.tmp, sProd, blockIdx.x);
}

// after the last iteration of a batch: has the loop ended?  (one workgroup; the next batch's first kernel would find out
// the same, one launch and one host round trip later)
__global__ __launch_bounds__(64) void cg_check_kernel(CgDev d, int k, uint32_t stamp) {
    __shared__ double tile[514];
    CgScalars& s = *d.s;
    if (!s.done) {  // (uniform)
        double rr, rz;
        sumChunksAll<true>(d.nChunks, d.partB, d.partC, tile, rr, rz);
        if (threadIdx.x == 0) {
            if (rr < s.threshold) {
                s.resNorm2 = rr;
                s.done = 1;
            } else if (k >= s.maxIter) {
                s.resNorm2 = rr;
                s.absNew = rz;
                s.beta = rz / s.absRing[(k - 1) & 1];
                s.it = k;
                s.done = 2;
            }
        }
    }
    if (threadIdx.x == 0 && d.hostFlag != nullptr) {  // tell the host (which watches these words) that the batch is over, and how
        d.hostFlag[0] = (uint32_t)s.done;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        d.hostFlag[1] = stamp;
    }
}

// ---- the steps around the loop (continuity.cpp runs them on the host when it has no device): same arithmetic
// which: 0  tmp = M c, chunk sums of c . tmp (jump energy before);  1  tmp = (M + lambda I) x (for the first residual);
//        2  tmp = M x, chunk sums of x . tmp (jump energy after)
__global__ __launch_bounds__(1024, 8) void cg_spmv_aux_kernel(CgDev d, int which) {
    __shared__ double sh[256];
    spmvChunk<false>(d, which == 0 ? d.c : d.x, nullptr, 0.0, nullptr, which == 1 ? d.s->lambda : 0.0, d.tmp, d.partA, sh);
}

// rhs = lambda c, x = rhs (solveWithGuess(old, old)), dinv = 1 / (lambda + diagonal entries of the row, in row order)
__global__ __launch_bounds__(256) void cg_setup_kernel(CgDev d) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n) return;
    const double lambda = d.s->lambda;
    const double rhs = d.c[i] * lambda;
    d.rhs[i] = rhs;
    d.x[i] = rhs;
    double diag = lambda;
    for (uint64_t k = d.rowPtr[i], end = d.rowPtr[i + 1]; k < end; ++k)
        if (d.col[k] == (uint32_t)i) diag += d.val[k];
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

// which: 0  jumpBefore = sum of partA;  2  jumpAfter = sum of partA;
//        1  |b|^2, |r|^2, r . p from partA / partB / partC -> threshold, absNew, and whether there is anything to iterate
__global__ __launch_bounds__(256) void cg_scalar_aux_kernel(CgDev d, int which) {
    __shared__ double sh[514];
    double a, b, c = 0.0, unused;
    sumChunksAll<true>(d.nChunks, d.partA, d.partB, sh, a, b);
    if (which == 1) sumChunksAll<false>(d.nChunks, d.partC, nullptr, sh, c, unused);
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
        s.absRing[0] = c;
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

// step 2 of iteration k: alpha from the chunk sums of p . tmp, then x, r, z and the chunk sums of r . r and r . z
__global__ __launch_bounds__(256) void cg_step2_kernel(CgDev d, int k) {
    __shared__ double sh[514];
    if (d.s->done) return;
    const double* p = (k & 1) ? d.rhs : d.p;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < d.n;
    // (this row's operands, asked for before the chunk sums are added up: they do not depend on alpha)
    double pi = 0.0, xi = 0.0, ri = 0.0, ti = 0.0, di = 0.0;
    if (live) pi = p[i], xi = d.x[i], ri = d.r[i], ti = d.tmp[i], di = d.dinv[i];
    double pAp, unused;
    sumChunksAll<false>(d.nChunks, d.partA, nullptr, sh, pAp, unused);
    const double alpha = d.s->absRing[k & 1] / pAp;
    double rr = 0.0, rz = 0.0;
    if (live) {
        d.x[i] = xi + alpha * pi;
        const double r = ri - alpha * ti;
        d.r[i] = r;
        const double z = di * r;
        d.z[i] = z;
        rr = r * r;
        rz = r * z;
    }
    const double a = blockChunkSum(rr, sh);
    const double b = blockChunkSum(rz, sh);
    if (threadIdx.x == 0) {
        d.partB[blockIdx.x] = a, d.partC[blockIdx.x] = b;
        if (blockIdx.x == 0) d.s->alpha = alpha;
    }
}

// everything before the loop: setup, jump energy of c, first residual, threshold
hipError_t launchCgStart(hipStream_t stream, const CgDev& d) {
    if (d.n == 0) return hipSuccess;
    const dim3 wide((unsigned)d.nChunks), one(1);
    hipLaunchKernelGGL(cg_setup_kernel, wide, dim3(256), 0, stream, d);
    if (d.nWg) hipLaunchKernelGGL(cg_rows_kernel, dim3((d.nWg + 256u) / 256u), dim3(256), 0, stream, d);
    auto spmvAux = [&](int which, const double* in) {
        if (d.nWg) {
            hipLaunchKernelGGL(cg_spmv_aux_e_kernel, dim3(d.nWg), dim3(1024), 0, stream, d, which);
            hipLaunchKernelGGL(cg_dot_kernel, wide, dim3(256), 0, stream, d, in, (const double*)d.tmp, 0);
        } else {
            hipLaunchKernelGGL(cg_spmv_aux_kernel, wide, dim3(1024), 0, stream, d, which);
        }
    };
    spmvAux(0, d.c);
    hipLaunchKernelGGL(cg_scalar_aux_kernel, one, dim3(256), 0, stream, d, 0);
    spmvAux(1, d.x);
    hipLaunchKernelGGL(cg_residual_kernel, wide, dim3(256), 0, stream, d);
    hipLaunchKernelGGL(cg_scalar_aux_kernel, one, dim3(256), 0, stream, d, 1);
    return hipGetLastError();
}

// after the loop: the jump energy of x
hipError_t launchCgFinish(hipStream_t stream, const CgDev& d) {
    if (d.n == 0) return hipSuccess;
    if (d.nWg) {
        hipLaunchKernelGGL(cg_spmv_aux_e_kernel, dim3(d.nWg), dim3(1024), 0, stream, d, 2);
        hipLaunchKernelGGL(cg_dot_kernel, dim3((unsigned)d.nChunks), dim3(256), 0, stream, d, (const double*)d.x, (const double*)d.tmp, 0);
    } else {
        hipLaunchKernelGGL(cg_spmv_aux_kernel, dim3((unsigned)d.nChunks), dim3(1024), 0, stream, d, 2);
    }
    hipLaunchKernelGGL(cg_scalar_aux_kernel, dim3(1), dim3(256), 0, stream, d, 2);
    return hipGetLastError();
}

hipError_t launchCgIterations(hipStream_t stream, const CgDev& d, int firstIteration, int iterations, uint32_t stamp) {
    if (d.n == 0 || iterations <= 0) return hipSuccess;
    if (d.nChunks != (d.n + kCgChunk - 1) / kCgChunk) return hipErrorInvalidValue;
    const dim3 wide((unsigned)d.nChunks), one(1);
    for (int k = firstIteration; k < firstIteration + iterations; ++k) {
        if (d.nWg) {
            hipLaunchKernelGGL(cg_step1e_kernel, dim3(d.nWg), dim3(1024), 0, stream, d, k);
            hipLaunchKernelGGL(cg_dot_kernel, wide, dim3(256), 0, stream, d, (const double*)((k & 1) ? d.rhs : d.p), (const double*)d.tmp, 1);
        } else {
            hipLaunchKernelGGL(cg_step1_kernel, wide, dim3(1024), 0, stream, d, k);
        }
        hipLaunchKernelGGL(cg_step2_kernel, wide, dim3(256), 0, stream, d, k);
    }
    hipLaunchKernelGGL(cg_check_kernel, one, dim3(64), 0, stream, d, firstIteration + iterations, stamp);
    return hipGetLastError();
}

}  // namespace hpsdf
