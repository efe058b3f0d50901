// The rows below the top degree of a split fit (FitBlock::split, device_types.hpp), by sum factorisation.
//
// A split fit's rows of top degree -- the only ones that enter its error (Octree.cpp:1062-1069), so the only ones a decision of Create
// ever sees -- come from the bit-exact kernel, which also leaves the field's values F[i][j][k] of the (4p+1)^3 Gauss-Legendre grid in the
// sample buffer.  The rows below them are promised to ~1e-16 of the coefficients' scale, not to the bit, so they may be computed any way
// that is accurate.  The direct way (fit_mfma_low_kernel: every sample against every row on the matrix cores) spends nq^3 * rows
// multiply-adds and as many VALU multiplies again to form the B operand P_a(x_i) P_b(y_j) P_c(z_k) per sample and row.  But the grid is
// a tensor grid and the basis a product basis, so
//     c[a][b][c] = S N_a N_b N_c  sum_k A_c[k]  sum_j A_b[j]  sum_i A_a[i] F[i][j][k],        A_a[q] = w_q P_a(x_q)
// is three small contractions, one axis at a time, keeping only a + b + c <= p - 1:
//     G1[a][j][k] = sum_i A_a[i] F[i][j][k]          p * nq^3 multiply-adds        (the only stage that reads the samples)
//     G2[a][b][k] = sum_j A_b[j] G1[a][j][k]         ~p^2/2 * nq^2
//     c[a][b][c]  = sum_k A_c[k] G2[a][b][k]         ~p^3/6 * nq
// -- 5 x fewer operations than the direct form at degree 4, 13 x at degree 8, no operand to form, and every one of them a fused
// multiply-add.  On this part the FP64 vector rate equals the FP64 matrix rate (78.6 TFLOP/s either way), and stage 1's output tile is
// only p <= 11 rows tall (a 16 x 16 x 4 matrix instruction would run 25-70 % full), so the stages run on the vector units: one workgroup
// per cell, a thread per few (j, k) columns in stage 1 (coalesced reads of the sample planes, four planes' loads in flight), per
// (a, b, k) in stage 2, per row in stage 3.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_types.hpp"
#include "launch.hpp"

namespace hpsdf {

namespace {

__host__ __device__ constexpr int lowCoef(int p) { return p == 6 ? 83 : (p + 1) * (p + 2) * (p + 3) / 6; }  // Utility.h:63-77 (count[6] == 83)

constexpr int kLowThreads = 256;
constexpr int kLowAhead = 4;  // sample planes whose loads a lane has in flight in stage 1

template <int DEG>
struct LowShape {
    static constexpr int NQ = 4 * DEG + 1, NQ2 = NQ * NQ;
    // columns of the sample grid a thread carries in stage 1: all of them in ONE pass over the planes (column tid + 256 c: a wave's
    // loads are 512 contiguous bytes).  Four columns whatever the degree (round 4) left degree 8 a second pass with 17 of 256 lanes at
    // work and degree 6 39 % of its lanes idle.
    static constexpr int COLS = (NQ2 + kLowThreads - 1) / kLowThreads;
    // G2[a][b][k] for a + b <= DEG - 1 only: row a holds DEG - a entries
    static constexpr int PAIRS = DEG * (DEG + 1) / 2;
    // basis indices of one pass of stage 1: as many as keep the workgroup within 52 KB of LDS (three workgroups a CU: two, with 80 KB,
    // left a SIMD two waves in front of its global loads), in passes of equal size (7 + 1 at degree 8 read every sample twice for one row)
    static constexpr int kBudget = 52 * 1024 - (DEG * NQ + PAIRS * NQ) * 8;
    static constexpr int AGmax = kBudget / (NQ2 * 8) < 1 ? 1 : kBudget / (NQ2 * 8);
    static constexpr int PASSES = (DEG + AGmax - 1) / AGmax;
    static constexpr int AG = (DEG + PASSES - 1) / PASSES;
};
__host__ __device__ constexpr int lowPair(int deg, int a, int b) { return a * deg - a * (a - 1) / 2 + b; }

template <int DEG>
__global__ __launch_bounds__(kLowThreads, 3) void fit_low_kernel(const FitTask* __restrict__ tasks, double* __restrict__ arena,
                                                              const DeviceTables* __restrict__ T, const double* __restrict__ samples,
                                                              const uint32_t* __restrict__ range, uint32_t first, uint32_t count) {
    constexpr int NQ = LowShape<DEG>::NQ, NQ2 = LowShape<DEG>::NQ2, AG = LowShape<DEG>::AG, COLS = LowShape<DEG>::COLS;
    constexpr int NROWS = lowCoef(DEG - 1);
    __shared__ double sA[DEG * NQ];                      // A_a[q] = w_q P_a(x_q), a < DEG (Octree::LpX, :988-1004)
    __shared__ double sG1[AG * NQ2];                     // this pass's G1[a][j][k]
    __shared__ double sG2[LowShape<DEG>::PAIRS * NQ];    // G2[a][b][k], a + b <= DEG - 1
    if (range != nullptr) first = range[0], count = range[1];
    if (blockIdx.x >= count) return;
    const FitTask& tk = tasks[first + blockIdx.x];
    const int tid = threadIdx.x;
    constexpr int gl = NQ * (NQ - 1) / 2;  // Legendre.h: rule n starts at n(n-1)/2 (:1016-1017)
    for (int q = tid; q < NQ; q += kLowThreads) {
        const double x = T->roots[gl + q], w = T->weights[gl + q];
        double m2 = 0.0, m1 = 1.0;
        sA[q] = w;
        for (int a = 1; a < DEG; ++a) {
            const double l = T->rec[a][0] * x * m1 - T->rec[a][1] * m2;
            m2 = m1, m1 = l;
            sA[a * NQ + q] = w * l;
        }
    }
    __syncthreads();
    const double* F = samples + tk.sampleOff;
    for (int a0 = 0; a0 < DEG; a0 += AG) {
        const int na = DEG - a0 < AG ? DEG - a0 : AG;
        // ---- stage 1: G1[a][col] = sum_i A_a[i] F[i][col], col = j * NQ + k; this lane's columns are tid, tid + 256, ...
        {
            double acc[AG][COLS];
#pragma unroll
            for (int a = 0; a < AG; ++a)
#pragma unroll
                for (int c = 0; c < COLS; ++c) acc[a][c] = 0.0;
#pragma unroll 1
            for (int i0 = 0; i0 < NQ; i0 += kLowAhead) {
                double f[kLowAhead][COLS];
#pragma unroll
                for (int u = 0; u < kLowAhead; ++u)
#pragma unroll
                    for (int c = 0; c < COLS; ++c) {
                        const int col = tid + c * kLowThreads;
                        f[u][c] = (i0 + u < NQ && col < NQ2) ? F[(size_t)(i0 + u) * NQ2 + col] : 0.0;
                    }
#pragma unroll
                for (int u = 0; u < kLowAhead; ++u)
#pragma unroll
                    for (int a = 0; a < AG; ++a) {
                        const double t = (a < na && i0 + u < NQ) ? sA[(a0 + a) * NQ + i0 + u] : 0.0;
#pragma unroll
                        for (int c = 0; c < COLS; ++c) acc[a][c] = __builtin_fma(t, f[u][c], acc[a][c]);
                    }
            }
#pragma unroll
            for (int a = 0; a < AG; ++a)
#pragma unroll
                for (int c = 0; c < COLS; ++c) {
                    const int col = tid + c * kLowThreads;
                    if (a < na && col < NQ2) sG1[a * NQ2 + col] = acc[a][c];
                }
        }
        __syncthreads();
        // ---- stage 2: G2[a][b][k] = sum_j A_b[j] G1[a][j][k] for b <= DEG - 1 - a
        for (int o = tid; o < na * DEG * NQ; o += kLowThreads) {
            const int al = o / (DEG * NQ), rem = o - al * (DEG * NQ), b = rem / NQ, k = rem - b * NQ;
            const int a = a0 + al;
            if (a + b > DEG - 1) continue;
            const double* g = sG1 + al * NQ2 + k;
            const double* tb = sA + b * NQ;
            double s = 0.0;
#pragma unroll 4
            for (int j = 0; j < NQ; ++j) s = __builtin_fma(tb[j], g[j * NQ], s);
            sG2[lowPair(DEG, a, b) * NQ + k] = s;
        }
        __syncthreads();
    }
    // ---- stage 3: c[a][b][c] = S N_a N_b N_c sum_k A_c[k] G2[a][b][k]; the rows in the reference's order (BasisIndexValues)
    double sc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) sc[a] = (double)(tk.bmax[a] - tk.bmin[a]) * 0.5;  // :1020 sizes() in f32
    const double S = sc[0] * (sc[1] * sc[2]);                                      // :1022
    const int depth = tk.depth;
    for (int r = tid; r < NROWS; r += kLowThreads) {
        const int a = T->bidx[r][0], b = T->bidx[r][1], c = T->bidx[r][2];
        const double* g = sG2 + lowPair(DEG, a, b) * NQ;
        const double* tc = sA + c * NQ;
        double s = 0.0;
#pragma unroll 4
        for (int k = 0; k < NQ; ++k) s = __builtin_fma(tc[k], g[k], s);
        arena[tk.outOff + r] = ((S * T->nl[a][depth]) * (T->nl[b][depth] * T->nl[c][depth])) * s;
    }
}

}  // namespace

// Rows [0, ncoef(degree - 1)) of the split fits of `degree` (2..11) from the sample buffer; the tasks are [dRange[0], +dRange[1]) when
// dRange is given (device-written; the grid then covers maxTasks), else [first, first + count).  One workgroup per task.
hipError_t launchFitLow(hipStream_t stream, int degree, const FitTask* dTasks, const uint32_t* dRange, uint32_t first, uint32_t count,
                        uint32_t maxTasks, double* dArena, const DeviceTables* dTables, const double* dSamples) {
    const uint32_t n = dRange ? maxTasks : count;
    if (n == 0) return hipSuccess;
    if (degree < 2 || degree > 11 || dSamples == nullptr) return hipErrorInvalidValue;
#define HPSDF_LOW_CASE(D)                                                                                                               \
    case D:                                                                                                                             \
        hipLaunchKernelGGL((fit_low_kernel<D>), dim3(n), dim3(kLowThreads), 0, stream, dTasks, dArena, dTables, dSamples, dRange, first, \
                           count);                                                                                                      \
        break;
    switch (degree) {
        HPSDF_LOW_CASE(2)
        HPSDF_LOW_CASE(3)
        HPSDF_LOW_CASE(4)
        HPSDF_LOW_CASE(5)
        HPSDF_LOW_CASE(6)
        HPSDF_LOW_CASE(7)
        HPSDF_LOW_CASE(8)
        HPSDF_LOW_CASE(9)
        HPSDF_LOW_CASE(10)
        HPSDF_LOW_CASE(11)
        default: break;
    }
#undef HPSDF_LOW_CASE
    return hipGetLastError();
}

}  // namespace hpsdf
