// Opt-in fast fit for degrees 4..11 on the matrix cores (hpsdf_ctx_set_fast_fit).
//
// Octree::FitPolynomial (Source/HP/Octree.cpp:1007-1093) is a contraction: with the basis row r = (a, b, c) and the sample
// s = (i, j, k) of the (4p+1)^3 Gauss-Legendre grid,
//     coeff[cell][r] = N_a N_b N_c * sum_s  Fw[cell][s] * V[s][r],      V[s][r] = P_a(x_i) P_b(x_j) P_c(x_k),
//     Fw[cell][s] = |cell| * w_i w_j w_k * F(sample s of the cell)                                        (:1028-1056)
// i.e. a GEMM  C[cells x rows] = Fw[cells x nq^3] * V[nq^3 x rows]  whose A operand is the field itself and whose B
// operand is a tensor product of three small tables.  fit_kernel (kernels.hip) evaluates it term by term in the
// reference's order without fused multiply-adds -- bit-identical to the CPU path, capped at 1/8 of the FP64 peak.  This
// kernel hands it to v_mfma_f64_16x16x4_f64: one workgroup = one tile of 16 cells, its four waves split the sample
// dimension four ways; per step of 4 samples every lane evaluates F at ONE (cell, sample) pair -- which is exactly its
// element of the 16 x 4 A fragment -- and forms its elements of the B fragments (4 x 16 per 16-row tile) from the
// P tables in LDS with two multiplications; the four partial accumulator sets are summed through LDS in wave order
// (a fixed order: the result does not depend on scheduling), normalised, written, and the error of :1062-1069 is
// reduced across the lanes.
//
// NOT bit-identical to the default path (fused multiply-add chains inside the matrix instruction, another summation
// order): coefficients agree to ~1e-15 relative, far inside the 1e-6 the north_star asks for, but refinement decisions
// that are exact ties in the default arithmetic (mirror-symmetric cells) may fall the other way.  Off by default.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "device_types.hpp"
#include "field_eval.hpp"
#include "launch.hpp"

namespace hpsdf {

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int mfmaCoef(int p) { return p == 6 ? 83 : (p + 1) * (p + 2) * (p + 3) / 6; }

// One block's contraction with a compile-time number of 16-row tiles TN: the loop body is one straight run of
// instructions -- B elements of the NEXT step (three LDS reads, two multiplications per tile) and the next A element are
// formed while the matrix pipe works through this step's TN instructions (64 cycles each): the scheduler is told to deal
// them out one MFMA, one tile's worth of LDS reads and VALU work at a time.
// LOW: the rows below the top degree of a split fit (FitBlock::split, device_types.hpp) -- the cells of a tile may differ in depth
// (the normalisation is looked up per cell), and no error is formed: the top-degree rows, which alone enter it, are fit_kernel's.
template <int KIND, int TN, int KW, bool LEFT, bool LOW = false>
__device__ __forceinline__ void mfmaContract(const FitBlock& blk, const FitTask* __restrict__ tasks, double* __restrict__ arena,
                                             double* __restrict__ errs, const DeviceTables* __restrict__ T, const FieldDev& field,
                                             const RootMap& rm, const double* sT, const double* sR, const double* sW, const double* sNl,
                                             double* sRed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int deg = blk.degree, nq = 4 * deg + 1, nq2 = nq * nq, G = blk.nTasks, depth = blk.depth;
    const int rowStart = blk.rowStart, rowEnd = blk.rowEnd;
    // ---- this lane's cell (A rows) and basis rows (B columns)
    const int cell = lane & 15, kq = lane >> 4;
    const bool cellLive = cell < G;
    const FitTask& tk = tasks[blk.firstTask + (cellLive ? cell : 0)];
    double sc[3], ce[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        sc[a] = (double)(tk.bmax[a] - tk.bmin[a]) * 0.5;     // :1020 sizes() in f32
        ce[a] = (double)((tk.bmin[a] + tk.bmax[a]) / 2.0f);  // :1021 center() in f32
    }
    const double S = cellLive ? prod3<LEFT>(sc[0], sc[1], sc[2]) : 0.0;  // :1022 (0: a cell slot beyond the block contributes nothing)
    const uint64_t sampleOff = tk.sampleOff;
    // this lane's row per tile: offsets of its three P rows into sT; rows beyond rowEnd read the zero row appended to sT
    const int zeroRow = (deg + 1) * nq;
    int off[TN];  // three 10-bit offsets per tile
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int r = rowStart + 16 * t + (lane & 15);
        const bool live = r < rowEnd;
        const int oa = live ? (int)T->bidx[r][0] * nq : zeroRow, ob = live ? (int)T->bidx[r][1] * nq : zeroRow;
        const int oc = live ? (int)T->bidx[r][2] * nq : zeroRow;
        off[t] = oa | (ob << 10) | (oc << 20);
    }
    double4_t acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    // ---- this wave's quarter of the (i, j) sample rows.  A step covers samples k = 4 c + kq of ONE row (nq = 4 p + 1: p + 1
    //      steps per row, the last one with a single live sample), so P_a(x_i) P_b(x_j) is the same for all four samples of
    //      a step and changes for the whole wave at once: per step and tile one LDS read and one multiplication remain.
    const int perRow = deg + 1;
    const int row0 = wave * nq2 / KW, row1 = (wave + 1) * nq2 / KW, nSteps = (row1 - row0) * perRow;
    int row = row0, c = 0;  // wave-uniform
    int i = row / nq, j = row - i * nq;
    double u[TN], xi = 0.0, yj = 0.0, wj = 0.0, wi = 0.0;
    auto newRow = [&]() {  // everything that depends on (i, j) only
#pragma unroll
        for (int t = 0; t < TN; ++t) u[t] = sT[(off[t] & 1023) + i] * sT[((off[t] >> 10) & 1023) + j];
        const double ux = sR[i] * sc[0] + ce[0], uy = sR[j] * sc[1] + ce[1];  // :1035-1039
        xi = ux * rm.bounds[0] + rm.centre[0];                               // :327
        yj = uy * rm.bounds[1] + rm.centre[1];
        wi = sW[i], wj = sW[j];
    };
    auto element = [&](double& a, double (&b)[TN]) {  // this lane's A element and B elements of sample (i, j, 4 c + kq)
        const int k = 4 * c + kq;
        const bool live = k < nq;
        const int kk = live ? k : 0;
        const double uz = sR[kk] * sc[2] + ce[2];
        const double wz = uz * rm.bounds[2] + rm.centre[2];
        double fv;
        if constexpr (KIND == kFieldAnalytic)
            fv = analyticEval<LEFT>(field, xi, yj, wz);
        else
            fv = field.samples[sampleOff + (uint64_t)(live && cellLive ? (i * nq + j) * nq + k : 0)];
        a = (live ? S : 0.0) * prod3<LEFT>(wi, wj, sW[kk]) * fv;  // :1040
#pragma unroll
        for (int t = 0; t < TN; ++t) b[t] = u[t] * sT[(off[t] >> 20) + kk];
    };
    double aCur = 0.0, bCur[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) bCur[t] = 0.0;
    if (nSteps > 0) {
        newRow();
        element(aCur, bCur);
    }
    for (int st = 0; st < nSteps; ++st) {
        // the step after this one (its operands are formed while the matrix pipe works)
        ++c;
        if (c == perRow) {
            c = 0, ++row, ++j;
            if (j == nq) j = 0, ++i;
            if (row < row1) newRow();
        }
        double aNext = 0.0, bNext[TN];
        if (st + 1 < nSteps) {
            element(aNext, bNext);
        } else {
#pragma unroll
            for (int t = 0; t < TN; ++t) bNext[t] = 0.0;
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aCur, bCur[t], acc[t], 0, 0, 0);
        // issue order: the next step's LDS reads first, then the matrix instructions with the next step's VALU work dealt
        // out between them (a wave issues in order: VALU work placed behind the last MFMA would wait for all of them)
        __builtin_amdgcn_sched_group_barrier(0x100, TN + 2, 0);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
        }
        aCur = aNext;
#pragma unroll
        for (int t = 0; t < TN; ++t) bCur[t] = bNext[t];
    }
    // ---- the KW partial sums, added in wave order (fixed: the result does not depend on which wave finishes first)
    for (int w = 0; w < KW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double* slot = &sRed[(t * 4 + q) * 64 + lane];
                    if (w == 0)
                        *slot = acc[t][q];
                    else
                        *slot = *slot + acc[t][q];
                }
        }
        __syncthreads();
    }
    // ---- normalise, write, error (:1058-1069).  D[row = (lane >> 4) + 4 q][col = lane & 15] of tile t: cell
    //      (lane >> 4) + 4 q, basis row rowStart + 16 t + (lane & 15).  Wave w finishes tiles w, w + 4, ...
    double e4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t = wave; t < TN; t += KW) {
        const int r = rowStart + 16 * t + (lane & 15);
        if (r < rowEnd) {
            const int ia = T->bidx[r][0], ib = T->bidx[r][1], ic = T->bidx[r][2];
            const double nrm = sNl[ia * 11 + depth] * sNl[ib * 11 + depth] * sNl[ic * 11 + depth];
            const bool top = ia + ib + ic == deg;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c2 = (lane >> 4) + 4 * q;
                if (c2 < G) {
                    const FitTask& tc = tasks[blk.firstTask + c2];
                    double nc = nrm;
                    if constexpr (LOW) {
                        const int dc = tc.depth;
                        nc = sNl[ia * 11 + dc] * sNl[ib * 11 + dc] * sNl[ic * 11 + dc];
                    }
                    const double v = sRed[(t * 4 + q) * 64 + lane] * nc;
                    arena[tc.outOff + (uint64_t)(r - rowStart)] = v;
                    if (top) e4[q] += v * v;
                }
            }
        }
    }
    if constexpr (LOW) return;
    // error: over the 16 lanes of a row group, then over the waves (through LDS, wave order)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int off = 8; off >= 1; off >>= 1) e4[q] += __shfl_xor(e4[q], off, 64);
    __syncthreads();
    double* sErr = sRed;  // [KW waves][16 cells]
    if ((lane & 15) == 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) sErr[wave * 16 + (lane >> 4) + 4 * q] = e4[q];
    __syncthreads();
    if (tid < G) {
        double e = 0.0;
        for (int w = 0; w < KW; ++w) e += sErr[w * 16 + tid];
        errs[tasks[blk.firstTask + tid].errSlot] = e;
    }
}

// DEG: the degree the launch serves.  A block is either a from-scratch fit (all ncoef(DEG) rows) or an incremental one
// (the rows of total degree DEG only, :846-851): two tile counts, both compile-time.
// KW waves per workgroup share a tile's samples: 8 (two per SIMD: one's VALU work runs beside the other's matrix
// instructions) where the accumulators leave room for two waves per SIMD, else 4.
template <int DEG>
struct MfmaWaves {
    static constexpr int value = DEG <= 8 ? 8 : 4;
};
template <int KIND, int DEG, bool LEFT>
__global__ __launch_bounds__(64 * MfmaWaves<DEG>::value) void fit_mfma_kernel(const FitBlock* __restrict__ blocks, const FitTask* __restrict__ tasks,
                                                       double* __restrict__ arena, double* __restrict__ errs,
                                                       const DeviceTables* __restrict__ T, FieldDev field, RootMap rm,
                                                       const uint32_t* __restrict__ range) {
    constexpr int NT = (mfmaCoef(DEG) + 15) / 16, NTI = (mfmaCoef(DEG) - mfmaCoef(DEG - 1) + 15) / 16;
    constexpr int NQ = 4 * DEG + 1, KW = MfmaWaves<DEG>::value;
    __shared__ double sT[(DEG + 2) * NQ];  // P_a(root_q), [DEG + 1][nq] (Octree::LpX, :988-1004), then a row of zeros
    __shared__ double sR[NQ], sW[NQ];
    __shared__ double sNl[13 * 11];
    __shared__ double sRed[NT * 4 * 64];   // partial accumulators, summed wave after wave
    uint32_t bIdx = blockIdx.x;
    if (range != nullptr) {
        if (bIdx >= range[1]) return;
        bIdx += range[0];
    }
    const FitBlock blk = blocks[bIdx];
    const int tid = threadIdx.x;
    constexpr int gl = NQ * (NQ - 1) / 2;  // Legendre.h: rule n starts at n(n-1)/2 (:1016-1017)
    for (int i = tid; i < 13 * 11; i += 64 * KW) sNl[i] = (&T->nl[0][0])[i];
    for (int q = tid; q < NQ; q += 64 * KW) {
        const double x = T->roots[gl + q];
        sR[q] = x;
        sW[q] = T->weights[gl + q];
        double m2 = 0.0, m1 = 1.0;
        sT[q] = 1.0;
        for (int i = 1; i <= DEG; ++i) {
            const double l = T->rec[i][0] * x * m1 - T->rec[i][1] * m2;
            m2 = m1, m1 = l;
            sT[i * NQ + q] = l;
        }
        sT[(DEG + 1) * NQ + q] = 0.0;
    }
    __syncthreads();
    if (blk.rowStart == 0)
        mfmaContract<KIND, NT, KW, LEFT>(blk, tasks, arena, errs, T, field, rm, sT, sR, sW, sNl, sRed);
    else
        mfmaContract<KIND, NTI, KW, LEFT>(blk, tasks, arena, errs, T, field, rm, sT, sR, sW, sNl, sRed);
}

// The rows below the top degree of the split fits of degree DEG (HPSDF_LOW_KERNEL=mfma; degrees >= 4 -- the default lower-rows kernel is fit_low.hip's): tasks
// [range[0], range[0] + range[1]) are from-scratch fits of degree DEG in any mix of depths, 16 to a workgroup; their field values
// are in the sample buffer (fit_kernel wrote them back while it fitted the top-degree rows, or the mesh sampler put them there);
// rows [0, ncoef(DEG - 1)) go to the start of every task's array.  No FitBlock list: a tile is 16 consecutive tasks.
template <int DEG, bool LEFT>
__global__ __launch_bounds__(64 * MfmaWaves<DEG>::value) void fit_mfma_low_kernel(const FitTask* __restrict__ tasks, double* __restrict__ arena,
                                                                                    const DeviceTables* __restrict__ T, FieldDev field, RootMap rm,
                                                                                    const uint32_t* __restrict__ range, uint32_t first, uint32_t count) {
    constexpr int NTL = (mfmaCoef(DEG - 1) + 15) / 16;
    constexpr int NQ = 4 * DEG + 1, KW = MfmaWaves<DEG>::value;
    __shared__ double sT[(DEG + 2) * NQ];
    __shared__ double sR[NQ], sW[NQ];
    __shared__ double sNl[13 * 11];
    __shared__ double sRed[NTL * 4 * 64];
    if (range != nullptr) first = range[0], count = range[1];
    if (blockIdx.x * (uint32_t)kMfmaCells >= count) return;
    FitBlock blk;
    blk.firstTask = first + blockIdx.x * (uint32_t)kMfmaCells;
    blk.nTasks = (uint16_t)(count - blockIdx.x * (uint32_t)kMfmaCells < (uint32_t)kMfmaCells ? count - blockIdx.x * (uint32_t)kMfmaCells : (uint32_t)kMfmaCells);
    blk.degree = DEG, blk.planesPerChunk = 1;
    blk.rowStart = 0, blk.rowEnd = (uint16_t)mfmaCoef(DEG - 1);
    blk.depth = 0, blk.weighted = 0, blk.split = 0, blk.pad1[0] = 0;
    const int tid = threadIdx.x;
    constexpr int gl = NQ * (NQ - 1) / 2;
    for (int i = tid; i < 13 * 11; i += 64 * KW) sNl[i] = (&T->nl[0][0])[i];
    for (int q = tid; q < NQ; q += 64 * KW) {
        const double x = T->roots[gl + q];
        sR[q] = x;
        sW[q] = T->weights[gl + q];
        double m2 = 0.0, m1 = 1.0;
        sT[q] = 1.0;
        for (int i = 1; i <= DEG; ++i) {
            const double l = T->rec[i][0] * x * m1 - T->rec[i][1] * m2;
            m2 = m1, m1 = l;
            sT[i * NQ + q] = l;
        }
        sT[(DEG + 1) * NQ + q] = 0.0;
    }
    __syncthreads();
    mfmaContract<kFieldSamples, NTL, KW, LEFT, true>(blk, tasks, arena, nullptr, T, field, rm, sT, sR, sW, sNl, sRed);
}

template <int KIND, bool LEFT>
void launchMfmaT(hipStream_t stream, int degree, const FitBlock* dBlocks, uint32_t nBlocks, const FitTask* dTasks, double* dArena, double* dErrs,
                 const DeviceTables* dTables, const FieldDev& field, const RootMap& rm, const uint32_t* dRange) {
#define HPSDF_MFMA_CASE(D)                                                                                                        \
    case D:                                                                                                                       \
        hipLaunchKernelGGL((fit_mfma_kernel<KIND, D, LEFT>), dim3(nBlocks), dim3(64 * MfmaWaves<D>::value), 0, stream, dBlocks, dTasks, dArena, dErrs, dTables, field, \
                           rm, dRange);                                                                                           \
        break;
    switch (degree) {
        HPSDF_MFMA_CASE(2)
        HPSDF_MFMA_CASE(3)
        HPSDF_MFMA_CASE(4)
        HPSDF_MFMA_CASE(5)
        HPSDF_MFMA_CASE(6)
        HPSDF_MFMA_CASE(7)
        HPSDF_MFMA_CASE(8)
        HPSDF_MFMA_CASE(9)
        HPSDF_MFMA_CASE(10)
        HPSDF_MFMA_CASE(11)
        default: break;
    }
#undef HPSDF_MFMA_CASE
}

}  // namespace

bool fitMfmaSupports(int degree, const FieldDev& field) {
    // (degrees 10 and 11 -- 18 and 23 tiles of accumulators -- take 351 and 438 of a lane's 512 registers at one wave per SIMD, no
    // scratch: instantiated since round 3; tests/test_gpu_parity.py compares every degree 2..11 with the bit-exact kernel cell by cell)
    // (degrees 2 and 3 are instantiated for the micro-benchmark -- one and two tiles of 16 rows, 62.5 % full; builds send only
    // degrees >= 4 here: builder.cpp fastDeg, frontier.hip frShape)
    return degree >= 2 && degree <= 11 && field.csgOp < 0 && (field.kind == kFieldAnalytic || field.kind == kFieldSamples);
}

// Blocks of at most 16 fits of one class (degree `degree`, any mix of from-scratch and incremental blocks).
hipError_t launchFitMfma(hipStream_t stream, int degree, const FitBlock* dBlocks, uint32_t nBlocks, const FitTask* dTasks, double* dArena,
                         double* dErrs, const DeviceTables* dTables, const FieldDev& field, const RootMap& rm, const uint32_t* dRange) {
    if (nBlocks == 0) return hipSuccess;
    if (!fitMfmaSupports(degree, field)) return hipErrorInvalidValue;
    if (field.kind == kFieldAnalytic) {
        if (field.leftAssoc) launchMfmaT<kFieldAnalytic, true>(stream, degree, dBlocks, nBlocks, dTasks, dArena, dErrs, dTables, field, rm, dRange);
        else launchMfmaT<kFieldAnalytic, false>(stream, degree, dBlocks, nBlocks, dTasks, dArena, dErrs, dTables, field, rm, dRange);
    } else {
        if (field.leftAssoc) launchMfmaT<kFieldSamples, true>(stream, degree, dBlocks, nBlocks, dTasks, dArena, dErrs, dTables, field, rm, dRange);
        else launchMfmaT<kFieldSamples, false>(stream, degree, dBlocks, nBlocks, dTasks, dArena, dErrs, dTables, field, rm, dRange);
    }
    return hipGetLastError();
}

// From which degree a from-scratch fit is split by default (a context's splitMinDegree starts with this).  The two kernels sample the
// field once (the exact one writes the values back), the exact kernel's per-sample work outside the contraction -- index arithmetic, the
// field, the weights -- is paid for the rows of top degree only (37 % of the rows at degree 5), and the rows below them cost next to
// nothing by sum factorisation (fit_low.hip); what the split adds is the samples' trip through memory, 16 bytes a sample.  Measured
// (16 384 cells, union3 field, ms exact -> split; profiles/r04_low_rows_sum_factorised.txt): degree 2 0.11 -> 0.16, 3: 0.41 -> 0.48,
// 4: 0.99 -> 1.14 (slower); 5: 2.27 -> 2.05, 6: 6.03 -> 4.42, 7: 11.99 -> 7.70, 8: 26.71 -> 14.04.  (With the direct contraction on the
// matrix cores, fit_mfma_low_kernel below, which round 4 started with: 4: 1.23, 5: 2.35, 6: 4.99, 7: 8.82, 8: 15.51.)  The default stays
// at 6 -- trees whose leaves stop at degree 5 keep the canonical bytes in the default mode --; HPSDF_SPLIT_MIN_DEGREE overrides it
// (2..12; 12 = never), hpsdf_ctx_set_split_min_degree a context's.
int fitSplitDefaultMinDegree() {
    static const int v = [] {
        int d = 6;
        if (const char* e = std::getenv("HPSDF_SPLIT_MIN_DEGREE")) d = std::atoi(e);
        return d < 2 ? 2 : (d > 12 ? 12 : d);
    }();
    return v;
}
bool fitSplitSupports(int degree, int minDegree) { return degree >= (minDegree < 2 ? 2 : minDegree) && degree <= 11; }
// Rows [0, ncoef(degree - 1)) of the split fits of `degree` (2..11) from the sample buffer; the tasks are [dRange[0], +dRange[1]) when
// dRange is given (device-written; the grid then covers maxTasks), else [first, first + count).
hipError_t launchFitMfmaLow(hipStream_t stream, int degree, const FitTask* dTasks, const uint32_t* dRange, uint32_t first, uint32_t count,
                            uint32_t maxTasks, double* dArena, const DeviceTables* dTables, const double* dSamples, const RootMap& rm, int leftAssoc) {
    const uint32_t n = dRange ? maxTasks : count;
    if (n == 0) return hipSuccess;
    // The default: the sum-factorised kernel of fit_low.hip (three one-axis contractions on the vector units).  HPSDF_LOW_KERNEL=mfma keeps
    // the direct contraction on the matrix cores below (degrees 4..11), which it replaced in round 4 (profiles/r04_low_rows_sum_factorised.txt).
    const char* lowEnv = std::getenv("HPSDF_LOW_KERNEL");  // (read per launch: a few launches a round; tests switch it)
    const bool direct = lowEnv && lowEnv[0] == 'm';
    if (!direct || degree < 4) return launchFitLow(stream, degree, dTasks, dRange, first, count, maxTasks, dArena, dTables, dSamples);
    if (!fitSplitSupports(degree, 4) || dSamples == nullptr) return hipErrorInvalidValue;
    FieldDev fd;
    std::memset(&fd, 0, sizeof fd);
    fd.kind = kFieldSamples, fd.csgOp = -1, fd.samples = dSamples, fd.leftAssoc = leftAssoc;
    const unsigned grid = (n + (uint32_t)kMfmaCells - 1u) / (uint32_t)kMfmaCells;
#define HPSDF_LOW_CASE(D)                                                                                                                  \
    case D:                                                                                                                                \
        if (fd.leftAssoc)                                                                                                                  \
            hipLaunchKernelGGL((fit_mfma_low_kernel<D, true>), dim3(grid), dim3(64 * MfmaWaves<D>::value), 0, stream, dTasks, dArena, dTables, fd, rm, \
                               dRange, first, count);                                                                                      \
        else                                                                                                                               \
            hipLaunchKernelGGL((fit_mfma_low_kernel<D, false>), dim3(grid), dim3(64 * MfmaWaves<D>::value), 0, stream, dTasks, dArena, dTables, fd, rm, \
                               dRange, first, count);                                                                                      \
        break;
    switch (degree) {
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
