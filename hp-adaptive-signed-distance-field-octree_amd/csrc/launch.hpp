// Host-side launch wrappers of kernels.hip.
#pragma once
#include <algorithm>
#include <vector>
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "device_types.hpp"

namespace hpsdf {

constexpr size_t kFitMaxLdsBytes = 60 * 1024;  // stays under the 64 KiB default dynamic-LDS limit
constexpr int kFitBlockThreads = 256;
constexpr unsigned kQueryMaxGrid = 256 * 8 * 4;  // workgroups of one Query launch (grid-stride beyond)
constexpr size_t kQueryFewPoints = 256;  // up to here Query is one launch of query_few_kernel (one lane per point)

constexpr size_t kFitChunkLdsBytes = 40 * 1024;  // sample planes staged per chunk (keeps >= 3 workgroups per CU)
size_t fitLdsBytes(int degree, int nTasks, int planes);
struct FitShape {
    int cells;           // fits per workgroup
    int cellsPerThread;  // 1, or 4 with cell blocking
    int planes;          // i-planes staged per chunk
    size_t ldsBytes;
};
// latencyBound: the field is a BVH traversal (mesh): one cell per workgroup, occupancy hides the gathers
FitShape fitShape(int degree, int nrows, uint32_t count, bool weighted, bool latencyBound = false);


hipError_t launchFit(hipStream_t stream, int degree, int cellsPerThread, const FitBlock* dBlocks, uint32_t nBlocks, size_t ldsBytes,
                     const FitTask* dTasks, double* dArena, double* dErrs, double* dMirror,  // dMirror: a second destination of the rows (host memory the device can write), or nullptr
                     const DeviceTables* dTables, const FieldDev& field, const RootMap& rm, const uint32_t* dRange = nullptr);
hipError_t launchFitMulti(hipStream_t stream, const FitBlock* dBlocks, uint32_t maxBlocks, size_t ldsBytes, const FitTask* dTasks,
                          double* dArena, double* dErrs, const DeviceTables* dTables, const FieldDev& field, const RootMap& rm,
                          const uint32_t* dCount);
hipError_t launchFitWeight(hipStream_t stream, const FitBlock* dBlocks, uint32_t nBlocks, size_t ldsBytes, const FitTask* dTasks,
                           const double* dArena, double* dMeans, const DeviceTables* dTables, const uint32_t* dCount = nullptr);
// opt-in fast fit of degrees 4..11 on the matrix cores (fit_mfma.hip): blocks of at most 16 fits, NOT bit-identical to launchFit
constexpr int kMfmaCells = 16;
bool fitMfmaSupports(int degree, const FieldDev& field);
hipError_t launchFitMfma(hipStream_t stream, int degree, const FitBlock* dBlocks, uint32_t nBlocks, const FitTask* dTasks, double* dArena,
                         double* dErrs, const DeviceTables* dTables, const FieldDev& field, const RootMap& rm, const uint32_t* dRange = nullptr);
// The default mode's from-scratch fits of degree 6..11 ("split", FitBlock::split; the threshold is the context's splitMinDegree): the rows of top degree by the bit-exact kernel (which
// also leaves the field values in the sample buffer), the rows below it by this launch on the matrix cores -- errors, hence every
// decision of the build, stay those of the all-exact fit; the lower rows agree with it to ~1e-17.
int fitSplitDefaultMinDegree();  // 6 unless HPSDF_SPLIT_MIN_DEGREE says otherwise
bool fitSplitSupports(int degree, int minDegree);
hipError_t launchFitLow(hipStream_t stream, int degree, const FitTask* dTasks, const uint32_t* dRange, uint32_t first, uint32_t count,
                        uint32_t maxTasks, double* dArena, const DeviceTables* dTables, const double* dSamples);  // fit_low.hip
hipError_t launchFitMfmaLow(hipStream_t stream, int degree, const FitTask* dTasks, const uint32_t* dRange, uint32_t first, uint32_t count,
                            uint32_t maxTasks, double* dArena, const DeviceTables* dTables, const double* dSamples, const RootMap& rm, int leftAssoc);
hipError_t launchQuery(hipStream_t stream, const TreeDev& t, const DeviceTables* dTables, const double* dXyz, size_t n,
                       double* dOut, double* dGrad, bool allInline, uint32_t* dDeferCount, uint32_t* dDeferIdx);
hipError_t launchQueryRay(hipStream_t stream, const TreeDev& t, const DeviceTables* dTables, const double* dOrigins,
                          const double* dDirs, const double* dTMax, size_t n, uint8_t* dHit, double* dT);
hipError_t launchSlicePoints(hipStream_t stream, double c, float minX, float minY, float step, uint32_t nSamples,
                             double* dXyz);
hipError_t launchFieldEval(hipStream_t stream, const FieldDev& f, const DeviceTables* dTables, const double* dXyz,
                           size_t n, double* dOut);
// mesh field, linear scan over all triangles (no BVH): Mesh::SignedDistanceAtPt(pt), Mesh.cpp:42-51,134-159
hipError_t launchMeshEvalWave(hipStream_t stream, const FieldDev& f, const double* dXyz, size_t n, double* dOut);
// rocPRIM's radix sort of (key, value) pairs on the low `bits` bits (mesh_build.hip); tmp == nullptr: tmpBytes receives the scratch size
hipError_t sortPairsU32(hipStream_t stream, void* tmp, size_t& tmpBytes, const uint32_t* keys, uint32_t* keysOut, const uint32_t* vals, uint32_t* valsOut,
                        size_t n, unsigned bits);
// the library's private stream-ordered pool of a device (mesh_build.hip; never the application's default pool), and its trim
hipMemPool_t meshPool(int dev);
void meshPoolTrim(int dev);

// a few points of a plain mesh field on the calling thread; hm: HOST copies of the field's arrays
void meshEvalHostPoints(const MeshDev& hm, const double* xyz, size_t n, double* out);
// dKeys: n x 8 bytes of DEVICE memory for the per-point (distance, triangle) keys, or nullptr when dOut itself is device memory
hipError_t launchMeshNaive(hipStream_t stream, const FieldDev& f, const double* dXyz, size_t n, double* dOut, unsigned long long* dKeys = nullptr);
hipError_t launchAcosfSelftest(hipStream_t stream, uint32_t first, uint32_t stride, size_t n, float* dOut);
constexpr int kTriRecordFloats = 12;  // MeshDev::triPos: a, b, c, cross(b - a, c - a)
constexpr int kTriPreFloats = 12;     // MeshDev::triPre: g hu | unit normal hv | unit edge vector, triangle index
// dSlotTri: which triangle sits in leaf slot s (nullptr: slot s = triangle s); either output may be nullptr (skipped)
hipError_t launchMeshTriPos(hipStream_t stream, const float* dVerts, const uint32_t* dTris, uint64_t nTris, float* dTriPos,
                            const uint32_t* dSlotTri, float* dTriPre);
// mesh fields: F at every sample of nTasks fits of one degree -> dSamples[FitTask::sampleOff + sample]
hipError_t launchMeshSample(hipStream_t stream, const FitTask* dTasks, uint32_t nTasks, int degree, const DeviceTables* dTables,
                            const FieldDev& field, const RootMap& rm, double* dSamples);
hipError_t launchMeshSampleRange(hipStream_t stream, const FitTask* dTasks, const uint32_t* dRange, uint32_t maxTasks, int degree,
                                 const DeviceTables* dTables, const FieldDev& field, const RootMap& rm, double* dSamples);
// ---- continuity solve on the device (cg.hip); the arithmetic is continuity.cpp's
constexpr uint64_t kCgChunk = 256;  // dot products are summed chunk by chunk (one workgroup's rows), then over the chunks
// The canonical sum of one chunk, e[0..count) with count <= 256 (missing elements count as +0.0): lane l of 64 adds
// e[l], e[l+64], e[l+128], e[l+192] in that order, then the lanes fold as a shuffle tree (l += l+32, l+16, ... l+1).
// Host and device both sum dot products this way, which is what makes their solves bit-identical.
inline double cgChunkSum(const double* e, uint64_t count) {
    double v[64];
    for (uint64_t l = 0; l < 64; ++l) {
        auto at = [&](uint64_t i) { return i < count ? e[i] : 0.0; };
        v[l] = ((at(l) + at(64 + l)) + at(128 + l)) + at(192 + l);
    }
    for (int off = 32; off >= 1; off >>= 1)
        for (int l = 0; l < off; ++l) v[l] = v[l] + v[l + off];
    return v[0];
}
// The canonical total of n chunk sums: the same shape again, level by level -- groups of 256 consecutive values go through
// cgChunkSum until one value is left (at least one level).  A wave does a level of <= 256 values in a dozen instructions;
// added up left to right instead (rounds 1 and 2a) the 325 chunk sums of an 83 k-unknown system cost every kernel that
// needed a scalar 5-6 us of one lane's dependent additions.
inline double cgCombine(const double* part, uint64_t n) {
    if (n == 0) return 0.0;
    std::vector<double> a(part, part + n), b;
    do {
        b.resize((a.size() + 255) / 256);
        for (uint64_t i = 0; i < b.size(); ++i) b[i] = cgChunkSum(a.data() + 256 * i, std::min<uint64_t>(256, a.size() - 256 * i));
        a.swap(b);
    } while (a.size() > 1);
    return a[0];
}
constexpr uint64_t kCgMaxChunksOnDevice = 65536;  // two levels of cgCombine: 16.7 M unknowns
constexpr uint32_t kCgEntriesPerWg = 4096;  // entries of M a workgroup of the SpMV multiplies (its rows may run kCgRowTail past them)
constexpr uint32_t kCgRowTail = 1024;
struct CgScalars {
    double absNew, alpha, beta, resNorm2, threshold, lambda;
    double tol, rhsNorm2, jumpBefore, jumpAfter;
    int32_t it, maxIter, done, pad;  // done: 0 iterating, 1 converged, 2 iteration cap, 3 zero right-hand side
    double absRing[2];               // r . z at the start of iteration k in absRing[k & 1] (read by workgroups of the kernel whose
                                     // first workgroup writes the other slot)
};
struct CgDev {
    uint64_t n, nChunks;
    // M in CSR as the host assembles it: entries of row r at [rowPtr[r], rowPtr[r + 1]), columns ascending
    const uint64_t* rowPtr;  // n + 1
    const uint32_t* col;
    const double* val;
    // the SpMV's work list: workgroup w takes the rows whose first entry lies in [w, w + 1) * kCgEntriesPerWg, i.e. rows
    // wgRow[w] .. wgRow[w + 1] - 1 (filled by launchCgStart).  nWg = 0: a row is longer than kCgRowTail, the row-owned
    // SpMV runs instead.
    uint32_t* wgRow;  // nWg + 1
    uint32_t nWg;
    const double* c;  // the block's coefficients (right-hand side / lambda, initial guess / lambda)
    double *dinv, *rhs;
    double *x, *r, *p, *z, *tmp, *partA, *partB, *partC;
    CgScalars* s;
    // pinned host memory the device writes at the end of a batch of iterations ([0] = CgScalars::done, [1] = the batch's stamp): the host
    // watches it instead of waking up from hipStreamSynchronize (null: no mirror)
    volatile uint32_t* hostFlag;
};
hipError_t launchCgStart(hipStream_t stream, const CgDev& d);   // setup, jump energy before, first residual, threshold
hipError_t launchCgIterations(hipStream_t stream, const CgDev& d, int firstIteration, int iterations, uint32_t stamp = 0);
hipError_t launchCgFinish(hipStream_t stream, const CgDev& d);  // jump energy after
hipError_t launchPack(hipStream_t stream, const PackItem* dItems, uint32_t nItems, const double* dArena, double* dOut);

}  // namespace hpsdf
