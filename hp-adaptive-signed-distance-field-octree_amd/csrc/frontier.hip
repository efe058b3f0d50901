// Device-side frontier of Octree::Create: selection of a round's jobs, task emission, the P/H decision and the tree
// bookkeeping all run on the GPU; the host enqueues a fixed sequence of launches per round and reads one small header.
//
// Reference: the build loop Octree::RunBuildThreadPool (Source/HP/Octree.cpp:194-309), the decision of
// Octree::TickBuildThread (:594-657), Subdivide / CornerAABB (:1096-1128) and ReallocCoeffs (:474-555), under the
// canonical round schedule of DESIGN.md section 3 -- the same schedule csrc/builder.cpp runs on the host (and the oracle on
// the CPU); the MemoryBlock must come out byte-identical to theirs.
//
// State in HBM (FrDev): the serialised node array itself (hpsdf_node, 56 B), per node the queued error (all-ones = not in
// the frontier), the parent index and up to 10 coefficient segments (a leaf of degree p that started at degree f owns rows
// [0, ncoef(f)) from its first fit and one more run of rows per P-refinement: nothing is copied when a degree rises).
//
// A round (r >= 1):
//   fr_select_kernel   top-K of the frontier by (error desc, node index asc) as an MSB radix select over the error's bit
//                      pattern: level 0 (exponent, 2048 bins) against a histogram kept incrementally -- its threshold bin
//                      is known since the previous round closed; everything strictly above the bin is taken, the bin
//                      itself becomes the candidate list
//   fr_batch_kernel    (one workgroup) refines the candidates digit by digit until <= 4096 remain, sorts those exactly,
//                      sorts the taken nodes by index -> the batch; counts the round's cell fits per shape class and
//                      lays out task list, workgroup list, arena rows and sample slots by prefix sums
//   fr_tasks_kernel    (grid) writes the FitTask / FitBlock lists
//   mesh_sample_kernel / fit_kernel   (kernels.hip) over device-written ranges: grids are upper bounds
//   fr_decide_kernel   (one workgroup) improvements (:814-825, :846-854), decision (:600-601), first child of every split
//   fr_update_kernel   (grid) child creation, queue, coefficient counts of the ancestors; workgroup 0 carries the running
//                      total in the reference's order (:253-290); the last workgroup to finish applies the stop rule
//                      (:216) and computes the next selection's threshold bin
//   fr_store_kernel    when the stop rule has fired: ReallocCoeffs -- every leaf's coeffsStart by walking up its
//                      ancestors, coefficients gathered into the packed store
// Round 0 (the 4096 coarse cells) is the same for every build and comes from a template: no selection, no task emission,
// and if the build stops there (the BASELINE thresholds do) the packed store is the arena itself.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "builder.hpp"
#include "frontier.hpp"
#include "launch.hpp"
#include "runtime.hpp"

namespace hpsdf {

namespace {

constexpr uint32_t kFrJobs = 4096;  // largest batch: K <= 4096, and round 0 is the 4096 depth-4 cells
constexpr uint32_t kFrTasks = 9 * kFrJobs;
constexpr int kFrSegs = 10;         // degrees 2..11: the first fit and at most nine P-refinements
constexpr int kFrDepths = kMaxDepth + 2;
constexpr int kFrClasses = 2 * (kMaxDegree + 1) * kFrDepths;  // (degree, from scratch | incremental, depth)
constexpr uint64_t kNotQueued = ~0ull;
constexpr uint64_t kOffMask = (1ull << 56) - 1;  // a segment's arena offset; the bits above hold the rank that owns it
constexpr uint32_t kFrSort = 4096;  // bitonic sort capacity (LDS): the fallback ordering of a batch
constexpr uint32_t kFrExact = 64;   // candidates left when the digit-by-digit refinement hands over to exact ranking
// Multi-rank builds: the last double of a rank's part of errs is its STATUS for the round's exchange.  A healthy rank zeroes it
// (fr_init_kernel / fr_tasks_kernel); a rank whose share of the round failed on the host (device memory) sets it to all ones and
// enters the exchange all the same; fr_round0_kernel / fr_decide_kernel find it and every rank leaves with an error instead of
// waiting in a later collective for a rank that has gone.
constexpr uint32_t kFrStatusPad = 8;

struct FrHdr {
    uint32_t nNodes, nQueued, nJobs, done;
    uint32_t round, maxDegree, maxDepth, overflow;
    uint32_t nTasks, nBlocks, takenCount, candCount;
    uint32_t above;   // selection, level 0: queued nodes in the exponent bins above the threshold bin t1
    int32_t t1;       // (-1: the whole frontier is the batch); both are left by the previous round's update kernel
    uint32_t nLeaves, arrive;
    uint32_t degBlocks[13][2];  // {first block, count} per degree: the range a fit launch walks
    uint32_t degTasks[13][2];   // {first task, count} per degree: the range a mesh-sampler launch walks
    uint32_t lowTasks[13][2];   // {first task, count} per degree: the from-scratch fits that are split (FitBlock::split), i.e. the
                                // range fit_mfma_low_kernel walks for the rows below the top degree
    uint64_t arenaUsed, sampleUsed, nCoeffs, pad1;
    uint64_t jobs, pRefines, hRefines, dropped, fits, samples;
    double total, target;
    // staging between the decide and update kernels of a round
    uint32_t rP, rH, rD, rMaxDeg;
    uint32_t rOps, rPad;
    uint32_t sliceFirst[9];  // multi-rank: rank r computes jobs [sliceFirst[r], sliceFirst[r + 1]) of the round
    uint32_t padS;
    uint64_t packCount[8];   // multi-rank: coefficients rank r contributes to the packed store
    uint64_t dbg[24];  // phase time stamps (s_memtime) of the one-workgroup kernels, read under HPSDF_TRACE
    int64_t rCoeffDelta;
    double rTotal;
    uint32_t hist1[2048];  // queued nodes per exponent bin (kept by update / batch)
    uint32_t hist2[2048];  // level-1 digits of the candidates of the current selection
};
constexpr size_t kFrHdrCopyBytes = offsetof(FrHdr, hist1);
static inline void frCpuRelax() {  // the host's spin on the header mirror
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
}

struct FrRound {  // shape classes of the current round (written by fr_batch_kernel, read by fr_tasks_kernel)
    uint32_t cCount[kFrClasses], cFirst[kFrClasses], cCursor[kFrClasses], cBlockFirst[kFrClasses], cBlocks[kFrClasses];
    uint64_t cArena[kFrClasses], cSample[kFrClasses];
    uint8_t cG[kFrClasses], cPlanes[kFrClasses];
    uint64_t arenaBase;
};

struct FrDev {
    FrHdr* hdr;
    FrHdr* hostHdr;  // the header's mirror in pinned host memory, as the device addresses it
    FrRound* rnd;
    hpsdf_node* nodes;
    uint64_t* qErr;     // bit pattern of the queued error; kNotQueued = not in the frontier
    uint32_t* parent;
    uint64_t* segOff;   // [node][kFrSegs] arena offsets (doubles)
    uint8_t* segFirst;  // degree of the node's first segment
    uint32_t* sub;      // coefficients below every interior node (kept incrementally from round 1 on)
    uint32_t* taken;    // selection: nodes taken so far (unordered)
    uint32_t* candA;
    uint32_t* candB;
    const uint32_t* batchIdx;  // the round's jobs in node-index order (round 0: the template's)
    const double* batchErr;
    const uint64_t* jobP;      // per job: arena offset of the P result / of the first H child
    const uint64_t* jobH;
    uint32_t* wBatchIdx;       // the same buffers, writable (rounds > 0)
    double* wBatchErr;
    uint64_t* wJobP;
    uint64_t* wJobH;
    uint8_t* kind;             // per job: 0 dropped, 1 P, 2 H
    uint32_t* base;            // per H job: index of its first child
    double* ops;               // the round's additions to the running total, densely, in job order
    FitTask* tasks;
    FitBlock* blocks;
    double* errs;       // [jobs][9]
    double* store;      // packed coefficients (ReallocCoeffs)
    const double* arena;
    uint64_t storeCap;
    uint32_t nodeCap, K;
    // multi-rank builds (world > 1): this rank fits a cost-balanced slice of every round's jobs
    int32_t rank, world;
    uint32_t errStride;    // doubles per rank in errs: rank r's slice sits at errs + r * errStride (all-gathered in place)
    uint8_t* jobOwner;     // per job: the rank that fits it
    uint32_t* packPos;     // [node][kFrSegs]: where the segment sits in its owner's pack buffer
    double* pack;          // [world][packStride] all-gathered pack buffers (this rank writes its own)
    uint64_t packStride;
    int32_t fastFit;       // degrees >= 4 fitted by fit_mfma.hip: 16 cells per workgroup
    int32_t splitFit;      // 0, or the lowest degree (.. 11) whose from-scratch fits are split: top-degree rows exact, the rest by fit_mfma_low_kernel
    // nearness weighting (Octree.cpp:1071-1092, 1209-1247): a fit keeps ONE full coefficient array (an incremental fit
    // carries the old rows over, :847), fit_weight_kernel leaves |mean FApprox| of every fit in `means`, the host turns
    // the means into weights with its libm (pow / exp: what the oracle calls) and fr_weigh_kernel scales the errors
    int32_t weighted;
    double* means;          // [jobs][9], pinned host memory as the device addresses it (written by fit_weight_kernel)
    const double* weights;  // [jobs][9], pinned host memory as the device addresses it (written by the host)
    uint32_t* hostFlag;     // pinned: {jobs of the round, stamp}: "the means are there"
};

__host__ __device__ inline uint32_t frCoef(int p) { return p == 6 ? 83u : (uint32_t)((p + 1) * (p + 2) * (p + 3) / 6); }
__host__ __device__ inline int frClass(int degree, bool incr, int depth) { return (2 * degree + (incr ? 1 : 0)) * kFrDepths + depth; }
__host__ __device__ inline size_t frLds(int degree, int g, int planes) {  // = fitLdsBytes (kernels.hip)
    const size_t nq = 4 * (size_t)degree + 1;
    return ((size_t)(degree + 1) * nq + 2 * nq + 8 * (size_t)g + (size_t)g * planes * nq * nq) * sizeof(double);
}
// workgroup shape of `count` fits of one class: what fitShape (kernels.hip) gives an unweighted, sampled-or-analytic fit
// splitFit: 0 = off, else the lowest degree whose from-scratch fits are split (the context's splitMinDegree)
__host__ __device__ inline bool frSplit(int splitFit, int degree, bool incr) { return splitFit > 0 && !incr && degree >= splitFit && degree <= 11; }
__host__ __device__ inline void frShape(int degree, bool incr, uint32_t count, int* cells, int* planes, bool fast = false, bool weighted = false,
                                        int split = 0) {
    if (fast && degree >= 4 && degree <= 11) {  // the matrix-core fit: one workgroup = one tile of 16 cells
        *cells = kMfmaCells, *planes = 1;
        return;
    }
    if (frSplit(split, degree, incr)) incr = true;  // the exact kernel fits the top-degree rows only: the shape of an incremental fit
    const int nrows = incr ? (int)(frCoef(degree) - frCoef(degree - 1)) : (int)frCoef(degree);
    int gmax = nrows > kFitBlockThreads ? 1 : kFitBlockThreads / nrows;
    while (gmax > 1 && frLds(degree, gmax, 1) > kFitMaxLdsBytes) --gmax;
    uint32_t spread = (count + 511u) / 512u;
    int cap = gmax;
    if (degree == 2) {  // as fitShape: measured best for degree 2
        if (count <= 4096u) spread = (count + 1023u) / 1024u;
        cap = gmax < 16 ? gmax : 16;
    }
    int g = (int)(spread < 1u ? 1u : spread);
    g = g < cap ? g : cap;
    const int nq = 4 * degree + 1;
    int pl = nq;
    while (pl > 1 && frLds(degree, g, pl) > kFitChunkLdsBytes) --pl;
    if (weighted) {  // as fitShape: the sample region is reused for the full coefficient array + 100 FApprox values of every cell
        const int need = (int)frCoef(degree) + 100;
        const int minPlanes = (need + nq * nq - 1) / (nq * nq);
        const int want = nq < minPlanes ? nq : minPlanes;
        pl = pl > want ? pl : want;
        while (g > 1 && frLds(degree, g, pl) > kFitMaxLdsBytes) --g;
    }
    *cells = g;
    *planes = pl;
}

// where job j's 9 errors sit: in the slice of the rank that fitted it
__device__ __forceinline__ size_t frErrSlot(const FrDev& d, const FrHdr* h, uint32_t j) {
    if (d.world == 1) return (size_t)j * 9;
    const uint32_t o = d.jobOwner[j];
    return (size_t)o * d.errStride + (size_t)(j - h->sliceFirst[o]) * 9;
}
__device__ __forceinline__ double* frStatusSlot(const FrDev& d, int r) { return d.errs + (size_t)r * d.errStride + (d.errStride - 1u); }
// the lowest rank whose status slot says "failed" (+ 1), 0 if none; threads 0 .. world - 1 look, the result is valid after the caller's barrier
__device__ __forceinline__ void frPeerCheck(const FrDev& d, uint32_t* sPeer) {
    if (d.world > 1 && threadIdx.x < (uint32_t)d.world &&
        (unsigned long long)__double_as_longlong(*(volatile double*)frStatusSlot(d, (int)threadIdx.x)) == ~0ull)
        atomicMax(sPeer, threadIdx.x + 1u);
}
// flop-proportional cost of one job (builder.cpp jobCost): balances the ranks' slices
__device__ __forceinline__ uint64_t frJobCost(int degree, int depth, bool coarse) {
    auto cube = [](uint64_t n) { return n * n * n; };
    if (coarse) return frCoef(2) * cube(9);
    uint64_t c = 0;
    if (depth < kMaxDepth) c += 8ull * frCoef(degree) * cube(4 * (uint64_t)degree + 1);
    if (degree < kMaxDegree - 1) c += (uint64_t)(frCoef(degree + 1) - frCoef(degree)) * cube(4 * (uint64_t)degree + 5);
    return c ? c : 1;
}

// digit `level` of the selection key (error bits, then ~index): larger key = earlier in the frontier's order
__device__ __forceinline__ uint32_t frDigit(int level, uint64_t bits, uint32_t idx) {
    switch (level) {
        case 0: return (uint32_t)(bits >> 52) & 2047u;
        case 1: return (uint32_t)(bits >> 41) & 2047u;
        case 2: return (uint32_t)(bits >> 30) & 2047u;
        case 3: return (uint32_t)(bits >> 19) & 2047u;
        case 4: return (uint32_t)(bits >> 8) & 2047u;
        case 5: return (uint32_t)bits & 255u;
        case 6: return (~idx >> 21) & 2047u;
        case 7: return (~idx >> 10) & 2047u;
        default: return ~idx & 1023u;
    }
}

// Threshold bin of a 2048-bin histogram in LDS for `need` entries taken from the top: the largest bin T with
// count(bins > T) < need <= count(bins >= T); *above = count(bins > T).  All threads call (the histogram must be complete
// and visible: the caller has synchronised); wave 0 works -- lane l owns bins 32 l .. 32 l + 31, a suffix scan over the
// lanes finds the lane where the running count crosses `need`, that lane walks its 32 bins.
__device__ void frThreshold(const uint32_t* hist, uint32_t need, uint32_t* /*sTmp*/, int* outT, uint32_t* outAbove) {
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        uint32_t own = 0;
        for (int k = 0; k < 32; ++k) own += hist[lane * 32 + k];
        uint32_t suf = own;  // inclusive suffix sum: bins of lanes >= this one
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_down(suf, off, 64);
            if (lane + off < 64) suf += v;
        }
        const uint32_t aboveLane = suf - own;  // bins of the lanes above this one
        const bool crossing = aboveLane < need && need <= suf;
        const unsigned long long m = __ballot(crossing);
        if (m == 0ull) {  // fewer than `need` entries in all: everything is taken
            if (lane == 0) *outT = -1, *outAbove = suf;
        } else if (crossing) {
            uint32_t cum = aboveLane;
            int bb = lane * 32 + 31;
            for (; bb > lane * 32 && cum + hist[bb] < need; --bb) cum += hist[bb];
            *outT = bb;
            *outAbove = cum;
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// selection, level 0 (the threshold bin was computed when the previous round's updates completed)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fr_select_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (h->done) return;
    __shared__ uint32_t sHist[2048];
    const uint32_t nNodes = h->nNodes;
    const int T = h->t1;
    for (int i = threadIdx.x; i < 2048; i += 256) sHist[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * 256u;
    bool anyCand = false;
    for (uint32_t base = blockIdx.x * 256u; base < nNodes; base += stride) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t bits = i < nNodes ? d.qErr[i] : kNotQueued;
        const bool queued = bits != kNotQueued;
        const int d0 = (int)frDigit(0, bits, i);
        const bool take = queued && d0 > T, cand = queued && d0 == T;
        const unsigned long long mt = __ballot(take), mc = __ballot(cand);
        if (mt) {
            uint32_t slot = 0;
            const int leader = __ffsll((long long)mt) - 1;
            if (lane == leader) slot = atomicAdd(&h->takenCount, (uint32_t)__popcll(mt));
            slot = __shfl(slot, leader, 64);
            if (take) d.taken[slot + (uint32_t)__popcll(mt & ((1ull << lane) - 1ull))] = i;
        }
        if (mc) {
            uint32_t slot = 0;
            const int leader = __ffsll((long long)mc) - 1;
            if (lane == leader) slot = atomicAdd(&h->candCount, (uint32_t)__popcll(mc));
            slot = __shfl(slot, leader, 64);
            if (cand) {
                d.candA[slot + (uint32_t)__popcll(mc & ((1ull << lane) - 1ull))] = i;
                atomicAdd(&sHist[frDigit(1, bits, i)], 1u);
                anyCand = true;
            }
        }
    }
    if (__syncthreads_or(anyCand ? 1 : 0))
        for (int i = threadIdx.x; i < 2048; i += 256)
            if (sHist[i]) atomicAdd(&h->hist2[i], sHist[i]);
}

// ---------------------------------------------------------------------------------------------------------------------
// selection, remaining levels + exact sort -> the batch; then the round's shape classes
// ---------------------------------------------------------------------------------------------------------------------
// In-place bitonic sort of n (a power of two, <= 4096) (key, val) pairs in LDS by "key descending, then val ascending".
__device__ void frBitonic(uint64_t* key, uint32_t* val, uint32_t n) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < n / 2; t += blockDim.x) {
                const uint32_t lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
                const bool up = ((lo & k) == 0);  // this run sorts "first before second"
                const uint64_t ka = key[lo], kb = key[hi];
                const uint32_t va = val[lo], vb = val[hi];
                const bool aFirst = ka > kb || (ka == kb && va < vb);
                if (aFirst != up && !(ka == kb && va == vb)) {
                    key[lo] = kb, key[hi] = ka;
                    val[lo] = vb, val[hi] = va;
                }
            }
            __syncthreads();
        }
    }
}
__device__ __forceinline__ uint32_t frPow2(uint32_t n) {
    uint32_t p = 2;
    while (p < n) p <<= 1;
    return p;
}

__global__ __launch_bounds__(1024) void fr_batch_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    FrRound* R = d.rnd;
    const uint32_t tid = threadIdx.x;
    if (h->done) {
        if (tid < 13) h->degBlocks[tid][0] = h->degBlocks[tid][1] = h->degTasks[tid][0] = h->degTasks[tid][1] = 0;
        if (tid == 0) h->nJobs = 0, h->nTasks = 0, h->nBlocks = 0;
        return;
    }
    __shared__ uint64_t sKey[kFrSort];
    __shared__ uint32_t sVal[kFrSort];
    __shared__ uint32_t sHist[2048];
    __shared__ uint32_t sTmp[1024];
    __shared__ uint32_t sCount[kFrClasses];
    __shared__ int sT;
    __shared__ uint32_t sAbove, sC, sNext;
#define FR_STAMP(k) do { if (tid == 0) h->dbg[k] = __builtin_readcyclecounter(); } while (0)
    FR_STAMP(0);
    const uint32_t nQ = h->nQueued;
    const uint32_t nJobs = nQ < d.K ? nQ : d.K;
    uint32_t need = nJobs - h->above;  // still to come out of the candidates
    uint32_t C = h->candCount;
    uint32_t* cur = d.candA;
    uint32_t* nxt = d.candB;
    int level = 1;
    if (C > kFrExact) {
        for (uint32_t i = tid; i < 2048; i += 1024) sHist[i] = h->hist2[i];
        __syncthreads();
    }
    uint32_t tk = h->takenCount;
    while (C > kFrExact && level <= 8) {  // refine by the digit of `level` (its histogram is in sHist)
        frThreshold(sHist, need, sTmp, &sT, &sAbove);
        const int T = sT;
        const uint32_t abv = sAbove;
        if (tid == 0) sC = 0, sNext = tk;
        __syncthreads();
        for (uint32_t i = tid; i < 2048; i += 1024) sHist[i] = 0;
        __syncthreads();
        for (uint32_t base = 0; base < C; base += 1024) {
            const uint32_t q = base + tid;
            if (q < C) {
                const uint32_t idx = cur[q];
                const uint64_t bits = d.qErr[idx];
                const int dg = (int)frDigit(level, bits, idx);
                if (dg > T) {
                    d.taken[atomicAdd(&sNext, 1u)] = idx;
                } else if (dg == T) {
                    nxt[atomicAdd(&sC, 1u)] = idx;
                    if (level < 8) atomicAdd(&sHist[frDigit(level + 1, bits, idx)], 1u);
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        need -= abv;
        C = sC;
        tk = sNext;
        __syncthreads();
        uint32_t* t = cur;
        cur = nxt, nxt = t;
        ++level;
    }
    // exact order of what is left (<= 64 candidates, unique keys): a candidate's rank is the number of candidates before
    // it in the frontier's order (error desc, index asc); ranks below `need` join the batch
    FR_STAMP(1);
    if (need && tid < 64) {
        const bool have = tid < C;
        const uint32_t idx = have ? cur[tid] : 0xFFFFFFFFu;
        const uint64_t key = have ? d.qErr[idx] : 0ull;
        uint32_t rank = 0;
        for (int l = 0; l < 64; ++l) {
            const uint64_t ok = (uint64_t)__shfl((long long)key, l, 64);
            const uint32_t oi = __shfl(idx, l, 64);
            rank += ((uint32_t)l < C && (ok > key || (ok == key && oi < idx))) ? 1u : 0u;
        }
        if (have && rank < need) d.taken[tk + rank] = idx;
    }
    __threadfence_block();
    __syncthreads();
    // the batch in node-index order: a bitmap over the nodes (bit = taken), prefix population counts, every set bit's
    // position is its rank (trees beyond 262144 nodes: bitonic sort of the indices)
    FR_STAMP(2);
    const uint32_t nNodes = h->nNodes;
    if (nNodes <= 8192u * 32u) {
        uint32_t* bm = reinterpret_cast<uint32_t*>(sKey);  // 8192 words
        const uint32_t words = (nNodes + 31u) >> 5;
        for (uint32_t w = tid; w < 8192u; w += 1024) bm[w] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < nJobs; i += 1024) {
            const uint32_t idx = d.taken[i];
            atomicOr(&bm[idx >> 5], 1u << (idx & 31u));
        }
        __syncthreads();
        uint32_t own = 0;  // thread t owns words 8 t .. 8 t + 7
        if (tid * 8u < words)
            for (int k = 0; k < 8; ++k) own += (uint32_t)__popc(bm[tid * 8 + k]);
        uint32_t inc = own;  // inclusive scan over the wave's lanes
        const int lane = (int)(tid & 63);
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(inc, off, 64);
            if (lane >= off) inc += v;
        }
        if (lane == 63) sTmp[tid >> 6] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < (tid >> 6); ++w) before += sTmp[w];
        uint32_t pos = before + inc - own;
        if (own)
            for (int k = 0; k < 8; ++k) {
                uint32_t bits = bm[tid * 8 + k];
                while (bits) {
                    const int bpos = __ffs((int)bits) - 1;
                    bits &= bits - 1u;
                    sVal[pos++] = (tid * 8u + (uint32_t)k) * 32u + (uint32_t)bpos;
                }
            }
        __syncthreads();
    } else {
        const uint32_t n2 = frPow2(nJobs);
        for (uint32_t i = tid; i < n2; i += 1024) {
            sKey[i] = 0;
            sVal[i] = i < nJobs ? d.taken[i] : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (nJobs > 1) frBitonic(sKey, sVal, n2);
    }
    FR_STAMP(3);
    for (uint32_t i = tid; i < 2048; i += 1024) sHist[i] = 0;
    for (uint32_t c = tid; c < (uint32_t)kFrClasses; c += 1024) sCount[c] = 0;
    __syncthreads();
    // ---- the jobs leave the frontier; every job becomes 1 (coarse) or up to 9 cell fits: count them per shape class.
    //      Thread t owns jobs 4 t .. 4 t + 3.  With several ranks every rank counts only the fits of its own slice: the
    //      slices are contiguous job ranges of (nearly) equal cost, cut where the host scheduler cuts them (builderSelect).
    uint64_t* sCost = sKey;  // (the bitmap is dead) inclusive prefix of the jobs' costs
    {
        uint32_t jIdx[4];
        int jP[4], jDep[4];
        bool jCoarse[4];
        uint64_t own = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = tid * 4u + (uint32_t)q;
            jIdx[q] = 0xFFFFFFFFu, jP[q] = 0, jDep[q] = 0, jCoarse[q] = false;
            if (j >= nJobs) continue;
            const uint32_t idx = sVal[j];
            const uint64_t bits = d.qErr[idx];
            const double e = __longlong_as_double((long long)bits);
            d.wBatchIdx[j] = idx;
            d.wBatchErr[j] = e;
            d.qErr[idx] = kNotQueued;
            atomicAdd(&sHist[frDigit(0, bits, idx)], 1u);
            const hpsdf_node& n = d.nodes[idx];
            jIdx[q] = idx, jP[q] = n.degree, jDep[q] = n.depth;
            jCoarse[q] = fabs(e - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;  // coarse, Octree.cpp:806,831
            own += frJobCost(jP[q], jDep[q], jCoarse[q]);
        }
        if (d.world > 1) {
            // inclusive scan of the costs (wave shuffles, then the 16 wave totals), slice ends by binary search
            uint64_t inc = own;
            const int lane = (int)(tid & 63);
            for (int off = 1; off < 64; off <<= 1) {
                const uint64_t v = (uint64_t)__shfl_up((long long)inc, off, 64);
                if (lane >= off) inc += v;
            }
            uint64_t* sWave = reinterpret_cast<uint64_t*>(sTmp);  // 16 totals (sTmp has 1024 words)
            if (lane == 63) sWave[tid >> 6] = inc;
            __syncthreads();
            uint64_t before = 0, total = 0;
            for (uint32_t w = 0; w < 16; ++w) {
                if (w < (tid >> 6)) before += sWave[w];
                total += sWave[w];
            }
            uint64_t run = before + inc - own;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t j = tid * 4u + (uint32_t)q;
                if (j >= nJobs) continue;
                run += frJobCost(jP[q], jDep[q], jCoarse[q]);
                sCost[j] = run;  // cost of jobs 0..j
            }
            __syncthreads();
            if (tid < (uint32_t)d.world - 1u) {  // end of rank tid's slice: the first i with cost(jobs < i) >= total (tid + 1) / world
                const uint64_t target = total * (uint64_t)(tid + 1) / (uint64_t)d.world;
                uint32_t lo = 0, hi = nJobs;  // answer in [0, nJobs]; cost(jobs < i) = i ? sCost[i - 1] : 0
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((mid ? sCost[mid - 1] : 0ull) >= target)
                        hi = mid;
                    else
                        lo = mid + 1;
                }
                sTmp[64 + tid] = lo;
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t start = 0;
                h->sliceFirst[0] = 0;
                for (int r = 0; r < d.world; ++r) {
                    uint32_t end = nJobs;
                    if (r + 1 < d.world) {
                        end = sTmp[64 + r];
                        end = end < start ? start : end;
                        end = end > nJobs ? nJobs : end;
                    }
                    h->sliceFirst[r + 1] = end;
                    sTmp[32 + r + 1] = end;
                    start = end;
                }
                sTmp[32] = 0;
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = tid * 4u + (uint32_t)q;
            if (j >= nJobs) continue;
            if (d.world > 1) {
                uint32_t o = 0;
                for (int r = 1; r < d.world; ++r) o += sTmp[32 + r] <= j ? 1u : 0u;
                d.jobOwner[j] = (uint8_t)o;
                if ((int)o != d.rank) continue;  // another rank's job: none of its fits here
            }
            const int p = jP[q], dep = jDep[q];
            if (jCoarse[q]) {
                atomicAdd(&sCount[frClass(2, false, dep)], 1u);  // :836-843
            } else {
                if (dep < kMaxDepth) atomicAdd(&sCount[frClass(p, false, dep + 1)], 8u);     // :814-822
                if (p < kMaxDegree - 1) atomicAdd(&sCount[frClass(p + 1, true, dep)], 1u);  // :846-851
            }
        }
    }
    __syncthreads();
    FR_STAMP(4);
    for (uint32_t i = tid; i < 2048; i += 1024)
        if (sHist[i]) h->hist1[i] -= sHist[i];
    // ---- shapes, then prefix sums over the classes (degree-major): tasks, workgroups, arena rows, samples.
    //      Thread c owns class c; the scan runs over 512 slots in LDS (sKey doubles as the 64-bit scan buffer).
    uint32_t myCount = 0, myBlocks = 0;
    uint64_t myRows = 0, mySamples = 0;
    int g = 1, pl = 1;
    if (tid < (uint32_t)kFrClasses) {
        myCount = sCount[tid];
        if (myCount) {
            const int deg = (int)tid / kFrDepths / 2;
            const bool incr = ((int)tid / kFrDepths) & 1;
            frShape(deg, incr, myCount, &g, &pl, d.fastFit != 0, d.weighted != 0, d.splitFit);
            myBlocks = (myCount + (uint32_t)g - 1u) / (uint32_t)g;
            const uint64_t nq = 4 * (uint64_t)deg + 1;
            // (a weighted fit owns a full array: the incremental one too)
            myRows = (uint64_t)((incr && !d.weighted) ? frCoef(deg) - frCoef(deg - 1) : frCoef(deg)) * myCount;
            mySamples = nq * nq * nq * myCount;
        }
    }
    __syncthreads();
    FR_STAMP(5);
    // four inclusive scans over 512 slots: counts and blocks in sTmp / sVal (32 bit), rows and samples in sKey halves
    uint64_t* sRows = sKey;
    uint64_t* sSmp = sKey + 512;
    uint32_t* sCnt = sTmp;
    uint32_t* sBlk = sVal;
    if (tid < 512) sCnt[tid] = myCount, sBlk[tid] = myBlocks, sRows[tid] = myRows, sSmp[tid] = mySamples;
    __syncthreads();
    for (uint32_t off = 1; off < 512; off <<= 1) {
        uint32_t a = 0, b = 0;
        uint64_t r = 0, sm = 0;
        if (tid < 512 && tid >= off) a = sCnt[tid - off], b = sBlk[tid - off], r = sRows[tid - off], sm = sSmp[tid - off];
        __syncthreads();
        if (tid < 512) sCnt[tid] += a, sBlk[tid] += b, sRows[tid] += r, sSmp[tid] += sm;
        __syncthreads();
    }
    FR_STAMP(6);
    if (tid < (uint32_t)kFrClasses) {
        R->cCount[tid] = myCount;
        R->cFirst[tid] = sCnt[tid] - myCount;
        R->cCursor[tid] = 0;
        R->cBlockFirst[tid] = sBlk[tid] - myBlocks;
        R->cBlocks[tid] = myBlocks;
        R->cArena[tid] = sRows[tid] - myRows;
        R->cSample[tid] = sSmp[tid] - mySamples;
        R->cG[tid] = (uint8_t)g;
        R->cPlanes[tid] = (uint8_t)pl;
    }
    if (tid < 13) {  // per degree: the classes [deg * 24, deg * 24 + 24)
        const uint32_t lo = tid * 2 * kFrDepths, hi = lo + 2 * kFrDepths - 1;
        const uint32_t t0 = lo ? sCnt[lo - 1] : 0u, b0 = lo ? sBlk[lo - 1] : 0u;
        h->degTasks[tid][0] = t0, h->degTasks[tid][1] = sCnt[hi] - t0;
        h->degBlocks[tid][0] = b0, h->degBlocks[tid][1] = sBlk[hi] - b0;
        // the from-scratch classes of a degree come first (frClass): their tasks are one contiguous run
        h->lowTasks[tid][0] = t0;
        h->lowTasks[tid][1] = frSplit(d.splitFit, (int)tid, false) ? sCnt[lo + kFrDepths - 1] - t0 : 0u;
    }
    if (tid == 0) {
        const uint32_t t = sCnt[kFrClasses - 1], b = sBlk[kFrClasses - 1];
        const uint64_t rows = sRows[kFrClasses - 1], smp = sSmp[kFrClasses - 1];
        h->nJobs = nJobs, h->nTasks = t, h->nBlocks = b;
        h->sampleUsed = smp;
        h->fits += t, h->samples += smp;
        R->arenaBase = h->arenaUsed;
        h->arenaUsed += rows;
    }
    FR_STAMP(7);
}

// the round's FitTask / FitBlock lists, grouped by shape class.  Grid-wide: sixteen lanes per job -- lane k < 8 writes the
// from-scratch fit of child k (EstimateHImprovement, :814-822), lane 8 the job's own fit (the coarse degree-2 fit :836-843
// or the incremental one :846-851); lanes 0 and 8 reserve the slots of their class -- then one lane per workgroup record.
__global__ __launch_bounds__(256) void fr_tasks_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (h->done) return;
    FrRound* R = d.rnd;
    const uint32_t nJobs = h->nJobs, nBlocks = h->nBlocks;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
    if (gid == 0 && d.world > 1) *frStatusSlot(d, d.rank) = 0.0;
    const uint64_t arenaBase = R->arenaBase;
    const int lane = threadIdx.x & 63;
    for (uint32_t base = (gid >> 4) - ((uint32_t)lane >> 4); base < nJobs; base += stride >> 4) {  // 4 jobs per wave, wave-uniform trip count
        const uint32_t j = base + ((uint32_t)lane >> 4);
        const int k = lane & 15;
        const bool live = j < nJobs;
        const hpsdf_node& n = d.nodes[d.batchIdx[live ? j : 0]];
        const int p = n.degree, dep = n.depth;
        const bool coarse = fabs(d.batchErr[live ? j : 0] - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
        const bool ours = live && (d.world == 1 || (int)d.jobOwner[j] == d.rank);
        const bool hasH = ours && !coarse && dep < kMaxDepth, hasP = ours && (coarse || p < kMaxDegree - 1);
        // this lane's fit, if any
        const bool mine = (k < 8 && hasH) || (k == 8 && hasP);
        const int deg = k < 8 ? p : (coarse ? 2 : p + 1);
        const bool incr = k == 8 && !coarse;
        const int depth = k < 8 ? dep + 1 : dep;
        const int c = frClass(deg, incr, depth);
        uint32_t slot = 0;
        if (mine && (k == 0 || k == 8)) slot = atomicAdd(&R->cCursor[c], k == 0 ? 8u : 1u);
        slot = __shfl(slot, (lane & ~15) | (k < 8 ? 0 : 8), 64) + (k < 8 ? (uint32_t)k : 0u);
        uint64_t outOff = ~0ull;
        if (mine) {
            const uint64_t rows = (incr && !d.weighted) ? frCoef(deg) - frCoef(deg - 1) : frCoef(deg);
            const uint64_t nq = 4 * (uint64_t)deg + 1;
            FitTask t;
            for (int a = 0; a < 3; ++a) {
                if (k < 8) {  // Octree::CornerAABB, :1096-1112
                    const float mid = (n.aabb_max[a] + n.aabb_min[a]) * 0.5f;
                    t.bmin[a] = (k >> a) & 1 ? mid : n.aabb_min[a];
                    t.bmax[a] = (k >> a) & 1 ? n.aabb_max[a] : mid;
                } else {
                    t.bmin[a] = n.aabb_min[a], t.bmax[a] = n.aabb_max[a];
                }
            }
            outOff = arenaBase + R->cArena[c] + (uint64_t)slot * rows;
            t.outOff = outOff;
            // weighted incremental fit: the cell's current array (one segment: fr_update_kernel keeps it that way), :847
            t.copyOff = (d.weighted && incr) ? (d.segOff[(size_t)d.batchIdx[j] * kFrSegs] & kOffMask) : ~0ull;
            t.sampleOff = R->cSample[c] + (uint64_t)slot * nq * nq * nq;
            t.errSlot = (uint32_t)frErrSlot(d, h, j) + (k < 8 ? 1u + (uint32_t)k : 0u);
            t.depth = (uint8_t)depth;
            t.pad[0] = (uint8_t)deg, t.pad[1] = t.pad[2] = 0;
            d.tasks[R->cFirst[c] + slot] = t;
        }
        if (live && k == 0) d.wJobH[j] = outOff;
        if (live && k == 8) d.wJobP[j] = outOff;
    }
    for (uint32_t b = gid; b < nBlocks; b += stride) {
        int lo = 0, hi = kFrClasses;  // the last class whose first workgroup is <= b is the one that owns b
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (R->cBlockFirst[mid] <= b)
                lo = mid;
            else
                hi = mid;
        }
        const int c = lo;
        const int deg = c / kFrDepths / 2;
        const bool incr = (c / kFrDepths) & 1;
        const uint32_t local = b - R->cBlockFirst[c], g = R->cG[c];
        FitBlock fb;
        fb.firstTask = R->cFirst[c] + local * g;
        const uint32_t left = R->cCount[c] - local * g;
        fb.nTasks = (uint16_t)(left < g ? left : g);
        fb.degree = (uint8_t)deg;
        fb.planesPerChunk = R->cPlanes[c];
        const bool split = frSplit(d.splitFit, deg, incr);
        fb.rowStart = (uint16_t)((incr || split) ? frCoef(deg - 1) : 0);
        fb.rowEnd = (uint16_t)frCoef(deg);
        fb.depth = (uint8_t)(c % kFrDepths);
        fb.weighted = d.weighted ? 1 : 0;
        fb.split = split ? 1 : 0;
        fb.pad1[0] = 0;
        d.blocks[b] = fb;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// decide: Octree.cpp:594-601 per job, and the index of every splitting job's first child
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fr_decide_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (h->done) return;
    __shared__ uint32_t sScan[16];
    __shared__ uint32_t sCnt[4];  // P, -, dropped, max degree
    __shared__ int sDelta;
    __shared__ uint32_t sPeer;
    const uint32_t tid = threadIdx.x, nJobs = h->nJobs, nNodes0 = h->nNodes;
    const bool round0 = h->round == 0;
    FR_STAMP(8);
    if (tid < 4) sCnt[tid] = 0;
    if (tid == 0) sDelta = 0, sPeer = 0;
    __syncthreads();
    frPeerCheck(d, &sPeer);
    // thread t owns jobs 4 t .. 4 t + 3 from the decision to the operand list: their errors stay in registers
    uint32_t nP = 0, nD = 0, maxDeg = 0;
    int delta = 0;
    int kd[4];
    double ev[4][9], be[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t j = tid * 4u + (uint32_t)q;
        kd[q] = 0;
        be[q] = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) ev[q][i] = 0.0;
        if (j >= nJobs) continue;
        const uint32_t idx = d.batchIdx[j];
        const double err = d.batchErr[j];
        be[q] = err;
        const int p = d.nodes[idx].degree, dep = d.nodes[idx].depth;
        const double* e = d.errs + frErrSlot(d, h, j);
        const bool coarse = fabs(err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
        const bool hasH = !coarse && dep < kMaxDepth, hasP = coarse || p < kMaxDegree - 1;
        if (hasP) ev[q][0] = e[0];
        if (hasH) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ev[q][1 + i] = e[1 + i];
        }
        const double pErr = ev[q][0];
        double pImp, hImp;
        if (coarse) {
            hImp = 0.0;   // :806-810
            pImp = pErr;  // :842
        } else {
            if (dep < kMaxDepth) {
                double maxNewErr = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) maxNewErr = maxNewErr < ev[q][1 + i] ? ev[q][1 + i] : maxNewErr;  // std::max
                hImp = (1.0 / (7.0 * (double)frCoef(p))) * (err - 8.0 * maxNewErr);  // :825
            } else {
                hImp = 0.0;
            }
            if (p < kMaxDegree - 1)
                pImp = (1.0 / (double)(frCoef(p + 1) - frCoef(p))) * (err - 8.0 * pErr);  // :854
            else
                pImp = 0.0;
        }
        bool refineP = p < (kMaxDegree - 1) && (dep == kMaxDepth || pImp > hImp);  // :600
        if (coarse) refineP = true;
        const bool refineH = dep < kMaxDepth && !refineP;  // :601
        const int kind = refineP ? 1 : (refineH ? 2 : 0);
        kd[q] = kind;
        d.kind[j] = (uint8_t)kind;
        if (kind == 1) {
            const int np = coarse ? 2 : p + 1;
            ++nP;
            maxDeg = (uint32_t)np > maxDeg ? (uint32_t)np : maxDeg;
            // coefficients gained (the template's subtree counts already stand for "every cell at degree 2")
            delta += (int)frCoef(np) - (int)(round0 ? frCoef(2) : frCoef(p));
        } else if (kind == 2) {
            delta += 7 * (int)frCoef(p);
        } else {
            ++nD;
        }
    }
    if (nP) atomicAdd(&sCnt[0], nP);
    if (nD) atomicAdd(&sCnt[2], nD);
    if (maxDeg) atomicMax(&sCnt[3], maxDeg);
    if (delta) atomicAdd(&sDelta, delta);
    FR_STAMP(9);
    // first child of every H job: node count so far + 8 x (H jobs before it).  The same scan (H count in the high half,
    // P count in the low half of one word) places every job's additions to the running total in a dense list: a P result
    // adds (newErr - initialErr) (:255); an H result subtracts initialErr once (:268) and adds its 8 children's errors (:272).
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) c += kd[q] == 2 ? 0x10000u : (kd[q] == 1 ? 1u : 0u);
    uint32_t inc = c;
    const int lane = (int)(tid & 63);
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(inc, off, 64);
        if (lane >= off) inc += v;
    }
    if (lane == 63) sScan[tid >> 6] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        if (w < (tid >> 6)) before += sScan[w];
        all += sScan[w];
    }
    FR_STAMP(10);
    uint32_t run = before + inc - c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t j = tid * 4u + (uint32_t)q;
        if (j >= nJobs) continue;
        const uint32_t hBefore = run >> 16, pBefore = run & 0xFFFFu;
        d.base[j] = nNodes0 + 8u * hBefore;
        double* o = d.ops + (pBefore + 9u * hBefore);
        if (kd[q] == 1) {
            o[0] = ev[q][0] - be[q];
            run += 1u;
        } else if (kd[q] == 2) {
            o[0] = be[q] * -1.0;  // total -= err  ==  total + (-err)
#pragma unroll
            for (int i = 0; i < 8; ++i) o[1 + i] = ev[q][1 + i];
            run += 0x10000u;
        }
    }
    if (tid == 0) {
        const uint32_t nH = all >> 16;
        h->rP = sCnt[0], h->rH = nH, h->rD = sCnt[2], h->rMaxDeg = sCnt[3];
        h->rOps = (all & 0xFFFFu) + 9u * nH;
        h->rPad = sPeer;  // (barriers lie between the check and here) a rank that failed this round, + 1
        h->dbg[11] = __builtin_readcyclecounter();
        h->rCoeffDelta = (int64_t)sDelta;
        h->arrive = 0;
        if (nNodes0 + 8u * nH > d.nodeCap) h->overflow = 1, h->done = 1;  // cannot happen: the host sizes for 8 K new nodes
    }
}

__device__ __forceinline__ double frReadLane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// Whoever arrives last closes the round: counters, stop rule (Octree.cpp:216), the threshold bin of the next selection,
// and the header's mirror in pinned host memory (the host waits for the stream and reads it there: no copy to launch).
__device__ void frCloseRound(const FrDev& d, uint32_t nJobs, bool round0, uint32_t* sHist, uint32_t* sTmp) {
    FrHdr* h = d.hdr;
    __shared__ int sLast, sT;
    __shared__ uint32_t sAbove;
    const uint32_t tid = threadIdx.x;
    __threadfence();
    __syncthreads();
    if (tid == 0) sLast = atomicAdd(&h->arrive, 1u) == gridDim.x - 1u ? 1 : 0;
    __syncthreads();
    if (!sLast) return;
    __threadfence();
    const uint32_t nP = *(volatile uint32_t*)&h->rP, nH = *(volatile uint32_t*)&h->rH, nD = *(volatile uint32_t*)&h->rD;
    const uint32_t nQ = h->nQueued - (round0 ? 0u : nJobs) + nP + 8u * nH;  // (round 0's batch never sat in the queue)
    const double total = *(volatile double*)&h->rTotal;
    const bool done = total < h->target || nQ == 0;  // Octree.cpp:216
    for (uint32_t i = tid; i < 2048; i += 256) {
        sHist[i] = *(volatile uint32_t*)&h->hist1[i];
        h->hist2[i] = 0;
    }
    __syncthreads();
    if (!done && nQ > d.K) {
        frThreshold(sHist, d.K, sTmp, &sT, &sAbove);
    } else {
        if (tid == 0) sT = -1, sAbove = nQ;
        __syncthreads();
    }
    if (tid == 0) {
        h->nQueued = nQ;
        h->nNodes += 8u * nH;
        h->total = total;
        h->jobs += nJobs, h->pRefines += nP, h->hRefines += nH, h->dropped += nD;
        h->nLeaves += 7u * nH;
        h->nCoeffs = (uint64_t)((int64_t)h->nCoeffs + *(volatile int64_t*)&h->rCoeffDelta);
        const uint32_t md = *(volatile uint32_t*)&h->rMaxDeg;
        if (md > h->maxDegree) h->maxDegree = md;
        h->round += 1;
        h->takenCount = 0, h->candCount = 0, h->arrive = 0;
        h->t1 = sT, h->above = sAbove;
        if (done && h->nCoeffs > d.storeCap) h->overflow = 2;  // the host grows the store and runs fr_store_kernel again
        h->done = done ? 1u : 0u;
        __threadfence();
    }
    __syncthreads();
    // the mirror: every word but the round number, a system-scope fence, then the round number -- the host watches that word
    constexpr uint32_t kRoundWord = offsetof(FrHdr, round) / 4;
    if (tid < kFrHdrCopyBytes / 4 && tid != kRoundWord)
        reinterpret_cast<volatile uint32_t*>(d.hostHdr)[tid] = reinterpret_cast<volatile uint32_t*>(h)[tid];
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        reinterpret_cast<volatile uint32_t*>(d.hostHdr)[kRoundWord] = reinterpret_cast<volatile uint32_t*>(h)[kRoundWord];
        __threadfence_system();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// update: Octree.cpp:243-299 in node-index order.  Workgroup 0 carries the running total (one dependent addition after
// the other, the reference's order); the others update tree and queue, eight lanes per job.  The workgroup that finishes
// last closes the round: counters, stop rule (:216), the next selection's threshold bin.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fr_update_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (h->done) return;  // (set by an earlier round, or by decide on overflow: uniform over the grid)
    __shared__ uint32_t sHist[2048];
    __shared__ uint32_t sTmp[256];
    __shared__ double sOps[2][2048];  // workgroup 0: the running total's operands, double-buffered
    const uint32_t tid = threadIdx.x, nJobs = h->nJobs;
    const bool round0 = h->round == 0;
    if (blockIdx.x == 0) {
        // The running total: one dependent addition after the other, in the dense order fr_decide_kernel laid out.  Wave 0
        // adds; waves 1..3 bring the next 2048 operands into the other half of the LDS buffer meanwhile (a lone wave
        // loading its own operands waits out an L2 round trip every 64 jobs: 129 us for round 0's 4096 additions).
        const uint32_t nOps = h->rOps;
        double total = h->total;
        for (uint32_t k = tid; k < 2048 && k < nOps; k += 256) sOps[0][k] = d.ops[k];
        __syncthreads();
        for (uint32_t c0 = 0, half = 0; c0 < nOps; c0 += 2048, half ^= 1u) {
            const uint32_t n = nOps - c0 < 2048u ? nOps - c0 : 2048u;
            if (tid >= 64) {
                const uint32_t nx = c0 + 2048;
                for (uint32_t k = tid - 64; k < 2048 && nx + k < nOps; k += 192) sOps[half ^ 1u][k] = d.ops[nx + k];
            } else {
                const double* src = sOps[half];
                uint32_t q = 0;
                for (; q + 16 <= n; q += 16) {
                    double o[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) o[k] = src[q + k];
#pragma unroll
                    for (int k = 0; k < 16; ++k) total = total + o[k];
                }
                for (; q < n; ++q) total = total + src[q];
            }
            __syncthreads();
        }
        if (tid == 0) h->rTotal = total;
    } else {
        for (uint32_t i = tid; i < 2048; i += 256) sHist[i] = 0;
        __syncthreads();
        const int sub = (int)(tid & 7);
        const uint32_t j = (blockIdx.x - 1u) * 32u + (tid >> 3);
        if (j < nJobs) {
            const uint32_t idx = d.batchIdx[j];
            const int kind = d.kind[j];
            const hpsdf_node par = d.nodes[idx];
            const int p = par.degree, dep = par.depth;
            const bool coarse = fabs(d.batchErr[j] - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
            int delta = 0;
            if (kind == 1) {  // :253-260, :286-290
                if (sub == 0) {
                    const int np = coarse ? 2 : p + 1;
                    // (weighted: the new array holds every row, so it is the node's one and only segment)
                    const int first = (coarse || d.weighted) ? np : (int)d.segFirst[idx];
                    if (coarse || d.weighted) d.segFirst[idx] = (uint8_t)np;
                    d.segOff[(size_t)idx * kFrSegs + (np - first)] = (d.jobP[j] & kOffMask) | ((uint64_t)(d.world == 1 ? 0 : d.jobOwner[j]) << 56);
                    d.nodes[idx].degree = (uint8_t)np;
                    const double pErr = d.errs[frErrSlot(d, h, j)];
                    const uint64_t bits = (uint64_t)__double_as_longlong(pErr);
                    d.qErr[idx] = bits;
                    atomicAdd(&sHist[frDigit(0, bits, idx)], 1u);
                    delta = (int)frCoef(np) - (int)(round0 ? frCoef(2) : frCoef(p));
                }
            } else if (kind == 2) {  // :262-279, :286-290; Octree::Subdivide :1115-1128
                const uint32_t c0 = d.base[j], ch = c0 + (uint32_t)sub;
                hpsdf_node c;
                c.child_idx = ~0ull;
                for (int a = 0; a < 3; ++a) {  // CornerAABB
                    const float mid = (par.aabb_max[a] + par.aabb_min[a]) * 0.5f;
                    c.aabb_min[a] = (sub >> a) & 1 ? mid : par.aabb_min[a];
                    c.aabb_max[a] = (sub >> a) & 1 ? par.aabb_max[a] : mid;
                }
                c.coeffs_start = 0;
                c.degree = (uint8_t)p;
                for (int a = 0; a < 7; ++a) c.pad0[a] = 0, c.pad1[a] = 0;
                c.depth = (uint8_t)(dep + 1);
                d.nodes[ch] = c;
                d.parent[ch] = idx;
                d.segFirst[ch] = (uint8_t)p;
                d.segOff[(size_t)ch * kFrSegs] = ((d.jobH[j] + (uint64_t)sub * frCoef(p)) & kOffMask) | ((uint64_t)(d.world == 1 ? 0 : d.jobOwner[j]) << 56);
                const double hErr = d.errs[frErrSlot(d, h, j) + 1 + sub];
                const uint64_t bits = (uint64_t)__double_as_longlong(hErr);
                d.qErr[ch] = bits;
                atomicAdd(&sHist[frDigit(0, bits, ch)], 1u);
                if (sub == 0) {
                    d.nodes[idx].child_idx = c0;
                    d.nodes[idx].degree = kInteriorDegree;
                    d.nodes[idx].coeffs_start = 0;
                    d.sub[idx] = 8u * frCoef(p);
                    delta = 7 * (int)frCoef(p);
                }
            }
            if (delta != 0) {  // coefficient counts of the ancestors (the root's is the header's nCoeffs)
                uint32_t a = d.parent[idx];
                while (a != 0) {
                    atomicAdd(&d.sub[a], (uint32_t)delta);
                    a = d.parent[a];
                }
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < 2048; i += 256)
            if (sHist[i]) atomicAdd(&h->hist1[i], sHist[i]);
    }
    frCloseRound(d, nJobs, round0, sHist, sTmp);
}

// Round 0 for itself: every job is a coarse cell that takes its degree-2 fit (:806-810, :836-843), so there is nothing to
// decide and no child to place -- workgroup 0 sums (newErr - 100) in job order straight from the fit's error slots, the
// others (one lane per cell) set degree, first segment and queued error.
__global__ __launch_bounds__(256) void fr_round0_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    __shared__ uint32_t sHist[2048];
    __shared__ uint32_t sTmp[256];
    __shared__ double sOps[2][2048];
    const uint32_t tid = threadIdx.x, nJobs = h->nJobs;
    if (blockIdx.x == 0) {
        double total = h->total;
        if (tid == 0) sTmp[0] = 0;
        __syncthreads();
        frPeerCheck(d, &sTmp[0]);
        for (uint32_t k = tid; k < 2048 && k < nJobs; k += 256) sOps[0][k] = d.errs[frErrSlot(d, h, k)] - d.batchErr[k];
        __syncthreads();
        const uint32_t peer = sTmp[0];
        for (uint32_t c0 = 0, half = 0; c0 < nJobs; c0 += 2048, half ^= 1u) {
            const uint32_t n = nJobs - c0 < 2048u ? nJobs - c0 : 2048u;
            if (tid >= 64) {
                const uint32_t nx = c0 + 2048;
                for (uint32_t k = tid - 64; k < 2048 && nx + k < nJobs; k += 192)
                    sOps[half ^ 1u][k] = d.errs[frErrSlot(d, h, nx + k)] - d.batchErr[nx + k];
            } else {
                const double* src = sOps[half];
                uint32_t q = 0;
                for (; q + 16 <= n; q += 16) {
                    double o[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) o[k] = src[q + k];
#pragma unroll
                    for (int k = 0; k < 16; ++k) total = total + o[k];
                }
                for (; q < n; ++q) total = total + src[q];
            }
            __syncthreads();
        }
        if (tid == 0) {
            h->rTotal = total;
            h->rPad = peer;
            h->rP = nJobs, h->rH = 0, h->rD = 0, h->rMaxDeg = 2, h->rCoeffDelta = 0;
        }
    } else {
        for (uint32_t i = tid; i < 2048; i += 256) sHist[i] = 0;
        __syncthreads();
        const uint32_t j = (blockIdx.x - 1u) * 256u + tid;
        if (j < nJobs) {
            const uint32_t idx = d.batchIdx[j];
            d.segFirst[idx] = 2;
            d.segOff[(size_t)idx * kFrSegs] = (d.jobP[j] & kOffMask) | ((uint64_t)(d.world == 1 ? 0 : d.jobOwner[j]) << 56);
            d.nodes[idx].degree = 2;
            const uint64_t bits = (uint64_t)__double_as_longlong(d.errs[frErrSlot(d, h, j)]);
            d.qErr[idx] = bits;
            atomicAdd(&sHist[frDigit(0, bits, idx)], 1u);
        }
        __syncthreads();
        for (uint32_t i = tid; i < 2048; i += 256)
            if (sHist[i]) atomicAdd(&h->hist1[i], sHist[i]);
    }
    frCloseRound(d, nJobs, true, sHist, sTmp);
}

// ---------------------------------------------------------------------------------------------------------------------
// ReallocCoeffs (Octree.cpp:474-555) once the stop rule has fired.  One wave per node.  A leaf's coeffsStart =
// coefficients of everything the depth-first walk (children 0..7 from the root) visits before it = over its
// ancestors-or-self a: the subtree sizes of a's earlier siblings.  Its rows are then gathered from the arena, segment by
// segment.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fr_store_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (!h->done || h->overflow) return;
    const uint32_t n = h->nNodes;
    const int lane = threadIdx.x & 63;
    for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) >> 6; i < n; i += gridDim.x * 4u) {
        const hpsdf_node nd = d.nodes[i];
        if (nd.degree == kInteriorDegree) continue;
        uint32_t start = 0;
        uint32_t a = i;
        while (a != 0) {  // lanes 0..6 look at the siblings before `a`
            const uint32_t par = d.parent[a];
            const uint32_t c0 = (uint32_t)d.nodes[par].child_idx;
            const uint32_t k = a - c0;
            uint32_t v = 0;
            if ((uint32_t)lane < k) {
                const hpsdf_node& sib = d.nodes[c0 + lane];
                v = sib.degree == kInteriorDegree ? d.sub[c0 + lane] : frCoef(sib.degree);
            }
            for (int off = 4; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);  // lanes 0..7 hold the sum
            start += __shfl(v, 0, 64);
            a = par;
        }
        if (lane == 0) d.nodes[i].coeffs_start = start;
        const int first = d.segFirst[i];
        for (int s = 0; s <= (int)nd.degree - first; ++s) {
            const uint32_t r0 = s == 0 ? 0u : frCoef(first + s - 1), r1 = frCoef(first + s);
            const uint64_t so = d.segOff[(size_t)i * kFrSegs + s];
            // one rank: straight from the arena; several: from the all-gathered pack buffers (fr_pack_kernel)
            const double* src = d.world == 1 ? d.arena + (so & kOffMask)
                                             : d.pack + (size_t)(so >> 56) * d.packStride + d.packPos[(size_t)i * kFrSegs + s];
            for (uint32_t r = r0 + (uint32_t)lane; r < r1; r += 64) d.store[(size_t)start + r] = src[r - r0];
        }
    }
}

// Several ranks: every rank holds the rows it fitted.  The packed store is reassembled from one all-gather of per-rank
// pack buffers: a rank's segments in (node index, segment) order.  fr_packpos_kernel (one workgroup) numbers them --
// thread t owns a contiguous run of nodes, per-rank running sums are scanned across the threads -- and leaves the
// per-rank totals in the header; fr_pack_kernel copies this rank's own segments into its buffer.
__global__ __launch_bounds__(1024) void fr_packpos_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (!h->done || h->overflow) return;
    __shared__ uint32_t sWave[16][8];
    const uint32_t n = h->nNodes, tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u, lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint32_t own[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = lo; i < hi; ++i) {
        const hpsdf_node& nd = d.nodes[i];
        if (nd.degree == kInteriorDegree) continue;
        const int first = d.segFirst[i];
        for (int sg = 0; sg <= (int)nd.degree - first; ++sg) {
            const uint32_t rows = frCoef(first + sg) - (sg == 0 ? 0u : frCoef(first + sg - 1));
            own[(d.segOff[(size_t)i * kFrSegs + sg] >> 56) & 7u] += rows;
        }
    }
    uint32_t inc[8];
    const int lane = (int)(tid & 63);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        inc[r] = own[r];
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(inc[r], off, 64);
            if (lane >= off) inc[r] += v;
        }
        if (lane == 63) sWave[tid >> 6][r] = inc[r];
    }
    __syncthreads();
    uint32_t run[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            if (w < (tid >> 6)) before += sWave[w][r];
            all += sWave[w][r];
        }
        run[r] = before + inc[r] - own[r];
        if (tid == 0) h->packCount[r] = all;
    }
    for (uint32_t i = lo; i < hi; ++i) {
        const hpsdf_node& nd = d.nodes[i];
        if (nd.degree == kInteriorDegree) continue;
        const int first = d.segFirst[i];
        for (int sg = 0; sg <= (int)nd.degree - first; ++sg) {
            const uint32_t rows = frCoef(first + sg) - (sg == 0 ? 0u : frCoef(first + sg - 1));
            const int o = (int)((d.segOff[(size_t)i * kFrSegs + sg] >> 56) & 7u);
            uint32_t pos = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r == o) pos = run[r], run[r] += rows;
            d.packPos[(size_t)i * kFrSegs + sg] = pos;
        }
    }
    __threadfence();
    __syncthreads();
    if (tid < kFrHdrCopyBytes / 4) reinterpret_cast<volatile uint32_t*>(d.hostHdr)[tid] = reinterpret_cast<volatile uint32_t*>(h)[tid];
    __threadfence_system();
}
__global__ __launch_bounds__(256) void fr_pack_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (!h->done || h->overflow) return;
    const uint32_t n = h->nNodes;
    const int lane = threadIdx.x & 63;
    double* mine = d.pack + (size_t)d.rank * d.packStride;
    for (uint32_t i = (blockIdx.x * 256u + threadIdx.x) >> 6; i < n; i += gridDim.x * 4u) {
        const hpsdf_node nd = d.nodes[i];
        if (nd.degree == kInteriorDegree) continue;
        const int first = d.segFirst[i];
        for (int sg = 0; sg <= (int)nd.degree - first; ++sg) {
            const uint64_t so = d.segOff[(size_t)i * kFrSegs + sg];
            if ((int)(so >> 56) != d.rank) continue;
            const uint32_t rows = frCoef(first + sg) - (sg == 0 ? 0u : frCoef(first + sg - 1));
            const double* src = d.arena + (so & kOffMask);
            double* dst = mine + d.packPos[(size_t)i * kFrSegs + sg];
            for (uint32_t r = (uint32_t)lane; r < rows; r += 64) dst[r] = src[r];
        }
    }
}

// Nearness weighting.  fr_means_done_kernel tells the host that the round's means have reached its memory (they were written
// there by fit_weight_kernel, the launch before this one: a kernel boundary on one stream); the host answers with the
// weights, and fr_weigh_kernel scales this rank's errors -- error * weight, Octree.cpp:1078-1086 -- before anything reads them.
__global__ void fr_means_done_kernel(FrDev d, uint32_t stamp) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    volatile uint32_t* f = d.hostFlag;
    f[0] = d.hdr->nJobs;
    __threadfence_system();
    f[1] = stamp;
    __threadfence_system();
}
__global__ __launch_bounds__(256) void fr_weigh_kernel(FrDev d, uint32_t stride) {  // stride 9: every error of a job; round 0: the first only
    const FrHdr* h = d.hdr;
    if (h->done) return;
    const uint32_t nJobs = h->nJobs;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nJobs * 9u) return;
    if (stride == 1u && i % 9u != 0u) return;
    d.errs[i] = d.errs[i] * d.weights[i];
}

// per-build initialisation: the uniformly refined tree (a copy of the context's template) and the header.  Round 0's
// batch, tasks and workgroups are the template's, used in place.
struct FrTemplate {
    const hpsdf_node* nodes;
    const uint32_t* parent;
    const uint32_t* sub;
    uint32_t nNodes, nLeaves, nTasks, nBlocks;  // nTasks / nBlocks / arenaRows / samples: this rank's share of round 0
    uint64_t arenaRows, samples;
    uint32_t sliceFirst[9];
};
__global__ __launch_bounds__(256) void fr_init_kernel(FrDev d, FrTemplate t, double target) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < t.nNodes) {
        d.nodes[i] = t.nodes[i];
        d.parent[i] = t.parent[i];
        d.sub[i] = t.sub[i];
        d.qErr[i] = kNotQueued;
        d.segFirst[i] = 2;
    }
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0 && d.world > 1) *frStatusSlot(d, d.rank) = 0.0;
    uint32_t* hw = reinterpret_cast<uint32_t*>(d.hdr);
    for (uint32_t w = threadIdx.x; w < sizeof(FrHdr) / 4; w += 256u) hw[w] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        FrHdr* h = d.hdr;
        h->nNodes = t.nNodes;
        h->nJobs = t.nLeaves;
        h->nLeaves = t.nLeaves;
        h->t1 = -1;
        h->total = 4096.0 * HPSDF_INITIAL_NODE_ERR;  // pow(8, 4) * INITIAL_NODE_ERR, Octree.cpp:212
        h->target = target;
        h->maxDegree = 2;
        h->nTasks = t.nTasks, h->nBlocks = t.nBlocks;
        h->degTasks[2][0] = 0, h->degTasks[2][1] = t.nTasks;
        h->degBlocks[2][0] = 0, h->degBlocks[2][1] = t.nBlocks;
        h->arenaUsed = t.arenaRows;
        h->sampleUsed = t.samples;
        h->fits = t.nTasks, h->samples = t.samples;
        h->nCoeffs = (uint64_t)t.nLeaves * frCoef(2);  // what the tree holds once round 0 has raised every cell to degree 2
        for (int r = 0; r < 9; ++r) h->sliceFirst[r] = t.sliceFirst[r];
    }
    if (d.world > 1)  // round 0's owners: the slices are equal runs of cells
        for (uint32_t j = threadIdx.x; j < t.nLeaves; j += 256u) {
            uint32_t o = 0;
            for (int r = 1; r < d.world; ++r) o += t.sliceFirst[r] <= j ? 1u : 0u;
            d.jobOwner[j] = (uint8_t)o;
        }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct FrontierWorkspace {
    int device = -1;
    bool inUse = false;
    FrDev d{};
    FrHdr* hostHdr = nullptr;  // pinned
    uint32_t nodeCap = 0;
    uint64_t arenaCap = 0, sampleCap = 0, storeCap = 0;
    double* arena = nullptr;
    double* samples = nullptr;
    // Template of the uniformly refined tree (Octree::UniformlyRefine, :112-191) and of round 0, which is the same for
    // every build: the 4096 depth-4 cells in index order, one from-scratch degree-2 fit each, cell j's rows at arena
    // offset 10 j (index order is the depth-first order of ReallocCoeffs, so a build that stops after round 0 finds
    // its packed coefficient store already sitting at the start of the arena).
    FrTemplate tmpl{};
    hpsdf_node* tmplNodes = nullptr;
    uint32_t* tmplParent = nullptr;
    uint32_t* tmplSub = nullptr;
    uint32_t* tmplLeaves = nullptr;
    double* tmplErr = nullptr;
    uint64_t* tmplJobP = nullptr;
    FitTask* tmplTasks = nullptr;
    FitBlock* tmplBlocks = nullptr;
    size_t tmplLds = 0;
    std::vector<hpsdf_node> hostNodesAfterRound0;  // the node array of a tree that stops after round 0, serialised
    char* pinned = nullptr;                        // staging of the finished block
    size_t pinnedCap = 0;
    // weighted builds: the round's |mean FApprox| values as the device writes them, the weights as the host answers, the
    // "means are there" word pair (pinned, coherent: both sides watch them while the other writes)
    double* hostMeans = nullptr;
    double* hostWeights = nullptr;
    uint32_t* hostFlag = nullptr;
    uint32_t flagStamp = 0;
    hipError_t ensureWeighting() {
        if (hostMeans) return hipSuccess;
        const size_t n = (size_t)kFrJobs * HPSDF_JOB_HEADER_DOUBLES;
        hipError_t e = hipHostMalloc((void**)&hostMeans, n * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostMalloc((void**)&hostWeights, n * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostMalloc((void**)&hostFlag, 64, hipHostMallocCoherent | hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&d.means, hostMeans, 0);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&d.weights, hostWeights, 0);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&d.hostFlag, hostFlag, 0);
        if (e == hipSuccess) {
            std::memset(hostMeans, 0, n * sizeof(double));
            for (size_t i = 0; i < n; ++i) hostWeights[i] = 1.0;
            hostFlag[0] = hostFlag[1] = 0;
        }
        return e;
    }
    // A round's fits are one launch per degree, and none of them fills the chip (a few hundred workgroups of two per CU):
    // degrees beyond the first go to side streams and run beside it
    static constexpr int kSide = 3;
    hipStream_t side[kSide] = {nullptr, nullptr, nullptr};
    hipEvent_t forkEv = nullptr, joinEv[kSide] = {nullptr, nullptr, nullptr};

    template <typename T>
    static hipError_t grow(T** p, size_t oldCount, size_t newCount, hipStream_t s, bool keep) {
        T* np = nullptr;
        hipError_t e = hipMalloc((void**)&np, newCount * sizeof(T));
        if (e != hipSuccess) return e;
        if (*p) {
            if (keep && oldCount) e = hipMemcpyAsync(np, *p, oldCount * sizeof(T), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            (void)hipFree(*p);
        }
        *p = np;
        return e;
    }
    hipError_t ensureNodes(uint32_t need, hipStream_t s) {
        if (need <= nodeCap) return hipSuccess;
        uint32_t nc = nodeCap ? nodeCap : 65536u;
        while (nc < need) nc *= 2;
        hipError_t e = grow(&d.nodes, nodeCap, nc, s, true);
        if (e == hipSuccess) e = grow(&d.qErr, nodeCap, nc, s, true);
        if (e == hipSuccess) e = grow(&d.parent, nodeCap, nc, s, true);
        if (e == hipSuccess) e = grow(&d.segOff, (size_t)nodeCap * kFrSegs, (size_t)nc * kFrSegs, s, true);
        if (e == hipSuccess) e = grow(&d.segFirst, nodeCap, nc, s, true);
        if (e == hipSuccess) e = grow(&d.sub, nodeCap, nc, s, true);
        if (e == hipSuccess) e = grow(&d.candA, nodeCap, nc, s, false);
        if (e == hipSuccess) e = grow(&d.candB, nodeCap, nc, s, false);
        if (e == hipSuccess && d.packPos) {
            e = grow(&d.packPos, 0, (size_t)nc * kFrSegs, s, false);
            if (e == hipSuccess) packPosCap = nc;
        }
        if (e == hipSuccess) nodeCap = nc, d.nodeCap = nc;
        return e;
    }
    hipError_t ensureArena(uint64_t need, uint64_t used, hipStream_t s) {
        if (need <= arenaCap) return hipSuccess;
        uint64_t nc = arenaCap ? arenaCap : (1ull << 22);
        while (nc < need) nc *= 2;
        hipError_t e = grow(&arena, used, nc, s, true);
        if (e == hipSuccess) arenaCap = nc, d.arena = arena;
        return e;
    }
    hipError_t ensureSamples(uint64_t need, hipStream_t s) {
        if (need <= sampleCap) return hipSuccess;
        uint64_t nc = sampleCap ? sampleCap : (1ull << 22);
        while (nc < need) nc *= 2;
        hipError_t e = grow(&samples, 0, nc, s, false);
        if (e == hipSuccess) sampleCap = nc;
        return e;
    }
    hipError_t ensureStore(uint64_t need, hipStream_t s) {
        if (need <= storeCap) return hipSuccess;
        uint64_t nc = storeCap ? storeCap : (1ull << 20);
        while (nc < need) nc *= 2;
        hipError_t e = grow(&d.store, 0, nc, s, false);
        if (e == hipSuccess) storeCap = nc, d.storeCap = nc;
        return e;
    }
    // multi-rank: errs is [world][4096 * 9], plus owners, pack positions, round-0 share
    int ranksCap = 1;
    uint64_t packCap = 0;
    FitTask* r0Tasks = nullptr;
    FitBlock* r0Blocks = nullptr;
    uint64_t* r0JobP = nullptr;
    int r0Rank = -1, r0World = -1;  // what (rank, world, error stride) the three r0 arrays were built for
    uint32_t r0ErrStride = 0;
    std::vector<FitTask> hostTmplTasks, hostR0Tasks;
    hipError_t ensureRanks(int world, hipStream_t s) {
        hipError_t e = hipSuccess;
        if (world > ranksCap) {
            e = grow(&d.errs, 0, (size_t)world * (kFrJobs * HPSDF_JOB_HEADER_DOUBLES + kFrStatusPad), s, false);
            if (e == hipSuccess) ranksCap = world;
        }
        if (e == hipSuccess && world > 1 && !d.jobOwner) {
            e = hipMalloc((void**)&d.jobOwner, kFrJobs);
            if (e == hipSuccess) e = hipMalloc((void**)&r0Tasks, kFrJobs * sizeof(FitTask));
            if (e == hipSuccess) e = hipMalloc((void**)&r0Blocks, kFrJobs * sizeof(FitBlock));
            if (e == hipSuccess) e = hipMalloc((void**)&r0JobP, kFrJobs * sizeof(uint64_t));
        }
        if (e == hipSuccess && world > 1 && packPosCap < nodeCap) {
            e = grow(&d.packPos, 0, (size_t)nodeCap * kFrSegs, s, false);
            if (e == hipSuccess) packPosCap = nodeCap;
        }
        return e;
    }
    uint32_t packPosCap = 0;
    hipError_t ensurePack(uint64_t need, hipStream_t s) {
        if (need <= packCap) return hipSuccess;
        uint64_t nc = packCap ? packCap : (1ull << 20);
        while (nc < need) nc *= 2;
        hipError_t e = grow(&d.pack, 0, nc, s, false);
        if (e == hipSuccess) packCap = nc;
        return e;
    }
    hipError_t ensurePinned(size_t need) {
        if (need <= pinnedCap) return hipSuccess;
        size_t nc = pinnedCap ? pinnedCap : (1u << 20);
        while (nc < need) nc *= 2;
        if (pinned) (void)hipHostFree(pinned);
        pinned = nullptr, pinnedCap = 0;
        hipError_t e = hipHostMalloc((void**)&pinned, nc, hipHostMallocDefault);
        if (e == hipSuccess) pinnedCap = nc;
        return e;
    }
    template <typename T>
    static hipError_t upload(T** dst, const std::vector<T>& v) {
        hipError_t e = hipMalloc((void**)dst, std::max<size_t>(1, v.size()) * sizeof(T));
        if (e == hipSuccess && !v.empty()) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
    hipError_t init(int dev, hipStream_t s) {
        device = dev;
        hipError_t e = hipMalloc((void**)&d.hdr, sizeof(FrHdr));
        if (e == hipSuccess) e = hipMalloc((void**)&d.rnd, sizeof(FrRound));
        if (e == hipSuccess) e = hipHostMalloc((void**)&hostHdr, sizeof(FrHdr), hipHostMallocCoherent | hipHostMallocMapped);  // (the host watches it while kernels write it)
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&d.hostHdr, hostHdr, 0);
        if (e == hipSuccess) e = hipMalloc((void**)&d.taken, (kFrJobs + 64) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.wBatchIdx, kFrJobs * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.wBatchErr, kFrJobs * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&d.wJobP, kFrJobs * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.wJobH, kFrJobs * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.kind, kFrJobs);
        if (e == hipSuccess) e = hipMalloc((void**)&d.base, kFrJobs * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.ops, (size_t)kFrJobs * 9 * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&d.tasks, kFrTasks * sizeof(FitTask));
        if (e == hipSuccess) e = hipMalloc((void**)&d.blocks, kFrTasks * sizeof(FitBlock));
        if (e == hipSuccess) e = hipMalloc((void**)&d.errs, (size_t)kFrJobs * HPSDF_JOB_HEADER_DOUBLES * sizeof(double));
        if (e != hipSuccess) return e;
        d.batchIdx = d.wBatchIdx, d.batchErr = d.wBatchErr, d.jobP = d.wJobP, d.jobH = d.wJobH;
        // the uniformly refined tree, from the host scheduler's own initialisation (builderBegin): identical indices
        hpsdf_build b;
        hpsdf_config cfg;
        hpsdf_config_default(&cfg);
        cfg.thread_count = 1;
        if (builderBegin(&b, &cfg, nullptr) != HPSDF_OK) return hipErrorUnknown;
        const uint32_t nT = (uint32_t)b.nodes.size();
        std::vector<uint32_t> parent(nT, 0), leaves, sub(nT, 0);
        for (uint32_t i = 0; i < nT; ++i) {
            if (b.nodes[i].child_idx != ~0ull)
                for (unsigned c = 0; c < 8; ++c) parent[b.nodes[i].child_idx + c] = i;
            else
                leaves.push_back(i);
        }
        const uint32_t nL = (uint32_t)leaves.size();
        if (nL > kFrJobs) return hipErrorUnknown;
        for (uint32_t i = nT; i-- > 1;) {  // children have larger indices than their parents
            const uint32_t own = b.nodes[i].child_idx == ~0ull ? frCoef(2) : sub[i];
            sub[parent[i]] += own;
        }
        // round 0: class (degree 2, from scratch, depth 4), slot j = job j
        int g = 1, pl = 1;
        frShape(2, false, nL, &g, &pl);
        std::vector<FitTask> tasks(nL);
        std::vector<uint64_t> jobP(nL);
        std::vector<double> errs(nL, HPSDF_INITIAL_NODE_ERR);
        for (uint32_t j = 0; j < nL; ++j) {
            const hpsdf_node& n = b.nodes[leaves[j]];
            FitTask& t = tasks[j];
            std::memset(&t, 0, sizeof t);
            for (int a = 0; a < 3; ++a) t.bmin[a] = n.aabb_min[a], t.bmax[a] = n.aabb_max[a];
            t.outOff = (uint64_t)j * frCoef(2);
            t.copyOff = ~0ull;
            t.sampleOff = (uint64_t)j * 729;
            t.errSlot = j * HPSDF_JOB_HEADER_DOUBLES;
            t.depth = n.depth;
            t.pad[0] = 2;
            jobP[j] = t.outOff;
        }
        std::vector<FitBlock> blocks((nL + g - 1) / g);
        for (uint32_t k = 0; k < blocks.size(); ++k) {
            FitBlock& fb = blocks[k];
            std::memset(&fb, 0, sizeof fb);
            fb.firstTask = k * (uint32_t)g;
            fb.nTasks = (uint16_t)std::min<uint32_t>((uint32_t)g, nL - k * (uint32_t)g);
            fb.degree = 2;
            fb.planesPerChunk = (uint8_t)pl;
            fb.rowStart = 0, fb.rowEnd = (uint16_t)frCoef(2);
            fb.depth = b.nodes[leaves[0]].depth;
        }
        tmplLds = frLds(2, g, pl);
        hostTmplTasks = tasks;
        hostNodesAfterRound0 = b.nodes;
        for (uint32_t j = 0; j < nL; ++j) {
            hostNodesAfterRound0[leaves[j]].degree = 2;
            hostNodesAfterRound0[leaves[j]].coeffs_start = (uint64_t)j * frCoef(2);
        }
        e = upload(&tmplNodes, b.nodes);
        if (e == hipSuccess) e = upload(&tmplParent, parent);
        if (e == hipSuccess) e = upload(&tmplSub, sub);
        if (e == hipSuccess) e = upload(&tmplLeaves, leaves);
        if (e == hipSuccess) e = upload(&tmplErr, errs);
        if (e == hipSuccess) e = upload(&tmplJobP, jobP);
        if (e == hipSuccess) e = upload(&tmplTasks, tasks);
        if (e == hipSuccess) e = upload(&tmplBlocks, blocks);
        tmpl.nodes = tmplNodes, tmpl.parent = tmplParent, tmpl.sub = tmplSub;
        tmpl.nNodes = nT, tmpl.nLeaves = nL, tmpl.nTasks = nL, tmpl.nBlocks = (uint32_t)blocks.size();
        tmpl.arenaRows = (uint64_t)nL * frCoef(2), tmpl.samples = (uint64_t)nL * 729;
        for (int k = 0; k < kSide && e == hipSuccess; ++k) {
            e = hipStreamCreateWithFlags(&side[k], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&joinEv[k], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&forkEv, hipEventDisableTiming);
        if (e == hipSuccess) e = ensureNodes(65536, s);
        if (e == hipSuccess) e = ensureArena(1ull << 22, 0, s);
        if (e == hipSuccess) e = ensureStore(1ull << 20, s);
        if (e == hipSuccess) e = ensurePinned(4u << 20);
        return e;
    }
    ~FrontierWorkspace() {
        if (device >= 0) (void)hipSetDevice(device);
        for (void* p : {(void*)d.hdr, (void*)d.rnd, (void*)d.nodes, (void*)d.qErr, (void*)d.parent, (void*)d.segOff, (void*)d.segFirst,
                        (void*)d.sub, (void*)d.taken, (void*)d.candA, (void*)d.candB, (void*)d.wBatchIdx, (void*)d.wBatchErr, (void*)d.wJobP,
                        (void*)d.wJobH, (void*)d.kind, (void*)d.base, (void*)d.ops, (void*)d.tasks, (void*)d.blocks, (void*)d.errs, (void*)d.store,
                        (void*)arena, (void*)samples, (void*)tmplNodes, (void*)tmplParent, (void*)tmplSub, (void*)tmplLeaves, (void*)tmplErr,
                        (void*)tmplJobP, (void*)tmplTasks, (void*)tmplBlocks, (void*)d.jobOwner, (void*)d.packPos, (void*)d.pack, (void*)r0Tasks,
                        (void*)r0Blocks, (void*)r0JobP})
            if (p) (void)hipFree(p);
        if (hostHdr) (void)hipHostFree(hostHdr);
        if (pinned) (void)hipHostFree(pinned);
        if (hostMeans) (void)hipHostFree(hostMeans);
        if (hostWeights) (void)hipHostFree(hostWeights);
        if (hostFlag) (void)hipHostFree(hostFlag);
        for (int k = 0; k < kSide; ++k) {
            if (side[k]) (void)hipStreamDestroy(side[k]);
            if (joinEv[k]) (void)hipEventDestroy(joinEv[k]);
        }
        if (forkEv) (void)hipEventDestroy(forkEv);
    }
};

bool frontierEligible(const hpsdf_config* cfg, const hpsdf_field* field, uint64_t K) {
    if (const char* e = std::getenv("HPSDF_HOST_FRONTIER"))
        if (e[0] == '1') return false;
    if (cfg->weighting_type > 2) return false;    // (unknown weighting: the host scheduler reports it)
    if (cfg->enable_logging) return false;        // the per-job log line is printed by the host scheduler
    const hpsdf_field* in = innermost(field);
    if (!in || (in->kind != kHostAnalytic && in->kind != kHostMesh)) return false;  // callbacks are sampled by host threads
    if (in->kind == kHostMesh) {
        const char* e = std::getenv("HPSDF_MESH_FUSED");
        if (e && e[0] == '1') return false;
    }
    const uint64_t k = K ? K : HPSDF_DEFAULT_JOBS_PER_ROUND;
    return k <= kFrJobs;
}

// A launch that failed (LDS over-subscription, a grid beyond the limits) used to surface a round later as "a round ended without
// advancing": every launch of the build loop reports its own failure at once.
#define FR_LAUNCH(kernel, grid, block, stream, ...)                                         \
    do {                                                                                    \
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                    \
        const hipError_t le_ = hipGetLastError();                                           \
        if (le_ != hipSuccess) return ::hpsdf::hipFail(le_, "launch of " #kernel);          \
    } while (0)

int frontierCreate(hpsdf_ctx* ctx, const hpsdf_config* cfgIn, const hpsdf_field* field, uint64_t K, void** block, size_t* size,
                   hpsdf_build_stats* stats, int rank, int world, hpsdf_allgather_fn gather, void* gatherUser) {
    if (world < 1 || world > 8 || rank < 0 || rank >= world) return fail(HPSDF_ERR_INVALID_ARGUMENT, "bad rank/world (1..8 ranks)");
    if (world > 1 && !gather) return fail(HPSDF_ERR_INVALID_ARGUMENT, "a multi-rank build needs an all-gather");
    const bool trace = std::getenv("HPSDF_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // Config::IsValid (Source/HP/Config.cpp:17-32), as builderBegin
    if (!(cfgIn->target_error_threshold > 0.0)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "targetErrorThreshold must be > 0");
    if (cfgIn->thread_count == 0) return fail(HPSDF_ERR_INVALID_ARGUMENT, "threadCount must be > 0");
    {
        float vol = 1.0f;
        for (int a = 0; a < 3; ++a) vol *= (cfgIn->root_max[a] - cfgIn->root_min[a]);
        if (!(vol > 0.0f)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "root volume must be > 0");
    }
    hpsdf_config cfg = *cfgIn;
    std::memset(cfg.pad0, 0, sizeof cfg.pad0);
    std::memset(cfg.pad1, 0, sizeof cfg.pad1);
    std::memset(cfg.pad2, 0, sizeof cfg.pad2);
    HPSDF_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    if (!ctx->frontierScratch) {
        auto w = std::make_shared<FrontierWorkspace>();
        const hipError_t e = w->init(ctx->device, s);
        if (e != hipSuccess) return hipFail(e, "frontier workspace");
        ctx->frontierScratch = w;
    }
    FrontierWorkspace* ws = static_cast<FrontierWorkspace*>(ctx->frontierScratch.get());
    if (ws->inUse) return fail(HPSDF_ERR_STATE, "one Create at a time per context");
    ws->inUse = true;
    struct Release {
        FrontierWorkspace* w;
        ~Release() { w->inUse = false; }
    } release{ws};
    const uint32_t Kj = (uint32_t)(K ? K : HPSDF_DEFAULT_JOBS_PER_ROUND);
    const bool weighted = cfg.weighting_type != 0;
    if (weighted) {
        if (cfg.weighting_type > 2) return fail(HPSDF_ERR_INVALID_ARGUMENT, "unknown nearnessWeighting.type");
        if (!(cfg.weighting_strength > 0.0)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "nearnessWeighting.strength must be > 0");
        // (on several ranks a weighted incremental fit needs the node's previous rows, which another rank may hold: the
        // host scheduler's sharded rounds exchange them, capi.cpp)
        if (world > 1) return fail(HPSDF_ERR_UNSUPPORTED, "weighted builds on several ranks run the host scheduler's rounds");
        const hipError_t e = ws->ensureWeighting();
        if (e != hipSuccess) return hipFail(e, "frontier weighting buffers");
    }
    ws->d.K = Kj;
    ws->d.rank = rank, ws->d.world = world;
    ws->d.weighted = weighted ? 1 : 0;
    ws->d.fastFit = (ctx->fitMode == HPSDF_FIT_FAST && field->kind != kHostTreeCsg && !weighted) ? 1 : 0;
    const bool splitMode = ctx->fitMode == HPSDF_FIT_SPLIT && !weighted;
    ws->d.splitFit = 0;
    ws->d.errStride = kFrJobs * HPSDF_JOB_HEADER_DOUBLES + kFrStatusPad;
    {
        const hipError_t e = ws->ensureRanks(world, s);
        if (e != hipSuccess) return hipFail(e, "frontier buffers");
    }
    const bool mesh = innermost(field)->kind == kHostMesh;
    auto exchange = [&](void* dBuf, size_t bytesPerRank, const char* what) -> int {  // in place: rank r's part at r * bytesPerRank
        if (world == 1) return HPSDF_OK;
        const int grc = gather(gatherUser, dBuf, bytesPerRank, (void*)s);
        if (grc != 0) return fail(HPSDF_ERR_STATE, std::string("the all-gather callback failed (") + what + ")");
        return HPSDF_OK;
    };

    // Several ranks: which exchange the other ranks enter next (0: none pending, 1: round 0's errors, 2: a later round's).  If this
    // rank's share fails while one is pending, it still enters that exchange -- with its status slot set (kFrStatusPad) -- before
    // it returns its error: the others then leave with HPSDF_ERR_STATE instead of waiting for a rank that has gone.
    int phase = 0;
    const char* injected = std::getenv("HPSDF_TEST_FAIL_RANK");  // tests: "<rank>:<round>"
    auto injectedFailure = [&](int round) {
        return injected && world > 1 && std::atoi(injected) == rank && std::strchr(injected, ':') && std::atoi(std::strchr(injected, ':') + 1) == round;
    };
    auto body = [&]() -> int {
    FieldDev fd;
    int rc;
    if (world > 1) phase = 1;
    if ((rc = makeFieldDev(field, nullptr, &fd))) return rc;
    if (injectedFailure(0)) return fail(HPSDF_ERR_OUT_OF_MEMORY, "injected failure (HPSDF_TEST_FAIL_RANK)");
    RootMap rm;
    for (int a = 0; a < 3; ++a) {
        rm.bounds[a] = (double)(cfg.root_max[a] - cfg.root_min[a]);          // Octree.cpp:324
        rm.centre[a] = (double)((cfg.root_min[a] + cfg.root_max[a]) / 2.0f);  // Octree.cpp:322
    }
    // launch-time LDS of a fit launch of `deg`: the largest shape the device may pick
    auto makeLdsTable = [](bool w) {
        std::vector<size_t> t(kMaxDegree + 1, 0);
        for (int deg = 1; deg <= kMaxDegree; ++deg)
            for (int incr = 0; incr < 2; ++incr) {
                int g, pl;
                frShape(deg, incr != 0, 1u << 20, &g, &pl, false, w);  // g = the class's largest
                for (int gg = 1; gg <= g; ++gg) {
                    const int nq = 4 * deg + 1;
                    int pp = nq;
                    while (pp > 1 && frLds(deg, gg, pp) > kFitChunkLdsBytes) --pp;
                    if (w) {  // frShape's weighted adjustment
                        const int minPlanes = ((int)frCoef(deg) + 100 + nq * nq - 1) / (nq * nq);
                        pp = std::max(pp, std::min(nq, minPlanes));
                        if (gg > 1 && frLds(deg, gg, pp) > kFitMaxLdsBytes) continue;  // (frShape stacks fewer cells)
                    }
                    t[deg] = std::max(t[deg], frLds(deg, gg, pp));
                }
            }
        return t;
    };
    static const std::vector<size_t> fitLdsPlain = makeLdsTable(false), fitLdsWeighted = makeLdsTable(true);
    const std::vector<size_t>& fitLdsTable = weighted ? fitLdsWeighted : fitLdsPlain;
    auto rowsPerJob = [](int pmax) {  // arena rows one job can need when no leaf exceeds degree pmax
        const int p = std::min(pmax, kMaxDegree - 1);
        return (uint64_t)8 * frCoef(p) + frCoef(std::min(p + 1, kMaxDegree));
    };
    auto samplesPerJob = [](int pmax) {
        const uint64_t a = 4 * (uint64_t)std::min(pmax, kMaxDegree - 1) + 1, b = a + 4;
        return 8 * a * a * a + b * b * b;
    };

    FrDev& d = ws->d;
    const FrTemplate& T = ws->tmpl;
    const FrHdr* hh = ws->hostHdr;
    double tSync = 0, tWeights = 0;
    // Weighted builds, once per round behind the fits: |mean FApprox| of every fit (fit_weight_kernel, straight into pinned
    // host memory) -> the weight, with the HOST's pow / exp (Octree.cpp:1224-1226, :1246: the libm the oracle and the host
    // scheduler call; a device pow would have to match it bit for bit) -> error * weight on the device (:1078-1086).
    // Everything else of the round -- selection, tasks, decision, bookkeeping, packing -- stays where it is.
    auto applyWeights = [&](const FitBlock* blocks, uint32_t maxBlocks, size_t lds, const FitTask* tasks, const uint32_t* dCount, bool round0) -> int {
        HPSDF_HIP(launchFitWeight(s, blocks, maxBlocks, lds, tasks, ws->arena, d.means, ctx->dTables, dCount));
        const uint32_t stamp = ++ws->flagStamp;
        FR_LAUNCH(fr_means_done_kernel, dim3(1), dim3(64), s, d, stamp);
        const double ts = now();
        {
            const volatile uint32_t* flag = ws->hostFlag;
            const double limit = ts + 2.0e3;
            while (flag[1] != stamp && now() < limit) frCpuRelax();
            if (flag[1] != stamp) HPSDF_HIP(hipStreamSynchronize(s));
            std::atomic_thread_fence(std::memory_order_acquire);
            if (flag[1] != stamp) return fail(HPSDF_ERR_STATE, "frontier: the round's means did not arrive");
        }
        const double tw = now();
        tSync += tw - ts;
        const uint32_t nJobs = std::min<uint32_t>(ws->hostFlag[0], kFrJobs);
        const double dd = std::sqrt(3.0), strength = cfg.weighting_strength;
        const double* mean = ws->hostMeans;
        double* w = ws->hostWeights;
        const uint32_t step = round0 ? 9u : 1u;  // round 0: every job is a coarse cell with one fit (slot 0 of its nine)
        for (uint32_t i = 0; i < nJobs * 9u; i += step) {
            if (cfg.weighting_type == 1) {
                const double k = std::pow(1.0 - mean[i] / dd, strength);
                w[i] = std::min<double>(1.0, std::max<double>(k, 0.0));
            } else {
                w[i] = std::exp(-1.0 * strength * mean[i] / dd);
            }
        }
        std::atomic_thread_fence(std::memory_order_release);
        tWeights += now() - tw;
        FR_LAUNCH(fr_weigh_kernel, dim3((std::max(1u, nJobs) * 9u + 255u) / 256u), dim3(256), s, d, round0 ? 1u : 9u);
        return HPSDF_OK;
    };
    uint8_t* early = nullptr;  // the block of a build that stops after round 0, begun before the device has finished
    struct FreeEarly {
        uint8_t** p;
        ~FreeEarly() { std::free(*p); }
    } freeEarly{&early};
    // ---- round 0: every cell of the uniformly refined tree, straight from the template
    FrTemplate T0 = T;  // (this rank's share when there are several)
    const FitTask* r0Tasks = ws->tmplTasks;
    const FitBlock* r0Blocks = ws->tmplBlocks;
    const uint64_t* r0JobP = ws->tmplJobP;
    size_t r0Lds = ws->tmplLds;
    T0.sliceFirst[0] = 0;
    for (int r = 1; r < 9; ++r) T0.sliceFirst[r] = T.nLeaves;
    if (world > 1) {
        // equal costs: slice ends where the scheduler's rule puts them (builderSelect: the first i with i c >= total (r + 1) / world)
        const uint64_t c = (uint64_t)frCoef(2) * 729ull, total = c * T.nLeaves;
        uint32_t start = 0;
        for (int r = 0; r < world; ++r) {
            uint32_t end = T.nLeaves;
            if (r + 1 < world) {
                const uint64_t target = total * (uint64_t)(r + 1) / (uint64_t)world;
                end = (uint32_t)((target + c - 1) / c);
                end = std::min(std::max(end, start), T.nLeaves);
            }
            T0.sliceFirst[r + 1] = end;
            start = end;
        }
        const uint32_t first = T0.sliceFirst[rank], count = T0.sliceFirst[rank + 1] - first;
        int g = 1, pl = 1;
        frShape(2, false, std::max(1u, count), &g, &pl);
        const uint32_t nBl = (count + (uint32_t)g - 1u) / (uint32_t)g;
        // this rank's task list, workgroup list and job -> arena map of round 0 depend on (rank, world, error stride) alone:
        // built and uploaded once, kept on the device for the Creates that follow (no upload, no wait per Create)
        if (ws->r0Rank != rank || ws->r0World != world || ws->r0ErrStride != ws->d.errStride) {
            std::vector<FitTask>& tk = ws->hostR0Tasks;
            tk.assign(ws->hostTmplTasks.begin() + first, ws->hostTmplTasks.begin() + first + count);
            std::vector<uint64_t> jobP(T.nLeaves, ~0ull & kOffMask);
            for (uint32_t q = 0; q < count; ++q) {
                tk[q].outOff = (uint64_t)q * frCoef(2);
                tk[q].sampleOff = (uint64_t)q * 729;
                tk[q].errSlot = (uint32_t)rank * ws->d.errStride + q * HPSDF_JOB_HEADER_DOUBLES;
                jobP[first + q] = tk[q].outOff;
            }
            std::vector<FitBlock> bl(nBl);
            for (uint32_t k = 0; k < bl.size(); ++k) {
                FitBlock& fb = bl[k];
                std::memset(&fb, 0, sizeof fb);
                fb.firstTask = k * (uint32_t)g;
                fb.nTasks = (uint16_t)std::min<uint32_t>((uint32_t)g, count - k * (uint32_t)g);
                fb.degree = 2;
                fb.planesPerChunk = (uint8_t)pl;
                fb.rowStart = 0, fb.rowEnd = (uint16_t)frCoef(2);
                fb.depth = ws->hostTmplTasks[0].depth;
            }
            ws->r0Rank = ws->r0World = -1;
            HPSDF_HIP(hipMemcpyAsync(ws->r0Tasks, tk.data(), count * sizeof(FitTask), hipMemcpyHostToDevice, s));
            HPSDF_HIP(hipMemcpyAsync(ws->r0Blocks, bl.data(), bl.size() * sizeof(FitBlock), hipMemcpyHostToDevice, s));
            HPSDF_HIP(hipMemcpyAsync(ws->r0JobP, jobP.data(), jobP.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
            HPSDF_HIP(hipStreamSynchronize(s));  // (pageable sources)
            ws->r0Rank = rank, ws->r0World = world, ws->r0ErrStride = ws->d.errStride;
        }
        r0Tasks = ws->r0Tasks, r0Blocks = ws->r0Blocks, r0JobP = ws->r0JobP;
        r0Lds = frLds(2, g, pl);
        T0.nTasks = count, T0.nBlocks = nBl;
        T0.arenaRows = (uint64_t)count * frCoef(2), T0.samples = (uint64_t)count * 729;
    }
    {
        hipError_t e = ws->ensureArena(std::max<uint64_t>(1, T0.arenaRows), 0, s);
        if (e == hipSuccess && mesh) e = ws->ensureSamples(std::max<uint64_t>(1, T0.samples), s);
        if (e != hipSuccess) return hipFail(e, "frontier buffers");
        FR_LAUNCH(fr_init_kernel, dim3((T.nNodes + 255) / 256), dim3(256), s, d, T0, cfg.target_error_threshold);
        FieldDev fdr = fd;
        if (mesh) {
            HPSDF_HIP(launchMeshSample(s, r0Tasks, T0.nTasks, 2, ctx->dTables, fd, rm, ws->samples));
            fdr.kind = kFieldSamples;
            fdr.samples = ws->samples;
        }
        HPSDF_HIP(launchFit(s, 2, 1, r0Blocks, T0.nBlocks, r0Lds, r0Tasks, ws->arena, d.errs, nullptr, ctx->dTables, fdr, rm));
        if (weighted && (rc = applyWeights(r0Blocks, T0.nBlocks, r0Lds, r0Tasks, nullptr, true))) return rc;
        if ((rc = exchange(d.errs, (size_t)d.errStride * sizeof(double), "round 0"))) return rc;
        phase = 0;
        FrDev d0 = d;
        d0.batchIdx = ws->tmplLeaves, d0.batchErr = ws->tmplErr, d0.jobP = r0JobP, d0.jobH = r0JobP;
        FR_LAUNCH(fr_round0_kernel, dim3(1 + (T.nLeaves + 255) / 256), dim3(256), s, d0);
        if (world == 1) {
            // a build that stops here has its packed store at the start of the arena: fetch it right behind the round
            HPSDF_HIP(hipMemcpyAsync(ws->pinned, ws->arena, T.arenaRows * sizeof(double), hipMemcpyDeviceToHost, s));
            // ... and everything else of its block is known in advance: write that part while the device works
            const uint64_t nc0 = T.arenaRows, nn0 = T.nNodes;
            early = (uint8_t*)std::malloc(8 + 8 * (size_t)nc0 + 8 + sizeof(hpsdf_node) * (size_t)nn0 + sizeof(hpsdf_config));
            if (early) {
                std::memcpy(early, &nc0, 8);
                std::memcpy(early + 8 + 8 * (size_t)nc0, &nn0, 8);
                std::memcpy(early + 16 + 8 * (size_t)nc0, ws->hostNodesAfterRound0.data(), sizeof(hpsdf_node) * (size_t)nn0);
                std::memcpy(early + 16 + 8 * (size_t)nc0 + sizeof(hpsdf_node) * (size_t)nn0, &cfg, sizeof cfg);
            }
        }
        const double ts = now();
        HPSDF_HIP(hipStreamSynchronize(s));
        tSync += now() - ts;
    }
    int rounds = 1;
    if (world > 1 && hh->rPad) return fail(HPSDF_ERR_STATE, "rank " + std::to_string(hh->rPad - 1) + " failed in round 0: its own error was returned there");
    d.errStride = Kj * HPSDF_JOB_HEADER_DOUBLES + kFrStatusPad;  // later rounds have at most K jobs: smaller parts to all-gather
    const bool stoppedAfterRound0 = world == 1 && hh->done && hh->overflow != 1;
    if (stoppedAfterRound0 && early && hh->nCoeffs == T.arenaRows && hh->nNodes == T.nNodes) {
        std::memcpy(early + 8, ws->pinned, 8 * (size_t)T.arenaRows);
        *block = early;
        *size = 8 + 8 * (size_t)T.arenaRows + 8 + sizeof(hpsdf_node) * (size_t)T.nNodes + sizeof(hpsdf_config);
        early = nullptr;
        if (stats) {
            std::memset(stats, 0, sizeof *stats);
            stats->rounds = hh->round, stats->jobs = hh->jobs, stats->p_refines = hh->pRefines, stats->h_refines = hh->hRefines;
            stats->dropped = hh->dropped, stats->fits = hh->fits, stats->samples = hh->samples;
            stats->n_nodes = hh->nNodes, stats->n_leaves = hh->nLeaves, stats->n_coeffs = hh->nCoeffs, stats->total_error = hh->total;
        }
        if (trace) std::fprintf(stderr, "[frontierCreate] us: total %.0f (waiting for the device %.0f, stopped after round 0)\n", now() - t0, tSync);
        return HPSDF_OK;
    }
    std::free(early);
    early = nullptr;
    uint32_t knownNodes = hh->nNodes, knownMaxDeg = hh->maxDegree;
    uint64_t knownArena = hh->arenaUsed;
    while (!hh->done) {
        // (as builderSelect: a total that is NaN or infinite never falls below the threshold -- the field is not finite somewhere)
        if (!(std::fabs(hh->total) <= DBL_MAX))
            return fail(HPSDF_ERR_INVALID_ARGUMENT, "the field is not a finite number at some sample point (the build's total error is NaN or infinite)");
        if (world > 1) phase = 2;
        if (injectedFailure(rounds)) return fail(HPSDF_ERR_OUT_OF_MEMORY, "injected failure (HPSDF_TEST_FAIL_RANK)");
        // capacities for this round (the device flags what the host failed to foresee; it cannot happen by these bounds)
        hipError_t e = ws->ensureNodes(knownNodes + 8u * Kj, s);
        if (e == hipSuccess) e = ws->ensureArena(knownArena + (uint64_t)Kj * rowsPerJob((int)knownMaxDeg), knownArena, s);
        // split fits (the default for from-scratch fits of degree >= 4) hand their field values to the matrix-core kernel through the
        // sample buffer, which mesh fields use anyway; a round whose samples would not fit 16 GB is fitted exactly throughout
        bool splitRound = splitMode && (int)knownMaxDeg >= ctx->splitMinDegree;
        if (e == hipSuccess && (mesh || splitRound)) {
            const uint64_t need = (uint64_t)Kj * samplesPerJob((int)knownMaxDeg);
            if (need > (1ull << 31)) {
                if (mesh) return fail(HPSDF_ERR_UNSUPPORTED, "round too large for the sampled mesh path");
                splitRound = false;
            } else {
                e = ws->ensureSamples(need, s);
            }
        }
        d.splitFit = splitRound ? std::max(2, ctx->splitMinDegree) : 0;
        if (e != hipSuccess) return hipFail(e, "frontier buffers");
        FR_LAUNCH(fr_select_kernel, dim3(std::min<uint32_t>(512u, (knownNodes + 255u) / 256u)), dim3(256), s, d);
        FR_LAUNCH(fr_batch_kernel, dim3(1), dim3(1024), s, d);
        FR_LAUNCH(fr_tasks_kernel, dim3((16u * Kj + 255u) / 256u), dim3(256), s, d);
        const int degHi = (int)std::min<uint32_t>(kMaxDegree - 1, knownMaxDeg + 1);
        const uint32_t taskBound = 9u * Kj;
        bool degHiDone = false;
        FieldDev fdr = fd;
        if (mesh) {
            for (int deg = 2; deg <= degHi; ++deg)
                HPSDF_HIP(launchMeshSampleRange(s, d.tasks, &d.hdr->degTasks[deg][0], std::min<uint32_t>(taskBound, 65535u), deg, ctx->dTables, fd,
                                                rm, ws->samples));
            fdr.kind = kFieldSamples;
            fdr.samples = ws->samples;
        } else if (splitRound) {
            fdr.samples = ws->samples;  // (the exact kernel writes the field values of split fits there)
        }
        // One launch for every degree of the round (kernels.hip fit_multi_kernel); the matrix-core fit and
        // HPSDF_FRONTIER_SPLIT_FITS=1 keep one launch per degree, side by side on three streams.
        static const bool splitFits = std::getenv("HPSDF_FRONTIER_SPLIT_FITS") != nullptr;
        if (!d.fastFit && !splitFits) {
            size_t lds = 0;
            for (int deg = 2; deg <= degHi; ++deg) lds = std::max(lds, fitLdsTable[deg]);
            HPSDF_HIP(launchFitMulti(s, d.blocks, taskBound, lds, d.tasks, ws->arena, d.errs, ctx->dTables, fdr, rm, &d.hdr->nBlocks));
            degHiDone = true;
        }
        const bool fork = !degHiDone && degHi > 2 && std::getenv("HPSDF_FRONTIER_ONE_STREAM") == nullptr;
        if (fork) HPSDF_HIP(hipEventRecord(ws->forkEv, s));
        bool used[FrontierWorkspace::kSide] = {false, false, false};
        for (int deg = 2; deg <= degHi && !degHiDone; ++deg) {
            hipStream_t fs = s;
            if (fork && deg > 2) {
                const int k = (deg - 3) % FrontierWorkspace::kSide;
                fs = ws->side[k];
                if (!used[k]) HPSDF_HIP(hipStreamWaitEvent(fs, ws->forkEv, 0));
                used[k] = true;
            }
            if (d.fastFit && deg >= 4 && deg <= 11)
                HPSDF_HIP(launchFitMfma(fs, deg, d.blocks, taskBound, d.tasks, ws->arena, d.errs, ctx->dTables, fdr, rm, &d.hdr->degBlocks[deg][0]));
            else
                HPSDF_HIP(launchFit(fs, deg <= 5 ? deg : 0, 1, d.blocks, taskBound, fitLdsTable[deg], d.tasks, ws->arena, d.errs, nullptr,
                                    ctx->dTables, fdr, rm, &d.hdr->degBlocks[deg][0]));
        }
        for (int k = 0; k < FrontierWorkspace::kSide; ++k)
            if (used[k]) {
                HPSDF_HIP(hipEventRecord(ws->joinEv[k], ws->side[k]));
                HPSDF_HIP(hipStreamWaitEvent(s, ws->joinEv[k], 0));
            }
        if (splitRound)  // the rows below the top degree of the split fits, from the samples the exact kernel left (H children: degree <= knownMaxDeg)
            for (int deg = std::max(2, ctx->splitMinDegree); deg <= (int)std::min<uint32_t>(knownMaxDeg, 11u); ++deg)
                HPSDF_HIP(launchFitMfmaLow(s, deg, d.tasks, &d.hdr->lowTasks[deg][0], 0u, 0u, 8u * Kj, ws->arena, ctx->dTables, ws->samples, rm));
        if (weighted) {
            size_t lds = 0;
            for (int deg = 2; deg <= degHi; ++deg) lds = std::max(lds, fitLdsTable[deg]);
            if ((rc = applyWeights(d.blocks, taskBound, lds, d.tasks, &d.hdr->nBlocks, false))) return rc;
        }
        if ((rc = exchange(d.errs, (size_t)d.errStride * sizeof(double), "a round's errors"))) return rc;
        phase = 0;
        FR_LAUNCH(fr_decide_kernel, dim3(1), dim3(1024), s, d);
        FR_LAUNCH(fr_update_kernel, dim3(1 + (Kj + 31) / 32), dim3(256), s, d);
        if (world == 1) FR_LAUNCH(fr_store_kernel, dim3(std::min<uint32_t>(2048u, (knownNodes + 8u * Kj + 3u) / 4u)), dim3(256), s, d);
        // The round is over for the host when the header's mirror shows the next round number: the update kernel's last
        // workgroup writes it into pinned memory behind a system-scope fence.  Watching that word costs ~3 us; waking up
        // from hipStreamSynchronize ~20 (everything launched next is ordered behind this round on the stream anyway;
        // the download of a finished build synchronises the stream itself).
        const double ts = now();
        {
            const volatile uint32_t* roundWord = &hh->round;
            const uint32_t want = (uint32_t)rounds + 1u;
            const double limit = ts + 2.0e3;  // two milliseconds of watching, then the ordinary wait
            while (*roundWord != want && now() < limit) frCpuRelax();
            if (*roundWord != want) {
                HPSDF_HIP(hipStreamSynchronize(s));
                if (*roundWord != want) {
                    // The stream is idle and the mirror still shows the old round: the round ended on a path that does not
                    // write it (fr_decide_kernel's capacity overflow makes fr_update_kernel return early), or the pinned
                    // mirror is not coherent on this system.  Fetch the header itself and let it decide.
                    HPSDF_HIP(hipMemcpy(ws->hostHdr, d.hdr, kFrHdrCopyBytes, hipMemcpyDeviceToHost));
                    if (hh->overflow == 1) return fail(HPSDF_ERR_STATE, "frontier: node capacity exceeded");
                    if (hh->round != want && !hh->done) return fail(HPSDF_ERR_STATE, "frontier: a round ended without advancing");
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        tSync += now() - ts;
        ++rounds;
        if (world > 1 && hh->rPad)
            return fail(HPSDF_ERR_STATE, "rank " + std::to_string(hh->rPad - 1) + " failed in round " + std::to_string(rounds - 1) + ": its own error was returned there");
        knownNodes = hh->nNodes, knownMaxDeg = hh->maxDegree, knownArena = hh->arenaUsed;
        if (trace) {
            std::fprintf(stderr, "[frontier round %d] batch phases (cycles):", rounds - 1);
            for (int k = 1; k <= 7; ++k) std::fprintf(stderr, " %lld", (long long)(hh->dbg[k] - hh->dbg[k - 1]));
            std::fprintf(stderr, " | decide:");
            for (int k = 9; k <= 11; ++k) std::fprintf(stderr, " %lld", (long long)(hh->dbg[k] - hh->dbg[k - 1]));
            std::fprintf(stderr, "\n");
        }
    }
    if (hh->overflow == 1) return fail(HPSDF_ERR_STATE, "frontier: node capacity exceeded");
    if (world > 1) {
        // the packed store from one all-gather of the ranks' pack buffers (each rank's own segments in node order)
        hipError_t e = ws->ensureStore(hh->nCoeffs, s);
        if (e != hipSuccess) return hipFail(e, "coefficient store");
        HPSDF_HIP(hipMemsetAsync(&d.hdr->overflow, 0, sizeof(uint32_t), s));
        FR_LAUNCH(fr_packpos_kernel, dim3(1), dim3(1024), s, d);
        HPSDF_HIP(hipStreamSynchronize(s));
        uint64_t stride = 1;
        for (int r = 0; r < world; ++r) stride = std::max<uint64_t>(stride, hh->packCount[r]);
        stride = (stride + 15) & ~15ull;
        e = ws->ensurePack((uint64_t)world * stride, s);
        if (e != hipSuccess) return hipFail(e, "pack buffers");
        d.packStride = stride;
        FR_LAUNCH(fr_pack_kernel, dim3(std::min<uint32_t>(2048u, (knownNodes + 3u) / 4u)), dim3(256), s, d);
        if ((rc = exchange(d.pack, (size_t)stride * sizeof(double), "the packed coefficients"))) return rc;
        FR_LAUNCH(fr_store_kernel, dim3(std::min<uint32_t>(2048u, (knownNodes + 3u) / 4u)), dim3(256), s, d);
        HPSDF_HIP(hipStreamSynchronize(s));
    } else if (hh->overflow == 2) {  // the packed store was too small: grow, run ReallocCoeffs again
        hipError_t e = ws->ensureStore(hh->nCoeffs, s);
        if (e != hipSuccess) return hipFail(e, "coefficient store");
        HPSDF_HIP(hipMemsetAsync(&d.hdr->overflow, 0, sizeof(uint32_t), s));
        FR_LAUNCH(fr_store_kernel, dim3(std::min<uint32_t>(2048u, (knownNodes + 3u) / 4u)), dim3(256), s, d);
        HPSDF_HIP(hipStreamSynchronize(s));
    }
    // Octree::ToMemoryBlock, Octree.cpp:424-456: [u64 nCoeffs][f64 x nCoeffs][u64 nNodes][Node x nNodes][Config]
    const uint64_t nc = hh->nCoeffs, nn = hh->nNodes;
    const size_t bytes = 8 + 8 * (size_t)nc + 8 + sizeof(hpsdf_node) * (size_t)nn + sizeof(hpsdf_config);
    uint8_t* p = (uint8_t*)std::malloc(bytes);
    if (!p) return fail(HPSDF_ERR_OUT_OF_MEMORY, "malloc of the memory block failed");
    const double tc = now();
    const hpsdf_node* nodesSrc;
    if (stoppedAfterRound0) {
        if (nc != T.arenaRows || nn != T.nNodes) {
            std::free(p);
            return fail(HPSDF_ERR_STATE, "frontier: round-0 tree does not match its template");
        }
        nodesSrc = ws->hostNodesAfterRound0.data();  // (its coefficients came with the header)
    } else {
        hipError_t e = ws->ensurePinned(8 * (size_t)nc + sizeof(hpsdf_node) * (size_t)nn);
        if (e == hipSuccess && nc) e = hipMemcpyAsync(ws->pinned, d.store, 8 * (size_t)nc, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(ws->pinned + 8 * (size_t)nc, d.nodes, sizeof(hpsdf_node) * (size_t)nn, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            std::free(p);
            return hipFail(e, "block download");
        }
        nodesSrc = reinterpret_cast<const hpsdf_node*>(ws->pinned + 8 * (size_t)nc);
    }
    const double tcopy = now() - tc;
    std::memcpy(p, &nc, 8);
    std::memcpy(p + 8, ws->pinned, 8 * (size_t)nc);
    std::memcpy(p + 8 + 8 * (size_t)nc, &nn, 8);
    std::memcpy(p + 16 + 8 * (size_t)nc, nodesSrc, sizeof(hpsdf_node) * (size_t)nn);
    std::memcpy(p + 16 + 8 * (size_t)nc + sizeof(hpsdf_node) * (size_t)nn, &cfg, sizeof cfg);
    *block = p;
    *size = bytes;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->rounds = hh->round, stats->jobs = hh->jobs, stats->p_refines = hh->pRefines, stats->h_refines = hh->hRefines;
        stats->dropped = hh->dropped, stats->fits = hh->fits, stats->samples = hh->samples;
        stats->n_nodes = nn, stats->n_leaves = hh->nLeaves, stats->n_coeffs = nc, stats->total_error = hh->total;
    }
    if (trace)
        std::fprintf(stderr, "[frontierCreate] us: total %.0f (waiting for the device %.0f over %d rounds, weights on the host %.0f, block download %.0f)\n",
                     now() - t0, tSync, rounds, tWeights, tcopy);
    return HPSDF_OK;
    };
    const int rcBody = body();
    if (rcBody && world > 1 && phase != 0) {
        const std::string own = hpsdf_last_error();
        const size_t stride = ws->d.errStride;
        const unsigned long long ones = ~0ull;
        if (hipMemcpyAsync(ws->d.errs + (size_t)rank * stride + (stride - 1), &ones, sizeof ones, hipMemcpyHostToDevice, s) == hipSuccess)
            (void)gather(gatherUser, ws->d.errs, stride * sizeof(double), (void*)s);
        (void)hipStreamSynchronize(s);
        return fail(rcBody, own);
    }
    return rcBody;
}

}  // namespace hpsdf
