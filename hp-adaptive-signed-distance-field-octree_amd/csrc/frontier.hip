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
// A round (r >= 1) is TWO launches around its fits (round 5; rounds 2-4 ran six small kernels, ~100 us of their own latencies a round):
//   fr_round_kernel    closes round r - 1 and opens round r in one launch of 2 + K / 128 workgroups of 1024 threads:
//        workgroups 1.. : eight lanes per job -- the decision (:594-601: improvements :814-825, :846-854), child creation
//                         (Subdivide / CornerAABB), queue and histogram, the round's additions to the running total as a dense list.
//                         A splitting job's first child is nNodes + 8 x (splitting jobs before it): every workgroup publishes its
//                         own counts and adds up its predecessors' (one hop, no chain), so nobody has to see all jobs first
//        the last one   : the running total in the reference's order (:253-290), one dependent v_add_f64 after the other, the
//                         operands double-buffered through LDS by twelve loader waves (frRunChain) -- beside everything else:
//                         nobody waits for it before the next round is prepared
//        workgroup 0    : when all have arrived: counters, the next selection's threshold bin; then -- trees up to
//                         HPSDF_FRONTIER_INLINE_NODES -- the NEXT round's selection (top-K of the frontier by (error desc, node index
//                         asc) as an MSB radix select over the error's bit pattern), the batch in node order, the round's shape
//                         classes and every job's record, on the assumption that the build goes on; the stop rule (:216) when the
//                         total has come (a build that stops takes the prepared round out of the header again); the header's
//                         mirror in pinned host memory tells the host.  Round 0's total (4096 operands that are there when the
//                         launch starts) it adds up itself
//   fr_emit_kernel     the round's FitTask / FitBlock lists from the job records (grid: 128 jobs a workgroup, a lane per fit)
//   mesh_sample_kernel / fit_multi_kernel   (kernels.hip) over device-written ranges: grids are upper bounds.  From the second
//                      round on both are enqueued without waiting for the header: a build that has stopped leaves them nothing to do
// Larger trees select with a grid: fr_select_kernel (level 0 against the histogram the updates keep), fr_batch_kernel (one
// workgroup: remaining digits, exact order, classes), fr_tasks_kernel (grid: the lists).
//   fr_subtree_kernel / fr_store_kernel   once the stop rule has fired: ReallocCoeffs -- the coefficient count of every subtree, every
//                      leaf's coeffsStart by walking up its ancestors, coefficients and node array written straight into pinned host
//                      memory in ToMemoryBlock's order (the host copies each part into the block as its flag appears)
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
// (fr_init_kernel, the round kernel's leader or fr_batch_kernel); a rank whose share of the round failed on the host (device memory) sets it to all ones and
// enters the exchange all the same; the round kernel's leader finds it (frPeerCheck) and every rank leaves with an error instead of
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
    uint64_t splitFits;   // from-scratch fits whose lower rows came from the split's second kernel (hpsdf_build_stats::split_fits)
    uint32_t splitRound, padR;  // FrDev::splitFit if the round being prepared splits its from-scratch fits, else 0 (decided by the batch)
    uint64_t roundBase, partStride;  // FrDev::replica: the round's fits write arena rows [roundBase + r partStride, + partStride) on rank r
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
    uint32_t opsArrive, stuck;  // fr_round_kernel: update workgroups whose operands are written; a wait that ran out (never, by construction)
    uint32_t chainStamp;  // round + 1 once rTotal holds that round's running total
    uint32_t landed;      // (mirror only) the build's stamp, written when round 0's closing launch starts: the fit before it has finished, so
                          // the rows it wrote into pinned host memory are there
    uint32_t stored[2];   // (mirror only) the build's stamp once fr_store_kernel's node array / coefficients are in pinned host memory
    uint32_t storeArrive[2];  // workgroups of fr_store_kernel that have finished a phase
    uint32_t hist1[2048];  // queued nodes per exponent bin (kept by update / batch)
    uint32_t hist2[2048];  // level-1 digits of the candidates of the current selection (grid selection only)
    uint64_t agg[kFrJobs / 128];  // fr_round_kernel: per update workgroup {round stamp, splitting jobs << 16 | P jobs}
};
constexpr size_t kFrHdrCopyBytes = offsetof(FrHdr, hist1);
static_assert(kFrHdrCopyBytes / 4 <= 1024, "frMirror copies the header with one thread a word");
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
    uint64_t cArenaO[8][kFrClasses];  // FrDev::replica: arena rows before class c within owner o's part of the round
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
    double* ops;               // the round's additions to the running total, densely, in job order
    uint32_t* jobRecA;         // per job of the round being opened, for fr_emit_kernel: H slot | P slot << 16 | degree << 28
    uint16_t* jobRecB;         //   coarse | ours << 1 | owner << 2 | depth << 5
    FitTask* tasks;
    FitBlock* blocks;
    double* errs;       // [jobs][9]
    double* store;      // the finished block's coefficients and, behind them, its node array: pinned host memory as the device addresses it
    const double* arena;
    uint32_t nodeCap, K;
    // multi-rank builds (world > 1): this rank fits a cost-balanced slice of every round's jobs
    int32_t rank, world;
    uint32_t errStride;    // doubles per rank in errs: rank r's slice sits at errs + r * errStride (all-gathered in place)
    uint32_t errStrideNext;  // ... in the round being prepared (round 0 exchanges 4096 jobs' errors, later rounds K)
    uint8_t* jobOwner;     // per job: the rank that fits it
    uint32_t* packPos;     // [node][kFrSegs]: where the segment sits in its owner's pack buffer
    double* pack;          // [world][packStride] all-gathered pack buffers (this rank writes its own)
    uint64_t packStride;
    int32_t fastFit;       // degrees >= 4 fitted by fit_mfma.hip: 16 cells per workgroup
    int32_t splitFit;      // 0, or the lowest degree (.. 11) whose from-scratch fits are split: top-degree rows exact, the rest by fit_low_kernel.
                           // Whether a round does split is the batch's decision (FrHdr::splitRound), by the host scheduler's rule
    uint64_t sampleCap;    // doubles the sample buffer holds (the split's hand-over needs the round's samples to fit)
    // nearness weighting (Octree.cpp:1071-1092, 1209-1247): a fit keeps ONE full coefficient array (an incremental fit
    // carries the old rows over, :847), fit_weight_kernel leaves |mean FApprox| of every fit in `means`, the host turns
    // the means into weights with its libm (pow / exp: what the oracle calls) and fr_weigh_kernel scales the errors
    uint32_t buildStamp;  // a number of the build (FrHdr::landed)
    uint32_t stamps;      // HPSDF_TRACE: the one-workgroup kernels leave their phases' times in FrHdr::dbg (a store a phase, and the next barrier waits for it)
    double target;         // targetErrorThreshold of the build (the header is initialised before the build is known: FrontierWorkspace::clean)
    int32_t weighted;
    // Weighted builds on several ranks: every rank's arena is a REPLICA.  An incremental weighted fit carries the cell's previous rows
    // over (:847), and those may have been fitted anywhere -- so a round's fits are laid out owner by owner at the same offsets on every
    // rank (FrHdr::roundBase / partStride), every rank fills its own part, and ONE in-place all-gather of the round's part of the arena
    // (beside the errors') leaves every rank with every row.  Segments then need no owner, the packed store no exchange of its own.
    int32_t replica;
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
constexpr size_t kFrFitLdsCap = 45 * 1024;
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
    // (45 KB of dynamic LDS + the fit kernel's 7 KB of static keep three workgroups on a CU at every degree.  Up to kFitMaxLdsBytes the
    // incremental fits of degrees 6, 8, 9 and 11 stacked one or two cells more -- and a round launched for "one degree more than the
    // host knows" ran two workgroups a CU: 238 us instead of 194)
    while (gmax > 1 && frLds(degree, gmax, 1) > (weighted ? kFitMaxLdsBytes : kFrFitLdsCap)) --gmax;
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
// selection, remaining levels + exact order -> the batch; then the round's shape classes (and, inline, its lists)
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

// LDS of the 1024-thread workgroups (fr_batch_kernel, fr_round_kernel): 62 KB, reused phase by phase -- who lives where is said
// where it happens.
static_assert(sizeof(FitTask) == 56 && sizeof(hpsdf_node) == 56, "frStoreRecords moves records of seven 8-byte words");
struct FrLds {
    uint64_t key[kFrSort];  // 32 KB: sort keys / node bitmap / cost prefix / scans + per-job records / the running total's operands
    uint32_t val[kFrSort];  // 16 KB: the batch in node order
    uint32_t hist[2048];    // radix histograms; later the class tables of the inline emission
    uint32_t tmp[1024];
    uint32_t count[kFrClasses];
    uint32_t wave[16];
    uint32_t slice[9];
    int t;
    uint32_t above, c, next, flag, stuck, anySplit, early;
    unsigned long long need;
    uint32_t countO[8][kFrClasses];  // FrDev::replica: fits per (owner, class)
    unsigned long long rowsO[8];
    // per wave: 64 records of 7 words (a FitTask, a serialised node: 56 bytes) on their way to memory.  A lane writes its record's
    // words, the wave reads the 448 words back in order and stores THOSE: runs of whole records instead of 8 bytes every 56
    // (which cost the lists of a 1024-job round 60 k cycles, most of them waiting for the stores to drain)
    uint64_t stage[16][64 * 7];
};
// One wave: record `lane` (7 words, w[]) goes to dst64 + base (in 8-byte words) if valid; the records of the eight lanes of an
// octet are contiguous in memory when `octets` (base = the octet's first record: taken from its first lane).
__device__ __forceinline__ void frStoreRecords(uint64_t* st, uint64_t* dst64, const uint64_t (&w)[7], uint64_t base, bool valid, bool octets) {
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int k = 0; k < 7; ++k) st[lane * 7 + k] = w[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const long long b = valid ? (long long)base : -1ll;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const int q = r * 64 + lane;                 // word q of the wave's 448
        const int rec = q / 7, off = q - rec * 7;    // record and word within it
        const int src = octets ? (rec & ~7) : rec;   // the lane that knows where it goes
        const long long sb = __shfl(b, src, 64);
        const uint64_t v = st[q];
        if (sb >= 0) dst64[(uint64_t)sb + (uint64_t)(octets ? (rec & 7) * 7 + off : off)] = v;
    }
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t frLoad(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Hand-offs between workgroups (MI355X_MICROARCH.md, "Valid forms"): the producer's waves drain their stores at the workgroup's barrier,
// ONE lane releases (the XCD's L2 is written back once, not once a wave: sixteen waves fencing cost two to four times one lane's fence)
// and then signals with a relaxed atomic; the consumer polls relaxed, ONE lane acquires when the poll has matched, the workgroup's
// barrier holds the other waves until the invalidate has completed, then plain loads.  The waits behind the fences are written out: the
// compiler may drop the one after the write-back when it thinks the wave has nothing in flight.
__device__ __forceinline__ void frRelease() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void frAcquire() {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#define FR_STAMP(k) do { if (d.stamps && threadIdx.x == 0) d.hdr->dbg[k] = __builtin_readcyclecounter(); } while (0)

// where job j's 9 errors go in the round being prepared (frErrSlot's rule with the coming round's stride), and that round's status slot
__device__ __forceinline__ uint32_t frErrSlotNext(const FrDev& d, const uint32_t* slice, uint32_t owner, uint32_t j) {
    if (d.world == 1) return j * 9u;
    return owner * d.errStrideNext + (j - slice[owner]) * 9u;
}
__device__ __forceinline__ double* frStatusSlotNext(const FrDev& d, int r) { return d.errs + (size_t)r * d.errStrideNext + (d.errStrideNext - 1u); }

// From the selection's level 0 to the round's shape classes.  nQ: queued nodes; above / cand / taken: level 0's counts
// (nodes in exponent bins above the threshold bin -- already in d.taken --, candidates in d.candA, entries of d.taken).
// INLINE (fr_round_kernel's workgroup 0, which has just run level 0 itself: L.hist holds the level-1 digits of the
// candidates): the workgroup also writes the round's FitTask / FitBlock lists -- a job's slots are reserved while its fits
// are counted, the per-job facts wait in LDS, one lane per (job, fit) writes a task.  Otherwise (fr_batch_kernel, behind
// fr_select_kernel): the histogram sits in the header, the classes go to d.rnd and fr_tasks_kernel follows.
// watchStamp (INLINE, a round prepared on the assumption that the build goes on): the closing round's number + 1 -- once the batch is
// chosen the workgroup looks whether the running total has arrived meanwhile and says "stop" (FrHdr::chainStamp == watchStamp and
// rTotal < d.target); it then returns true with nothing of the header touched (what it wrote -- d.taken, the candidate lists -- is not
// read after the stop), and the rest of the preparation, more than half of it, is not spent on a round that will not be fitted.
template <bool INLINE>
__device__ bool frBatchBody(const FrDev& d, FrLds& L, uint32_t nQ, uint32_t above, uint32_t cand, uint32_t taken, uint32_t nNodes,
                            uint32_t watchStamp = 0) {
    FrHdr* h = d.hdr;
    FrRound* R = d.rnd;
    const uint32_t tid = threadIdx.x;
    uint64_t* sKey = L.key;
    uint32_t* sVal = L.val;
    uint32_t* sHist = L.hist;
    uint32_t* sTmp = L.tmp;
    uint32_t* sCount = L.count;
    FR_STAMP(0);
    const uint32_t nJobs = nQ < d.K ? nQ : d.K;
    uint32_t need = nJobs - above;  // still to come out of the candidates
    uint32_t C = cand;
    uint32_t* cur = d.candA;
    uint32_t* nxt = d.candB;
    int level = 1;
    if (!INLINE && C > kFrExact) {
        for (uint32_t i = tid; i < 2048; i += 1024) sHist[i] = h->hist2[i];
        __syncthreads();
    }
    uint32_t tk = taken;
    while (C > kFrExact && level <= 8) {  // refine by the digit of `level` (its histogram is in sHist)
        frThreshold(sHist, need, sTmp, &L.t, &L.above);
        const int T = L.t;
        const uint32_t abv = L.above;
        if (tid == 0) L.c = 0, L.next = tk;
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
                    d.taken[atomicAdd(&L.next, 1u)] = idx;
                } else if (dg == T) {
                    nxt[atomicAdd(&L.c, 1u)] = idx;
                    if (level < 8) atomicAdd(&sHist[frDigit(level + 1, bits, idx)], 1u);
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        need -= abv;
        C = L.c;
        tk = L.next;
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
    // position is its rank (trees beyond 262144 nodes: bitonic sort of the indices).  A thread owns 8 words, or one word
    // while the tree has at most 32768 nodes (eight times the threads then share the bit loop).
    FR_STAMP(2);
    if (nNodes <= 8192u * 32u) {
        uint32_t* bm = reinterpret_cast<uint32_t*>(sKey);  // 8192 words
        const uint32_t words = (nNodes + 31u) >> 5;
        const uint32_t per = words <= 1024u ? 1u : 8u;
        for (uint32_t w = tid; w < per * 1024u; w += 1024) bm[w] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < nJobs; i += 1024) {
            const uint32_t idx = d.taken[i];
            atomicOr(&bm[idx >> 5], 1u << (idx & 31u));
        }
        __syncthreads();
        uint32_t own = 0;
        for (uint32_t k = 0; k < per; ++k) own += (uint32_t)__popc(bm[tid * per + k]);
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
            for (uint32_t k = 0; k < per; ++k) {
                uint32_t bits = bm[tid * per + k];
                while (bits) {
                    const int bpos = __ffs((int)bits) - 1;
                    bits &= bits - 1u;
                    sVal[pos++] = (tid * per + k) * 32u + (uint32_t)bpos;
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
    if (INLINE && watchStamp != 0 && tid == 0) {
        uint32_t early = 0;
        if (frLoad(&h->chainStamp) == watchStamp) {
            frAcquire();
            early = *(volatile double*)&h->rTotal < d.target ? 1u : 0u;  // Octree.cpp:216
        }
        L.early = early;
    }
    for (uint32_t i = tid; i < 2048; i += 1024) sHist[i] = 0;
    for (uint32_t c = tid; c < (uint32_t)kFrClasses; c += 1024) sCount[c] = 0;
    if (d.replica)
        for (uint32_t c = tid; c < 8u * (uint32_t)kFrClasses; c += 1024) (&L.countO[0][0])[c] = 0;
    if (tid < 9) L.slice[tid] = tid ? nJobs : 0u;
    __syncthreads();
    if (INLINE && watchStamp != 0 && L.early) return true;
    // ---- the jobs leave the frontier; every job becomes 1 (coarse) or up to 9 cell fits: count them per shape class.
    //      Thread t owns jobs 4 t .. 4 t + 3.  With several ranks every rank counts only the fits of its own slice: the
    //      slices are contiguous job ranges of (nearly) equal cost, cut where the host scheduler cuts them (builderSelect).
    uint64_t* sCost = sKey;  // (the bitmap is dead) inclusive prefix of the jobs' costs
    // (INLINE) per-job records for fr_emit_kernel: A = H slot | P slot << 16 | degree << 28; B = coarse | ours << 1 | owner << 2 | depth << 5
    uint32_t* sJobA = d.jobRecA;
    uint16_t* sJobB = d.jobRecB;
    // (replica) per job the lists its fits belong to, owner x class (0xFFFF: none): words 2048.. of sKey, behind the scans' 2048
    uint16_t* sKeyH = reinterpret_cast<uint16_t*>(reinterpret_cast<uint32_t*>(sKey) + 2048);
    uint16_t* sKeyP = sKeyH + kFrSort;
    {
        int jP[4], jDep[4];
        bool jCoarse[4];
        uint64_t own = 0;
        // (every job's error and node record asked for before anything is stored: with the stores in between the four jobs' loads
        // went out one after the other -- the compiler must assume that a store to qErr changes the next job's load -- 14 k cycles)
        uint32_t jIdx[4];
        uint64_t jBits[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = tid * 4u + (uint32_t)q;
            jP[q] = 0, jDep[q] = 0, jCoarse[q] = false, jIdx[q] = 0, jBits[q] = 0;
            if (j < nJobs) jIdx[q] = sVal[j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = tid * 4u + (uint32_t)q;
            if (j >= nJobs) continue;
            jBits[q] = d.qErr[jIdx[q]];
            const hpsdf_node& n = d.nodes[jIdx[q]];
            jP[q] = n.degree, jDep[q] = n.depth;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = tid * 4u + (uint32_t)q;
            if (j >= nJobs) continue;
            const uint32_t idx = jIdx[q];
            const uint64_t bits = jBits[q];
            const double e = __longlong_as_double((long long)bits);
            d.wBatchIdx[j] = idx;
            d.wBatchErr[j] = e;
            d.qErr[idx] = kNotQueued;
            atomicAdd(&sHist[frDigit(0, bits, idx)], 1u);
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
                L.slice[0] = 0;
                for (int r = 0; r < d.world; ++r) {
                    uint32_t end = nJobs;
                    if (r + 1 < d.world) {
                        end = sTmp[64 + r];
                        end = end < start ? start : end;
                        end = end > nJobs ? nJobs : end;
                    }
                    h->sliceFirst[r + 1] = end;
                    L.slice[r + 1] = end;
                    start = end;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = tid * 4u + (uint32_t)q;
            if (j >= nJobs) continue;
            uint32_t o = 0;
            if (d.world > 1) {
                for (int r = 1; r < d.world; ++r) o += L.slice[r] <= j ? 1u : 0u;
                d.jobOwner[j] = (uint8_t)o;
            }
            const bool ours = d.world == 1 || (int)o == d.rank;  // another rank's job: none of its fits here
            const int p = jP[q], dep = jDep[q];
            uint32_t slotH = 0, slotP = 0;
            if (d.replica) {
                // every job takes its place in its OWNER's lists, and the same place on every rank -- the arena is laid out alike
                // everywhere -- so not by an atomic's luck: the keys wait in LDS and one wave hands the slots out in job order (below)
                const uint32_t none = 0xFFFFu, base = o * (uint32_t)kFrClasses;
                sKeyH[j] = (uint16_t)((!jCoarse[q] && dep < kMaxDepth) ? base + (uint32_t)frClass(p, false, dep + 1) : none);
                sKeyP[j] = (uint16_t)(jCoarse[q] ? base + (uint32_t)frClass(2, false, dep) : (p < kMaxDegree - 1 ? base + (uint32_t)frClass(p + 1, true, dep) : none));
            } else if (ours) {
                if (jCoarse[q]) {
                    slotP = atomicAdd(&sCount[frClass(2, false, dep)], 1u);  // :836-843
                } else {
                    if (dep < kMaxDepth) slotH = atomicAdd(&sCount[frClass(p, false, dep + 1)], 8u);     // :814-822
                    if (p < kMaxDegree - 1) slotP = atomicAdd(&sCount[frClass(p + 1, true, dep)], 1u);  // :846-851
                }
            }
            if (INLINE || d.replica) {
                sJobA[j] = slotH | (slotP << 16) | ((uint32_t)p << 28);
                sJobB[j] = (uint16_t)((jCoarse[q] ? 1u : 0u) | (ours ? 2u : 0u) | (o << 2) | ((uint32_t)dep << 5));
            }
        }
    }
    __syncthreads();
    if (d.replica && tid < 64) {
        // 64 jobs at a time, in job order: the lanes that share a key take consecutive slots behind the key's running count
        uint32_t* flat = &L.countO[0][0];
        for (uint32_t j0 = 0; j0 < nJobs; j0 += 64u) {
            const uint32_t j = j0 + tid;
            uint32_t slots[2] = {0u, 0u};
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const uint32_t key = j < nJobs ? (which == 0 ? sKeyH[j] : sKeyP[j]) : 0xFFFFu, step = which == 0 ? 8u : 1u;
                bool pending = key != 0xFFFFu;
                unsigned long long left = __ballot(pending);
                while (left) {
                    const uint32_t k0 = __shfl(key, __ffsll((long long)left) - 1, 64);
                    const unsigned long long m = __ballot(pending && key == k0);
                    const uint32_t before = flat[k0];
                    if (pending && key == k0) slots[which] = before + step * (uint32_t)__popcll(m & ((1ull << tid) - 1ull));
                    __builtin_amdgcn_wave_barrier();
                    if (tid == (uint32_t)(__ffsll((long long)m) - 1)) flat[k0] = before + step * (uint32_t)__popcll(m);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    pending = pending && key != k0;
                    left = __ballot(pending);
                }
            }
            if (j < nJobs) sJobA[j] = (sJobA[j] & 0xF0000000u) | slots[0] | (slots[1] << 16);
        }
    }
    __syncthreads();
    if (d.replica) {  // this rank's own fits: its lists in the owner table
        for (uint32_t c = tid; c < (uint32_t)kFrClasses; c += 1024) sCount[c] = L.countO[d.rank][c];
        __syncthreads();
    }
    FR_STAMP(4);
    for (uint32_t i = tid; i < 2048; i += 1024)
        if (sHist[i]) h->hist1[i] -= sHist[i];
    // ---- shapes, then prefix sums over the classes (degree-major): tasks, workgroups, arena rows, samples.
    //      Thread c owns class c; the scan runs over 512 slots in LDS (the first 2048 words of sKey hold the two 64-bit scans,
    //      the halves of sTmp the task and workgroup counts).
    uint32_t myCount = 0, myBlocks = 0;
    uint64_t myRows = 0, mySamples = 0;
    int g = 1, pl = 1;
    // Does this round split its from-scratch fits (HPSDF_FIT_SPLIT)?  The host scheduler's rule (builderCompute): some class of the
    // round can be split, and the round's samples -- all of them: the buffer is addressed by FitTask::sampleOff -- number at most 2^31
    // and fit the buffer the host could get.
    if (tid == 0) L.need = 0, L.anySplit = 0;
    __syncthreads();
    if (tid < (uint32_t)kFrClasses && sCount[tid]) {
        const int deg = (int)tid / kFrDepths / 2;
        const bool incr = ((int)tid / kFrDepths) & 1;
        const uint64_t nq = 4 * (uint64_t)deg + 1;
        atomicAdd(&L.need, (unsigned long long)(nq * nq * nq * sCount[tid]));
        if (frSplit(d.splitFit, deg, incr)) L.anySplit = 1;
    }
    __syncthreads();
    const int splitFit = (L.anySplit && L.need <= (1ull << 31) && L.need <= d.sampleCap) ? d.splitFit : 0;
    if (tid < (uint32_t)kFrClasses) {
        myCount = sCount[tid];
        if (myCount) {
            const int deg = (int)tid / kFrDepths / 2;
            const bool incr = ((int)tid / kFrDepths) & 1;
            frShape(deg, incr, myCount, &g, &pl, d.fastFit != 0, d.weighted != 0, splitFit);
            myBlocks = (myCount + (uint32_t)g - 1u) / (uint32_t)g;
            const uint64_t nq = 4 * (uint64_t)deg + 1;
            // (a weighted fit owns a full array: the incremental one too)
            myRows = (uint64_t)((incr && !d.weighted) ? frCoef(deg) - frCoef(deg - 1) : frCoef(deg)) * myCount;
            mySamples = nq * nq * nq * myCount;
        }
    }
    __syncthreads();
    FR_STAMP(5);
    uint64_t* sRows = sKey;
    uint64_t* sSmp = sKey + 512;
    uint32_t* sCnt = sTmp;
    uint32_t* sBlk = sTmp + 512;
    {
        // inclusive scans of the four per-class figures over the first 512 threads: inside a wave by shuffles, the eight wave totals
        // through LDS (two barriers; a Hillis-Steele scan over LDS was nine steps of two)
        uint32_t a = myCount, b = myBlocks;
        uint64_t r = myRows, sm = mySamples;
        const int lane = (int)(tid & 63);
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t a2 = __shfl_up(a, off, 64), b2 = __shfl_up(b, off, 64);
            const uint64_t r2 = (uint64_t)__shfl_up((long long)r, off, 64), s2 = (uint64_t)__shfl_up((long long)sm, off, 64);
            if (lane >= off) a += a2, b += b2, r += r2, sm += s2;
        }
        uint32_t* wCnt = L.wave;                                   // [8] + [8]
        uint64_t* wRow = reinterpret_cast<uint64_t*>(sKey + 1024);  // [8] + [8] (behind the scans' 1024 slots)
        if (tid < 512 && lane == 63) wCnt[tid >> 6] = a, wCnt[8 + (tid >> 6)] = b, wRow[tid >> 6] = r, wRow[8 + (tid >> 6)] = sm;
        __syncthreads();
        if (tid < 512) {
            for (uint32_t w = 0; w < (tid >> 6); ++w) a += wCnt[w], b += wCnt[8 + w], r += wRow[w], sm += wRow[8 + w];
            sCnt[tid] = a, sBlk[tid] = b, sRows[tid] = r, sSmp[tid] = sm;
        }
        __syncthreads();
    }
    FR_STAMP(6);
    const uint32_t nTasks = sCnt[kFrClasses - 1], nBlocks = sBlk[kFrClasses - 1];
    const uint64_t rowsAll = sRows[kFrClasses - 1], smpAll = sSmp[kFrClasses - 1];
    const uint64_t arenaBase = *(volatile uint64_t*)&h->arenaUsed;
    uint32_t exCnt = 0, exBlk = 0;
    uint64_t exRows = 0, exSmp = 0;
    if (tid < (uint32_t)kFrClasses) exCnt = sCnt[tid] - myCount, exBlk = sBlk[tid] - myBlocks, exRows = sRows[tid] - myRows, exSmp = sSmp[tid] - mySamples;
    if (tid < (uint32_t)kFrClasses) {
        R->cCount[tid] = myCount;
        R->cFirst[tid] = exCnt;
        R->cCursor[tid] = 0;
        R->cBlockFirst[tid] = exBlk;
        R->cBlocks[tid] = myBlocks;
        R->cArena[tid] = exRows;
        R->cSample[tid] = exSmp;
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
        const uint32_t low = frSplit(splitFit, (int)tid, false) ? sCnt[lo + kFrDepths - 1] - t0 : 0u;
        h->lowTasks[tid][1] = low;
        if (low) atomicAdd(reinterpret_cast<unsigned long long*>(&h->splitFits), (unsigned long long)low);
    }
    if (d.replica && tid < (uint32_t)d.world) {  // owner tid's part of the round: its classes one after the other (a weighted fit owns a full array)
        unsigned long long run = 0;
        for (int c = 0; c < kFrClasses; ++c) {
            R->cArenaO[tid][c] = run;
            run += (unsigned long long)L.countO[tid][c] * frCoef(c / kFrDepths / 2);
        }
        L.rowsO[tid] = run;
    }
    __syncthreads();  // (every thread has read the header's arenaUsed)
    if (tid == 0) {
        h->nJobs = nJobs, h->nTasks = nTasks, h->nBlocks = nBlocks;
        h->splitRound = (uint32_t)splitFit;
        h->sampleUsed = smpAll;
        h->fits += nTasks, h->samples += smpAll;
        R->arenaBase = arenaBase;
        if (d.replica) {
            unsigned long long stride = 16;
            for (int r = 0; r < d.world; ++r) stride = L.rowsO[r] > stride ? L.rowsO[r] : stride;
            stride = (stride + 15ull) & ~15ull;  // equal parts, whole lines
            h->roundBase = arenaBase, h->partStride = stride;
            h->arenaUsed = arenaBase + (uint64_t)d.world * stride;
        } else
        h->arenaUsed = arenaBase + rowsAll;
        if (!INLINE && d.world > 1) *frStatusSlotNext(d, d.rank) = 0.0;  // (inline: the leader, once nobody reads the closing round's errors any more)
    }
    FR_STAMP(7);
    FR_STAMP(8);
    return false;
}

// The round's FitTask / FitBlock lists behind the inline batch (fr_round_kernel's leader): the shape classes are in d.rnd, every job's
// slots and facts in d.jobRecA / B, the batch in d.wBatchIdx.  Workgroup w takes jobs 128 w .. 128 w + 127: one lane per fit, the
// records through the wave's staging buffer (frStoreRecords) -- first the from-scratch fits of the children (EstimateHImprovement,
// :814-822; lane 8 j + k is child k of job j, a job's eight tasks are contiguous), then every job's own fit (the coarse degree-2 fit
// :836-843 or the incremental one :846-851) and where the job's results will lie in the arena; then its share of the workgroup records.
// (The leader wrote these lists itself at first: 60 k cycles of one CU for a 1024-job round.)
struct FrEmitLds {
    uint32_t tabTask[kFrClasses], tabBlk[kFrClasses], tabCount[kFrClasses], tabShape[kFrClasses];
    uint64_t tabArena[kFrClasses], tabSample[kFrClasses];
    uint32_t slice[9];
    uint64_t stage[16][64 * 7];
    uint64_t tabArenaO[8][kFrClasses];  // FrDev::replica
};
__global__ __launch_bounds__(1024) void fr_emit_kernel(FrDev d) {
    const FrHdr* h = d.hdr;
    if (h->done) return;
    const uint32_t nJobs = h->nJobs, nBlocks = h->nBlocks, tid = threadIdx.x;
    const FrRound* R = d.rnd;
    __shared__ FrEmitLds E;
    if (tid < (uint32_t)kFrClasses) {
        E.tabTask[tid] = R->cFirst[tid], E.tabBlk[tid] = R->cBlockFirst[tid], E.tabCount[tid] = R->cCount[tid];
        E.tabShape[tid] = (uint32_t)R->cG[tid] | ((uint32_t)R->cPlanes[tid] << 8);
        E.tabArena[tid] = R->cArena[tid], E.tabSample[tid] = R->cSample[tid];
    }
    if (tid < 9) E.slice[tid] = h->sliceFirst[tid];
    if (d.replica)
        for (uint32_t c = tid; c < 8u * (uint32_t)kFrClasses; c += 1024) (&E.tabArenaO[0][0])[c] = (&R->cArenaO[0][0])[c];
    const uint64_t arenaBase = R->arenaBase, partStride = h->partStride;
    // rows before slot 0 of class c in the round's part of the arena (replica: within owner o's part)
    auto arenaOf = [&](uint32_t o, int c) { return d.replica ? arenaBase + (uint64_t)o * partStride + E.tabArenaO[o][c] : arenaBase + E.tabArena[c]; };
    const int wv = (int)(tid >> 6);
    uint64_t* tasks64 = reinterpret_cast<uint64_t*>(d.tasks);
    auto word2 = [](float lo, float hi) { return (uint64_t)__float_as_uint(lo) | ((uint64_t)__float_as_uint(hi) << 32); };
    // this lane's child fit: job and node first (independent of the tables)
    const uint32_t jH = blockIdx.x * 128u + (tid >> 3), jP = blockIdx.x * 128u + tid;
    const int k = (int)(tid & 7u);
    const bool liveH = jH < nJobs, liveP = tid < 128u && jP < nJobs;
    float bn[3] = {0, 0, 0}, bx[3] = {0, 0, 0}, pn[3] = {0, 0, 0}, px[3] = {0, 0, 0};
    uint32_t ja = 0, jb = 0, pa = 0, pb = 0, pIdx = 0;
    if (liveH) {
        const hpsdf_node& n = d.nodes[d.wBatchIdx[jH]];
        for (int a = 0; a < 3; ++a) bn[a] = n.aabb_min[a], bx[a] = n.aabb_max[a];
        ja = d.jobRecA[jH], jb = d.jobRecB[jH];
    }
    if (liveP) {
        pIdx = d.wBatchIdx[jP];
        const hpsdf_node& n = d.nodes[pIdx];
        for (int a = 0; a < 3; ++a) pn[a] = n.aabb_min[a], px[a] = n.aabb_max[a];
        pa = d.jobRecA[jP], pb = d.jobRecB[jP];
    }
    __syncthreads();
    if (blockIdx.x * 128u < nJobs) {
        {
            const int p = (int)(ja >> 28), dep = (int)(jb >> 5);
            const bool hasH = liveH && (jb & 2u) != 0 && (jb & 1u) == 0 && dep < kMaxDepth;
            const int c = frClass(p, false, dep + 1 <= kMaxDepth ? dep + 1 : kMaxDepth);
            const uint32_t slot0 = ja & 0xFFFFu;
            const uint64_t rows = frCoef(p), nq = 4 * (uint64_t)p + 1;
            float mn[3], mx[3];
            for (int a = 0; a < 3; ++a) {  // Octree::CornerAABB, :1096-1112
                const float mid = (bx[a] + bn[a]) * 0.5f;
                mn[a] = (k >> a) & 1 ? mid : bn[a];
                mx[a] = (k >> a) & 1 ? bx[a] : mid;
            }
            uint64_t w[7];
            w[0] = word2(mn[0], mn[1]), w[1] = word2(mn[2], mx[0]), w[2] = word2(mx[1], mx[2]);
            w[3] = arenaOf((jb >> 2) & 7u, c) + (uint64_t)(slot0 + (uint32_t)k) * rows;  // outOff
            w[4] = ~0ull;                                                                // copyOff
            w[5] = E.tabSample[c] + (uint64_t)(slot0 + (uint32_t)k) * nq * nq * nq;      // sampleOff
            w[6] = (uint64_t)(frErrSlotNext(d, E.slice, (jb >> 2) & 7u, liveH ? jH : 0u) + 1u + (uint32_t)k) | ((uint64_t)(dep + 1) << 32) | ((uint64_t)p << 40);
            frStoreRecords(E.stage[wv], tasks64, w, ((uint64_t)E.tabTask[c] + slot0) * 7u, hasH, true);
        }
        if (tid < 128u) {  // (waves 0 and 1)
            const int p = (int)(pa >> 28), dep = (int)(pb >> 5);
            const bool coarse = (pb & 1u) != 0, ours = (pb & 2u) != 0;
            const bool hasP = liveP && ours && (coarse || p < kMaxDegree - 1);
            // (replica: where ANY rank's results will lie -- the update notes them for every job)
            const bool anyH = liveP && (ours || d.replica) && !coarse && dep < kMaxDepth, anyP = liveP && (ours || d.replica) && (coarse || p < kMaxDegree - 1);
            const uint32_t owner = (pb >> 2) & 7u;
            const int deg = coarse ? 2 : (p + 1 <= kMaxDegree ? p + 1 : kMaxDegree);
            const bool incr = !coarse;
            const int c = frClass(deg, incr, dep), cH = frClass(p, false, dep + 1 <= kMaxDepth ? dep + 1 : kMaxDepth);
            const uint32_t slot = (pa >> 16) & 4095u;
            const uint64_t rows = (incr && !d.weighted) ? frCoef(deg) - frCoef(deg - 1) : frCoef(deg), nq = 4 * (uint64_t)deg + 1;
            const uint64_t outP = arenaOf(owner, c) + (uint64_t)slot * rows;
            uint64_t w[7];
            w[0] = word2(pn[0], pn[1]), w[1] = word2(pn[2], px[0]), w[2] = word2(px[1], px[2]);
            w[3] = outP;
            // weighted incremental fit: the cell's current array (one segment: the update keeps it that way), :847
            w[4] = (hasP && d.weighted && incr) ? (d.segOff[(size_t)pIdx * kFrSegs] & kOffMask) : ~0ull;
            w[5] = E.tabSample[c] + (uint64_t)slot * nq * nq * nq;
            w[6] = (uint64_t)frErrSlotNext(d, E.slice, (pb >> 2) & 7u, liveP ? jP : 0u) | ((uint64_t)dep << 32) | ((uint64_t)deg << 40);
            frStoreRecords(E.stage[wv], tasks64, w, ((uint64_t)E.tabTask[c] + slot) * 7u, hasP, false);
            if (liveP) {
                d.wJobP[jP] = anyP ? outP : ~0ull;
                d.wJobH[jP] = anyH ? arenaOf(owner, cH) + (uint64_t)(pa & 0xFFFFu) * frCoef(p) : ~0ull;
            }
        }
    }
    for (uint32_t b = blockIdx.x * 1024u + tid; b < nBlocks; b += gridDim.x * 1024u) {
        int lo = 0, hi = kFrClasses;  // the last class whose first workgroup is <= b is the one that owns b
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (E.tabBlk[mid] <= b)
                lo = mid;
            else
                hi = mid;
        }
        const int c = lo;
        const int deg = c / kFrDepths / 2;
        const bool incr = (c / kFrDepths) & 1;
        const uint32_t gg = E.tabShape[c] & 255u, local = b - E.tabBlk[c];
        FitBlock fb;
        fb.firstTask = E.tabTask[c] + local * gg;
        const uint32_t left = E.tabCount[c] - local * gg;
        fb.nTasks = (uint16_t)(left < gg ? left : gg);
        fb.degree = (uint8_t)deg;
        fb.planesPerChunk = (uint8_t)(E.tabShape[c] >> 8);
        const bool split = frSplit((int)h->splitRound, deg, incr);
        fb.rowStart = (uint16_t)((incr || split) ? frCoef(deg - 1) : 0);
        fb.rowEnd = (uint16_t)frCoef(deg);
        fb.depth = (uint8_t)(c % kFrDepths);
        fb.weighted = d.weighted ? 1 : 0;
        fb.split = split ? 1 : 0;
        fb.pad1[0] = 0;
        d.blocks[b] = fb;
    }
}

__global__ __launch_bounds__(1024) void fr_batch_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    const uint32_t tid = threadIdx.x;
    if (h->done) {
        if (tid < 13) h->degBlocks[tid][0] = h->degBlocks[tid][1] = h->degTasks[tid][0] = h->degTasks[tid][1] = 0;
        if (tid == 0) h->nJobs = 0, h->nTasks = 0, h->nBlocks = 0;
        return;
    }
    __shared__ FrLds L;
    frBatchBody<false>(d, L, h->nQueued, h->above, h->candCount, h->takenCount, h->nNodes);
}

// the round's FitTask / FitBlock lists, grouped by shape class (behind fr_batch_kernel).  Grid-wide: sixteen lanes per job -- lane
// k < 8 writes the from-scratch fit of child k (EstimateHImprovement, :814-822), lane 8 the job's own fit (the coarse degree-2 fit
// :836-843 or the incremental one :846-851); lanes 0 and 8 reserve the slots of their class -- then one lane per workgroup record.
__global__ __launch_bounds__(256) void fr_tasks_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (h->done) return;
    FrRound* R = d.rnd;
    const uint32_t nJobs = h->nJobs, nBlocks = h->nBlocks;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
    const uint64_t arenaBase = R->arenaBase;
    const int lane = threadIdx.x & 63;
    for (uint32_t base = (gid >> 4) - ((uint32_t)lane >> 4); base < nJobs; base += stride >> 4) {  // 4 jobs per wave, wave-uniform trip count
        const uint32_t j = base + ((uint32_t)lane >> 4);
        const int k = lane & 15;
        const bool live = j < nJobs;
        const hpsdf_node& n = d.nodes[d.batchIdx[live ? j : 0]];
        const int p = n.degree, dep = n.depth;
        const bool coarse = fabs(d.batchErr[live ? j : 0] - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
        const uint32_t owner = d.world == 1 ? 0u : d.jobOwner[live ? j : 0];
        const bool ours = live && (d.world == 1 || (int)owner == d.rank);
        const bool hasH = ours && !coarse && dep < kMaxDepth, hasP = ours && (coarse || p < kMaxDegree - 1);
        // this lane's fit, if any
        const bool mine = (k < 8 && hasH) || (k == 8 && hasP);
        const int deg = k < 8 ? p : (coarse ? 2 : p + 1);
        const bool incr = k == 8 && !coarse;
        const int depth = k < 8 ? dep + 1 : dep;
        const int c = frClass(deg, incr, depth);
        uint32_t slot = 0;
        if (d.replica) {  // (the batch has noted every job's slots in its owner's lists: d.jobRecA)
            const uint32_t ja = d.jobRecA[live ? j : 0];
            slot = k < 8 ? (ja & 0xFFFFu) + (uint32_t)k : (ja >> 16) & 4095u;
        } else {
            if (mine && (k == 0 || k == 8)) slot = atomicAdd(&R->cCursor[c], k == 0 ? 8u : 1u);
            slot = __shfl(slot, (lane & ~15) | (k < 8 ? 0 : 8), 64) + (k < 8 ? (uint32_t)k : 0u);
        }
        uint64_t outOff = ~0ull;
        if (d.replica && live && k <= 8 && (k < 8 ? (!coarse && dep < kMaxDepth) : (coarse || p < kMaxDegree - 1)))
            outOff = arenaBase + (uint64_t)owner * h->partStride + R->cArenaO[owner][c] + (uint64_t)slot * frCoef(deg);
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
            if (!d.replica) outOff = arenaBase + R->cArena[c] + (uint64_t)slot * rows;
            t.outOff = outOff;
            // weighted incremental fit: the cell's current array (one segment: the update keeps it that way), :847
            t.copyOff = (d.weighted && incr) ? (d.segOff[(size_t)d.batchIdx[j] * kFrSegs] & kOffMask) : ~0ull;
            t.sampleOff = R->cSample[c] + (uint64_t)slot * nq * nq * nq;
            t.errSlot = frErrSlotNext(d, h->sliceFirst, owner, j) + (k < 8 ? 1u + (uint32_t)k : 0u);
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
        const bool split = frSplit((int)h->splitRound, deg, incr);
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
// fr_round_kernel: closes a round (Octree.cpp:243-299 in node-index order, :594-601 per job, :216) and opens the next
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kFrInlineNodes = 131072;  // trees up to here are selected by the closing workgroup itself (128 passes of 1024 lanes; its bitmap
                                             // holds 262 144 nodes.  65 536 sent union3 @ 4.5e-9, K = 4096 -- 58 k nodes + 32 k of room for a round -- to the
                                             // grid selection in its last rounds: 5.8 ms against 4.5)
constexpr unsigned long long kFrWaitTicks = 200000000ull;  // two seconds of the 100 MHz clock: a wait this long means a workgroup is gone

// one lane waits until *p >= want, then acquires (the counter's writers release before they add); false: the wait ran out
__device__ bool frWaitAtLeast(const uint32_t* p, uint32_t want) {
    const unsigned long long t0 = wall_clock64();
    while (frLoad(p) < want) {
        for (int k = 0; k < 64 && frLoad(p) < want; ++k) __builtin_amdgcn_s_sleep(1);
        if (frLoad(p) < want && wall_clock64() - t0 > kFrWaitTicks) return false;
    }
    frAcquire();
    return true;
}

// Workgroups 1 .. gridDim.x - 2: 128 jobs each, eight lanes per job.
__device__ void frUpdateJobs(const FrDev& d, FrLds& L) {
    FrHdr* h = d.hdr;
    const uint32_t tid = threadIdx.x, w = blockIdx.x - 1u;
    const int lane = (int)(tid & 63), sub = (int)(tid & 7), wave = (int)(tid >> 6), b0 = lane & ~7;
    const uint32_t j = w * 128u + (tid >> 3);
    // (what depends on the job number alone is asked for together with the header: the batch arrays hold kFrJobs entries whatever the round's size)
    const uint32_t nJobs = h->nJobs, nNodes0 = h->nNodes, stamp = h->round + 1u;
    uint32_t idx = d.batchIdx[j];
    const double err = d.batchErr[j];
    const uint32_t owner = d.world == 1 ? 0u : d.jobOwner[j];
    const bool round0 = stamp == 1u;
    if (w * 128u >= nJobs) {  // no jobs here (the grid is sized for K): arrive all the same -- the leader rewrites the header only when
        if (tid == 0) {       // every workgroup of the launch has read it
            if (!round0) atomicAdd(&h->opsArrive, 1u);
            atomicAdd(&h->arrive, 1u);
        }
        return;
    }
    const bool live = j < nJobs;
    for (uint32_t i = tid; i < 2048; i += 1024) L.hist[i] = 0;
    if (tid == 0) L.c = 0, L.next = 0, L.stuck = 0;  // largest new degree, coefficients gained (as an int)
    double e = 0.0, e0 = 0.0;
    int p = 0, dep = 0;
    bool coarse = false;
    float bmn[3] = {0, 0, 0}, bmx[3] = {0, 0, 0};
    if (live) {
        const hpsdf_node& n = d.nodes[idx];
        const size_t slot = frErrSlot(d, h, j);
        coarse = fabs(err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
        const double ek = d.errs[slot + 1 + sub], ep = d.errs[slot];  // (nine doubles a job, whether or not a fit wrote them)
        p = n.degree, dep = n.depth;
        for (int a = 0; a < 3; ++a) bmn[a] = n.aabb_min[a], bmx[a] = n.aabb_max[a];
        const bool hasH = !coarse && dep < kMaxDepth, hasP = coarse || p < kMaxDegree - 1;
        if (hasH) e = ek;
        if (hasP && sub == 0) e0 = ep;
    } else {
        idx = 0;
    }
    __syncthreads();
    // ---- the decision, Octree.cpp:594-601, on every lane of the job alike
    const double pErr = __shfl(e0, b0, 64);
    double maxNewErr = 0.0;
    for (int i = 0; i < 8; ++i) {
        const double v = __shfl(e, b0 + i, 64);
        maxNewErr = maxNewErr < v ? v : maxNewErr;  // std::max
    }
    double pImp, hImp;
    if (coarse) {
        hImp = 0.0;   // :806-810
        pImp = pErr;  // :842
    } else {
        hImp = dep < kMaxDepth ? (1.0 / (7.0 * (double)frCoef(p))) * (err - 8.0 * maxNewErr) : 0.0;           // :825
        pImp = p < kMaxDegree - 1 ? (1.0 / (double)(frCoef(p + 1) - frCoef(p))) * (err - 8.0 * pErr) : 0.0;  // :854
    }
    bool refineP = p < (kMaxDegree - 1) && (dep == kMaxDepth || pImp > hImp);  // :600
    if (coarse) refineP = true;
    const bool refineH = dep < kMaxDepth && !refineP;  // :601
    const int kind = !live ? 0 : (refineP ? 1 : (refineH ? 2 : 0));
    const int np = coarse ? 2 : p + 1;
    int delta = 0;
    if (kind == 1)
        delta = (int)frCoef(np) - (int)(round0 ? frCoef(2) : frCoef(p));  // (the header's count already stands for "every cell at degree 2")
    else if (kind == 2)
        delta = 7 * (int)frCoef(p);
    // ---- splitting / P jobs before this one: in the wave by ballots, in the workgroup over the 16 wave counts, in the round
    //      over the counts the workgroups before this one have published
    const bool lead = live && sub == 0;
    const unsigned long long mH = __ballot(lead && kind == 2), mP = __ballot(lead && kind == 1);
    const unsigned long long below = (1ull << b0) - 1ull;
    if (lane == 0) L.wave[wave] = ((uint32_t)__popcll(mH) << 16) | (uint32_t)__popcll(mP);
    if (lead && kind == 1) atomicMax(&L.c, (uint32_t)np);
    if (lead && delta) atomicAdd(reinterpret_cast<int*>(&L.next), delta);
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int ww = 0; ww < 16; ++ww) {
        const uint32_t v = L.wave[ww];
        if (ww < wave) before += v;
        all += v;
    }
    if (tid == 0) {
        __hip_atomic_store(&h->agg[w], ((uint64_t)stamp << 32) | all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the round's counts (the last workgroup sizes the operand list from them: ahead of the operands' "written" signal)
        const uint32_t mine = nJobs - w * 128u < 128u ? nJobs - w * 128u : 128u, nP = all & 0xFFFFu, nH = all >> 16;
        if (nP) atomicAdd(&h->rP, nP);
        if (nH) atomicAdd(&h->rH, nH);
        if (mine - nP - nH) atomicAdd(&h->rD, mine - nP - nH);
        if (L.c) atomicMax(&h->rMaxDeg, L.c);
        const int dl = *reinterpret_cast<int*>(&L.next);
        if (dl) atomicAdd(reinterpret_cast<unsigned long long*>(&h->rCoeffDelta), (unsigned long long)(long long)dl);
    }
    if (tid < 64) {
        uint32_t v = 0;
        if (tid < w) {
            uint64_t a = __hip_atomic_load(&h->agg[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(a >> 32) != stamp) {
                const unsigned long long t0 = wall_clock64();
                do {
                    __builtin_amdgcn_s_sleep(1);
                    a = __hip_atomic_load(&h->agg[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((uint32_t)(a >> 32) != stamp && wall_clock64() - t0 < kFrWaitTicks);
                if ((uint32_t)(a >> 32) != stamp) L.stuck = 1, a = 0;
            }
            v = (uint32_t)a;
        }
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (tid == 0) L.above = v;
    }
    __syncthreads();
    const bool stuck = L.stuck != 0;
    const uint32_t run = L.above + before;
    const uint32_t hBefore = (run >> 16) + (uint32_t)__popcll(mH & below), pBefore = (run & 0xFFFFu) + (uint32_t)__popcll(mP & below);
    const uint32_t c0 = nNodes0 + 8u * hBefore;          // first child of a splitting job
    const bool full = kind == 2 && c0 + 8u > d.nodeCap;  // cannot happen: the host sizes for 8 K new nodes
    if (full && sub == 0) atomicExch(&h->overflow, 1u);
    if (stuck && tid == 0) atomicExch(&h->stuck, 1u);
    const bool apply = live && !full && !stuck;
    // ---- the job's additions to the running total, densely, in job order: a P result adds (newErr - initialErr) (:255); an H
    //      result subtracts initialErr once (:268) and adds its 8 children's errors (:272).  (Round 0: the last workgroup reads
    //      the errors themselves.)
    if (!round0 && apply) {
        double* o = d.ops + (pBefore + 9u * hBefore);
        if (kind == 1) {
            if (sub == 0) o[0] = pErr - err;
        } else if (kind == 2) {
            if (sub == 0) o[0] = err * -1.0;  // total -= err  ==  total + (-err)
            o[1 + sub] = e;
        }
    }
    // ---- tree and queue.  (The coefficient counts of the subtrees, which ReallocCoeffs needs, are added up once, when the build
    //      has stopped -- fr_subtree_kernel; walking up every job's ancestors here was a chain of dependent atomics a round.)
    if (apply && kind == 1) {  // :253-260, :286-290
        if (sub == 0) {
            // (weighted: the new array holds every row, so it is the node's one and only segment)
            const int first = (coarse || d.weighted) ? np : (int)d.segFirst[idx];
            if (coarse || d.weighted) d.segFirst[idx] = (uint8_t)np;
            d.segOff[(size_t)idx * kFrSegs + (np - first)] = (d.jobP[j] & kOffMask) | ((uint64_t)(d.replica ? 0u : owner) << 56);
            d.nodes[idx].degree = (uint8_t)np;
            const uint64_t bits = (uint64_t)__double_as_longlong(pErr);
            d.qErr[idx] = bits;
            atomicAdd(&L.hist[frDigit(0, bits, idx)], 1u);
        }
    }
    {   // :262-279, :286-290; Octree::Subdivide :1115-1128 -- the eight children of a splitting job are eight consecutive nodes:
        // lane `sub` makes child `sub`, the wave stores its 64 records in order (frStoreRecords)
        const bool split = apply && kind == 2;
        const uint32_t ch = c0 + (uint32_t)sub;
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) {  // CornerAABB
            const float mid = (bmx[a] + bmn[a]) * 0.5f;
            mn[a] = (sub >> a) & 1 ? mid : bmn[a];
            mx[a] = (sub >> a) & 1 ? bmx[a] : mid;
        }
        auto word2 = [](float lo, float hi) { return (uint64_t)__float_as_uint(lo) | ((uint64_t)__float_as_uint(hi) << 32); };
        uint64_t wd[7];
        wd[0] = ~0ull;  // child_idx: a leaf
        wd[1] = word2(mn[0], mn[1]), wd[2] = word2(mn[2], mx[0]), wd[3] = word2(mx[1], mx[2]);
        wd[4] = 0;                        // coeffs_start
        wd[5] = (uint64_t)(uint32_t)p;    // degree, pad0
        wd[6] = (uint64_t)(uint32_t)(dep + 1);  // depth, pad1
        frStoreRecords(L.stage[wave], reinterpret_cast<uint64_t*>(d.nodes), wd, (uint64_t)c0 * 7u, split, true);
        if (split) {
            d.parent[ch] = idx;
            d.sub[ch] = 0;
            d.segFirst[ch] = (uint8_t)p;
            d.segOff[(size_t)ch * kFrSegs] = ((d.jobH[j] + (uint64_t)sub * frCoef(p)) & kOffMask) | ((uint64_t)(d.replica ? 0u : owner) << 56);
            const uint64_t bits = (uint64_t)__double_as_longlong(e);
            d.qErr[ch] = bits;
            atomicAdd(&L.hist[frDigit(0, bits, ch)], 1u);
            if (sub == 0) {
                d.nodes[idx].child_idx = c0;
                d.nodes[idx].degree = kInteriorDegree;
                d.nodes[idx].coeffs_start = 0;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < 2048; i += 1024)
        if (L.hist[i]) atomicAdd(&h->hist1[i], L.hist[i]);
    __syncthreads();  // (every wave's stores and atomics have drained)
    if (tid == 0) {
        frRelease();
        if (!round0) __hip_atomic_fetch_add(&h->opsArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&h->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// all threads: every header word but the round number, a system-scope fence, then the round number -- the host watches that word
__device__ __forceinline__ void frMirror(const FrDev& d) {
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kRoundWord = offsetof(FrHdr, round) / 4;
    __syncthreads();  // (the header's writers have drained their stores; the copy reads past the L1)
    constexpr uint32_t kLandedWord = offsetof(FrHdr, landed) / 4;  // (landed and stored[] live in the mirror alone)
    if (tid < kFrHdrCopyBytes / 4 && tid != kRoundWord && (tid < kLandedWord || tid > kLandedWord + 2))
        reinterpret_cast<volatile uint32_t*>(d.hostHdr)[tid] = reinterpret_cast<volatile uint32_t*>(d.hdr)[tid];
    __syncthreads();  // (... and the copying waves theirs)
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the words above are in host memory before the round word is
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        reinterpret_cast<volatile uint32_t*>(d.hostHdr)[kRoundWord] = reinterpret_cast<volatile uint32_t*>(d.hdr)[kRoundWord];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// The running total, Octree.cpp:253-290 -- one dependent addition after the other in the reference's order, from the dense operand
// list the update workgroups have written (round 0: straight from the fits' error slots).  One workgroup: wave 0 adds, alone on its
// SIMD (waves 4, 8 and 12 do nothing); the twelve waves of the other three SIMDs bring the next 2048 operands into the other half of an
// LDS buffer meanwhile, each lane's loads in flight together (a lone wave loading its own operands waits out an L2 round trip every
// 64 jobs: 129 us for round 0's 4096 additions; three loader waves taking their elements one round trip after the other still left
// the adder waiting).  Rounds 2-5 added with v_add_f64, 8.6 cycles an addition with a fresh operand (tools/chain_lab.hip tried what else
// could feed it -- DPP broadcasts, SGPR operands through scalar loads, one active lane, a hand-scheduled loop -- nothing did better);
// since round 6 the additions run four at a time on the matrix pipe, in the same order with the same roundings: 5.2 cycles (below).
// All waves that are still there call (barriers inside).
__device__ double frRunChain(const FrDev& d, FrLds& L, double total, uint32_t nOps, bool round0) {
    const FrHdr* h = d.hdr;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const bool adder = wave == 0, loader = (wave & 3u) != 0;
    const uint32_t ltid = (wave - 1u - (wave >> 2)) * 64u + (tid & 63u);  // 0 .. 767 among the loaders
    double* sOps = reinterpret_cast<double*>(L.key);                         // [2][2048]
    auto operand = [&](uint32_t k) { return round0 ? d.errs[frErrSlot(d, h, k)] - d.batchErr[k] : d.ops[k]; };
    auto loadChunk = [&](uint32_t first, double* dst) {  // operands [first, first + 2048) (those below nOps): three a lane
        double v[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const uint32_t k = ltid + (uint32_t)u * 768u;
            v[u] = (k < 2048u && first + k < nOps) ? operand(first + k) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const uint32_t k = ltid + (uint32_t)u * 768u;
            if (k < 2048u && first + k < nOps) dst[k] = v[u];
        }
    };
    if (loader) loadChunk(0, sOps);
    __syncthreads();
    if (d.stamps && tid == 0) d.hdr->dbg[15] = __builtin_readcyclecounter();
    for (uint32_t c0 = 0, half = 0; c0 < nOps; c0 += 2048, half ^= 1u) {
        const uint32_t n = nOps - c0 < 2048u ? nOps - c0 : 2048u;
        if (loader) {
            if (c0 + 2048u < nOps) loadChunk(c0 + 2048u, sOps + (half ^ 1u) * 2048u);
        } else if (adder) {
            // FOUR additions an instruction, on the matrix pipe (round 6).  v_mfma_f64_4x4x4_4b_f64 computes D = A B + C with k = 4; with
            // B = 1 every product is exact, and the hardware accumulates the four terms of an output one after the other, each through
            // a fused multiply-add -- i.e. D = (((C + a0) + a1) + a2) + a3 with IEEE rounding after every step, the very additions of
            // Octree.cpp:253-290 in the reference's order (tools/mfma_chain_lab.hip: 262 144 random sums of mixed magnitudes and signs,
            // signed zeros, subnormals and half-ulp ties bit for bit equal to the sequential v_add_f64 sum -- and unequal to the reversed
            // and to the exactly rounded sum).  A dependent chain of them costs 20.6 cycles an instruction = 5.2 cycles an addition
            // against v_add_f64's 8.7 (rounds 2-5: 9.3 with its operands fed from LDS).  Output (0, 0) of block 0 -- lane 0 -- sums
            // the A operands of lanes 0, 16, 32, 48 in that order, so lane l supplies operand 4 t + (l >> 4) of step t; the other
            // fifteen outputs of every block compute sums nobody reads.  Sixteen steps' operands are in registers while the next
            // sixteen's LDS reads are in flight.
            const double* src = sOps + half * 2048u;
            uint32_t q = 0;
            {
                const uint32_t lane = tid & 63u;
                const double* mine = src + (lane >> 4);
                const uint32_t steps = n >> 2, batches = steps >> 4;
                uint32_t t = 0;
                if (batches) {
                    // two register batches taking turns (no copies between them: the wait for a batch's reads then sits in front of ITS
                    // additions, behind the other batch's)
                    double ra[16], rb[16];
#define FR_CHAIN_LOAD(r, batch)                                                                  \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) r[u] = mine[4u * (16u * (batch) + (uint32_t)u)];
#define FR_CHAIN_ADD(r) \
    _Pragma("unroll") for (int u = 0; u < 16; ++u) total = __builtin_amdgcn_mfma_f64_4x4x4f64(r[u], 1.0, total, 0, 0, 0);
                    FR_CHAIN_LOAD(ra, 0u)
                    uint32_t b = 0;
                    for (; b + 2u <= batches; b += 2u) {
                        FR_CHAIN_LOAD(rb, b + 1u)
                        __builtin_amdgcn_sched_barrier(0);  // (the other batch's reads are issued before this batch's additions)
                        FR_CHAIN_ADD(ra)
                        if (b + 2u < batches) {
                            FR_CHAIN_LOAD(ra, b + 2u)
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        FR_CHAIN_ADD(rb)
                    }
                    if (b < batches) {
                        FR_CHAIN_ADD(ra)
                    }
#undef FR_CHAIN_LOAD
#undef FR_CHAIN_ADD
                    t = batches << 4;
                }
                for (; t < steps; ++t) total = __builtin_amdgcn_mfma_f64_4x4x4f64(mine[4u * t], 1.0, total, 0, 0, 0);
                q = steps << 2;
            }
            for (; q < n; ++q) total = total + src[q];
            if (d.stamps && tid == 0 && c0 == 0) d.hdr->dbg[16] = __builtin_readcyclecounter();
        }
        __syncthreads();
    }
    if (tid == 0) sOps[0] = total;  // (wave 0's lanes all carry it; the other waves get it through LDS)
    __syncthreads();
    return sOps[0];
}

// The last workgroup: the running total of rounds >= 1, beside everything else -- nobody waits for it before the next round is
// prepared: the leader only needs the total for the stop rule.  (Round 0's total the leader adds up itself: its operands are there when
// the launch starts, and most builds at everyday thresholds end with it -- no hand-over on their critical path.)
__device__ void frChainTotal(const FrDev& d, FrLds& L, bool chain0) {
    FrHdr* h = d.hdr;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    if (wave != 0 && (wave & 3u) == 0) return;  // (whole waves: the barriers below count the waves that are still there)
    const uint32_t stamp = h->round + 1u;
    const bool round0 = stamp == 1u;
    double total = h->total;
    if (tid == 0) L.stuck = 0;
    __syncthreads();
    // (the header has been read: the leader may rewrite it.  The round's counters it leaves alone until this workgroup has finished.)
    const uint32_t nJobs0 = h->nJobs;
    if (tid == 0) __hip_atomic_fetch_add(&h->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (nothing is published: "I have read")
    if (round0 && !chain0) return;
    uint32_t nOps = nJobs0;  // round 0: every cell's (newErr - initialErr), there since the launch began
    if (!round0) {
        if (tid == 0 && !frWaitAtLeast(&h->opsArrive, gridDim.x - 2u)) L.stuck = 1;  // (acquires)
        __syncthreads();
        nOps = frLoad(&h->rP) + 9u * frLoad(&h->rH);
        if (L.stuck) nOps = 0;
    }
    if (d.stamps && tid == 0) h->dbg[13] = __builtin_readcyclecounter();
    total = frRunChain(d, L, total, nOps, round0);
    if (tid == 0) {
        if (d.stamps) h->dbg[14] = __builtin_readcyclecounter();
        if (L.stuck) atomicExch(&h->stuck, 1u);
        h->rTotal = total;
        frRelease();
        __hip_atomic_store(&h->chainStamp, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Workgroup 0.  When every update workgroup has arrived: counters, the threshold bin of the next selection; then the stop rule
// (:216) -- if the running total is there already (round 0: always), else the next round is prepared first, on the assumption that
// the build goes on (`pre`: selection, batch and lists, by this workgroup), and the rule is applied when the total has come: a
// build that stops takes the prepared round's entries out of the header again (what else the preparation touched -- queue marks,
// histogram, lists -- is not read after the stop).  Last the header's mirror in pinned host memory (the host watches the round
// word: no copy).
__device__ void frLeadRound(const FrDev& d, FrLds& L, int pre, bool chain0) {
    FrHdr* h = d.hdr;
    const uint32_t tid = threadIdx.x, nJobs = h->nJobs, round = h->round, stamp = round + 1u;
    const bool round0 = round == 0;
    const uint32_t nNodes0 = h->nNodes, nQ0 = h->nQueued;
    // (the counters the close adds to, asked for now -- with the rest of the header, one round trip -- so that the close only stores)
    const uint64_t jobs0 = h->jobs, pRef0 = h->pRefines, hRef0 = h->hRefines, drop0 = h->dropped, nCoeffs0 = h->nCoeffs;
    const uint32_t nLeaves0 = h->nLeaves, maxDeg0 = h->maxDegree;
    const uint64_t fits0 = h->fits, samples0 = h->samples, arena0 = h->arenaUsed, split0 = h->splitFits;  // (what a prepared round adds to)
    const double target = d.target;
    if (tid == 0) L.flag = 0, L.stuck = 0;
    if (tid == 0 && round0) {
        *(volatile uint32_t*)&d.hostHdr->landed = d.buildStamp;
        __threadfence_system();
    }
    __syncthreads();
    frPeerCheck(d, &L.flag);
    FR_STAMP(9);
    double total0 = 0.0;
    const bool ownChain = round0 && !chain0;
    if (ownChain) {  // the round's total, here and now: every cell's (newErr - initialErr) in cell order
        if (d.stamps && tid == 0) h->dbg[13] = __builtin_readcyclecounter();
        total0 = frRunChain(d, L, h->total, nJobs, true);
        if (d.stamps && tid == 0) h->dbg[14] = __builtin_readcyclecounter();
    }
    if (tid == 0 && !frWaitAtLeast(&h->arrive, gridDim.x - 1u)) L.stuck = 1;  // (the update workgroups, and the last one's "header read"; acquires)
    __syncthreads();
    FR_STAMP(10);
    // ---- the round closes
    const uint32_t nP = frLoad(&h->rP), nH = frLoad(&h->rH), nD = frLoad(&h->rD), md = frLoad(&h->rMaxDeg);
    const long long cd = (long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&h->rCoeffDelta), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t overflow = frLoad(&h->overflow) | ((L.stuck || frLoad(&h->stuck)) ? 4u : 0u);
    const uint32_t nQ = nQ0 - (round0 ? 0u : nJobs) + nP + 8u * nH;  // (round 0's batch never sat in the queue)
    const uint32_t nNodes = nNodes0 + 8u * nH;
    // (a build whose total after round 0 says "stop" -- most builds at everyday thresholds -- selects nothing any more: no threshold)
    const bool stopsHere = ownChain && total0 < target;
    if (!stopsHere)
        for (uint32_t i = tid; i < 2048; i += 1024) L.hist[i] = frLoad(&h->hist1[i]);
    if (tid == 0) {
        const bool there = !ownChain && frLoad(&h->chainStamp) == stamp;
        if (there) frAcquire();  // (rTotal is read below)
        L.c = (ownChain || there) ? 1u : 0u;
    }
    __syncthreads();
    bool haveTotal = L.c != 0;
    const bool canGoOn = nQ != 0 && overflow == 0;
    if (canGoOn && nQ > d.K && !stopsHere) {
        frThreshold(L.hist, d.K, L.tmp, &L.t, &L.above);
    } else {
        if (tid == 0) L.t = -1, L.above = nQ;
        __syncthreads();
    }
    const int T = L.t;
    const uint32_t above = L.above;
    // the total: wait for it now unless the next round can be prepared meanwhile
    auto awaitTotal = [&]() {
        if (tid == 0 && !frWaitAtLeast(&h->chainStamp, stamp)) L.stuck = 1;  // (stamps only grow within a build)
        __syncthreads();
        if (L.stuck) overflow |= 4u;
    };
    const bool speculate = !haveTotal && pre && canGoOn;
    if (!haveTotal && !speculate) awaitTotal(), haveTotal = true;
    double total = 0.0;
    bool done = false;
    if (haveTotal) {
        if (ownChain) {
            total = total0;
        } else {
            total = *(volatile double*)&h->rTotal;  // (behind the acquire of whoever saw the stamp, and a barrier)
        }
        done = total < target || !canGoOn || overflow != 0;  // Octree.cpp:216
    }
    if (tid == 0) {
        h->nQueued = nQ;
        h->nNodes = nNodes;
        h->jobs = jobs0 + nJobs, h->pRefines = pRef0 + nP, h->hRefines = hRef0 + nH, h->dropped = drop0 + nD;
        h->nLeaves = nLeaves0 + 7u * nH;
        h->nCoeffs = (uint64_t)((int64_t)nCoeffs0 + cd);
        if (md > maxDeg0) h->maxDegree = md;
        h->round = stamp;
        h->takenCount = 0, h->candCount = 0;
        h->rPad = L.flag;  // a rank that failed this round, + 1
        h->t1 = T, h->above = above;
    }
    FR_STAMP(11);
    __syncthreads();
    if (pre && !done) {
        // ---- the next round's selection, level 0: everything in the exponent bins above T is taken, bin T is the candidate list
        //      (its level-1 digits counted as they are found)
        for (uint32_t i = tid; i < 2048; i += 1024) L.hist[i] = 0;
        if (tid == 0) L.c = 0, L.next = 0;
        __syncthreads();
        {
            const int lane = (int)(tid & 63);
            for (uint32_t base = 0; base < nNodes; base += 1024u) {
                const uint32_t i = base + tid;
                const uint64_t bits = i < nNodes ? d.qErr[i] : kNotQueued;
                const bool queued = bits != kNotQueued;
                const int d0 = (int)frDigit(0, bits, i);
                const bool take = queued && d0 > T, cand = queued && d0 == T;
                const unsigned long long mt = __ballot(take), mc = __ballot(cand);
                if (mt) {
                    uint32_t s = 0;
                    const int leader = __ffsll((long long)mt) - 1;
                    if (lane == leader) s = atomicAdd(&L.next, (uint32_t)__popcll(mt));
                    s = __shfl(s, leader, 64);
                    if (take) d.taken[s + (uint32_t)__popcll(mt & ((1ull << lane) - 1ull))] = i;
                }
                if (mc) {
                    uint32_t s = 0;
                    const int leader = __ffsll((long long)mc) - 1;
                    if (lane == leader) s = atomicAdd(&L.c, (uint32_t)__popcll(mc));
                    s = __shfl(s, leader, 64);
                    if (cand) {
                        d.candA[s + (uint32_t)__popcll(mc & ((1ull << lane) - 1ull))] = i;
                        atomicAdd(&L.hist[frDigit(1, bits, i)], 1u);
                    }
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        const uint32_t cand = L.c, taken = L.next;
        __syncthreads();
        const bool stopped = frBatchBody<true>(d, L, nQ, above, cand, taken, nNodes, haveTotal ? 0u : stamp);
        __syncthreads();
        if (stopped) {  // the total came while the batch was being chosen, and it says "stop": nothing to take back
            haveTotal = true, done = true;
            total = *(volatile double*)&h->rTotal;
        }
        if (!haveTotal) {
            awaitTotal();
            total = *(volatile double*)&h->rTotal;
            done = total < target || overflow != 0;
            if (done && tid == 0) h->fits = fits0, h->samples = samples0, h->arenaUsed = arena0, h->sampleUsed = 0, h->splitFits = split0;
        }
    }
    if (tid == 0) {
        h->total = total;
        // (the total is there, so the last workgroup has finished with the round's counters)
        h->arrive = 0, h->opsArrive = 0;
        h->rP = 0, h->rH = 0, h->rD = 0, h->rMaxDeg = 0, h->rCoeffDelta = 0;
        if (overflow) h->overflow = overflow;
        h->done = done ? 1u : 0u;
        if (done) h->nJobs = 0, h->nTasks = 0, h->nBlocks = 0;
        if (!done && pre && d.world > 1) *frStatusSlotNext(d, d.rank) = 0.0;  // this rank's status for the coming round's exchange
    }
    if (done && tid < 13)
        h->degBlocks[tid][0] = h->degBlocks[tid][1] = h->degTasks[tid][0] = h->degTasks[tid][1] = h->lowTasks[tid][0] = h->lowTasks[tid][1] = 0;
    if (!done && !pre)
        for (uint32_t i = tid; i < 2048; i += 1024) h->hist2[i] = 0;  // (the grid selection's level-1 histogram)
    FR_STAMP(12);
    frMirror(d);
}

// flags: bit 0 = the leader prepares the next round itself (trees up to HPSDF_FRONTIER_INLINE_NODES); bit 1 = round 0's running total is
// added up by the last workgroup, beside the leader's preparation, like every later round's (the host sets it when this context's last
// build went on past round 0: the leader adding it up itself is the shorter way only for a build that stops there)
__global__ __launch_bounds__(1024) void fr_round_kernel(FrDev d, int flags) {
    const int pre = flags & 1;
    const bool chain0 = (flags & 2) != 0;
    if (d.hdr->done) return;  // (set by an earlier launch: uniform over the grid -- this launch's leader sets it only after every workgroup has arrived)
    __shared__ FrLds L;
    if (blockIdx.x == 0)
        frLeadRound(d, L, pre, chain0);
    else if (blockIdx.x == gridDim.x - 1u)
        frChainTotal(d, L, chain0);
    else
        frUpdateJobs(d, L);
}

// ---------------------------------------------------------------------------------------------------------------------
// The coefficient count of every subtree, once the stop rule has fired: every leaf adds its own to each of its ancestors but the
// root (whose count is the header's nCoeffs).  d.sub is zero before: fr_init_kernel, and no round touches it.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fr_subtree_kernel(FrDev d) {
    const FrHdr* h = d.hdr;
    if (!h->done || h->overflow) return;
    // The uniformly refined tree's own nodes -- indices below 4681, levels 0 to 4 -- are ancestors of everything: a workgroup (256
    // consecutive nodes, i.e. relatives) adds its leaves up in LDS first and touches each of them once (every leaf for itself: ten
    // thousand atomics on the same eight words, 131 us; the first three levels alone in LDS: 21 us -- the eight children of a split
    // depth-4 cell still met at their parent's word)
    constexpr uint32_t kTop = 4736;  // >= (8^5 - 1) / 7 = 4681
    __shared__ uint32_t sTop[kTop];
    for (uint32_t k = threadIdx.x; k < kTop; k += 256) sTop[k] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < h->nNodes) {
        const uint32_t deg = d.nodes[i].degree;
        if (deg != kInteriorDegree) {
            const uint32_t c = frCoef((int)deg);
            uint32_t a = d.parent[i];
            while (a != 0) {
                if (a < kTop)
                    atomicAdd(&sTop[a], c);
                else
                    atomicAdd(&d.sub[a], c);
                a = d.parent[a];
            }
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kTop; k += 256)
        if (sTop[k]) atomicAdd(&d.sub[k], sTop[k]);
}

// ---------------------------------------------------------------------------------------------------------------------
// ReallocCoeffs (Octree.cpp:474-555) once the stop rule has fired.  One wave per node.  A leaf's coeffsStart =
// coefficients of everything the depth-first walk (children 0..7 from the root) visits before it = over its
// ancestors-or-self a: the subtree sizes of a's earlier siblings.  Its rows are then gathered from the arena, segment by
// segment.
// ---------------------------------------------------------------------------------------------------------------------
// Workgroup b takes nodes 64 b .. 64 b + 63, in two phases (256 nodes a workgroup left a 12 000-node tree 48 workgroups -- a fifth of
// the chip's CUs -- and every lane a dozen rows, each behind its own chain of loads: 56 us for 1.9 MB); d.store is pinned HOST memory (ToMemoryBlock's order, :424-456:
// coefficients, then the node array), written by the stores themselves -- no copy to launch, and the host, which watches
// FrHdr::stored, moves the node array into the block while the coefficients are still on their way.
//   1  the first wave, one lane per node: coeffsStart (lanes walk up alone: the earlier siblings' sizes level by level), the serialised node through
//      the wave's staging buffer (whole records, in order)
//   2  the workgroup's leaves' rows: one lane per row -- a prefix sum over the 64 leaves' row counts, a search for the row's leaf, the
//      segment it lies in -- so that every lane has a load in flight and a leaf's rows leave as one run
constexpr uint32_t kStoreNodes = 64;  // nodes per workgroup of fr_store_kernel (one wave's worth: four times the workgroups of a node per thread)
__global__ __launch_bounds__(256) void fr_store_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (!h->done || h->overflow) return;
    const uint32_t n = h->nNodes, tid = threadIdx.x;
    const uint64_t nCoeffs = h->nCoeffs;
    __shared__ uint64_t sStage[64 * 7];
    __shared__ uint32_t sStart[kStoreNodes], sRows[kStoreNodes + 1];
    if (tid < 64) {  // ---- phase 1, the first wave: a lane per node
        const uint32_t i = blockIdx.x * kStoreNodes + tid;
        const bool live = i < n;
        hpsdf_node nd;
        nd.child_idx = ~0ull, nd.degree = kInteriorDegree, nd.depth = 0, nd.coeffs_start = 0;
        if (live) nd = d.nodes[i];
        const bool leaf = live && nd.degree != kInteriorDegree;
        uint32_t start = 0;
        if (leaf) {
            uint32_t a = i;
            while (a != 0) {
                const uint32_t par = d.parent[a];
                const uint32_t c0 = (uint32_t)d.nodes[par].child_idx;
                for (uint32_t sIdx = c0; sIdx < a; ++sIdx) {
                    const uint32_t dg = d.nodes[sIdx].degree;
                    start += dg == kInteriorDegree ? d.sub[sIdx] : frCoef((int)dg);
                }
                a = par;
            }
        }
        {
            uint64_t w[7];
            const uint64_t* src = reinterpret_cast<const uint64_t*>(&nd);
#pragma unroll
            for (int k = 0; k < 7; ++k) w[k] = src[k];
            if (leaf) w[4] = (uint64_t)start;  // (word 4: coeffs_start, @32)
            // the wave's 64 nodes are consecutive: one run
            frStoreRecords(sStage, reinterpret_cast<uint64_t*>(d.store + nCoeffs), w, (uint64_t)i * 7u, live, false);
        }
        sStart[tid] = start;
        // inclusive scan of the row counts over the wave
        uint32_t inc = leaf ? frCoef((int)nd.degree) : 0u;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(inc, off, 64);
            if ((int)tid >= off) inc += v;
        }
        sRows[tid + 1] = inc;
        if (tid == 0) sRows[0] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's stores have left)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: they are in host memory before the count says so
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (atomicAdd(&h->storeArrive[0], 1u) == gridDim.x - 1u) {  // the node array is complete
                *(volatile uint32_t*)&d.hostHdr->stored[0] = d.buildStamp;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    __syncthreads();
    // ---- phase 2, every wave: a lane per row
    const uint32_t total = sRows[kStoreNodes];
    for (uint32_t e = tid; e < total; e += 256u) {
        uint32_t lo = 0, hi = kStoreNodes;  // the leaf whose rows hold element e: the last one with sRows[leaf] <= e
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (sRows[mid] <= e)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t node = blockIdx.x * kStoreNodes + lo, r = e - sRows[lo];
        const int first = d.segFirst[node];
        int sg = 0;  // the segment row r lies in: rows [coef(first + sg - 1), coef(first + sg))
        while (r >= frCoef(first + sg)) ++sg;
        const uint32_t r0 = sg == 0 ? 0u : frCoef(first + sg - 1);
        const uint64_t so = d.segOff[(size_t)node * kFrSegs + sg];
        // one rank: straight from the arena; several: from the all-gathered pack buffers (fr_pack_kernel)
        const double* src = (d.world == 1 || d.replica) ? d.arena + (so & kOffMask) : d.pack + (size_t)(so >> 56) * d.packStride + d.packPos[(size_t)node * kFrSegs + sg];
        d.store[(size_t)sStart[lo] + r] = src[r - r0];
    }
    __syncthreads();  // (every wave's stores have left)
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(&h->storeArrive[1], 1u) == gridDim.x - 1u) {
            *(volatile uint32_t*)&d.hostHdr->stored[1] = d.buildStamp;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
// (base, count: this rank's run of error slots -- on one rank all of them, from 0; on several, rank r's part of the errors)
__global__ __launch_bounds__(256) void fr_weigh_kernel(FrDev d, uint32_t stride, uint32_t base, uint32_t count) {  // stride 9: every error of a job; round 0: the first only
    const FrHdr* h = d.hdr;
    if (h->done) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    if (stride == 1u && i % 9u != 0u) return;
    d.errs[base + i] = d.errs[base + i] * d.weights[base + i];
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
__global__ __launch_bounds__(256) void fr_init_kernel(FrDev d, FrTemplate t) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < t.nNodes) {
        d.nodes[i] = t.nodes[i];
        d.parent[i] = t.parent[i];
        d.sub[i] = 0;  // (fr_subtree_kernel, when the build has stopped)
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
    uint64_t arenaCap = 0, sampleCap = 0;
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
    // fr_init_kernel has run for (cleanRank, cleanWorld) and nothing since: a build that ends well leaves the workspace ready for the
    // next one (the launch runs while the host hands the block over), so a Create starts with its first fit
    bool clean = false;
    bool lastWentOn = false;  // this context's last build went on past round 0 (fr_round_kernel's flag bit 1: a guess about timing, never about results)
    int cleanRank = -1, cleanWorld = -1;
    uint32_t buildStamp = 0;
    std::vector<hpsdf_node> hostNodesAfterRound0;  // the node array of a tree that stops after round 0, serialised
    char* pinned = nullptr;                        // staging of the finished block
    double* pinnedDev = nullptr;                   // ... as the device addresses it
    size_t pinnedCap = 0;
    // weighted builds: the round's |mean FApprox| values as the device writes them, the weights as the host answers, the
    // "means are there" word pair (pinned, coherent: both sides watch them while the other writes)
    double* hostMeans = nullptr;
    double* hostWeights = nullptr;
    uint32_t* hostFlag = nullptr;
    uint32_t flagStamp = 0;
    hipError_t ensureWeighting() {
        if (hostMeans) return hipSuccess;
        const size_t n = 8 * ((size_t)kFrJobs * HPSDF_JOB_HEADER_DOUBLES + kFrStatusPad);  // (indexed like errs: up to eight ranks' parts)
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
    // device bytes per node of capacity (ensureNodes' arrays; hpsdf_ctx_set_build_limits counts with it)
    static constexpr uint64_t kBytesPerNode = sizeof(hpsdf_node) + sizeof(uint64_t) /* qErr */ + 2 * sizeof(uint32_t) /* parent, segFirst */ +
                                              (uint64_t)kFrSegs * sizeof(uint64_t) /* segOff */ + sizeof(uint64_t) /* sub */ + 2 * sizeof(uint32_t) /* candA, candB */;
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
        const uint64_t nc = (need + (1ull << 20) - 1) & ~((1ull << 20) - 1);  // to the need (GBs at high degrees), not to a power of two
        hipError_t e = grow(&samples, 0, nc, s, false);
        if (e == hipSuccess) sampleCap = nc;
        return e;
    }
    // multi-rank: errs is [world][4096 * 9], plus owners, pack positions, round-0 share
    int ranksCap = 1;
    uint64_t packCap = 0;
    FitTask* r0Tasks = nullptr;
    FitBlock* r0Blocks = nullptr;
    uint64_t* r0JobP = nullptr;
    int r0Rank = -1, r0World = -1;  // what (rank, world, error stride, replica) the three r0 arrays were built for
    bool r0Replica = false;
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
        pinned = nullptr, pinnedDev = nullptr, pinnedCap = 0;
        hipError_t e = hipHostMalloc((void**)&pinned, nc, hipHostMallocCoherent | hipHostMallocMapped);  // (round 0's fit writes it while the host may look)
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&pinnedDev, pinned, 0);
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
        if (e == hipSuccess) e = hipMalloc((void**)&d.ops, (size_t)kFrJobs * 9 * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&d.jobRecA, kFrJobs * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.jobRecB, kFrJobs * sizeof(uint16_t));
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
        if (e == hipSuccess) e = ensurePinned(4u << 20);
        return e;
    }
    ~FrontierWorkspace() {
        if (device >= 0) (void)hipSetDevice(device);
        for (void* p : {(void*)d.hdr, (void*)d.rnd, (void*)d.nodes, (void*)d.qErr, (void*)d.parent, (void*)d.segOff, (void*)d.segFirst,
                        (void*)d.sub, (void*)d.taken, (void*)d.candA, (void*)d.candB, (void*)d.wBatchIdx, (void*)d.wBatchErr, (void*)d.wJobP,
                        (void*)d.wJobH, (void*)d.ops, (void*)d.jobRecA, (void*)d.jobRecB, (void*)d.tasks, (void*)d.blocks, (void*)d.errs,
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

bool frontierEligible(const hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, uint64_t K) {
    if (const char* e = std::getenv("HPSDF_HOST_FRONTIER"))
        if (e[0] == '1') return false;
    if (cfg->weighting_type > 2) return false;    // (unknown weighting: the host scheduler reports it)
    if (cfg->enable_logging) return false;        // the per-job log line is printed by the host scheduler
    const hpsdf_field* in = innermost(field);
    if (!in || (in->kind != kHostAnalytic && in->kind != kHostMesh)) return false;  // callbacks are sampled by host threads
    if (in->kind == kHostMesh) {
        const char* e = std::getenv("HPSDF_MESH_FUSED");
        if (e && e[0] == '1') return false;
        if (meshFaceRuleReference(ctx)) return false;  // (the sampler's shared traversal assumes the default face rule: builder.cpp fits with the per-point one)
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
        if (frSyncEveryLaunch()) {                                                          \
            std::fprintf(stderr, "[frontier] " #kernel " ...");                             \
            const hipError_t se_ = hipStreamSynchronize(stream);                            \
            std::fprintf(stderr, " %s\n", hipGetErrorString(se_));                          \
        }                                                                                   \
    } while (0)
static bool frSyncEveryLaunch() {  // HPSDF_FRONTIER_SYNC=1: a fault names its kernel (diagnostic; the host's waits then find every round closed)
    static const bool on = std::getenv("HPSDF_FRONTIER_SYNC") != nullptr;
    return on;
}

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
        const hipError_t e = ws->ensureWeighting();
        if (e != hipSuccess) return hipFail(e, "frontier weighting buffers");
    }
    ws->d.K = Kj;
    ws->d.rank = rank, ws->d.world = world;
    ws->d.weighted = weighted ? 1 : 0;
    // (on several ranks a weighted incremental fit needs the node's previous rows, which another rank may have fitted: the arenas are
    // replicas of each other then -- FrDev::replica)
    const bool replica = weighted && world > 1;
    ws->d.replica = replica ? 1 : 0;
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
    // replica: the round's part of the arena while its exchange is still to come -- a failing rank enters it too, so round 0's part is
    // known before anything can fail (it depends on the template and the world size alone: equal runs of cells, builderSelect's cut).
    // Kept as an OFFSET into the arena (doubles) and a size: the arena can move between the moment the part is known and the moment a
    // failure sends this rank into the exchange (ensureArena reallocates), and a pointer taken earlier would then name freed memory.
    bool partPending = false;
    uint64_t partOff = 0;
    size_t arenaPartBytes = 0;
    if (replica) {
        const uint64_t nL = ws->tmpl.nLeaves;
        uint32_t start = 0, maxCount = 1;
        for (int r = 0; r < world; ++r) {
            uint32_t end = (uint32_t)nL;
            const uint64_t c = (uint64_t)frCoef(2) * 729ull, total = c * nL;  // (the cut of the round-0 section below)
            if (r + 1 < world) end = std::min<uint32_t>(std::max<uint32_t>((uint32_t)((total * (uint64_t)(r + 1) / (uint64_t)world + c - 1) / c), start), (uint32_t)nL);
            maxCount = std::max(maxCount, end - start);
            start = end;
        }
        partPending = true, partOff = 0;
        arenaPartBytes = (size_t)((((uint64_t)maxCount * frCoef(2)) + 15ull) & ~15ull) * sizeof(double);
    }
#ifdef HPSDF_TEST_HOOKS  // (lib/libhpsdf_hooks.so, built for tests/: the production library does not look at the variable)
    const char* injected = std::getenv("HPSDF_TEST_FAIL_RANK");  // "<rank>:<round>"
#else
    constexpr const char* injected = nullptr;  // (the statements that look at it fold away: the production library holds no trace of the hook)
#endif
    auto injectedFailure = [&](int round) {
        return injected && world > 1 && std::atoi(injected) == rank && std::strchr(injected, ':') && std::atoi(std::strchr(injected, ':') + 1) == round;
    };
    auto body = [&]() -> int {
    FieldDev fd;
    int rc;
    if (world > 1) phase = 1;
    if ((rc = makeFieldDev(ctx, field, nullptr, &fd))) return rc;
    if (injectedFailure(0)) return fail(HPSDF_ERR_OUT_OF_MEMORY, kInjectedFailureMsg);
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
    // (base, nMine: this rank's run of error slots and its jobs of the round)
    auto applyWeights = [&](const FitBlock* blocks, uint32_t maxBlocks, size_t lds, const FitTask* tasks, const uint32_t* dCount, bool round0, uint32_t base,
                            uint32_t nMine) -> int {
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
        const uint32_t nJobs = world == 1 ? std::min<uint32_t>(ws->hostFlag[0], kFrJobs) : std::min<uint32_t>(nMine, kFrJobs);
        const double dd = std::sqrt(3.0), strength = cfg.weighting_strength;
        const double* mean = ws->hostMeans + base;
        double* w = ws->hostWeights + base;
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
        FR_LAUNCH(fr_weigh_kernel, dim3((std::max(1u, nJobs) * 9u + 255u) / 256u), dim3(256), s, d, round0 ? 1u : 9u, base, nJobs * 9u);
        return HPSDF_OK;
    };
    uint8_t* early = nullptr;  // the block of a build that stops after round 0, begun before the device has finished
    bool earlyCopied = false;  // ... its coefficients are in it
    struct FreeEarly {
        hpsdf_ctx* ctx;
        uint8_t** p;
        ~FreeEarly() { ctx->freeBlock(*p); }
    } freeEarly{ctx, &early};
    // ---- round 0: every cell of the uniformly refined tree, straight from the template
    FrTemplate T0 = T;  // (this rank's share when there are several)
    const FitTask* r0Tasks = ws->tmplTasks;
    const FitBlock* r0Blocks = ws->tmplBlocks;
    const uint64_t* r0JobP = ws->tmplJobP;
    size_t r0Lds = ws->tmplLds;
    uint64_t part0 = 0;  // replica: doubles per rank in round 0's part of the arena
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
        // (replica: rank r's cells at r x part0 -- equal parts for the in-place all-gather -- and every cell's place known everywhere)
        uint32_t maxCount = 1;
        for (int r = 0; r < world; ++r) maxCount = std::max(maxCount, T0.sliceFirst[r + 1] - T0.sliceFirst[r]);
        part0 = ((uint64_t)maxCount * frCoef(2) + 15ull) & ~15ull;
        // this rank's task list, workgroup list and job -> arena map of round 0 depend on (rank, world, error stride, replica) alone:
        // built and uploaded once, kept on the device for the Creates that follow (no upload, no wait per Create)
        if (ws->r0Rank != rank || ws->r0World != world || ws->r0ErrStride != ws->d.errStride || ws->r0Replica != replica) {
            std::vector<FitTask>& tk = ws->hostR0Tasks;
            tk.assign(ws->hostTmplTasks.begin() + first, ws->hostTmplTasks.begin() + first + count);
            std::vector<uint64_t> jobP(T.nLeaves, ~0ull & kOffMask);
            for (uint32_t q = 0; q < count; ++q) {
                tk[q].outOff = (replica ? (uint64_t)rank * part0 : 0ull) + (uint64_t)q * frCoef(2);
                tk[q].sampleOff = (uint64_t)q * 729;
                tk[q].errSlot = (uint32_t)rank * ws->d.errStride + q * HPSDF_JOB_HEADER_DOUBLES;
                jobP[first + q] = tk[q].outOff;
            }
            if (replica)
                for (int r = 0; r < world; ++r)
                    for (uint32_t j = T0.sliceFirst[r]; j < T0.sliceFirst[r + 1]; ++j) jobP[j] = (uint64_t)r * part0 + (uint64_t)(j - T0.sliceFirst[r]) * frCoef(2);
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
            ws->r0Rank = rank, ws->r0World = world, ws->r0ErrStride = ws->d.errStride, ws->r0Replica = replica;
        }
        r0Tasks = ws->r0Tasks, r0Blocks = ws->r0Blocks, r0JobP = ws->r0JobP;
        r0Lds = frLds(2, g, pl);
        T0.nTasks = count, T0.nBlocks = nBl;
        T0.arenaRows = replica ? (uint64_t)world * part0 : (uint64_t)count * frCoef(2), T0.samples = (uint64_t)count * 729;
    }
    // ---- what a round needs beyond its lists
    // (later rounds have at most K jobs: smaller parts to all-gather than round 0's 4096)
    const uint32_t stride0 = kFrJobs * HPSDF_JOB_HEADER_DOUBLES + kFrStatusPad, strideK = Kj * HPSDF_JOB_HEADER_DOUBLES + kFrStatusPad;
    const uint32_t inlineNodes = [] {
        const char* e = std::getenv("HPSDF_FRONTIER_INLINE_NODES");  // tests: 0 sends every round through the grid selection
        return e ? (uint32_t)std::strtoul(e, nullptr, 10) : kFrInlineNodes;
    }();
    // fr_round_kernel closes round r and opens round r + 1 in one launch, so what round r + 1 can need -- nodes for round r's
    // splits, arena rows and sample slots for round r + 1's fits -- is bounded from what the host knows when it launches it: the tree
    // after round r - 1 (whose largest degree can have risen by one since).  Also decides whether round r + 1's from-scratch fits
    // are split (ctx->fitMode): they are unless the sample buffer they hand their field values over in cannot be had.
    bool splitOpen = false, samplesTooLarge = false;
    uint64_t measuredLimit = 0;  // (hpsdf_ctx_set_build_limits' default, measured at most once per Create: checkBuildLimits)
    auto prepareNext = [&](uint32_t knownNodes, uint64_t knownArena, uint32_t knownMaxDeg, int roundsDone) -> int {
        const int degBound = (int)std::min<uint32_t>(knownMaxDeg + 1u, kMaxDegree);
        const uint64_t needNodes = (uint64_t)knownNodes + 8ull * Kj;
        const uint64_t needArena = knownArena + (replica ? (uint64_t)world * ((uint64_t)Kj * rowsPerJob(degBound) + 16) : (uint64_t)Kj * rowsPerJob(degBound));
        {   // hpsdf_ctx_set_build_limits: the round about to open against the context's bounds -- before anything is allocated for it, so
            // that a runaway build ends here with its statistics in the message instead of minutes later in a failed hipMalloc
            // (a mesh build cannot do without its sample buffer; the hand-over buffer of split fits -- at most 2^31 samples, 16 GiB, whatever
            // the tree's size -- is not part of what grows without bound and is not counted: counting it would make the two schedulers'
            // split decisions depend on the limit, and they promise the same bytes)
            const uint64_t needSamples = mesh ? std::min<uint64_t>((uint64_t)Kj * samplesPerJob(degBound), 1ull << 31) : 0ull;
            const uint64_t bytes = needNodes * FrontierWorkspace::kBytesPerNode + (needArena + needSamples) * sizeof(double);
            const uint64_t held = (uint64_t)ws->nodeCap * FrontierWorkspace::kBytesPerNode + (ws->arenaCap + ws->sampleCap) * sizeof(double);
            const uint64_t growBytes = needNodes * FrontierWorkspace::kBytesPerNode + needArena * sizeof(double);  // (the default limit's subject)
            const int lrc = checkBuildLimits(ctx, knownNodes, bytes, growBytes, held, &measuredLimit, (uint64_t)roundsDone, roundsDone ? hh->total : 8.0 * 8.0 * 8.0 * 8.0 * HPSDF_INITIAL_NODE_ERR,
                                             cfg.target_error_threshold);
            if (lrc) return lrc;
        }
        if (needNodes > 0xFFFFFFF0ull) return fail(HPSDF_ERR_BUILD_LIMIT, "build limit: the device-side frontier indexes nodes with 32 bits");
        hipError_t e = ws->ensureNodes((uint32_t)needNodes, s);
        if (e == hipSuccess) e = ws->ensureArena(needArena, knownArena, s);
        if (e != hipSuccess) return hipFail(e, "frontier buffers");
        splitOpen = splitMode && degBound >= ctx->splitMinDegree;
        samplesTooLarge = false;
        if (mesh || splitOpen) {
            // (the round's own count decides on the device whether it splits -- the host scheduler's rule, FrHdr::splitRound; the host
            // provides for what the round can need, up to the 2^31 samples a split round may have)
            uint64_t need = (uint64_t)Kj * samplesPerJob(degBound);
            if (need > (1ull << 31)) {
                if (mesh) samplesTooLarge = true;  // (a mesh build fails when that round comes)
                need = 1ull << 31;
            }
            if (!samplesTooLarge && (e = ws->ensureSamples(need, s)) != hipSuccess) {
                if (mesh) return hipFail(e, "frontier sample buffer");
                (void)hipGetLastError();  // no room for the hand-over buffer: the exact fit needs none
                splitOpen = false;
            }
        }
        d.sampleCap = ws->sampleCap;
        return HPSDF_OK;
    };
    // a round's fits, from the lists the device wrote (kernels.hip; grids are upper bounds)
    auto launchRoundFits = [&](uint32_t knownMaxDeg, bool splitRound) -> int {
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
                HPSDF_HIP(launchFit(fs, deg <= 8 ? deg : 0, 1, d.blocks, taskBound, fitLdsTable[deg], d.tasks, ws->arena, d.errs, nullptr,
                                    ctx->dTables, fdr, rm, &d.hdr->degBlocks[deg][0]));
        }
        for (int k = 0; k < FrontierWorkspace::kSide; ++k)
            if (used[k]) {
                HPSDF_HIP(hipEventRecord(ws->joinEv[k], ws->side[k]));
                HPSDF_HIP(hipStreamWaitEvent(s, ws->joinEv[k], 0));
            }
        if (splitRound)  // the rows below the top degree of the split fits, from the samples the exact kernel left (H children: degree <= knownMaxDeg)
            for (int deg = std::max(2, ctx->splitMinDegree); deg <= (int)std::min<uint32_t>(knownMaxDeg, 11u); ++deg)
                HPSDF_HIP(launchFitMfmaLow(s, deg, d.tasks, &d.hdr->lowTasks[deg][0], 0u, 0u, 8u * Kj, ws->arena, ctx->dTables, ws->samples, rm, fd.leftAssoc));
        if (weighted) {
            size_t lds = 0;
            for (int deg = 2; deg <= degHi; ++deg) lds = std::max(lds, fitLdsTable[deg]);
            int rcw;
            const uint32_t mine = world == 1 ? 0u : hh->sliceFirst[rank + 1] - hh->sliceFirst[rank];  // (one rank: the device says how many)
            if ((rcw = applyWeights(d.blocks, taskBound, lds, d.tasks, &d.hdr->nBlocks, false, world == 1 ? 0u : (uint32_t)rank * strideK, mine))) return rcw;
        }
        return HPSDF_OK;
    };
    // The round is over for the host when the header's mirror shows the next round number: the closing workgroup writes it into
    // pinned memory behind a system-scope fence.  Watching that word costs ~3 us; waking up from hipStreamSynchronize ~20.
    auto waitForRound = [&](uint32_t want) -> int {
        const double ts = now();
        const volatile uint32_t* roundWord = &hh->round;
        const double limit = ts + 2.0e3;  // two milliseconds of watching, then the ordinary wait
        while (*roundWord != want && now() < limit) frCpuRelax();
        if (*roundWord != want) {
            HPSDF_HIP(hipStreamSynchronize(s));
            if (*roundWord != want) {
                // The stream is idle and the mirror still shows the old round: the pinned mirror is not coherent on this system.
                // Fetch the header itself and let it decide.
                HPSDF_HIP(hipMemcpy(ws->hostHdr, d.hdr, kFrHdrCopyBytes, hipMemcpyDeviceToHost));
                if (hh->round != want) return fail(HPSDF_ERR_STATE, "frontier: a round ended without advancing");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        tSync += now() - ts;
        return HPSDF_OK;
    };

    d.errStride = stride0, d.errStrideNext = strideK;
    d.buildStamp = ++ws->buildStamp ? ws->buildStamp : ++ws->buildStamp;  // (never 0)
    d.stamps = trace ? 1u : 0u;
    ws->hostHdr->round = 0, ws->hostHdr->done = 0;  // (the mirror still shows the previous build's last round: the waits below watch it)
    auto initDev = [&] {  // what fr_init_kernel sees: the arrays as they are now, round 0's stride (where this rank's status for round 0's exchange lies)
        FrDev di = d;
        di.errStride = stride0;
        return di;
    };
    {
        hipError_t e = ws->ensureArena(std::max<uint64_t>(1, T0.arenaRows), 0, s);
        if (e == hipSuccess && mesh) e = ws->ensureSamples(std::max<uint64_t>(1, T0.samples), s);
        if (e != hipSuccess) return hipFail(e, "frontier buffers");
        if ((rc = prepareNext(T.nNodes, T0.arenaRows, 2u, 0))) return rc;
        d.target = cfg.target_error_threshold;
        if (!ws->clean || ws->cleanRank != rank || ws->cleanWorld != world) FR_LAUNCH(fr_init_kernel, dim3((T.nNodes + 255) / 256), dim3(256), s, initDev(), T0);
        ws->clean = false;
        FieldDev fdr = fd;
        if (mesh) {
            HPSDF_HIP(launchMeshSample(s, r0Tasks, T0.nTasks, 2, ctx->dTables, fd, rm, ws->samples));
            fdr.kind = kFieldSamples;
            fdr.samples = ws->samples;
        }
        // (one rank: the rows also go straight into pinned host memory -- if the build stops after this round they are its packed store)
        HPSDF_HIP(launchFit(s, 2, 1, r0Blocks, T0.nBlocks, r0Lds, r0Tasks, ws->arena, d.errs, world == 1 ? ws->pinnedDev : nullptr, ctx->dTables, fdr, rm));
        if (weighted && (rc = applyWeights(r0Blocks, T0.nBlocks, r0Lds, r0Tasks, nullptr, true, world == 1 ? 0u : (uint32_t)rank * stride0,
                                           T0.sliceFirst[rank + 1] - T0.sliceFirst[rank])))
            return rc;
        if (replica) {
            if (arenaPartBytes != (size_t)part0 * sizeof(double)) return fail(HPSDF_ERR_STATE, "frontier: round 0's part of the arena");
            partPending = false;  // (entered: a gather that fails is the callback's failure, not a reason to enter it again)
            if ((rc = exchange(ws->arena, arenaPartBytes, "round 0's rows"))) return rc;
        }
        if ((rc = exchange(d.errs, (size_t)stride0 * sizeof(double), "round 0"))) return rc;
        phase = 0;
        d.errStride = strideK;  // (of the exchange that comes next: where a failing rank leaves its status)
    }
    int rounds = 0;  // rounds the host has seen closed
    uint32_t knownNodes = T.nNodes, knownMaxDeg = 2;
    uint64_t knownArena = T0.arenaRows;
    for (;;) {
        // close round `rounds`, open the next
        const bool pre = (uint64_t)knownNodes + 8ull * Kj <= inlineNodes;
        const bool splitCur = splitOpen, tooLargeCur = samplesTooLarge;
        FrDev dk = d;
        dk.splitFit = splitCur ? std::max(2, ctx->splitMinDegree) : 0;
        if (rounds == 0) {
            dk.batchIdx = ws->tmplLeaves, dk.batchErr = ws->tmplErr, dk.jobP = r0JobP, dk.jobH = r0JobP;
            dk.errStride = stride0;
        }
        FR_LAUNCH(fr_round_kernel, dim3(2 + ((rounds == 0 ? T.nLeaves : Kj) + 127u) / 128u), dim3(1024), s, dk, (pre ? 1 : 0) | (ws->lastWentOn && pre ? 2 : 0));
        // From the second round on the fits are launched without waiting for the header (one rank, no host step in between): what they
        // need to know is on the device -- their lists and counts -- and a build that has stopped leaves them nothing to do.  The
        // largest degree may have risen once more than the host knows.
        const bool noBlind = std::getenv("HPSDF_FRONTIER_NO_BLIND") != nullptr;  // (tests, comparisons)
        // Behind round 0 too when this context's last build went on past it (the guess that also hands round 0's total to the last
        // workgroup): the second round's fits then start when the closing launch ends instead of a host round trip later (~15 us); a
        // build that does stop there leaves two empty launches behind.
        const bool blind = (rounds >= 1 || ws->lastWentOn) && pre && world == 1 && !weighted && !frSyncEveryLaunch() && !noBlind;
        if (blind) {
            if (mesh && tooLargeCur) return fail(HPSDF_ERR_UNSUPPORTED, "round too large for the sampled mesh path");
            d.splitFit = dk.splitFit;
            FrDev de = d;  // (not dk: round 0's closing launch reads the template's batch, the lists are written for the next one)
            FR_LAUNCH(fr_emit_kernel, dim3((Kj + 127u) / 128u), dim3(1024), s, de);
            if ((rc = launchRoundFits(std::min<uint32_t>(knownMaxDeg + 1u, kMaxDegree), splitCur))) return rc;
        }
        if (rounds == 0 && world == 1) {
            // a build that stops here has its packed store in pinned memory already (the fit wrote it there too), and everything else
            // of its block is known in advance: write that part while the device works
            const uint64_t nc0 = T.arenaRows, nn0 = T.nNodes;
            early = (uint8_t*)ctx->allocBlock(8 + 8 * (size_t)nc0 + 8 + sizeof(hpsdf_node) * (size_t)nn0 + sizeof(hpsdf_config));
            if (early) {
                std::memcpy(early, &nc0, 8);
                std::memcpy(early + 8 + 8 * (size_t)nc0, &nn0, 8);
                std::memcpy(early + 16 + 8 * (size_t)nc0, ws->hostNodesAfterRound0.data(), sizeof(hpsdf_node) * (size_t)nn0);
                std::memcpy(early + 16 + 8 * (size_t)nc0 + sizeof(hpsdf_node) * (size_t)nn0, &cfg, sizeof cfg);
                // ... and the rows themselves as soon as the closing launch says that the fit before it has finished -- long before it
                // can say whether the build stops here (the running total is 4096 dependent additions away): 320 KB out of memory the
                // GPU has just written take a core ~20 us, which now pass while the device works.  In pieces, with an eye on the round
                // word: a build that goes on must not wait for a copy it will not use.
                const volatile uint32_t* landed = &hh->landed;
                const volatile uint32_t* roundWord = &hh->round;
                const double limit = now() + 2.0e3;
                while (*landed != d.buildStamp && *roundWord == 0 && now() < limit) frCpuRelax();
                if (*landed == d.buildStamp) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    const size_t total = 8 * (size_t)nc0, piece = 32768;
                    size_t at = 0;
                    for (; at < total && (*roundWord == 0 || *(const volatile uint32_t*)&hh->done); at += piece) std::memcpy(early + 8 + at, ws->pinned + at, std::min(piece, total - at));
                    earlyCopied = at >= total;
                }
            }
        }
        if ((rc = waitForRound((uint32_t)rounds + 1u))) return rc;
        ++rounds;
        if (world > 1 && hh->rPad)
            return fail(HPSDF_ERR_STATE, "rank " + std::to_string(hh->rPad - 1) + " failed in round " + std::to_string(rounds - 1) + ": its own error was returned there");
        if (hh->overflow & 1u) return fail(HPSDF_ERR_STATE, "frontier: node capacity exceeded");
        if (hh->overflow & 4u) return fail(HPSDF_ERR_STATE, "frontier: a workgroup of the round's kernel did not arrive");
        if (trace) {
            std::fprintf(stderr, "[frontier round %d] lead (cycles): wait-all %lld close %lld", rounds - 1, (long long)(hh->dbg[10] - hh->dbg[9]),
                         (long long)(hh->dbg[11] - hh->dbg[10]));
            if (!hh->done && pre) {
                std::fprintf(stderr, " | select %lld batch", (long long)(hh->dbg[0] - hh->dbg[11]));
                for (int k = 1; k <= 8; ++k) std::fprintf(stderr, " %lld", (long long)(hh->dbg[k] - hh->dbg[k - 1]));
                std::fprintf(stderr, " | total+commit %lld", (long long)(hh->dbg[12] - hh->dbg[8]));
            }
            std::fprintf(stderr, " | chain %lld (first operands %lld, first 2048 additions %lld)", (long long)(hh->dbg[14] - hh->dbg[13]), (long long)(hh->dbg[15] - hh->dbg[13]),
                         (long long)(hh->dbg[16] - hh->dbg[15]));
            std::fprintf(stderr, "\n");
        }
        if (rounds == 1 && world == 1 && hh->done && early && hh->nCoeffs == T.arenaRows && hh->nNodes == T.nNodes) {
            if (!earlyCopied) std::memcpy(early + 8, ws->pinned, 8 * (size_t)T.arenaRows);
            hipLaunchKernelGGL(fr_init_kernel, dim3((T.nNodes + 255) / 256), dim3(256), 0, s, initDev(), T0);  // for the next build
            ws->clean = hipGetLastError() == hipSuccess, ws->cleanRank = rank, ws->cleanWorld = world;
            ws->lastWentOn = false;
            *block = early;
            *size = 8 + 8 * (size_t)T.arenaRows + 8 + sizeof(hpsdf_node) * (size_t)T.nNodes + sizeof(hpsdf_config);
            early = nullptr;
            if (stats) {
                std::memset(stats, 0, sizeof *stats);
                stats->rounds = hh->round, stats->jobs = hh->jobs, stats->p_refines = hh->pRefines, stats->h_refines = hh->hRefines;
                stats->dropped = hh->dropped, stats->fits = hh->fits, stats->samples = hh->samples;
                stats->n_nodes = hh->nNodes, stats->n_leaves = hh->nLeaves, stats->n_coeffs = hh->nCoeffs, stats->total_error = hh->total;
                stats->fit_mode = (uint64_t)ctx->fitMode, stats->split_fits = hh->splitFits, stats->device_frontier = 1;
            }
            if (trace) std::fprintf(stderr, "[frontierCreate] us: total %.0f (waiting for the device %.0f, stopped after round 0)\n", now() - t0, tSync);
            return HPSDF_OK;
        }
        ctx->freeBlock(early);
        early = nullptr;
        if (hh->done) break;
        // (as builderSelect: a total that is NaN or infinite never falls below the threshold -- the field is not finite somewhere)
        if (!(std::fabs(hh->total) <= DBL_MAX))
            return fail(HPSDF_ERR_INVALID_ARGUMENT, "the field is not a finite number at some sample point (the build's total error is NaN or infinite)");
        knownNodes = hh->nNodes, knownMaxDeg = hh->maxDegree, knownArena = hh->arenaUsed;
        if (world > 1) phase = 2;
        // (replica: the round's rows are exchanged too -- from here on a rank that fails enters that exchange as well.  With the grid
        // selection the part is known only when the batch kernel has run: below)
        if (replica && pre) partPending = true, partOff = hh->roundBase, arenaPartBytes = (size_t)hh->partStride * sizeof(double);
        if (!pre) {  // a large tree: the selection as a grid, then batch and lists (the header's arenaUsed lags one round: bound it)
            FrDev dl = d;
            dl.splitFit = dk.splitFit;
            FR_LAUNCH(fr_select_kernel, dim3(std::min<uint32_t>(512u, (knownNodes + 255u) / 256u)), dim3(256), s, dl);
            FR_LAUNCH(fr_batch_kernel, dim3(1), dim3(1024), s, dl);
            FR_LAUNCH(fr_tasks_kernel, dim3((16u * Kj + 255u) / 256u), dim3(256), s, dl);
            if (replica) {
                // Where the round's part of the arena lies is the batch kernel's decision, and the exchange below needs it.  The
                // context's stream does not synchronise with the null stream (hipStreamNonBlocking): a plain hipMemcpy right behind the
                // launches would fetch the PREVIOUS round's roundBase / partStride.  Wait for the three kernels, then fetch.
                HPSDF_HIP(hipStreamSynchronize(s));
                HPSDF_HIP(hipMemcpy(ws->hostHdr, d.hdr, kFrHdrCopyBytes, hipMemcpyDeviceToHost));
                partPending = true, partOff = hh->roundBase, arenaPartBytes = (size_t)hh->partStride * sizeof(double);
            }
        }
        // (a rank whose share fails from here on knows which exchanges the others are heading for, and how large their parts are: with
        // the grid selection that is only true once the batch kernel has spoken, hence the order.  A failed launch of the three
        // selection kernels themselves -- a lost device, not a full one -- is not covered: that rank enters the errors' exchange alone.)
        if (injectedFailure(rounds)) return fail(HPSDF_ERR_OUT_OF_MEMORY, kInjectedFailureMsg);
        if (mesh && tooLargeCur) return fail(HPSDF_ERR_UNSUPPORTED, "round too large for the sampled mesh path");
        if (pre && !blind) {
            FrDev de = d;
            de.splitFit = dk.splitFit;
            FR_LAUNCH(fr_emit_kernel, dim3((Kj + 127u) / 128u), dim3(1024), s, de);
        }
        // capacities of the launch that closes this round; the arena may move: nothing that writes it is in flight
        if ((rc = prepareNext(knownNodes, knownArena + (pre ? 0ull : (uint64_t)(replica ? world : 1) * ((uint64_t)Kj * rowsPerJob((int)knownMaxDeg) + 16)), knownMaxDeg, rounds))) return rc;
        d.splitFit = dk.splitFit;
        if (!blind && (rc = launchRoundFits(knownMaxDeg, splitCur))) return rc;
        if (replica) {
            partPending = false;
            if ((rc = exchange(ws->arena + partOff, arenaPartBytes, "a round's rows"))) return rc;  // (the arena may have moved: offset, not pointer)
        }
        if (frSyncEveryLaunch()) std::fprintf(stderr, "[frontier] fits of round %d ... %s\n", rounds, hipGetErrorString(hipStreamSynchronize(s)));
        if ((rc = exchange(d.errs, (size_t)strideK * sizeof(double), "a round's errors"))) return rc;
        phase = 0;
    }
    // ReallocCoeffs, and Octree::ToMemoryBlock (Octree.cpp:424-456: [u64 nCoeffs][f64 x nCoeffs][u64 nNodes][Node x nNodes][Config]): the
    // device writes coefficients and node array into pinned host memory (fr_store_kernel), the host moves each part into the block
    // when the mirror says it is there -- the node array while the coefficients are still landing
    const uint64_t nc = hh->nCoeffs, nn = hh->nNodes;
    {
        hipError_t e = ws->ensurePinned(8 * (size_t)nc + sizeof(hpsdf_node) * (size_t)nn);
        if (e != hipSuccess) return hipFail(e, "block staging");
        d.store = ws->pinnedDev;
        FR_LAUNCH(fr_subtree_kernel, dim3(((uint32_t)nn + 255u) / 256u), dim3(256), s, d);
        if (world > 1 && !replica) {
            // the packed store from one all-gather of the ranks' pack buffers (each rank's own segments in node order)
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
        }
        FR_LAUNCH(fr_store_kernel, dim3(((uint32_t)nn + kStoreNodes - 1u) / kStoreNodes), dim3(256), s, d);
    }
    const size_t bytes = 8 + 8 * (size_t)nc + 8 + sizeof(hpsdf_node) * (size_t)nn + sizeof(hpsdf_config);
    uint8_t* p = (uint8_t*)ctx->allocBlock(bytes);
    if (!p) {
        (void)hipStreamSynchronize(s);
        return fail(HPSDF_ERR_OUT_OF_MEMORY, "malloc of the memory block failed");
    }
    const double tc = now();
    std::memcpy(p, &nc, 8);
    std::memcpy(p + 8 + 8 * (size_t)nc, &nn, 8);
    std::memcpy(p + 16 + 8 * (size_t)nc + sizeof(hpsdf_node) * (size_t)nn, &cfg, sizeof cfg);
    for (int part = 0; part < 2; ++part) {  // 0: the node array, 1: the coefficients
        const volatile uint32_t* flag = &hh->stored[part];
        const double limit = now() + 2.0e3;
        while (*flag != d.buildStamp && now() < limit) frCpuRelax();
        if (*flag != d.buildStamp) {  // (two milliseconds of watching, then the ordinary wait: the launch has finished, its stores are there)
            const hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                ctx->freeBlock(p);
                return hipFail(e, "block download");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (part == 0)
            std::memcpy(p + 16 + 8 * (size_t)nc, ws->pinned + 8 * (size_t)nc, sizeof(hpsdf_node) * (size_t)nn);
        else
            std::memcpy(p + 8, ws->pinned, 8 * (size_t)nc);
    }
    const double tcopy = now() - tc;
    *block = p;
    *size = bytes;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->rounds = hh->round, stats->jobs = hh->jobs, stats->p_refines = hh->pRefines, stats->h_refines = hh->hRefines;
        stats->dropped = hh->dropped, stats->fits = hh->fits, stats->samples = hh->samples;
        stats->n_nodes = nn, stats->n_leaves = hh->nLeaves, stats->n_coeffs = nc, stats->total_error = hh->total;
        stats->fit_mode = (uint64_t)ctx->fitMode, stats->split_fits = hh->splitFits, stats->device_frontier = 1;
    }
    hipLaunchKernelGGL(fr_init_kernel, dim3((T.nNodes + 255) / 256), dim3(256), 0, s, initDev(), T0);  // for the next build
    ws->clean = hipGetLastError() == hipSuccess, ws->cleanRank = rank, ws->cleanWorld = world;
    ws->lastWentOn = rounds > 1;
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
        if (partPending) (void)gather(gatherUser, ws->arena + partOff, arenaPartBytes, (void*)s);  // (replica: the others exchange the round's rows first)
        // (this rank's errors of the round were never written: zeros instead of whatever the buffer held -- the others' round kernels
        // read every job's errors before their leader looks at the status)
        if (hipMemsetAsync(ws->d.errs + (size_t)rank * stride, 0, stride * sizeof(double), s) == hipSuccess &&
            hipMemcpyAsync(ws->d.errs + (size_t)rank * stride + (stride - 1), &ones, sizeof ones, hipMemcpyHostToDevice, s) == hipSuccess)
            (void)gather(gatherUser, ws->d.errs, stride * sizeof(double), (void*)s);
        (void)hipStreamSynchronize(s);
        return fail(rcBody, own);
    }
    if (rcBody) {
        // A build that fails leaves nothing in flight: the round kernel's leader publishes the round number -- which is what the host
        // watches -- BEFORE it prepares the next round, so a failure noticed right there (another rank's status, an overflow) would
        // otherwise return with that kernel still writing the header's mirror and the lists, into buffers the caller may free next.
        const std::string own = hpsdf_last_error();
        (void)hipStreamSynchronize(s);
        (void)hipGetLastError();
        return fail(rcBody, own);
    }
    return rcBody;
}

}  // namespace hpsdf
