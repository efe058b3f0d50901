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
// A round:
//   fr_select_kernel   top-K of the frontier by (error desc, node index asc) as an MSB radix select over the error's bit
//                      pattern: level 0 (exponent, 2048 bins) against a histogram kept incrementally by the apply kernel,
//                      everything strictly above the threshold bin is taken, the bin itself becomes the candidate list
//   fr_batch_kernel    (one workgroup) refines the candidates digit by digit until <= 4096 remain, sorts those exactly,
//                      sorts the taken nodes by index -> the batch; then emits the round's FitTask / FitBlock lists
//                      grouped by shape, the per-degree launch ranges, arena and sample offsets
//   mesh_sample_kernel / fit_kernel   (kernels.hip) over device-written ranges: grids are upper bounds
//   fr_apply_kernel    (one workgroup) improvements (:814-825, :846-854), decision (:600-601), child creation, queue
//                      updates, the running total in the reference's order (one lane, :253-290), stop rule (:216)
//   fr_totals_kernel + fr_store_kernel   when the stop rule has fired: ReallocCoeffs -- subtree coefficient counts
//                      bottom-up, every leaf's coeffsStart by walking up its ancestors, coefficients gathered into the
//                      packed store
// Round 0 (the 4096 coarse cells) skips the selection: its batch is the uniformly refined tree's leaves in index order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
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
constexpr uint32_t kFrSort = 4096;  // exact sort capacity (LDS)

struct FrHdr {
    uint32_t nNodes, nQueued, nJobs, done;
    uint32_t round, maxDegree, maxDepth, overflow;
    uint32_t nTasks, nBlocks, takenCount, candCount;
    uint32_t above;
    int32_t t1;
    uint32_t nLeaves, pad0;
    uint32_t degBlocks[13][2];  // {first block, count} per degree: the range a fit launch walks
    uint32_t degTasks[13][2];   // {first task, count} per degree: the range a mesh-sampler launch walks
    uint64_t arenaUsed, sampleUsed, nCoeffs, pad1;
    uint64_t jobs, pRefines, hRefines, dropped, fits, samples;
    double total, target;
    uint32_t hist1[2048];  // queued nodes per exponent bin (kept by apply / batch)
    uint32_t hist2[2048];  // level-1 digits of the candidates of the current selection
};

struct FrDev {
    FrHdr* hdr;
    hpsdf_node* nodes;
    uint64_t* qErr;     // bit pattern of the queued error; kNotQueued = not in the frontier
    uint32_t* parent;
    uint64_t* segOff;   // [node][kFrSegs] arena offsets (doubles)
    uint8_t* segFirst;  // degree of the node's first segment
    uint32_t* sub;      // finalize: coefficients in the subtree
    uint32_t* taken;    // selection: nodes taken so far (unordered)
    uint32_t* candA;
    uint32_t* candB;
    uint32_t* batchIdx;
    double* batchErr;
    uint64_t* jobP;     // per job: arena offset of the P result / of the first H child
    uint64_t* jobH;
    FitTask* tasks;
    FitBlock* blocks;
    double* errs;       // [jobs][9]
    double* store;      // packed coefficients (ReallocCoeffs)
    const double* arena;
    uint64_t storeCap;
    uint32_t nodeCap, K;
};

__host__ __device__ inline uint32_t frCoef(int p) { return p == 6 ? 83u : (uint32_t)((p + 1) * (p + 2) * (p + 3) / 6); }
__host__ __device__ inline int frClass(int degree, bool incr, int depth) { return (2 * degree + (incr ? 1 : 0)) * kFrDepths + depth; }
__host__ __device__ inline size_t frLds(int degree, int g, int planes) {  // = fitLdsBytes (kernels.hip)
    const size_t nq = 4 * (size_t)degree + 1;
    return ((size_t)(degree + 1) * nq + 2 * nq + 8 * (size_t)g + (size_t)g * planes * nq * nq) * sizeof(double);
}
// workgroup shape of `count` fits of one class: what fitShape (kernels.hip) gives an unweighted, sampled-or-analytic fit
__host__ __device__ inline void frShape(int degree, bool incr, uint32_t count, int* cells, int* planes) {
    const int nrows = incr ? (int)(frCoef(degree) - frCoef(degree - 1)) : (int)frCoef(degree);
    int gmax = nrows > kFitBlockThreads ? 1 : kFitBlockThreads / nrows;
    while (gmax > 1 && frLds(degree, gmax, 1) > kFitMaxLdsBytes) --gmax;
    const uint32_t spread = (count + 511u) / 512u;
    int g = (int)(spread < 1u ? 1u : spread);
    g = g < gmax ? g : gmax;
    const int nq = 4 * degree + 1;
    int pl = nq;
    while (pl > 1 && frLds(degree, g, pl) > kFitChunkLdsBytes) --pl;
    *cells = g;
    *planes = pl;
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
// count(bins > T) < need <= count(bins >= T); *above = count(bins > T).  All threads call; sTmp: blockDim.x words.
__device__ void frThreshold(const uint32_t* hist, uint32_t need, uint32_t* sTmp, int* outT, uint32_t* outAbove) {
    const int nt = (int)blockDim.x, per = 2048 / nt, tid = (int)threadIdx.x;
    uint32_t s = 0;
    for (int k = 0; k < per; ++k) s += hist[tid * per + k];
    sTmp[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint32_t cum = 0;
        int t = nt - 1;
        for (; t > 0 && cum + sTmp[t] < need; --t) cum += sTmp[t];
        int b = t * per + per - 1;
        for (; b > t * per && cum + hist[b] < need; --b) cum += hist[b];
        *outT = b;
        *outAbove = cum;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// selection, level 0
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fr_select_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (h->done) return;
    __shared__ uint32_t sHist[2048];
    __shared__ uint32_t sTmp[256];
    __shared__ int sT;
    __shared__ uint32_t sAbove;
    const uint32_t nQ = h->nQueued, nNodes = h->nNodes;
    for (int i = threadIdx.x; i < 2048; i += 256) sHist[i] = h->hist1[i];
    __syncthreads();
    if (nQ <= d.K) {  // the whole frontier is this round's batch
        if (threadIdx.x == 0) sT = -1, sAbove = nQ;
        __syncthreads();
    } else {
        frThreshold(sHist, d.K, sTmp, &sT, &sAbove);
    }
    const int T = sT;
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) sHist[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t stride = gridDim.x * 256u;
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
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (sHist[i]) atomicAdd(&h->hist2[i], sHist[i]);
    if (blockIdx.x == 0 && threadIdx.x == 0) h->t1 = T, h->above = sAbove;
}

// ---------------------------------------------------------------------------------------------------------------------
// selection, remaining levels + exact sort; then the round's task and workgroup lists
// ---------------------------------------------------------------------------------------------------------------------
// In-place bitonic sort of n = 4096 (key, val) pairs in LDS by "key descending, then val ascending"; 1024 threads.
__device__ void frBitonic(uint64_t* key, uint32_t* val) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t k = 2; k <= kFrSort; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < kFrSort / 2; t += blockDim.x) {
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

struct FrBatchLds {  // carved out of dynamic LDS; the sort arrays are dead once the batch is written
    uint64_t* key;    // [4096]
    uint32_t* val;    // [4096]
    uint32_t* hist;   // [2048]
    uint32_t* tmp;    // [1024]
    uint32_t* cCount;  // [kFrClasses] tasks per class
    uint32_t* cFirst;  // first task
    uint32_t* cCursor;
    uint32_t* cBlockFirst;
    uint32_t* cBlocks;
    uint64_t* cArena;   // first arena row (relative to the round's base)
    uint64_t* cSample;  // first sample
    uint8_t* cG;
    uint8_t* cPlanes;
    uint8_t* jDeg;   // [4096] per job
    uint8_t* jDepth;
    uint8_t* jCoarse;
};
constexpr size_t kFrBatchLdsBytes = 4096 * 8 + 4096 * 4 + 2048 * 4 + 1024 * 4 + kFrClasses * (5 * 4 + 2 * 8 + 2) + 3 * 4096 + 64;

__global__ __launch_bounds__(1024) void fr_batch_kernel(FrDev d, int skipSelect) {
    FrHdr* h = d.hdr;
    if (h->done) {
        if (threadIdx.x == 0) {
            h->nJobs = 0, h->nTasks = 0, h->nBlocks = 0;
            for (int g = 0; g < 13; ++g) h->degBlocks[g][0] = h->degBlocks[g][1] = h->degTasks[g][0] = h->degTasks[g][1] = 0;
        }
        return;
    }
    extern __shared__ unsigned char frRaw[];
    FrBatchLds L;
    {
        unsigned char* p = frRaw;
        L.key = (uint64_t*)p, p += 4096 * 8;
        L.cArena = (uint64_t*)p, p += kFrClasses * 8;
        L.cSample = (uint64_t*)p, p += kFrClasses * 8;
        L.val = (uint32_t*)p, p += 4096 * 4;
        L.hist = (uint32_t*)p, p += 2048 * 4;
        L.tmp = (uint32_t*)p, p += 1024 * 4;
        L.cCount = (uint32_t*)p, p += kFrClasses * 4;
        L.cFirst = (uint32_t*)p, p += kFrClasses * 4;
        L.cCursor = (uint32_t*)p, p += kFrClasses * 4;
        L.cBlockFirst = (uint32_t*)p, p += kFrClasses * 4;
        L.cBlocks = (uint32_t*)p, p += kFrClasses * 4;
        L.cG = p, p += kFrClasses;
        L.cPlanes = p, p += kFrClasses;
        L.jDeg = p, p += 4096;
        L.jDepth = p, p += 4096;
        L.jCoarse = p, p += 4096;
    }
    __shared__ int sT;
    __shared__ uint32_t sAbove, sCount, sNext;
    const uint32_t tid = threadIdx.x;
    uint32_t nJobs;
    if (skipSelect) {
        nJobs = h->nJobs;  // batchIdx / batchErr were put there by the initialisation (round 0)
    } else {
        const uint32_t nQ = h->nQueued;
        nJobs = nQ < d.K ? nQ : d.K;
        uint32_t need = nJobs - h->above;  // still to come out of the candidates
        uint32_t C = h->candCount;
        uint32_t* cur = d.candA;
        uint32_t* nxt = d.candB;
        for (uint32_t i = tid; i < 2048; i += 1024) L.hist[i] = h->hist2[i];
        __syncthreads();
        int level = 1;
        while (C > kFrSort && level <= 8) {  // refine by the digit of `level` (its histogram is in L.hist)
            frThreshold(L.hist, need, L.tmp, &sT, &sAbove);
            const int T = sT;
            const uint32_t abv = sAbove;
            if (tid == 0) sCount = 0, sNext = h->takenCount;
            __syncthreads();
            for (uint32_t i = tid; i < 2048; i += 1024) L.hist[i] = 0;
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
                        nxt[atomicAdd(&sCount, 1u)] = idx;
                        if (level < 8) atomicAdd(&L.hist[frDigit(level + 1, bits, idx)], 1u);
                    }
                }
            }
            __syncthreads();
            need -= abv;
            C = sCount;
            if (tid == 0) h->takenCount = sNext;
            __syncthreads();
            uint32_t* t = cur;
            cur = nxt, nxt = t;
            ++level;
        }
        // exact order of what is left (<= 4096 candidates; unique keys)
        for (uint32_t i = tid; i < kFrSort; i += 1024) {
            if (i < C) {
                const uint32_t idx = cur[i];
                L.key[i] = d.qErr[idx], L.val[i] = idx;
            } else {
                L.key[i] = 0, L.val[i] = 0xFFFFFFFFu;  // behind every real entry
            }
        }
        __syncthreads();
        if (C > 1) frBitonic(L.key, L.val);
        const uint32_t tk = h->takenCount;
        for (uint32_t i = tid; i < need; i += 1024) d.taken[tk + i] = L.val[i];
        __syncthreads();
        // the batch in node-index order
        for (uint32_t i = tid; i < kFrSort; i += 1024) {
            L.key[i] = 0;
            L.val[i] = i < nJobs ? d.taken[i] : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (nJobs > 1) frBitonic(L.key, L.val);
        for (uint32_t i = tid; i < 2048; i += 1024) L.hist[i] = 0;
        __syncthreads();
        for (uint32_t j = tid; j < nJobs; j += 1024) {
            const uint32_t idx = L.val[j];
            const uint64_t bits = d.qErr[idx];
            d.batchIdx[j] = idx;
            d.batchErr[j] = __longlong_as_double((long long)bits);
            d.qErr[idx] = kNotQueued;
            atomicAdd(&L.hist[frDigit(0, bits, idx)], 1u);
        }
        __syncthreads();
        for (uint32_t i = tid; i < 2048; i += 1024)
            if (L.hist[i]) h->hist1[i] -= L.hist[i];
        __syncthreads();
    }

    // ---- tasks: every job becomes 1 (coarse) or up to 9 cell fits, grouped by shape class
    for (uint32_t c = tid; c < (uint32_t)kFrClasses; c += 1024) L.cCount[c] = 0;
    __syncthreads();
    for (uint32_t j = tid; j < nJobs; j += 1024) {
        const hpsdf_node& n = d.nodes[d.batchIdx[j]];
        const double e = d.batchErr[j];
        const bool coarse = fabs(e - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;  // Octree.cpp:806,831
        const int p = n.degree, dep = n.depth;
        L.jDeg[j] = (uint8_t)p, L.jDepth[j] = (uint8_t)dep, L.jCoarse[j] = coarse ? 1 : 0;
        if (coarse) {
            atomicAdd(&L.cCount[frClass(2, false, dep)], 1u);  // :836-843
        } else {
            if (dep < kMaxDepth) atomicAdd(&L.cCount[frClass(p, false, dep + 1)], 8u);     // :814-822
            if (p < kMaxDegree - 1) atomicAdd(&L.cCount[frClass(p + 1, true, dep)], 1u);  // :846-851
        }
    }
    __syncthreads();
    for (uint32_t c = tid; c < (uint32_t)kFrClasses; c += 1024) {
        int g = 1, pl = 1;
        const int deg = (int)c / kFrDepths / 2;
        const bool incr = ((int)c / kFrDepths) & 1;
        if (L.cCount[c]) frShape(deg, incr, L.cCount[c], &g, &pl);
        L.cG[c] = (uint8_t)g, L.cPlanes[c] = (uint8_t)pl;
        L.cBlocks[c] = L.cCount[c] ? (L.cCount[c] + (uint32_t)g - 1u) / (uint32_t)g : 0u;
    }
    __syncthreads();
    if (tid == 0) {  // prefix over the classes (degree-major): tasks, workgroups, arena rows, samples
        uint32_t t = 0, b = 0;
        uint64_t rows = 0, smp = 0;
        for (int deg = 0; deg <= kMaxDegree; ++deg) {
            const uint32_t t0 = t, b0 = b;
            for (int v = 0; v < 2 * kFrDepths; ++v) {
                const int c = deg * 2 * kFrDepths + v;
                const uint32_t cnt = L.cCount[c];
                L.cFirst[c] = t, L.cCursor[c] = 0, L.cBlockFirst[c] = b, L.cArena[c] = rows, L.cSample[c] = smp;
                if (cnt) {
                    const bool incr = v >= kFrDepths;
                    const uint64_t r = incr ? frCoef(deg) - frCoef(deg - 1) : frCoef(deg);
                    const uint64_t nq = 4 * (uint64_t)deg + 1;
                    t += cnt, b += L.cBlocks[c], rows += r * cnt, smp += nq * nq * nq * cnt;
                }
            }
            h->degTasks[deg][0] = t0, h->degTasks[deg][1] = t - t0;
            h->degBlocks[deg][0] = b0, h->degBlocks[deg][1] = b - b0;
        }
        h->nJobs = nJobs, h->nTasks = t, h->nBlocks = b;
        h->sampleUsed = smp;
        h->fits += t, h->samples += smp;
        sNext = 0;
        // rows of this round start at the arena's current end
        L.tmp[0] = (uint32_t)(rows & 0xFFFFFFFFu), L.tmp[1] = (uint32_t)(rows >> 32);
    }
    __syncthreads();
    const uint64_t arenaBase = h->arenaUsed;
    for (uint32_t j = tid; j < nJobs; j += 1024) {
        const hpsdf_node& n = d.nodes[d.batchIdx[j]];
        const int p = L.jDeg[j], dep = L.jDepth[j];
        const uint32_t slot0 = j * HPSDF_JOB_HEADER_DOUBLES;
        auto emit = [&](int deg, bool incr, const float* bmin, const float* bmax, int depth, uint32_t errSlot, uint32_t slot) -> uint64_t {
            const int c = frClass(deg, incr, depth);
            const uint64_t rows = incr ? frCoef(deg) - frCoef(deg - 1) : frCoef(deg);
            const uint64_t nq = 4 * (uint64_t)deg + 1;
            FitTask& t = d.tasks[L.cFirst[c] + slot];
            for (int a = 0; a < 3; ++a) t.bmin[a] = bmin[a], t.bmax[a] = bmax[a];
            t.outOff = arenaBase + L.cArena[c] + (uint64_t)slot * rows;
            t.copyOff = ~0ull;
            t.sampleOff = L.cSample[c] + (uint64_t)slot * nq * nq * nq;
            t.errSlot = errSlot;
            t.depth = (uint8_t)depth;
            t.pad[0] = (uint8_t)deg, t.pad[1] = t.pad[2] = 0;
            return t.outOff;
        };
        uint64_t pOff = ~0ull, hOff = ~0ull;
        if (L.jCoarse[j]) {
            const uint32_t s = atomicAdd(&L.cCursor[frClass(2, false, dep)], 1u);
            pOff = emit(2, false, n.aabb_min, n.aabb_max, dep, slot0, s);
        } else {
            if (dep < kMaxDepth) {
                const uint32_t s = atomicAdd(&L.cCursor[frClass(p, false, dep + 1)], 8u);
                for (unsigned i = 0; i < 8; ++i) {
                    float cmin[3], cmax[3];
                    for (int a = 0; a < 3; ++a) {  // Octree::CornerAABB, :1096-1112
                        const float mid = (n.aabb_max[a] + n.aabb_min[a]) * 0.5f;
                        cmin[a] = (i >> a) & 1u ? mid : n.aabb_min[a];
                        cmax[a] = (i >> a) & 1u ? n.aabb_max[a] : mid;
                    }
                    const uint64_t o = emit(p, false, cmin, cmax, dep + 1, slot0 + 1 + i, s + i);
                    if (i == 0) hOff = o;
                }
            }
            if (p < kMaxDegree - 1) {
                const uint32_t s = atomicAdd(&L.cCursor[frClass(p + 1, true, dep)], 1u);
                pOff = emit(p + 1, true, n.aabb_min, n.aabb_max, dep, slot0, s);
            }
        }
        d.jobP[j] = pOff, d.jobH[j] = hOff;
    }
    for (int c = 0; c < kFrClasses; ++c) {
        const uint32_t nb = L.cBlocks[c];
        if (!nb) continue;
        const int deg = c / kFrDepths / 2;
        const bool incr = (c / kFrDepths) & 1;
        for (uint32_t b = tid; b < nb; b += 1024) {
            FitBlock fb;
            fb.firstTask = L.cFirst[c] + b * L.cG[c];
            const uint32_t left = L.cCount[c] - b * L.cG[c];
            fb.nTasks = (uint16_t)(left < L.cG[c] ? left : L.cG[c]);
            fb.degree = (uint8_t)deg;
            fb.planesPerChunk = L.cPlanes[c];
            fb.rowStart = (uint16_t)(incr ? frCoef(deg - 1) : 0);
            fb.rowEnd = (uint16_t)frCoef(deg);
            fb.depth = (uint8_t)(c % kFrDepths);
            fb.weighted = 0;
            fb.pad1[0] = fb.pad1[1] = 0;
            d.blocks[L.cBlockFirst[c] + b] = fb;
        }
    }
    for (uint32_t i = tid; i < nJobs * HPSDF_JOB_HEADER_DOUBLES; i += 1024) d.errs[i] = 0.0;
    __syncthreads();
    if (tid == 0) h->arenaUsed = arenaBase + ((uint64_t)L.tmp[0] | ((uint64_t)L.tmp[1] << 32));
}

// ---------------------------------------------------------------------------------------------------------------------
// apply: Octree.cpp:594-601 (decision) and :243-299 (bookkeeping), jobs in node-index order
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fr_apply_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (h->done) return;
    __shared__ uint8_t sKind[kFrJobs];    // 0 dropped, 1 P, 2 H
    __shared__ uint32_t sBase[kFrJobs];   // H: index of the first child
    __shared__ uint32_t sHist[2048];
    __shared__ uint32_t sScan[1024];
    __shared__ double sOps[64 * 9];
    __shared__ uint32_t sCnt[4];          // P, H, dropped, max degree
    __shared__ double sTotal;
    const uint32_t tid = threadIdx.x, nJobs = h->nJobs, nNodes0 = h->nNodes;
    for (uint32_t i = tid; i < 2048; i += 1024) sHist[i] = 0;
    if (tid < 4) sCnt[tid] = 0;
    __syncthreads();
    // ---- decisions
    for (uint32_t j = tid; j < nJobs; j += 1024) {
        const uint32_t idx = d.batchIdx[j];
        const double err = d.batchErr[j];
        const int p = d.nodes[idx].degree, dep = d.nodes[idx].depth;
        const double* e = d.errs + (size_t)j * HPSDF_JOB_HEADER_DOUBLES;
        const double pErr = e[0];
        const bool coarse = fabs(err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
        double pImp, hImp;
        if (coarse) {
            hImp = 0.0;   // :806-810
            pImp = pErr;  // :842
        } else {
            if (dep < kMaxDepth) {
                double maxNewErr = 0.0;
                for (int i = 0; i < 8; ++i) maxNewErr = maxNewErr < e[1 + i] ? e[1 + i] : maxNewErr;  // std::max
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
        sKind[j] = refineP ? 1 : (refineH ? 2 : 0);
    }
    __syncthreads();
    // ---- first child of every H job: node count so far + 8 x (H jobs before it); thread t scans jobs 4t..4t+3
    {
        uint32_t c = 0;
        for (uint32_t j = tid * 4; j < tid * 4 + 4 && j < nJobs; ++j) c += sKind[j] == 2 ? 1u : 0u;
        sScan[tid] = c;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t v = tid >= off ? sScan[tid - off] : 0u;
            __syncthreads();
            sScan[tid] += v;
            __syncthreads();
        }
        uint32_t run = sScan[tid] - c;
        for (uint32_t j = tid * 4; j < tid * 4 + 4 && j < nJobs; ++j) {
            sBase[j] = nNodes0 + 8u * run;
            run += sKind[j] == 2 ? 1u : 0u;
        }
    }
    __syncthreads();
    const uint32_t nH = sScan[1023];
    if (nNodes0 + 8u * nH > d.nodeCap) {  // the host sizes the node arrays for 8 K new nodes per round: cannot happen
        if (tid == 0) h->overflow = 1, h->done = 1;
        return;
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (wave == 0) {
        // ---- the running total, in the reference's order: one lane, one addition after the other.  A P result adds
        //      (newErr - initialErr) (:255); an H result subtracts initialErr once (:268) and adds its 8 children's errors
        //      (:272).  The other lanes stage the operands of 64 jobs at a time, densely, in LDS.
        double total = h->total;
        for (uint32_t base = 0; base < nJobs; base += 64) {
            const uint32_t j = base + lane;
            const int kind = j < nJobs ? sKind[j] : 0;
            const uint32_t nOps = kind == 1 ? 1u : (kind == 2 ? 9u : 0u);
            uint32_t pos = nOps;  // inclusive scan over the lanes
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(pos, off, 64);
                if (lane >= off) pos += v;
            }
            const uint32_t totalOps = __shfl(pos, 63, 64);
            pos -= nOps;
            if (kind == 1) {
                sOps[pos] = d.errs[(size_t)j * 9] - d.batchErr[j];
            } else if (kind == 2) {
                sOps[pos] = d.batchErr[j] * -1.0;  // total -= err  ==  total + (-err)
                for (int i = 0; i < 8; ++i) sOps[pos + 1 + i] = d.errs[(size_t)j * 9 + 1 + i];
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0)
                for (uint32_t q = 0; q < totalOps; ++q) total = total + sOps[q];
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) sTotal = total;
    } else {
        // ---- tree and queue updates: lane group of 8 = one job (its 8 children when it splits)
        const int sub = lane & 7;
        uint32_t nP = 0, nD = 0, maxDeg = 0;
        for (uint32_t j = (uint32_t)(wave - 1) * 8 + (uint32_t)(lane >> 3); j < nJobs; j += 15 * 8) {
            const uint32_t idx = d.batchIdx[j];
            const int kind = sKind[j];
            const int p = d.nodes[idx].degree, dep = d.nodes[idx].depth;
            const bool coarse = fabs(d.batchErr[j] - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
            if (kind == 1) {  // :253-260, :286-290
                if (sub == 0) {
                    const int np = coarse ? 2 : p + 1;
                    if (coarse) d.segFirst[idx] = 2;
                    d.segOff[(size_t)idx * kFrSegs + (np - d.segFirst[idx])] = d.jobP[j];
                    d.nodes[idx].degree = (uint8_t)np;
                    const double pErr = d.errs[(size_t)j * 9];
                    const uint64_t bits = (uint64_t)__double_as_longlong(pErr);
                    d.qErr[idx] = bits;
                    atomicAdd(&sHist[frDigit(0, bits, idx)], 1u);
                    ++nP;
                    maxDeg = (uint32_t)np > maxDeg ? (uint32_t)np : maxDeg;
                }
            } else if (kind == 2) {  // :262-279, :286-290; Octree::Subdivide :1115-1128
                const uint32_t c0 = sBase[j], ch = c0 + (uint32_t)sub;
                const hpsdf_node par = d.nodes[idx];
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
                d.segOff[(size_t)ch * kFrSegs] = d.jobH[j] + (uint64_t)sub * frCoef(p);
                const double hErr = d.errs[(size_t)j * 9 + 1 + sub];
                const uint64_t bits = (uint64_t)__double_as_longlong(hErr);
                d.qErr[ch] = bits;
                atomicAdd(&sHist[frDigit(0, bits, ch)], 1u);
                __builtin_amdgcn_wave_barrier();
                if (sub == 0) {  // after the parent has been read by all eight lanes
                    d.nodes[idx].child_idx = c0;
                    d.nodes[idx].degree = kInteriorDegree;
                    d.nodes[idx].coeffs_start = 0;
                }
            } else if (sub == 0) {
                ++nD;  // :643-655: keeps its basis, never queued again
            }
        }
        if (nP) atomicAdd(&sCnt[0], nP);
        if (nD) atomicAdd(&sCnt[2], nD);
        if (maxDeg) atomicMax(&sCnt[3], maxDeg);
    }
    __syncthreads();
    for (uint32_t i = tid; i < 2048; i += 1024) {
        if (sHist[i]) h->hist1[i] += sHist[i];
        h->hist2[i] = 0;
    }
    if (tid == 0) {
        const uint32_t nP = sCnt[0], nD = sCnt[2];
        const uint32_t nQ = h->nQueued - (h->round == 0 ? 0u : nJobs) + nP + 8u * nH;
        // (round 0's batch was never counted in nQueued: the initialisation hands it over directly)
        h->nQueued = nQ;
        h->nNodes = nNodes0 + 8u * nH;
        h->total = sTotal;
        h->jobs += nJobs, h->pRefines += nP, h->hRefines += nH, h->dropped += nD;
        h->nLeaves += 7u * nH;
        if (sCnt[3] > h->maxDegree) h->maxDegree = sCnt[3];
        h->round += 1;
        h->takenCount = 0, h->candCount = 0, h->above = 0, h->t1 = -1;
        h->done = (sTotal < h->target || nQ == 0) ? 1u : 0u;  // Octree.cpp:216
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ReallocCoeffs (Octree.cpp:474-555) once the stop rule has fired
// ---------------------------------------------------------------------------------------------------------------------
// coefficients below every interior node, level by level from the deepest (one workgroup; children sit at depth + 1)
__global__ __launch_bounds__(1024) void fr_totals_kernel(FrDev d) {
    FrHdr* h = d.hdr;
    if (!h->done || h->overflow) return;
    const uint32_t n = h->nNodes;
    for (int dep = kMaxDepth; dep >= 0; --dep) {
        for (uint32_t i = threadIdx.x; i < n; i += 1024) {
            const hpsdf_node& nd = d.nodes[i];
            if (nd.depth != dep || nd.degree != kInteriorDegree) continue;
            uint32_t s = 0;
            for (unsigned c = 0; c < 8; ++c) {
                const hpsdf_node& ch = d.nodes[nd.child_idx + c];
                s += ch.degree == kInteriorDegree ? d.sub[nd.child_idx + c] : frCoef(ch.degree);
            }
            d.sub[i] = s;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint64_t total = d.sub[0];
        h->nCoeffs = total;
        if (total > d.storeCap) h->overflow = 2;  // the host grows the store and runs the two kernels again
    }
}

// One wave per node.  A leaf's coeffsStart = coefficients of everything the depth-first walk (children 0..7 from the
// root) visits before it = over its ancestors-or-self a: the subtree sizes of a's earlier siblings.  Its rows are then
// gathered from the arena, segment by segment.
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
            const double* src = d.arena + d.segOff[(size_t)i * kFrSegs + s];
            for (uint32_t r = r0 + (uint32_t)lane; r < r1; r += 64) d.store[(size_t)start + r] = src[r - r0];
        }
    }
}

// per-build initialisation: the uniformly refined tree (a copy of the context's template), the header, round 0's batch
__global__ __launch_bounds__(256) void fr_init_kernel(FrDev d, const hpsdf_node* tmplNodes, const uint32_t* tmplParent,
                                                      const uint32_t* tmplLeaves, uint32_t nTmpl, uint32_t nLeaves) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nTmpl) {
        d.nodes[i] = tmplNodes[i];
        d.parent[i] = tmplParent[i];
        d.qErr[i] = kNotQueued;
        d.segFirst[i] = 2;
    }
    if (i < nLeaves) {
        d.batchIdx[i] = tmplLeaves[i];
        d.batchErr[i] = HPSDF_INITIAL_NODE_ERR;
    }
    uint32_t* hw = reinterpret_cast<uint32_t*>(d.hdr);
    for (uint32_t w = i; w < sizeof(FrHdr) / 4; w += gridDim.x * 256u) hw[w] = 0;
}
// (a kernel of its own: the stores below must come after every workgroup's zeroing above)
__global__ void fr_init_hdr_kernel(FrDev d, uint32_t nTmpl, uint32_t nLeaves, double target) {
    FrHdr* h = d.hdr;
    h->nNodes = nTmpl;
    h->nJobs = nLeaves;
    h->nLeaves = nLeaves;
    h->t1 = -1;
    h->total = 4096.0 * HPSDF_INITIAL_NODE_ERR;  // pow(8, 4) * INITIAL_NODE_ERR, Octree.cpp:212
    h->target = target;
    h->maxDegree = 2;
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
    // template of the uniformly refined tree (Octree::UniformlyRefine, :112-191)
    hpsdf_node* tmplNodes = nullptr;
    uint32_t* tmplParent = nullptr;
    uint32_t* tmplLeaves = nullptr;
    uint32_t nTmpl = 0, nTmplLeaves = 0;
    char* pinned = nullptr;  // staging of the finished block
    size_t pinnedCap = 0;

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
        if (e == hipSuccess) e = grow(&d.sub, nodeCap, nc, s, false);
        if (e == hipSuccess) e = grow(&d.candA, nodeCap, nc, s, false);
        if (e == hipSuccess) e = grow(&d.candB, nodeCap, nc, s, false);
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
    hipError_t init(int dev, hipStream_t s) {
        device = dev;
        hipError_t e = hipMalloc((void**)&d.hdr, sizeof(FrHdr));
        if (e == hipSuccess) e = hipHostMalloc((void**)&hostHdr, sizeof(FrHdr), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void**)&d.taken, (kFrJobs + 64) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.batchIdx, kFrJobs * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.batchErr, kFrJobs * sizeof(double));
        if (e == hipSuccess) e = hipMalloc((void**)&d.jobP, kFrJobs * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.jobH, kFrJobs * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void**)&d.tasks, kFrTasks * sizeof(FitTask));
        if (e == hipSuccess) e = hipMalloc((void**)&d.blocks, kFrTasks * sizeof(FitBlock));
        if (e == hipSuccess) e = hipMalloc((void**)&d.errs, (size_t)kFrJobs * HPSDF_JOB_HEADER_DOUBLES * sizeof(double));
        if (e != hipSuccess) return e;
        // the uniformly refined tree, from the host scheduler's own initialisation (builderBegin): identical indices
        hpsdf_build b;
        hpsdf_config cfg;
        hpsdf_config_default(&cfg);
        cfg.thread_count = 1;
        if (builderBegin(&b, &cfg, nullptr) != HPSDF_OK) return hipErrorUnknown;
        nTmpl = (uint32_t)b.nodes.size();
        std::vector<uint32_t> parent(nTmpl, 0), leaves;
        for (uint32_t i = 0; i < nTmpl; ++i) {
            if (b.nodes[i].child_idx != ~0ull)
                for (unsigned c = 0; c < 8; ++c) parent[b.nodes[i].child_idx + c] = i;
            else
                leaves.push_back(i);
        }
        nTmplLeaves = (uint32_t)leaves.size();
        if (nTmplLeaves > kFrJobs) return hipErrorUnknown;
        e = hipMalloc((void**)&tmplNodes, nTmpl * sizeof(hpsdf_node));
        if (e == hipSuccess) e = hipMalloc((void**)&tmplParent, nTmpl * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&tmplLeaves, nTmplLeaves * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(tmplNodes, b.nodes.data(), nTmpl * sizeof(hpsdf_node), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(tmplParent, parent.data(), nTmpl * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(tmplLeaves, leaves.data(), nTmplLeaves * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = ensureNodes(65536, s);
        if (e == hipSuccess) e = ensureArena(1ull << 22, 0, s);
        if (e == hipSuccess) e = ensureStore(1ull << 20, s);
        if (e == hipSuccess) e = ensurePinned(4u << 20);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)fr_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFrBatchLdsBytes);
        return e;
    }
    ~FrontierWorkspace() {
        if (device >= 0) (void)hipSetDevice(device);
        for (void* p : {(void*)d.hdr, (void*)d.nodes, (void*)d.qErr, (void*)d.parent, (void*)d.segOff, (void*)d.segFirst, (void*)d.sub,
                        (void*)d.taken, (void*)d.candA, (void*)d.candB, (void*)d.batchIdx, (void*)d.batchErr, (void*)d.jobP, (void*)d.jobH,
                        (void*)d.tasks, (void*)d.blocks, (void*)d.errs, (void*)d.store, (void*)arena, (void*)samples, (void*)tmplNodes,
                        (void*)tmplParent, (void*)tmplLeaves})
            if (p) (void)hipFree(p);
        if (hostHdr) (void)hipHostFree(hostHdr);
        if (pinned) (void)hipHostFree(pinned);
    }
};

bool frontierEligible(const hpsdf_config* cfg, const hpsdf_field* field, uint64_t K) {
    if (const char* e = std::getenv("HPSDF_HOST_FRONTIER"))
        if (e[0] == '1') return false;
    if (cfg->weighting_type != 0) return false;   // the weight is pow/exp of the host's libm
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

int frontierCreate(hpsdf_ctx* ctx, const hpsdf_config* cfgIn, const hpsdf_field* field, uint64_t K, void** block, size_t* size,
                   hpsdf_build_stats* stats) {
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
    std::shared_ptr<FrontierWorkspace> wsKeep;
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
    ws->d.K = Kj;
    const bool mesh = innermost(field)->kind == kHostMesh;

    FieldDev fd;
    int rc;
    if ((rc = makeFieldDev(field, nullptr, &fd))) return rc;
    RootMap rm;
    for (int a = 0; a < 3; ++a) {
        rm.bounds[a] = (double)(cfg.root_max[a] - cfg.root_min[a]);          // Octree.cpp:324
        rm.centre[a] = (double)((cfg.root_min[a] + cfg.root_max[a]) / 2.0f);  // Octree.cpp:322
    }
    // launch-time LDS of a fit launch of `deg`: the largest shape the device may pick
    auto fitLds = [](int deg) {
        size_t m = 0;
        for (int incr = 0; incr < 2; ++incr) {
            if (incr && deg == 0) continue;
            for (uint32_t count : {1u, 512u, 1024u, 2048u, 4096u, 8192u, 16384u, 40000u}) {
                int g, pl;
                frShape(deg, incr != 0, count, &g, &pl);
                for (int gg = 1; gg <= g; ++gg) {
                    int pp = 4 * deg + 1;
                    while (pp > 1 && frLds(deg, gg, pp) > kFitChunkLdsBytes) --pp;
                    m = std::max(m, frLds(deg, gg, pp));
                }
            }
        }
        return m;
    };
    auto rowsPerJob = [](int pmax) {  // arena rows one job can need when no leaf exceeds degree pmax
        const int p = std::min(pmax, kMaxDegree - 1);
        return (uint64_t)8 * frCoef(p) + frCoef(std::min(p + 1, kMaxDegree));
    };
    auto samplesPerJob = [](int pmax) {
        const uint64_t a = 4 * (uint64_t)std::min(pmax, kMaxDegree - 1) + 1, b = a + 4;
        return 8 * a * a * a + b * b * b;
    };

    FrDev& d = ws->d;
    hipLaunchKernelGGL(fr_init_kernel, dim3((ws->nTmpl + 255) / 256), dim3(256), 0, s, d, ws->tmplNodes, ws->tmplParent, ws->tmplLeaves,
                       ws->nTmpl, ws->nTmplLeaves);
    hipLaunchKernelGGL(fr_init_hdr_kernel, dim3(1), dim3(1), 0, s, d, ws->nTmpl, ws->nTmplLeaves, cfg.target_error_threshold);
    uint32_t knownNodes = ws->nTmpl, knownMaxDeg = 2;
    uint64_t knownArena = 0;
    double tSync = 0;
    int rounds = 0;
    const FrHdr* hh = ws->hostHdr;
    for (;; ++rounds) {
        const uint32_t jobsBound = rounds == 0 ? ws->nTmplLeaves : Kj;
        // capacities for this round (the device flags what the host failed to foresee; it cannot happen by these bounds)
        hipError_t e = ws->ensureNodes(knownNodes + 8u * jobsBound, s);
        if (e == hipSuccess) e = ws->ensureArena(knownArena + (uint64_t)jobsBound * rowsPerJob((int)knownMaxDeg), knownArena, s);
        if (e == hipSuccess && mesh) {
            const uint64_t need = (uint64_t)jobsBound * (rounds == 0 ? 729ull : samplesPerJob((int)knownMaxDeg));
            if (need > (1ull << 31)) return fail(HPSDF_ERR_UNSUPPORTED, "round too large for the sampled mesh path");
            e = ws->ensureSamples(need, s);
        }
        if (e != hipSuccess) return hipFail(e, "frontier buffers");
        if (rounds > 0) hipLaunchKernelGGL(fr_select_kernel, dim3(std::min<uint32_t>(1024u, (knownNodes + 8u * Kj + 255u) / 256u)), dim3(256), 0, s, d);
        hipLaunchKernelGGL(fr_batch_kernel, dim3(1), dim3(1024), kFrBatchLdsBytes, s, d, rounds == 0 ? 1 : 0);
        const int degLo = 2, degHi = rounds == 0 ? 2 : (int)std::min<uint32_t>(kMaxDegree, knownMaxDeg + 1);
        const uint32_t taskBound = rounds == 0 ? ws->nTmplLeaves : 9u * Kj;
        FieldDev fdr = fd;
        if (mesh) {
            for (int deg = degLo; deg <= degHi; ++deg)
                HPSDF_HIP(launchMeshSampleRange(s, d.tasks, &d.hdr->degTasks[deg][0], std::min<uint32_t>(taskBound, 65535u), deg, ctx->dTables, fd,
                                                rm, ws->samples));
            fdr.kind = kFieldSamples;
            fdr.samples = ws->samples;
        }
        for (int deg = degLo; deg <= degHi; ++deg)
            HPSDF_HIP(launchFit(s, deg <= 5 ? deg : 0, 1, d.blocks, taskBound, fitLds(deg), d.tasks, ws->arena, d.errs, nullptr, ctx->dTables, fdr,
                                rm, &d.hdr->degBlocks[deg][0]));
        hipLaunchKernelGGL(fr_apply_kernel, dim3(1), dim3(1024), 0, s, d);
        hipLaunchKernelGGL(fr_totals_kernel, dim3(1), dim3(1024), 0, s, d);
        hipLaunchKernelGGL(fr_store_kernel, dim3(std::min<uint32_t>(2048u, (knownNodes + 8u * jobsBound + 3u) / 4u)), dim3(256), 0, s, d);
        HPSDF_HIP(hipMemcpyAsync(ws->hostHdr, d.hdr, offsetof(FrHdr, hist1), hipMemcpyDeviceToHost, s));
        const double ts = now();
        HPSDF_HIP(hipStreamSynchronize(s));
        tSync += now() - ts;
        if (hh->overflow == 1) return fail(HPSDF_ERR_STATE, "frontier: node capacity exceeded");
        knownNodes = hh->nNodes, knownMaxDeg = hh->maxDegree, knownArena = hh->arenaUsed;
        if (hh->done) break;
    }
    if (hh->overflow == 2) {  // the packed store was too small: grow, run ReallocCoeffs again
        hipError_t e = ws->ensureStore(hh->nCoeffs, s);
        if (e != hipSuccess) return hipFail(e, "coefficient store");
        HPSDF_HIP(hipMemsetAsync(&d.hdr->overflow, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(fr_totals_kernel, dim3(1), dim3(1024), 0, s, d);
        hipLaunchKernelGGL(fr_store_kernel, dim3(std::min<uint32_t>(2048u, (knownNodes + 3u) / 4u)), dim3(256), 0, s, d);
        HPSDF_HIP(hipMemcpyAsync(ws->hostHdr, d.hdr, offsetof(FrHdr, hist1), hipMemcpyDeviceToHost, s));
        HPSDF_HIP(hipStreamSynchronize(s));
        if (hh->overflow) return fail(HPSDF_ERR_STATE, "frontier: coefficient store overflow");
    }
    // Octree::ToMemoryBlock, Octree.cpp:424-456: [u64 nCoeffs][f64 x nCoeffs][u64 nNodes][Node x nNodes][Config]
    const uint64_t nc = hh->nCoeffs, nn = hh->nNodes;
    const size_t bytes = 8 + 8 * (size_t)nc + 8 + sizeof(hpsdf_node) * (size_t)nn + sizeof(hpsdf_config);
    uint8_t* p = (uint8_t*)std::malloc(bytes);
    if (!p) return fail(HPSDF_ERR_OUT_OF_MEMORY, "malloc of the memory block failed");
    const double tc = now();
    hipError_t e = ws->ensurePinned(bytes);
    if (e == hipSuccess && nc) e = hipMemcpyAsync(ws->pinned, d.store, 8 * (size_t)nc, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ws->pinned + 8 * (size_t)nc, d.nodes, sizeof(hpsdf_node) * (size_t)nn, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        std::free(p);
        return hipFail(e, "block download");
    }
    const double tcopy = now() - tc;
    std::memcpy(p, &nc, 8);
    std::memcpy(p + 8, ws->pinned, 8 * (size_t)nc);
    std::memcpy(p + 8 + 8 * (size_t)nc, &nn, 8);
    std::memcpy(p + 16 + 8 * (size_t)nc, ws->pinned + 8 * (size_t)nc, sizeof(hpsdf_node) * (size_t)nn);
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
        std::fprintf(stderr, "[frontierCreate] us: total %.0f (waiting for the device %.0f over %d rounds, block download %.0f)\n", now() - t0, tSync,
                     rounds + 1, tcopy);
    return HPSDF_OK;
}

}  // namespace hpsdf
