#include "builder.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "launch.hpp"

using namespace hpsdf;

namespace {

const uint64_t kNone = ~0ull;

inline uint64_t ncoef(int p) { return tables().coeffCount[p]; }

// strict total order of the frontier: larger error first, then smaller node index
inline bool before(const hpsdf_build::HeapEnt& a, const hpsdf_build::HeapEnt& b) {
    return a.err > b.err || (a.err == b.err && a.idx < b.idx);
}
struct HeapLess {  // std heap keeps the "largest" on top: largest = first in the order above
    bool operator()(const hpsdf_build::HeapEnt& a, const hpsdf_build::HeapEnt& b) const { return before(b, a); }
};

void initNode(hpsdf_node& n) {  // Source/HP/Node.cpp:5-15, padding zeroed
    std::memset(&n, 0, sizeof(n));
    n.child_idx = kNone;
    for (int a = 0; a < 3; ++a) {
        n.aabb_min[a] = FLT_MAX;
        n.aabb_max[a] = -FLT_MAX;
    }
    n.degree = kInteriorDegree;
    n.depth = kMaxDepth + 1;
}

// Octree::CornerAABB, Octree.cpp:1096-1112: f32 midpoint split, bit d of i selects the upper half on axis d
void cornerBox(const float* bmin, const float* bmax, unsigned i, float* omin, float* omax) {
    for (int d = 0; d < 3; ++d) {
        const float mid = (bmax[d] + bmin[d]) * 0.5f;
        omin[d] = (i >> d) & 1u ? mid : bmin[d];
        omax[d] = (i >> d) & 1u ? bmax[d] : mid;
    }
}

// Octree::Subdivide, Octree.cpp:1115-1128: children appended as a block of 8
void subdivide(hpsdf_build* b, uint64_t idx) {
    const uint64_t first = b->nodes.size();
    b->nodes[idx].child_idx = first;
    b->nodes.resize(first + 8);
    b->segHead.resize(first + 8, -1);
    b->segTail.resize(first + 8, -1);
    for (unsigned i = 0; i < 8; ++i) {
        hpsdf_node& c = b->nodes[first + i];
        initNode(c);
        cornerBox(b->nodes[idx].aabb_min, b->nodes[idx].aabb_max, i, c.aabb_min, c.aabb_max);
        c.depth = (uint8_t)(b->nodes[idx].depth + 1);
    }
}

// Octree::UniformlyRefine, Octree.cpp:112-191: depth-first, a node is split when first reached,
// depth-4 cells become degree-0 leaves queued with the initial error
void uniformRefine(hpsdf_build* b, uint64_t idx, int depth) {
    if (depth < 4) {
        subdivide(b, idx);
        const uint64_t c = b->nodes[idx].child_idx;
        for (unsigned i = 0; i < 8; ++i) uniformRefine(b, c + i, depth + 1);
    } else {
        b->nodes[idx].degree = 0;
        b->heap.push_back({idx, HPSDF_INITIAL_NODE_ERR});
    }
}

void appendSeg(hpsdf_build* b, uint64_t node, const hpsdf_build::Seg& s) {
    b->segs.push_back(s);
    const int64_t id = (int64_t)b->segs.size() - 1;
    if (b->segHead[node] < 0)
        b->segHead[node] = id;
    else
        b->segs[b->segTail[node]].next = id;
    b->segTail[node] = id;
}

}  // namespace

hpsdf_build::~hpsdf_build() {
    if (ws) {
        if (ownsWs) {
            ws->release();
            delete ws;
        } else {
            ws->inUse = false;
        }
    }
}

namespace {
int acquireWorkspace(hpsdf_build* b, hpsdf_ctx* ctx) {
    if (b->ws) return HPSDF_OK;
    if (!ctx->ws.inUse) {
        ctx->ws.inUse = true;
        ctx->ws.device = ctx->device;
        b->ws = &ctx->ws;
        b->ownsWs = false;
    } else {
        b->ws = new Workspace();
        b->ws->device = ctx->device;
        b->ownsWs = true;
    }
    return HPSDF_OK;
}
}  // namespace

namespace hpsdf {

// flop-proportional cost of one job, used only to balance the slices
uint64_t jobCost(int degree, int depth, bool coarse) {
    auto cube = [](uint64_t n) { return n * n * n; };
    if (coarse) return ncoef(2) * cube(9);
    uint64_t c = 0;
    if (depth < kMaxDepth) c += 8 * ncoef(degree) * cube(4 * degree + 1);
    if (degree < kMaxDegree - 1) c += (ncoef(degree + 1) - ncoef(degree)) * cube(4 * degree + 5);
    return c ? c : 1;
}

int builderBegin(hpsdf_build* b, const hpsdf_config* cfg, const hpsdf_build_opts* opts) {
    // Config::IsValid, Source/HP/Config.cpp:17-32 (asserts there, status codes here)
    if (!(cfg->target_error_threshold > 0.0)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "targetErrorThreshold must be > 0");
    if (cfg->thread_count == 0) return fail(HPSDF_ERR_INVALID_ARGUMENT, "threadCount must be > 0");
    {
        float vol = 1.0f;
        for (int a = 0; a < 3; ++a) vol *= (cfg->root_max[a] - cfg->root_min[a]);
        if (!(vol > 0.0f)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "root volume must be > 0");
    }
    if (cfg->weighting_type > 2) return fail(HPSDF_ERR_INVALID_ARGUMENT, "unknown nearnessWeighting.type");
    if (cfg->weighting_type != 0) {
        if (!(cfg->weighting_strength > 0.0)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "nearnessWeighting.strength must be > 0");
    }
    b->weighted = cfg->weighting_type != 0;
    b->cfg = *cfg;
    std::memset(b->cfg.pad0, 0, sizeof b->cfg.pad0);
    std::memset(b->cfg.pad1, 0, sizeof b->cfg.pad1);
    std::memset(b->cfg.pad2, 0, sizeof b->cfg.pad2);
    if (opts) {
        if (opts->world < 1 || opts->rank < 0 || opts->rank >= opts->world)
            return fail(HPSDF_ERR_INVALID_ARGUMENT, "bad rank/world");
        b->K = opts->max_jobs_per_round ? opts->max_jobs_per_round : HPSDF_DEFAULT_JOBS_PER_ROUND;
        b->rank = opts->rank;
        b->world = opts->world;
    }
    // CreateRoot, Octree.cpp:792-801
    b->nodes.resize(1);
    b->segHead.assign(1, -1);
    b->segTail.assign(1, -1);
    initNode(b->nodes[0]);
    b->nodes[0].depth = 0;
    for (int a = 0; a < 3; ++a) {
        b->nodes[0].aabb_min[a] = -0.5f;
        b->nodes[0].aabb_max[a] = 0.5f;
    }
    subdivide(b, 0);
    for (unsigned i = 0; i < 8; ++i) uniformRefine(b, b->nodes[0].child_idx + i, 1);
    std::make_heap(b->heap.begin(), b->heap.end(), HeapLess());
    b->total = std::pow(8, 4) * HPSDF_INITIAL_NODE_ERR;  // Octree.cpp:212
    b->slices.assign(b->world, {0, 0});
    return HPSDF_OK;
}

int builderSelect(hpsdf_build* b, uint64_t* nJobs) {
    *nJobs = 0;
    if (b->roundOpen) return fail(HPSDF_ERR_STATE, "previous round not applied");
    if (b->finished) return HPSDF_OK;
    // A field that is NaN or infinite at a sample poisons that cell's error and the running total for good: the reference's loop
    // (`finished = total < threshold || queue.empty()`, :216) then refines until memory ends.  Fail instead, after the first round.
    if (!(std::fabs(b->total) <= DBL_MAX))
        return fail(HPSDF_ERR_INVALID_ARGUMENT, "the field is not a finite number at some sample point (the build's total error is NaN or infinite)");
    if (b->total < b->cfg.target_error_threshold || b->heap.empty()) {  // Octree.cpp:216
        b->finished = true;
        return HPSDF_OK;
    }
    const uint64_t want = b->stats.rounds == 0 ? b->heap.size() : std::min<uint64_t>(b->K, b->heap.size());
    b->batch.resize(want);
    bool sorted = false;
    if (want == b->heap.size()) {  // the whole frontier (always the case in round 0): no heap work needed
        if (b->stats.rounds == 0) {
            // round 0 is every leaf of the uniformly refined tree with the initial error: listing them by node index
            // gives the batch already in order (sorting 4096 heap-ordered entries was 12 % of a Create at 1e-5)
            bool same = true;
            for (uint64_t j = 0; j < want && same; ++j) same = std::fabs(b->heap[j].err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
            uint64_t k = 0;
            for (uint64_t i = 0; i < b->nodes.size() && same; ++i)
                if (b->nodes[i].child_idx == kNone) {
                    same = k < want;
                    if (same) {
                        b->batch[k] = b->heap[0];  // (err, any other field) as pushed
                        b->batch[k++].idx = i;
                    }
                }
            sorted = same && k == want;
        }
        if (!sorted) b->batch.assign(b->heap.begin(), b->heap.end());
        b->heap.clear();
    } else {
        for (uint64_t i = 0; i < want; ++i) {
            std::pop_heap(b->heap.begin(), b->heap.end(), HeapLess());
            b->batch[i] = b->heap.back();
            b->heap.pop_back();
        }
    }
    if (!sorted)
        std::sort(b->batch.begin(), b->batch.end(),
                  [](const hpsdf_build::HeapEnt& x, const hpsdf_build::HeapEnt& y) { return x.idx < y.idx; });
    // contiguous cost-balanced slices, the same on every rank
    std::vector<uint64_t> prefix(want + 1, 0);
    for (uint64_t j = 0; j < want; ++j) {
        const hpsdf_node& n = b->nodes[b->batch[j].idx];
        const bool coarse = std::fabs(b->batch[j].err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
        prefix[j + 1] = prefix[j] + jobCost(n.degree, n.depth, coarse);
    }
    uint64_t start = 0;
    for (int r = 0; r < b->world; ++r) {
        uint64_t end = want;
        if (r + 1 < b->world) {
            const uint64_t target = (uint64_t)(((unsigned __int128)prefix[want] * (uint64_t)(r + 1)) / (uint64_t)b->world);
            end = (uint64_t)(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());
            end = std::min(std::max(end, start), want);
        }
        b->slices[r] = {start, end - start};
        start = end;
    }
    b->jobOut.assign(b->slices[b->rank].count, hpsdf_build::JobOut());
    b->roundOpen = true;
    b->computed = false;
    *nJobs = want;
    return HPSDF_OK;
}

int builderJobs(const hpsdf_build* b, hpsdf_job* out) {
    if (!b->roundOpen) return fail(HPSDF_ERR_STATE, "no open round");
    for (size_t j = 0; j < b->batch.size(); ++j) {
        const hpsdf_node& n = b->nodes[b->batch[j].idx];
        hpsdf_job& o = out[j];
        std::memset(&o, 0, sizeof o);
        o.node_idx = b->batch[j].idx;
        for (int a = 0; a < 3; ++a) {
            o.aabb_min[a] = n.aabb_min[a];
            o.aabb_max[a] = n.aabb_max[a];
        }
        o.err = b->batch[j].err;
        o.degree = n.degree;
        o.depth = n.depth;
        o.coarse = std::fabs(o.err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;  // Octree.cpp:806,831
    }
    return HPSDF_OK;
}

// The round's sample buffer (mesh fields: F at every sample; split fits: the values the exact kernel hands to the lower rows' kernel),
// sized to the need -- gigabytes at high degrees -- not to the next power of two.
static hipError_t ensureSampleBuffer(hpsdf_ctx* ctx, Workspace& ws, uint64_t need) {
    if (need <= ws.meshSamplesCap) return hipSuccess;
    const uint64_t nc = (need + (1ull << 20) - 1) & ~((1ull << 20) - 1);
    hipError_t e = hipStreamSynchronize(ctx->stream);  // an earlier round's kernels may still read the old buffer
    if (e != hipSuccess) return e;
    if (ws.meshSamples) (void)hipFree(ws.meshSamples);
    ws.meshSamples = nullptr, ws.meshSamplesCap = 0;
    e = hipMalloc((void**)&ws.meshSamples, nc * sizeof(double));
    if (e == hipSuccess) ws.meshSamplesCap = nc;
    return e;
}

// -----------------------------------------------------------------------------------------------
// GPU leg of a round: every job of this rank's slice becomes 1 (coarse) or up to 9 cell fits
// (EstimateHImprovement: 8 children from scratch, Octree.cpp:814-822; EstimatePImprovement: the
// new rows of degree p+1, :846-851).
// -----------------------------------------------------------------------------------------------
int builderCompute(hpsdf_build* b, hpsdf_ctx* ctx, const hpsdf_field* field) {
    if (!b->roundOpen) return fail(HPSDF_ERR_STATE, "no open round");
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "round_compute needs a device context (there is no CPU path)");
    if (!field) return fail(HPSDF_ERR_INVALID_ARGUMENT, "field is null");
    HPSDF_HIP(hipSetDevice(ctx->device));
    acquireWorkspace(b, ctx);
    Workspace& ws = *b->ws;
    const Tables& T = tables();
    const hpsdf_build::Slice sl = b->slices[b->rank];
    const bool sampled = innermost(field)->kind == kHostCallback;
    // Mesh fields are sampled by their own kernel, wave by wave, and fitted from the samples like a callback field;
    // past kMeshSampleCap samples in one round the fit kernel samples for itself (one workgroup per cell).
    constexpr uint64_t kMeshSampleCap = 1ull << 30;  // 8 GB of f64
    // (HPSDF_MESH_FUSED=1 forces that path: the two must build identical trees, tests/test_gpu_parity.py)
    bool meshSampled = innermost(field)->kind == kHostMesh;
    if (meshSampled) {
        const char* e = std::getenv("HPSDF_MESH_FUSED");
        if (e && e[0] == '1') meshSampled = false;
        if (meshFaceRuleReference(ctx)) meshSampled = false;  // (the sampler's shared traversal assumes the default face rule)
    }

    // ---- pass 1: count the fits of every shape.  A class = (degree, from-scratch | incremental, depth):
    //      all cells of a workgroup share these, so the kernel forms each basis product once.
    constexpr int kDepths = kMaxDepth + 2;
    constexpr int kClasses = 2 * (kMaxDegree + 1) * kDepths;
    auto classOf = [](int degree, bool incremental, int depth) { return (2 * degree + (incremental ? 1 : 0)) * kDepths + depth; };
    std::vector<uint32_t> classCount(kClasses, 0);
    for (uint64_t jl = 0; jl < sl.count; ++jl) {
        const hpsdf_build::HeapEnt& e = b->batch[sl.first + jl];
        const hpsdf_node& n = b->nodes[e.idx];
        if (std::fabs(e.err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON) {
            classCount[classOf(2, false, n.depth)] += 1;  // Octree.cpp:836-843: degree-2 fit from scratch
            continue;
        }
        if (n.depth < kMaxDepth) classCount[classOf(n.degree, false, n.depth + 1)] += 8;          // :814-822
        if (n.degree < kMaxDegree - 1) classCount[classOf(n.degree + 1, true, n.depth)] += 1;  // :846-851
    }
    std::vector<uint32_t> classFirst(kClasses + 1, 0);
    for (int c = 0; c < kClasses; ++c) classFirst[c + 1] = classFirst[c] + classCount[c];
    const uint32_t nTasks = classFirst[kClasses];
    {   // hpsdf_ctx_set_build_limits: this round against the context's bounds, before anything is allocated for it (the tree lives in
        // host memory with this scheduler: the device holds the coefficient arena and the round's samples)
        uint64_t rows = 0, samples = 0;
        for (int c = 0; c < kClasses; ++c) {
            const int deg = c / kDepths / 2;
            const uint64_t nq = 4 * (uint64_t)deg + 1;
            rows += (uint64_t)classCount[c] * T.coeffCount[deg];
            samples += (uint64_t)classCount[c] * nq * nq * nq;
        }
        if (!sampled && !meshSampled) samples = 0;  // (the hand-over buffer of split fits is optional: ensureSampleBuffer's failure makes the round exact)
        const uint64_t bytes = (b->arenaUsed + rows + std::min<uint64_t>(samples, 1ull << 31)) * sizeof(double);
        const uint64_t held = (ws.arenaCap + ws.meshSamplesCap) * sizeof(double);
        const uint64_t growBytes = (b->arenaUsed + rows) * sizeof(double);  // (what grows with the tree: the default limit's subject)
        const int lrc = checkBuildLimits(ctx, b->nodes.size(), bytes, growBytes, held, &b->measuredLimit, b->stats.rounds, b->total, b->cfg.target_error_threshold);
        if (lrc) return lrc;
    }
    if (meshSampled) {
        uint64_t need = 0;
        for (int c = 0; c < kClasses; ++c) {
            const uint64_t nq = 4 * (uint64_t)(c / kDepths / 2) + 1;
            need += classCount[c] * nq * nq * nq;
        }
        meshSampled = need <= kMeshSampleCap;
    }
    const bool meshFused = innermost(field)->kind == kHostMesh && !meshSampled;
    // opt-in fast fit (hpsdf_ctx_set_fast_fit): degrees >= 4 of unweighted, non-CSG fields go to the matrix cores
    const bool fastOn = ctx->fitMode == HPSDF_FIT_FAST && !b->weighted && !meshFused && field->kind != kHostTreeCsg;
    auto fastDeg = [&](int deg) { return fastOn && deg >= 4 && deg <= 11; };
    // the default (HPSDF_FIT_SPLIT): a from-scratch fit of degree >= splitMinDegree (6 unless set otherwise) is cut in two -- its rows of top degree by the bit-exact kernel
    // (they alone enter the error, Octree.cpp:1062-1069), the rows below them by fit_low_kernel (HPSDF_LOW_KERNEL=mfma: fit_mfma_low_kernel) from the same samples.  Not for
    // weighted builds (the weight reads every row) nor for mesh fits that sample inside the fit kernel; a round whose samples would
    // not fit the sample buffer's 16 GB is fitted exactly throughout.
    // (round 0 -- every coarse cell's degree-2 fit -- is never split: the device-side frontier fits it through its own path, and the two
    // schedulers promise the same bytes in the same mode)
    bool splitOn = ctx->fitMode == HPSDF_FIT_SPLIT && !b->weighted && !meshFused && b->stats.rounds > 0;
    if (splitOn) {
        uint64_t need = 0;
        bool any = false;
        for (int c = 0; c < kClasses; ++c) {
            const int deg = c / kDepths / 2;
            const uint64_t nq = 4 * (uint64_t)deg + 1;
            need += classCount[c] * nq * nq * nq;
            any |= classCount[c] && !((c / kDepths) & 1) && fitSplitSupports(deg, ctx->splitMinDegree);
        }
        splitOn = any && need <= (1ull << 31);
        // the hand-over buffer of the split fits, BEFORE the workgroup table says "split": if there is no room for it the round is fitted
        // exactly (the exact fit needs no such buffer) instead of failing the build
        if (splitOn && !sampled && !meshSampled && ensureSampleBuffer(ctx, ws, need) != hipSuccess) {
            (void)hipGetLastError();
            splitOn = false;
        }
    }
    auto splitClass = [&](int deg, bool incr) { return splitOn && !incr && fitSplitSupports(deg, ctx->splitMinDegree); };
    b->stats.fit_mode = (uint64_t)ctx->fitMode;
    for (int c = 0; c < kClasses; ++c)
        if (splitClass(c / kDepths / 2, (c / kDepths) & 1)) b->stats.split_fits += classCount[c];

    // ---- workgroup table
    uint32_t nBlocks = 0;
    std::vector<FitShape> classShape(kClasses);
    std::vector<uint32_t> classBlockFirst(kClasses + 1, 0);
    for (int c = 0; c < kClasses; ++c) {
        classBlockFirst[c] = nBlocks;
        if (!classCount[c]) continue;
        const int deg = c / kDepths / 2;
        const bool incr = (c / kDepths) & 1;
        const int nrows = (incr || splitClass(deg, incr)) ? (int)(T.coeffCount[deg] - T.coeffCount[deg - 1]) : (int)T.coeffCount[deg];
        classShape[c] = fitShape(deg, nrows, classCount[c], b->weighted, meshFused);
        if (fastDeg(deg)) classShape[c] = FitShape{kMfmaCells, 1, 1, 0};  // one workgroup = one 16-cell tile of the matrix-core fit
        nBlocks += (classCount[c] + classShape[c].cells - 1) / classShape[c].cells;
    }
    classBlockFirst[kClasses] = nBlocks;
    hipError_t he = ws.tasks.ensure(std::max<uint32_t>(1, nTasks));
    if (he == hipSuccess) he = ws.blocks.ensure(std::max<uint32_t>(1, nBlocks));
    // errs: [jobs][9] errors, followed (weighted builds) by [jobs][9] |mean FApprox| values
    const uint64_t nSlots = std::max<uint64_t>(1, sl.count * HPSDF_JOB_HEADER_DOUBLES);
    if (he == hipSuccess) he = ws.errs.ensure(2 * nSlots);
    if (he != hipSuccess) return hipFail(he, "workspace allocation");
    {
        uint32_t bi = 0;
        for (int c = 0; c < kClasses; ++c) {
            if (!classCount[c]) continue;
            const int deg = c / kDepths / 2, g = classShape[c].cells;
            const bool incr = (c / kDepths) & 1;
            for (uint32_t i = 0; i < classCount[c]; i += g) {
                FitBlock& fb = ws.blocks.host[bi++];
                std::memset(&fb, 0, sizeof fb);
                fb.firstTask = classFirst[c] + i;
                fb.nTasks = (uint16_t)std::min<uint32_t>(g, classCount[c] - i);
                fb.degree = (uint8_t)deg;
                fb.planesPerChunk = (uint8_t)classShape[c].planes;
                fb.rowStart = (uint16_t)((incr || splitClass(deg, incr)) ? T.coeffCount[deg - 1] : 0);
                fb.rowEnd = (uint16_t)T.coeffCount[deg];
                fb.depth = (uint8_t)(c % kDepths);
                fb.weighted = b->weighted ? 1 : 0;
                fb.split = splitClass(deg, incr) ? 1 : 0;
            }
        }
    }

    // ---- pass 2: fill the tasks, grouped by shape; arena offsets and sample offsets in job order
    std::vector<uint32_t> cursor(classFirst.begin(), classFirst.end() - 1);
    uint64_t arenaNeed = 0, sampleNeed = 0;
    auto addTask = [&](int deg, bool incr, const float* bmin, const float* bmax, int depth, uint32_t errSlot,
                       uint64_t copyOff = kNone) {
        const uint64_t rows = (incr && !b->weighted) ? T.coeffCount[deg] - T.coeffCount[deg - 1] : T.coeffCount[deg];
        FitTask& t = ws.tasks.host[cursor[classOf(deg, incr, depth)]++];
        for (int a = 0; a < 3; ++a) {
            t.bmin[a] = bmin[a];
            t.bmax[a] = bmax[a];
        }
        t.outOff = b->arenaUsed + arenaNeed;
        t.copyOff = copyOff;
        t.sampleOff = sampleNeed;
        t.errSlot = errSlot;
        t.depth = (uint8_t)depth;
        t.pad[0] = (uint8_t)deg;  // host-side note for the sampler below
        t.pad[1] = t.pad[2] = 0;
        arenaNeed += rows;
        const uint64_t nq = 4 * (uint64_t)deg + 1;
        sampleNeed += nq * nq * nq;
        return t.outOff;
    };
    for (uint64_t jl = 0; jl < sl.count; ++jl) {
        const hpsdf_build::HeapEnt& e = b->batch[sl.first + jl];
        const hpsdf_node& n = b->nodes[e.idx];
        hpsdf_build::JobOut& jo = b->jobOut[jl];
        const uint32_t slot0 = (uint32_t)(jl * HPSDF_JOB_HEADER_DOUBLES);
        if (std::fabs(e.err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON) {
            jo.pOff = addTask(2, false, n.aabb_min, n.aabb_max, n.depth, slot0);
            jo.pHost = 0;
            b->stats.fits += 1;
            continue;
        }
        const int p = n.degree, d = n.depth;
        if (d < kMaxDepth) {  // the reference fits depth-11 children of depth-10 cells and discards them (:601)
            jo.hOff = b->arenaUsed + arenaNeed;
            jo.hHost = 0;
            for (unsigned i = 0; i < 8; ++i) {
                float cmin[3], cmax[3];
                cornerBox(n.aabb_min, n.aabb_max, i, cmin, cmax);
                addTask(p, false, cmin, cmax, d + 1, slot0 + 1 + i);
            }
            b->stats.fits += 8;
        }
        if (p < kMaxDegree - 1) {  // degree 11 is never raised (:600)
            uint64_t prev = kNone;
            if (b->weighted) {
                const int64_t sg = b->segHead[e.idx];
                if (sg < 0 || b->segs[sg].hostStore || (b->segs[sg].owner != b->rank && !b->segs[sg].local))
                    return fail(HPSDF_ERR_STATE, "weighted incremental fit without the node's previous coefficients in HBM "
                                                 "(on N ranks: exchange the accepted rows after every round, hpsdf_build_rows_*)");
                prev = b->segs[sg].off;
            }
            jo.pOff = addTask(p + 1, true, n.aabb_min, n.aabb_max, d, slot0, prev);
            jo.pHost = 0;
            b->stats.fits += 1;
        }
    }
    b->stats.samples += sampleNeed;

    // ---- arena: grow by reallocation (offsets are stable, the pointer is not kept anywhere)
    if (b->arenaUsed + arenaNeed > ws.arenaCap) {
        uint64_t nc = std::max<uint64_t>(ws.arenaCap * 2, 1ull << 22);
        while (nc < b->arenaUsed + arenaNeed) nc *= 2;
        double* na = nullptr;
        HPSDF_HIP(hipMalloc((void**)&na, nc * sizeof(double)));
        if (ws.arena) {
            if (b->arenaUsed)
                HPSDF_HIP(hipMemcpyAsync(na, ws.arena, b->arenaUsed * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            HPSDF_HIP(hipStreamSynchronize(ctx->stream));
            HPSDF_HIP(hipFree(ws.arena));
        }
        ws.arena = na;
        ws.arenaCap = nc;
    }
    b->arenaUsed += arenaNeed;

    // ---- host-evaluated field: sample every fit's grid with thread_count workers, ship the values
    const double* dSamples = nullptr;
    if (sampled) {
        he = ws.samples.ensure(std::max<uint64_t>(1, sampleNeed));
        if (he != hipSuccess) return hipFail(he, "sample buffer");
        const hpsdf_field* cbf = innermost(field);
        double* vals = ws.samples.host;
        double rb[3], rc3[3];
        for (int a = 0; a < 3; ++a) {
            rb[a] = (double)(b->cfg.root_max[a] - b->cfg.root_min[a]);
            rc3[a] = (double)((b->cfg.root_min[a] + b->cfg.root_max[a]) / 2.0f);
        }
        const unsigned nThreads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(b->cfg.thread_count, 256));
        auto worker = [&](unsigned tIdx) {
            for (uint32_t ti = tIdx; ti < nTasks; ti += nThreads) {
                const FitTask& tk = ws.tasks.host[ti];
                const int nq = 4 * (int)tk.pad[0] + 1, gl = glOffset(nq);
                double sc[3], ce[3];
                for (int a = 0; a < 3; ++a) {
                    sc[a] = (double)(tk.bmax[a] - tk.bmin[a]) * 0.5;
                    ce[a] = (double)((tk.bmin[a] + tk.bmax[a]) / 2.0f);
                }
                uint64_t o = tk.sampleOff;
                for (int i = 0; i < nq; ++i)
                    for (int j = 0; j < nq; ++j)
                        for (int k = 0; k < nq; ++k) {
                            const double u[3] = {T.roots[gl + i] * sc[0] + ce[0], T.roots[gl + j] * sc[1] + ce[1],
                                                 T.roots[gl + k] * sc[2] + ce[2]};
                            const double w[3] = {u[0] * rb[0] + rc3[0], u[1] * rb[1] + rc3[1], u[2] * rb[2] + rc3[2]};
                            vals[o++] = cbf->cb(w, tIdx, cbf->user);
                        }
            }
        };
        if (nThreads == 1) {
            worker(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nThreads; ++t) pool.emplace_back(worker, t);
            for (auto& th : pool) th.join();
        }
        HPSDF_HIP(hipMemcpyAsync(ws.samples.dev, vals, sampleNeed * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        dSamples = ws.samples.dev;
    }

    FieldDev fd;
    int rc;
    if ((rc = makeFieldDev(ctx, field, dSamples, &fd))) return rc;
    RootMap rm;
    for (int a = 0; a < 3; ++a) {
        rm.bounds[a] = (double)(b->cfg.root_max[a] - b->cfg.root_min[a]);          // Octree.cpp:324
        rm.centre[a] = (double)((b->cfg.root_min[a] + b->cfg.root_max[a]) / 2.0f);  // Octree.cpp:322
    }
    // pinned sources: the copies, the memset and the kernel are all asynchronous on the context stream
    if (nTasks) {
        HPSDF_HIP(hipMemcpyAsync(ws.tasks.dev, ws.tasks.host, nTasks * sizeof(FitTask), hipMemcpyHostToDevice, ctx->stream));
        HPSDF_HIP(hipMemcpyAsync(ws.blocks.dev, ws.blocks.host, nBlocks * sizeof(FitBlock), hipMemcpyHostToDevice, ctx->stream));
    }
    if ((meshSampled || (splitOn && !sampled)) && nTasks) {  // (split fits of an analytic field leave their values here for the matrix-core kernel)
        HPSDF_HIP(ensureSampleBuffer(ctx, ws, sampleNeed));
        // the tasks of one degree are contiguous (classes are ordered degree-major)
        for (int deg = 0; deg <= kMaxDegree && meshSampled; ++deg) {
            const uint32_t first = classFirst[classOf(deg, false, 0)];
            const uint32_t last = deg == kMaxDegree ? nTasks : classFirst[classOf(deg + 1, false, 0)];
            if (last > first)
                HPSDF_HIP(launchMeshSample(ctx->stream, ws.tasks.dev + first, last - first, deg, ctx->dTables, fd, rm, ws.meshSamples));
        }
        if (meshSampled) fd.kind = kFieldSamples;  // the fit reads what the sampler wrote (a csg wrapper still applies on top)
        fd.samples = ws.meshSamples;
    }
    HPSDF_HIP(hipMemsetAsync(ws.errs.dev, 0, (b->weighted ? 2 : 1) * nSlots * sizeof(double), ctx->stream));
    // one launch per shape class: the degree is a compile-time constant of the kernel
    // the blocks of one degree are contiguous (classes are ordered degree-major) and carry their own rows and
    // depth, so one launch per (degree, cells-per-thread) run covers them
    bool anyFast = false;
    for (int deg = 0; deg <= kMaxDegree; ++deg) anyFast |= fastDeg(deg);
    if (!anyFast && nBlocks) {  // every degree in one launch (kernels.hip fit_multi_kernel: longest fits first, all degrees share the chip)
        size_t ldsAll = 0;
        for (int c = 0; c < kClasses; ++c)
            if (classCount[c]) ldsAll = std::max(ldsAll, classShape[c].ldsBytes);
        HPSDF_HIP(launchFitMulti(ctx->stream, ws.blocks.dev, nBlocks, ldsAll, ws.tasks.dev, ws.arena, ws.errs.dev, ctx->dTables, fd, rm, nullptr));
        if (b->weighted)  // Octree.cpp:1071-1092: the weight's |mean FApprox|, from the coefficients just written
            HPSDF_HIP(launchFitWeight(ctx->stream, ws.blocks.dev, nBlocks, ldsAll, ws.tasks.dev, ws.arena, ws.errs.dev + nSlots, ctx->dTables));
    }
    for (int c = 0; c < kClasses && anyFast;) {
        if (!classCount[c]) {
            ++c;
            continue;
        }
        const int deg = c / kDepths / 2, cpt = classShape[c].cellsPerThread;
        size_t ldsBytes = 0;
        int e = c;
        while (e < kClasses && e / kDepths / 2 == deg && (!classCount[e] || classShape[e].cellsPerThread == cpt)) {
            if (classCount[e]) ldsBytes = std::max(ldsBytes, classShape[e].ldsBytes);
            ++e;
        }
        if (fastDeg(deg))
            HPSDF_HIP(launchFitMfma(ctx->stream, deg, ws.blocks.dev + classBlockFirst[c], classBlockFirst[e] - classBlockFirst[c], ws.tasks.dev,
                                    ws.arena, ws.errs.dev, ctx->dTables, fd, rm));
        else
            HPSDF_HIP(launchFit(ctx->stream, deg, cpt, ws.blocks.dev + classBlockFirst[c], classBlockFirst[e] - classBlockFirst[c],
                                ldsBytes, ws.tasks.dev, ws.arena, ws.errs.dev, nullptr, ctx->dTables, fd, rm));
        if (b->weighted)  // Octree.cpp:1071-1092: the weight's |mean FApprox|, from the coefficients just written
            HPSDF_HIP(launchFitWeight(ctx->stream, ws.blocks.dev + classBlockFirst[c], classBlockFirst[e] - classBlockFirst[c], ldsBytes,
                                      ws.tasks.dev, ws.arena, ws.errs.dev + nSlots, ctx->dTables));
        c = e;
    }
    if (splitOn)  // the rows below the top degree of the split fits: the from-scratch tasks of a degree are one contiguous run
        for (int deg = std::max(2, ctx->splitMinDegree); deg <= 11; ++deg) {
            const uint32_t first = classFirst[classOf(deg, false, 0)], count = classFirst[classOf(deg, true, 0)] - first;
            if (count)
                HPSDF_HIP(launchFitMfmaLow(ctx->stream, deg, ws.tasks.dev, nullptr, first, count, 0u, ws.arena, ctx->dTables, fd.samples, rm, fd.leftAssoc));
        }
    b->computed = true;
    return HPSDF_OK;
}

int builderInject(hpsdf_build* b, uint64_t job, const double* pCoeffs, const double* hCoeffs) {
    if (!b->roundOpen) return fail(HPSDF_ERR_STATE, "no open round");
    const hpsdf_build::Slice sl = b->slices[b->rank];
    if (job < sl.first || job >= sl.first + sl.count) return fail(HPSDF_ERR_INVALID_ARGUMENT, "job not in this rank's slice");
    const hpsdf_build::HeapEnt& e = b->batch[job];
    const hpsdf_node& n = b->nodes[e.idx];
    const bool coarse = std::fabs(e.err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
    hpsdf_build::JobOut& jo = b->jobOut[job - sl.first];
    if (pCoeffs) {
        const uint64_t rs = (coarse || b->weighted) ? 0 : ncoef(n.degree), re = coarse ? ncoef(2) : ncoef(n.degree + 1);
        jo.pOff = b->hostStore.size();
        jo.pHost = 1;
        b->hostStore.insert(b->hostStore.end(), pCoeffs + rs, pCoeffs + re);
    }
    if (hCoeffs && !coarse) {
        jo.hOff = b->hostStore.size();
        jo.hHost = 1;
        b->hostStore.insert(b->hostStore.end(), hCoeffs, hCoeffs + 8 * ncoef(n.degree));
    }
    return HPSDF_OK;
}

// Octree.cpp:594-601 (decision) and :243-299 (bookkeeping), in node-index order
int builderApply(hpsdf_build* b, const double* headers) {
    if (!b->roundOpen) return fail(HPSDF_ERR_STATE, "no open round");
    const Tables& T = tables();
    int owner = 0;
    b->rowItems.clear();
    const bool exchangeRows = b->weighted && b->world > 1;
    for (uint64_t j = 0; j < b->batch.size(); ++j) {
        while (owner + 1 < b->world && j >= b->slices[owner].first + b->slices[owner].count) ++owner;
        const bool mine = owner == b->rank;
        const hpsdf_build::JobOut* jo = mine ? &b->jobOut[j - b->slices[b->rank].first] : nullptr;
        const uint64_t idx = b->batch[j].idx;
        const double err = b->batch[j].err;
        const int p = b->nodes[idx].degree, d = b->nodes[idx].depth;
        const double* h = headers + j * HPSDF_JOB_HEADER_DOUBLES;
        const double pErr = h[0];
        const bool coarse = std::fabs(err - HPSDF_INITIAL_NODE_ERR) < DBL_EPSILON;
        double pImp, hImp;
        if (coarse) {
            hImp = 0.0;   // :806-810
            pImp = pErr;  // :842
        } else {
            if (d < kMaxDepth) {
                double maxNewErr = 0.0;
                for (int i = 0; i < 8; ++i) maxNewErr = std::max<double>(maxNewErr, h[1 + i]);
                hImp = (1.0 / (7.0 * (double)T.coeffCount[p])) * (err - 8.0 * maxNewErr);  // :825
            } else {
                hImp = 0.0;
            }
            if (p < kMaxDegree - 1)
                pImp = (1.0 / (double)(T.coeffCount[p + 1] - T.coeffCount[p])) * (err - 8.0 * pErr);  // :854
            else
                pImp = 0.0;
        }
        bool refineP = p < (kMaxDegree - 1) && (d == kMaxDepth || pImp > hImp);  // :600
        if (coarse) refineP = true;  // a zero-error coarse fit would otherwise take the H branch with no child fits
        const bool refineH = d < kMaxDepth && !refineP;                          // :601
        b->stats.jobs++;
        if (refineP) {  // :253-260, :286-290
            const int np = coarse ? 2 : p + 1;
            hpsdf_build::Seg s;
            s.off = mine ? jo->pOff : kNone;
            s.rowStart = (coarse || b->weighted) ? 0u : (uint32_t)T.coeffCount[p];
            s.rowEnd = (uint32_t)T.coeffCount[np];
            if (b->weighted) b->segHead[idx] = b->segTail[idx] = -1;  // the new array holds every row
            s.owner = owner;
            s.hostStore = mine ? jo->pHost : 0;
            s.next = -1;
            if (mine && s.off == kNone) return fail(HPSDF_ERR_STATE, "P result of an owned job was never computed");
            appendSeg(b, idx, s);
            if (exchangeRows) b->rowItems.push_back({(int64_t)b->segs.size() - 1, owner, s.rowEnd});
            b->nodes[idx].degree = (uint8_t)np;
            b->total += (pErr - err);
            b->heap.push_back({idx, pErr});
            std::push_heap(b->heap.begin(), b->heap.end(), HeapLess());
            b->stats.p_refines++;
        } else if (refineH) {  // :262-279, :286-290
            if (mine && jo->hOff == kNone) return fail(HPSDF_ERR_STATE, "H result of an owned job was never computed");
            b->segHead[idx] = b->segTail[idx] = -1;  // parent basis dropped
            b->nodes[idx].degree = kInteriorDegree;
            subdivide(b, idx);
            b->total -= err;
            const uint64_t c0 = b->nodes[idx].child_idx;
            for (unsigned i = 0; i < 8; ++i) {
                hpsdf_build::Seg s;
                s.off = mine ? jo->hOff + (uint64_t)i * T.coeffCount[p] : kNone;
                s.rowStart = 0;
                s.rowEnd = (uint32_t)T.coeffCount[p];
                s.owner = owner;
                s.hostStore = mine ? jo->hHost : 0;
                s.next = -1;
                appendSeg(b, c0 + i, s);
                if (exchangeRows) b->rowItems.push_back({(int64_t)b->segs.size() - 1, owner, s.rowEnd});
                b->nodes[c0 + i].degree = (uint8_t)p;
                b->total += h[1 + i];
                b->heap.push_back({c0 + i, h[1 + i]});
                std::push_heap(b->heap.begin(), b->heap.end(), HeapLess());
            }
            b->stats.h_refines++;
        } else {
            b->stats.dropped++;  // :643-655: keeps its basis, never queued again
        }
        if (b->cfg.enable_logging)  // :292-296
            std::printf("\n%.11f\t%zu", b->total, b->nodes.size());
    }
    b->stats.rounds++;
    b->stats.total_error = b->total;
    b->roundOpen = false;
    return HPSDF_OK;
}

// ---- weighted builds on N ranks: after every round the ranks hand each other the arrays that round accepted
// (hpsdf.h: hpsdf_build_rows_*).  counts[r] = doubles rank r contributes, items in job order.
int builderRowsCounts(const hpsdf_build* b, uint64_t* counts) {
    if (b->roundOpen) return fail(HPSDF_ERR_STATE, "round still open: apply it first");
    for (int r = 0; r < b->world; ++r) counts[r] = 0;
    for (const auto& it : b->rowItems) counts[it.owner] += it.count;
    return HPSDF_OK;
}

// this rank's accepted arrays of the last round, contiguous, in item order
int builderRowsPackHost(hpsdf_build* b, hpsdf_ctx* ctx, double* out) {
    if (b->roundOpen) return fail(HPSDF_ERR_STATE, "round still open: apply it first");
    uint64_t pos = 0;
    bool anyDevice = false;
    for (const auto& it : b->rowItems) {
        if (it.owner != b->rank) continue;
        const hpsdf_build::Seg& sg = b->segs[it.seg];
        if (sg.hostStore) {
            std::memcpy(out + pos, b->hostStore.data() + sg.off, it.count * sizeof(double));
        } else {
            if (!ctx || !b->ws || !b->ws->arena) return fail(HPSDF_ERR_NO_DEVICE, "rows in HBM need the device context that computed them");
            if (!anyDevice) HPSDF_HIP(hipSetDevice(ctx->device));
            anyDevice = true;
            HPSDF_HIP(hipMemcpyAsync(out + pos, b->ws->arena + sg.off, it.count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        }
        pos += it.count;
    }
    if (anyDevice) HPSDF_HIP(hipStreamSynchronize(ctx->stream));
    return HPSDF_OK;
}

// parts[r] = what rank r packed (parts[rank] is not read).  With a device context the rows go to this rank's arena
// (one upload per rank), without one to the host store (CPU tests); either way the segments now have a local copy.
int builderRowsUnpackHost(hpsdf_build* b, hpsdf_ctx* ctx, const double* const* parts) {
    if (b->roundOpen) return fail(HPSDF_ERR_STATE, "round still open: apply it first");
    std::vector<uint64_t> counts(b->world, 0), base(b->world, 0), cur(b->world, 0);
    for (const auto& it : b->rowItems) counts[it.owner] += it.count;
    uint64_t need = 0;
    for (int r = 0; r < b->world; ++r)
        if (r != b->rank) {
            if (counts[r] && (!parts || !parts[r])) return fail(HPSDF_ERR_INVALID_ARGUMENT, "missing rows of a rank");
            need += counts[r];
        }
    if (need == 0) {
        b->rowItems.clear();
        return HPSDF_OK;
    }
    if (ctx) {
        HPSDF_HIP(hipSetDevice(ctx->device));
        acquireWorkspace(b, ctx);
        Workspace& ws = *b->ws;
        if (b->arenaUsed + need > ws.arenaCap) {  // grow by reallocation, as builderCompute does
            uint64_t nc = std::max<uint64_t>(ws.arenaCap * 2, 1ull << 22);
            while (nc < b->arenaUsed + need) nc *= 2;
            double* na = nullptr;
            HPSDF_HIP(hipMalloc((void**)&na, nc * sizeof(double)));
            if (ws.arena) {
                if (b->arenaUsed)
                    HPSDF_HIP(hipMemcpyAsync(na, ws.arena, b->arenaUsed * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                HPSDF_HIP(hipStreamSynchronize(ctx->stream));
                HPSDF_HIP(hipFree(ws.arena));
            }
            ws.arena = na;
            ws.arenaCap = nc;
        }
        uint64_t at = b->arenaUsed;
        for (int r = 0; r < b->world; ++r) {
            if (r == b->rank || !counts[r]) continue;
            base[r] = at;
            HPSDF_HIP(hipMemcpyAsync(ws.arena + at, parts[r], counts[r] * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            at += counts[r];
        }
        HPSDF_HIP(hipStreamSynchronize(ctx->stream));  // the caller's buffers may go away
        b->arenaUsed = at;
    } else {
        for (int r = 0; r < b->world; ++r) {
            if (r == b->rank || !counts[r]) continue;
            base[r] = b->hostStore.size();
            b->hostStore.insert(b->hostStore.end(), parts[r], parts[r] + counts[r]);
        }
    }
    for (const auto& it : b->rowItems) {
        if (it.owner != b->rank) {
            hpsdf_build::Seg& sg = b->segs[it.seg];
            sg.off = base[it.owner] + cur[it.owner];
            sg.hostStore = ctx ? 0 : 1;
            sg.local = 1;
        }
        cur[it.owner] += it.count;
    }
    b->rowItems.clear();
    return HPSDF_OK;
}

// the rows a node holds right now (its whole chain), if this rank has them: out[0 .. coeffCount[degree])
int builderNodeRowsHost(hpsdf_build* b, hpsdf_ctx* ctx, uint64_t node, double* out, uint64_t* n) {
    if (node >= b->nodes.size()) return fail(HPSDF_ERR_INVALID_ARGUMENT, "no such node");
    uint64_t rows = 0;
    bool anyDevice = false;
    for (int64_t s = b->segHead[node]; s >= 0; s = b->segs[s].next) {
        const hpsdf_build::Seg& sg = b->segs[s];
        if (sg.rowStart != rows) return fail(HPSDF_ERR_STATE, "coefficient segments are not contiguous");
        if (sg.owner != b->rank && !sg.local) return fail(HPSDF_ERR_STATE, "the node's rows live on another rank");
        if (out) {
            if (sg.hostStore) {
                std::memcpy(out + sg.rowStart, b->hostStore.data() + sg.off, (sg.rowEnd - sg.rowStart) * sizeof(double));
            } else {
                if (!ctx || !b->ws || !b->ws->arena) return fail(HPSDF_ERR_NO_DEVICE, "rows in HBM need the device context that computed them");
                if (!anyDevice) HPSDF_HIP(hipSetDevice(ctx->device));
                anyDevice = true;
                HPSDF_HIP(hipMemcpyAsync(out + sg.rowStart, b->ws->arena + sg.off, (sg.rowEnd - sg.rowStart) * sizeof(double), hipMemcpyDeviceToHost,
                                         ctx->stream));
            }
        }
        rows = sg.rowEnd;
    }
    if (anyDevice) HPSDF_HIP(hipStreamSynchronize(ctx->stream));
    if (n) *n = rows;
    return HPSDF_OK;
}

// Octree::ReallocCoeffs, Octree.cpp:474-555: children 0..7 depth first from the root, leaves
// packed in visit order
int builderLayout(hpsdf_build* b) {
    if (b->roundOpen) return fail(HPSDF_ERR_STATE, "round still open");
    const Tables& T = tables();
    b->layout.clear();
    b->packCounts.assign(b->world, 0);
    uint64_t cursor = 0, leaves = 0;
    struct Frame {
        uint64_t node;
        int child;
    };
    std::vector<Frame> st;
    st.push_back({0, 0});
    while (!st.empty()) {
        Frame& f = st.back();
        if (f.child == 8) {
            st.pop_back();
            continue;
        }
        const uint64_t n = b->nodes[f.node].child_idx + (uint64_t)f.child++;
        if (b->nodes[n].child_idx == kNone) {
            ++leaves;
            b->nodes[n].coeffs_start = cursor;
            uint32_t row = 0;
            for (int64_t s = b->segHead[n]; s >= 0; s = b->segs[s].next) {
                const hpsdf_build::Seg& sg = b->segs[s];
                if (sg.rowStart != row) return fail(HPSDF_ERR_STATE, "coefficient segments are not contiguous");
                b->layout.push_back({sg.off, cursor + sg.rowStart, sg.rowEnd - sg.rowStart, sg.owner, sg.hostStore});
                b->packCounts[sg.owner] += sg.rowEnd - sg.rowStart;
                row = sg.rowEnd;
            }
            cursor += T.coeffCount[b->nodes[n].degree];
        } else {
            b->nodes[n].coeffs_start = 0;  // the reference leaves a stale pointer here
            st.push_back({n, 0});
        }
    }
    b->nCoeffsTotal = cursor;
    b->stats.n_nodes = b->nodes.size();
    b->stats.n_leaves = leaves;
    b->stats.n_coeffs = cursor;
    b->laidOut = true;
    return HPSDF_OK;
}

int builderPackDevice(hpsdf_build* b, hpsdf_ctx* ctx, double** dPack, uint64_t* n) {
    if (!b->laidOut) return fail(HPSDF_ERR_STATE, "call hpsdf_build_layout first");
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "pack_device needs a device context");
    HPSDF_HIP(hipSetDevice(ctx->device));
    acquireWorkspace(b, ctx);
    Workspace& ws = *b->ws;
    uint64_t pos = 0, nItems = 0;
    for (const auto& l : b->layout)
        if (l.owner == b->rank) {
            nItems += l.hostStore ? 0 : 1;
            pos += l.count;
        }
    hipError_t he = ws.pack.ensure(std::max<uint64_t>(1, pos));
    if (he == hipSuccess) he = ws.items.ensure(std::max<uint64_t>(1, nItems));
    if (he != hipSuccess) return hipFail(he, "pack buffers");
    uint64_t at = 0, it = 0;
    for (const auto& l : b->layout) {
        if (l.owner != b->rank) continue;
        if (!l.hostStore) ws.items.host[it++] = PackItem{l.src, at, l.count, 0};
        at += l.count;
    }
    if (nItems) {
        HPSDF_HIP(hipMemcpyAsync(ws.items.dev, ws.items.host, nItems * sizeof(PackItem), hipMemcpyHostToDevice, ctx->stream));
        HPSDF_HIP(launchPack(ctx->stream, ws.items.dev, (uint32_t)nItems, ws.arena, ws.pack.dev));
    }
    if (dPack) *dPack = ws.pack.dev;
    if (n) *n = pos;
    return HPSDF_OK;
}

int builderPackHost(hpsdf_build* b, hpsdf_ctx* ctx, double* out) {
    if (!b->laidOut) return fail(HPSDF_ERR_STATE, "call hpsdf_build_layout first");
    bool anyDevice = false;
    for (const auto& l : b->layout) anyDevice |= (l.owner == b->rank && !l.hostStore);
    if (anyDevice) {
        double* dp = nullptr;
        uint64_t n = 0;
        int rc = builderPackDevice(b, ctx, &dp, &n);
        if (rc) return rc;
        HPSDF_HIP(hipMemcpyAsync(b->ws->pack.host, dp, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HPSDF_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(out, b->ws->pack.host, n * sizeof(double));
    }
    uint64_t pos = 0;
    for (const auto& l : b->layout) {
        if (l.owner != b->rank) continue;
        if (l.hostStore) std::memcpy(out + pos, b->hostStore.data() + l.src, l.count * sizeof(double));
        pos += l.count;
    }
    return HPSDF_OK;
}

// Octree::ToMemoryBlock, Octree.cpp:424-456:
//   [u64 nCoeffs][f64 x nCoeffs][u64 nNodes][Node x nNodes][Config], malloc-owned
int builderAssemble(hpsdf_build* b, const double* const* packs, void** block, size_t* size) {
    if (!b->laidOut) return fail(HPSDF_ERR_STATE, "call hpsdf_build_layout first");
    const size_t bytes = 8 + 8 * (size_t)b->nCoeffsTotal + 8 + sizeof(hpsdf_node) * b->nodes.size() + sizeof(hpsdf_config);
    uint8_t* p = (uint8_t*)std::malloc(bytes);
    if (!p) return fail(HPSDF_ERR_OUT_OF_MEMORY, "malloc of the memory block failed");
    const uint64_t nc = b->nCoeffsTotal, nn = b->nodes.size();
    std::memcpy(p, &nc, 8);
    double* store = (double*)(p + 8);
    std::memset(store, 0, 8 * (size_t)nc);
    std::vector<uint64_t> cur(b->world, 0);
    for (const auto& l : b->layout) {
        if (!packs || !packs[l.owner]) {
            std::free(p);
            return fail(HPSDF_ERR_INVALID_ARGUMENT, "missing pack buffer of a rank");
        }
        std::memcpy(store + l.dst, packs[l.owner] + cur[l.owner], l.count * sizeof(double));
        cur[l.owner] += l.count;
    }
    uint8_t* q = p + 8 + 8 * (size_t)nc;
    std::memcpy(q, &nn, 8);
    std::memcpy(q + 8, b->nodes.data(), sizeof(hpsdf_node) * b->nodes.size());
    std::memcpy(q + 8 + sizeof(hpsdf_node) * b->nodes.size(), &b->cfg, sizeof(hpsdf_config));
    *block = p;
    *size = bytes;
    return HPSDF_OK;
}

}  // namespace hpsdf
