// C ABI of include/hpsdf.h: contexts, fields, tree upload + batched Query, the stepwise build.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <limits>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "builder.hpp"
#include "continuity.hpp"
#include "frontier.hpp"
#include "block_check.hpp"
#include "launch.hpp"
#include "runtime.hpp"

using namespace hpsdf;

namespace hpsdf {

int loadObj(const char* path, std::vector<float>& verts, std::vector<uint64_t>& tris, std::string& err);  // obj.cpp

static thread_local std::string g_lastError;
static thread_local hpsdf_continuity_stats g_lastContinuity = {};

void setError(const std::string& msg) { g_lastError = msg; }
int fail(int code, const std::string& msg) {
    g_lastError = msg;
    return code;
}
int hipFail(hipError_t e, const char* what) {
    g_lastError = std::string(what) + ": " + hipGetErrorString(e);
    // the runtime keeps a failed call's code as its "last error" until somebody asks for it: taken here, so that the launch checks of the
    // NEXT call (hipGetLastError() behind every kernel launch) do not report this call's failure again (seen in tools/fuzz_split.py: the
    // build after one that ran out of memory was refused with "launch of fr_init_kernel: out of memory")
    (void)hipGetLastError();
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) ? HPSDF_ERR_NO_DEVICE
                                                                                                 : HPSDF_ERR_HIP;
}

const hpsdf_field* innermost(const hpsdf_field* f) {
    while (f && f->kind == kHostTreeCsg) f = f->inner;
    return f;
}

static int envLeftAssoc() {
    const char* e = std::getenv("HPSDF_REDUCTION_ORDER");  // "left" / "1": (a . b) . c from the start
    return e && (e[0] == 'l' || e[0] == 'L' || e[0] == '1');
}
static std::atomic<int> gLeftAssoc{envLeftAssoc()};
int reductionLeftAssoc(const hpsdf_ctx* ctx) { return ctx && ctx->reductionOrder >= 0 ? ctx->reductionOrder : gLeftAssoc.load(std::memory_order_relaxed); }
void setReductionLeftAssoc(int left) { gLeftAssoc.store(left != 0, std::memory_order_relaxed); }

// hpsdf_set_mesh_face_rule(): 0 = a face-case point that has left its triangle is replaced by the boundary's closest point (default: every
// evaluation path then gives one answer), 1 = the reference's point whatever its weights (Source/Meshing/Utility.cpp:5-97)
static int envMeshFaceRule() {
    const char* e = std::getenv("HPSDF_MESH_FACE_RULE");  // "reference" / "1"
    return e && (e[0] == 'r' || e[0] == 'R' || e[0] == '1');
}
static std::atomic<int> gMeshFaceReference{envMeshFaceRule()};
int meshFaceRuleReference(const hpsdf_ctx* ctx) { return ctx && ctx->meshFaceRule >= 0 ? ctx->meshFaceRule : gMeshFaceReference.load(std::memory_order_relaxed); }
void setMeshFaceRuleReference(int on) { gMeshFaceReference.store(on != 0, std::memory_order_relaxed); }
float meshFaceTolOfSlack(const hpsdf_ctx* ctx) { return meshFaceRuleReference(ctx) ? std::numeric_limits<float>::infinity() : 0.25f; }

uint64_t buildByteLimit(const hpsdf_ctx* ctx, uint64_t growBytes, uint64_t held, uint64_t* measured) {
    constexpr uint64_t kNone = ~0ull, kMeasureFrom = 256ull << 20;
    if (ctx->limitBytes) return ctx->limitBytes;
    if (growBytes <= kMeasureFrom) return kNone;  // (nothing is measured for builds that stay small: every BASELINE config)
    if (*measured == 0) {  // the default: 1/64 of what the device could give this build, at least 1 GiB
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) {
            (void)hipGetLastError();
            freeB = 0;
        }
        *measured = std::max<uint64_t>(1ull << 30, ((uint64_t)freeB + held) / 64);
    }
    return *measured;
}
// bytes: everything the round about to open needs on the device; growBytes: the part of it that grows with the tree (nodes, coefficient
// arena).  A limit the caller set bounds `bytes`.  The DEFAULT exists to end a build that will not end by itself (the reference's default
// Config()) in seconds instead of minutes, and bounds `growBytes` only: a round's sample buffer -- a mesh field's, K x up to 150 000
// samples a job, at most 2^31 samples = 16 GiB whatever the tree's size -- is what an ordinary mesh build at K = 4096 needs (2.9 GiB at
// degree 4) and says nothing about whether the build runs away.  (1/64: 4.5 GiB on an idle MI355X, a tree of some 10 M nodes; at 1/256 a
// mesh build at 1e-8 that was two thirds of the way to its threshold with 3.7 M nodes was refused.)
int checkBuildLimits(const hpsdf_ctx* ctx, uint64_t nodes, uint64_t bytes, uint64_t growBytes, uint64_t held, uint64_t* measured, uint64_t rounds, double total, double target) {
    constexpr uint64_t kNone = ~0ull;
    const uint64_t maxNodes = ctx->limitNodes ? ctx->limitNodes : kNone;
    const uint64_t maxBytes = buildByteLimit(ctx, growBytes, held, measured);
    const uint64_t counted = ctx->limitBytes ? bytes : growBytes;
    const char* how = ctx->limitBytes ? "hpsdf_ctx_set_build_limits" : "the default, on nodes and coefficients: 1/64 of the device memory that was free, at least 1 GiB";
    if (nodes <= maxNodes && counted <= maxBytes) return HPSDF_OK;
    char msg[704];
    std::snprintf(msg, sizeof msg,
                  "build limit: after %llu rounds the tree has %llu nodes (limit %s%llu) and the next round needs %.3f GiB of device memory (limit %.3f GiB, %s); "
                  "total error %.3e against the threshold %.3e -- the threshold may be below what the error estimate reaches on this field. "
                  "hpsdf_ctx_set_build_limits(ctx, max_nodes, max_bytes) raises the limits (UINT64_MAX: none)",
                  (unsigned long long)rounds, (unsigned long long)nodes, maxNodes == kNone ? "none, " : "", (unsigned long long)(maxNodes == kNone ? 0 : maxNodes),
                  (double)counted / (double)(1ull << 30), maxBytes == kNone ? 0.0 : (double)maxBytes / (double)(1ull << 30), maxBytes == kNone ? "none" : how, total, target);
    return fail(HPSDF_ERR_BUILD_LIMIT, msg);
}

int makeFieldDev(const hpsdf_ctx* ctx, const hpsdf_field* f, const double* dSamples, FieldDev* out) {
    std::memset(out, 0, sizeof(*out));
    out->csgOp = -1;
    if (f->kind == kHostTreeCsg) {
        if (!f->oldTree || !f->inner) return fail(HPSDF_ERR_INVALID_ARGUMENT, "csg field without tree or inner field");
        if (f->inner->kind == kHostTreeCsg) return fail(HPSDF_ERR_UNSUPPORTED, "nested csg fields are not supported");
        out->csgOp = f->csgOp;
        out->oldTree = f->oldTree->dev;
        f = f->inner;
    }
    switch (f->kind) {
        case kHostAnalytic:
            out->kind = kFieldAnalytic;
            out->nPrims = (int32_t)f->prims.size();
            for (size_t i = 0; i < f->prims.size(); ++i) out->prims[i] = f->prims[i];
            break;
        case kHostCallback:
            out->kind = kFieldSamples;
            out->samples = dSamples;
            break;
        case kHostMesh:
            out->kind = kFieldMesh;
            out->mesh.verts = f->dVerts;
            out->mesh.tris = f->dTris;
            out->mesh.halfEdges = f->dHalfEdges;
            out->mesh.triPos = reinterpret_cast<const float4*>(f->dTriPos);
            out->mesh.triPre = reinterpret_cast<const float4*>(f->dTriPre);
            out->mesh.bvh = f->dBvh;
            out->mesh.slabs = f->dSlabs;
            out->mesh.leafLog2 = f->leafLog2;
            {
                const char* pc = std::getenv("HPSDF_MESH_POOL_CAP");  // tests: a small pool sends lanes through the overflow path
                out->mesh.poolCap = pc ? (uint32_t)std::strtoul(pc, nullptr, 10) : 0xFFFFFFFFu;
            }
            out->mesh.nTris = f->nTris;
            out->mesh.nNodes = f->nBvhNodes;
            out->mesh.stats = f->dStats;
            out->mesh.faceTolOfSlack = meshFaceTolOfSlack(ctx);
            break;
        default:
            return fail(HPSDF_ERR_INVALID_ARGUMENT, "unknown field kind");
    }
    out->leftAssoc = reductionLeftAssoc(ctx);
    return HPSDF_OK;
}

}  // namespace hpsdf

// ---- host-array entry points: cached device + pinned scratch ---------------------------------------------
namespace {
constexpr size_t kPinnedPathBytes = 1u << 20;  // below this, user memory is staged through the pinned buffer
constexpr size_t kZeroCopyBytes = 4096;        // below this the kernel works on the pinned buffer itself: a scalar
                                               // Query(pt) then costs a launch and a wait, not two copies as well
inline size_t alignUp(size_t b) { return (b + 255) & ~(size_t)255; }

int ensureHostScratch(hpsdf_ctx* ctx, size_t devBytes, size_t pinBytes) {
    if (ctx->hostDevCap < devBytes) {
        if (ctx->hostDev) HPSDF_HIP(hipFree(ctx->hostDev));
        ctx->hostDev = nullptr;
        ctx->hostDevCap = 0;
        size_t cap = 1u << 16;
        while (cap < devBytes) cap *= 2;
        HPSDF_HIP(hipMalloc((void**)&ctx->hostDev, cap));
        ctx->hostDevCap = cap;
    }
    if (ctx->hostPinCap < pinBytes) {
        if (ctx->hostPin) HPSDF_HIP(hipHostFree(ctx->hostPin));
        ctx->hostPin = nullptr;
        ctx->hostPinCap = 0;
        size_t cap = 1u << 16;
        while (cap < pinBytes) cap *= 2;
        HPSDF_HIP(hipHostMalloc((void**)&ctx->hostPin, cap, hipHostMallocDefault));
        ctx->hostPinCap = cap;
        void* dp = nullptr;
        ctx->hostPinDev = hipHostGetDevicePointer(&dp, ctx->hostPin, 0) == hipSuccess ? (char*)dp : nullptr;
    }
    return HPSDF_OK;
}

// One host call: `ins` are copied to the device, `run` launches on the context stream, `outs` come back.
// inout arrays are both (rows the kernel leaves untouched keep the caller's values).
struct HostArray {
    const void* src;  // host source (nullptr: output only)
    void* dst;        // host destination (nullptr: input only)
    size_t bytes;
    char* dev = nullptr;
};
template <typename Run>
int hostCall(hpsdf_ctx* ctx, HostArray* arrays, int nArrays, Run&& run) {
    std::lock_guard<std::mutex> guard(ctx->hostLock);
    HPSDF_HIP(hipSetDevice(ctx->device));
    size_t total = 0;
    for (int a = 0; a < nArrays; ++a) total += alignUp(arrays[a].bytes);
    const bool staged = total <= kPinnedPathBytes;
    int rc = ensureHostScratch(ctx, total, staged ? total : 0);
    if (rc) return rc;
    static const bool zeroCopyOff = std::getenv("HPSDF_NO_ZEROCOPY") != nullptr;  // measurement knob
    const bool zeroCopy = total <= kZeroCopyBytes && ctx->hostPinDev != nullptr && !zeroCopyOff;
    size_t off = 0;
    for (int a = 0; a < nArrays; ++a) {
        arrays[a].dev = zeroCopy ? ctx->hostPinDev + off : ctx->hostDev + off;
        if (arrays[a].src) {
            const void* from = arrays[a].src;
            if (staged) {
                std::memcpy(ctx->hostPin + off, arrays[a].src, arrays[a].bytes);
                from = ctx->hostPin + off;
            }
            if (!zeroCopy) HPSDF_HIP(hipMemcpyAsync(arrays[a].dev, from, arrays[a].bytes, hipMemcpyHostToDevice, ctx->stream));
        }
        off += alignUp(arrays[a].bytes);
    }
    rc = run();
    if (rc) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    off = 0;
    for (int a = 0; a < nArrays; ++a) {
        if (arrays[a].dst && !zeroCopy)
            HPSDF_HIP(hipMemcpyAsync(staged ? (void*)(ctx->hostPin + off) : arrays[a].dst, arrays[a].dev, arrays[a].bytes,
                                     hipMemcpyDeviceToHost, ctx->stream));
        off += alignUp(arrays[a].bytes);
    }
    HPSDF_HIP(hipStreamSynchronize(ctx->stream));
    if (staged) {
        off = 0;
        for (int a = 0; a < nArrays; ++a) {
            if (arrays[a].dst) std::memcpy(arrays[a].dst, ctx->hostPin + off, arrays[a].bytes);
            off += alignUp(arrays[a].bytes);
        }
    }
    return HPSDF_OK;
}
}  // namespace

#define HPSDF_TRY                                                            \
    try {
#define HPSDF_CATCH                                                          \
    }                                                                        \
    catch (const std::bad_alloc&) {                                          \
        return fail(HPSDF_ERR_OUT_OF_MEMORY, "host allocation failed");      \
    }                                                                        \
    catch (const std::exception& ex) {                                       \
        return fail(HPSDF_ERR_STATE, std::string("exception: ") + ex.what()); \
    }

extern "C" {

const char* hpsdf_last_error(void) { return g_lastError.c_str(); }
const char* hpsdf_version(void) { return "hpsdf-gfx950 0.4"; }  // (minor = HPSDF_ABI_VERSION)
int hpsdf_abi_version(void) { return HPSDF_ABI_VERSION; }

int hpsdf_config_default(hpsdf_config* c) {
    if (!c) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null config");
    std::memset(c, 0, sizeof(*c));
    c->target_error_threshold = std::pow(10, -10);
    c->weighting_type = 0;
    c->continuity_enforce = 1;
    c->continuity_strength = 8.0;
    const unsigned hc = std::thread::hardware_concurrency();
    c->thread_count = hc ? hc : 1;
    for (int a = 0; a < 3; ++a) {
        c->root_min[a] = -0.5f;
        c->root_max[a] = 0.5f;
    }
    return HPSDF_OK;
}

int hpsdf_tables_get(double* roots, double* weights, double* nl, double* rec, uint64_t* count, uint64_t* bidx,
                     uint64_t* sumToN) {
    HPSDF_TRY
    const Tables& T = tables();
    if (roots) std::memcpy(roots, T.roots, sizeof T.roots);
    if (weights) std::memcpy(weights, T.weights, sizeof T.weights);
    if (nl) std::memcpy(nl, T.normalisedLengths, sizeof T.normalisedLengths);
    if (rec) std::memcpy(rec, T.recurrence, sizeof T.recurrence);
    if (count) std::memcpy(count, T.coeffCount, sizeof T.coeffCount);
    if (bidx) std::memcpy(bidx, T.basisIndex, sizeof T.basisIndex);
    if (sumToN) std::memcpy(sumToN, T.sumToN, sizeof T.sumToN);
    return HPSDF_OK;
    HPSDF_CATCH
}

// ---------------------------------------------------------------------------- context
int hpsdf_ctx_create(int device, void* stream, hpsdf_ctx** out) {
    HPSDF_TRY
    if (!out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(HPSDF_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0"));
    if (device < 0 || device >= n) return fail(HPSDF_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    HPSDF_HIP(hipSetDevice(device));
    hpsdf_ctx* c = new hpsdf_ctx();
    c->device = device;
    c->splitMinDegree = fitSplitDefaultMinDegree();
    if (const char* fm = std::getenv("HPSDF_FIT_MODE")) {  // the mode a context starts with (hpsdf_ctx_set_fit_mode changes it)
        if (!std::strcmp(fm, "exact")) c->fitMode = HPSDF_FIT_EXACT;
        else if (!std::strcmp(fm, "fast")) c->fitMode = HPSDF_FIT_FAST;
        else if (!std::strcmp(fm, "split")) c->fitMode = HPSDF_FIT_SPLIT;
    }
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete c;
            return hipFail(se, "hipStreamCreate");
        }
        c->ownsStream = true;
    }
    // constant tables -> HBM
    const Tables& T = tables();
    DeviceTables* h = new DeviceTables();
    std::memset(h, 0, sizeof(*h));
    std::memcpy(h->roots, T.roots, sizeof T.roots);
    std::memcpy(h->weights, T.weights, sizeof T.weights);
    std::memcpy(h->nl, T.normalisedLengths, sizeof T.normalisedLengths);
    std::memcpy(h->rec, T.recurrence, sizeof T.recurrence);
    for (int i = 0; i < kMaxCoeffs; ++i) {
        h->bidx[i][0] = (uint8_t)T.basisIndex[i][0];
        h->bidx[i][1] = (uint8_t)T.basisIndex[i][1];
        h->bidx[i][2] = (uint8_t)T.basisIndex[i][2];
        h->bidx[i][3] = (uint8_t)(T.basisIndex[i][0] + T.basisIndex[i][1] + T.basisIndex[i][2]);
    }
    for (int i = 0; i <= kMaxDegree; ++i) h->count[i] = (uint32_t)T.coeffCount[i];
    hipError_t me = hipMalloc((void**)&c->dTables, sizeof(DeviceTables));
    if (me == hipSuccess) me = hipMemcpy(c->dTables, h, sizeof(DeviceTables), hipMemcpyHostToDevice);
    delete h;
    if (me != hipSuccess) {
        hpsdf_ctx_destroy(c);
        return hipFail(me, "table upload");
    }
    *out = c;
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_ctx_destroy(hpsdf_ctx* c) {
    if (!c) return HPSDF_OK;
    (void)hipSetDevice(c->device);
    // (nothing of this context's may still be running when its buffers go: a build leaves its tables' reset on the stream for the next
    // one, and pinned host memory the device writes -- the header's mirror -- is freed below)
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    if (c->dTables) (void)hipFree(c->dTables);
    if (c->hostDev) (void)hipFree(c->hostDev);
    if (c->hostPin) (void)hipHostFree(c->hostPin);
    if (c->dDefer) (void)hipFree(c->dDefer);
    if (c->dDeferCount) (void)hipFree(c->dDeferCount);
    c->ws.release();
    if (c->ownsStream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return HPSDF_OK;
}

int hpsdf_ctx_set_stream(hpsdf_ctx* c, void* stream) {
    if (!c) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null ctx");
    if (c->ownsStream && c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
        c->ownsStream = false;
    }
    c->stream = (hipStream_t)stream;
    return HPSDF_OK;
}

int hpsdf_ctx_set_fit_mode(hpsdf_ctx* c, int mode) {
    if (!c) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null context");
    if (mode != HPSDF_FIT_EXACT && mode != HPSDF_FIT_SPLIT && mode != HPSDF_FIT_FAST) return fail(HPSDF_ERR_INVALID_ARGUMENT, "unknown fit mode");
    c->fitMode = mode;
    return HPSDF_OK;
}
int hpsdf_ctx_get_fit_mode(hpsdf_ctx* c, int* mode) {
    if (!c || !mode) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    *mode = c->fitMode;
    return HPSDF_OK;
}
int hpsdf_ctx_set_build_limits(hpsdf_ctx* c, uint64_t max_nodes, uint64_t max_bytes) {
    if (!c) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null context");
    c->limitNodes = max_nodes, c->limitBytes = max_bytes;
    return HPSDF_OK;
}
int hpsdf_ctx_get_build_limits(const hpsdf_ctx* c, uint64_t* max_nodes, uint64_t* max_bytes) {
    if (!c) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null context");
    if (max_nodes) *max_nodes = c->limitNodes;
    if (max_bytes) *max_bytes = c->limitBytes;
    return HPSDF_OK;
}
int hpsdf_ctx_set_split_min_degree(hpsdf_ctx* c, int degree) {
    if (!c) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null context");
    if (degree < 2 || degree > 12) return fail(HPSDF_ERR_INVALID_ARGUMENT, "split_min_degree: 2..12 (12 = never split)");
    c->splitMinDegree = degree;
    return HPSDF_OK;
}
void hpsdf_set_mesh_face_rule(int reference) { setMeshFaceRuleReference(reference); }
int hpsdf_get_mesh_face_rule(void) { return meshFaceRuleReference(nullptr); }
void hpsdf_set_reduction_order(int left_assoc) { setReductionLeftAssoc(left_assoc); }
int hpsdf_get_reduction_order(void) { return reductionLeftAssoc(nullptr); }
// per context (-1: follow the process-wide setting above)
int hpsdf_ctx_set_reduction_order(hpsdf_ctx* c, int left_assoc) {
    if (!c) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null context");
    if (left_assoc < -1 || left_assoc > 1) return fail(HPSDF_ERR_INVALID_ARGUMENT, "reduction order: -1 (the process-wide setting), 0 or 1");
    c->reductionOrder = left_assoc;
    return HPSDF_OK;
}
int hpsdf_ctx_get_reduction_order(const hpsdf_ctx* c, int* left_assoc) {
    if (!c || !left_assoc) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    *left_assoc = reductionLeftAssoc(c);
    return HPSDF_OK;
}
int hpsdf_ctx_set_mesh_face_rule(hpsdf_ctx* c, int reference) {
    if (!c) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null context");
    if (reference < -1 || reference > 1) return fail(HPSDF_ERR_INVALID_ARGUMENT, "mesh face rule: -1 (the process-wide setting), 0 or 1");
    c->meshFaceRule = reference;
    return HPSDF_OK;
}
int hpsdf_ctx_get_mesh_face_rule(const hpsdf_ctx* c, int* reference) {
    if (!c || !reference) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    *reference = meshFaceRuleReference(c);
    return HPSDF_OK;
}
int hpsdf_ctx_set_fast_fit(hpsdf_ctx* c, int on) { return hpsdf_ctx_set_fit_mode(c, on ? HPSDF_FIT_FAST : HPSDF_FIT_EXACT); }

int hpsdf_ctx_synchronize(hpsdf_ctx* c) {
    if (!c) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null ctx");
    HPSDF_HIP(hipSetDevice(c->device));
    HPSDF_HIP(hipStreamSynchronize(c->stream));
    return HPSDF_OK;
}

void* hpsdf_ctx_stream(hpsdf_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ---------------------------------------------------------------------------- fields
int hpsdf_field_create_analytic(const hpsdf_prim* prims, int n, hpsdf_field** out) {
    HPSDF_TRY
    if (!prims || !out || n < 1 || n > HPSDF_MAX_PRIMS)
        return fail(HPSDF_ERR_INVALID_ARGUMENT, "analytic field needs 1..HPSDF_MAX_PRIMS primitives");
    for (int i = 0; i < n; ++i)
        if (prims[i].kind < 0 || prims[i].kind > HPSDF_PRIM_PLANE || prims[i].op < 0 || prims[i].op > HPSDF_OP_SUBTRACT)
            return fail(HPSDF_ERR_INVALID_ARGUMENT, "unknown primitive kind or op");
    hpsdf_field* f = new hpsdf_field();
    f->kind = kHostAnalytic;
    f->prims.assign(prims, prims + n);
    *out = f;
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_field_create_callback(hpsdf_callback cb, void* user, hpsdf_field** out) {
    HPSDF_TRY
    if (!cb || !out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null callback");
    hpsdf_field* f = new hpsdf_field();
    f->kind = kHostCallback;
    f->cb = cb;
    f->user = user;
    *out = f;
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_field_create_mesh(hpsdf_ctx* ctx, const float* verts, uint64_t nVerts, const uint64_t* tris, uint64_t nTris,
                            hpsdf_field** out) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "mesh fields live in HBM: a device context is required");
    if (!verts || !tris || !out || nVerts == 0 || nTris == 0) return fail(HPSDF_ERR_INVALID_ARGUMENT, "empty mesh");
    if (nVerts > 0x7FFFFFFFull || nTris > kMeshMaxTris) return fail(HPSDF_ERR_UNSUPPORTED, "mesh too large: at most 2^27 - 1 triangles");
    HPSDF_HIP(hipSetDevice(ctx->device));
    hpsdf_field* f = new hpsdf_field();
    f->kind = kHostMesh;
    f->device = ctx->device;
    hipError_t e = hipSuccess;
    // half-edge twins and BVH on the device (mesh_build.hip); HPSDF_MESH_HOST_BUILD=1, non-manifold input and single
    // triangles take the host preparation of mesh.cpp (same field values either way: any BVH gives the same distances)
    int fallback = 1;
    {
        const char* hb = std::getenv("HPSDF_MESH_HOST_BUILD");
        if (!(hb && hb[0] == '1')) {
            const int brc = meshBuildDevice(ctx, verts, nVerts, tris, nTris, f, &fallback);
            if (brc != HPSDF_OK) {
                delete f;
                return brc;
            }
        }
    }
    if (fallback == 2) {  // non-manifold: the twins from the host's sequential pairing, everything else is on the device already
        std::vector<uint32_t> he;
        if (!hostHalfEdges(tris, nTris, nVerts, &he)) {
            hpsdf_field_destroy(f);
            return fail(HPSDF_ERR_OPEN_MESH, "mesh is not closed: an edge has no twin (Mesh::CreateHalfEdges)");
        }
        e = hipMemcpy(f->dHalfEdges, he.data(), he.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    } else if (fallback) {
        for (uint64_t i = 0; i < 3 * nTris; ++i)
            if (tris[i] >= nVerts) {
                delete f;
                return fail(HPSDF_ERR_INVALID_ARGUMENT, "triangle " + std::to_string(i / 3) + " refers to vertex " + std::to_string(tris[i]) +
                                                            " of " + std::to_string(nVerts));
            }
        HostMesh hm;
        if (!prepareMesh(verts, nVerts, tris, nTris, &hm)) {
            delete f;
            return fail(HPSDF_ERR_OPEN_MESH, "mesh is not closed: an edge has no twin (Mesh::CreateHalfEdges)");
        }
        f->nVerts = (uint32_t)nVerts;
        f->nTris = (uint32_t)nTris;
        f->nBvhNodes = (uint32_t)hm.bvh.size();
        auto up = [&](void** d, const void* h, size_t bytes) -> hipError_t {
            hipError_t ue = hipMalloc(d, bytes);
            if (ue != hipSuccess) return ue;
            return hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        };
        e = up((void**)&f->dVerts, hm.verts.data(), hm.verts.size() * sizeof(float));
        if (e == hipSuccess) e = up((void**)&f->dTris, hm.tris.data(), hm.tris.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&f->dTriPos, (size_t)nTris * kTriRecordFloats * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&f->dTriPre, (size_t)nTris * kTriPreFloats * sizeof(float));
        if (e == hipSuccess) e = launchMeshTriPos(ctx->stream, f->dVerts, f->dTris, nTris, f->dTriPos, nullptr, f->dTriPre);  // after the blocking uploads
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = up((void**)&f->dHalfEdges, hm.halfEdges.data(), hm.halfEdges.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = up((void**)&f->dBvh, hm.bvh.data(), hm.bvh.size() * sizeof(BvhNode));
    }
#ifdef HPSDF_MESH_STATS_BUILD
    if (e == hipSuccess && std::getenv("HPSDF_MESH_STATS")) {
        e = hipMalloc((void**)&f->dStats, 8 * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(f->dStats, 0, 8 * sizeof(unsigned long long));
    }
#endif
    if (e != hipSuccess) {
        hpsdf_field_destroy(f);
        return hipFail(e, "mesh upload");
    }
    *out = f;
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_obj_load(const char* path, float** verts, uint64_t* nVerts, uint64_t** tris, uint64_t* nTris) {
    HPSDF_TRY
    if (!path || !verts || !nVerts || !tris || !nTris) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    *verts = nullptr, *tris = nullptr, *nVerts = 0, *nTris = 0;
    std::vector<float> v;
    std::vector<uint64_t> t;
    std::string err;
    const int rc = loadObj(path, v, t, err);
    if (rc) return fail(rc, err);
    *verts = (float*)std::malloc(v.size() * sizeof(float));
    *tris = (uint64_t*)std::malloc(t.size() * sizeof(uint64_t));
    if (!*verts || !*tris) {
        std::free(*verts), std::free(*tris);
        *verts = nullptr, *tris = nullptr;
        return fail(HPSDF_ERR_OUT_OF_MEMORY, "malloc failed");
    }
    std::memcpy(*verts, v.data(), v.size() * sizeof(float));
    std::memcpy(*tris, t.data(), t.size() * sizeof(uint64_t));
    *nVerts = v.size() / 3;
    *nTris = t.size() / 3;
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_field_create_tree_csg(const hpsdf_tree* old, int op, const hpsdf_field* inner, hpsdf_field** out) {
    HPSDF_TRY
    if (!old || !inner || !out || op < 0 || op > HPSDF_OP_SUBTRACT)
        return fail(HPSDF_ERR_INVALID_ARGUMENT, "bad csg field arguments");
    if (inner->kind == kHostTreeCsg) return fail(HPSDF_ERR_UNSUPPORTED, "nested csg fields are not supported");
    hpsdf_field* f = new hpsdf_field();
    f->kind = kHostTreeCsg;
    f->oldTree = old;
    f->csgOp = op;
    f->inner = inner;
    *out = f;
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_field_destroy(hpsdf_field* f) {
    if (!f) return HPSDF_OK;
    if (f->kind == kHostMesh && f->device >= 0) {
        (void)hipSetDevice(f->device);
        if (f->dBlock) {
            (void)hipFree(f->dBlock);
            hpsdf::meshPoolTrim(f->device);
        } else {
            if (f->dVerts) (void)hipFree(f->dVerts);
            if (f->dTris) (void)hipFree(f->dTris);
            if (f->dTriPos) (void)hipFree(f->dTriPos);
            if (f->dTriPre) (void)hipFree(f->dTriPre);
            if (f->dHalfEdges) (void)hipFree(f->dHalfEdges);
            if (f->dBvh) (void)hipFree(f->dBvh);
        }
        if (f->dStats) (void)hipFree(f->dStats);
    }
    delete f;
    return HPSDF_OK;
}

int hpsdf_field_mesh_stats(const hpsdf_field* f, uint64_t out[8], int reset) {
    HPSDF_TRY
    if (!f || !out || f->kind != kHostMesh) return fail(HPSDF_ERR_INVALID_ARGUMENT, "not a mesh field");
    if (!f->dStats) return fail(HPSDF_ERR_UNSUPPORTED, "traversal counters need a diagnostic build (-DHPSDF_MESH_STATS_BUILD) and HPSDF_MESH_STATS=1");
    HPSDF_HIP(hipSetDevice(f->device));
    HPSDF_HIP(hipDeviceSynchronize());
    HPSDF_HIP(hipMemcpy(out, f->dStats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HPSDF_HIP(hipMemset(f->dStats, 0, 8 * sizeof(uint64_t)));
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_field_eval_device(hpsdf_ctx* ctx, const hpsdf_field* f, const double* dXyz, size_t n, double* dOut) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!f || (!dXyz && n) || (!dOut && n)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (innermost(f)->kind == kHostCallback) return fail(HPSDF_ERR_UNSUPPORTED, "callback fields are evaluated on the host");
    HPSDF_HIP(hipSetDevice(ctx->device));
    FieldDev fd;
    int rc = makeFieldDev(ctx, f, nullptr, &fd);
    if (rc) return rc;
    // A plain mesh field goes through the sampler's traversal, 64 consecutive points per walk (meshSignedDistanceWaveQ:
    // dense nodes walked by the wave, sparse subtrees pooled, tests compacted) -- the same bits as the per-point traversal
    // and 4 (random points) to 8 (points sorted by cell) times its speed on a 2 M-triangle mesh.
    // (Not under hpsdf_set_mesh_face_rule(1): the shared traversal's bounds -- slabs, in-plane rectangles -- bound the DISTANCE to a
    // triangle, and the reference's face-case point can lie below it; the per-point traversal prunes by boxes alone, as the reference's does.)
    if (fd.kind == kFieldMesh && fd.csgOp < 0 && !meshFaceRuleReference(ctx))
        HPSDF_HIP(launchMeshEvalWave(ctx->stream, fd, dXyz, n, dOut));
    else
        HPSDF_HIP(launchFieldEval(ctx->stream, fd, ctx->dTables, dXyz, n, dOut));
    return HPSDF_OK;
    HPSDF_CATCH
}

static int hostRoundTrip(hpsdf_ctx* ctx, const double* xyz, size_t n, double* out,
                         int (*run)(hpsdf_ctx*, const void*, const double*, size_t, double*), const void* obj) {
    if (n == 0) return HPSDF_OK;
    HostArray arr[2] = {{xyz, nullptr, n * 3 * sizeof(double)}, {nullptr, out, n * sizeof(double)}};
    return hostCall(ctx, arr, 2, [&] { return run(ctx, obj, (const double*)arr[0].dev, n, (double*)arr[1].dev); });
}

// Up to how many points / rays a *_host call is answered on the calling thread.  The defaults are where the two paths cost the same on an
// MI355X box (tools/host_api_latency.py, profiles/r04_host_call_thresholds.txt): a launch round trip is ~15 us (~55 us for the mesh
// traversal), a point on the host 0.045 us (Query), 0.07 us (gradient), <= 9 us (a ray: up to 200 Query steps), ~23 us (mesh distance).
// HPSDF_HOST_QUERY_POINTS / _GRADIENT_POINTS / _RAYS / _MESH_POINTS override them (read once).
static size_t hostLimit(const char* env, size_t dflt) {
    const char* e = std::getenv(env);
    if (!e) return dflt;
    const long v = std::atol(e);
    return v < 0 ? 0 : (size_t)v;
}
static size_t hostQueryLimit() { static const size_t v = hostLimit("HPSDF_HOST_QUERY_POINTS", kHostQueryPoints); return v; }
static size_t hostGradientLimit() { static const size_t v = hostLimit("HPSDF_HOST_GRADIENT_POINTS", kHostGradientPoints); return v; }
static size_t hostRayLimit() { static const size_t v = hostLimit("HPSDF_HOST_RAYS", kHostRays); return v; }
static size_t hostMeshLimit() { static const size_t v = hostLimit("HPSDF_HOST_MESH_POINTS", kHostMeshPoints); return v; }
// HPSDF_SMALL_QUERIES_ON_DEVICE=1 (read once) sends calls of a few points through query_few_kernel as before round 4: a measurement knob
static bool smallQueriesOnHost() {
    static const bool on = [] {
        const char* e = std::getenv("HPSDF_SMALL_QUERIES_ON_DEVICE");
        return !(e && e[0] == '1');
    }();
    return on;
}

// What the host copies of a mesh field's arrays weigh, and up to where they are made at all: beyond HPSDF_HOST_MESH_MIRROR_MB (default
// 512) a call of a few points is a launch like any other -- a mirror of a 2 M-triangle mesh is ~330 MB of pageable memory, fetched
// behind a device-wide synchronisation (the field's arrays may still be being written on some stream), under the field's lock, and kept
// until hpsdf_field_release_host_copies() or the field's destruction.
static size_t meshMirrorBytes(const hpsdf_field* f) {
    return (size_t)f->nVerts * 12 + (size_t)f->nTris * (12 + 12 + 4 * (size_t)(kTriRecordFloats + kTriPreFloats)) + (size_t)f->nBvhNodes * sizeof(BvhNode);
}
static size_t hostMeshMirrorLimit() { static const size_t v = hostLimit("HPSDF_HOST_MESH_MIRROR_MB", 512) << 20; return v; }
// The host copies of a mesh field's arrays (made once, by the first call of a few points)
static int meshHostMirror(const hpsdf_field* f, std::shared_ptr<hpsdf_field::HostMirror>* out) {
    std::lock_guard<std::mutex> guard(f->hostMirrorLock);
    if (!f->hostMirror) {
        HPSDF_HIP(hipSetDevice(f->device));
        auto m = std::make_shared<hpsdf_field::HostMirror>();
        m->verts.resize(3 * (size_t)f->nVerts);
        m->tris.resize(3 * (size_t)f->nTris);
        m->halfEdges.resize(3 * (size_t)f->nTris);
        m->triPos.resize((size_t)kTriRecordFloats * f->nTris);
        m->triPre.resize((size_t)kTriPreFloats * f->nTris);
        m->bvh.resize(std::max<size_t>(1, f->nBvhNodes));
        HPSDF_HIP(hipDeviceSynchronize());  // (the field's arrays may still be being written on some stream)
        HPSDF_HIP(hipMemcpy(m->verts.data(), f->dVerts, m->verts.size() * sizeof(float), hipMemcpyDeviceToHost));
        HPSDF_HIP(hipMemcpy(m->tris.data(), f->dTris, m->tris.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HPSDF_HIP(hipMemcpy(m->halfEdges.data(), f->dHalfEdges, m->halfEdges.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HPSDF_HIP(hipMemcpy(m->triPos.data(), f->dTriPos, m->triPos.size() * sizeof(float), hipMemcpyDeviceToHost));
        HPSDF_HIP(hipMemcpy(m->triPre.data(), f->dTriPre, m->triPre.size() * sizeof(float), hipMemcpyDeviceToHost));
        if (f->nBvhNodes) HPSDF_HIP(hipMemcpy(m->bvh.data(), f->dBvh, (size_t)f->nBvhNodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
        MeshDev& d = m->dev;
        d.verts = m->verts.data(), d.tris = m->tris.data(), d.halfEdges = m->halfEdges.data();
        d.triPos = reinterpret_cast<const float4*>(m->triPos.data());
        d.triPre = reinterpret_cast<const float4*>(m->triPre.data());
        d.bvh = m->bvh.data(), d.slabs = nullptr;  // (the per-point traversal uses boxes and triangle records only)
        d.nTris = f->nTris, d.nNodes = f->nBvhNodes, d.leafLog2 = f->leafLog2, d.poolCap = 0, d.stats = nullptr;
        f->hostMirror = m;
    }
    *out = f->hostMirror;
    return HPSDF_OK;
}

int hpsdf_field_release_host_copies(hpsdf_field* f) {
    if (!f) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null field");
    std::lock_guard<std::mutex> guard(f->hostMirrorLock);
    f->hostMirror.reset();  // (calls in flight hold their own reference)
    return HPSDF_OK;
}

int hpsdf_field_eval_host(hpsdf_ctx* ctx, const hpsdf_field* f, const double* xyz, size_t n, double* out) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!f || (!xyz && n) || (!out && n)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    // Mesh::SignedDistanceAtPt(pt, bvh): a call of a few points on a plain mesh field never reaches the device (~57 us as a launch;
    // a user's SDF lambda that calls it per sample -- the reference's own usage, Mesh.cpp:54-63 -- would cost minutes per Create)
    if (n && n <= hostMeshLimit() && f->kind == kHostMesh && f->nTris >= 1 && f->nBvhNodes >= 1 && smallQueriesOnHost() &&
        meshMirrorBytes(f) <= hostMeshMirrorLimit()) {
        std::shared_ptr<hpsdf_field::HostMirror> m;
        const int rc = meshHostMirror(f, &m);
        if (rc) return rc;
        MeshDev hm = m->dev;
        hm.faceTolOfSlack = meshFaceTolOfSlack(ctx);  // (the rule at the time of the call, as on the device)
        meshEvalHostPoints(hm, xyz, n, out);
        return HPSDF_OK;
    }
    return hostRoundTrip(
        ctx, xyz, n, out,
        [](hpsdf_ctx* c, const void* o, const double* d, size_t m, double* r) {
            return hpsdf_field_eval_device(c, (const hpsdf_field*)o, d, m, r);
        },
        f);
    HPSDF_CATCH
}

static int meshNaiveDevice(hpsdf_ctx* ctx, const hpsdf_field* f, const double* dXyz, size_t n, double* dOut) {
    if (f->kind != kHostMesh) return fail(HPSDF_ERR_INVALID_ARGUMENT, "the linear scan is defined for mesh fields");
    HPSDF_HIP(hipSetDevice(ctx->device));
    FieldDev fd;
    int rc = makeFieldDev(ctx, f, nullptr, &fd);
    if (rc) return rc;
    // tiny calls run on the pinned host buffer as the device sees it (hostCall): the scan's atomics then go to the device-side
    // staging buffer, which such a call leaves unused and which holds the call's arrays (32 bytes a point) at least
    unsigned long long* keys = nullptr;
    if (ctx->hostPinDev && (char*)dOut >= ctx->hostPinDev && (char*)dOut < ctx->hostPinDev + ctx->hostPinCap) {
        if (ctx->hostDevCap < n * sizeof(double)) return fail(HPSDF_ERR_STATE, "the staging buffer is smaller than the call");
        keys = reinterpret_cast<unsigned long long*>(ctx->hostDev);
    }
    HPSDF_HIP(launchMeshNaive(ctx->stream, fd, dXyz, n, dOut, keys));
    return HPSDF_OK;
}

static int meshWaveDevice(hpsdf_ctx* ctx, const hpsdf_field* f, const double* dXyz, size_t n, double* dOut) {
    if (f->kind != kHostMesh) return fail(HPSDF_ERR_INVALID_ARGUMENT, "the shared traversal is defined for mesh fields");
    HPSDF_HIP(hipSetDevice(ctx->device));
    FieldDev fd;
    int rc = makeFieldDev(ctx, f, nullptr, &fd);
    if (rc) return rc;
    if (meshFaceRuleReference(ctx))
        return fail(HPSDF_ERR_UNSUPPORTED, "the shared traversal's bounds assume the default face rule (hpsdf_set_mesh_face_rule(0)): use hpsdf_field_eval_* or the scan");
    HPSDF_HIP(launchMeshEvalWave(ctx->stream, fd, dXyz, n, dOut));
    return HPSDF_OK;
}

int hpsdf_field_eval_wave_host(hpsdf_ctx* ctx, const hpsdf_field* f, const double* xyz, size_t n, double* out) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!f || (!xyz && n) || (!out && n)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return hostRoundTrip(
        ctx, xyz, n, out,
        [](hpsdf_ctx* c, const void* o, const double* d, size_t m, double* r) { return meshWaveDevice(c, (const hpsdf_field*)o, d, m, r); }, f);
    HPSDF_CATCH
}

// the per-point stack traversal (what a mesh field wrapped by a tree-CSG and the fused mesh fit run): diagnostics
static int meshLaneDevice(hpsdf_ctx* ctx, const hpsdf_field* f, const double* dXyz, size_t n, double* dOut) {
    if (f->kind != kHostMesh) return fail(HPSDF_ERR_INVALID_ARGUMENT, "the per-point traversal is defined for mesh fields");
    HPSDF_HIP(hipSetDevice(ctx->device));
    FieldDev fd;
    int rc = makeFieldDev(ctx, f, nullptr, &fd);
    if (rc) return rc;
    HPSDF_HIP(launchFieldEval(ctx->stream, fd, ctx->dTables, dXyz, n, dOut));
    return HPSDF_OK;
}
int hpsdf_field_eval_lane_host(hpsdf_ctx* ctx, const hpsdf_field* f, const double* xyz, size_t n, double* out) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!f || (!xyz && n) || (!out && n)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return hostRoundTrip(
        ctx, xyz, n, out,
        [](hpsdf_ctx* c, const void* o, const double* d, size_t m, double* r) { return meshLaneDevice(c, (const hpsdf_field*)o, d, m, r); }, f);
    HPSDF_CATCH
}

int hpsdf_field_eval_naive_host(hpsdf_ctx* ctx, const hpsdf_field* f, const double* xyz, size_t n, double* out) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!f || (!xyz && n) || (!out && n)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return hostRoundTrip(
        ctx, xyz, n, out,
        [](hpsdf_ctx* c, const void* o, const double* d, size_t m, double* r) { return meshNaiveDevice(c, (const hpsdf_field*)o, d, m, r); }, f);
    HPSDF_CATCH
}

int hpsdf_selftest_acosf(hpsdf_ctx* ctx, uint32_t first_bits, uint32_t stride, size_t n, float* out) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!out && n) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return HPSDF_OK;
    HostArray arr[1] = {{nullptr, out, n * sizeof(float)}};
    return hostCall(ctx, arr, 1, [&] {
        HPSDF_HIP(launchAcosfSelftest(ctx->stream, first_bits, stride, n, (float*)arr[0].dev));
        return (int)HPSDF_OK;
    });
    HPSDF_CATCH
}

// ---------------------------------------------------------------------------- tree + query
int hpsdf_tree_upload(hpsdf_ctx* ctx, const void* block, size_t size, hpsdf_tree** out) {
    HPSDF_TRY
    if (!out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "trees are queried on the GPU: a device context is required");
    if (!block || size < 16 + sizeof(hpsdf_config)) return fail(HPSDF_ERR_BAD_BLOCK, "block too small");
    const uint8_t* p = (const uint8_t*)block;
    uint64_t nCoeffs, nNodes;
    std::memcpy(&nCoeffs, p, 8);
    if (nCoeffs > (size - 16 - sizeof(hpsdf_config)) / 8) return fail(HPSDF_ERR_BAD_BLOCK, "coefficient count exceeds block");
    const double* coeffs = (const double*)(p + 8);
    std::memcpy(&nNodes, p + 8 + 8 * nCoeffs, 8);
    const size_t need = 8 + 8 * (size_t)nCoeffs + 8 + sizeof(hpsdf_node) * (size_t)nNodes + sizeof(hpsdf_config);
    if (nNodes == 0 || nNodes > (size_t)0xFFFFFFF0u || need != size) return fail(HPSDF_ERR_BAD_BLOCK, "node count does not match block size");
    if (nCoeffs > 0xFFFFFFFFull) return fail(HPSDF_ERR_UNSUPPORTED, "more than 2^32 coefficients");
    std::vector<hpsdf_node> nodes(nNodes);
    std::memcpy(nodes.data(), p + 16 + 8 * nCoeffs, sizeof(hpsdf_node) * nNodes);
    hpsdf_config cfg;
    std::memcpy(&cfg, p + 16 + 8 * nCoeffs + sizeof(hpsdf_node) * nNodes, sizeof cfg);

    // device mirror: validate that the tree is the dyadic octree the descent recomputes
    const Tables& T = tables();
    std::vector<NodeRec> recs(nNodes, NodeRec{0, 0});
    if (nodes[0].degree != kInteriorDegree || nodes[0].child_idx == ~0ull)
        return fail(HPSDF_ERR_UNSUPPORTED, "root must be an interior node (Octree::CreateRoot always splits it)");
    for (int a = 0; a < 3; ++a)
        if (nodes[0].aabb_min[a] != -0.5f || nodes[0].aabb_max[a] != 0.5f)
            return fail(HPSDF_ERR_UNSUPPORTED, "internal root box must be [-0.5,0.5]^3 (Octree.cpp:798)");
    if (nNodes < 9) return fail(HPSDF_ERR_BAD_BLOCK, "an interior root needs its 8 children");
    BlockTreeInfo walk;
    {
        std::string why;
        const int vrc = checkBlockTree(nodes.data(), nNodes, nCoeffs, T.coeffCount, false, false, &walk, why);
        if (vrc) return fail(vrc, why);
    }
    std::vector<double> padded;
    padded.reserve(nCoeffs + 16 * nNodes);
    const uint64_t leaves = walk.leaves;
    const int maxDeg = walk.maxDegree, maxDepth = walk.maxDepth, minLeafDepth = walk.minLeafDepth;
    for (const uint64_t i : walk.order) {
        const hpsdf_node& n = nodes[i];
        if (n.degree == kInteriorDegree) {
            recs[i] = NodeRec{(uint32_t)n.child_idx, kInteriorTag};
            for (unsigned c = 0; c < 8; ++c) {
                const hpsdf_node& ch = nodes[n.child_idx + c];
                for (int d = 0; d < 3; ++d) {
                    const float mid = (n.aabb_max[d] + n.aabb_min[d]) * 0.5f;
                    const float emin = (c >> d) & 1u ? mid : n.aabb_min[d], emax = (c >> d) & 1u ? n.aabb_max[d] : mid;
                    if (ch.aabb_min[d] != emin || ch.aabb_max[d] != emax)
                        return fail(HPSDF_ERR_UNSUPPORTED, "child boxes are not midpoint octants of their parent");
                }
            }
        } else {
            // device mirror: every leaf's block starts on a 128-byte line (the wave-cooperative fetch of
            // query_general_kernel moves whole lines; a degree-2 leaf is one line, a degree-3 leaf two)
            recs[i] = NodeRec{(uint32_t)padded.size(), (uint32_t)n.degree};
            padded.insert(padded.end(), coeffs + n.coeffs_start, coeffs + n.coeffs_start + T.coeffCount[n.degree]);
            padded.resize((padded.size() + 15) & ~(size_t)15, 0.0);
        }
    }
    if (padded.size() > 0xFFFFFFF0ull) return fail(HPSDF_ERR_UNSUPPORTED, "more than 2^32 coefficients");
    // dense table of the deepest complete level (<= 5): table[path] = node reached by that octant path
    const int topDepth = std::max(1, std::min(5, minLeafDepth));
    std::vector<TopEntry> top((size_t)1 << (3 * topDepth));
    std::vector<NodeRec> topRec(top.size());
    for (size_t code = 0; code < top.size(); ++code) {  // code = x + side * (y + side * z), cell coordinates at topDepth
        const size_t mask = ((size_t)1 << topDepth) - 1;
        const size_t kx = code & mask, ky = (code >> topDepth) & mask, kz = code >> (2 * topDepth);
        uint64_t cur = 0;
        for (int l = topDepth - 1; l >= 0; --l)  // bit l of a coordinate picks the upper half at that level (Octree.cpp:1101)
            cur = nodes[cur].child_idx + ((kx >> l) & 1u) + 2 * ((ky >> l) & 1u) + 4 * ((kz >> l) & 1u);
        TopEntry& te = top[code];
        std::memset(&te, 0, sizeof te);
        te.a = recs[cur].a;
        te.b = recs[cur].b;
        topRec[code] = recs[cur];
        if (te.b <= 2u)  // small leaf: coefficients ride in the same line
            std::memcpy(te.c, coeffs + nodes[cur].coeffs_start, sizeof(double) * T.coeffCount[te.b]);
    }
    HPSDF_HIP(hipSetDevice(ctx->device));
    hpsdf_tree* t = new hpsdf_tree();
    t->device = ctx->device;
    t->nNodes = nNodes;
    t->nCoeffs = nCoeffs;
    t->nLeaves = leaves;
    t->maxDegree = maxDeg;
    t->maxDepth = maxDepth;
    t->allInline = maxDeg <= 2 && maxDepth <= topDepth;
    t->config = cfg;
    hipError_t e = hipMalloc((void**)&t->dNodes, nNodes * sizeof(NodeRec));
    if (e == hipSuccess) e = hipMalloc((void**)&t->dCoeffs, std::max<size_t>(2, padded.size()) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&t->dTop, top.size() * sizeof(TopEntry));
    if (e == hipSuccess) e = hipMalloc((void**)&t->dTopRec, topRec.size() * sizeof(NodeRec));
    if (e == hipSuccess) e = hipMemcpy(t->dTopRec, topRec.data(), topRec.size() * sizeof(NodeRec), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->dNodes, recs.data(), nNodes * sizeof(NodeRec), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->dTop, top.data(), top.size() * sizeof(TopEntry), hipMemcpyHostToDevice);
    if (e == hipSuccess && !padded.empty())
        e = hipMemcpy(t->dCoeffs, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hpsdf_tree_destroy(t);
        return hipFail(e, "tree upload");
    }
    t->dev.nodes = t->dNodes;
    t->dev.top = t->dTop;
    t->dev.topRec = t->dTopRec;
    t->dev.coeffs = t->dCoeffs;
    t->dev.topDepth = topDepth;
    t->dev.maxDegree = maxDeg;
    for (int j = 0; j < 3; ++j) t->dev.nlTop[j] = T.normalisedLengths[j][topDepth];
    for (int a = 0; a < 3; ++a) {
        t->dev.rootCentre[a] = (double)((cfg.root_min[a] + cfg.root_max[a]) / 2.0f);  // Octree.cpp:419
        t->dev.rootInvSizes[a] = (double)(1.0f / (cfg.root_max[a] - cfg.root_min[a]));  // Octree.cpp:420
    }
    t->paddedCount = padded.size();
    *out = t;
    return HPSDF_OK;
    HPSDF_CATCH
}

// The host copies a scalar-sized call works from (host_query.cpp), on first use: the device mirror's records and coefficients.
int hpsdf_tree::hostCopies() const {
    if (hostReady.load(std::memory_order_acquire)) return HPSDF_OK;
    std::lock_guard<std::mutex> guard(hostLock);
    if (hostReady.load(std::memory_order_relaxed)) return HPSDF_OK;
    HPSDF_HIP(hipSetDevice(device));
    hRecs.resize(nNodes);
    hPadded.resize(std::max<uint64_t>(2, paddedCount));
    HPSDF_HIP(hipMemcpy(hRecs.data(), dNodes, nNodes * sizeof(hpsdf::NodeRec), hipMemcpyDeviceToHost));  // (the upload's copies were synchronous)
    if (paddedCount) HPSDF_HIP(hipMemcpy(hPadded.data(), dCoeffs, paddedCount * sizeof(double), hipMemcpyDeviceToHost));
    hostReady.store(true, std::memory_order_release);
    return HPSDF_OK;
}

int hpsdf_tree_destroy(hpsdf_tree* t) {
    if (!t) return HPSDF_OK;
    (void)hipSetDevice(t->device);
    if (t->dNodes) (void)hipFree(t->dNodes);
    if (t->dTop) (void)hipFree(t->dTop);
    if (t->dTopRec) (void)hipFree(t->dTopRec);
    if (t->dCoeffs) (void)hipFree(t->dCoeffs);
    delete t;
    return HPSDF_OK;
}

int hpsdf_tree_info(const hpsdf_tree* t, uint64_t* nNodes, uint64_t* nCoeffs, uint64_t* nLeaves, int* maxDegree,
                    int* maxDepth) {
    if (!t) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null tree");
    if (nNodes) *nNodes = t->nNodes;
    if (nCoeffs) *nCoeffs = t->nCoeffs;
    if (nLeaves) *nLeaves = t->nLeaves;
    if (maxDegree) *maxDegree = t->maxDegree;
    if (maxDepth) *maxDepth = t->maxDepth;
    return HPSDF_OK;
}

// Query / QueryWithGradient over device arrays (dGrad == nullptr: values only)
static int queryDevice(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* dXyz, size_t n, double* dOut, double* dGrad) {
    if (t->device != ctx->device) return fail(HPSDF_ERR_INVALID_ARGUMENT, "tree lives on another device");
    HPSDF_HIP(hipSetDevice(ctx->device));
    const size_t kChunk = (size_t)1 << 31;  // deferred indices are 32-bit
    for (size_t off = 0; off < n; off += kChunk) {
        const size_t m = std::min(kChunk, n - off);
        // Query* is const and callable from many threads in the reference (Octree.h:71-78): the deferred-point scratch is
        // the one piece of context state these entry points touch -- grown and handed to the launch under one lock
        // (launches on the context stream run in order, so sharing the buffer between them is safe)
        std::unique_lock<std::mutex> guard(ctx->scratchLock, std::defer_lock);
        if (t->maxDegree > 3) {
            guard.lock();
            if (!ctx->dDeferCount) HPSDF_HIP(hipMalloc((void**)&ctx->dDeferCount, (2 * kQueryMaxGrid + 1) * sizeof(uint32_t)));
            const size_t need = m + (size_t)256 * kQueryMaxGrid + 4096;  // every workgroup's run is whole tiles
            if (ctx->deferCap < need) {
                if (ctx->dDefer) HPSDF_HIP(hipFree(ctx->dDefer));
                ctx->dDefer = nullptr;
                ctx->deferCap = 0;
                HPSDF_HIP(hipMalloc((void**)&ctx->dDefer, need * sizeof(uint32_t)));
                ctx->deferCap = need;
            }
        }
        TreeDev td = t->dev;
        td.leftAssoc = reductionLeftAssoc(ctx);
        HPSDF_HIP(launchQuery(ctx->stream, td, ctx->dTables, dXyz + 3 * off, m, dOut + off, dGrad ? dGrad + 3 * off : nullptr,
                              t->allInline, ctx->dDeferCount, ctx->dDefer));
    }
    return HPSDF_OK;
}

int hpsdf_query_device(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* dXyz, size_t n, double* dOut) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!t || (!dXyz && n) || (!dOut && n)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return queryDevice(ctx, t, dXyz, n, dOut, nullptr);
    HPSDF_CATCH
}

int hpsdf_query_host(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* xyz, size_t n, double* out) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!t || (!xyz && n) || (!out && n)) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (n <= hostQueryLimit() && smallQueriesOnHost()) {  // a scalar Query(pt): ~0.05 us here, ~15 us as a launch (host_query.cpp)
        if (const int hc = t->hostCopies()) return hc;
        for (size_t i = 0; i < n; ++i) out[i] = hostQueryPoint(*t, xyz + 3 * i);
        return HPSDF_OK;
    }
    return hostRoundTrip(
        ctx, xyz, n, out,
        [](hpsdf_ctx* c, const void* o, const double* d, size_t m, double* r) {
            return hpsdf_query_device(c, (const hpsdf_tree*)o, d, m, r);
        },
        t);
    HPSDF_CATCH
}

int hpsdf_query_gradient_device(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* dXyz, size_t n, double* dOut,
                                double* dGrad) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!t || (n && (!dXyz || !dOut || !dGrad))) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return queryDevice(ctx, t, dXyz, n, dOut, dGrad);
    HPSDF_CATCH
}

int hpsdf_query_gradient_host(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* xyz, size_t n, double* out,
                              double* grad) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!t || (n && (!xyz || !out || !grad))) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return HPSDF_OK;
    if (n <= hostGradientLimit() && smallQueriesOnHost()) {
        if (const int hc = t->hostCopies()) return hc;
        for (size_t i = 0; i < n; ++i) hostQueryPointWithGradient(*t, xyz + 3 * i, out + i, grad + 3 * i, reductionLeftAssoc(ctx));
        return HPSDF_OK;
    }
    // rows of points outside the root keep what the caller passed in (the reference leaves its output untouched)
    HostArray arr[3] = {{xyz, nullptr, n * 3 * sizeof(double)}, {nullptr, out, n * sizeof(double)}, {grad, grad, n * 3 * sizeof(double)}};
    return hostCall(ctx, arr, 3, [&] {
        return hpsdf_query_gradient_device(ctx, t, (const double*)arr[0].dev, n, (double*)arr[1].dev, (double*)arr[2].dev);
    });
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_query_ray_device(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* dOrigins, const double* dDirs,
                           const double* dTMax, size_t n, uint8_t* dHit, double* dT) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!t || (n && (!dOrigins || !dDirs || !dTMax || !dHit || !dT))) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (t->device != ctx->device) return fail(HPSDF_ERR_INVALID_ARGUMENT, "tree lives on another device");
    HPSDF_HIP(hipSetDevice(ctx->device));
    HPSDF_HIP(launchQueryRay(ctx->stream, t->dev, ctx->dTables, dOrigins, dDirs, dTMax, n, dHit, dT));
    return HPSDF_OK;
    HPSDF_CATCH
}

namespace {
// frees its device buffers on every exit path
struct DevBufs {
    std::vector<void*> p;
    ~DevBufs() {
        for (void* q : p)
            if (q) (void)hipFree(q);
    }
    hipError_t alloc(void** out, size_t bytes) {
        hipError_t e = hipMalloc(out, bytes ? bytes : 8);
        if (e == hipSuccess) p.push_back(*out);
        return e;
    }
};
}  // namespace

int hpsdf_query_ray_host(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* origins, const double* dirs,
                         const double* tMax, size_t n, uint8_t* hit, double* tOut) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!t || (n && (!origins || !dirs || !tMax || !hit || !tOut))) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return HPSDF_OK;
    if (n <= hostRayLimit() && smallQueriesOnHost()) {  // a scalar QueryRay(ray, tMax, t): <= 200 host Query steps instead of a launch
        if (const int hc = t->hostCopies()) return hc;
        for (size_t i = 0; i < n; ++i) hit[i] = hostQueryRay(*t, origins + 3 * i, dirs + 3 * i, tMax[i], tOut + i) ? 1 : 0;
        return HPSDF_OK;
    }
    // t of a miss keeps the caller's value (the reference leaves t_ untouched)
    HostArray arr[5] = {{origins, nullptr, n * 3 * sizeof(double)}, {dirs, nullptr, n * 3 * sizeof(double)},
                        {tMax, nullptr, n * sizeof(double)},        {tOut, tOut, n * sizeof(double)},
                        {nullptr, hit, n}};
    return hostCall(ctx, arr, 5, [&] {
        return hpsdf_query_ray_device(ctx, t, (const double*)arr[0].dev, (const double*)arr[1].dev, (const double*)arr[2].dev, n,
                                      (uint8_t*)arr[4].dev, (double*)arr[3].dev);
    });
    HPSDF_CATCH
}

static int queryDevice(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* dXyz, size_t n, double* dOut, double* dGrad);

// (u8)f64 as the reference's x86-64 build performs it: truncating conversion to a 32-bit integer
// (out of range and NaN give INT_MIN), then the low byte
static uint8_t f64ToU8(double q) {
    int32_t v;
    if (!(q > -2147483649.0 && q < 2147483648.0))
        v = INT32_MIN;
    else
        v = (int32_t)q;
    return (uint8_t)(v & 0xFF);
}

int hpsdf_function_slice(hpsdf_ctx* ctx, const hpsdf_tree* t, double c, const float* viewMin, const float* viewMax,
                         uint64_t nSamples, uint8_t* rgb, double* values) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!t || !viewMin || !viewMax || !rgb) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (nSamples == 0 || nSamples > 32768) return fail(HPSDF_ERR_INVALID_ARGUMENT, "n_samples must be in [1, 32768]");
    if (t->device != ctx->device) return fail(HPSDF_ERR_INVALID_ARGUMENT, "tree lives on another device");
    HPSDF_HIP(hipSetDevice(ctx->device));
    const size_t total = (size_t)nSamples * nSamples;
    DevBufs bufs;
    double *dV = nullptr, *dP = nullptr;
    HPSDF_HIP(bufs.alloc((void**)&dV, total * sizeof(double)));
    HPSDF_HIP(bufs.alloc((void**)&dP, total * 3 * sizeof(double)));
    const float step = (viewMax[0] - viewMin[0]) / (float)nSamples;  // Octree.cpp:1149
    HPSDF_HIP(launchSlicePoints(ctx->stream, c, viewMin[0], viewMin[1], step, (uint32_t)nSamples, dP));
    {
        const int rc = queryDevice(ctx, t, dP, total, dV, nullptr);
        if (rc) return rc;
    }
    std::vector<double> own;
    double* v = values;
    if (!v) {
        own.resize(total);
        v = own.data();
    }
    HPSDF_HIP(hipMemcpyAsync(v, dV, total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HPSDF_HIP(hipStreamSynchronize(ctx->stream));
    // Octree.cpp:1140-1167 ranges, :1172-1196 bytes
    double posFirst = DBL_MAX, posSecond = 0.0, negFirst = 0.0, negSecond = DBL_MAX * -1.0;
    for (size_t p = 0; p < total; ++p) {
        const double s = v[p];
        if (s > (double)0.000001f) {
            posFirst = std::min(s, posFirst);
            posSecond = std::max(s, posSecond);
        } else {
            negFirst = std::min(s, negFirst);
            negSecond = std::max(s, negSecond);
        }
    }
    for (size_t p = 0; p < total; ++p) {
        const float u = (float)v[p];
        uint8_t* px = rgb + 3 * p;
        if (u > 0.0f) {
            px[0] = 0, px[1] = f64ToU8(255 * ((double)u - posSecond) / (posFirst - posSecond)), px[2] = 0;
        } else {
            px[0] = 0, px[1] = 0, px[2] = f64ToU8(255 * ((double)u - negFirst) / (negSecond - negFirst));
        }
    }
    return HPSDF_OK;
    HPSDF_CATCH
}

// ---------------------------------------------------------------------------- build
int hpsdf_build_begin(const hpsdf_config* cfg, const hpsdf_build_opts* opts, hpsdf_build** out) {
    HPSDF_TRY
    if (!cfg || !out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    hpsdf_build* b = new hpsdf_build();
    int rc = builderBegin(b, cfg, opts);
    if (rc) {
        delete b;
        return rc;
    }
    *out = b;
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_build_destroy(hpsdf_build* b) {
    delete b;
    return HPSDF_OK;
}

int hpsdf_build_round_select(hpsdf_build* b, uint64_t* nJobs) {
    HPSDF_TRY
    if (!b || !nJobs) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return builderSelect(b, nJobs);
    HPSDF_CATCH
}

int hpsdf_build_round_jobs(const hpsdf_build* b, hpsdf_job* out) {
    HPSDF_TRY
    if (!b || !out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return builderJobs(b, out);
    HPSDF_CATCH
}

int hpsdf_build_round_slice(const hpsdf_build* b, int rank, uint64_t* first, uint64_t* count) {
    if (!b || rank < 0 || rank >= b->world) return fail(HPSDF_ERR_INVALID_ARGUMENT, "bad rank");
    if (!b->roundOpen) return fail(HPSDF_ERR_STATE, "no open round");
    if (first) *first = b->slices[rank].first;
    if (count) *count = b->slices[rank].count;
    return HPSDF_OK;
}

int hpsdf_build_round_max_slice(const hpsdf_build* b, uint64_t* n) {
    if (!b || !n) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (!b->roundOpen) return fail(HPSDF_ERR_STATE, "no open round");
    uint64_t m = 0;
    for (const auto& s : b->slices) m = std::max(m, s.count);
    *n = m;
    return HPSDF_OK;
}

int hpsdf_build_round_compute(hpsdf_build* b, hpsdf_ctx* ctx, const hpsdf_field* field) {
    HPSDF_TRY
    if (!b) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null build");
    return builderCompute(b, ctx, field);
    HPSDF_CATCH
}

int hpsdf_build_round_results_device(hpsdf_build* b, double** dHeaders, uint64_t* n) {
    if (!b) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null build");
    if (!b->roundOpen || !b->computed) return fail(HPSDF_ERR_STATE, "round not computed");
    if (b->weighted) return fail(HPSDF_ERR_UNSUPPORTED, "weighted builds apply the weight on the host: use hpsdf_build_round_results_host");
    if (dHeaders) *dHeaders = b->ws ? b->ws->errs.dev : nullptr;
    if (n) *n = b->slices[b->rank].count * HPSDF_JOB_HEADER_DOUBLES;
    return HPSDF_OK;
}

int hpsdf_build_round_results_host(hpsdf_build* b, hpsdf_ctx* ctx, double* out) {
    HPSDF_TRY
    if (!b || !ctx || !out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (!b->roundOpen || !b->computed) return fail(HPSDF_ERR_STATE, "round not computed");
    const uint64_t n = b->slices[b->rank].count * HPSDF_JOB_HEADER_DOUBLES;
    HPSDF_HIP(hipSetDevice(ctx->device));
    const uint64_t nSlots = std::max<uint64_t>(1, n);
    if (n) {
        HPSDF_HIP(hipMemcpyAsync(b->ws->errs.host, b->ws->errs.dev, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (b->weighted)
            HPSDF_HIP(hipMemcpyAsync(b->ws->errs.host + nSlots, b->ws->errs.dev + nSlots, n * sizeof(double),
                                     hipMemcpyDeviceToHost, ctx->stream));
    }
    HPSDF_HIP(hipStreamSynchronize(ctx->stream));
    if (n) std::memcpy(out, b->ws->errs.host, n * sizeof(double));
    if (b->weighted) {
        // Octree.cpp:1078-1086: error * weight, the weight from |mean FApprox| (:1224-1226 / :1246)
        const double d = std::sqrt(3.0), strength = b->cfg.weighting_strength;
        const double* mean = b->ws->errs.host + nSlots;
        for (uint64_t i = 0; i < n; ++i) {
            double w;
            if (b->cfg.weighting_type == 1) {
                const double k = std::pow(1.0 - mean[i] / d, strength);
                w = std::min<double>(1.0, std::max<double>(k, 0.0));
            } else {
                w = std::exp(-1.0 * strength * mean[i] / d);
            }
            out[i] = out[i] * w;
        }
    }
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_build_round_apply(hpsdf_build* b, const double* headers) {
    HPSDF_TRY
    if (!b || !headers) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return builderApply(b, headers);
    HPSDF_CATCH
}

int hpsdf_build_rows_counts(const hpsdf_build* b, uint64_t* counts_per_rank) {
    HPSDF_TRY
    if (!b || !counts_per_rank) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return builderRowsCounts(b, counts_per_rank);
    HPSDF_CATCH
}

int hpsdf_build_rows_pack_host(hpsdf_build* b, hpsdf_ctx* ctx, double* out) {
    HPSDF_TRY
    if (!b || !out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return builderRowsPackHost(b, ctx, out);
    HPSDF_CATCH
}

int hpsdf_build_rows_unpack_host(hpsdf_build* b, hpsdf_ctx* ctx, const double* const* parts) {
    HPSDF_TRY
    if (!b) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null build");
    return builderRowsUnpackHost(b, ctx, parts);
    HPSDF_CATCH
}

int hpsdf_build_node_rows_host(hpsdf_build* b, hpsdf_ctx* ctx, uint64_t node_idx, double* out, uint64_t* n_rows) {
    HPSDF_TRY
    if (!b) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null build");
    return builderNodeRowsHost(b, ctx, node_idx, out, n_rows);
    HPSDF_CATCH
}

int hpsdf_build_round_inject(hpsdf_build* b, uint64_t job, const double* pCoeffs, const double* hCoeffs) {
    HPSDF_TRY
    if (!b) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null build");
    return builderInject(b, job, pCoeffs, hCoeffs);
    HPSDF_CATCH
}

int hpsdf_build_layout(hpsdf_build* b, uint64_t* nCoeffsTotal, uint64_t* counts) {
    HPSDF_TRY
    if (!b) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null build");
    int rc = builderLayout(b);
    if (rc) return rc;
    if (nCoeffsTotal) *nCoeffsTotal = b->nCoeffsTotal;
    if (counts)
        for (int r = 0; r < b->world; ++r) counts[r] = b->packCounts[r];
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_build_pack_device(hpsdf_build* b, hpsdf_ctx* ctx, double** dPack, uint64_t* n) {
    HPSDF_TRY
    if (!b) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null build");
    return builderPackDevice(b, ctx, dPack, n);
    HPSDF_CATCH
}

int hpsdf_build_pack_host(hpsdf_build* b, hpsdf_ctx* ctx, double* out) {
    HPSDF_TRY
    if (!b || !out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return builderPackHost(b, ctx, out);
    HPSDF_CATCH
}

int hpsdf_build_assemble(hpsdf_build* b, const double* const* packs, void** block, size_t* size) {
    HPSDF_TRY
    if (!b || !block || !size) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    return builderAssemble(b, packs, block, size);
    HPSDF_CATCH
}

int hpsdf_build_get_stats(const hpsdf_build* b, hpsdf_build_stats* out) {
    if (!b || !out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    *out = b->stats;
    out->total_error = b->total;
    out->n_nodes = b->nodes.size();
    return HPSDF_OK;
}

// A block that builderAssemble malloc'd, for a context with its own block allocator (hpsdf_ctx_set_block_allocator): moved there.
static int adoptBlock(hpsdf_ctx* ctx, void** block, size_t size) {
    if (!ctx->blockAlloc) return HPSDF_OK;
    void* q = ctx->allocBlock(size);
    if (!q) {
        std::free(*block);
        *block = nullptr;
        return fail(HPSDF_ERR_OUT_OF_MEMORY, "the block allocator returned no memory");
    }
    std::memcpy(q, *block, size);
    std::free(*block);
    *block = q;
    return HPSDF_OK;
}

void hpsdf_ctx_set_block_allocator(hpsdf_ctx* ctx, hpsdf_block_alloc_fn alloc, hpsdf_block_release_fn release, void* user) {
    if (!ctx) return;
    ctx->blockAlloc = alloc, ctx->blockRelease = alloc ? release : nullptr, ctx->blockUser = alloc ? user : nullptr;
}

int hpsdf_create(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, uint64_t K, void** block,
                 size_t* size, hpsdf_build_stats* stats) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "Create runs on the GPU: a device context is required");
    if (!cfg || !field || !block || !size) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (frontierEligible(ctx, cfg, field, K)) {  // frontier, decision and bookkeeping on the device (frontier.hip)
        int frc = frontierCreate(ctx, cfg, field, K, block, size, stats);
        std::memset(&g_lastContinuity, 0, sizeof g_lastContinuity);
        if (!frc && cfg->continuity_enforce) {  // Octree.cpp:341-344
            std::string err;
            frc = continuityPostProcess(*block, *size, 0.0, 0, 0, &g_lastContinuity, err, ctx);
            if (frc) {
                setError(err);
                ctx->freeBlock(*block);
                *block = nullptr;
                *size = 0;
            }
        }
        return frc;
    }
    hpsdf_build_opts o;
    std::memset(&o, 0, sizeof o);
    o.max_jobs_per_round = K;
    o.rank = 0;
    o.world = 1;
    const bool trace = std::getenv("HPSDF_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tSel = 0, tCmp = 0, tRes = 0, tApp = 0, t0 = now(), t1;
    hpsdf_build* b = nullptr;
    int rc = hpsdf_build_begin(cfg, &o, &b);
    if (rc) return rc;
    // whatever throws below (vector growth, std::thread creation in builderCompute ...), the build's destructor runs and
    // hands the context's workspace back (ws->inUse)
    std::unique_ptr<hpsdf_build> owner(b);
    const double tBegin = now() - t0;
    std::vector<double> headers;
    for (;;) {
        uint64_t n = 0;
        t1 = now();
        if ((rc = builderSelect(b, &n))) break;
        tSel += now() - t1;
        if (n == 0) break;
        headers.resize(n * HPSDF_JOB_HEADER_DOUBLES);
        t1 = now();
        if ((rc = builderCompute(b, ctx, field))) break;
        tCmp += now() - t1;
        t1 = now();
        if ((rc = hpsdf_build_round_results_host(b, ctx, headers.data()))) break;
        tRes += now() - t1;
        t1 = now();
        if ((rc = builderApply(b, headers.data()))) break;
        tApp += now() - t1;
    }
    t1 = now();
    if (!rc) rc = builderLayout(b);
    const double tLay = now() - t1;
    t1 = now();
    double tPack = 0, tAsm = 0;
    if (!rc) {
        std::vector<double> pack(std::max<uint64_t>(1, b->packCounts[0]));
        rc = builderPackHost(b, ctx, pack.data());
        tPack = now() - t1;
        t1 = now();
        const double* packs[1] = {pack.data()};
        if (!rc) rc = builderAssemble(b, packs, block, size);
        if (!rc) rc = adoptBlock(ctx, block, *size);
        tAsm = now() - t1;
    }
    if (!rc && stats) hpsdf_build_get_stats(b, stats);
    t1 = now();
    std::memset(&g_lastContinuity, 0, sizeof g_lastContinuity);
    if (!rc && cfg->continuity_enforce) {  // Octree.cpp:341-344
        std::string err;
        rc = continuityPostProcess(*block, *size, 0.0, 0, 0, &g_lastContinuity, err, ctx);
        if (rc) {
            setError(err);
            ctx->freeBlock(*block);
            *block = nullptr;
            *size = 0;
        }
    }
    const double tCont = now() - t1;
    t1 = now();
    owner.reset();
    if (trace)
        std::fprintf(stderr, "[hpsdf_create] us: begin %.0f select %.0f compute(host+launch) %.0f results(wait+D2H) %.0f apply %.0f "
                             "layout %.0f pack %.0f assemble %.0f continuity %.0f destroy %.0f total %.0f\n",
                     tBegin, tSel, tCmp, tRes, tApp, tLay, tPack, tAsm, tCont, now() - t1, now() - t0);
    return rc;
    HPSDF_CATCH
}

// The sharded build for what the host scheduler owns (host callbacks, nearness weighting, logging, K > 4096): the round
// loop of hp-adaptive-..._amd/distributed.py in C++, over the caller's all-gather.  The scheduler keeps its results in host
// memory (the weight's pow / exp run there), so every exchange is staged through one device buffer: this rank's part up,
// the in-place all-gather, everything down.  Per round: the 9 errors of every job; for weighted builds also the arrays the
// round accepted (hpsdf_build_rows_*); at the end the packed coefficients.
static int createShardedOnHostScheduler(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, uint64_t K, int rank, int world,
                                        hpsdf_allgather_fn gather, void* user, void** block, size_t* size, hpsdf_build_stats* stats) {
    hpsdf_build_opts o;
    std::memset(&o, 0, sizeof o);
    o.max_jobs_per_round = K;
    o.rank = rank;
    o.world = world;
    hpsdf_build* b = nullptr;
    int rc = hpsdf_build_begin(cfg, &o, &b);
    if (rc) return rc;
    std::unique_ptr<hpsdf_build> owner(b);
    HPSDF_HIP(hipSetDevice(ctx->device));
    struct Stage {
        double* d = nullptr;
        uint64_t cap = 0;
        ~Stage() {
            if (d) (void)hipFree(d);
        }
    } stage;
    std::vector<double> all;
    uint64_t stride = 0;  // doubles per rank in `all` after an exchange
    // every rank's `mine` (count doubles, padded to `pad`) -> all[world][stride], stride = pad + 1: the last double of a rank's
    // part is its STATUS (0: fine).  A rank whose share of the round failed (localRc: device memory, a launch) still enters the
    // exchange, with its status set, and returns its own error afterwards; the others find the status, and return
    // HPSDF_ERR_STATE instead of waiting in the next collective for a rank that has left.
#ifdef HPSDF_TEST_HOOKS  // (lib/libhpsdf_hooks.so, built for tests/: the production library does not look at the variable)
    const char* injected = std::getenv("HPSDF_TEST_FAIL_RANK");  // "<rank>:<exchange number>"
#else
    constexpr const char* injected = nullptr;  // (the statements that look at it fold away: the production library holds no trace of the hook)
#endif
    int exchanges = 0;
    auto exchange = [&](const double* mine, uint64_t count, uint64_t pad, const char* what, int localRc) -> int {
        if (injected && !localRc && std::atoi(injected) == rank && std::strchr(injected, ':') && std::atoi(std::strchr(injected, ':') + 1) == exchanges)
            localRc = fail(HPSDF_ERR_OUT_OF_MEMORY, kInjectedFailureMsg);
        ++exchanges;
        std::string ownError = localRc ? std::string(hpsdf_last_error()) : std::string();
        stride = pad + 1;
        const uint64_t total = (uint64_t)world * stride;
        if (stage.cap < total) {
            if (stage.d) (void)hipFree(stage.d);
            stage.d = nullptr, stage.cap = 0;
            uint64_t want = total + total / 2;
            hipError_t me = hipMalloc((void**)&stage.d, want * sizeof(double));
            if (me != hipSuccess) {  // without the headroom, then: what the exchange itself needs
                (void)hipGetLastError();
                want = total;
                me = hipMalloc((void**)&stage.d, want * sizeof(double));
            }
            // (no buffer at all: this rank has nothing to enter the in-place exchange with -- the one failure the status word cannot carry)
            if (me != hipSuccess) return localRc ? fail(localRc, ownError) : hipFail(me, "staging buffer of the exchange");
            stage.cap = want;
        }
        // a failing copy is this rank's failure like any other: it enters the exchange with its status set (the peers leave with
        // HPSDF_ERR_STATE instead of waiting for it in the next collective) and returns its own error afterwards
        if (count && !localRc) {
            const hipError_t ce = hipMemcpyAsync(stage.d + (uint64_t)rank * stride, mine, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
            if (ce != hipSuccess) localRc = hipFail(ce, "upload of this rank's part of the exchange"), ownError = hpsdf_last_error();
        }
        const double status = (double)localRc;
        {
            const hipError_t ce = hipMemcpyAsync(stage.d + (uint64_t)rank * stride + pad, &status, sizeof(double), hipMemcpyHostToDevice, ctx->stream);
            if (ce != hipSuccess && !localRc) localRc = hipFail(ce, "upload of this rank's status word"), ownError = hpsdf_last_error();
        }
        const int grc = gather(user, stage.d, stride * sizeof(double), (void*)ctx->stream);
        if (localRc) return fail(localRc, ownError);
        if (grc) return fail(HPSDF_ERR_STATE, std::string("the all-gather callback failed (") + what + "): " + std::to_string(grc));
        all.resize(total);
        HPSDF_HIP(hipMemcpyAsync(all.data(), stage.d, total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HPSDF_HIP(hipStreamSynchronize(ctx->stream));
        for (int r = 0; r < world; ++r)
            if (all[(uint64_t)r * stride + pad] != 0.0)
                return fail(HPSDF_ERR_STATE, "rank " + std::to_string(r) + " failed in this round (status " + std::to_string((int)all[(uint64_t)r * stride + pad]) +
                                                 ", " + what + "): its own error was returned there");
        return HPSDF_OK;
    };
    std::vector<double> mine, headers;
    std::vector<uint64_t> counts(world);
    std::vector<const double*> parts(world);
    for (;;) {
        uint64_t n = 0;
        if ((rc = builderSelect(b, &n))) return rc;
        if (n == 0) break;
        // (this rank's share of the round: what can fail here fails on this rank alone -- the exchange carries the news)
        rc = builderCompute(b, ctx, field);
        const uint64_t myCount = b->slices[rank].count * HPSDF_JOB_HEADER_DOUBLES;
        uint64_t maxSlice = 0;
        for (const auto& sl : b->slices) maxSlice = std::max(maxSlice, sl.count);
        mine.resize(std::max<uint64_t>(1, myCount));
        if (!rc) rc = hpsdf_build_round_results_host(b, ctx, mine.data());
        const uint64_t pad = maxSlice * HPSDF_JOB_HEADER_DOUBLES;
        if ((rc = exchange(mine.data(), myCount, pad, "a round's errors", rc))) return rc;
        headers.resize(n * HPSDF_JOB_HEADER_DOUBLES);
        for (int r = 0; r < world; ++r)
            if (b->slices[r].count)
                std::memcpy(headers.data() + b->slices[r].first * HPSDF_JOB_HEADER_DOUBLES, all.data() + (uint64_t)r * stride,
                            b->slices[r].count * HPSDF_JOB_HEADER_DOUBLES * sizeof(double));
        if ((rc = builderApply(b, headers.data()))) return rc;
        if (b->weighted) {  // the arrays this round accepted go to every rank (Octree.cpp:847 copies a node's previous rows)
            if ((rc = builderRowsCounts(b, counts.data()))) return rc;
            uint64_t rpad = 0;
            for (uint64_t c : counts) rpad = std::max(rpad, c);
            if (rpad) {
                mine.resize(std::max<uint64_t>(1, counts[rank]));
                rc = builderRowsPackHost(b, ctx, mine.data());
                if ((rc = exchange(mine.data(), counts[rank], rpad, "a round's accepted rows", rc))) return rc;
                for (int r = 0; r < world; ++r) parts[r] = all.data() + (uint64_t)r * stride;
                if ((rc = builderRowsUnpackHost(b, ctx, parts.data()))) return rc;
            }
        }
    }
    if ((rc = builderLayout(b))) return rc;
    uint64_t ppad = 1;
    for (uint64_t c : b->packCounts) ppad = std::max(ppad, c);
    mine.resize(std::max<uint64_t>(1, b->packCounts[rank]));
    rc = builderPackHost(b, ctx, mine.data());
    if ((rc = exchange(mine.data(), b->packCounts[rank], ppad, "the packed coefficients", rc))) return rc;
    for (int r = 0; r < world; ++r) parts[r] = all.data() + (uint64_t)r * stride;
    if ((rc = builderAssemble(b, parts.data(), block, size))) return rc;
    if ((rc = adoptBlock(ctx, block, *size))) return rc;
    if (stats) hpsdf_build_get_stats(b, stats);
    return HPSDF_OK;
}

int hpsdf_create_distributed(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, uint64_t K, int rank, int world,
                             hpsdf_allgather_fn gather, void* user, void** block, size_t* size, hpsdf_build_stats* stats) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "Create runs on the GPU: a device context is required");
    if (!cfg || !field || !block || !size) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    if (world == 1) return hpsdf_create(ctx, cfg, field, K, block, size, stats);
    if (world < 1 || rank < 0 || rank >= world) return fail(HPSDF_ERR_INVALID_ARGUMENT, "rank outside [0, world)");
    if (!gather) return fail(HPSDF_ERR_INVALID_ARGUMENT, "an all-gather callback is required for world > 1");
    // (the device-side frontier packs a segment's owner rank into three bits: more than 8 ranks take the host scheduler's rounds)
    // (... and a weighted incremental fit needs the node's previous rows, which another rank may hold: the host scheduler's
    // sharded rounds hand them over)
    int rc = (frontierEligible(ctx, cfg, field, K) && world <= 8)
                 ? frontierCreate(ctx, cfg, field, K, block, size, stats, rank, world, gather, user)
                 : createShardedOnHostScheduler(ctx, cfg, field, K, rank, world, gather, user, block, size, stats);
    std::memset(&g_lastContinuity, 0, sizeof g_lastContinuity);
    if (!rc && cfg->continuity_enforce) {  // Octree.cpp:341-344: every rank on its identical copy (deterministic: no exchange)
        std::string err;
        rc = continuityPostProcess(*block, *size, 0.0, 0, 0, &g_lastContinuity, err, ctx);
        if (rc) {
            setError(err);
            ctx->freeBlock(*block);
            *block = nullptr;
            *size = 0;
        }
    }
    return rc;
    HPSDF_CATCH
}

// ---------------------------------------------------------------------------- continuity (host)
int hpsdf_continuity_post_process(void* block, size_t size, double tol, int maxIter, uint64_t threads,
                                  hpsdf_continuity_stats* stats) {
    HPSDF_TRY
    std::string err;
    const int rc = continuityPostProcess(block, size, tol, maxIter, threads, stats, err);
    if (rc) return fail(rc, err);
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_continuity_post_process_device(hpsdf_ctx* ctx, void* block, size_t size, double tol, int maxIter, uint64_t threads,
                                         hpsdf_continuity_stats* stats) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required (hpsdf_continuity_post_process solves on the host)");
    std::string err;
    const int rc = continuityPostProcess(block, size, tol, maxIter, threads, stats, err, ctx);
    if (rc) return fail(rc, err);
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_continuity_matrix(const void* block, size_t size, uint64_t threads, uint64_t** rowPtr, uint64_t** col,
                            double** val, hpsdf_continuity_stats* stats) {
    HPSDF_TRY
    if (!rowPtr || !col || !val) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null output");
    std::string err;
    const int rc = continuityMatrix(block, size, threads, rowPtr, col, val, stats, err);
    if (rc) return fail(rc, err);
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_continuity_matrix_device(hpsdf_ctx* ctx, const void* block, size_t size, uint64_t** rowPtr, uint64_t** col, double** val,
                                   hpsdf_continuity_stats* stats) {
    HPSDF_TRY
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!block || !rowPtr || !col || !val) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    std::string err;
    const int rc = continuityMatrixDevice(ctx, block, size, rowPtr, col, val, stats, err);
    if (rc) return fail(rc, err);
    return HPSDF_OK;
    HPSDF_CATCH
}

int hpsdf_continuity_last_stats(hpsdf_continuity_stats* out) {
    if (!out) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null out");
    *out = g_lastContinuity;
    return HPSDF_OK;
}

// ---------------------------------------------------------------------------- micro-benchmark
static int benchFit(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, int degree, int depth, uint64_t nCells, int repeats,
                    double* msPerLaunch, double* coeffsOut, double* errsOut);
int hpsdf_bench_fit(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, int degree, int depth,
                    uint64_t nCells, int repeats, double* msPerLaunch) {
    HPSDF_TRY
    if (!msPerLaunch) return fail(HPSDF_ERR_INVALID_ARGUMENT, "bad bench arguments");
    return benchFit(ctx, cfg, field, degree, depth, nCells, repeats, msPerLaunch, nullptr, nullptr);
    HPSDF_CATCH
}
int hpsdf_fit_cells(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, int degree, int depth, uint64_t nCells, double* coeffs,
                    double* errs) {
    HPSDF_TRY
    if (!coeffs || !errs) return fail(HPSDF_ERR_INVALID_ARGUMENT, "null argument");
    double ms = 0.0;
    return benchFit(ctx, cfg, field, degree, depth, nCells, 1, &ms, coeffs, errs);
    HPSDF_CATCH
}
static int benchFit(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, int degree, int depth, uint64_t nCells, int repeats,
                    double* msPerLaunch, double* coeffsOut, double* errsOut) {
    {
    if (!ctx) return fail(HPSDF_ERR_NO_DEVICE, "a device context is required");
    if (!cfg || !field || !msPerLaunch || degree < 0 || degree > kMaxDegree || depth < 0 || depth > kMaxDepth || nCells == 0 ||
        repeats < 1)
        return fail(HPSDF_ERR_INVALID_ARGUMENT, "bad bench arguments");
    if (innermost(field)->kind == kHostCallback) return fail(HPSDF_ERR_UNSUPPORTED, "bench_fit needs a device field");
    HPSDF_HIP(hipSetDevice(ctx->device));
    const Tables& T = tables();
    const uint64_t nc = T.coeffCount[degree];
    const int nrows = (int)nc;
    FitShape shape = fitShape(degree, nrows, (uint32_t)std::min<uint64_t>(nCells, 0xFFFFFFFFull), false);
    bool fast = false, split = false;
    {
        FieldDev probe;
        if (ctx->fitMode == HPSDF_FIT_FAST && makeFieldDev(ctx, field, nullptr, &probe) == HPSDF_OK && fitMfmaSupports(degree, probe)) fast = true;
        // the default: top-degree rows exact, the rows below them on the matrix cores from the samples the exact kernel leaves
        if (ctx->fitMode == HPSDF_FIT_SPLIT && fitSplitSupports(degree, ctx->splitMinDegree) && innermost(field)->kind != kHostMesh) split = true;
    }
    const uint64_t rowsTop = nc - (degree > 0 ? T.coeffCount[degree - 1] : 0);
    if (fast) shape = FitShape{kMfmaCells, 1, 1, 0};  // the matrix-core fit: one workgroup per 16-cell tile
    if (split) shape = fitShape(degree, (int)rowsTop, (uint32_t)std::min<uint64_t>(nCells, 0xFFFFFFFFull), false);
    const uint64_t nq3 = (4 * (uint64_t)degree + 1) * (4 * (uint64_t)degree + 1) * (4 * (uint64_t)degree + 1);
    const int g = shape.cells, planes = shape.planes;
    std::vector<FitTask> tasks(nCells);
    const uint64_t side = 1ull << depth;
    const float h = 1.0f / (float)side;
    for (uint64_t i = 0; i < nCells; ++i) {
        const uint64_t c = i % (side * side * side);
        const uint64_t ix = c % side, iy = (c / side) % side, iz = c / (side * side);
        FitTask& t = tasks[i];
        std::memset(&t, 0, sizeof t);
        t.bmin[0] = -0.5f + (float)ix * h, t.bmin[1] = -0.5f + (float)iy * h, t.bmin[2] = -0.5f + (float)iz * h;
        for (int a = 0; a < 3; ++a) t.bmax[a] = t.bmin[a] + h;
        t.outOff = i * nc;
        t.copyOff = ~0ull;
        t.sampleOff = i * nq3;
        t.errSlot = (uint32_t)i;
        t.depth = (uint8_t)depth;
    }
    std::vector<FitBlock> blocks;
    for (uint64_t i = 0; i < nCells; i += g) {
        FitBlock fb;
        std::memset(&fb, 0, sizeof fb);
        fb.firstTask = (uint32_t)i;
        fb.nTasks = (uint16_t)std::min<uint64_t>(g, nCells - i);
        fb.degree = (uint8_t)degree;
        fb.planesPerChunk = (uint8_t)planes;
        fb.depth = (uint8_t)depth;
        fb.rowStart = (uint16_t)(split ? nc - rowsTop : 0);
        fb.rowEnd = (uint16_t)nc;
        fb.split = split ? 1 : 0;
        blocks.push_back(fb);
    }
    FitTask* dT = nullptr;
    FitBlock* dB = nullptr;
    double *dA = nullptr, *dE = nullptr, *dS = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = HPSDF_OK;
    hipError_t e = hipMalloc((void**)&dT, tasks.size() * sizeof(FitTask));
    if (e == hipSuccess) e = hipMalloc((void**)&dB, blocks.size() * sizeof(FitBlock));
    if (e == hipSuccess) e = hipMalloc((void**)&dA, nCells * nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&dE, nCells * sizeof(double));
    if (e == hipSuccess && split) e = hipMalloc((void**)&dS, nCells * nq3 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(dT, tasks.data(), tasks.size() * sizeof(FitTask), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dB, blocks.data(), blocks.size() * sizeof(FitBlock), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    FieldDev fd;
    RootMap rm;
    if (e == hipSuccess) rc = makeFieldDev(ctx, field, nullptr, &fd);
    for (int a = 0; a < 3; ++a) {
        rm.bounds[a] = (double)(cfg->root_max[a] - cfg->root_min[a]);
        rm.centre[a] = (double)((cfg->root_min[a] + cfg->root_max[a]) / 2.0f);
    }
    const size_t lds = shape.ldsBytes;
    if (split) fd.samples = dS;
    if (e == hipSuccess && rc == HPSDF_OK) {
        auto launch = [&]() {
            if (fast) return launchFitMfma(ctx->stream, degree, dB, (uint32_t)blocks.size(), dT, dA, dE, ctx->dTables, fd, rm);
            hipError_t le = launchFit(ctx->stream, degree, shape.cellsPerThread, dB, (uint32_t)blocks.size(), lds, dT, dA, dE, nullptr, ctx->dTables, fd, rm);
            if (le == hipSuccess && split)
                le = launchFitMfmaLow(ctx->stream, degree, dT, nullptr, 0u, (uint32_t)nCells, 0u, dA, ctx->dTables, dS, rm, fd.leftAssoc);
            return le;
        };
        e = launch();  // warm-up
        if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
        for (int r = 0; r < repeats && e == hipSuccess; ++r) e = launch();
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        *msPerLaunch = (double)ms / repeats;
        if (e == hipSuccess && coeffsOut) e = hipMemcpy(coeffsOut, dA, nCells * nc * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess && errsOut) e = hipMemcpy(errsOut, dE, nCells * sizeof(double), hipMemcpyDeviceToHost);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(dT);
    (void)hipFree(dB);
    (void)hipFree(dA);
    (void)hipFree(dE);
    if (dS) (void)hipFree(dS);
    if (rc) return rc;
    if (e != hipSuccess) return hipFail(e, "bench_fit");
    return HPSDF_OK;
    }
}

}  // extern "C"
