// Host runtime objects behind the opaque handles of include/hpsdf.h.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "device_types.hpp"
#include "tables.hpp"

namespace hpsdf {

// A device buffer with a pinned host twin (async copies need no host synchronisation).
template <typename T>
struct Staged {
    T* dev = nullptr;
    T* host = nullptr;
    uint64_t cap = 0;
    hipError_t ensure(uint64_t n) {
        if (n <= cap) return hipSuccess;
        uint64_t nc = cap ? cap : 1024;
        while (nc < n) nc *= 2;
        release();
        hipError_t e = hipMalloc((void**)&dev, nc * sizeof(T));
        if (e == hipSuccess) e = hipHostMalloc((void**)&host, nc * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = nc;
        return e;
    }
    void release() {
        if (dev) (void)hipFree(dev);
        if (host) (void)hipHostFree(host);
        dev = host = nullptr;
        cap = 0;
    }
};

// Per-context scratch of the build: the coefficient arena and the per-round staging buffers.  Kept
// across Create() calls (hipMalloc/hipFree cost more than a whole coarse round).
struct Workspace {
    int device = -1;
    bool inUse = false;
    double* arena = nullptr;
    uint64_t arenaCap = 0;
    Staged<FitTask> tasks;
    Staged<FitBlock> blocks;
    Staged<double> errs;
    Staged<double> samples;
    double* meshSamples = nullptr;  // mesh fields: the round's F values (device only)
    uint64_t meshSamplesCap = 0;
    Staged<double> pack;
    Staged<PackItem> items;
    void release() {
        if (device >= 0) (void)hipSetDevice(device);
        if (arena) (void)hipFree(arena);
        arena = nullptr;
        arenaCap = 0;
        tasks.release();
        blocks.release();
        errs.release();
        samples.release();
        if (meshSamples) (void)hipFree(meshSamples);
        meshSamples = nullptr;
        meshSamplesCap = 0;
        pack.release();
        items.release();
    }
};

}  // namespace hpsdf

struct hpsdf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    hpsdf::DeviceTables* dTables = nullptr;
    // hpsdf_ctx_set_fit_mode: how from-scratch fits of higher degree run (split: from splitMinDegree, 6 by default; fast: from 4).  HPSDF_FIT_SPLIT (default): rows of top degree bit-exact, the
    // rows below them on the matrix cores (errors and so every decision canonical); HPSDF_FIT_EXACT: every row bit-exact (the
    // canonical bytes); HPSDF_FIT_FAST: every row of every fit of degree >= 4 on the matrix cores (errors within ~1e-15, ties may flip)
    int fitMode = HPSDF_FIT_SPLIT;
    int splitMinDegree = 6;  // HPSDF_FIT_SPLIT: from-scratch fits of this degree and above are split (hpsdf_ctx_set_split_min_degree)
    // hpsdf_ctx_set_reduction_order / hpsdf_ctx_set_mesh_face_rule: -1 = this context follows the process-wide setting
    // (hpsdf_set_reduction_order / hpsdf_set_mesh_face_rule, which stand for the reference's build flags), 0 / 1 = its own.  Two
    // Octrees of one process can differ in them; read when a launch is prepared, like the process-wide ones.
    int reductionOrder = -1;
    int meshFaceRule = -1;
    // hpsdf_ctx_set_build_limits: a Create whose tree outgrows either bound when a round opens is refused with
    // HPSDF_ERR_BUILD_LIMIT (0 = the default: a share of the device's free memory; UINT64_MAX = none)
    uint64_t limitNodes = 0, limitBytes = 0;
    hpsdf::Workspace ws;
    // Scratch of the *_host entry points (host arrays in, host arrays out): one device buffer and one pinned buffer,
    // kept across calls -- a scalar Query(pt) through the C++ drop-in must not pay two hipMalloc/hipFree pairs.
    // hostLock serialises those entry points per context (Octree::Query* is const and callable from many threads in
    // the reference, Octree.h:71-78).
    std::shared_ptr<void> frontierScratch;    // frontier.hip: the device-side frontier's buffers, kept between Creates
    std::shared_ptr<void> continuityScratch;  // continuity.cpp: matrix + solver vectors kept between post-processes
    char* hostDev = nullptr;
    size_t hostDevCap = 0;
    char* hostPin = nullptr;
    char* hostPinDev = nullptr;  // the pinned buffer as the device sees it (tiny calls run on it directly)
    size_t hostPinCap = 0;
    std::mutex hostLock;
    std::mutex scratchLock;  // growth of dDefer / dDeferCount below (the *_device query entry points)
    // Query scratch for trees with leaves of degree > 3: per-workgroup lists of the points finished lane by lane
    uint32_t* dDefer = nullptr;
    uint64_t deferCap = 0;
    uint32_t* dDeferCount = nullptr;
    // hpsdf_ctx_set_block_allocator: where the memory blocks Create hands out come from (default: malloc / free)
    void* (*blockAlloc)(size_t, void*) = nullptr;
    void (*blockRelease)(void*, void*) = nullptr;
    void* blockUser = nullptr;
    void* allocBlock(size_t bytes) const { return blockAlloc ? blockAlloc(bytes, blockUser) : std::malloc(bytes); }
    void freeBlock(void* p) const {  // a block the build began and will not return
        if (p == nullptr) return;
        if (blockAlloc) {
            if (blockRelease) blockRelease(p, blockUser);
        } else {
            std::free(p);
        }
    }
};

struct hpsdf_tree {
    int device = 0;
    hpsdf::NodeRec* dNodes = nullptr;
    hpsdf::TopEntry* dTop = nullptr;
    hpsdf::NodeRec* dTopRec = nullptr;
    double* dCoeffs = nullptr;
    hpsdf::TreeDev dev{};
    uint64_t nNodes = 0, nCoeffs = 0, nLeaves = 0;
    int maxDegree = 0, maxDepth = 0;
    bool allInline = false;  // every leaf sits in the top table with degree <= 2: query_kernel serves it
    hpsdf_config config{};
    // the block's node array and packed coefficients as uploaded: calls of a few points are answered from them on the calling
    // thread (host_query.cpp), with the kernels' statements in the kernels' order
    // What calls of a few points are answered from on the calling thread (host_query.cpp): the device mirror's own arrays -- 8-byte node
    // records, coefficients with every leaf on its own lines -- fetched by the FIRST such call (hostCopies()), not kept by every
    // tree from its upload on.
    mutable std::mutex hostLock;
    mutable std::vector<hpsdf::NodeRec> hRecs;
    mutable std::vector<double> hPadded;
    mutable std::atomic<bool> hostReady{false};
    uint64_t paddedCount = 0;
    int hostCopies() const;  // HPSDF_OK, or the status of the failed download
};

enum HostFieldKind { kHostAnalytic = 0, kHostCallback = 1, kHostMesh = 2, kHostTreeCsg = 3 };

struct hpsdf_field {
    int kind = kHostAnalytic;
    std::vector<hpsdf_prim> prims;
    hpsdf_callback cb = nullptr;
    void* user = nullptr;
    // mesh (device resident)
    int device = -1;
    void* dBlock = nullptr;  // device-built meshes: ONE allocation holding the five arrays below (which then are not freed singly)
    float* dVerts = nullptr;
    uint32_t* dTris = nullptr;
    float* dTriPos = nullptr;
    float* dTriPre = nullptr;
    uint32_t* dHalfEdges = nullptr;
    hpsdf::BvhNode* dBvh = nullptr;
    hpsdf::NodeSlab* dSlabs = nullptr;  // device-built meshes only (part of dBlock)
    uint32_t leafLog2 = 0;              // leaves hold at most 1 << leafLog2 triangles
    uint32_t nTris = 0, nVerts = 0, nBvhNodes = 0;
    unsigned long long* dStats = nullptr;  // 4 counters, only under HPSDF_MESH_STATS=1
    // host copies of the arrays above, made by the first call of one or two points (hpsdf_field_eval_host, n <= kHostMeshPoints): such calls
    // are answered on the calling thread (kernels.hip, meshEvalHostPoints) -- what Mesh::SignedDistanceAtPt(pt, bvh) inside a user's
    // SDF lambda costs decides whether code written against the reference is usable as it is
    struct HostMirror {
        std::vector<float> verts, triPos, triPre;
        std::vector<uint32_t> tris, halfEdges;
        std::vector<hpsdf::BvhNode> bvh;
        hpsdf::MeshDev dev{};
    };
    mutable std::shared_ptr<HostMirror> hostMirror;
    mutable std::mutex hostMirrorLock;
    // csg wrapper
    const hpsdf_tree* oldTree = nullptr;
    int csgOp = -1;
    const hpsdf_field* inner = nullptr;
};

namespace hpsdf {

void setError(const std::string& msg);
int fail(int code, const std::string& msg);
int hipFail(hipError_t e, const char* what);

#define HPSDF_HIP(call)                                           \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return ::hpsdf::hipFail(e_, #call); \
    } while (0)

// Query / QueryWithGradient of one point on the calling thread (host_query.cpp): the kernels' values bit for bit
constexpr size_t kHostQueryPoints = 32;     // Query calls of up to this many points never reach the device (capi.cpp: hostQueryLimit)
constexpr size_t kHostGradientPoints = 32;  // QueryWithGradient
constexpr size_t kHostRays = 32;            // QueryRay
constexpr size_t kHostMeshPoints = 2;       // mesh signed distance (hpsdf_field_eval_host on a plain mesh field): ~23-50 us a point on the host, a launch
                                            // round trip ~55 us + ~3 us a point -- from three points on the device is the faster one
double hostQueryPoint(const hpsdf_tree& t, const double* xyz);
void hostQueryPointWithGradient(const hpsdf_tree& t, const double* xyz, double* out, double* grad, int leftAssoc);
bool hostQueryRay(const hpsdf_tree& t, const double* origin, const double* dir, double tMax, double* tOut);

// innermost non-CSG field and the FieldDev the kernels take
const hpsdf_field* innermost(const hpsdf_field* f);
#ifdef HPSDF_TEST_HOOKS
constexpr const char* kInjectedFailureMsg = "injected failure (HPSDF_TEST_FAIL_RANK)";
#else
constexpr const char* kInjectedFailureMsg = "injected failure";  // (unreachable: the hook is not compiled in)
#endif
int makeFieldDev(const hpsdf_ctx* ctx, const hpsdf_field* f, const double* dSamples, FieldDev* out);
// hpsdf_ctx_set_build_limits, checked by both schedulers when a round opens.  nodes: the tree's; bytes: the device memory this rank's
// build state needs for the round about to open -- node arrays, coefficient arena, a mesh build's sample buffer; NOT the hand-over buffer
// of split fits (bounded by 2^31 samples whatever the tree's size: it is not what runs away);
// held: what the build's buffers hold now (counted as available when the default limit is
// measured); *measured: the build's cache of that measurement (0 = not taken yet; one hipMemGetInfo per Create, and only once a build
// needs more than 256 MiB).  HPSDF_OK, or HPSDF_ERR_BUILD_LIMIT with the message set.
uint64_t buildByteLimit(const hpsdf_ctx* ctx, uint64_t bytes, uint64_t held, uint64_t* measured);  // the bound on bytes in effect (UINT64_MAX: none)
int checkBuildLimits(const hpsdf_ctx* ctx, uint64_t nodes, uint64_t bytes, uint64_t growBytes, uint64_t held, uint64_t* measured, uint64_t rounds, double total, double target);
// the two semantic switches as a context sees them: its own setting, or the process-wide one when it has none (ctx may be null)
int meshFaceRuleReference(const hpsdf_ctx* ctx);  // hpsdf_[ctx_]set_mesh_face_rule(): 1 = the reference's face-case point whatever its weights
float meshFaceTolOfSlack(const hpsdf_ctx* ctx);   // MeshDev::faceTolOfSlack for launches prepared now
int reductionLeftAssoc(const hpsdf_ctx* ctx);     // hpsdf_[ctx_]set_reduction_order()
void setReductionLeftAssoc(int left);

// host mesh preparation (mesh.cpp)
// std::allocator that leaves trivially constructible elements uninitialised on resize(): the big mesh arrays are
// written in full by parallel workers right after they are sized, and zero-filling them first costs more than that
template <typename T>
struct DefaultInit : std::allocator<T> {
    template <typename U>
    struct rebind {
        using other = DefaultInit<U>;
    };
    template <typename U, typename... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0)
            ::new ((void*)p) U;
        else
            ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <typename T>
using RawVector = std::vector<T, DefaultInit<T>>;

struct HostMesh {
    std::vector<float> verts;
    std::vector<uint32_t> tris;
    std::vector<uint32_t> halfEdges;
    RawVector<BvhNode> bvh;
};
// returns false when the mesh is not closed (Mesh::CreateHalfEdges, Mesh.cpp:87-131)
bool prepareMesh(const float* verts, uint64_t nVerts, const uint64_t* tris, uint64_t nTris, HostMesh* out);
// the twins alone (non-manifold meshes whose BVH, slabs and records the device has built)
bool hostHalfEdges(const uint64_t* tris, uint64_t nTris, uint64_t nVerts, std::vector<uint32_t>* he);
// the same on the device (mesh_build.hip): fills f's device pointers; *fallback = 1 asks for prepareMesh instead, 2 for hostHalfEdges
// (everything but the twins is in place: f->dHalfEdges waits for them)
int meshBuildDevice(hpsdf_ctx* ctx, const float* verts, uint64_t nVerts, const uint64_t* tris, uint64_t nTris, hpsdf_field* f, int* fallback);

}  // namespace hpsdf
