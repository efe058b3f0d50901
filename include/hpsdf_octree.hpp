// SDF::Octree / SDF::Config / MemoryBlock -- C++ drop-in over the C ABI of hpsdf.h.
//
// Mirrors the public surface of the reference class (Include/HP/Octree.h:37-86,
// Include/HP/Config.h:12-43, Include/Utility/MemoryBlock.h:5-9) for the hot path:
// Create, Query, ToMemoryBlock, FromMemoryBlock, Clear, copy/move, GetRootAABB and
// the three CSG rebuilds.  Everything numeric happens in libhpsdf.so on the GPU;
// this header only marshals.  It is header-only on purpose: the compiled library
// has no C++ types in its ABI, so it does not care which Eigen (if any) the host
// program uses.
//
// Differences from the reference, all additive or documented in DESIGN.md:
//   * Query(const double* xyz, n, out) -- the batched form: one pass of the GPU kernels over the
//     points.  A scalar Query(pt) / QueryWithGradient(pt, n) -- any call of up to 32 points -- is
//     answered on the calling thread from the tree handle's copy of the block (~0.1 us, the
//     kernels' values bit for bit: csrc/host_query.cpp), so loops over Query(pt) (the reference's
//     own tests and benchmarks: HPUnitTests.cpp:64-75, HPBenchmarks.cpp:105-109) run as they are.
//   * Create(config, DeviceField) -- fields the GPU evaluates itself (analytic
//     primitives, triangle meshes); Create(config, std::function) still works and
//     samples the callback with config.threadCount host threads per round.
//   * errors: the reference asserts; here a failed call throws SDF::Error carrying
//     hpsdf_last_error() (Query outside the root still returns DBL_MAX).
//   * QueryRay(origins, dirs, tMax, n, hit, t) -- batched sphere tracing on the GPU; QueryRay(ray, tMax, t)
//     is the same path with n = 1.  OutputFunctionSlice needs no stb: the BMP is written here.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <atomic>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hpsdf.h"

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && __has_include(<Eigen/Geometry>) && !defined(HPSDF_NO_EIGEN)
#define HPSDF_HAVE_EIGEN 1
#include <Eigen/Core>
#include <Eigen/Geometry>
#endif
#endif

// Literals.h of the reference: on LP64 Linux "u32" is 8 bytes wide
typedef double f64;
typedef float f32;
typedef int i16;
typedef long int i32;
typedef unsigned char u8;
typedef unsigned short u16;
typedef long unsigned int u32;
typedef long long unsigned int u64;
typedef size_t usize;
constexpr usize PATH_MAX_LEN = 1024;      // Literals.h:12
constexpr f32 EPSILON_F32 = 0.000001f;    // Literals.h:13

#ifndef HPSDF_HAVE_EIGEN
// Just enough of the two Eigen types the public API mentions, for hosts without Eigen.
namespace Eigen {
template <typename T>
struct HpsdfVec3 {
    T v[3];
    HpsdfVec3() : v{T(0), T(0), T(0)} {}
    HpsdfVec3(T x, T y, T z) : v{x, y, z} {}
    T& x() { return v[0]; }
    T& y() { return v[1]; }
    T& z() { return v[2]; }
    const T& x() const { return v[0]; }
    const T& y() const { return v[1]; }
    const T& z() const { return v[2]; }
    T& operator()(int i) { return v[i]; }
    const T& operator()(int i) const { return v[i]; }
    const T* data() const { return v; }
    T* data() { return v; }
    HpsdfVec3 operator+(const HpsdfVec3& o) const { return {T(v[0] + o.v[0]), T(v[1] + o.v[1]), T(v[2] + o.v[2])}; }
    HpsdfVec3 operator-(const HpsdfVec3& o) const { return {T(v[0] - o.v[0]), T(v[1] - o.v[1]), T(v[2] - o.v[2])}; }
    HpsdfVec3 operator*(T s) const { return {T(v[0] * s), T(v[1] * s), T(v[2] * s)}; }
    HpsdfVec3 operator/(T s) const { return {T(v[0] / s), T(v[1] / s), T(v[2] / s)}; }
    HpsdfVec3 cwiseProduct(const HpsdfVec3& o) const { return {T(v[0] * o.v[0]), T(v[1] * o.v[1]), T(v[2] * o.v[2])}; }
    // Eigen's reduction of three elements: a . (b . c) by default; a double vector follows hpsdf_set_reduction_order() (include/hpsdf.h:
    // a vectorised Eigen reduces a Vector3d as (a . b) . c), a float vector is a . (b . c) in every build
    static T sum3(T a, T b, T c) { return (sizeof(T) == 8 && hpsdf_get_reduction_order()) ? (a + b) + c : a + (b + c); }
    T dot(const HpsdfVec3& o) const { return sum3(v[0] * o.v[0], v[1] * o.v[1], v[2] * o.v[2]); }
    T squaredNorm() const { return sum3(v[0] * v[0], v[1] * v[1], v[2] * v[2]); }
    T norm() const { return std::sqrt(squaredNorm()); }
};
typedef HpsdfVec3<double> Vector3d;
typedef HpsdfVec3<float> Vector3f;
typedef HpsdfVec3<int> Vector3i;
template <typename T>
struct HpsdfBox3 {
    HpsdfVec3<T> lo, hi;
    HpsdfBox3()
        : lo(std::numeric_limits<T>::max(), std::numeric_limits<T>::max(), std::numeric_limits<T>::max()),
          hi(std::numeric_limits<T>::lowest(), std::numeric_limits<T>::lowest(), std::numeric_limits<T>::lowest()) {}
    HpsdfBox3(const HpsdfVec3<T>& a, const HpsdfVec3<T>& b) : lo(a), hi(b) {}
    HpsdfVec3<T>& min() { return lo; }
    HpsdfVec3<T>& max() { return hi; }
    const HpsdfVec3<T>& min() const { return lo; }
    const HpsdfVec3<T>& max() const { return hi; }
    HpsdfVec3<T> sizes() const { return hi - lo; }
    HpsdfVec3<T> center() const { return (lo + hi) / T(2); }
    bool contains(const HpsdfVec3<T>& p) const {  // both ends inclusive, as Eigen's
        return lo.x() <= p.x() && p.x() <= hi.x() && lo.y() <= p.y() && p.y() <= hi.y() && lo.z() <= p.z() && p.z() <= hi.z();
    }
    /// uniform point of the box through std::rand, as Eigen's AlignedBox::sample() (internal::random<T>(0, 1))
    HpsdfVec3<T> sample() const {
        HpsdfVec3<T> r;
        for (int d = 0; d < 3; ++d) r(d) = lo(d) + (hi(d) - lo(d)) * (T(0) + (T(1) - T(0)) * T(std::rand()) / T(RAND_MAX));
        return r;
    }
    T volume() const { return (hi.x() - lo.x()) * (hi.y() - lo.y()) * (hi.z() - lo.z()); }
};
typedef HpsdfBox3<float> AlignedBox3f;
typedef HpsdfBox3<double> AlignedBox3d;
}  // namespace Eigen
#endif

struct MemoryBlock {  // Include/Utility/MemoryBlock.h:5-9 (global namespace there too)
    usize size;
    void* ptr;
};

namespace SDF {

constexpr usize BASIS_MAX_DEGREE = HPSDF_BASIS_MAX_DEGREE;
constexpr usize TREE_MAX_DEPTH = HPSDF_TREE_MAX_DEPTH;

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};
inline void check(int rc) {
    if (rc != HPSDF_OK) throw Error(rc, hpsdf_last_error());
}

// Same fields, names and defaults as the reference Config; converts to the 80-byte POD.
struct Config {
    struct NearnessWeighting {
        enum Type : u8 { None = 0, Polynomial = 1, Exponential = 2 } type;
        f64 strength;
    } nearnessWeighting;
    struct Continuity {
        bool enforce;
        f64 strength;
    } continuity;
    bool enableLogging;
    f64 targetErrorThreshold;
    u32 threadCount;
    Eigen::AlignedBox3f root;

    Config() {  // Source/HP/Config.cpp:5-14
        hpsdf_config d;
        hpsdf_config_default(&d);
        fromPod(d);
    }
    void IsValid() const {  // Source/HP/Config.cpp:17-32 (asserts there)
        if (!(targetErrorThreshold > 0.0)) throw Error(HPSDF_ERR_INVALID_ARGUMENT, "targetErrorThreshold must be > 0");
        if (threadCount == 0) throw Error(HPSDF_ERR_INVALID_ARGUMENT, "threadCount must be > 0");
        if (!(root.volume() > 0.0f)) throw Error(HPSDF_ERR_INVALID_ARGUMENT, "root volume must be > 0");
        if (nearnessWeighting.type != NearnessWeighting::None && !(nearnessWeighting.strength > 0.0))
            throw Error(HPSDF_ERR_INVALID_ARGUMENT, "nearnessWeighting.strength must be > 0");
        if (continuity.enforce && !(continuity.strength > 0.0))
            throw Error(HPSDF_ERR_INVALID_ARGUMENT, "continuity.strength must be > 0");
    }
    hpsdf_config toPod() const {
        hpsdf_config d;
        std::memset(&d, 0, sizeof d);
        d.weighting_type = (uint8_t)nearnessWeighting.type;
        d.weighting_strength = nearnessWeighting.strength;
        d.continuity_enforce = continuity.enforce ? 1 : 0;
        d.continuity_strength = continuity.strength;
        d.enable_logging = enableLogging ? 1 : 0;
        d.target_error_threshold = targetErrorThreshold;
        d.thread_count = threadCount;
        for (int a = 0; a < 3; ++a) {
            d.root_min[a] = root.min()(a);
            d.root_max[a] = root.max()(a);
        }
        return d;
    }
    void fromPod(const hpsdf_config& d) {
        nearnessWeighting.type = (NearnessWeighting::Type)d.weighting_type;
        nearnessWeighting.strength = d.weighting_strength;
        continuity.enforce = d.continuity_enforce != 0;
        continuity.strength = d.continuity_strength;
        enableLogging = d.enable_logging != 0;
        targetErrorThreshold = d.target_error_threshold;
        threadCount = (u32)d.thread_count;
        root = Eigen::AlignedBox3f(Eigen::Vector3f(d.root_min[0], d.root_min[1], d.root_min[2]),
                                   Eigen::Vector3f(d.root_max[0], d.root_max[1], d.root_max[2]));
    }
};

/// Defines a line R(t) = O + t * D, where ||D||_2 = 1   (Include/HP/Ray.h:8-22, Source/HP/Ray.cpp:5-15)
struct Ray {
    Ray(const Eigen::Vector3d& origin_, const Eigen::Vector3d& direction_) : origin(origin_), direction(direction_) {
        for (int a = 0; a < 3; ++a) {
            invDirection(a) = 1.0 / direction_(a);
            sign(a) = invDirection(a) < 0.0;
        }
    }
    Eigen::Vector3d origin;
    Eigen::Vector3d direction;
    Eigen::Vector3d invDirection;
    Eigen::Vector3i sign;

    /// Returns whether the ray intersects the box and, if so, stores the slab parameters in a_ and b_
    /// (Source/HP/Ray.cpp:18-68; host-side helper, the batched QueryRay does the same on the GPU)
    bool IntersectAABB(const Eigen::AlignedBox3d& aabb_, Eigen::Vector3d& a_, Eigen::Vector3d& b_) const {
        const Eigen::Vector3d bounds[2] = {aabb_.min(), aabb_.max()};
        a_(0) = (bounds[sign(0)](0) - origin(0)) * invDirection(0);
        b_(0) = (bounds[1 - sign(0)](0) - origin(0)) * invDirection(0);
        a_(1) = (bounds[sign(1)](1) - origin(1)) * invDirection(1);
        b_(1) = (bounds[1 - sign(1)](1) - origin(1)) * invDirection(1);
        if ((a_(0) > b_(1)) || (a_(1) > b_(0))) return false;
        if (a_(1) > a_(0)) a_(0) = a_(1);
        if (b_(1) < b_(0)) b_(0) = b_(1);
        a_(2) = (bounds[sign(2)](2) - origin(2)) * invDirection(2);
        b_(2) = (bounds[1 - sign(2)](2) - origin(2)) * invDirection(2);
        if ((a_(0) > b_(2)) || (a_(2) > b_(0))) return false;
        if (a_(2) > a_(0)) a_(0) = a_(2);
        if (b_(2) < b_(0)) b_(0) = b_(2);
        return true;
    }
};

// A field the GPU evaluates itself (no host callback in the loop).
class DeviceField {
   public:
    DeviceField() = default;
    DeviceField(const DeviceField&) = delete;
    DeviceField& operator=(const DeviceField&) = delete;
    DeviceField(DeviceField&& o) noexcept : f_(o.f_) { o.f_ = nullptr; }
    ~DeviceField() { hpsdf_field_destroy(f_); }
    static DeviceField Analytic(const std::vector<hpsdf_prim>& prims) {
        DeviceField d;
        check(hpsdf_field_create_analytic(prims.data(), (int)prims.size(), &d.f_));
        return d;
    }
    static DeviceField Sphere(double cx, double cy, double cz, double r) {
        hpsdf_prim p;
        std::memset(&p, 0, sizeof p);
        p.kind = HPSDF_PRIM_SPHERE;
        p.p[0] = cx, p.p[1] = cy, p.p[2] = cz, p.p[3] = r;
        return Analytic({p});
    }
    /// BASELINE configs[1]: union of a sphere, a box and a torus
    static DeviceField Union3() {
        hpsdf_prim p[3];
        std::memset(p, 0, sizeof p);
        p[0].kind = HPSDF_PRIM_SPHERE, p[0].op = HPSDF_OP_UNION;
        p[0].p[0] = -0.2, p[0].p[1] = -0.15, p[0].p[2] = 0.1, p[0].p[3] = 0.18;
        p[1].kind = HPSDF_PRIM_BOX, p[1].op = HPSDF_OP_UNION;
        p[1].p[0] = 0.15, p[1].p[1] = 0.2, p[1].p[2] = -0.1, p[1].p[3] = 0.12, p[1].p[4] = 0.10, p[1].p[5] = 0.15;
        p[2].kind = HPSDF_PRIM_TORUS_Y, p[2].op = HPSDF_OP_UNION;
        p[2].p[0] = 0.0, p[2].p[1] = -0.2, p[2].p[2] = -0.2, p[2].p[3] = 0.15, p[2].p[4] = 0.05;
        return Analytic({p[0], p[1], p[2]});
    }
    const hpsdf_field* handle() const { return f_; }

   private:
    friend class Octree;
    hpsdf_field* f_ = nullptr;
};

class Octree {
   public:
    typedef std::function<f64(const Eigen::Vector3d& pt_, const u32 threadIdx_)> Func;

    Octree() = default;
    ~Octree() { release(); }

    Octree(const Octree& o) { copyFrom(o); }  // deep copy, Octree.cpp:24-45
    Octree& operator=(const Octree& o) {
        if (this != &o) {
            release();
            copyFrom(o);
        }
        return *this;
    }
    Octree(Octree&& o) noexcept { steal(o); }  // Octree.cpp:76-86
    Octree& operator=(Octree&& o) noexcept {
        if (this != &o) {
            release();
            steal(o);
        }
        return *this;
    }

    /// Which GPU / stream later calls use (default: device 0, library-owned stream).
    void SetDevice(int device, void* hipStream = nullptr) {
        if (ctx_) {
            dropTree();
            hpsdf_ctx_destroy(ctx_);
            ctx_ = nullptr;
        }
        device_ = device;
        stream_ = hipStream;
    }
    /// Jobs per round of the canonical schedule (K); part of the result's definition.
    void SetJobsPerRound(uint64_t k) { jobsPerRound_ = k; }
    /// Additive: where fits may leave the reference's term-by-term summation (hpsdf_ctx_set_fit_mode).  Default HPSDF_FIT_SPLIT: errors,
    /// decisions and topology are those of the bit-exact path by construction; the rows below the top degree of a from-scratch fit of
    /// degree >= 6 (sum-factorised from the same samples) agree with it to ~1e-17.  HPSDF_FIT_EXACT: every row bit-exact;
    /// HPSDF_FIT_FAST (= SetFastFit(true)): every row of every fit of degree >= 4 on the matrix cores.
    void SetFitMode(int mode) {
        ensureCtx();
        check(hpsdf_ctx_set_fit_mode(ctx_, mode));
        fitMode_ = mode;
    }
    void SetFastFit(bool on) { SetFitMode(on ? HPSDF_FIT_FAST : HPSDF_FIT_SPLIT); }
    /// Additive: bounds on Create (hpsdf_ctx_set_build_limits).  The reference's build has none: a threshold below what the error
    /// estimate reaches on a field -- the default Config()'s 1e-10 on most -- refines until memory ends.  0 = the default (nodes:
    /// unbounded; bytes of nodes and coefficients: 1/64 of the free device memory, at least 1 GiB), UINT64_MAX = none.  A Create that crosses a limit throws SDF::Error
    /// with status HPSDF_ERR_BUILD_LIMIT and a message that names rounds, nodes, bytes and the error reached.
    void SetBuildLimits(uint64_t maxNodes, uint64_t maxBytes) {
        ensureCtx();
        check(hpsdf_ctx_set_build_limits(ctx_, maxNodes, maxBytes));
        limitNodes_ = maxNodes, limitBytes_ = maxBytes;
    }
    /// Additive, this object only (hpsdf_ctx_set_reduction_order / hpsdf_ctx_set_mesh_face_rule): -1 = follow the process-wide
    /// setting, 0 / 1 = this Octree's own.  Two Octrees of one process can differ.
    void SetReductionOrderHere(int leftAssoc) {
        ensureCtx();
        check(hpsdf_ctx_set_reduction_order(ctx_, leftAssoc));
        reductionOrder_ = leftAssoc;
    }
    void SetMeshFaceRuleHere(int reference) {
        ensureCtx();
        check(hpsdf_ctx_set_mesh_face_rule(ctx_, reference));
        meshFaceRule_ = reference;
    }
    /// Additive, process-wide (hpsdf_set_reduction_order): Eigen's Vector3d prod() / norm() / normalize() as (a . b) . c -- what an SSE2
    /// build of Eigen computes, by our reading -- instead of a . (b . c); set it once, before any Create / Query, to match your build.
    static void SetReductionOrder(bool leftAssoc) { hpsdf_set_reduction_order(leftAssoc ? 1 : 0); }
    /// Additive: Create() over `world` GPUs of one node -- one Octree per GPU (SetDevice), every rank calls Create with
    /// the same config and field; `gather` is the in-place all-gather of hpsdf_create_distributed (for RCCL:
    /// hpsdf_rccl::AllGather with an hpsdf_rccl::Comm as `user`, include/hpsdf_rccl.hpp).  Every rank ends with the
    /// identical tree.  Works for every field and config: device-evaluated fields (weighted configs included) take the
    /// device-side frontier on up to 8 ranks, std::function fields the host scheduler, sharded over the same gather.
    void SetRanks(int rank, int world, hpsdf_allgather_fn gather, void* user) {
        rank_ = rank, world_ = world, gather_ = gather, gatherUser_ = user;
    }

    /// Approximates F_ using the parameters in config_   (Octree.h:50)
    void Create(const Config& config_, Func F_) {
        hpsdf_field* f = nullptr;
        Func fn = std::move(F_);
        check(hpsdf_field_create_callback(&Octree::trampoline, &fn, &f));
        FieldGuard g{f};
        createFrom(config_, f);
    }
    /// Same, with a field evaluated on the GPU.
    void Create(const Config& config_, const DeviceField& F_) { createFrom(config_, F_.f_); }
    /// Same, with a field handle of the C ABI (e.g. Meshing::BVH::Field()); the handle stays the caller's
    void Create(const Config& config_, const hpsdf_field* F_) {
        if (!F_) throw Error(HPSDF_ERR_INVALID_ARGUMENT, "null field");
        createFrom(config_, F_);
    }

    /// Resultant SDF = Min(oldF, F_)   (Octree.h:53, Octree.cpp:355-374)
    void UnionSDF(Func F_) { csg(HPSDF_OP_UNION, std::move(F_)); }
    /// Resultant SDF = Max(-oldF, F_)  (Octree.h:56, Octree.cpp:377-387)
    void SubtractSDF(Func F_) { csg(HPSDF_OP_SUBTRACT, std::move(F_)); }
    /// Resultant SDF = Max(oldF, F_)   (Octree.h:59, Octree.cpp:390-400)
    void IntersectSDF(Func F_) { csg(HPSDF_OP_INTERSECT, std::move(F_)); }
    void UnionSDF(const DeviceField& F_) { csg(HPSDF_OP_UNION, F_.f_); }
    void SubtractSDF(const DeviceField& F_) { csg(HPSDF_OP_SUBTRACT, F_.f_); }
    void IntersectSDF(const DeviceField& F_) { csg(HPSDF_OP_INTERSECT, F_.f_); }

    /// Resets the tree   (Octree.h:62)
    void Clear() {
        dropTree();
        std::free(block_);
        block_ = nullptr;
        size_ = 0;
    }

    /// Creates an octree from a previously serialised version; the block stays the caller's   (Octree.h:65)
    void FromMemoryBlock(MemoryBlock octBlock_) {
        if (!octBlock_.size || !octBlock_.ptr) throw Error(HPSDF_ERR_BAD_BLOCK, "empty MemoryBlock");
        if (octBlock_.size < 16 + sizeof(hpsdf_config))  // two counts + Config: nothing smaller can be a block
            throw Error(HPSDF_ERR_BAD_BLOCK, "MemoryBlock too small");
        Clear();
        block_ = std::malloc(octBlock_.size);
        if (!block_) throw Error(HPSDF_ERR_OUT_OF_MEMORY, "malloc failed");
        std::memcpy(block_, octBlock_.ptr, octBlock_.size);
        size_ = octBlock_.size;
        readConfig();
        uploadTree();
    }

    /// Serialises an octree to a memory block owned by malloc (caller frees)   (Octree.h:68)
    MemoryBlock ToMemoryBlock() const {
        MemoryBlock b = {0, nullptr};
        if (!block_) return b;
        b.ptr = std::malloc(size_);
        if (!b.ptr) throw Error(HPSDF_ERR_OUT_OF_MEMORY, "malloc failed");
        std::memcpy(b.ptr, block_, size_);
        b.size = size_;
        return b;
    }

    /// Returns the approximated distance from F = 0; DBL_MAX outside the root   (Octree.h:71)
    f64 Query(const Eigen::Vector3d& pt_) const {
        const double xyz[3] = {pt_(0), pt_(1), pt_(2)};
        double out = 0.0;
        Query(xyz, 1, &out);
        return out;
    }
    /// Batched Query over host arrays (xyz interleaved)
    void Query(const double* xyz, usize n, double* out) const {
        hpsdf_tree* t = deviceTree();
        if (!t) throw Error(HPSDF_ERR_STATE, "Query on an empty octree");
        check(hpsdf_query_host(ctx_, t, xyz, n, out));
    }
    /// Batched Query over device (HBM) arrays, asynchronous on the context stream
    void QueryDevice(const double* d_xyz, usize n, double* d_out) const {
        hpsdf_tree* t = deviceTree();
        if (!t) throw Error(HPSDF_ERR_STATE, "Query on an empty octree");
        check(hpsdf_query_device(ctx_, t, d_xyz, n, d_out));
    }

    /// As with Query, but with the unit "gradient" calculated via CD   (Octree.h:78, Octree.cpp:749-789)
    f64 QueryWithGradient(const Eigen::Vector3d& pt_, Eigen::Vector3d& unitNormal_) const {
        const double xyz[3] = {pt_(0), pt_(1), pt_(2)};
        double out = 0.0, g[3] = {unitNormal_(0), unitNormal_(1), unitNormal_(2)};
        QueryWithGradient(xyz, 1, &out, g);
        unitNormal_ = Eigen::Vector3d(g[0], g[1], g[2]);
        return out;
    }
    /// Batched form; rows of grad for points outside the root are left untouched
    void QueryWithGradient(const double* xyz, usize n, double* out, double* grad) const {
        hpsdf_tree* t = deviceTree();
        if (!t) throw Error(HPSDF_ERR_STATE, "Query on an empty octree");
        check(hpsdf_query_gradient_host(ctx_, t, xyz, n, out, grad));
    }

    /// Sphere tracing along ray_ (<= 200 Query steps).  As in the reference (Octree.cpp:705-746), t_ receives
    /// the field value at the stopping point on a hit and is left untouched otherwise   (Octree.h:75)
    bool QueryRay(const Ray& ray_, const f64 tMax_, f64& t_) const {
        const double o[3] = {ray_.origin(0), ray_.origin(1), ray_.origin(2)};
        const double d[3] = {ray_.direction(0), ray_.direction(1), ray_.direction(2)};
        uint8_t hit = 0;
        double t = t_;
        QueryRay(o, d, &tMax_, 1, &hit, &t);
        if (hit) t_ = t;
        return hit != 0;
    }
    /// Batched form over host arrays; t rows of misses are left untouched
    void QueryRay(const double* origins, const double* dirs, const double* tMax, usize n, uint8_t* hit, double* t) const {
        hpsdf_tree* tr = deviceTree();
        if (!tr) throw Error(HPSDF_ERR_STATE, "Query on an empty octree");
        check(hpsdf_query_ray_host(ctx_, tr, origins, dirs, tMax, n, hit, t));
    }

    /// Outputs an image of the z = c_ slice over viewArea_ to <fName_>.bmp   (Octree.h:83-86, Octree.cpp:1131-1206;
    /// the reference needs stb_image_write for this, here the 24-bit BMP is written directly)
    void OutputFunctionSlice(const char* fName_, const f64 c_, const Eigen::AlignedBox3f& viewArea_) const {
        hpsdf_tree* t = deviceTree();
        if (!t) throw Error(HPSDF_ERR_STATE, "Query on an empty octree");
        const uint64_t n = 2048;
        std::vector<uint8_t> rgb(n * n * 3);
        const float vmin[3] = {viewArea_.min()(0), viewArea_.min()(1), viewArea_.min()(2)};
        const float vmax[3] = {viewArea_.max()(0), viewArea_.max()(1), viewArea_.max()(2)};
        check(hpsdf_function_slice(ctx_, t, c_, vmin, vmax, n, rgb.data(), nullptr));
        const std::string path = std::string(fName_) + ".bmp";
        std::FILE* fh = std::fopen(path.c_str(), "wb");
        if (!fh) throw Error(HPSDF_ERR_STATE, "cannot open " + path);
        const uint32_t rowBytes = (uint32_t)(3 * n), pad = (4 - rowBytes % 4) % 4;
        const uint32_t fileSize = 14 + 40 + (rowBytes + pad) * (uint32_t)n;
        uint8_t hdr[54] = {0};
        auto put32 = [&](int off, uint32_t v) { std::memcpy(hdr + off, &v, 4); };
        auto put16 = [&](int off, uint16_t v) { std::memcpy(hdr + off, &v, 2); };
        hdr[0] = 'B', hdr[1] = 'M';
        put32(2, fileSize), put32(10, 54), put32(14, 40), put32(18, (uint32_t)n), put32(22, (uint32_t)n);
        put16(26, 1), put16(28, 24);
        std::fwrite(hdr, 1, sizeof hdr, fh);
        std::vector<uint8_t> row(rowBytes + pad, 0);
        for (uint64_t i = n; i-- > 0;) {  // bottom-up rows, BGR
            for (uint64_t j = 0; j < n; ++j) {
                const uint8_t* px = &rgb[3 * (i * n + j)];
                row[3 * j] = px[2], row[3 * j + 1] = px[1], row[3 * j + 2] = px[0];
            }
            std::fwrite(row.data(), 1, row.size(), fh);
        }
        std::fclose(fh);
    }

    /// Returns the aabb of the root node   (Octree.h:81)
    Eigen::AlignedBox3f GetRootAABB() const { return config_.root; }

    const hpsdf_build_stats& LastBuildStats() const { return stats_; }
    const hpsdf_continuity_stats& LastContinuityStats() const { return continuity_; }
    const Config& GetConfig() const { return config_; }

   private:
    struct FieldGuard {
        hpsdf_field* f;
        ~FieldGuard() { hpsdf_field_destroy(f); }
    };
    static double trampoline(const double* pt, uint64_t threadIdx, void* user) {
        const Func* fn = static_cast<const Func*>(user);
        return (*fn)(Eigen::Vector3d(pt[0], pt[1], pt[2]), (u32)threadIdx);
    }
    void ensureCtx() const {
        if (!ctx_) {
            // (this header hands the library structs by size: refuse a library built with another hpsdf.h -- HPSDF_ABI_VERSION)
            if (hpsdf_abi_version() != HPSDF_ABI_VERSION)
                throw Error(HPSDF_ERR_STATE, "libhpsdf.so has ABI version " + std::to_string(hpsdf_abi_version()) + ", this program was compiled against " +
                                                 std::to_string(HPSDF_ABI_VERSION) + " (include/hpsdf.h): recompile one of them");
            check(hpsdf_ctx_create(device_, stream_, &ctx_));
            if (fitMode_ >= 0) check(hpsdf_ctx_set_fit_mode(ctx_, fitMode_));  // (a copy keeps the mode its source was given)
            if (limitNodes_ || limitBytes_) check(hpsdf_ctx_set_build_limits(ctx_, limitNodes_, limitBytes_));
            if (reductionOrder_ >= 0) check(hpsdf_ctx_set_reduction_order(ctx_, reductionOrder_));
            if (meshFaceRule_ >= 0) check(hpsdf_ctx_set_mesh_face_rule(ctx_, meshFaceRule_));
        }
    }
    void dropTree() {
        hpsdf_tree_destroy(tree_);
        tree_ = nullptr;
        mirrorPending_.store(false, std::memory_order_release);
    }
    // The device mirror Query* and the CSG operations work on.  A tree this object has just built gets it on first use, not
    // inside Create (the reference's Create ends when the tree is built; the mirror -- tables and a line-aligned copy of the
    // coefficients, ~0.4 ms of host work -- is a cost of the first query here).  Query* are const and may be called from
    // several threads (Octree.h:71-78): the first caller builds it under a lock.
    hpsdf_tree* deviceTree() const {
        if (mirrorPending_.load(std::memory_order_acquire)) {
            std::lock_guard<std::mutex> guard(mirrorLock_);
            if (mirrorPending_.load(std::memory_order_relaxed)) {
                check(hpsdf_tree_upload(ctx_, block_, size_, &tree_));
                mirrorPending_.store(false, std::memory_order_release);
            }
        }
        return tree_;
    }
    void release() {
        Clear();
        hpsdf_ctx_destroy(ctx_);
        ctx_ = nullptr;
    }
    void readConfig() {
        hpsdf_config pod;
        std::memcpy(&pod, (const uint8_t*)block_ + size_ - sizeof pod, sizeof pod);
        config_.fromPod(pod);
    }
    void uploadTree() {
        ensureCtx();
        dropTree();
        check(hpsdf_tree_upload(ctx_, block_, size_, &tree_));
    }
    void createFrom(const Config& config, const hpsdf_field* f) {
        config.IsValid();
        ensureCtx();
        const hpsdf_config pod = config.toPod();
        void* blk = nullptr;
        size_t sz = 0;
        if (world_ > 1)  // one process / thread per GPU: this rank's part of the sharded build (SetRanks)
            check(hpsdf_create_distributed(ctx_, &pod, f, jobsPerRound_, rank_, world_, gather_, gatherUser_, &blk, &sz, &stats_));
        else
            check(hpsdf_create(ctx_, &pod, f, jobsPerRound_, &blk, &sz, &stats_));
        Clear();  // Octree.cpp:315 (the old tree may have been the CSG operand until now)
        block_ = blk;
        size_ = sz;
        config_ = config;
        mirrorPending_.store(true, std::memory_order_release);  // (deviceTree())
        // continuity.enforce: hpsdf_create has already run the host-side post-process (Octree.cpp:341-344)
        // on the block; LastContinuityStats() reports it.
        hpsdf_continuity_last_stats(&continuity_);
    }
    void csg(int op, const hpsdf_field* inner) {
        hpsdf_tree* t = deviceTree();
        if (!t) throw Error(HPSDF_ERR_STATE, "CSG on an empty octree");
        hpsdf_field* f = nullptr;
        check(hpsdf_field_create_tree_csg(t, op, inner, &f));
        FieldGuard g{f};
        createFrom(config_, f);  // Create(oldTree.config, ...), Octree.cpp:373
    }
    void csg(int op, Func F_) {
        hpsdf_field* inner = nullptr;
        Func fn = std::move(F_);
        check(hpsdf_field_create_callback(&Octree::trampoline, &fn, &inner));
        FieldGuard g{inner};
        csg(op, inner);
    }
    // everything that is a setting of the object rather than its tree: a copy or a moved-into Octree builds the way its source did
    void copySettings(const Octree& o) {
        device_ = o.device_;
        stream_ = o.stream_;
        jobsPerRound_ = o.jobsPerRound_;
        rank_ = o.rank_, world_ = o.world_;
        gather_ = o.gather_, gatherUser_ = o.gatherUser_;
        fitMode_ = o.fitMode_;
        limitNodes_ = o.limitNodes_, limitBytes_ = o.limitBytes_;
        reductionOrder_ = o.reductionOrder_, meshFaceRule_ = o.meshFaceRule_;
        config_ = o.config_;
        stats_ = o.stats_;
        continuity_ = o.continuity_;
    }
    void copyFrom(const Octree& o) {
        copySettings(o);
        if (o.block_) {
            block_ = std::malloc(o.size_);
            if (!block_) throw Error(HPSDF_ERR_OUT_OF_MEMORY, "malloc failed");
            std::memcpy(block_, o.block_, o.size_);
            size_ = o.size_;
            uploadTree();
        }
    }
    void steal(Octree& o) {
        copySettings(o);
        ctx_ = o.ctx_;
        tree_ = o.tree_;
        mirrorPending_.store(o.mirrorPending_.load(), std::memory_order_release);
        o.mirrorPending_.store(false);
        block_ = o.block_;
        size_ = o.size_;
        o.ctx_ = nullptr;
        o.tree_ = nullptr;
        o.block_ = nullptr;
        o.size_ = 0;
    }

    int device_ = 0;
    void* stream_ = nullptr;
    uint64_t jobsPerRound_ = 0;
    int rank_ = 0, world_ = 1;
    int fitMode_ = -1;  // -1: the context's default
    uint64_t limitNodes_ = 0, limitBytes_ = 0;      // SetBuildLimits (0: the library's defaults)
    int reductionOrder_ = -1, meshFaceRule_ = -1;  // Set...Here (-1: the process-wide settings)
    hpsdf_allgather_fn gather_ = nullptr;
    void* gatherUser_ = nullptr;
    mutable hpsdf_ctx* ctx_ = nullptr;
    mutable hpsdf_tree* tree_ = nullptr;
    mutable std::atomic<bool> mirrorPending_{false};  // block_ is a fresh build whose device mirror has not been made yet
    mutable std::mutex mirrorLock_;
    void* block_ = nullptr;  // serialised tree: [nCoeffs][coeffs][nNodes][nodes][config]
    size_t size_ = 0;
    Config config_;
    hpsdf_build_stats stats_{};
    hpsdf_continuity_stats continuity_{};
};

}  // namespace SDF
