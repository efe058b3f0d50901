// Meshing::ObjParser / Meshing::Mesh / Meshing::BVH -- C++ drop-in over the C ABI of hpsdf.h for the mesh
// side of the hot path.
//
// Mirrors the public surface of the reference's Meshing library (Include/Meshing/ObjParser.h:15-52,
// Include/Meshing/Mesh.h:45-77, Include/Meshing/BVH.h:17-30) as far as an SDF build touches it:
//   ObjParser::Load / GetVertices / GetTriIndices / GetVertexNormals / Clear
//   Mesh::CreateFromObj / SignedDistanceAtPt(pt, bvh, threadIdx) / CalculateMeshAABB / Clear
//   BVH::Create(mesh) / Clear
// The reference evaluates one point at a time on the calling host thread (per-thread priority queue, f32);
// here BVH::Create uploads the mesh, its twin half-edges and a BVH to the GPU and batched distance queries run
// there.  The single-point signature SignedDistanceAtPt(pt, bvh) is answered on the calling thread from host copies of
// the device-built arrays (a few microseconds, the device's bits), so an SDF lambda that calls it per sample works as
// it does against the reference; the BVH-less O(n) scan SignedDistanceAtPt(pt) is a launch per call.  Additive:
//   Mesh::SignedDistanceAtPt(const float* xyz, n, float* out, bvh)   batched
//   BVH::Field()   a SDF::DeviceField-compatible handle for Octree::Create(config, field): the fit kernel
//                  samples the mesh on the GPU, no host callback in the loop.
// NNOctree and the reference's bottom-up BVH pairing (Source/Meshing/NNOctree.cpp, BVH.cpp:26-260) are build
// details of its CPU BVH and have no counterpart: any BVH yields the same closest triangle.
#pragma once

#include <memory>
#include <mutex>
#include <vector>

#include "hpsdf_octree.hpp"

namespace Meshing {

class BVH;

class ObjParser {  // Include/Meshing/ObjParser.h:15-52
   public:
    /// Loads a set of vertices, normals and triangles into the object   (ObjParser.cpp:11-35)
    bool Load(const char* objPath_) {
        Clear();
        float* v = nullptr;
        uint64_t* t = nullptr;
        uint64_t nv = 0, nt = 0;
        if (hpsdf_obj_load(objPath_, &v, &nv, &t, &nt) != HPSDF_OK) return false;
        vertices.resize(nv);
        for (uint64_t i = 0; i < nv; ++i) vertices[i] = Eigen::Vector3f(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
        triIndices.assign(t, t + 3 * nt);
        std::free(v);
        std::free(t);
        CalculateVertexNormals();
        return vertices.size() && triIndices.size() && vertexNormals.size();
    }
    void Clear() {
        vertices.clear();
        vertexNormals.clear();
        triIndices.clear();
    }
    const std::vector<u32>& GetTriIndices() const { return triIndices; }
    const std::vector<Eigen::Vector3f>& GetVertices() const { return vertices; }
    const std::vector<Eigen::Vector3f>& GetVertexNormals() const { return vertexNormals; }

   private:
    void CalculateVertexNormals() {  // ObjParser.cpp:141-163: sum of unit face normals, normalised
        vertexNormals.assign(vertices.size(), Eigen::Vector3f(0.0f, 0.0f, 0.0f));
        auto sub = [](const Eigen::Vector3f& a, const Eigen::Vector3f& b) { return Eigen::Vector3f(a(0) - b(0), a(1) - b(1), a(2) - b(2)); };
        auto unit = [](Eigen::Vector3f n) {
            const float z = n(0) * n(0) + (n(1) * n(1) + n(2) * n(2));
            if (z > 0.0f) {
                const float l = std::sqrt(z);
                n = Eigen::Vector3f(n(0) / l, n(1) / l, n(2) / l);
            }
            return n;
        };
        for (size_t i = 0; i + 2 < triIndices.size(); i += 3) {
            const Eigen::Vector3f ab = sub(vertices[triIndices[i + 1]], vertices[triIndices[i]]);
            const Eigen::Vector3f ac = sub(vertices[triIndices[i + 2]], vertices[triIndices[i]]);
            const Eigen::Vector3f n = unit(Eigen::Vector3f(ab(1) * ac(2) - ab(2) * ac(1), ab(2) * ac(0) - ab(0) * ac(2),
                                                           ab(0) * ac(1) - ab(1) * ac(0)));
            for (int k = 0; k < 3; ++k) {
                Eigen::Vector3f& d = vertexNormals[triIndices[i + k]];
                d = Eigen::Vector3f(d(0) + n(0), d(1) + n(1), d(2) + n(2));
            }
        }
        for (auto& n : vertexNormals) n = unit(n);
    }
    std::vector<u32> triIndices;
    std::vector<Eigen::Vector3f> vertices;
    std::vector<Eigen::Vector3f> vertexNormals;
};

class Mesh {  // Include/Meshing/Mesh.h:45-77
   public:
    void Clear() {
        triIndices.clear();
        vertices.clear();
        vertexNormals.clear();
        scan_.reset();
    }
    /// Creates a mesh from a .obj filepath   (Mesh.cpp:15-39).  Whether the mesh is closed -- the reference's
    /// CreateHalfEdges check (Mesh.cpp:87-131) -- is established when a BVH is created from it.
    bool CreateFromObj(const char* objPath_) {
        Clear();
        ObjParser parser;
        if (!parser.Load(objPath_)) return false;
        triIndices = parser.GetTriIndices();
        vertices = parser.GetVertices();
        vertexNormals = parser.GetVertexNormals();
        return true;
    }
    /// Creates a mesh from arrays (additive): 3 floats per vertex, 3 zero-based indices per CCW triangle
    void CreateFromArrays(const float* verts, usize nVerts, const uint64_t* tris, usize nTris) {
        Clear();
        vertices.resize(nVerts);
        for (usize i = 0; i < nVerts; ++i) vertices[i] = Eigen::Vector3f(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
        triIndices.assign(tris, tris + 3 * nTris);
    }
    /// > 0 implies outside mesh, O(n) scan over every triangle   (Mesh.h:53, Mesh.cpp:42-51,134-159) -- on the GPU, one
    /// wave per point; the mesh goes to HBM on the first call
    inline f32 SignedDistanceAtPt(const Eigen::Vector3f& pt_);
    /// Batched form of the O(n) scan (additive)
    inline void SignedDistanceAtPt(const float* xyz, usize n, float* out);
    /// > 0 implies outside mesh   (Mesh.h:54, Mesh.cpp:54-63) -- on the calling thread (host copies of the device-built arrays)
    inline f32 SignedDistanceAtPt(const Eigen::Vector3f& pt_, const BVH& bvh_, const u32 threadIdx_ = 0);
    /// Batched form: xyz interleaved f32 points in, f32 signed distances out (host arrays)
    inline void SignedDistanceAtPt(const float* xyz, usize n, float* out, const BVH& bvh_) const;
    /// Returns a bounding volume for the mesh   (Mesh.h:57, Mesh.cpp:66-84)
    Eigen::AlignedBox3f CalculateMeshAABB() const {
        Eigen::AlignedBox3f box;
        for (const auto& v : vertices)
            for (int a = 0; a < 3; ++a) {
                if (v(a) < box.min()(a)) box.min()(a) = v(a);
                if (v(a) > box.max()(a)) box.max()(a) = v(a);
            }
        return box;
    }
    const std::vector<u32>& GetTriIndices() const { return triIndices; }
    const std::vector<Eigen::Vector3f>& GetVertices() const { return vertices; }

   private:
    friend class BVH;
    std::vector<u32> triIndices;
    std::vector<Eigen::Vector3f> vertexNormals;
    std::vector<Eigen::Vector3f> vertices;
    std::shared_ptr<BVH> scan_;  // device copy of the mesh for the BVH-less overloads (made on first use)
};

class BVH {  // Include/Meshing/BVH.h:17-30
   public:
    BVH() = default;
    BVH(const BVH&) = delete;
    BVH& operator=(const BVH&) = delete;
    ~BVH() {
        Clear();
        hpsdf_ctx_destroy(ctx_);
    }
    /// Which GPU / stream the mesh lives on (default: device 0, library-owned stream); call before Create
    void SetDevice(int device, void* hipStream = nullptr) {
        Clear();
        hpsdf_ctx_destroy(ctx_);
        ctx_ = nullptr;
        device_ = device;
        stream_ = hipStream;
    }
    void Clear() {
        hpsdf_field_destroy(field_);
        field_ = nullptr;
    }
    /// Creates a BVH from a mesh   (BVH.h:26, BVH.cpp:26-72).  false: the mesh is not closed -- the reference
    /// reports that from Mesh::CreateFromObj already (CreateHalfEdges, Mesh.cpp:121-128) -- or no GPU.
    bool Create(const Mesh& mesh_) {
        Clear();
        if (!ctx_ && hpsdf_ctx_create(device_, stream_, &ctx_) != HPSDF_OK) return false;
        // the mesh's own arrays go to the library as they are: a Vector3f is three packed floats (Eigen's as well as the
        // stand-in's) and the reference's "u32" is an 8-byte unsigned integer (Literals.h:9)
        static_assert(sizeof(Eigen::Vector3f) == 3 * sizeof(float), "Vector3f is three packed floats");
        static_assert(sizeof(u32) == sizeof(uint64_t), "the reference's u32 is 8 bytes wide");
        return hpsdf_field_create_mesh(ctx_, reinterpret_cast<const float*>(mesh_.vertices.data()), mesh_.vertices.size(),
                                       reinterpret_cast<const uint64_t*>(mesh_.triIndices.data()), mesh_.triIndices.size() / 3, &field_) == HPSDF_OK;
    }
    /// The mesh as a field Octree::Create samples on the GPU: octree.Create(config, bvh.Field())
    const hpsdf_field* Field() const { return field_; }
    hpsdf_ctx* Context() const { return ctx_; }

   private:
    int device_ = 0;
    void* stream_ = nullptr;
    hpsdf_ctx* ctx_ = nullptr;
    hpsdf_field* field_ = nullptr;
};

inline void Mesh::SignedDistanceAtPt(const float* xyz, usize n, float* out, const BVH& bvh_) const {
    if (!bvh_.Field()) throw SDF::Error(HPSDF_ERR_STATE, "BVH::Create has not succeeded");
    std::vector<double> in(3 * n), res(n);
    for (usize i = 0; i < 3 * n; ++i) in[i] = (double)xyz[i];  // f32 -> f64 -> f32 is exact
    SDF::check(hpsdf_field_eval_host(bvh_.Context(), bvh_.Field(), in.data(), n, res.data()));
    for (usize i = 0; i < n; ++i) out[i] = (float)res[i];  // the field value is the f32 distance widened
}
inline void Mesh::SignedDistanceAtPt(const float* xyz, usize n, float* out) {
    std::shared_ptr<BVH> scan;
    {
        // (the device copy is made by whoever comes first; the reference's method has no state and is called from several threads)
        static std::mutex firstUse;
        std::lock_guard<std::mutex> guard(firstUse);
        if (!scan_) {
            auto b = std::make_shared<BVH>();
            if (!b->Create(*this)) throw SDF::Error(HPSDF_ERR_STATE, hpsdf_last_error());
            scan_ = b;
        }
        scan = scan_;
    }
    std::vector<double> in(3 * n), res(n);
    for (usize i = 0; i < 3 * n; ++i) in[i] = (double)xyz[i];
    SDF::check(hpsdf_field_eval_naive_host(scan->Context(), scan->Field(), in.data(), n, res.data()));
    for (usize i = 0; i < n; ++i) out[i] = (float)res[i];
}
inline f32 Mesh::SignedDistanceAtPt(const Eigen::Vector3f& pt_) {
    const float xyz[3] = {pt_(0), pt_(1), pt_(2)};
    float out = 0.0f;
    SignedDistanceAtPt(xyz, 1, &out);
    return out;
}
inline f32 Mesh::SignedDistanceAtPt(const Eigen::Vector3f& pt_, const BVH& bvh_, const u32) {
    // (a call of a few points is answered on the calling thread from host copies of the field's arrays: ~2 us, the device's bits)
    if (!bvh_.Field()) throw SDF::Error(HPSDF_ERR_STATE, "BVH::Create has not succeeded");
    const double in[3] = {(double)pt_(0), (double)pt_(1), (double)pt_(2)};  // f32 -> f64 -> f32 is exact
    double res = 0.0;
    SDF::check(hpsdf_field_eval_host(bvh_.Context(), bvh_.Field(), in, 1, &res));
    return (float)res;
}

}  // namespace Meshing
