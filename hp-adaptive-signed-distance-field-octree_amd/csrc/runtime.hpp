// Host runtime objects behind the opaque handles of include/hpsdf.h.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

#include "device_types.hpp"
#include "tables.hpp"

struct hpsdf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    hpsdf::DeviceTables* dTables = nullptr;
};

struct hpsdf_tree {
    int device = 0;
    hpsdf::NodeRec* dNodes = nullptr;
    hpsdf::TopEntry* dTop = nullptr;
    double* dCoeffs = nullptr;
    hpsdf::TreeDev dev{};
    uint64_t nNodes = 0, nCoeffs = 0, nLeaves = 0;
    int maxDegree = 0, maxDepth = 0;
    hpsdf_config config{};
};

enum HostFieldKind { kHostAnalytic = 0, kHostCallback = 1, kHostMesh = 2, kHostTreeCsg = 3 };

struct hpsdf_field {
    int kind = kHostAnalytic;
    std::vector<hpsdf_prim> prims;
    hpsdf_callback cb = nullptr;
    void* user = nullptr;
    // mesh (device resident)
    int device = -1;
    float* dVerts = nullptr;
    uint32_t* dTris = nullptr;
    uint32_t* dHalfEdges = nullptr;
    float* dBvhBoxes = nullptr;
    int32_t* dBvhChild = nullptr;
    uint32_t nTris = 0, nVerts = 0, nBvhNodes = 0;
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

// innermost non-CSG field and the FieldDev the kernels take
const hpsdf_field* innermost(const hpsdf_field* f);
int makeFieldDev(const hpsdf_field* f, const double* dSamples, FieldDev* out);

// host mesh preparation (mesh.cpp)
struct HostMesh {
    std::vector<float> verts;
    std::vector<uint32_t> tris;
    std::vector<uint32_t> halfEdges;
    std::vector<float> bvhBoxes;
    std::vector<int32_t> bvhChild;
};
// returns false when the mesh is not closed (Mesh::CreateHalfEdges, Mesh.cpp:87-131)
bool prepareMesh(const float* verts, uint64_t nVerts, const uint64_t* tris, uint64_t nTris, HostMesh* out);

}  // namespace hpsdf
