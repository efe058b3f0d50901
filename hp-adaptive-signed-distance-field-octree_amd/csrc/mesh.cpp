// Host-side mesh preparation for the device mesh field: twin half-edges and a BVH.
//
// Only what the per-sample path needs is built: the reference's
// OBJ parser, vertex normals, NNOctree and bottom-up BVH pairing are one-off
// preprocessing outside the hot path.  Any BVH gives the same signed distance:
// the device traversal breaks distance ties towards the lower triangle index,
// i.e. it returns what the reference's linear scan (Mesh.cpp:134-159) returns.
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <numeric>

#include "runtime.hpp"

namespace hpsdf {
namespace {

// Mesh::CreateHalfEdges, Mesh.cpp:87-131: edge i runs tris[i] -> tris[next(i)]; its twin is the
// half-edge of the reversed vertex pair.  A directed edge seen twice keeps its first owner
// (std::map::insert does not overwrite).
bool twinHalfEdges(const std::vector<uint32_t>& tris, std::vector<uint32_t>& he) {
    const size_t ne = tris.size();
    he.assign(ne, 0xFFFFFFFFu);
    size_t cap = 16;
    while (cap < ne * 2) cap <<= 1;
    std::vector<uint64_t> keys(cap, ~0ull);
    std::vector<uint32_t> vals(cap, 0);
    auto slotOf = [&](uint64_t k) {
        uint64_t h = k * 0x9E3779B97F4A7C15ull;
        h ^= h >> 31;
        return (size_t)(h & (cap - 1));
    };
    for (size_t i = 0; i < ne; ++i) {
        const uint32_t a = tris[i], b = (i % 3 == 2) ? tris[i - 2] : tris[i + 1];
        const uint64_t rev = ((uint64_t)b << 32) | a, fwd = ((uint64_t)a << 32) | b;
        size_t s = slotOf(rev);
        bool found = false;
        while (keys[s] != ~0ull) {
            if (keys[s] == rev) {
                he[vals[s]] = (uint32_t)i;
                he[i] = vals[s];
                found = true;
                break;
            }
            s = (s + 1) & (cap - 1);
        }
        if (found) continue;
        s = slotOf(fwd);
        bool dup = false;
        while (keys[s] != ~0ull) {
            if (keys[s] == fwd) {
                dup = true;
                break;
            }
            s = (s + 1) & (cap - 1);
        }
        if (!dup) {
            keys[s] = fwd;
            vals[s] = (uint32_t)i;
        }
    }
    for (size_t i = 0; i < ne; ++i)
        if (he[i] == 0xFFFFFFFFu) return false;
    return true;
}

struct Builder {
    const std::vector<float>& verts;
    const std::vector<uint32_t>& tris;
    std::vector<float> triBox;  // 6 per triangle
    std::vector<float> cen;     // 3 per triangle
    std::vector<uint32_t> ids;
    std::vector<float>& boxes;
    std::vector<int32_t>& child;

    int32_t newNode() {
        const int32_t id = (int32_t)(child.size() / 2);
        child.push_back(0);
        child.push_back(0);
        boxes.resize(boxes.size() + 6);
        return id;
    }
    // median split on the longest centroid axis; a leaf reference is ~triangle
    int32_t build(size_t lo, size_t hi) {
        if (hi - lo == 1) return ~(int32_t)ids[lo];
        const int32_t node = newNode();
        float bmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, bmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        float cmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (size_t i = lo; i < hi; ++i) {
            const uint32_t t = ids[i];
            for (int a = 0; a < 3; ++a) {
                bmin[a] = std::min(bmin[a], triBox[6 * t + a]);
                bmax[a] = std::max(bmax[a], triBox[6 * t + 3 + a]);
                cmin[a] = std::min(cmin[a], cen[3 * t + a]);
                cmax[a] = std::max(cmax[a], cen[3 * t + a]);
            }
        }
        for (int a = 0; a < 3; ++a) {
            boxes[6 * (size_t)node + a] = bmin[a];
            boxes[6 * (size_t)node + 3 + a] = bmax[a];
        }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
        const size_t mid = lo + (hi - lo) / 2;
        std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, [&](uint32_t x, uint32_t y) {
            const float cx = cen[3 * x + axis], cy = cen[3 * y + axis];
            return cx < cy || (cx == cy && x < y);
        });
        const int32_t l = build(lo, mid);
        const int32_t r = build(mid, hi);
        child[2 * (size_t)node] = l;
        child[2 * (size_t)node + 1] = r;
        return node;
    }
};

}  // namespace

bool prepareMesh(const float* verts, uint64_t nVerts, const uint64_t* tris, uint64_t nTris, HostMesh* out) {
    out->verts.assign(verts, verts + 3 * nVerts);
    out->tris.resize(3 * nTris);
    for (uint64_t i = 0; i < 3 * nTris; ++i) {
        if (tris[i] >= nVerts) return false;
        out->tris[i] = (uint32_t)tris[i];
    }
    if (!twinHalfEdges(out->tris, out->halfEdges)) return false;
    Builder b{out->verts, out->tris, {}, {}, {}, out->bvhBoxes, out->bvhChild};
    b.triBox.resize(6 * nTris);
    b.cen.resize(3 * nTris);
    b.ids.resize(nTris);
    std::iota(b.ids.begin(), b.ids.end(), 0u);
    for (uint64_t t = 0; t < nTris; ++t)
        for (int a = 0; a < 3; ++a) {
            const float v0 = out->verts[3 * out->tris[3 * t] + a], v1 = out->verts[3 * out->tris[3 * t + 1] + a],
                        v2 = out->verts[3 * out->tris[3 * t + 2] + a];
            const float lo = std::min(v0, std::min(v1, v2)), hi = std::max(v0, std::max(v1, v2));
            b.triBox[6 * t + a] = lo;
            b.triBox[6 * t + 3 + a] = hi;
            b.cen[3 * t + a] = 0.5f * (lo + hi);
        }
    out->bvhBoxes.clear();
    out->bvhChild.clear();
    if (nTris == 1) {  // degenerate: a root with the single triangle on both sides
        out->bvhChild = {~0, ~0};
        out->bvhBoxes.assign(b.triBox.begin(), b.triBox.begin() + 6);
        return true;
    }
    b.build(0, nTris);
    return true;
}

}  // namespace hpsdf
