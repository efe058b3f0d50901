// Host-side mesh preparation for the device mesh field: twin half-edges and a BVH.
//
// Only what the per-sample path needs is built: the reference's
// OBJ parser, vertex normals, NNOctree and bottom-up BVH pairing are one-off
// preprocessing outside the hot path.  Any BVH gives the same signed distance:
// the device traversal breaks distance ties towards the lower triangle index,
// i.e. it returns what the reference's linear scan (Mesh.cpp:134-159) returns.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cstring>
#include <numeric>
#include <thread>

#include "runtime.hpp"

namespace hpsdf {
namespace {

unsigned workerCount() {
    unsigned hc = std::thread::hardware_concurrency();
    return hc == 0 ? 1 : (hc > 16 ? 16 : hc);
}

template <typename F>
void parallelFor(unsigned n, F&& f) {  // f(worker) for worker in [0, n)
    std::vector<std::thread> th;
    for (unsigned w = 1; w < n; ++w) th.emplace_back([&f, w] { f(w); });
    f(0);
    for (auto& t : th) t.join();
}

// Mesh::CreateHalfEdges, Mesh.cpp:87-131: edge i runs tris[i] -> tris[next(i)]; its twin is the
// half-edge of the reversed vertex pair.  A directed edge seen twice keeps its first owner
// (std::map::insert does not overwrite).  The two directions of an edge hash to the same bucket, every worker
// owns the buckets b = worker (mod workers) and walks the edges in index order with a table of its own, so the
// result is the sequential one for any worker count.
bool twinHalfEdges(const std::vector<uint32_t>& tris, std::vector<uint32_t>& he) {
    const size_t ne = tris.size();
    he.assign(ne, 0xFFFFFFFFu);
    const unsigned workers = ne < (1u << 16) ? 1 : workerCount();
    auto bucketOf = [&](uint32_t a, uint32_t b) {
        const uint64_t k = ((uint64_t)std::min(a, b) << 32) | std::max(a, b);
        return (unsigned)(((k * 0x9E3779B97F4A7C15ull) >> 40) % workers);
    };
    parallelFor(workers, [&](unsigned w) {
        size_t cap = 16;
        while (cap < (ne / workers + 1) * 3) cap <<= 1;
        std::vector<uint64_t> keys(cap, ~0ull);
        std::vector<uint32_t> vals(cap, 0);
        auto slotOf = [&](uint64_t k) {
            uint64_t h = k * 0x9E3779B97F4A7C15ull;
            h ^= h >> 31;
            return (size_t)(h & (cap - 1));
        };
        for (size_t i = 0; i < ne; ++i) {
            const uint32_t a = tris[i], b = (i % 3 == 2) ? tris[i - 2] : tris[i + 1];
            if (workers > 1 && bucketOf(a, b) != w) continue;
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
    });
    for (size_t i = 0; i < ne; ++i)
        if (he[i] == 0xFFFFFFFFu) return false;
    return true;
}

struct Builder {
    RawVector<float> triBox;  // 6 per triangle
    RawVector<float> cen;     // 3 per triangle
    RawVector<uint32_t> ids;
    RawVector<BvhNode>& nodes;  // preallocated: a subtree over m triangles owns m - 1 consecutive nodes (preorder)

    // Median split on the longest centroid axis.  `node` is the preorder index of this subtree's root, so the
    // layout is fixed before anything is built and the upper levels can hand their right halves to other threads
    // (the result is the sequential build's, byte for byte).  Returns the child reference (node index, or
    // ~triangle for a leaf) and the box of the subtree.
    int32_t build(size_t lo, size_t hi, int32_t node, int spawnDepth, float* bmin, float* bmax) {
        if (hi - lo == 1) {
            const uint32_t t = ids[lo];
            for (int a = 0; a < 3; ++a) {
                bmin[a] = triBox[6 * t + a];
                bmax[a] = triBox[6 * t + 3 + a];
            }
            return ~(int32_t)(t << kMeshLeafShift);  // a leaf of one triangle: slot = triangle (device_types.hpp)
        }
        float cmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmax[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (size_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                cmin[a] = std::min(cmin[a], cen[3 * ids[i] + a]);
                cmax[a] = std::max(cmax[a], cen[3 * ids[i] + a]);
            }
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
        const size_t mid = lo + (hi - lo) / 2;
        std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, [&](uint32_t x, uint32_t y) {
            const float cx = cen[3 * x + axis], cy = cen[3 * y + axis];
            return cx < cy || (cx == cy && x < y);
        });
        float l0[3], h0[3], l1[3], h1[3];
        int32_t c0, c1;
        const int32_t leftNode = node + 1, rightNode = node + (int32_t)(mid - lo);  // left subtree: mid - lo - 1 nodes
        if (spawnDepth > 0 && hi - lo > 4096) {
            std::thread right([&] { c1 = build(mid, hi, rightNode, spawnDepth - 1, l1, h1); });
            c0 = build(lo, mid, leftNode, spawnDepth - 1, l0, h0);
            right.join();
        } else {
            c0 = build(lo, mid, leftNode, 0, l0, h0);
            c1 = build(mid, hi, rightNode, 0, l1, h1);
        }
        BvhNode& n = nodes[node];
        std::memset(&n, 0, sizeof n);
        for (int a = 0; a < 3; ++a) {
            n.lo0[a] = l0[a], n.hi0[a] = h0[a], n.lo1[a] = l1[a], n.hi1[a] = h1[a];
            bmin[a] = std::min(l0[a], l1[a]);
            bmax[a] = std::max(h0[a], h1[a]);
        }
        n.c0 = c0;
        n.c1 = c1;
        return node;
    }
};

}  // namespace

// Only the twins (Mesh::CreateHalfEdges, Mesh.cpp:87-131, with its sequential semantics for edges that occur twice): what a
// non-manifold mesh still needs from the host when everything else was built on the device.
bool hostHalfEdges(const uint64_t* tris, uint64_t nTris, uint64_t nVerts, std::vector<uint32_t>* he) {
    std::vector<uint32_t> t(3 * nTris);
    for (uint64_t i = 0; i < 3 * nTris; ++i) {
        if (tris[i] >= nVerts) return false;
        t[i] = (uint32_t)tris[i];
    }
    return twinHalfEdges(t, *he);
}

bool prepareMesh(const float* verts, uint64_t nVerts, const uint64_t* tris, uint64_t nTris, HostMesh* out) {
    const bool trace = std::getenv("HPSDF_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    out->verts.assign(verts, verts + 3 * nVerts);
    out->tris.resize(3 * nTris);
    for (uint64_t i = 0; i < 3 * nTris; ++i) {
        if (tris[i] >= nVerts) return false;
        out->tris[i] = (uint32_t)tris[i];
    }
    const double t1 = now();
    if (!twinHalfEdges(out->tris, out->halfEdges)) return false;
    const double t2 = now();
    out->bvh.resize(nTris > 1 ? nTris - 1 : 1);  // uninitialised: every node is written by the build
    Builder b{{}, {}, {}, out->bvh};
    b.triBox.resize(6 * nTris);
    b.cen.resize(3 * nTris);
    b.ids.resize(nTris);
    const unsigned workers = nTris < (1u << 16) ? 1 : workerCount();
    parallelFor(workers, [&](unsigned w) {
        for (uint64_t t = nTris * w / workers, end = nTris * (w + 1) / workers; t < end; ++t) {
            b.ids[t] = (uint32_t)t;
            for (int a = 0; a < 3; ++a) {
                const float v0 = out->verts[3 * out->tris[3 * t] + a], v1 = out->verts[3 * out->tris[3 * t + 1] + a],
                            v2 = out->verts[3 * out->tris[3 * t + 2] + a];
                const float lo = std::min(v0, std::min(v1, v2)), hi = std::max(v0, std::max(v1, v2));
                b.triBox[6 * t + a] = lo;
                b.triBox[6 * t + 3 + a] = hi;
                b.cen[3 * t + a] = 0.5f * (lo + hi);
            }
        }
    });
    if (nTris == 1) {  // degenerate: a root with the single triangle on both sides
        BvhNode n;
        std::memset(&n, 0, sizeof n);
        for (int a = 0; a < 3; ++a) {
            n.lo0[a] = n.lo1[a] = b.triBox[a];
            n.hi0[a] = n.hi1[a] = b.triBox[3 + a];
        }
        n.c0 = n.c1 = ~0;
        out->bvh[0] = n;
        return true;
    }
    float bmin[3], bmax[3];
    unsigned hc = std::thread::hardware_concurrency();
    int spawnDepth = 0;  // 2^spawnDepth concurrent subtrees, at most 16
    while (spawnDepth < 4 && (2u << spawnDepth) <= (hc ? hc : 1)) ++spawnDepth;
    const double t3 = now();
    b.build(0, nTris, 0, spawnDepth, bmin, bmax);
    if (trace)
        std::fprintf(stderr, "[prepareMesh] %llu tris: copy %.1f ms, half-edges %.1f ms, boxes %.1f ms, bvh %.1f ms\n",
                     (unsigned long long)nTris, t1 - t0, t2 - t1, t3 - t2, now() - t3);
    return true;
}

}  // namespace hpsdf
