// Mesh preparation on the device: what Meshing::Mesh::CreateHalfEdges (Source/Meshing/Mesh.cpp:87-131) and
// Meshing::BVH::Create (Source/Meshing/BVH.cpp:26-260) do on the CPU -- the twin of every half-edge, and a bounding
// volume hierarchy over the triangles -- done where the mesh is going anyway.  mesh.cpp keeps the host versions (a
// parallel median-split build): 0.15 s for 2.1 M triangles, ten times the Create that follows; this is a few ms.
//
// Twins.  A closed manifold mesh has every directed edge (a, b) exactly once and its reverse (b, a) exactly once.  All
// directed edges go into an open-addressing table keyed by (a << 32 | b) (64-bit compare-and-swap); a second pass looks
// (b, a) up.  For such meshes the result does not depend on insertion order, so it is what the reference's sequential
// std::map pass gives.  A directed edge met twice (non-manifold) or a reverse that is missing (open mesh) is flagged;
// the caller falls back to the host path for the first (whose sequential semantics it reproduces) and reports the
// second as HPSDF_ERR_OPEN_MESH like the reference's `return false` (:121-128).
//
// BVH.  Any hierarchy gives the same distances (the traversal prunes only strictly farther boxes and breaks ties
// towards the lower triangle index), so the reference's bottom-up pairing is not reproduced; this is a
// linear BVH (Karras 2012): 63-bit Morton codes of the triangle-box centres, a stable radix sort (rocPRIM), every inner
// node's range and split found independently from the sorted codes (equal codes ordered by position), boxes fitted
// bottom-up by the second thread to reach a node.  Deterministic: the sort is stable, ranges and splits are pure
// functions of the sorted keys, boxes are min / max.  Node layout as the traversal wants it: 64 bytes holding BOTH
// children's boxes and references (>= 0 an inner node, < 0 a leaf); node 0 is the root.  A leaf is a run of up to
// `leafTris` triangles that are consecutive along the curve (= every subtree that small): the sorted order is the slot
// order of MeshDev::triPre, so a leaf's lower-bound records are neighbours in memory, and the two bottom levels of the
// binary tree -- three quarters of its nodes -- are never visited.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>

#include <mutex>

#include "device_types.hpp"
#include "launch.hpp"
#include "runtime.hpp"

namespace hpsdf {

namespace {

struct MeshBuildFlags {
    unsigned long long badIndex;  // smallest corner (3 t + k) whose vertex index is out of range, or ~0
    uint32_t nonManifold;         // a directed edge occurs twice
    uint32_t open;                // a directed edge without its reverse
    uint32_t boundsLo[3], boundsHi[3];  // scene box of the triangle-box centres, as order-preserving integers
};

__device__ __forceinline__ uint32_t orderedBits(float f) {  // monotone float -> uint
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fromOrderedBits(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// u64 corner indices -> u32, range check
__global__ __launch_bounds__(256) void mb_tris_kernel(const uint64_t* __restrict__ in, uint64_t nCorners, uint64_t nVerts,
                                                      uint32_t* __restrict__ out, MeshBuildFlags* flags) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nCorners) return;
    const uint64_t v = in[c];
    if (v >= nVerts) {
        atomicMin(&flags->badIndex, (unsigned long long)c);
        out[c] = 0;
    } else {
        out[c] = (uint32_t)v;
    }
}

// triangle boxes (as the host build: min / max of the three corners per axis), centres, scene bounds of the centres
__global__ __launch_bounds__(256) void mb_boxes_kernel(const float* __restrict__ triPos, uint32_t nTris, float* __restrict__ triBox,
                                                       MeshBuildFlags* flags) {
    __shared__ uint32_t sLo[3], sHi[3];
    if (threadIdx.x < 3) sLo[threadIdx.x] = 0xFFFFFFFFu, sHi[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < nTris) {
        const float* p = triPos + (size_t)kTriRecordFloats * t;  // a xyz, b xyz, c xyz, normal
        for (int a = 0; a < 3; ++a) {
            const float lo = fminf(p[a], fminf(p[3 + a], p[6 + a])), hi = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a]));
            triBox[6 * (size_t)t + a] = lo;
            triBox[6 * (size_t)t + 3 + a] = hi;
            const uint32_t c = orderedBits(0.5f * (lo + hi));
            atomicMin(&sLo[a], c);
            atomicMax(&sHi[a], c);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        atomicMin(&flags->boundsLo[threadIdx.x], sLo[threadIdx.x]);
        atomicMax(&flags->boundsHi[threadIdx.x], sHi[threadIdx.x]);
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // 21 bits -> every third bit
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void mb_morton_kernel(const float* __restrict__ triBox, uint32_t nTris, const MeshBuildFlags* flags,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ ids) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nTris) return;
    uint64_t code = 0;
    for (int a = 0; a < 3; ++a) {
        const float lo = fromOrderedBits(flags->boundsLo[a]), hi = fromOrderedBits(flags->boundsHi[a]);
        const float c = 0.5f * (triBox[6 * (size_t)t + a] + triBox[6 * (size_t)t + 3 + a]);
        const float ext = hi - lo;
        float u = ext > 0.0f ? (c - lo) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        const uint32_t q = (uint32_t)fminf(u * 2097152.0f, 2097151.0f);
        code |= spread21(q) << (2 - a);
    }
    keys[t] = code;
    ids[t] = t;
}

// common-prefix length of sorted keys i and j (Karras): equal codes are told apart by their position
__device__ __forceinline__ int mbDelta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// inner node i of n - 1: its range of sorted leaves and its split
__global__ __launch_bounds__(256) void mb_hierarchy_kernel(const uint64_t* __restrict__ keys, int n, int leafTris,
                                                           BvhNode* __restrict__ nodes, int32_t* __restrict__ parent /* [2n - 1]: inner 0..n-2, leaves n-1.. */,
                                                           int32_t* __restrict__ ranges /* [3 (n - 1)]: first slot, split, last slot */) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n - 1) return;
    const int d = mbDelta(keys, n, i, i + 1) - mbDelta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = mbDelta(keys, n, i, i - d);
    int lmax = 2;
    while (mbDelta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (mbDelta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = mbDelta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (mbDelta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = d > 0 ? i : j, hi = d > 0 ? j : i;
    const bool leaf0 = lo == gamma, leaf1 = hi == gamma + 1;
    BvhNode& nd = nodes[i];
    // children as the traversal sees them: a subtree of <= leafTris sorted triangles is one leaf (slots lo..gamma, or
    // gamma + 1..hi); the inner nodes below it still get their boxes (mb_fit_kernel climbs through them) but no reference
    const int n0 = gamma - lo + 1, n1 = hi - gamma;
    nd.c0 = n0 <= leafTris ? ~(int32_t)(((uint32_t)lo << kMeshLeafShift) | (uint32_t)(n0 - 1)) : gamma;
    nd.c1 = n1 <= leafTris ? ~(int32_t)(((uint32_t)(gamma + 1) << kMeshLeafShift) | (uint32_t)(n1 - 1)) : gamma + 1;
    nd.pad[0] = nd.pad[1] = 0;
    ranges[3 * (size_t)i] = lo, ranges[3 * (size_t)i + 1] = gamma, ranges[3 * (size_t)i + 2] = hi;
    parent[leaf0 ? (n - 1) + gamma : gamma] = i * 2;          // (child slot in the low bit)
    parent[leaf1 ? (n - 1) + gamma + 1 : gamma + 1] = i * 2 + 1;
    if (i == 0) parent[0] = -1;
}

// boxes bottom-up: one thread per leaf; the second thread to reach a node owns it from there.  The hand-over between the
// two threads (possibly on different XCDs, i.e. behind different L2s) goes through agent-scope atomic stores and loads of
// the box (three 8-byte words) ordered against the arrival counter by a plain s_waitcnt: a __threadfence() per level
// writes the XCD's L2 back each time and made this kernel 12 ms for 2 M triangles (now well under one).
__global__ __launch_bounds__(256) void mb_fit_kernel(const uint32_t* __restrict__ ids, const float* __restrict__ triBox, int n,
                                                     BvhNode* __restrict__ nodes, const int32_t* __restrict__ parent, uint32_t* __restrict__ arrived) {
    const int leaf = (int)(blockIdx.x * 256u + threadIdx.x);
    if (leaf >= n) return;
    const uint32_t t = ids[leaf];
    union Box {
        float f[6];  // lo xyz, hi xyz: the node's lo0[3] hi0[3] (or lo1 hi1) in memory order
        unsigned long long w[3];
    } box;
    for (int a = 0; a < 6; ++a) box.f[a] = triBox[6 * (size_t)t + a];
    int32_t p = parent[(n - 1) + leaf];
    while (p >= 0) {
        const int node = p >> 1, slot = p & 1;
        BvhNode& nd = nodes[node];
        unsigned long long* mine = reinterpret_cast<unsigned long long*>(slot ? nd.lo1 : nd.lo0);
        unsigned long long* other = reinterpret_cast<unsigned long long*>(slot ? nd.lo0 : nd.lo1);
        for (int k = 0; k < 3; ++k) __hip_atomic_store(mine + k, box.w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the box has reached the coherence point before the arrival does
        if (__hip_atomic_fetch_add(&arrived[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // the sibling carries on
        Box sib;
        for (int k = 0; k < 3; ++k) sib.w[k] = __hip_atomic_load(other + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int a = 0; a < 3; ++a) {
            box.f[a] = fminf(box.f[a], sib.f[a]);
            box.f[3 + a] = fmaxf(box.f[3 + a], sib.f[3 + a]);
        }
        p = parent[node];
    }
}

// Slabs (NodeSlab, device_types.hpp): for every child of every node the traversal can visit, a unit vector n, a point g and
// two radii such that every point x of the child's triangles has |n . (x - g)| <= e and |x - g| <= rho.  One wave per node;
// a child's triangles are the sorted slots [a, b], read twice: the sum of the (area-weighted) normals gives n; the range of
// n . (v - c) over the vertices, c the centre of the child's box, puts g in the middle of the slab and gives e, the largest
// |v - c| gives rho (both with explicit allowances for their own rounding, below) -- what rounding leaves open beyond that
// is a few ulps of the mesh's extent and belongs to the traversal's slack (2e-6 of that extent, meshSlack).  The bound is valid for ANY n of unit length: how n was chosen only
// decides how thin the slab is (a patch that bends back on itself gets a thick one and is pruned by its ball and box alone).
// Children of more than kSlabMaxTris triangles get the ball around their box and e = -1, "no slab": up there the boxes
// decide and a wave would loop for too long.  (1024 until round 6, on the argument that a surface is rarely flat at that scale: it
// need not be flat, only flatter than its box is thick, and the walk's dense visits are mostly ABOVE that size -- 131 072: Create on
// the 2.1 M-triangle torus at 1e-6 32.7 -> 31.9 ms, on the 1.3 M-triangle icosphere 12.3 -> 11.4, preparation unchanged; a million: the same,
// and the largest nodes' waves double the preparation: profiles/r06_mesh_steps.txt)
#ifndef HPSDF_SLAB_MAX_TRIS
#define HPSDF_SLAB_MAX_TRIS 131072
#endif
constexpr int kSlabMaxTris = HPSDF_SLAB_MAX_TRIS;
__device__ __forceinline__ float mbWaveSum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float mbWaveMax(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// One child's slab from its triangles [a, b] (sorted slots), by the NT threads that call this together: NT = 64, one wave (children of
// up to kSlabWaveTris triangles: mb_slab_kernel), or NT = 1024, a whole workgroup (larger children, mb_slab_big_kernel: one wave
// walking 131 072 triangles twice was 2.7 ms of a 3.5 ms preparation, alone on its CU).  sRed: 16 x 3 floats of LDS (NT = 1024).
template <int NT>
__device__ __forceinline__ void mbChildSlab(const uint32_t* __restrict__ slotTri, const float* __restrict__ triPos, int a, int b, const float* blo,
                                            const float* bhi, float* sRed, float4* og_, float4* on_) {
    const int tid = threadIdx.x % NT, lane = threadIdx.x & 63;
    const float inf = __builtin_inff();
    const float cx = 0.5f * (blo[0] + bhi[0]), cy = 0.5f * (blo[1] + bhi[1]), cz = 0.5f * (blo[2] + bhi[2]);
    auto sum3 = [&](float& x, float& y, float& z) {
        x = mbWaveSum(x), y = mbWaveSum(y), z = mbWaveSum(z);
        if constexpr (NT > 64) {
            __syncthreads();
            if (lane == 0) sRed[3 * (threadIdx.x >> 6)] = x, sRed[3 * (threadIdx.x >> 6) + 1] = y, sRed[3 * (threadIdx.x >> 6) + 2] = z;
            __syncthreads();
            x = y = z = 0.0f;
            for (int w = 0; w < NT / 64; ++w) x += sRed[3 * w], y += sRed[3 * w + 1], z += sRed[3 * w + 2];  // (every thread, the same order)
        }
    };
    auto max3 = [&](float& x, float& y, float& z) {
        x = mbWaveMax(x), y = mbWaveMax(y), z = mbWaveMax(z);
        if constexpr (NT > 64) {
            __syncthreads();
            if (lane == 0) sRed[3 * (threadIdx.x >> 6)] = x, sRed[3 * (threadIdx.x >> 6) + 1] = y, sRed[3 * (threadIdx.x >> 6) + 2] = z;
            __syncthreads();
            x = y = z = -inf;
            for (int w = 0; w < NT / 64; ++w) x = fmaxf(x, sRed[3 * w]), y = fmaxf(y, sRed[3 * w + 1]), z = fmaxf(z, sRed[3 * w + 2]);
        }
    };
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    for (int s = a + tid; s <= b; s += NT) {
        const float* p = triPos + (size_t)kTriRecordFloats * slotTri[s];
        nx += p[9], ny += p[10], nz += p[11];
    }
    sum3(nx, ny, nz);
    const float len = sqrtf(nx * nx + (ny * ny + nz * nz));
    if (len > 0.0f && len < inf) {
        nx /= len, ny /= len, nz /= len;
        if (!(fabsf(nx * nx + (ny * ny + nz * nz) - 1.0f) <= 1e-6f)) nx = ny = nz = 0.0f;
    } else {
        nx = ny = nz = 0.0f;
    }
    // one pass over the vertices: the range of n . (v - c) and the largest |v - c|, c the centre of the child's box
    float tminNeg = -inf, tmax = -inf, r2 = 0.0f;
    for (int s = a + tid; s <= b; s += NT) {
        const float* p = triPos + (size_t)kTriRecordFloats * slotTri[s];
        for (int v = 0; v < 3; ++v) {
            const float dx = p[3 * v] - cx, dy = p[3 * v + 1] - cy, dz = p[3 * v + 2] - cz;
            const float t = nx * dx + (ny * dy + nz * dz);
            tminNeg = fmaxf(tminNeg, -t), tmax = fmaxf(tmax, t);
            r2 = fmaxf(r2, dx * dx + (dy * dy + dz * dz));
        }
    }
    max3(tminNeg, tmax, r2);
    const float tmin = -tminNeg;
    // g in the middle of the slab.  With g = c + mid n (rounded: each coordinate off by <= u |g_i|, u = 2^-24):
    //   n . (v - g) = n . (v - c) - mid |n|^2 - n . (rounding of g)   =>  |n . (v - g)| <= (tmax - tmin) / 2 + 4 u |v - c| + 6 u |mid| + 2 u M
    //   |v - g| <= |v - c| + |mid| (1 + 3 u) + 2 u M                      (M: the largest coordinate around here)
    // -- both widened by 1e-5 of themselves and 4e-7 of the local scale, i.e. twice what the terms above come to
    const float mid = 0.5f * (tmin + tmax), rc = sqrtf(r2);
    const float gx = cx + mid * nx, gy = cy + mid * ny, gz = cz + mid * nz;
    const float local = fmaxf(fmaxf(fabsf(cx), fabsf(cy)), fabsf(cz)) + rc;
    const float e = (0.5f * (tmax - tmin)) * 1.00001f + 4e-7f * local;
    const float rho = (rc + fabsf(mid)) * 1.00001f + 4e-7f * local;
    if (rho < inf && e < inf && gx - gx == 0.0f && gy - gy == 0.0f && gz - gz == 0.0f) {  // (non-finite input: no bound)
        *og_ = make_float4(gx, gy, gz, rho);
        *on_ = make_float4(nx, ny, nz, e);
    }
}
constexpr int kSlabWaveTris = 1024;  // children of more triangles than this go to mb_slab_big_kernel's list
__global__ __launch_bounds__(256) void mb_slab_kernel(const int32_t* __restrict__ ranges, const uint32_t* __restrict__ slotTri,
                                                      const float* __restrict__ triPos, int n, int leafTris, const BvhNode* __restrict__ nodes,
                                                      NodeSlab* __restrict__ slabs, uint32_t* __restrict__ bigList, uint32_t bigCap) {
    const int node = (int)(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (node >= n - 1) return;
    const int lane = threadIdx.x & 63;
    const int lo = ranges[3 * (size_t)node], gamma = ranges[3 * (size_t)node + 1], hi = ranges[3 * (size_t)node + 2];
    if (node != 0 && hi - lo + 1 <= leafTris) return;  // below a leaf: nothing refers to this node (the root is always visited)
    const BvhNode& nd = nodes[node];
    for (int child = 0; child < 2; ++child) {
        const int a = child ? gamma + 1 : lo, b = child ? hi : gamma;
        const float* blo = child ? nd.lo1 : nd.lo0;
        const float* bhi = child ? nd.hi1 : nd.hi0;
        const float cx = 0.5f * (blo[0] + bhi[0]), cy = 0.5f * (blo[1] + bhi[1]), cz = 0.5f * (blo[2] + bhi[2]);
        // (no slab: the ball around the box, and e = -1 tells the walk to leave the test out)
        const float hx = 0.5f * (bhi[0] - blo[0]), hy = 0.5f * (bhi[1] - blo[1]), hz = 0.5f * (bhi[2] - blo[2]);
        float4 og = make_float4(cx, cy, cz, sqrtf(hx * hx + (hy * hy + hz * hz)) * 1.00001f + 1e-30f), on = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
        const int count = b - a + 1;
        if (count <= kSlabWaveTris) {
            mbChildSlab<64>(slotTri, triPos, a, b, blo, bhi, nullptr, &og, &on);
        } else if (count <= kSlabMaxTris && bigList != nullptr && lane == 0) {
            const uint32_t at = atomicAdd(bigList, 1u);  // (a whole workgroup takes it, behind this kernel: the record below stands until then, and if the list is full)
            if (at < bigCap) bigList[1u + at] = (uint32_t)node * 2u + (uint32_t)child;
        }
        if (lane == 0) {
            if (child)
                slabs[node].g1 = og, slabs[node].n1 = on;
            else
                slabs[node].g0 = og, slabs[node].n0 = on;
        }
    }
}
__global__ __launch_bounds__(1024) void mb_slab_big_kernel(const int32_t* __restrict__ ranges, const uint32_t* __restrict__ slotTri,
                                                           const float* __restrict__ triPos, const BvhNode* __restrict__ nodes,
                                                           NodeSlab* __restrict__ slabs, const uint32_t* __restrict__ bigList, uint32_t bigCap) {
    __shared__ float sRed[3 * 16];
    const uint32_t have = bigList[0] < bigCap ? bigList[0] : bigCap;
    if (blockIdx.x >= have) return;
    const uint32_t entry = bigList[1u + blockIdx.x];
    const int node = (int)(entry >> 1), child = (int)(entry & 1u);
    const int lo = ranges[3 * (size_t)node], gamma = ranges[3 * (size_t)node + 1], hi = ranges[3 * (size_t)node + 2];
    const int a = child ? gamma + 1 : lo, b = child ? hi : gamma;
    const BvhNode& nd = nodes[node];
    float4 og = child ? slabs[node].g1 : slabs[node].g0, on = child ? slabs[node].n1 : slabs[node].n0;  // (the ball, "no slab": kept if the input is not finite)
    mbChildSlab<1024>(slotTri, triPos, a, b, child ? nd.lo1 : nd.lo0, child ? nd.hi1 : nd.hi0, sRed, &og, &on);
    if (threadIdx.x == 0) {
        if (child)
            slabs[node].g1 = og, slabs[node].n1 = on;
        else
            slabs[node].g0 = og, slabs[node].n0 = on;
    }
}

// ---- twins
__device__ __forceinline__ uint64_t mbHash(uint64_t k) {
    k *= 0x9E3779B97F4A7C15ull;
    return k ^ (k >> 29);
}
__global__ __launch_bounds__(256) void mb_edges_insert_kernel(const uint32_t* __restrict__ tris, uint64_t nCorners, unsigned long long* __restrict__ tabKey,
                                                              uint32_t* __restrict__ tabVal, uint64_t mask, MeshBuildFlags* flags) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nCorners) return;
    const uint32_t a = tris[c], b = (c % 3 == 2) ? tris[c - 2] : tris[c + 1];  // Mesh.cpp:96-103
    const unsigned long long key = ((unsigned long long)a << 32) | b;
    uint64_t s = mbHash(key) & mask;
    for (;;) {
        const unsigned long long old = atomicCAS(&tabKey[s], ~0ull, key);
        if (old == ~0ull) {
            tabVal[s] = (uint32_t)c;
            return;
        }
        if (old == key) {  // the same directed edge twice
            atomicOr(&flags->nonManifold, 1u);
            return;
        }
        s = (s + 1) & mask;
    }
}
__global__ __launch_bounds__(256) void mb_edges_lookup_kernel(const uint32_t* __restrict__ tris, uint64_t nCorners, const unsigned long long* __restrict__ tabKey,
                                                              const uint32_t* __restrict__ tabVal, uint64_t mask, uint32_t* __restrict__ halfEdges,
                                                              MeshBuildFlags* flags) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nCorners) return;
    const uint32_t a = tris[c], b = (c % 3 == 2) ? tris[c - 2] : tris[c + 1];
    const unsigned long long rev = ((unsigned long long)b << 32) | a;
    uint64_t s = mbHash(rev) & mask;
    for (;;) {
        const unsigned long long k = tabKey[s];
        if (k == rev) {
            halfEdges[c] = tabVal[s];
            return;
        }
        if (k == ~0ull) {
            halfEdges[c] = 0xFFFFFFFFu;
            atomicOr(&flags->open, 1u);
            return;
        }
        s = (s + 1) & mask;
    }
}

}  // namespace

// The library's private stream-ordered pool of device `dev` (created on first use, one per device, never the device's default
// pool).  It holds on to at most kMeshPoolKeep bytes of freed blocks (HPSDF_MESH_POOL_KEEP_MB overrides): what a destroyed mesh
// field occupied beyond that goes back to the driver, so other allocators in the process (PyTorch's, RCCL's) do not starve.
static uint64_t meshPoolKeepBytes() {
    static const uint64_t keep = [] {
        uint64_t mb = 1024;
        if (const char* v = std::getenv("HPSDF_MESH_POOL_KEEP_MB")) mb = std::strtoull(v, nullptr, 10);
        return mb << 20;
    }();
    return keep;
}
static std::mutex gMeshPoolLock;
static hipMemPool_t gMeshPools[64];
hipMemPool_t meshPool(int dev) {
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(gMeshPoolLock);
    if (!gMeshPools[dev]) {
        hipMemPoolProps props;
        std::memset(&props, 0, sizeof props);
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) != hipSuccess || !pool) {
            (void)hipGetLastError();
            return nullptr;
        }
        uint64_t keep = meshPoolKeepBytes();
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        (void)hipGetLastError();
        gMeshPools[dev] = pool;
    }
    return gMeshPools[dev];
}
// after a mesh field's block has been freed: give back what the pool holds beyond its bound
void meshPoolTrim(int dev) {
    if (dev < 0 || dev >= 64) return;
    hipMemPool_t pool;
    {
        std::lock_guard<std::mutex> g(gMeshPoolLock);
        pool = gMeshPools[dev];
    }
    if (pool) (void)hipMemPoolTrimTo(pool, meshPoolKeepBytes()), (void)hipGetLastError();
}

// Everything a mesh field needs, from host arrays.  Returns HPSDF_OK and fills the device pointers, or an error / a
// request to fall back: *fallback = 1 asks the caller to run the host preparation instead (a mesh too small to be worth it: the
// device buffers are released), *fallback = 2 to supply the twins of a non-manifold mesh from the host (everything else is built and
// stays); device buffers are released on any non-OK return.
int meshBuildDevice(hpsdf_ctx* ctx, const float* verts, uint64_t nVerts, const uint64_t* tris, uint64_t nTris, hpsdf_field* f, int* fallback) {
    *fallback = 0;
    const bool trace = std::getenv("HPSDF_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (nTris < 2) {
        *fallback = 1;
        return HPSDF_OK;
    }
    hipStream_t s = ctx->stream;
    int leafTris = 8;  // triangles per leaf (1..kMeshLeafMax); HPSDF_MESH_LEAF_TRIS overrides (experiments)
    if (const char* lt = std::getenv("HPSDF_MESH_LEAF_TRIS")) leafTris = std::atoi(lt);
    leafTris = leafTris < 1 ? 1 : (leafTris > (int)kMeshLeafMax ? (int)kMeshLeafMax : leafTris);
    const char* ns = std::getenv("HPSDF_MESH_NO_SLABS");  // experiments: boxes only
    const bool noSlabs = ns && ns[0] == '1';
    const uint64_t nCorners = 3 * nTris;
    const int n = (int)nTris;
    uint64_t* dTris64 = nullptr;
    float* dTriBox = nullptr;
    uint64_t *dKeys = nullptr, *dKeysOut = nullptr;
    uint32_t *dIds = nullptr, *dIdsOut = nullptr, *dArrived = nullptr, *dTabVal = nullptr, *dBigList = nullptr;
    // children of more than 1024 triangles (mb_slab_big_kernel's list): disjoint on every level of the tree -- a few thousand on millions of triangles
    const uint32_t bigCap = (uint32_t)std::min<uint64_t>(65535, 8 * (nTris / 1024) + 64);
    unsigned long long* dTabKey = nullptr;
    int32_t *dParent = nullptr, *dRanges = nullptr;
    MeshBuildFlags* dFlags = nullptr;
    void* dSortTmp = nullptr;
    size_t sortTmpBytes = 0;
    uint64_t tabSize = 1;
    while (tabSize < 2 * nCorners) tabSize <<= 1;
    MeshBuildFlags hf;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sortTmpBytes, dKeys, dKeysOut, dIds, dIdsOut, (size_t)nTris, 0, 63, s);
    // two allocations in all (hipMalloc / hipFree cost ~0.1-0.3 ms apiece): the field's five arrays, and the temporaries
    char* fieldBlock = nullptr;
    char* tempBlock = nullptr;
    bool tempPooled = false;
    {
        auto carve = [](size_t& at, size_t bytes) {
            const size_t o = at;
            at = (at + bytes + 255) & ~(size_t)255;
            return o;
        };
        size_t fb = 0;
        const size_t oVerts = carve(fb, 3 * nVerts * sizeof(float)), oTris = carve(fb, nCorners * sizeof(uint32_t)),
                     oTriPos = carve(fb, (size_t)nTris * kTriRecordFloats * sizeof(float)),
                     oTriPre = carve(fb, (size_t)nTris * kTriPreFloats * sizeof(float)), oHe = carve(fb, nCorners * sizeof(uint32_t)),
                     oBvh = carve(fb, (size_t)(n - 1) * sizeof(BvhNode)), oSlab = carve(fb, (size_t)(n - 1) * sizeof(NodeSlab));
        size_t tb = 0;
        const size_t oT64 = carve(tb, nCorners * sizeof(uint64_t)), oBox = carve(tb, 6 * nTris * sizeof(float)),
                     oK = carve(tb, nTris * sizeof(uint64_t)), oK2 = carve(tb, nTris * sizeof(uint64_t)), oI = carve(tb, nTris * sizeof(uint32_t)),
                     oI2 = carve(tb, nTris * sizeof(uint32_t)), oArr = carve(tb, nTris * sizeof(uint32_t)),
                     oPar = carve(tb, 2 * nTris * sizeof(int32_t)), oRan = carve(tb, 3 * nTris * sizeof(int32_t)), oTK = carve(tb, tabSize * sizeof(unsigned long long)),
                     oTV = carve(tb, tabSize * sizeof(uint32_t)), oFl = carve(tb, sizeof(MeshBuildFlags)),
                     oBig = carve(tb, ((size_t)bigCap + 1) * sizeof(uint32_t)),
                     oSort = carve(tb, sortTmpBytes ? sortTmpBytes : 16);
        // Both blocks come from a stream-ordered pool of the library's OWN (meshPool: the application's default pool and its
        // attributes are left alone), which keeps up to a bounded amount of what is freed: a plain hipMalloc of a few hundred
        // megabytes costs anything between 0.05 and 15 ms here, more than the whole build.  The field's block is still released by
        // hipFree (hpsdf_field_destroy: it waits for the device, as before, then trims the pool to its bound); if the pool
        // declines (HPSDF_MESH_NO_POOL=1, or a runtime without it), hipMalloc it is.
        static const bool noPool = std::getenv("HPSDF_MESH_NO_POOL") != nullptr;
        if (e == hipSuccess && !noPool) {
            int dev = 0;
            hipMemPool_t pool = hipGetDevice(&dev) == hipSuccess ? meshPool(dev) : nullptr;
            if (pool) {
                if (hipMallocFromPoolAsync((void**)&fieldBlock, fb, pool, s) != hipSuccess) fieldBlock = nullptr, (void)hipGetLastError();
                if (fieldBlock && hipMallocFromPoolAsync((void**)&tempBlock, tb, pool, s) == hipSuccess) {
                    tempPooled = true;
                } else {
                    tempBlock = nullptr, (void)hipGetLastError();
                }
            }
        }
        if (e == hipSuccess && !fieldBlock) e = hipMalloc((void**)&fieldBlock, fb);
        if (e == hipSuccess && !tempBlock) e = hipMalloc((void**)&tempBlock, tb);
        if (e == hipSuccess) {
            f->dBlock = fieldBlock;
            f->dVerts = (float*)(fieldBlock + oVerts), f->dTris = (uint32_t*)(fieldBlock + oTris), f->dTriPos = (float*)(fieldBlock + oTriPos);
            f->dTriPre = (float*)(fieldBlock + oTriPre);
            f->dHalfEdges = (uint32_t*)(fieldBlock + oHe), f->dBvh = (BvhNode*)(fieldBlock + oBvh);
            f->dSlabs = (NodeSlab*)(fieldBlock + oSlab);
            dRanges = (int32_t*)(tempBlock + oRan);
            dTris64 = (uint64_t*)(tempBlock + oT64), dTriBox = (float*)(tempBlock + oBox), dKeys = (uint64_t*)(tempBlock + oK);
            dKeysOut = (uint64_t*)(tempBlock + oK2), dIds = (uint32_t*)(tempBlock + oI), dIdsOut = (uint32_t*)(tempBlock + oI2);
            dArrived = (uint32_t*)(tempBlock + oArr), dParent = (int32_t*)(tempBlock + oPar), dTabKey = (unsigned long long*)(tempBlock + oTK);
            dTabVal = (uint32_t*)(tempBlock + oTV), dFlags = (MeshBuildFlags*)(tempBlock + oFl), dSortTmp = tempBlock + oSort;
            dBigList = (uint32_t*)(tempBlock + oBig);
        }
    }
    auto freeTemps = [&] {
        if (tempBlock) (void)(tempPooled ? hipFreeAsync(tempBlock, s) : hipFree(tempBlock));  // (stream-ordered: after the kernels that use it)
        tempBlock = nullptr;
    };
    auto freeField = [&] {
        if (fieldBlock) (void)hipFree(fieldBlock);
        fieldBlock = nullptr;
        f->dBlock = nullptr;
        f->dVerts = nullptr, f->dTris = nullptr, f->dTriPos = nullptr, f->dTriPre = nullptr, f->dHalfEdges = nullptr, f->dBvh = nullptr;
        f->dSlabs = nullptr;
    };
    const double t1 = now();
    if (e == hipSuccess) e = hipMemcpyAsync(f->dVerts, verts, 3 * nVerts * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dTris64, tris, nCorners * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        std::memset(&hf, 0, sizeof hf);
        hf.badIndex = ~0ull;
        for (int a = 0; a < 3; ++a) hf.boundsLo[a] = 0xFFFFFFFFu, hf.boundsHi[a] = 0u;
        e = hipMemcpyAsync(dFlags, &hf, sizeof hf, hipMemcpyHostToDevice, s);
    }
    if (e == hipSuccess) e = hipMemsetAsync(dTabKey, 0xFF, tabSize * sizeof(unsigned long long), s);
    if (e == hipSuccess) e = hipMemsetAsync(dArrived, 0, nTris * sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(dBigList, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) {
        freeTemps(), freeField();
        return hipFail(e, "mesh preparation buffers");
    }
    const double t2 = now();
    const unsigned gc = (unsigned)((nCorners + 255) / 256), gt = (unsigned)((nTris + 255) / 256);
    hipLaunchKernelGGL(mb_tris_kernel, dim3(gc), dim3(256), 0, s, dTris64, nCorners, nVerts, f->dTris, dFlags);
    // The twin search (two hash-table kernels, ~1 ms on 2 M triangles: random 8-byte atomics) needs the 32-bit indices only, the BVH chain
    // (boxes -> codes -> sort -> hierarchy -> boxes bottom-up -> slabs, ~1.5 ms, a dozen dependent launches) everything but the twins: they
    // run side by side on two streams (a side stream and two events per host thread and device, made once).  HPSDF_MESH_ONE_STREAM=1: in a row.
    struct Side {
        int dev = -1;
        hipStream_t stream = nullptr;
        hipEvent_t fork = nullptr, join = nullptr;
    };
    static thread_local Side side;
    hipStream_t es = s;  // the stream of the twin search
    {
        static const bool oneStream = std::getenv("HPSDF_MESH_ONE_STREAM") != nullptr;
        int dev = -1;
        if (!oneStream && hipGetDevice(&dev) == hipSuccess) {
            if (side.dev != dev) {
                Side fresh;
                if (hipStreamCreateWithFlags(&fresh.stream, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&fresh.fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&fresh.join, hipEventDisableTiming) == hipSuccess) {
                    fresh.dev = dev;
                    side = fresh;  // (one per host thread and device for the life of the thread; an earlier device's is left to the runtime)
                } else {
                    (void)hipGetLastError();
                }
            }
            if (side.dev == dev && hipEventRecord(side.fork, s) == hipSuccess && hipStreamWaitEvent(side.stream, side.fork, 0) == hipSuccess)
                es = side.stream;
            else
                (void)hipGetLastError();
        }
    }
    hipLaunchKernelGGL(mb_edges_insert_kernel, dim3(gc), dim3(256), 0, es, f->dTris, nCorners, dTabKey, dTabVal, tabSize - 1, dFlags);
    hipLaunchKernelGGL(mb_edges_lookup_kernel, dim3(gc), dim3(256), 0, es, f->dTris, nCorners, dTabKey, dTabVal, tabSize - 1, f->dHalfEdges, dFlags);
    bool joined = es == s;
    if (!joined && hipEventRecord(side.join, es) == hipSuccess) joined = true;  // (waited for below, behind the BVH chain)
    e = launchMeshTriPos(s, f->dVerts, f->dTris, nTris, f->dTriPos, nullptr, nullptr);
    hipLaunchKernelGGL(mb_boxes_kernel, dim3(gt), dim3(256), 0, s, f->dTriPos, (uint32_t)nTris, dTriBox, dFlags);
    hipLaunchKernelGGL(mb_morton_kernel, dim3(gt), dim3(256), 0, s, dTriBox, (uint32_t)nTris, dFlags, dKeys, dIds);
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(dSortTmp, sortTmpBytes, dKeys, dKeysOut, dIds, dIdsOut, (size_t)nTris, 0, 63, s);
    if (e == hipSuccess) e = launchMeshTriPos(s, f->dVerts, f->dTris, nTris, nullptr, dIdsOut, f->dTriPre);  // slot order = sorted order
    hipLaunchKernelGGL(mb_hierarchy_kernel, dim3(gt), dim3(256), 0, s, dKeysOut, n, leafTris, f->dBvh, dParent, dRanges);
    hipLaunchKernelGGL(mb_fit_kernel, dim3(gt), dim3(256), 0, s, dIdsOut, dTriBox, n, f->dBvh, dParent, dArrived);
    if (!noSlabs) {
        hipLaunchKernelGGL(mb_slab_kernel, dim3((unsigned)((n - 1 + 3) / 4)), dim3(256), 0, s, dRanges, dIdsOut, f->dTriPos, n, leafTris, f->dBvh,
                           f->dSlabs, dBigList, bigCap);
        // children of more than kSlabWaveTris triangles: a workgroup each, from the list the kernel above wrote (the grid is its capacity)
        hipLaunchKernelGGL(mb_slab_big_kernel, dim3(bigCap), dim3(1024), 0, s, dRanges, dIdsOut, f->dTriPos, f->dBvh, f->dSlabs, dBigList, bigCap);
    }
    if (es != s) {
        // join: everything behind this point on s (the flags' download, the temporaries' stream-ordered release) is behind the twin search too
        hipError_t je = joined ? hipStreamWaitEvent(s, side.join, 0) : hipErrorUnknown;
        if (je != hipSuccess) je = hipStreamSynchronize(es);  // (no event: wait for the side stream on the host instead)
        if (e == hipSuccess) e = je;
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&hf, dFlags, sizeof hf, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    const double t3 = now();
    freeTemps();
    if (trace)
        std::fprintf(stderr, "[meshBuildDevice] %llu triangles, ms: allocations %.2f, uploads (issue) %.2f, kernels + wait %.2f, frees %.2f\n",
                     (unsigned long long)nTris, t1 - t0, t2 - t1, t3 - t2, now() - t3);
    if (e != hipSuccess) {
        freeField();
        return hipFail(e, "mesh preparation");
    }
    if (hf.badIndex != ~0ull) {
        freeField();
        return fail(HPSDF_ERR_INVALID_ARGUMENT, "triangle " + std::to_string(hf.badIndex / 3) + " refers to a vertex beyond the " +
                                                    std::to_string(nVerts) + " given");
    }
    if (hf.nonManifold) {
        // a directed edge occurs twice: the reference's sequential std::map pass decides which copy is paired (the first), and only
        // the host reproduces that (hostHalfEdges).  The BVH, the slabs and the triangle records do not depend on the twins: they stay.
        f->nVerts = (uint32_t)nVerts;
        f->nTris = (uint32_t)nTris;
        f->nBvhNodes = (uint32_t)(nTris - 1);
        if (noSlabs) f->dSlabs = nullptr;
        f->leafLog2 = 0;
        while ((1 << f->leafLog2) < leafTris) ++f->leafLog2;
        *fallback = 2;
        return HPSDF_OK;
    }
    if (hf.open) {
        freeField();
        return fail(HPSDF_ERR_OPEN_MESH, "mesh is not closed: an edge has no twin (Mesh::CreateHalfEdges)");
    }
    f->nVerts = (uint32_t)nVerts;
    f->nTris = (uint32_t)nTris;
    f->nBvhNodes = (uint32_t)(nTris - 1);
    if (noSlabs) f->dSlabs = nullptr;
    f->leafLog2 = 0;
    while ((1 << f->leafLog2) < leafTris) ++f->leafLog2;
    return HPSDF_OK;
}

hipError_t sortPairsU32(hipStream_t stream, void* tmp, size_t& tmpBytes, const uint32_t* keys, uint32_t* keysOut, const uint32_t* vals, uint32_t* valsOut,
                        size_t n, unsigned bits) {
    return rocprim::radix_sort_pairs(tmp, tmpBytes, keys, keysOut, vals, valsOut, n, 0u, bits, stream);
}

}  // namespace hpsdf
