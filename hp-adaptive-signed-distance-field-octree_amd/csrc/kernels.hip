// gfx950 kernels of the hp-adaptive SDF octree hot path.
//
// Built with -ffp-contract=off: the reference CPU path runs on baseline x86-64
// (no FMA), so every multiply-add below is a separate v_mul_f64 / v_add_f64 and
// every sum runs in the reference's order.  That makes the GPU results
// bit-identical to the CPU restatement (oracle/), which is what keeps the
// octree topology identical (near-ties in the refinement decisions and in the
// heap order would otherwise flip on 1-ulp differences).
//
// Kernels:
//   fit_kernel     Octree::FitPolynomial (Octree.cpp:1007-1093): sampling of F on
//                  the Gauss-Legendre grid fused with the L2 projection and the
//                  error of Octree.cpp:1062-1069.
//   query_kernel   Octree::Query + Octree::FApprox (Octree.cpp:662-702, 859-901).
//   field_kernel   F at arbitrary points (test/diagnostic).
//   pack_kernel    Octree::ReallocCoeffs gather (Octree.cpp:510-552).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstdlib>

#include "acosf_host_libm.hpp"
#include "device_types.hpp"
#include "field_eval.hpp"
#include "launch.hpp"

namespace hpsdf {

// ---- mesh signed distance (all f32) ----------------------------------------
// Source/Meshing/Utility.cpp:5-97, Source/Meshing/Mesh.cpp:54-63,162-242.
// The per-point path below (closest-point routine, pseudo-normals, bounds, the stack traversal) also compiles for the HOST: calls of a
// few points on a plain mesh field are answered on the calling thread from a host copy of the field's arrays (meshEvalHostPoints, at the
// end of this file) -- same statements, -ffp-contract=off on both sides, IEEE divide and square root: the device's bits.
#define HPSDF_HD __host__ __device__
// a bound's square root: the raw 1-ulp instruction on the device, sqrtf on the host (bounds only have to be conservative; their slack is
// four orders of magnitude wider than either)
HPSDF_HD __forceinline__ float boundSqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
struct V3 {
    float x, y, z;
};
HPSDF_HD __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
HPSDF_HD __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
HPSDF_HD __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
HPSDF_HD __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
HPSDF_HD __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
HPSDF_HD __forceinline__ float sqnorm(V3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
HPSDF_HD __forceinline__ V3 normalized(V3 a) {
    const float z = sqnorm(a);
    if (z > 0.0f) {
        const float n = sqrtf(z);
        return {a.x / n, a.y / n, a.z / n};
    }
    return a;
}
HPSDF_HD __forceinline__ V3 meshVert(const MeshDev& m, uint32_t i) {
    return {m.verts[3 * i], m.verts[3 * i + 1], m.verts[3 * i + 2]};
}

constexpr float kEpsF32 = 0.000001f;  // Include/Utility/Literals.h:13

// A query point with a coordinate that is not a finite number has no closest triangle: every comparison of the reference's
// search fails, its bestTri stays -1 and Mesh::SignedDistanceAtPt reads out of bounds (Mesh.cpp:139,157; BVH.cpp:281,343).
// Here such a point takes no part in a traversal and its value is this NaN, on every path.
HPSDF_HD __forceinline__ bool meshPointFinite(V3 p) { return fabsf(p.x) <= FLT_MAX && fabsf(p.y) <= FLT_MAX && fabsf(p.z) <= FLT_MAX; }
HPSDF_HD __forceinline__ float meshNoTriangle() { return hpsdfAcosfBits(0xFFFFFFFFu); }

// What closestSimplex falls back on when the reference's face case has left the triangle: the point the weights describe lies outside,
// so the triangle's closest point is on its boundary -- the nearest of the closest points of its three edges (a + t ab, t the clamped
// projection: well conditioned in f32 whatever the triangle's shape; ties go to ab, then bc, then ca).  A function of its own, called:
// inlined into closestSimplex's five copies inside mesh_sample_kernel it made the kernel 14 % slower WITHOUT ever running (3 000 more
// instructions in the hot loops: the instruction cache), behind a call 3 % (2.1 M-triangle torus at 1e-6: 30.5 ms without any of
// this, 34.5 inlined, 31.6 called).  (If the weights were wrong -- cancellation on a needle, the point really inside -- the answer is off by
// at most the needle's width, upwards: a distance that is too large never breaks a bound.)
HPSDF_HD __noinline__ int closestOnBoundary(V3 pt, V3 a, V3 b, V3 c, V3& q) {
    int code = 0;
    float best = __builtin_inff();
    auto edge = [&](V3 p0, V3 p1, int edgeCode, int v0, int v1) {
        const V3 e = p1 - p0;
        const float den = dot(e, e);
        float t = den > 0.0f ? dot(pt - p0, e) / den : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const V3 x = p0 + t * e;
        const float d = sqnorm(pt - x);
        if (d < best) best = d, q = x, code = t <= 0.0f ? v0 : (t >= 1.0f ? v1 : edgeCode);
    };
    edge(a, b, 4, 0, 1);
    edge(b, c, 5, 1, 2);
    edge(a, c, 6, 0, 2);
    if (!(best < __builtin_inff())) q = a, code = 0;  // (nothing finite: the vertex, like the reference's first case)
    return code;
}

// returns simplex*4 + simplexIdx; closest point in q
// (n: the triangle's unnormalised normal cross(b - a, c - a), precomputed per triangle by mesh_tripos_kernel with these very
// operations -- the value the reference recomputes in every call, Utility.cpp:41)
// The reference's routine, operation by operation, with ONE stated exception (`tol`, a distance: a quarter of the traversal's
// slack).  Vertex and edge cases return points OF the triangle (a vertex; a + t ab with 0 < t < 1).  The face case forms
// q = u a + v b + w c from barycentric quotients and returns it whatever the weights are: the absolute 1e-6 guards of the edge tests
// (snom > eps ...) are lengths SQUARED, so beside a short edge of a needle they let points through that lie well outside the
// triangle -- q is then a point of the triangle's PLANE, up to eps / (shortest altitude) away from it, and its distance lies
// BELOW the triangle's.  No bound can be a bound on that: a search that comes across the needle returns the artefact, one that has
// pruned it (by its box, rightly) does not, and two traversals disagree (tools/fuzz_mesh_bvh.py seeds 100758, 501177: a sphere
// squashed 1000 : 1).  The weights say exactly where q is -- a negative weight m puts it |m| altitudes beyond the opposite edge,
// i.e. |m| |n| / |edge| outside -- so: a face-case point farther than `tol` outside its triangle is not taken by a search that could
// make it its best (`best`: the caller's squared distance so far); it takes the boundary's closest point instead (closestOnBoundary).  Every traversal and the O(n) scan kernel share this function, so they
// agree bit for bit on every mesh; against the reference the value differs exactly where the reference's is such an artefact.
HPSDF_HD int closestSimplex(V3 pt, V3 a, V3 b, V3 c, V3 n, float tol, float best, V3& q) {
    const V3 ab = b - a, ac = c - a, bc = c - b;
    const float snom = dot(pt - a, ab), sdenom = dot(pt - b, a - b);
    const float tnom = dot(pt - a, ac), tdenom = dot(pt - c, a - c);
    if (snom < kEpsF32 && tnom < kEpsF32) {
        q = a;
        return 0;
    }
    const float unom = dot(pt - b, bc), udenom = dot(pt - c, b - c);
    if (sdenom < kEpsF32 && unom < kEpsF32) {
        q = b;
        return 1;
    }
    if (tdenom < kEpsF32 && udenom < kEpsF32) {
        q = c;
        return 2;
    }
    const float vc = dot(n, cross(a - pt, b - pt));
    if (vc < kEpsF32 && snom > kEpsF32 && sdenom > kEpsF32) {
        q = a + (snom / (snom + sdenom)) * ab;
        return 4;
    }
    const float va = dot(n, cross(b - pt, c - pt));
    if (va < kEpsF32 && unom > kEpsF32 && udenom > kEpsF32) {
        q = b + (unom / (unom + udenom)) * bc;
        return 5;
    }
    const float vb = dot(n, cross(c - pt, a - pt));
    if (vb < kEpsF32 && tnom > kEpsF32 && tdenom > kEpsF32) {
        q = a + (tnom / (tnom + tdenom)) * ac;
        return 6;
    }
    const float u = va / (va + vb + vc);
    const float v = vb / (va + vb + vc);
    const float w = 1.0f - u - v;
    q = (u * a + v * b) + w * c;
    const float m = fminf(u, fminf(v, w));  // (NaN weights -- a triangle without area -- stay the reference's NaN point)
    // Two cheap tests in front, for every face-case result.  The weight alone: an altitude is at most the mesh's extent E and
    // tol = 5e-7 max(E, largest coordinate), so a weight above -5e-7 cannot put q more than tol outside.  And whether the value could
    // win at all (d <= best; ties count, they go to the lower triangle index): one that cannot is left as it is -- the triangle's
    // proper distance is larger still, so it loses either way.  On a fine mesh negative weights are the rule, not the exception (the
    // reference's absolute guards send most outside points of a small triangle here), but a test beats its caller's best once or twice
    // per search: what passes both is measured exactly, and the three divisions of closestOnBoundary are spent on potential winners
    // only.  (Substituting at the call sites instead of here cost 18 VGPRs and a wave per SIMD.)  best = +inf passes
    // everything: the winner's recomputation, which must equal what the search stored for it (a winner whose face-case point had left
    // the triangle was, by this rule, substituted when it won).
    if (m < -5e-7f && sqnorm(pt - q) <= best) {
        const float e2 = m == u ? sqnorm(bc) : (m == v ? sqnorm(ac) : sqnorm(ab));  // the edge opposite the negative weight
        if ((m * m) * sqnorm(n) > (tol * tol) * e2) return closestOnBoundary(pt, a, b, c, q);
    }
    return 8;
}

HPSDF_HD V3 faceNormal(const MeshDev& m, uint32_t t) {
    const V3 a = meshVert(m, m.tris[3 * t]), b = meshVert(m, m.tris[3 * t + 1]), c = meshVert(m, m.tris[3 * t + 2]);
    return normalized(cross(b - a, c - a));
}

HPSDF_HD V3 pseudoNormal(const MeshDev& m, uint32_t t, int code) {
    const int simplex = code >> 2, sidx = code & 3;
    if (simplex == 2) return faceNormal(m, t);
    if (simplex == 1) {  // Mesh.cpp:201-215
        const uint32_t adj = m.halfEdges[3 * t + sidx] / 3;
        const float pif = (float)3.14159265359;
        return normalized(pif * faceNormal(m, t) + pif * faceNormal(m, adj));
    }
    // vertex: walk the half-edge fan, Mesh.cpp:218-242
    V3 n = {0.0f, 0.0f, 0.0f};
    uint32_t he = 3 * t + sidx, cur = t;
    int guard = 0;
    do {
        const V3 t0 = meshVert(m, m.tris[3 * cur]), t1 = meshVert(m, m.tris[3 * cur + 1]), t2 = meshVert(m, m.tris[3 * cur + 2]);
        const int k = he % 3;  // (selected, not indexed: an indexed array of registers lives in scratch memory)
        const V3 pk = k == 0 ? t0 : (k == 1 ? t1 : t2), pk1 = k == 0 ? t1 : (k == 1 ? t2 : t0), pk2 = k == 0 ? t2 : (k == 1 ? t0 : t1);
        const V3 ab = pk1 - pk;
        const V3 ac = pk2 - pk;
        const float ang = hpsdfAcosf(dot(normalized(ab), normalized(ac)));  // std::acos of the HOST's libm, bit for bit
        n = n + ang * faceNormal(m, cur);
        he = m.halfEdges[he];
        he = ((he % 3) == 2) ? (he - 2) : (he + 1);
        cur = he / 3;
    } while (cur != t && ++guard < 4096);
    return normalized(n);
}

// A leaf reference (BvhNode::c0 / c1 < 0): its first slot and how many, and the triangle a slot holds.
HPSDF_HD __forceinline__ uint32_t leafFirst(int32_t c) { return (uint32_t)~c >> kMeshLeafShift; }
HPSDF_HD __forceinline__ uint32_t leafCount(int32_t c) { return ((uint32_t)~c & (kMeshLeafMax - 1u)) + 1u; }
// A triPre record (three float4 per leaf slot): g.xyz hu | n.xyz hv | u.xyz triangle -- the triangle lies in the plane through g
// across the unit normal n, inside the rectangle |u . (x - g)| <= hu, |v . (x - g)| <= hv of that plane (u a unit vector along its
// longest edge, v = n x u).  n = u = 0, hu = 0, hv = rho degrades it to the ball of radius rho around g (slivers whose normal cancels).
struct TriPre {
    float4 g, n, u;
};
HPSDF_HD __forceinline__ TriPre loadTriPre(const MeshDev& m, uint32_t slot) {
    return TriPre{m.triPre[3 * (size_t)slot], m.triPre[3 * (size_t)slot + 1], m.triPre[3 * (size_t)slot + 2]};
}
HPSDF_HD __forceinline__ uint32_t triPreTriangle(const TriPre& r) { return hpsdfAcosfWord(r.u.w); }
HPSDF_HD __forceinline__ uint32_t slotTriangle(const MeshDev& m, uint32_t slot) { return hpsdfAcosfWord(m.triPre[3 * (size_t)slot + 2].w); }

// The lower-bound test on a triPre record: with s = n . (p - g) and a = u . (p - g) the squared distance of p from the triangle is
// at least s^2 + max(|a| - hu, 0)^2 + max(sqrt(|p - g|^2 - s^2 - a^2) - hv, 0)^2 -- two dozen instructions against the two hundred
// of the closest-point test.  (Until the middle of round 3 the triangle was bounded by a circle in its plane; the rectangle costs
// five instructions more and has half the area for the 4 : 1 triangles of a stretched grid: 17 -> 10 candidates per sample on
// the displaced torus with the final distance known; no change on equilateral triangles.)
// A box only says "the triangle is somewhere in here": for a sample at distance D from a surface tessellated at size h every
// triangle whose box dips into the ball passes the box test, a patch ~sqrt(2 D h) wide (~300 triangles per sample on a
// 1.3 M-triangle sphere); the plane-and-rectangle bound leaves the ones within ~h.
// A triangle is dropped only if the bound exceeds the best distance by `slack` = 2e-6 of the mesh's scale (its extent, or
// its largest coordinate if that is larger: f32 positions round at that scale; meshSlack has the error budget) and by
// 5e-6 of itself -- so the winner is still exactly the exhaustive scan's (test_mesh_bvh_equals_linear_scan_bitwise, the
// fuzzers).  rejectBound = what the bound is compared with; a NaN bound never drops anything.
// (Bounds, unlike the closest-point arithmetic, need not follow the reference operation by operation: their dot products
// are fused multiply-adds -- three instructions instead of five -- and the square root is the raw v_sqrt_f32, 1 ulp; the
// slack they are compared with is four orders of magnitude wider than either.)
HPSDF_HD __forceinline__ float dotF(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }
HPSDF_HD __forceinline__ float triLowerBound2(V3 p, const TriPre& r) {
    const V3 dx = p - V3{r.g.x, r.g.y, r.g.z};
    const float sd = dotF(V3{r.n.x, r.n.y, r.n.z}, dx), ad = dotF(V3{r.u.x, r.u.y, r.u.z}, dx);
    const float lat2 = __builtin_fmaf(-ad, ad, __builtin_fmaf(-sd, sd, dotF(dx, dx)));
    const float ou = fmaxf(fabsf(ad) - r.g.w, 0.0f);
    const float ov = fmaxf(boundSqrt(fmaxf(lat2, 0.0f)) - r.n.w, 0.0f);
    return __builtin_fmaf(ov, ov, __builtin_fmaf(ou, ou, sd * sd));
}
// The slack (a distance) a lower bound must exceed the best distance by before anything is dropped.  What it has to cover
// (u = 2^-24, D the distance, M the largest coordinate; every f32 subtraction p - g is relatively exact, so most errors
// scale with D and are absorbed by rejectBound's factor 1.00001 on the square, i.e. 5e-6 D):
//   the reference's closest point q = a + t ab (or (u a + v b) + w c) is rounded where it is formed: <= 3 u M off the
//     triangle, so its distance may come out that much below the true one                                  1.8e-7 M
//   the record's plane misses the triangle's vertices by e <= 4e-7 of the scale (mesh_tripre_kernel checks) 4.0e-7 M
//   n . (p - g), |p - g|^2 - (n . (p - g))^2, |n| - 1: ~16 u D                                              (factor)
// 2e-6 of the scale is three and a half times their sum.  (Round 2 ran with 2e-5; the margin it adds around every foot
// point, sqrt(2 D slack), was most of what the samples far from the surface queued: a triangle's width and more.)
HPSDF_HD __forceinline__ float meshSlack(const BvhNode& root) {
    float e2 = 0.0f, big = 0.0f;
    for (int a = 0; a < 3; ++a) {
        const float hi = fmaxf(root.hi0[a], root.hi1[a]), lo = fminf(root.lo0[a], root.lo1[a]);
        e2 += (hi - lo) * (hi - lo);
        big = fmaxf(big, fmaxf(fabsf(hi), fabsf(lo)));
    }
    return 2e-6f * fmaxf(sqrtf(e2), big);
}
// (closestSimplex's face-case tolerance is MeshDev::faceTolOfSlack of that slack -- a quarter by default: a point it accepts lies at
// most that far outside its triangle, so a distance it returns is at most a quarter of the slack, plus the 3 u M of forming q, below
// the triangle's true distance)
HPSDF_HD __forceinline__ float rejectBound(float best, float slack) {
    const float r = sqrtf(best) + slack;
    return r * r * 1.00001f;
}

// Closest triangle by stack traversal of the device BVH, nearer child first.  Ties on squared distance go to
// the lower triangle index, and a box is pruned only when it is strictly farther than the running best (with a
// guard band for f32 rounding of the box distance), so the winner equals the linear scan of
// Mesh::ClosestTriangleToPt (Mesh.cpp:134-159) whatever the visiting order.  `hint` (the winner of the
// caller's previous, nearby query) is tested first so that the bound is tight from the start.
HPSDF_HD float meshSignedDistance(const MeshDev& m, V3 pt, uint32_t& hint) {
    if (!meshPointFinite(pt)) return meshNoTriangle();
    float best = FLT_MAX, reject = __builtin_inff();
    const float slack = meshSlack(m.bvh[0]);
    uint32_t bestTri = 0xFFFFFFFFu;
    int bestCode = 8;
    V3 bestQ = {0.0f, 0.0f, 0.0f};
    auto visitTri = [&](uint32_t t) {
        V3 q;
        const float4 tp[3] = {m.triPos[3 * (size_t)t], m.triPos[3 * (size_t)t + 1], m.triPos[3 * (size_t)t + 2]};
        const int code = closestSimplex(pt, V3{tp[0].x, tp[0].y, tp[0].z}, V3{tp[0].w, tp[1].x, tp[1].y}, V3{tp[1].z, tp[1].w, tp[2].x}, V3{tp[2].y, tp[2].z, tp[2].w}, m.faceTolOfSlack * slack, best, q);
        const float d = sqnorm(pt - q);
        if (d < best || (d == best && t < bestTri)) {
            best = d;
            reject = rejectBound(d, slack);
            bestTri = t;
            bestCode = code;
            bestQ = q;
        }
    };
    auto visitLeaf = [&](int32_t c) {
        for (uint32_t k = 0, first = leafFirst(c), cnt = leafCount(c); k < cnt; ++k) {
            const TriPre rec = loadTriPre(m, first + k);
            if (!(triLowerBound2(pt, rec) > reject)) visitTri(triPreTriangle(rec));
        }
    };
    auto boxDist = [&](const float* lo, const float* hi) {
        const float cx = fminf(fmaxf(pt.x, lo[0]), hi[0]);
        const float cy = fminf(fmaxf(pt.y, lo[1]), hi[1]);
        const float cz = fminf(fmaxf(pt.z, lo[2]), hi[2]);
        return sqnorm(pt - V3{cx, cy, cz});
    };
    auto worthIt = [&](float d) { return !(d > best * 1.00001f + 1e-30f); };
    if (hint < m.nTris) visitTri(hint);
    // one deferred sibling per level: the host's median-split tree is <= 31 levels deep, the device's linear BVH (63-bit
    // Morton codes, equal codes split by position) at most 63 + 32
    int32_t stack[96];
    float stackD[96];
    int sp = 0;
    stack[sp] = 0;
    stackD[sp++] = 0.0f;
    while (sp > 0) {
        --sp;
        if (!worthIt(stackD[sp])) continue;  // the bound may have tightened since the push
        const BvhNode n = m.bvh[stack[sp]];
        const float d0 = boxDist(n.lo0, n.hi0), d1 = boxDist(n.lo1, n.hi1);
        // nearer child first; leaves are resolved at once (they tighten the bound for the sibling)
        const bool swap = d1 < d0;
        const int32_t ca = swap ? n.c1 : n.c0, cb = swap ? n.c0 : n.c1;
        const float da = swap ? d1 : d0, db = swap ? d0 : d1;
        int32_t pushA = -1;
        if (worthIt(da)) {
            if (ca < 0)
                visitLeaf(ca);
            else
                pushA = ca;
        }
        if (worthIt(db)) {
            if (cb < 0)
                visitLeaf(cb);
            else if (sp < 95) {
                stack[sp] = cb;
                stackD[sp++] = db;
            }
        }
        if (pushA >= 0 && sp < 96) {  // on top: popped next
            stack[sp] = pushA;
            stackD[sp++] = da;
        }
    }
    if (bestTri == 0xFFFFFFFFu) return meshNoTriangle();  // (every triangle's distance overflowed)
    hint = bestTri;
    const V3 nrm = pseudoNormal(m, bestTri, bestCode);
    const V3 d = pt - bestQ;
    const float sign = dot(nrm, d) > 0.0f ? 1.0f : -1.0f;
    return sign * sqrtf(sqnorm(d));
}

// The same query for the 64 points of a wave at once (the samples of one cell's grid: a tight cluster).  The wave
// walks ONE traversal: a node is visited if any lane still needs it, its 64 bytes are fetched once (every lane reads
// the same address), every lane keeps its own best and tests a triangle only if its own bound asks for it -- so each
// lane ends with exactly what its own traversal finds (pruning is per lane, ties go to the lower triangle index,
// the visiting order does not matter), while the gathers that dominate the per-lane version (64 lanes x dozens of
// scattered 64-byte nodes) become a few dozen uniform loads.  `stack` : kMeshStack ints of LDS owned by this wave (two entries per level of a BVH at most 31 levels deep).
// Every lane of the wave must call this (inactive lanes with active = false).
constexpr int kMeshStack = 128;
__device__ float meshSignedDistanceWave(const MeshDev& m, V3 pt, bool activeIn, uint32_t& hint, int32_t* stack) {
    const bool active = activeIn && meshPointFinite(pt);
    float best = FLT_MAX;
    float bound = __builtin_inff();  // best * 1.00001f + 1e-30f, kept beside best: what a box distance is compared with
    float reject = __builtin_inff();  // what a triangle's lower bound is compared with (rejectBound)
    const float slack = meshSlack(m.bvh[0]);
    uint32_t bestTri = 0xFFFFFFFFu;
    int bestCode = 8;
    V3 bestQ = {0.0f, 0.0f, 0.0f};
    auto visitTri = [&](uint32_t t) {
        V3 q;
        const float4 tp[3] = {m.triPos[3 * (size_t)t], m.triPos[3 * (size_t)t + 1], m.triPos[3 * (size_t)t + 2]};
        const int code = closestSimplex(pt, V3{tp[0].x, tp[0].y, tp[0].z}, V3{tp[0].w, tp[1].x, tp[1].y}, V3{tp[1].z, tp[1].w, tp[2].x}, V3{tp[2].y, tp[2].z, tp[2].w}, m.faceTolOfSlack * slack, best, q);
        const float d = sqnorm(pt - q);
        if (d < best || (d == best && t < bestTri)) {
            best = d;
            bound = d * 1.00001f + 1e-30f;
            reject = rejectBound(d, slack);
            bestTri = t;
            bestCode = code;
            bestQ = q;
        }
    };
    auto visitLeaf = [&](int32_t c) {
        for (uint32_t k = 0, first = leafFirst(c), cnt = leafCount(c); k < cnt; ++k) {
            const TriPre rec = loadTriPre(m, first + k);
            if (!(triLowerBound2(pt, rec) > reject)) visitTri(triPreTriangle(rec));
        }
    };
    auto boxDist = [&](const float* lo, const float* hi) {  // clamp = median of (p, lo, hi): lo <= hi in every box
        const float cx = __builtin_amdgcn_fmed3f(pt.x, lo[0], hi[0]);
        const float cy = __builtin_amdgcn_fmed3f(pt.y, lo[1], hi[1]);
        const float cz = __builtin_amdgcn_fmed3f(pt.z, lo[2], hi[2]);
        return sqnorm(pt - V3{cx, cy, cz});
    };
    auto worthIt = [&](float d) { return active && !(d > bound); };
    if (active && hint < m.nTris) visitTri(hint);
    const int lane = threadIdx.x & 63;
    // The node being processed lives in registers, the stack only holds the deferred siblings.  Which node comes next is
    // known as soon as the two box tests are in -- the nearer wanted child, else the top of the stack -- so its 64 bytes
    // are asked for BEFORE this node's triangle tests run and arrive behind them (same visiting order as a plain
    // push-both / pop loop; the winner does not depend on the order anyway).
    int sp = 0;  // wave-uniform
#ifdef HPSDF_MESH_STATS_BUILD
    unsigned nVisits = 0, nTriInstr = 0, nTriLanes = 0;
#endif
    BvhNode n = m.bvh[0];
    for (;;) {
#ifdef HPSDF_MESH_STATS_BUILD
        ++nVisits;
#endif
        const float d0 = boxDist(n.lo0, n.hi0), d1 = boxDist(n.lo1, n.hi1);
        const bool w0 = worthIt(d0), w1 = worthIt(d1);
        const unsigned long long b0 = __ballot(w0), b1 = __ballot(w1);
        const int32_t c0 = n.c0, c1 = n.c1;
        int32_t next = -1;
        const bool push0 = c0 >= 0 && b0 != 0ull, push1 = c1 >= 0 && b1 != 0ull;
        if (push0 && push1) {
            // the one that is nearer for the first lane that wants child 0 goes first, its sibling waits on the stack
            const int l0 = __ffsll((long long)b0) - 1;
            const float a0 = __shfl(d0, l0, 64), a1 = __shfl(d1, l0, 64);
            const bool firstIs1 = __builtin_amdgcn_readfirstlane((int)(a1 < a0)) != 0;
            next = firstIs1 ? c1 : c0;
            if (sp < kMeshStack) {
                if (lane == 0) stack[sp] = firstIs1 ? c0 : c1;
                ++sp;
            }
        } else if (push0 || push1) {
            next = push0 ? c0 : c1;
        } else if (sp > 0) {
            --sp;
            next = __builtin_amdgcn_readfirstlane(stack[sp]);
        }
        const BvhNode nn = m.bvh[next >= 0 ? next : 0];  // (the root again when the walk is over: never used)
        // leaves are resolved at once (they tighten the bounds for everything still to come)
        if (c0 < 0 && b0 != 0ull) {
            if (w0) visitLeaf(c0);
#ifdef HPSDF_MESH_STATS_BUILD
            if (m.stats) ++nTriInstr, nTriLanes += (unsigned)__popcll(b0);
#endif
        }
        if (c1 < 0 && b1 != 0ull) {
            if (w1 && worthIt(d1)) visitLeaf(c1);
#ifdef HPSDF_MESH_STATS_BUILD
            if (m.stats) ++nTriInstr, nTriLanes += (unsigned)__popcll(b1);
#endif
        }
        // The test on the two padding words (always zero, mesh.cpp) keeps all sixteen dwords of the prefetch live across
        // the triangle tests: with them dead the register allocator reuses their SGPRs at once and has to wait for the
        // load right where it was issued.  (An empty asm with "s" inputs would do too, but turns every BVH fetch of the
        // kernel into a vector load.)
        if (next < 0 || (nn.pad[0] & nn.pad[1]) == 0xFFFFFFFFu) break;
        n = nn;
    }
#ifdef HPSDF_MESH_STATS_BUILD  // a global atomic in the kernel makes every BVH fetch a vector load: diagnostic builds only
    if (m.stats && lane == 0) {
        atomicAdd(m.stats + 0, 1ull), atomicAdd(m.stats + 1, (unsigned long long)nVisits);
        atomicAdd(m.stats + 2, (unsigned long long)nTriInstr), atomicAdd(m.stats + 3, (unsigned long long)nTriLanes);
    }
#endif
    float r = activeIn ? meshNoTriangle() : 0.0f;
    if (active && bestTri != 0xFFFFFFFFu) {
        hint = bestTri;
        const V3 nrm = pseudoNormal(m, bestTri, bestCode);
        const V3 d = pt - bestQ;
        const float sign = dot(nrm, d) > 0.0f ? 1.0f : -1.0f;
        r = sign * sqrtf(sqnorm(d));
    }
    return r;
}

// A node fetched for the whole wave: the index is wave-uniform and nothing writes the BVH while a kernel walks it, so the
// 64 bytes go through the scalar cache into SGPRs (one s_load_dwordx16) instead of 64 lanes asking the texture path for
// the same line.  The compiler only does that for memory it knows to be invariant -- the constant address space says so;
// through the generic pointer it falls back to four vector loads as soon as the kernel contains an LDS atomic.
__device__ __forceinline__ BvhNode loadNodeUniform(const BvhNode* base, int32_t idx) {
    typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
    const ConstWords w = (ConstWords)(uintptr_t)(base + idx);
    BvhNode n;
    n.lo0[0] = __uint_as_float(w[0]), n.lo0[1] = __uint_as_float(w[1]), n.lo0[2] = __uint_as_float(w[2]);
    n.hi0[0] = __uint_as_float(w[3]), n.hi0[1] = __uint_as_float(w[4]), n.hi0[2] = __uint_as_float(w[5]);
    n.lo1[0] = __uint_as_float(w[6]), n.lo1[1] = __uint_as_float(w[7]), n.lo1[2] = __uint_as_float(w[8]);
    n.hi1[0] = __uint_as_float(w[9]), n.hi1[1] = __uint_as_float(w[10]), n.hi1[2] = __uint_as_float(w[11]);
    n.c0 = (int32_t)w[12], n.c1 = (int32_t)w[13], n.pad[0] = w[14], n.pad[1] = w[15];
    return n;
}

// The same traversal with the triangle tests COMPACTED and FILTERED (what mesh_sample_kernel runs).  In
// meshSignedDistanceWave a leaf is tested the moment it is met, by the lanes whose bound asks for it: ~13 of 64 on a smooth
// 1.3 M-triangle mesh, i.e. the closest-point code runs at a fifth of the machine's width.  Here a leaf only appends one
// (lane, leaf) pair per lane that wants it to ring A in LDS.  Whenever 64 >> leafLog2 pairs are there the wave runs 64
// lower-bound tests at once (triLowerBound2; lane l takes slot l & (W - 1) of pair l >> leafLog2, W = 1 << leafLog2 the most a
// leaf holds; the point comes from its owner's registers by ds_bpermute).  Survivors go to ring B as (lane, triangle);
// whenever 64 are there the wave runs 64 closest-point tests at once and merges every result into its owner's best with ONE
// 64-bit LDS atomic min on (distance bits << 32 | triangle): smallest distance, ties to the lower index -- the linear scan's
// rule (Mesh.cpp:134-159), whatever the order.
// Pruning works on bounds that are refreshed after every closest-point batch: a stale (looser) bound only adds pairs,
// never drops one, so every lane still ends with exactly the triangle its own exhaustive scan finds.  The winner's closest
// point and simplex are recomputed once at the end (same function, same bits).
// Round 3: a child is wanted by a lane only if BOTH its box and its slab (NodeSlab: mesh_build.hip) come within the lane's
// best distance.  A tilted patch of size H fills its box, so by the box alone a sample at distance D wants every patch within
// ~sqrt(D H) of its foot point, at every level of the tree; the slab of a smooth patch is thin and leaves the patches within
// ~H.  The per-lane descent that seeds the bounds follows the smaller of the two children's combined bounds and so ends in
// the leaf under the sample (by boxes alone: a few triangles off, and everything in between passes the lower-bound test).
#ifndef HPSDF_MESH_ABL
#define HPSDF_MESH_ABL 0  // lab builds only (tools/mesh_ablation.sh): what the sampler's phases cost, by leaving one out or running it twice
#endif
constexpr uint32_t kMeshPoolCap = 256;  // (lane, node) pairs a wave's pool holds
#ifndef HPSDF_MESH_SPARSE
#define HPSDF_MESH_SPARSE 16            // a child that at most this many lanes want goes to the pool instead of being walked by the wave
#endif
struct MeshWaveLds {
    unsigned long long best[64];  // per lane: (squared distance bits << 32 | triangle) of the nearest triangle so far
    float px[64], py[64], pz[64]; // per lane: its sample
    float rj[64];                 // per lane: what a lower bound is compared with, rejectBound(best): refreshed with best
    uint32_t nNode[kMeshPoolCap]; // pool N (a stack): (lane, inner node) pairs waiting for their two box / slab tests
    uint32_t aRef[256];           // ring A: (lane, leaf reference) waiting for the lower-bound tests of the leaf's slots
    uint32_t bTri[128];           // ring B: (lane, triangle) waiting for the closest-point test
    int32_t stack[kMeshStack];    // the walk's deferred siblings
    uint8_t nLane[kMeshPoolCap];
    uint8_t aLane[256];
    uint8_t bLane[128];
    uint8_t redo[64];             // lanes whose pairs found the pool full
#ifdef HPSDF_MESH_STALE_STATS
    float bLb[128];               // (diagnostic) ring B: the lower bound each pair passed with
#endif
};
// lower bound of the squared distance from p to anything inside the slab |n . (x - g)| <= e cut by the ball |x - g| <= rho
// (g = (g.xyz), rho = g.w, n = nh.xyz, e = nh.w): along n at least |n . (p - g)| - e, across it at least the distance of p
// from the axis through g minus rho.  Raw v_sqrt_f32 (1 ulp): the caller's slack is four orders of magnitude wider.
__device__ __forceinline__ float slabLowerBound2(V3 p, float4 g, float4 nh) {
    const V3 dx = p - V3{g.x, g.y, g.z};
    const float sd = dotF(V3{nh.x, nh.y, nh.z}, dx);
    const float al = fmaxf(fabsf(sd) - nh.w, 0.0f);
    const float off = fmaxf(__builtin_amdgcn_sqrtf(fmaxf(__builtin_fmaf(-sd, sd, dotF(dx, dx)), 0.0f)) - g.w, 0.0f);
    return __builtin_fmaf(off, off, al * al);
}
__device__ __forceinline__ NodeSlab loadSlabUniform(const NodeSlab* base, int32_t idx) {
    typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
    const ConstWords w = (ConstWords)(uintptr_t)(base + idx);
    NodeSlab s;
    s.g0 = make_float4(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3]));
    s.n0 = make_float4(__uint_as_float(w[4]), __uint_as_float(w[5]), __uint_as_float(w[6]), __uint_as_float(w[7]));
    s.g1 = make_float4(__uint_as_float(w[8]), __uint_as_float(w[9]), __uint_as_float(w[10]), __uint_as_float(w[11]));
    s.n1 = make_float4(__uint_as_float(w[12]), __uint_as_float(w[13]), __uint_as_float(w[14]), __uint_as_float(w[15]));
    return s;
}
__device__ float meshSignedDistanceWaveQ(const MeshDev& m, V3 pt, bool activeIn, MeshWaveLds& L) {
    const bool active = activeIn && meshPointFinite(pt);
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    L.best[lane] = ((unsigned long long)__float_as_uint(FLT_MAX) << 32) | 0xFFFFFFFFull;
    L.px[lane] = pt.x, L.py[lane] = pt.y, L.pz[lane] = pt.z;
    L.redo[lane] = 0;
    uint32_t nCount = 0, aHead = 0, aCount = 0, bHead = 0, bCount = 0;  // wave-uniform
    float bound = __builtin_inff();   // what a box distance is compared with: best * 1.00001f + 1e-30f
    float reject = __builtin_inff();  // what a lower bound is compared with: rejectBound(best)
    const uint32_t lg = m.leafLog2, perBatch = 64u >> lg;               // pairs of ring A one lower-bound batch takes
    const bool slabs = m.slabs != nullptr;
    const uint32_t poolCap = m.poolCap < 128u ? 128u : (m.poolCap > kMeshPoolCap ? kMeshPoolCap : m.poolCap);
    const float inf = __builtin_inff();
#ifdef HPSDF_MESH_STATS_BUILD
    unsigned nVisits = 0, nBound = 0, nClosest = 0;  // [1] nodes visited, [2] pairs through the lower-bound test, [3] through the closest-point test
    unsigned nPairs = 0, nBoundBatches = 0, nClosestBatches = 0, nSeedExact = 0;  // [4] (lane, leaf) pairs, [5] [6] batches, [7] lanes whose seed was the answer
    float seedBest = 0.0f;
    unsigned nPoolPairs = 0;
#endif
    const float slack = meshSlack(loadNodeUniform(m.bvh, 0));
    // the owner's sample and bounds as the batches see them: always the latest best (the closest-point batches write it)
    auto ownerPoint = [&](int o) { return V3{L.px[o], L.py[o], L.pz[o]}; };
    auto ownerBest = [&](int o) { return __uint_as_float((uint32_t)(L.best[o] >> 32)); };
    auto closestBatch = [&](uint32_t cnt) {  // the first cnt (<= 64) pairs of ring B
        const bool on = (uint32_t)lane < cnt;
        const uint32_t at = (bHead + (uint32_t)lane) & 127u;
        const uint32_t t = on ? L.bTri[at] : 0u;
        const int src = on ? (int)L.bLane[at] : lane;
        const V3 p = ownerPoint(src);
#ifdef HPSDF_MESH_STALE_STATS  // [7]: pairs whose bound no longer passes when their closest-point test runs (the owner's best has improved since)
        nSeedExact += (unsigned)__popcll(__ballot(on && L.bLb[at] > L.rj[src]));
#endif
        if (on) {
            V3 q;
            const float4 tp[3] = {m.triPos[3 * (size_t)t], m.triPos[3 * (size_t)t + 1], m.triPos[3 * (size_t)t + 2]};
            closestSimplex(p, V3{tp[0].x, tp[0].y, tp[0].z}, V3{tp[0].w, tp[1].x, tp[1].y}, V3{tp[1].z, tp[1].w, tp[2].x}, V3{tp[2].y, tp[2].z, tp[2].w}, m.faceTolOfSlack * slack, ownerBest(src), q);  // (a stale best only substitutes more often than needed)
            const float d = sqnorm(p - q);
            atomicMin(&L.best[src], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)t);
#if HPSDF_MESH_ABL == 5  // (lab: the closest-point test twice -- what the batches cost is the difference)
            {
                V3 p2 = p, q2;
                asm volatile("" : "+v"(p2.x));
                closestSimplex(p2, V3{tp[0].x, tp[0].y, tp[0].z}, V3{tp[0].w, tp[1].x, tp[1].y}, V3{tp[1].z, tp[1].w, tp[2].x}, V3{tp[2].y, tp[2].z, tp[2].w}, m.faceTolOfSlack * slack, ownerBest(src), q2);
                const float d2 = sqnorm(p2 - q2);
                asm volatile("" ::"v"(d2));
            }
#endif
        }
        __builtin_amdgcn_wave_barrier();
        const float best = ownerBest(lane);
        bound = best * 1.00001f + 1e-30f;
        reject = rejectBound(best, slack);
        L.rj[lane] = reject;
        bHead = (bHead + cnt) & 127u;
        bCount -= cnt;
#ifdef HPSDF_MESH_STATS_BUILD
        nClosest += cnt;
#ifndef HPSDF_MESH_VISIT_HIST
        ++nClosestBatches;
#endif
#endif
    };
    auto boundBatch = [&](uint32_t pairs) {  // the first `pairs` (<= perBatch) pairs of ring A; ring B holds < 64 on entry
        const uint32_t e = (uint32_t)lane >> lg, k = (uint32_t)lane & ((1u << lg) - 1u);
        const uint32_t at = (aHead + e) & 255u;
        const bool have = e < pairs;
        const int32_t ref = have ? (int32_t)L.aRef[at] : -1;
        const int src = have ? (int)L.aLane[at] : lane;
        const bool on = have && k < leafCount(ref);
        const uint32_t slot = leafFirst(ref) + k;
        const V3 p = ownerPoint(src);
        const float rj = L.rj[src];
        bool pass = false;
        uint32_t tri = 0u;
        float lbv = 0.0f;
        if (on) {
            const TriPre rec = loadTriPre(m, slot);
            tri = triPreTriangle(rec);
            lbv = triLowerBound2(p, rec);
            pass = !(lbv > rj);  // (a NaN passes)
#if HPSDF_MESH_ABL == 6  // (lab: the lower-bound test twice)
            {
                V3 p2 = p;
                asm volatile("" : "+v"(p2.x));
                const float lb2 = triLowerBound2(p2, rec);
                asm volatile("" ::"v"(lb2));
            }
#endif
        }
        const unsigned long long pb = __ballot(pass);
        if (pass) {
            const uint32_t pos = (bHead + bCount + (uint32_t)__popcll(pb & below)) & 127u;
            L.bTri[pos] = tri;
            L.bLane[pos] = (uint8_t)src;
#ifdef HPSDF_MESH_STALE_STATS
            L.bLb[pos] = lbv;
#endif
        }
        bCount += (uint32_t)__popcll(pb);
        aHead = (aHead + pairs) & 255u;
        aCount -= pairs;
#ifdef HPSDF_MESH_STATS_BUILD
        nBound += (unsigned)__popcll(__ballot(on));
#ifndef HPSDF_MESH_VISIT_HIST
        ++nBoundBatches;
#endif
#endif
        __builtin_amdgcn_wave_barrier();
        while (bCount >= 64) closestBatch(64);
    };
    auto boxDist6 = [&](V3 p, float lx, float ly, float lz, float hx, float hy, float hz) {  // clamp = median of (p, lo, hi): lo <= hi in every box
        const float cx = __builtin_amdgcn_fmed3f(p.x, lx, hx);
        const float cy = __builtin_amdgcn_fmed3f(p.y, ly, hy);
        const float cz = __builtin_amdgcn_fmed3f(p.z, lz, hz);
        const V3 dd = p - V3{cx, cy, cz};
        return dotF(dd, dd);
    };
#define HPSDF_BOX0(p, nd) boxDist6(p, (nd).lo0[0], (nd).lo0[1], (nd).lo0[2], (nd).hi0[0], (nd).hi0[1], (nd).hi0[2])
#define HPSDF_BOX1(p, nd) boxDist6(p, (nd).lo1[0], (nd).lo1[1], (nd).lo1[2], (nd).hi1[0], (nd).hi1[1], (nd).hi1[2])
    // Seeds.  Everything the walk below queues for a lane is what lies within the lane's best distance so far, and a best
    // that is off by a fraction f of the distance D admits everything within sqrt(2 f) D of the foot point: 1 % is already
    // 0.14 D, a dozen triangles across on a fine mesh.  So before anything is queued every lane gets a best distance that is
    // the final one for most samples, in three steps (HPSDF_MESH_STATS_BUILD counts how many):
    //   1. it walks down to ONE leaf of its own, towards the child whose centre is nearer (lower bounds decide badly here:
    //      a sample sits inside both children's balls and slabs half of the time, and then their tilt decides), and tests
    //      the leaf's triangles, the one with the smallest lower bound first, the others only if their bound allows;
    //   2. it tries the triangles its six neighbours in the wave's 4 x 4 x 4 block of samples ended with, if their lower
    //      bound allows, nearest bound first (a descent that took a wrong turn high up ends several leaves away; the
    //      neighbouring sample's, three triangles further on, most likely did not) -- twice;
    //   3. it walks over the mesh: while the closest point lies on an edge (or corner) of its triangle, the triangle across
    //      that edge is tried -- the distance falls with every step, and the walk ends on the foot point's triangle unless
    //      the surface folds in between.
    // None of this has to be right: the walk below finds whatever is nearer.
#ifndef HPSDF_SEED_EXCHANGE
#define HPSDF_SEED_EXCHANGE 2
#endif
#ifndef HPSDF_SEED_WALK
#define HPSDF_SEED_WALK 6
#endif
    {
        float best = FLT_MAX, rj = inf;
        uint32_t bestTri = 0xFFFFFFFFu, bestSlot = 0xFFFFFFFFu;
        int bestCode = 8;
        auto tryTriangle = [&](uint32_t t, uint32_t slot) {
            V3 q;
            const float4 tp[3] = {m.triPos[3 * (size_t)t], m.triPos[3 * (size_t)t + 1], m.triPos[3 * (size_t)t + 2]};
            const int code = closestSimplex(pt, V3{tp[0].x, tp[0].y, tp[0].z}, V3{tp[0].w, tp[1].x, tp[1].y}, V3{tp[1].z, tp[1].w, tp[2].x}, V3{tp[2].y, tp[2].z, tp[2].w}, m.faceTolOfSlack * slack, best, q);
            const float d = sqnorm(pt - q);
            const bool better = d < best || (d == best && t < bestTri);
            if (better) best = d, bestTri = t, bestSlot = slot, bestCode = code, rj = rejectBound(d, slack);
            return better;
        };
        uint32_t restMask = 0u, seedFirst = 0u;
        if (active) {
            int32_t c = 0;
            do {
                const BvhNode nd = m.bvh[c];
                bool second;
                if (slabs) {
                    const NodeSlab ns = m.slabs[c];
                    second = sqnorm(pt - V3{ns.g1.x, ns.g1.y, ns.g1.z}) < sqnorm(pt - V3{ns.g0.x, ns.g0.y, ns.g0.z});
                } else {
                    second = HPSDF_BOX1(pt, nd) < HPSDF_BOX0(pt, nd);
                }
                c = second ? nd.c1 : nd.c0;
            } while (c >= 0);
            const uint32_t first = leafFirst(c), cnt = leafCount(c);
            seedFirst = first;
            uint32_t kMin = 0;
            float lbMin = inf;
            for (uint32_t k = 0; k < cnt; ++k) {
                const float lb = triLowerBound2(pt, loadTriPre(m, first + k));
                if (lb < lbMin) lbMin = lb, kMin = k;
            }
            tryTriangle(slotTriangle(m, first + kMin), first + kMin);
            uint32_t rest = 0;  // the other slots whose bound the first test's distance allows
            for (uint32_t k = 0; k < cnt; ++k)
                if (k != kMin && !(triLowerBound2(pt, loadTriPre(m, first + k)) > rj)) rest |= 1u << k;
            restMask = rest;
        }
        // (the remaining candidates of all lanes side by side: as many rounds as the lane with the most of them has -- two or
        // three -- instead of one round per slot of the leaf)
        while (__ballot(restMask != 0u) != 0ull) {
            if (restMask != 0u) {
                const uint32_t k = (uint32_t)__ffs((int)restMask) - 1u;
                restMask &= restMask - 1u;
                const TriPre rec = loadTriPre(m, seedFirst + k);
                if (!(triLowerBound2(pt, rec) > rj)) tryTriangle(triPreTriangle(rec), seedFirst + k);
            }
        }
        for (int pass = 0; pass < HPSDF_SEED_EXCHANGE; ++pass) {
            uint32_t candSlot = 0xFFFFFFFFu, candTri = 0u;
            float candLb = inf;
#pragma unroll
            for (int nb = 0; nb < 6; ++nb) {
                const int off = nb < 2 ? 1 : (nb < 4 ? 4 : 16);
                const int src = (lane + ((nb & 1) ? off : 64 - off)) & 63;
                const uint32_t slot = (uint32_t)__shfl((int)bestSlot, src, 64);
                if (active && slot != 0xFFFFFFFFu && slot != bestSlot) {
                    const TriPre rec = loadTriPre(m, slot);
                    const float lb = triLowerBound2(pt, rec);
                    if (lb < candLb && !(lb > rj)) candLb = lb, candSlot = slot, candTri = triPreTriangle(rec);
                }
            }
            if (candSlot != 0xFFFFFFFFu) tryTriangle(candTri, candSlot);
        }
        {
            // (one triangle per lane and step: at a corner the edge that starts there first, and if that does not help the
            // edge that ends there in the next step)
            bool moving = active;
            int second = -1;  // the other edge of a corner whose first edge did not help
            for (int step = 0; step < HPSDF_SEED_WALK && __ballot(moving) != 0ull; ++step) {
                if (moving) {
                    int e = second;
                    second = -1;
                    if (e < 0 && bestCode != 8) {
                        e = bestCode >= 4 ? bestCode - 4 : bestCode;
                        if (bestCode < 4) second = (bestCode + 2) % 3;
                    }
                    moving = false;
                    if (e >= 0) {
                        const uint32_t t = m.halfEdges[3 * bestTri + (uint32_t)e] / 3u;
                        if (tryTriangle(t, 0xFFFFFFFFu))
                            moving = true, second = -1;
                        else
                            moving = second >= 0;
                    }
                }
            }
        }
        if (active) {
            L.best[lane] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)bestTri;
            bound = best * 1.00001f + 1e-30f;
            reject = rj;
        }
        L.rj[lane] = reject;
#ifdef HPSDF_MESH_STATS_BUILD
        seedBest = best;
#endif
    }
    __builtin_amdgcn_wave_barrier();
    // The walk: the wave visits a node if any lane still wants it, every lane against its own bound -- as long as MANY lanes
    // want it.  High up every lane wants the same few nodes and a visit (one 128-byte fetch through the scalar cache, two
    // box and two slab tests across the wave) serves them all; near the leaves the samples' foot points lie a dozen triangles
    // apart, half of the visits serve four lanes or fewer, and the wave-wide tests run for them alone.  So a child that at
    // most HPSDF_MESH_SPARSE lanes want is not walked: each of those lanes drops a (lane, node) pair into a pool in LDS, and
    // whenever 64 pairs are there one batch tests 64 pairs at once -- lane l fetches ITS pair's node and slab, tests both
    // children for the pair's owner (whose sample and bounds come from LDS), inner children that pass go back into the pool
    // (the nearer one on top), leaf children to ring A.  (The pool for EVERYTHING, from the root, was tried first: a lane
    // wants ~170 nodes, 30 lanes share a node on average, and the pool moved 1.4 MB of nodes per wave through the vector
    // memory path where the walk moves 45 KB through the scalar cache -- 10.1 against 7.1 ms.)  Every pair is tested against
    // its owner's own bound, so each lane still ends with exactly what its own exhaustive scan finds.  If the pool is ever
    // full, the owners concerned are marked and walk the tree once more at the end, without a pool.
    auto poolBatch = [&]() {
        const uint32_t cnt = nCount < 64u ? nCount : 64u;
        const bool on = (uint32_t)lane < cnt;
        const uint32_t at = on ? nCount - 1u - (uint32_t)lane : 0u;  // lane 0 takes the top of the stack
        const uint32_t node = on ? L.nNode[at] : 0u;
        const int o = on ? (int)L.nLane[at] : lane;
        nCount -= cnt;
        const V3 p = ownerPoint(o);
        const float bd = ownerBest(o) * 1.00001f + 1e-30f, rj = L.rj[o];
        bool w0 = false, w1 = false;
        int32_t c0 = 0, c1 = 0;
        float k0 = 0.0f, k1 = 0.0f;
        if (on) {
            const BvhNode nd = m.bvh[node];
            k0 = HPSDF_BOX0(p, nd), k1 = HPSDF_BOX1(p, nd);
            w0 = !(k0 > bd), w1 = !(k1 > bd);
            c0 = nd.c0, c1 = nd.c1;
            if (slabs) {
                const NodeSlab ns = m.slabs[node];
                if (ns.n0.w >= 0.0f) {
                    const float sb = slabLowerBound2(p, ns.g0, ns.n0);
                    w0 = w0 && !(sb > rj), k0 = fmaxf(k0, sb);
                }
                if (ns.n1.w >= 0.0f) {
                    const float sb = slabLowerBound2(p, ns.g1, ns.n1);
                    w1 = w1 && !(sb > rj), k1 = fmaxf(k1, sb);
                }
            }
        }
#ifdef HPSDF_MESH_STATS_BUILD
        nPoolPairs += cnt;
#endif
        const bool i0 = w0 && c0 >= 0, i1 = w1 && c1 >= 0, l0 = w0 && c0 < 0, l1 = w1 && c1 < 0;
        {   // inner children back to the pool: the farther one first, so that the nearer one is popped first
            const unsigned long long bAny = __ballot(i0 || i1), bTwo = __ballot(i0 && i1);
            const uint32_t total = (uint32_t)(__popcll(bAny) + __popcll(bTwo));
            if (nCount + total <= poolCap) {
                if (i0 || i1) {
                    const uint32_t pos = nCount + (uint32_t)(__popcll(bAny & below) + __popcll(bTwo & below));
                    const bool both = i0 && i1, nearIs1 = k1 < k0;
                    L.nNode[pos] = (uint32_t)(both ? (nearIs1 ? c0 : c1) : (i0 ? c0 : c1));
                    L.nLane[pos] = (uint8_t)o;
                    if (both) {
                        L.nNode[pos + 1u] = (uint32_t)(nearIs1 ? c1 : c0);
                        L.nLane[pos + 1u] = (uint8_t)o;
                    }
                }
                nCount += total;
            } else if (i0 || i1) {
                L.redo[o] = 1;  // no room: this owner walks the tree again at the end
            }
        }
        {   // leaf children to ring A (which holds < perBatch on entry: at most 63 + 128 of its 256)
            const unsigned long long bAny = __ballot(l0 || l1), bTwo = __ballot(l0 && l1);
            if (l0 || l1) {
                const uint32_t pos = aHead + aCount + (uint32_t)(__popcll(bAny & below) + __popcll(bTwo & below));
                L.aRef[pos & 255u] = (uint32_t)(l0 ? c0 : c1);
                L.aLane[pos & 255u] = (uint8_t)o;
                if (l0 && l1) {
                    L.aRef[(pos + 1u) & 255u] = (uint32_t)c1;
                    L.aLane[(pos + 1u) & 255u] = (uint8_t)o;
                }
            }
            aCount += (uint32_t)(__popcll(bAny) + __popcll(bTwo));
#if defined(HPSDF_MESH_STATS_BUILD) && !defined(HPSDF_MESH_VISIT_HIST)
            nPairs += (unsigned)(__popcll(bAny) + __popcll(bTwo));
#endif
        }
        __builtin_amdgcn_wave_barrier();
        while (aCount >= perBatch) boundBatch(perBatch);
    };
    auto walk = [&](bool act, uint32_t sparse) {  // sparse = 0: no pool, the wave walks everything some lane wants
        BvhNode n = loadNodeUniform(m.bvh, 0);
        NodeSlab sl{};
        if (slabs) sl = loadSlabUniform(m.slabs, 0);
        int sp = 0;  // wave-uniform
        for (;;) {
#ifdef HPSDF_MESH_STATS_BUILD
            ++nVisits;
#endif
            const float d0 = HPSDF_BOX0(pt, n), d1 = HPSDF_BOX1(pt, n);
            bool w0 = act && !(d0 > bound), w1 = act && !(d1 > bound);
            if (slabs) {  // (wave-uniform conditions: the slab came through the scalar cache)
                if (sl.n0.w >= 0.0f && __ballot(w0) != 0ull) w0 = w0 && !(slabLowerBound2(pt, sl.g0, sl.n0) > reject);
                if (sl.n1.w >= 0.0f && __ballot(w1) != 0ull) w1 = w1 && !(slabLowerBound2(pt, sl.g1, sl.n1) > reject);
            }
            const unsigned long long b0 = __ballot(w0), b1 = __ballot(w1);
            const int32_t c0 = n.c0, c1 = n.c1;
#ifdef HPSDF_MESH_VISIT_HIST  // how many lanes a visit serves: [4] <= 4 lanes wanted one of the children, [5] <= 8, [6] <= 16, [7] <= 32
            {
                const int pc = __popcll(b0 | b1);
                nPairs += pc <= 4 ? 1u : 0u, nBoundBatches += pc > 4 && pc <= 8 ? 1u : 0u, nClosestBatches += pc > 8 && pc <= 16 ? 1u : 0u;
                nSeedExact += pc > 16 && pc <= 32 ? 1u : 0u;
            }
#endif
            bool push0 = c0 >= 0 && b0 != 0ull, push1 = c1 >= 0 && b1 != 0ull;
            // inner children few lanes want: into the pool (if it has room for them and for what a batch can add)
            if (push0 && (uint32_t)__popcll(b0) <= sparse && nCount + (uint32_t)__popcll(b0) + 64u <= poolCap) {
                if (w0) {
                    const uint32_t pos = nCount + (uint32_t)__popcll(b0 & below);
                    L.nNode[pos] = (uint32_t)c0;
                    L.nLane[pos] = (uint8_t)lane;
                }
                nCount += (uint32_t)__popcll(b0);
                push0 = false;
            }
            if (push1 && (uint32_t)__popcll(b1) <= sparse && nCount + (uint32_t)__popcll(b1) + 64u <= poolCap) {
                if (w1) {
                    const uint32_t pos = nCount + (uint32_t)__popcll(b1 & below);
                    L.nNode[pos] = (uint32_t)c1;
                    L.nLane[pos] = (uint8_t)lane;
                }
                nCount += (uint32_t)__popcll(b1);
                push1 = false;
            }
            int32_t next = -1;
            if (push0 && push1) {
                // the one that is nearer for the first lane that wants child 0 goes first, its sibling waits on the stack
                const int l0 = __ffsll((long long)b0) - 1;
                const float a0 = __shfl(d0, l0, 64), a1 = __shfl(d1, l0, 64);
                const bool firstIs1 = __builtin_amdgcn_readfirstlane((int)(a1 < a0)) != 0;
                next = firstIs1 ? c1 : c0;
                if (sp < kMeshStack) {
                    if (lane == 0) L.stack[sp] = firstIs1 ? c0 : c1;
                    ++sp;
                }
            } else if (push0 || push1) {
                next = push0 ? c0 : c1;
            } else if (sp > 0) {
                --sp;
                next = __builtin_amdgcn_readfirstlane(L.stack[sp]);
            }
            // the next node's 128 bytes are asked for before this node's leaves are queued and tested
            const int32_t nextIdx = __builtin_amdgcn_readfirstlane(next >= 0 ? next : 0);  // (the root again when the walk is over: never used)
            const BvhNode nn = loadNodeUniform(m.bvh, nextIdx);
            NodeSlab sn{};
            if (slabs) sn = loadSlabUniform(m.slabs, nextIdx);
            for (int side = 0; side < 2; ++side) {  // leaves: one (lane, leaf) pair per lane that wants it
                const int32_t c = side ? c1 : c0;
                const unsigned long long b = side ? b1 : b0;
                if (c >= 0 || b == 0ull) continue;
                if (side ? w1 : w0) {
                    const uint32_t pos = (aHead + aCount + (uint32_t)__popcll(b & below)) & 255u;
                    L.aRef[pos] = (uint32_t)c;
                    L.aLane[pos] = (uint8_t)lane;
                }
                aCount += (uint32_t)__popcll(b);
#if defined(HPSDF_MESH_STATS_BUILD) && !defined(HPSDF_MESH_VISIT_HIST)
                nPairs += (unsigned)__popcll(b);
#endif
                __builtin_amdgcn_wave_barrier();
                while (aCount >= perBatch) boundBatch(perBatch);
            }
            __builtin_amdgcn_wave_barrier();
            while (nCount >= 64u) poolBatch();
            // (the test on the two padding words, always zero, keeps all sixteen dwords of the prefetch live across the
            // leaf tests: with them dead the register allocator reuses their SGPRs at once and waits for the load right here)
            if (next < 0 || (nn.pad[0] & nn.pad[1]) == 0xFFFFFFFFu) break;
            n = nn;
            sl = sn;
        }
        while (nCount > 0u) poolBatch();
        while (aCount) boundBatch(aCount < perBatch ? aCount : perBatch);
        if (bCount) closestBatch(bCount);
    };
#if HPSDF_MESH_ABL != 2  // (lab 2: seeds only)
    walk(active, (uint32_t)HPSDF_MESH_SPARSE);
    {
        const bool again = L.redo[lane] != 0;
        if (__ballot(again) != 0ull) walk(again, 0u);
    }
#endif
#ifdef HPSDF_MESH_STATS_BUILD
#ifndef HPSDF_MESH_VISIT_HIST
#ifndef HPSDF_MESH_STALE_STATS
    nSeedExact = (unsigned)__popcll(__ballot(active && seedBest == ownerBest(lane)));
#endif
#endif
#ifdef HPSDF_MESH_POOL_STATS  // [7]: (lane, node) pairs that went through the pool
    nSeedExact = nPoolPairs;
#endif
#ifdef HPSDF_MESH_SEED_STATS  // how far off the seeds are: [5] within 1e-4 of the final distance, [6] within 1e-2, [4] within 10 %
    {
        const float rs = sqrtf(seedBest), rf = sqrtf(ownerBest(lane));
        nBoundBatches = (unsigned)__popcll(__ballot(active && rs <= rf * 1.0001f));
        nClosestBatches = (unsigned)__popcll(__ballot(active && rs <= rf * 1.01f));
        nPairs = (unsigned)__popcll(__ballot(active && rs <= rf * 1.1f));
    }
#endif
    if (m.stats && lane == 0) {
        atomicAdd(m.stats + 0, 1ull), atomicAdd(m.stats + 1, (unsigned long long)nVisits);
        atomicAdd(m.stats + 2, (unsigned long long)nBound), atomicAdd(m.stats + 3, (unsigned long long)nClosest);
        atomicAdd(m.stats + 4, (unsigned long long)nPairs), atomicAdd(m.stats + 5, (unsigned long long)nBoundBatches);
        atomicAdd(m.stats + 6, (unsigned long long)nClosestBatches), atomicAdd(m.stats + 7, (unsigned long long)nSeedExact);
    }
#endif
    float r = activeIn ? meshNoTriangle() : 0.0f;
    if (active && (uint32_t)(L.best[lane] & 0xFFFFFFFFull) != 0xFFFFFFFFu) {
        const uint32_t bestTri = (uint32_t)(L.best[lane] & 0xFFFFFFFFull);
        V3 bestQ;
        const float4 tp[3] = {m.triPos[3 * (size_t)bestTri], m.triPos[3 * (size_t)bestTri + 1], m.triPos[3 * (size_t)bestTri + 2]};
#if HPSDF_MESH_ABL == 1  // (lab: no recomputation of the winner's closest point, no pseudo-normal)
        (void)tp, (void)bestQ;
        r = sqrtf(ownerBest(lane));
#elif HPSDF_MESH_ABL == 9  // (lab: the closest point, but the face normal for every case)
        const int bestCode = closestSimplex(pt, V3{tp[0].x, tp[0].y, tp[0].z}, V3{tp[0].w, tp[1].x, tp[1].y}, V3{tp[1].z, tp[1].w, tp[2].x}, V3{tp[2].y, tp[2].z, tp[2].w}, m.faceTolOfSlack * slack, __builtin_inff(), bestQ);
        (void)bestCode;
        const V3 nrm = pseudoNormal(m, bestTri, 8);
        const V3 d = pt - bestQ;
        const float sign = dot(nrm, d) > 0.0f ? 1.0f : -1.0f;
        r = sign * sqrtf(sqnorm(d));
#else
        const int bestCode = closestSimplex(pt, V3{tp[0].x, tp[0].y, tp[0].z}, V3{tp[0].w, tp[1].x, tp[1].y}, V3{tp[1].z, tp[1].w, tp[2].x}, V3{tp[2].y, tp[2].z, tp[2].w}, m.faceTolOfSlack * slack, __builtin_inff(), bestQ);
        const V3 nrm = pseudoNormal(m, bestTri, bestCode);
        const V3 d = pt - bestQ;
        const float sign = dot(nrm, d) > 0.0f ? 1.0f : -1.0f;
        r = sign * sqrtf(sqnorm(d));
#endif
    }
    return r;
}
#undef HPSDF_BOX0
#undef HPSDF_BOX1

// ---------------------------------------------------------------------------
// tree evaluation: Octree::Query (Octree.cpp:662-702) and FApprox (:859-901)
// ---------------------------------------------------------------------------

// Basis index table in graded order (total degree, then first and second index).
struct BasisIdx {
    unsigned char v[456][3];
    constexpr BasisIdx() : v() {
        int row = 0;
        for (int p = 0; p <= 12; ++p)
            for (int a = 0; a <= p; ++a)
                for (int b = 0; a + b <= p; ++b) {
                    v[row][0] = (unsigned char)a;
                    v[row][1] = (unsigned char)b;
                    v[row][2] = (unsigned char)(p - a - b);
                    ++row;
                }
    }
};
__device__ constexpr BasisIdx kBasis{};
__host__ __device__ constexpr int coeffCount(int p) {
    // the reference's (u32)(1/6.0 * (p+1)*(p+2)*(p+3)) evaluates to 83 for p = 6
    return p == 6 ? 83 : (p + 1) * (p + 2) * (p + 3) / 6;
}

// sNl: [13][11] normalisation table, sRec: [13][2] recurrence constants (LDS).  cv holds the leaf's
// coefficients; values and summation order are those of Octree.cpp:888-898.
template <int P, int NV>
__device__ __forceinline__ double evalLeafVals(const double (&cv)[NV], double ux, double uy, double uz, int depth,
                                               const double* sNl, const double* sRec) {
    constexpr int N = coeffCount(P);
    static_assert(NV >= N, "coefficient registers");
    double tx[P + 1], ty[P + 1], tz[P + 1];
    tx[0] = ty[0] = tz[0] = sNl[depth];
    double xm2 = 0.0, xm1 = 1.0, ym2 = 0.0, ym1 = 1.0, zm2 = 0.0, zm1 = 1.0;
#pragma unroll
    for (int j = 1; j <= P; ++j) {
        const double r0 = sRec[2 * j], r1 = sRec[2 * j + 1], nl = sNl[j * 11 + depth];
        const double lx = r0 * ux * xm1 - r1 * xm2;
        const double ly = r0 * uy * ym1 - r1 * ym2;
        const double lz = r0 * uz * zm1 - r1 * zm2;
        xm2 = xm1, xm1 = lx, ym2 = ym1, ym1 = ly, zm2 = zm1, zm1 = lz;
        tx[j] = lx * nl, ty[j] = ly * nl, tz[j] = lz * nl;
    }
    double f = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double lp = tx[kBasis.v[i][0]];
        lp = lp * ty[kBasis.v[i][1]];
        lp = lp * tz[kBasis.v[i][2]];
        f = f + cv[i] * lp;
    }
    return f;
}

// FApprox for the lanes of a wave whose leaves have different degrees <= P, in ONE pass: the basis rows of degree d are the first
// coeffCount(d) rows of the degree-P basis (Utility.h's table is ordered by total degree) and the sum runs row by row, so the value of
// a leaf of degree d is the running sum after row coeffCount(d) - 1 -- the very additions evalLeafVals<d> performs, on the same
// Legendre values (the recurrence is the same for j <= d) -- and each lane keeps the running sum at its own degree's last row.  A wave
// with degree-2 and degree-3 leaves used to run both bodies one after the other (tools/query_general_floor.py: the polynomial is a
// fifth of query_general's time); rows beyond a lane's degree multiply whatever its registers hold there: never read.
template <int P, int NV>
__device__ __forceinline__ double evalLeafValsMixed(const double (&cv)[NV], double ux, double uy, double uz, int depth, uint32_t degree,
                                                    const double* sNl, const double* sRec) {
    constexpr int N = coeffCount(P);
    static_assert(NV >= N, "coefficient registers");
    double tx[P + 1], ty[P + 1], tz[P + 1];
    tx[0] = ty[0] = tz[0] = sNl[depth];
    double xm2 = 0.0, xm1 = 1.0, ym2 = 0.0, ym1 = 1.0, zm2 = 0.0, zm1 = 1.0;
#pragma unroll
    for (int j = 1; j <= P; ++j) {
        const double r0 = sRec[2 * j], r1 = sRec[2 * j + 1], nl = sNl[j * 11 + depth];
        const double lx = r0 * ux * xm1 - r1 * xm2;
        const double ly = r0 * uy * ym1 - r1 * ym2;
        const double lz = r0 * uz * zm1 - r1 * zm2;
        xm2 = xm1, xm1 = lx, ym2 = ym1, ym1 = ly, zm2 = zm1, zm1 = lz;
        tx[j] = lx * nl, ty[j] = ly * nl, tz[j] = lz * nl;
    }
    double f = 0.0, mine = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double lp = tx[kBasis.v[i][0]];
        lp = lp * ty[kBasis.v[i][1]];
        lp = lp * tz[kBasis.v[i][2]];
        f = f + cv[i] * lp;
#pragma unroll
        for (int dgr = 0; dgr < P; ++dgr)
            if (i == coeffCount(dgr) - 1) mine = degree == (uint32_t)dgr ? f : mine;
    }
    return degree == (uint32_t)P ? f : mine;
}

// Octree::FApproxWithGradient (Octree.cpp:904-985) for a compile-time degree: the value as FApprox, the "gradient"
// as the reference forms it -- per axis k the central difference of sum_r c_r * Lhat_{idx[r][k]}(u_k +- eps), i.e. with
// the other two axes' factors left out (:956-968) -- then normalised.  Same statements, same order as
// queryPointWithGradient's any-degree loop (and as the oracle), with the tables in registers.
template <int P, int NV>
__device__ __forceinline__ double evalLeafGradVals(const double (&cv)[NV], const double (&u)[3], int depth, const double* sNl,
                                                   const double* sRec, double (&g)[3], int left) {
    constexpr int N = coeffCount(P);
    static_assert(NV >= N, "coefficient registers");
    const double eps = 0.0001;
    double L0[3][P + 1];  // normalised Legendre values at u, per axis: the value's factors
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double Lp[P + 1], Lm[P + 1];
        L0[k][0] = Lp[0] = Lm[0] = sNl[depth];
        double a2 = 0.0, a1 = 1.0, b2 = 0.0, b1 = 1.0, c2 = 0.0, c1 = 1.0;
#pragma unroll
        for (int j = 1; j <= P; ++j) {
            const double r0 = sRec[2 * j], r1 = sRec[2 * j + 1], nl = sNl[j * 11 + depth];
            const double a0 = r0 * u[k] * a1 - r1 * a2;          // :937
            const double b0 = r0 * (u[k] + eps) * b1 - r1 * b2;  // :941
            const double c0 = r0 * (u[k] - eps) * c1 - r1 * c2;  // :945
            a2 = a1, a1 = a0, b2 = b1, b1 = b0, c2 = c1, c1 = c0;
            L0[k][j] = a0 * nl, Lp[j] = b0 * nl, Lm[j] = c0 * nl;
        }
        double p1 = 0.0, m1 = 0.0;
#pragma unroll
        for (int r = 0; r < N; ++r) {  // :956-968
            p1 = p1 + cv[r] * Lp[kBasis.v[r][k]];
            m1 = m1 + cv[r] * Lm[kBasis.v[r][k]];
        }
        g[k] = (p1 - m1) / (2.0 * eps);
    }
    const double z = left ? sum3<true>(g[0] * g[0], g[1] * g[1], g[2] * g[2]) : sum3<false>(g[0] * g[0], g[1] * g[1], g[2] * g[2]);  // Eigen normalize()
    if (z > 0.0) {
        const double nrm = sqrt(z);
        g[0] = g[0] / nrm, g[1] = g[1] / nrm, g[2] = g[2] / nrm;
    }
    double f = 0.0;  // :972-984
#pragma unroll
    for (int r = 0; r < N; ++r) {
        double lp = L0[0][kBasis.v[r][0]];
        lp = lp * L0[1][kBasis.v[r][1]];
        lp = lp * L0[2][kBasis.v[r][2]];
        f = f + cv[r] * lp;
    }
    return f;
}

// FApproxWithGradient for a wave's mix of degrees <= P in one pass, as evalLeafValsMixed: every running sum -- the two one-sided sums
// of each axis and the value -- is kept at the last row of the lane's own degree.
// (NODIV: lab builds only, tools/query_general_floor.py -- the six IEEE divisions and the square root left out)
template <int P, int NV, bool NODIV = false>
__device__ __forceinline__ double evalLeafGradValsMixed(const double (&cv)[NV], const double (&u)[3], int depth, uint32_t degree, const double* sNl,
                                                        const double* sRec, double (&g)[3], int left) {
    constexpr int N = coeffCount(P);
    static_assert(NV >= N, "coefficient registers");
    const double eps = 0.0001;
    double L0[3][P + 1];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double Lp[P + 1], Lm[P + 1];
        L0[k][0] = Lp[0] = Lm[0] = sNl[depth];
        double a2 = 0.0, a1 = 1.0, b2 = 0.0, b1 = 1.0, c2 = 0.0, c1 = 1.0;
#pragma unroll
        for (int j = 1; j <= P; ++j) {
            const double r0 = sRec[2 * j], r1 = sRec[2 * j + 1], nl = sNl[j * 11 + depth];
            const double a0 = r0 * u[k] * a1 - r1 * a2;          // :937
            const double b0 = r0 * (u[k] + eps) * b1 - r1 * b2;  // :941
            const double c0 = r0 * (u[k] - eps) * c1 - r1 * c2;  // :945
            a2 = a1, a1 = a0, b2 = b1, b1 = b0, c2 = c1, c1 = c0;
            L0[k][j] = a0 * nl, Lp[j] = b0 * nl, Lm[j] = c0 * nl;
        }
        double p1 = 0.0, m1 = 0.0, pMine = 0.0, mMine = 0.0;
#pragma unroll
        for (int r = 0; r < N; ++r) {  // :956-968
            p1 = p1 + cv[r] * Lp[kBasis.v[r][k]];
            m1 = m1 + cv[r] * Lm[kBasis.v[r][k]];
#pragma unroll
            for (int dgr = 0; dgr < P; ++dgr)
                if (r == coeffCount(dgr) - 1) pMine = degree == (uint32_t)dgr ? p1 : pMine, mMine = degree == (uint32_t)dgr ? m1 : mMine;
        }
        if (degree != (uint32_t)P) p1 = pMine, m1 = mMine;
        if constexpr (NODIV)
            g[k] = p1 - m1;
        else
            g[k] = (p1 - m1) / (2.0 * eps);
    }
    if constexpr (!NODIV) {
        const double z = left ? sum3<true>(g[0] * g[0], g[1] * g[1], g[2] * g[2]) : sum3<false>(g[0] * g[0], g[1] * g[1], g[2] * g[2]);  // Eigen normalize()
        if (z > 0.0) {
            const double nrm = sqrt(z);
            g[0] = g[0] / nrm, g[1] = g[1] / nrm, g[2] = g[2] / nrm;
        }
    }
    double f = 0.0, mine = 0.0;  // :972-984
#pragma unroll
    for (int r = 0; r < N; ++r) {
        double lp = L0[0][kBasis.v[r][0]];
        lp = lp * L0[1][kBasis.v[r][1]];
        lp = lp * L0[2][kBasis.v[r][2]];
        f = f + cv[r] * lp;
#pragma unroll
        for (int dgr = 0; dgr < P; ++dgr)
            if (r == coeffCount(dgr) - 1) mine = degree == (uint32_t)dgr ? f : mine;
    }
    return degree == (uint32_t)P ? f : mine;
}

// Legendre recurrence constants (2j-1)/j and (j-1)/j (Include/HP/Utility.h:112-127): IEEE divisions of small
// integers, so the compile-time values are the table's values.
__host__ __device__ constexpr double recA(int j) { return j == 0 ? 0.0 : (2.0 * j - 1.0) / j; }
__host__ __device__ constexpr double recB(int j) { return j == 0 ? 0.0 : (j - 1.0) / j; }

// Same evaluation for a leaf sitting at the top-table level: its depth is uniform, so the normalisation
// factors arrive as kernel arguments (scalars) and the recurrence constants are literals.
template <int P, int NV>
__device__ __forceinline__ double evalLeafTop(const double (&cv)[NV], double ux, double uy, double uz,
                                              const double* __restrict__ nl) {
    constexpr int N = coeffCount(P);
    double tx[P + 1], ty[P + 1], tz[P + 1];
    tx[0] = ty[0] = tz[0] = nl[0];
    double xm2 = 0.0, xm1 = 1.0, ym2 = 0.0, ym1 = 1.0, zm2 = 0.0, zm1 = 1.0;
#pragma unroll
    for (int j = 1; j <= P; ++j) {
        const double lx = recA(j) * ux * xm1 - recB(j) * xm2;
        const double ly = recA(j) * uy * ym1 - recB(j) * ym2;
        const double lz = recA(j) * uz * zm1 - recB(j) * zm2;
        xm2 = xm1, xm1 = lx, ym2 = ym1, ym1 = ly, zm2 = zm1, zm1 = lz;
        tx[j] = lx * nl[j], ty[j] = ly * nl[j], tz[j] = lz * nl[j];
    }
    double f = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double lp = tx[kBasis.v[i][0]];
        lp = lp * ty[kBasis.v[i][1]];
        lp = lp * tz[kBasis.v[i][2]];
        f = f + cv[i] * lp;
    }
    return f;
}

// c is 16-byte aligned in the device mirror (hpsdf_tree_upload pads every leaf to an even count), so
// the coefficients come in as double2.
template <int P>
__device__ __forceinline__ double evalLeafFixed(const double* __restrict__ c, double ux, double uy, double uz, int depth,
                                                const double* sNl, const double* sRec) {
    constexpr int N = coeffCount(P);
    double cv[N + 1];
    const double2* __restrict__ c2 = reinterpret_cast<const double2*>(c);
#pragma unroll
    for (int i = 0; i < (N + 1) / 2; ++i) {
        const double2 v = c2[i];
        cv[2 * i] = v.x;
        cv[2 * i + 1] = v.y;
    }
    return evalLeafVals<P>(cv, ux, uy, uz, depth, sNl, sRec);
}

// any degree (tables in private memory, dynamically indexed)
__device__ __noinline__ double evalLeafGeneric(const double* __restrict__ c, int degree, double ux, double uy, double uz,
                                               int depth, const double* sNl, const double* sRec) {
    double t[3][13];
    const double u[3] = {ux, uy, uz};
    for (int a = 0; a < 3; ++a) {
        t[a][0] = sNl[depth];
        double m2 = 0.0, m1 = 1.0;
        for (int j = 1; j <= degree; ++j) {
            const double l = sRec[2 * j] * u[a] * m1 - sRec[2 * j + 1] * m2;
            m2 = m1, m1 = l;
            t[a][j] = l * sNl[j * 11 + depth];
        }
    }
    double f = 0.0;
    const int n = coeffCount(degree);
    for (int i = 0; i < n; ++i) {
        double lp = t[0][kBasis.v[i][0]];
        lp = lp * t[1][kBasis.v[i][1]];
        lp = lp * t[2][kBasis.v[i][2]];
        f = f + c[i] * lp;
    }
    return f;
}

// MAXP: the largest leaf degree of the tree this instantiation serves (2, 3, 5 unrolled; 12 adds the
// generic path).  A smaller MAXP keeps registers (and so latency-hiding waves) for the common trees.
template <int MAXP>
__device__ __forceinline__ double evalLeaf(const double* __restrict__ c, int degree, double ux, double uy, double uz,
                                           int depth, const double* sNl, const double* sRec) {
    if constexpr (MAXP <= 2) {
        if (degree == 2) return evalLeafFixed<2>(c, ux, uy, uz, depth, sNl, sRec);
        if (degree == 1) return evalLeafFixed<1>(c, ux, uy, uz, depth, sNl, sRec);
        return evalLeafFixed<0>(c, ux, uy, uz, depth, sNl, sRec);
    } else {
        switch (degree) {
            case 0: return evalLeafFixed<0>(c, ux, uy, uz, depth, sNl, sRec);
            case 1: return evalLeafFixed<1>(c, ux, uy, uz, depth, sNl, sRec);
            case 2: return evalLeafFixed<2>(c, ux, uy, uz, depth, sNl, sRec);
            case 3: return evalLeafFixed<3>(c, ux, uy, uz, depth, sNl, sRec);
            default:
                if constexpr (MAXP >= 5) {
                    if (degree == 4) return evalLeafFixed<4>(c, ux, uy, uz, depth, sNl, sRec);
                    if (degree == 5) return evalLeafFixed<5>(c, ux, uy, uz, depth, sNl, sRec);
                }
                if constexpr (MAXP > 5) return evalLeafGeneric(c, degree, ux, uy, uz, depth, sNl, sRec);
                return 0.0;
        }
    }
}

// One point through the tree.  (x,y,z) in world coordinates.
template <int MAXP>
__device__ __forceinline__ double queryPoint(const TreeDev& t, double x, double y, double z, const double* sNl,
                                             const double* sRec) {
    // Octree.cpp:665
    const double px = (x - t.rootCentre[0]) * t.rootInvSizes[0];
    const double py = (y - t.rootCentre[1]) * t.rootInvSizes[1];
    const double pz = (z - t.rootCentre[2]) * t.rootInvSizes[2];
    // :668 containment on the f32 cast, both ends inclusive; NaN fails
    const float fx = (float)px, fy = (float)py, fz = (float)pz;
    if (!(fx >= -0.5f && fx <= 0.5f && fy >= -0.5f && fy <= 0.5f && fz >= -0.5f && fz <= 0.5f)) return DBL_MAX;
    // :674-701.  The mid-plane of a cell is its centre; centres are exact dyadics.  The levels that are
    // complete in this tree (topDepth of them; Octree::UniformlyRefine makes that 4) need no node reads:
    // the same comparisons give the path, and one table lookup gives the node reached.
    double cx = 0.0, cy = 0.0, cz = 0.0, q = 0.25;
    uint32_t ix = 0, iy = 0, iz = 0;
    int depth = 0;
    for (; depth < t.topDepth; ++depth) {
        const bool ux = px >= cx, uy = py >= cy, uz = pz >= cz;
        ix = ix * 2u + (ux ? 1u : 0u);
        iy = iy * 2u + (uy ? 1u : 0u);
        iz = iz * 2u + (uz ? 1u : 0u);
        cx = ux ? cx + q : cx - q;
        cy = uy ? cy + q : cy - q;
        cz = uz ? cz + q : cz - q;
        q = q * 0.5;
    }
    const uint32_t code = ix + ((iy + (iz << t.topDepth)) << t.topDepth);  // the table is indexed by cell (x, y, z)
    // One 128-byte line per top-level cell: the node record and, for a leaf of degree <= 2, its
    // coefficients inline -- the common case costs a single L2 line per point.  The coefficient loads
    // do not wait for the record (same line, issued together).
    const TopEntry* __restrict__ e = t.top + code;
    const uint2 hdr = *reinterpret_cast<const uint2*>(e);
    double cv[10];
    {
        const double2* __restrict__ c2 = reinterpret_cast<const double2*>(e->c);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double2 v = c2[i];
            cv[2 * i] = v.x;
            cv[2 * i + 1] = v.y;
        }
    }
    NodeRec rec{hdr.x, hdr.y};
    if (rec.b <= 2u) {  // leaf at the table level, coefficients already here
        const double s = (double)(2 << depth);
        const double ux = (px - cx) * s, uy = (py - cy) * s, uz = (pz - cz) * s;
        if (rec.b == 2u) return evalLeafVals<2>(cv, ux, uy, uz, depth, sNl, sRec);
        if (rec.b == 1u) return evalLeafVals<1>(cv, ux, uy, uz, depth, sNl, sRec);
        return evalLeafVals<0>(cv, ux, uy, uz, depth, sNl, sRec);
    }
    while (rec.b == kInteriorTag) {
        const bool ux = px >= cx, uy = py >= cy, uz = pz >= cz;
        const uint32_t idx = rec.a + (ux ? 1u : 0u) + (uy ? 2u : 0u) + (uz ? 4u : 0u);
        cx = ux ? cx + q : cx - q;
        cy = uy ? cy + q : cy - q;
        cz = uz ? cz + q : cz - q;
        q = q * 0.5;
        ++depth;
        rec = t.nodes[idx];
    }
    // :862  unitPt = (pt - centre) * (2 << depth)
    const double s = (double)(2 << depth);
    return evalLeaf<MAXP>(t.coeffs + rec.a, (int)rec.b, (px - cx) * s, (py - cy) * s, (pz - cz) * s, depth, sNl, sRec);
}

__device__ __forceinline__ void stageQueryTables(const DeviceTables* T, double* sNl, double* sRec) {
    for (int i = threadIdx.x; i < 13 * 11; i += blockDim.x) sNl[i] = (&T->nl[0][0])[i];
    for (int i = threadIdx.x; i < 26; i += blockDim.x) sRec[i] = (&T->rec[0][0])[i];
}

// Cell of the complete top level that holds p (unit-cube coordinates), per axis: index k and cell centre c.
// The comparison chain "p >= mid-plane" of Octree.cpp:674-701, level by level, selects the cell k with
// lo_k <= p < lo_k + h (h = 2^-topDepth, lo_k = -0.5 + k h, all exact dyadics; k clamped to the grid because the
// containment test ran on the f32 cast).  k is computed directly -- floor((p+0.5)/h) can be off by one when
// p + 0.5 rounds across a cell boundary, so it is corrected by the same exact comparisons the chain would make.
__device__ __forceinline__ void topCell(const double (&p3)[3], int topDepth, int (&k3)[3], double (&c3)[3]) {
    const int side = 1 << topDepth;
    const double h = 1.0 / (double)side, fside = (double)side;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int k = (int)floor((p3[a] + 0.5) * fside);
        k = k < 0 ? 0 : (k > side - 1 ? side - 1 : k);
        double lo = -0.5 + (double)k * h;
        if (p3[a] < lo && k > 0) {
            --k;
            lo = lo - h;
        } else if (p3[a] >= lo + h && k < side - 1) {
            ++k;
            lo = lo + h;
        }
        k3[a] = k;
        c3[a] = lo + 0.5 * h;
    }
}

// Batched Query, trees whose leaves ALL sit in the top table with degree <= 2 (what the BASELINE thresholds
// produce: 4096 depth-4 leaves of degree 2).  Random points share nothing, so per point the tree costs one
// 128-byte top-table line out of L2; fetched lane-by-lane that is 6 divergent 16-byte requests per point and the
// texture addresser becomes the limit.  Here the wave fetches cooperatively: in step k the 8 lanes of every
// group read the 8 consecutive 16-byte chunks of the line of the group's k-th point, straight into LDS
// (global_load_lds_dwordx4: lane-linear destination, per-lane source), i.e. 8 whole lines per
// wave-instruction instead of 64 fragments; afterwards every lane reads back its own point's row.
// Every other tree goes through query_general_kernel below.
// GRAD: QueryWithGradient (Octree.cpp:749-789, 904-985) on the same trees -- the same fetch, value and "gradient" from the row
// (evalLeafGradVals); rows of grad for points outside the root are left untouched, as the reference leaves its output argument.
template <int TOPD, bool DEDUPE, bool GRAD>
__device__ __forceinline__ void queryTopBody(const TreeDev& t, const double* __restrict__ xyz, size_t n, double* __restrict__ out,
                                             double* __restrict__ grad, const double* sNl, const double* sRec) {
    // per wave: 4 steps x 64 lanes x 16 B (two passes; less LDS = more waves).  Each step's kilobyte is followed by
    // 32 bytes of padding: a lane reads row (sub & 3), so without it the four lanes of a group hit the same banks
    // one kilobyte apart (measured: 70 % of the LDS cycles were bank conflicts).
    __shared__ double2 sRows[4][4][66];
    __shared__ uint32_t sRunCell[4][64];  // ordered input: the cell of every run of equal cells in the wave (below)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane & ~7, sub = lane & 7;
    for (size_t base = (size_t)blockIdx.x * 256; base < n; base += (size_t)gridDim.x * 256) {
        const size_t i = base + threadIdx.x;
        const bool valid = i < n;
        const size_t il = valid ? i : n - 1;
        // (plain loads: non-temporal ones, 3 x 8 bytes or 16 + 8, cost the kernel 30 us -- neighbouring lanes share the points' lines, and
        // without the caches every lane fetches them for itself; the RESULTS leave non-temporally: below)
        const double x = xyz[3 * il], y = xyz[3 * il + 1], z = xyz[3 * il + 2];
        // Octree.cpp:665
        const double p3[3] = {(x - t.rootCentre[0]) * t.rootInvSizes[0], (y - t.rootCentre[1]) * t.rootInvSizes[1],
                              (z - t.rootCentre[2]) * t.rootInvSizes[2]};
        // :668 containment on the f32 cast, both ends inclusive; NaN fails
        const float fx = (float)p3[0], fy = (float)p3[1], fz = (float)p3[2];
        const bool inside = fx >= -0.5f && fx <= 0.5f && fy >= -0.5f && fy <= 0.5f && fz >= -0.5f && fz <= 0.5f;
        const int topDepth = TOPD > 0 ? TOPD : t.topDepth;
        int k3[3];
        double c3[3];
        topCell(p3, topDepth, k3, c3);
        uint32_t code = (uint32_t)(k3[0] + ((k3[1] + (k3[2] << topDepth)) << topDepth));
        if (!inside) code = 0;  // any valid line; the result is DBL_MAX
        // lane (group g, sub k) owns the row that step k writes at lanes 8g..8g+7: [record][c0 c1]..[c8 c9].
        // Two passes of four steps through a 4 KB per-wave window (measured: 110 vs 121 us for one 8 KB pass).
        // Points of a group that fall into the same cell share one fetch: step k runs for a group only if its k-th
        // point is the first of the group in its cell, and every lane reads the row of the first point of its own cell
        // (coherent point sets -- grids, slices, rays -- move a fraction of the lines; random points lose nothing but a
        // few compares: the kernel is bound by the L2 -> CU line traffic).
        uint2 hdr = make_uint2(0u, 0u);
        double cv[10];
        // ORDERED INPUT (round 6).  Grids, slices, rays and cell-sorted point sets arrive as RUNS of consecutive points in one cell: a
        // lane opens a run if its cell differs from the lane's before it.  With at most 32 runs in the wave -- random points have 64 --
        // the wave needs one row per RUN, not one per point or per group of eight: the lanes that open a run publish its cell, step s
        // fetches the rows of runs 8s .. 8s + 7 (eight lanes a row, as ever), and every lane reads back the row of its own run; four steps at
        // most, one for a wave that crosses a handful of cells, and no second pass.  A wave that lies in ONE cell altogether -- the rule in
        // cell-sorted sets, 2 441 points a cell at 10 M points -- asks for its row through the scalar cache and evaluates from SGPRs: no
        // LDS, no vector memory at all.  The rows are the same bytes whichever way they come, so the values are too
        // (test_query_ordered_point_sets_bitwise).  What the group-wise dedupe of rounds 1-5 cost such sets was its bookkeeping: eight
        // shuffles and ~60 compares a tile on a kernel whose arithmetic is 228 vector instructions a tile and which, on ordered input, is
        // bound by exactly that.
        bool viaRuns = false;
        if (DEDUPE) {
            const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)code, (int)code, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const bool opens = lane == 0 || before != code;
            const unsigned long long heads = __ballot(opens);
            const int nRuns = __popcll(heads);  // wave-uniform
            if (nRuns == 1) {
                const uint32_t cell = (uint32_t)__builtin_amdgcn_readfirstlane((int)code);
                typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
                const ConstWords w = (ConstWords)(uintptr_t)(t.top + cell);
                hdr = make_uint2(w[0], w[1]);
#pragma unroll
                for (int c = 0; c < 10; ++c) cv[c] = __longlong_as_double((long long)(((unsigned long long)w[5 + 2 * c] << 32) | (unsigned long long)w[4 + 2 * c]));
                viaRuns = true;
            } else if (nRuns <= 32) {
                // (the wave's row of sRunCell from an index the compiler cannot see through: hoisted out of the tile loop, that address is
                // one register more than the kernel's seven waves a SIMD leave it, and it went to scratch)
                uint32_t waveV = (uint32_t)threadIdx.x >> 6;
                asm volatile("" : "+v"(waveV));
                uint32_t* runCell = sRunCell[waveV];
                const uint32_t myRun = __builtin_amdgcn_mbcnt_hi((uint32_t)(heads >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)heads, 0u)) + (opens ? 1u : 0u) - 1u;
                if (opens) runCell[myRun] = code;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int nSteps = (nRuns + 7) >> 3;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < nSteps) {
                        const int run = 8 * k + (lane >> 3);
                        if (run < nRuns && sub < 6) {  // bytes 96..127 of an entry are padding
                            const char* src = reinterpret_cast<const char*>(t.top + runCell[run]) + sub * 16;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                             (__attribute__((address_space(3))) void*)&sRows[wave][k][0], 16, 0, 0);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                {
                    const double2* row = &sRows[wave][myRun >> 3][(myRun & 7u) * 8u];
                    hdr = *reinterpret_cast<const uint2*>(row);
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        const double2 v = row[1 + c];
                        cv[2 * c] = v.x;
                        cv[2 * c + 1] = v.y;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();  // the window is rewritten by the next tile
                viaRuns = true;
            }
        }
        if (!viaRuns) {
        uint32_t ck[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ck[k] = __shfl(code, grp | k, 64);
        int firstOfMine = sub;  // first point of the group in this lane's cell
        uint32_t needMask = 0xFFu;  // bit k: point k is the first of its cell in the group
        // wave-uniform gate: on random points (almost) no wave has two neighbouring lanes in one cell and the
        // bookkeeping below is skipped; on point sets with SOME order (more than 32 runs, yet neighbours that share cells) it pays
        if (DEDUPE && __any(__shfl_xor(code, 1, 64) == code)) {
#pragma unroll
            for (int k = 7; k >= 0; --k) firstOfMine = ck[k] == code ? k : firstOfMine;
            needMask = 1u;
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                bool seen = false;
#pragma unroll
                for (int j = 0; j < k; ++j) seen = seen || (ck[j] == ck[k]);
                needMask |= seen ? 0u : (1u << k);
            }
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const char* src = reinterpret_cast<const char*>(t.top + ck[pass * 4 + k]) + sub * 16;
                if (sub < 6 && ((needMask >> (pass * 4 + k)) & 1u))  // bytes 96..127 of an entry are padding
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&sRows[wave][k][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if ((firstOfMine >> 2) == pass) {
                const double2* row = &sRows[wave][firstOfMine & 3][grp];
                hdr = *reinterpret_cast<const uint2*>(row);
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const double2 v = row[1 + c];
                    cv[2 * c] = v.x;
                    cv[2 * c + 1] = v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();  // the window is rewritten by the next pass / tile
        }
        }
        double r = DBL_MAX;  // :668-671 outside the root
        if constexpr (GRAD) {
            double g[3] = {0.0, 0.0, 0.0};
            if (inside) {
                const double s = (double)(2 << topDepth);  // :862
                const double u[3] = {(p3[0] - c3[0]) * s, (p3[1] - c3[1]) * s, (p3[2] - c3[2]) * s};
                if (hdr.y == 2u)
                    r = evalLeafGradVals<2>(cv, u, topDepth, sNl, sRec, g, t.leftAssoc);
                else if (hdr.y == 1u)
                    r = evalLeafGradVals<1>(cv, u, topDepth, sNl, sRec, g, t.leftAssoc);
                else
                    r = evalLeafGradVals<0>(cv, u, topDepth, sNl, sRec, g, t.leftAssoc);
            }
            if (valid) {
                __builtin_nontemporal_store(r, &out[i]);
                if (inside) {
                    __builtin_nontemporal_store(g[0], &grad[3 * i]);
                    __builtin_nontemporal_store(g[1], &grad[3 * i + 1]);
                    __builtin_nontemporal_store(g[2], &grad[3 * i + 2]);
                }
            }
        } else {
            if (inside) {
                // :862  unitPt = (pt - centre) * (2 << depth)
                const double s = (double)(2 << topDepth);
                const double ux = (p3[0] - c3[0]) * s, uy = (p3[1] - c3[1]) * s, uz = (p3[2] - c3[2]) * s;
                // (the normalisation factors through a barrier the optimiser cannot see through: it hoists nl[0]^2 and nl[0]^3 out of the
                // tile loop otherwise, into registers that the kernel's seven waves a SIMD do not have -- they went to scratch and came
                // back once a tile; one multiply a tile is cheaper than one memory operation)
                double nlT[3] = {t.nlTop[0], t.nlTop[1], t.nlTop[2]};
                asm volatile("" : "+s"(nlT[0]));
                if (hdr.y == 2u)
                    r = evalLeafTop<2>(cv, ux, uy, uz, nlT);
                else if (hdr.y == 1u)
                    r = evalLeafTop<1>(cv, ux, uy, uz, nlT);
                else
                    r = evalLeafTop<0>(cv, ux, uy, uz, nlT);
            }
            if (valid) __builtin_nontemporal_store(r, &out[i]);
        }
    }
}

template <int TOPD, bool DEDUPE>
__global__ __launch_bounds__(256, 7) void query_kernel(TreeDev t, const double* __restrict__ xyz, size_t n,
                                                    double* __restrict__ out) {
    queryTopBody<TOPD, DEDUPE, false>(t, xyz, n, out, nullptr, nullptr, nullptr);
}

// QueryWithGradient on the trees query_kernel serves (every leaf in the top table, degree <= 2): one line a point like Query, where
// the any-tree kernel (query_general_grad_kernel) pays a record lookup, a walk and a second round trip.
template <int TOPD>
__global__ __launch_bounds__(256, 4) void query_grad_kernel(TreeDev t, const DeviceTables* __restrict__ T, const double* __restrict__ xyz,
                                                            size_t n, double* __restrict__ out, double* __restrict__ grad) {
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    stageQueryTables(T, sNl, sRec);
    __syncthreads();
    queryTopBody<TOPD, true, true>(t, xyz, n, out, grad, sNl, sRec);
}

// 16-byte chunks a leaf of degree d occupies in the device mirror (blocks are 128-byte aligned there)
__device__ __forceinline__ uint32_t leafChunks(uint32_t degree) { return ((uint32_t)coeffCount((int)degree) + 1u) >> 1; }

// Batched Query, any tree.  Per lane: the top cell by arithmetic, its 8-byte record from the thin top table
// (32 KB at depth 4: L1/L2 traffic only), then the walk down to the leaf (Octree.cpp:674-701).  The leaves'
// coefficients are then fetched by the wave as a whole, as in query_kernel, straight from the coefficient mirror
// (every leaf block starts on a 128-byte line there): in step k the 8 lanes of a group fetch chunks 0..7 of the
// leaf of the group's k-th point -- a whole degree-2 leaf, the first line of a degree-3 one -- and one extra step
// per pass brings chunks 8..9 of four degree-3 leaves at a time (two lanes each).  Two passes of 4 + 1 steps
// through a 5 KB per-wave window, i.e. two L2 round trips per tile whatever the mix of degrees <= 3 (one pass of
// 8 + 2 steps through a 10 KB window measured 3 % slower: fewer workgroups per CU).
// Leaves of degree > 3 are rare at the thresholds in use (a few dozen among thousands): DEFER appends their points to
// the workgroup's own run of deferIdx (an LDS counter, no global atomic -- one global atomic per wave serialised
// the first version of this at 2 ms per 10 M points) and query_deep_kernel finishes them lane by lane; keeping
// that code out of this kernel's loop keeps the loop at ~100 VGPRs (the 16-wave kernel has room up to 128, and finishes degrees 4-5
// itself behind its last tile: DEEP below).
// GRAD: QueryWithGradient (Octree.cpp:749-789) -- the same walk and fetch, value and "gradient" evaluated together;
// rows of grad for points outside the root are left untouched, as the reference leaves its output argument.
// WAVES: waves per workgroup (tile = 64 WAVES points).  LDSTOP (depth-4 top level only): the thin top table, 32 KB,
// sits in LDS -- the kernel runs at 4 waves per SIMD for its registers anyway, so one 16-wave workgroup per CU with
// ~120 KB of LDS costs no occupancy and takes the record lookup (an L2 round trip and 64 scattered 8-byte requests
// per wave) off the dependent chain.
constexpr size_t queryGeneralLdsBytes(int waves, bool ldsTop) {
    return (size_t)waves * 5 * 66 * sizeof(double2) + (size_t)waves * 64 * sizeof(uint32_t) + (ldsTop ? 4096 * sizeof(NodeRec) : 0);
}
// LAB (tools/query_general_floor.py, HPSDF_QUERY_LAB=n; results are NOT the tree's values): what the kernel's time is made of, by taking
// one link of its chain out at a time -- 1: no polynomial (the fetched rows are touched, not evaluated); 2: every lane fetches the leaf
// of its wave's first lane (the same instructions, but every line after the first is a hit: no gather traffic); 3: no second line for
// degree-3 leaves; 4: no walk below the top table (the top record is taken for the leaf)
// DEEP > 0 (values only, trees whose degrees stop at DEEP): the workgroup finishes its own deferred points behind its last tile, one lane
// each with queryPoint<DEEP> -- the lists are per workgroup, so nothing has to be scanned or waited for across workgroups, and the scan
// launch and the second pass (with the two kernel boundaries they bring: ~15 us behind a 176 us kernel for 10 M points on union3 @ 1e-7)
// are not launched at all.
template <int TOPD, bool DEFER, bool GRAD, int WAVES, bool LDSTOP, int LAB = 0, int DEEP = 0>
__device__ __forceinline__ void queryGeneralBody(const TreeDev& t, const DeviceTables* __restrict__ T,
                                                 const double* __restrict__ xyz, size_t n, double* __restrict__ out,
                                                 double* __restrict__ grad, uint32_t tilesPerWg,
                                                 uint32_t* __restrict__ deferCount, uint32_t* __restrict__ deferIdx) {
    static_assert(!LDSTOP || TOPD == 4, "the LDS copy of the thin table is sized for depth 4");
    constexpr int TILE = WAVES * 64;
    extern __shared__ double2 sDyn[];
    double2(*sRows)[5][66] = reinterpret_cast<double2(*)[5][66]>(sDyn);           // [WAVES][5][66]
    uint32_t(*sInfo)[64] = reinterpret_cast<uint32_t(*)[64]>(sDyn + WAVES * 5 * 66);  // [WAVES][64]
    NodeRec* sTop = reinterpret_cast<NodeRec*>(&sInfo[WAVES][0]);                  // [4096] when LDSTOP
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    __shared__ uint32_t sDeferred;
    stageQueryTables(T, sNl, sRec);
    if constexpr (LDSTOP) {
        for (int q = threadIdx.x; q < 4096; q += TILE) sTop[q] = t.topRec[q];
    }
    if (threadIdx.x == 0) sDeferred = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane & ~7, sub = lane & 7;
    const size_t segStart = (size_t)blockIdx.x * tilesPerWg * TILE;  // this workgroup's run of deferIdx
    // The points of a tile are asked for while the previous tile still waits for its last coefficient fetch (below): a wave's tile is a
    // chain of dependent round trips -- points from HBM (~1.8 us under load), record, walk, two coefficient fetches from L2 (~0.8 us
    // each), ~0.5 us of arithmetic -- and with 16 waves on a CU the kernel runs at what that chain's length allows, not at a bandwidth
    // limit (profiles/r02w_query_general_counters.txt: the texture addresser is busy half of the time).  The first link now overlaps
    // the previous tile's last fetch and its evaluation.
    typedef double qg_double2 __attribute__((ext_vector_type(2)));
    qg_double2 nxy = {0.0, 0.0};
    double nz = 0.0;
    {
        const size_t i0 = (size_t)blockIdx.x * TILE + threadIdx.x;
        if ((size_t)blockIdx.x * TILE < n) {
            const size_t il0 = i0 < n ? i0 : n - 1;
            nxy.x = xyz[3 * il0], nxy.y = xyz[3 * il0 + 1], nz = xyz[3 * il0 + 2];
        }
    }
    for (size_t base = (size_t)blockIdx.x * TILE; base < n; base += (size_t)gridDim.x * TILE) {
        const size_t i = base + threadIdx.x;
        const bool valid = i < n;
        const double x = nxy.x, y = nxy.y, z = nz;
        const size_t nextBase = base + (size_t)gridDim.x * TILE;
        const bool more = nextBase < n;  // workgroup-uniform
        const double p3[3] = {(x - t.rootCentre[0]) * t.rootInvSizes[0], (y - t.rootCentre[1]) * t.rootInvSizes[1],
                              (z - t.rootCentre[2]) * t.rootInvSizes[2]};  // Octree.cpp:665
        const float fx = (float)p3[0], fy = (float)p3[1], fz = (float)p3[2];
        const bool inside = fx >= -0.5f && fx <= 0.5f && fy >= -0.5f && fy <= 0.5f && fz >= -0.5f && fz <= 0.5f;  // :668
        const int topDepth = TOPD > 0 ? TOPD : t.topDepth;
        int k3[3];
        double c3[3];
        topCell(p3, topDepth, k3, c3);
        uint32_t code = (uint32_t)(k3[0] + ((k3[1] + (k3[2] << topDepth)) << topDepth));
        if (!inside) code = 0;
        NodeRec rec = LDSTOP ? sTop[code] : t.topRec[code];
        int depth = topDepth;
        double q = 0.25 / (double)(1 << topDepth);  // a quarter of the cell size: from a centre to its children's
        if constexpr (LAB == 4) {
            if (rec.b == kInteriorTag) rec.a = 0, rec.b = 2;
        }
        while (rec.b == kInteriorTag) {              // :674-701 below the complete levels
            const bool ux = p3[0] >= c3[0], uy = p3[1] >= c3[1], uz = p3[2] >= c3[2];
            const uint32_t idx = rec.a + (ux ? 1u : 0u) + (uy ? 2u : 0u) + (uz ? 4u : 0u);
            c3[0] = ux ? c3[0] + q : c3[0] - q;
            c3[1] = uy ? c3[1] + q : c3[1] - q;
            c3[2] = uz ? c3[2] + q : c3[2] - q;
            q = q * 0.5;
            ++depth;
            rec = t.nodes[idx];
        }
        const uint32_t degree = rec.b;
        const bool coop = inside && degree <= 3u;
        // :862 / :907  unitPt = (pt - centre) * (2 << depth) -- formed now, so that the point and the centre need no
        // registers across the fetch
        const double sc = (double)(2 << depth);
        const double u[3] = {(p3[0] - c3[0]) * sc, (p3[1] - c3[1]) * sc, (p3[2] - c3[2]) * sc};
        // what the fetching lanes need to know about this lane's leaf: block offset (a multiple of 16 doubles, so its
        // low four bits are free) and chunk count (0 = nothing to fetch)
        const uint32_t infoMine = rec.a | (coop ? leafChunks(degree) : 0u);
        double cv[20];
        // ORDERED INPUT (round 6; as in queryTopBody).  Grids, slices, rays and sorted point sets reach this kernel as RUNS of consecutive
        // points in one leaf.  With at most 32 runs in the wave (random points: 64) the wave fetches one block per RUN in ONE pass -- step s
        // the first lines of runs 8s .. 8s + 7, one more step the second lines of up to 32 degree-3 leaves, two lanes each -- and every
        // lane reads back its run's block: one round trip to L2 instead of two on this kernel's chain of dependent round trips, and a
        // quarter of the window's traffic.  The same bytes whichever way they come (test_query_ordered_point_sets_bitwise).
        bool viaRuns = false;
        if constexpr (LAB == 0) {
            const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)infoMine, (int)infoMine, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
            const bool opens = lane == 0 || before != infoMine;
            const unsigned long long heads = __ballot(opens);
            const int nRuns = __popcll(heads);  // wave-uniform
            if (nRuns <= 32) {
                viaRuns = true;
                const uint32_t myRun = __builtin_amdgcn_mbcnt_hi((uint32_t)(heads >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)heads, 0u)) + (opens ? 1u : 0u) - 1u;
                if (opens) sInfo[wave][myRun] = infoMine;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int nSteps = (nRuns + 7) >> 3;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < nSteps) {
                        const int run = 8 * k + (lane >> 3);
                        const uint32_t inf = run < nRuns ? sInfo[wave][run] : 0u;
                        const char* src = reinterpret_cast<const char*>(t.coeffs) + (size_t)(inf & ~15u) * 8u + (uint32_t)sub * 16u;
                        if ((uint32_t)sub < (inf & 15u))
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                             (__attribute__((address_space(3))) void*)&sRows[wave][k][0], 16, 0, 0);
                    }
                }
                if (__any(coop && degree == 3u)) {  // chunks 8..9 of the runs' degree-3 leaves: two lanes a run
                    const int run = lane >> 1;
                    const uint32_t inf = run < nRuns ? sInfo[wave][run] : 0u;
                    const char* src = reinterpret_cast<const char*>(t.coeffs) + (size_t)(inf & ~15u) * 8u + (8u + (uint32_t)(lane & 1)) * 16u;
                    if ((inf & 15u) > 8u)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                         (__attribute__((address_space(3))) void*)&sRows[wave][4][0], 16, 0, 0);
                }
                if (more) {  // the next tile's points behind the fetch, as below
                    const size_t in = nextBase + threadIdx.x;
                    const double* np = xyz + 3 * (in < n ? in : n - 1);
                    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx2 %1, %2, off offset:16\n\ts_waitcnt vmcnt(2)"
                                 : "=&v"(nxy), "=&v"(nz)
                                 : "v"(np)
                                 : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_wave_barrier();
                {
                    const double2* row = &sRows[wave][myRun >> 3][(myRun & 7u) * 8u];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const double2 v = row[c];
                        cv[2 * c] = v.x;
                        cv[2 * c + 1] = v.y;
                    }
                    const double2* tail = &sRows[wave][4][2u * myRun];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const double2 v = tail[c];
                        cv[16 + 2 * c] = v.x;
                        cv[16 + 2 * c + 1] = v.y;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();  // the window is rewritten by the next tile
            }
        }
        if (!viaRuns) {
        sInfo[wave][lane] = infoMine;
        __builtin_amdgcn_wave_barrier();
        if constexpr (LAB == 2) {
            const uint32_t first = sInfo[wave][0];
            __builtin_amdgcn_wave_barrier();
            sInfo[wave][lane] = first;
            __builtin_amdgcn_wave_barrier();
        }
        const bool second = LAB != 3 && __any(coop && degree == 3u);  // wave-uniform: somebody needs chunks 8..9
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // chunks 0..7 of the leaf of point (group, 4 pass + k)
                const uint32_t inf = sInfo[wave][grp + pass * 4 + k];
                const char* src = reinterpret_cast<const char*>(t.coeffs) + (size_t)(inf & ~15u) * 8u + (uint32_t)sub * 16u;
                if ((uint32_t)sub < (inf & 15u))
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&sRows[wave][k][0], 16, 0, 0);
            }
            if (second) {  // chunks 8..9 of the leaves of points (group, 4 pass + 0..3): two lanes each
                const uint32_t inf = sInfo[wave][grp + 4 * pass + (sub >> 1)];
                const char* src = reinterpret_cast<const char*>(t.coeffs) + (size_t)(inf & ~15u) * 8u + (8u + (uint32_t)(sub & 1)) * 16u;
                if ((inf & 15u) > 8u)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)&sRows[wave][4][0], 16, 0, 0);
            }
            if (pass == 1 && more) {
                // the next tile's points, issued BEHIND this tile's last coefficient fetch: memory operations of a wave complete in
                // order, so "all but the two youngest" (vmcnt(2)) is exactly "the coefficients are in LDS", and the two point loads
                // stay in flight across the read-back and the evaluation (written out as instructions: the count in the wait must be
                // the number of loads, which the compiler is free to merge or split; the values are claimed further down, before the
                // stores, by a wait of their own).  Plain loads: until the end of round 5 these carried `nt`, which made every lane fetch the
                // lines its neighbours share for itself -- 193 -> 168 us on union3 @ 1e-7 without it.  The results do leave non-temporally.
                const size_t in = nextBase + threadIdx.x;
                const double* np = xyz + 3 * (in < n ? in : n - 1);
                asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx2 %1, %2, off offset:16\n\ts_waitcnt vmcnt(2)"
                             : "=&v"(nxy), "=&v"(nz)
                             : "v"(np)
                             : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_wave_barrier();
            if ((sub >> 2) == pass) {
                const double2* row = &sRows[wave][sub & 3][grp];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const double2 v = row[c];
                    cv[2 * c] = v.x;
                    cv[2 * c + 1] = v.y;
                }
                const double2* tail = &sRows[wave][4][grp + 2 * (sub & 3)];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double2 v = tail[c];
                    cv[16 + 2 * c] = v.x;
                    cv[16 + 2 * c + 1] = v.y;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();  // the window is rewritten by the next pass / tile
        }
        }
        double r = DBL_MAX;  // :668-671
        double g[3] = {0.0, 0.0, 0.0};
        bool defer = false;
        if (LAB == 1 && inside) {
            r = (cv[0] + cv[9]) + (cv[16] + cv[19]) + u[0] * u[1] + u[2] + (double)depth;
        } else if (inside) {
            if constexpr (GRAD) {
                if (degree > 3u)
                    defer = valid;
                else if constexpr (LAB == 5)  // (lab: the gradient kernel with the value's arithmetic only -- what the six one-sided sums, the divisions and the square root cost)
                    r = evalLeafValsMixed<3>(cv, u[0], u[1], u[2], depth, degree, sNl, sRec), g[0] = r, g[1] = u[1], g[2] = u[2];
                else
                    r = evalLeafGradValsMixed<3, 20, LAB == 6>(cv, u, depth, degree, sNl, sRec, g, t.leftAssoc);
            } else {
                // one pass for the wave's mix of degrees (evalLeafValsMixed)
                if (degree > 3u)
                    defer = valid;
                else
                    r = evalLeafValsMixed<3>(cv, u[0], u[1], u[2], depth, degree, sNl, sRec);
            }
        }
        // the next tile's points have had the evaluation's time to arrive: claim them before this tile's stores are issued (a wait
        // behind the stores would wait for those too)
        if (more) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxy), "+v"(nz)::"memory");
        if constexpr (DEFER) {
            const unsigned long long dmask = __ballot(defer);
            if (dmask) {  // one LDS atomic per wave reserves the slots
                uint32_t slot = 0;
                const int leader = __ffsll((long long)dmask) - 1;
                if (lane == leader) slot = atomicAdd(&sDeferred, (uint32_t)__popcll(dmask));
                slot = __shfl(slot, leader, 64);
                if (defer) deferIdx[segStart + slot + (uint32_t)__popcll(dmask & ((1ull << lane) - 1ull))] = (uint32_t)i;
            }
        }
        if (valid && !defer) {
            __builtin_nontemporal_store(r, &out[i]);
            if constexpr (GRAD) {
                if (inside) {
                    __builtin_nontemporal_store(g[0], &grad[3 * i]);
                    __builtin_nontemporal_store(g[1], &grad[3 * i + 1]);
                    __builtin_nontemporal_store(g[2], &grad[3 * i + 2]);
                }
            }
        }
    }
    if constexpr (DEFER && DEEP > 0) {
        static_assert(!GRAD, "the fused second pass writes values only");
        __syncthreads();  // every wave's indices are written (and visible inside the workgroup), the count is final
        const uint32_t nd = sDeferred;
        for (uint32_t j = threadIdx.x; j < nd; j += TILE) {
            const size_t i = deferIdx[segStart + j];
            __builtin_nontemporal_store(queryPoint<DEEP>(t, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], sNl, sRec), &out[i]);
        }
    } else if constexpr (DEFER) {
        __syncthreads();
        if (threadIdx.x == 0) deferCount[blockIdx.x] = sDeferred;
    }
}

// Values only: ~104 VGPRs, 4 waves per SIMD (forcing 5 spills: measured 223 -> 338 us on union3@1e-7); with the
// gradient ~140 VGPRs, 3 waves.  (Sharing one fetch between the points of a group that land in the same leaf, as
// query_kernel does, measured slower here on grids and sorted points alike: this kernel is latency-, not traffic-bound.
// Evaluating every lane of a mixed wave with the degree-3 code over zero-padded rows -- one pass instead of the
// degree-2 and degree-3 passes -- is bit-identical but needs whole lines for degree-2 leaves: 229 -> 247 us.)
template <int TOPD, bool DEFER>
__global__ __launch_bounds__(256) void query_general_kernel(TreeDev t, const DeviceTables* __restrict__ T,
                                                               const double* __restrict__ xyz, size_t n,
                                                               double* __restrict__ out, uint32_t tilesPerWg,
                                                               uint32_t* __restrict__ deferCount,
                                                               uint32_t* __restrict__ deferIdx) {
    queryGeneralBody<TOPD, DEFER, false, 4, false>(t, T, xyz, n, out, nullptr, tilesPerWg, deferCount, deferIdx);
}
// depth-4 top level: 16 waves per workgroup, thin table in LDS
template <bool DEFER, int LAB = 0, int DEEP = 0>
__global__ __launch_bounds__(1024) void query_general_lds_kernel(TreeDev t, const DeviceTables* __restrict__ T,
                                                                 const double* __restrict__ xyz, size_t n,
                                                                 double* __restrict__ out, uint32_t tilesPerWg,
                                                                 uint32_t* __restrict__ deferCount,
                                                                 uint32_t* __restrict__ deferIdx) {
    queryGeneralBody<4, DEFER, false, 16, true, LAB, DEEP>(t, T, xyz, n, out, nullptr, tilesPerWg, deferCount, deferIdx);
}
template <int TOPD, bool DEFER, int LAB = 0>
__global__ __launch_bounds__(256, 3) void query_general_grad_kernel(TreeDev t, const DeviceTables* __restrict__ T,
                                                                    const double* __restrict__ xyz, size_t n,
                                                                    double* __restrict__ out, double* __restrict__ grad,
                                                                    uint32_t tilesPerWg, uint32_t* __restrict__ deferCount,
                                                                    uint32_t* __restrict__ deferIdx) {
    queryGeneralBody<TOPD, DEFER, true, 4, false, LAB>(t, T, xyz, n, out, grad, tilesPerWg, deferCount, deferIdx);
}

// Exclusive scan of the per-workgroup deferred counts (nWg <= 8192): offsets[b] = points deferred by workgroups
// < b, offsets[nWg] = their total.  One workgroup of 1024 threads, eight counts each.
__global__ __launch_bounds__(1024) void defer_scan_kernel(const uint32_t* __restrict__ counts, uint32_t nWg,
                                                          uint32_t* __restrict__ offsets) {
    __shared__ uint32_t sSum[1024];
    const uint32_t tid = threadIdx.x;
    uint32_t local[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t b = tid * 8 + k;
        local[k] = b < nWg ? counts[b] : 0u;
        sum += local[k];
    }
    sSum[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan
        const uint32_t v = tid >= d ? sSum[tid - d] : 0u;
        __syncthreads();
        sSum[tid] += v;
        __syncthreads();
    }
    uint32_t run = sSum[tid] - sum;  // exclusive prefix of this thread's eight
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t b = tid * 8 + k;
        if (b < nWg) offsets[b] = run;
        run += local[k];
    }
    if (tid == 1023) offsets[nWg] = sSum[1023];
}

// j-th deferred point overall -> its index: the workgroup whose run holds it (binary search in the scanned counts),
// then the slot inside that run.
__device__ __forceinline__ size_t deferredPoint(uint32_t j, const uint32_t* __restrict__ offsets, uint32_t nWg,
                                                uint32_t tilesPerWg, const uint32_t* __restrict__ deferIdx) {
    uint32_t lo = 0, hi = nWg;  // offsets[lo] <= j < offsets[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= j)
            lo = mid;
        else
            hi = mid;
    }
    return deferIdx[(size_t)lo * tilesPerWg * 256 + (j - offsets[lo])];
}

// Second pass of Query for the points query_general_kernel deferred (leaves of degree > 3), one lane each, dense
// over the concatenation of the per-workgroup lists.
template <int MAXP>
__global__ __launch_bounds__(256) void query_deep_kernel(TreeDev t, const DeviceTables* __restrict__ T,
                                                         const double* __restrict__ xyz, double* __restrict__ out,
                                                         uint32_t tilesPerWg, uint32_t nWg,
                                                         const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ deferIdx) {
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    const uint32_t total = offsets[nWg];
    if ((uint32_t)(blockIdx.x * blockDim.x) >= total) return;  // workgroup-uniform
    stageQueryTables(T, sNl, sRec);
    __syncthreads();
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const size_t i = deferredPoint(j, offsets, nWg, tilesPerWg, deferIdx);
        out[i] = queryPoint<MAXP>(t, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], sNl, sRec);
    }
}

// Query for a handful of points (what Octree::Query(pt) sends: n = 1): one lane per point, any tree, ONE launch -- the
// batched kernels' tile machinery and, for trees with leaves of degree > 3, their scan + second pass are three more
// launches that a scalar call would pay for nothing.  Same queryPoint as the deferred pass: same values.
template <int MAXP>
__global__ __launch_bounds__(64) void query_few_kernel(TreeDev t, const DeviceTables* __restrict__ T, const double* __restrict__ xyz,
                                                       uint32_t n, double* __restrict__ out) {
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    stageQueryTables(T, sNl, sRec);
    __syncthreads();
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) out[i] = queryPoint<MAXP>(t, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], sNl, sRec);
}

// Octree::QueryWithGradient + FApproxWithGradient (Octree.cpp:749-789, 904-985) for point i, any degree, one lane
__device__ void queryPointWithGradient(const TreeDev& t, size_t i, const double* __restrict__ xyz, double* __restrict__ out,
                                       double* __restrict__ grad, const double* sNl, const double* sRec) {
    const double p[3] = {(xyz[3 * i] - t.rootCentre[0]) * t.rootInvSizes[0],
                         (xyz[3 * i + 1] - t.rootCentre[1]) * t.rootInvSizes[1],
                         (xyz[3 * i + 2] - t.rootCentre[2]) * t.rootInvSizes[2]};
    const float fx = (float)p[0], fy = (float)p[1], fz = (float)p[2];
    if (!(fx >= -0.5f && fx <= 0.5f && fy >= -0.5f && fy <= 0.5f && fz >= -0.5f && fz <= 0.5f)) {
        out[i] = DBL_MAX;
        return;
    }
    double c[3] = {0.0, 0.0, 0.0}, q = 0.25;
    int depth = 0;
    NodeRec rec = t.nodes[0];
    while (rec.b == kInteriorTag) {
        uint32_t idx = rec.a;
        for (int a = 0; a < 3; ++a) {
            const bool up = p[a] >= c[a];
            idx += up ? (1u << a) : 0u;
            c[a] = up ? c[a] + q : c[a] - q;
        }
        q = q * 0.5;
        ++depth;
        rec = t.nodes[idx];
    }
    const int degree = (int)rec.b;
    const double* __restrict__ co = t.coeffs + rec.a;
    const double eps = 0.0001, s = (double)(2 << depth);
    double L[13][3][3];
    for (int a = 0; a < 3; ++a) {
        const double u = (p[a] - c[a]) * s;  // :907
        L[0][a][0] = L[0][a][1] = L[0][a][2] = sNl[depth];
        double a2 = 0.0, a1 = 1.0, b2 = 0.0, b1 = 1.0, c2 = 0.0, c1 = 1.0;
        for (int j = 1; j <= degree; ++j) {
            const double r0 = sRec[2 * j], r1 = sRec[2 * j + 1], nl = sNl[j * 11 + depth];
            const double a0 = r0 * u * a1 - r1 * a2;          // :937
            const double b0 = r0 * (u + eps) * b1 - r1 * b2;  // :941
            const double c0 = r0 * (u - eps) * c1 - r1 * c2;  // :945
            a2 = a1, a1 = a0, b2 = b1, b1 = b0, c2 = c1, c1 = c0;
            L[j][a][0] = a0 * nl, L[j][a][1] = b0 * nl, L[j][a][2] = c0 * nl;
        }
    }
    const int nc = coeffCount(degree);
    double g[3];
    for (int k = 0; k < 3; ++k) {  // :956-968
        double p1 = 0.0, m1 = 0.0;
        for (int r = 0; r < nc; ++r) {
            p1 = p1 + co[r] * L[kBasis.v[r][k]][k][1];
            m1 = m1 + co[r] * L[kBasis.v[r][k]][k][2];
        }
        g[k] = (p1 - m1) / (2.0 * eps);
    }
    const double z = t.leftAssoc ? sum3<true>(g[0] * g[0], g[1] * g[1], g[2] * g[2]) : sum3<false>(g[0] * g[0], g[1] * g[1], g[2] * g[2]);  // Eigen normalize()
    if (z > 0.0) {
        const double nrm = sqrt(z);
        g[0] = g[0] / nrm, g[1] = g[1] / nrm, g[2] = g[2] / nrm;
    }
    double f = 0.0;  // :972-984
    for (int r = 0; r < nc; ++r) {
        double lp = L[kBasis.v[r][0]][0][0];
        lp = lp * L[kBasis.v[r][1]][1][0];
        lp = lp * L[kBasis.v[r][2]][2][0];
        f = f + co[r] * lp;
    }
    out[i] = f;
    grad[3 * i] = g[0], grad[3 * i + 1] = g[1], grad[3 * i + 2] = g[2];
}

// QueryWithGradient for a handful of points in one launch (the scalar call of the drop-in), like query_few_kernel
__global__ __launch_bounds__(64) void query_grad_few_kernel(TreeDev t, const DeviceTables* __restrict__ T, const double* __restrict__ xyz,
                                                            uint32_t n, double* __restrict__ out, double* __restrict__ grad) {
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    stageQueryTables(T, sNl, sRec);
    __syncthreads();
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) queryPointWithGradient(t, i, xyz, out, grad, sNl, sRec);
}

// Octree::QueryWithGradient + FApproxWithGradient (Octree.cpp:749-789, 904-985), any degree, one lane per point:
// QueryWithGradient for one point whose leaf has degree 4 or 5, with the degree at compile time: the walk of queryPointWithGradient,
// then the leaf's coefficients into registers and evalLeafGradVals<P> -- the statements of the any-degree loop above, unrolled, its
// tables of basis values in registers instead of 936 bytes of scratch indexed through kBasis (the dense pass over union3 @ 1e-7's
// deferred points: 64 -> 20 us).  Any other degree goes through the any-degree code.
template <int P>
__device__ __forceinline__ void leafGradFixed(const TreeDev& t, const double* __restrict__ co, const double (&u)[3], int depth, size_t i,
                                              double* __restrict__ out, double* __restrict__ grad, const double* sNl, const double* sRec) {
    constexpr int N = coeffCount(P);
    double cv[N];
#pragma unroll
    for (int r = 0; r < N; ++r) cv[r] = co[r];
    double g[3];
    out[i] = evalLeafGradVals<P>(cv, u, depth, sNl, sRec, g, t.leftAssoc);
    grad[3 * i] = g[0], grad[3 * i + 1] = g[1], grad[3 * i + 2] = g[2];
}
__device__ __forceinline__ void queryPointWithGradient45(const TreeDev& t, size_t i, const double* __restrict__ xyz, double* __restrict__ out,
                                                         double* __restrict__ grad, const double* sNl, const double* sRec) {
    const double p[3] = {(xyz[3 * i] - t.rootCentre[0]) * t.rootInvSizes[0], (xyz[3 * i + 1] - t.rootCentre[1]) * t.rootInvSizes[1],
                         (xyz[3 * i + 2] - t.rootCentre[2]) * t.rootInvSizes[2]};
    const float fx = (float)p[0], fy = (float)p[1], fz = (float)p[2];
    if (!(fx >= -0.5f && fx <= 0.5f && fy >= -0.5f && fy <= 0.5f && fz >= -0.5f && fz <= 0.5f)) {
        out[i] = DBL_MAX;
        return;
    }
    double c[3] = {0.0, 0.0, 0.0}, q = 0.25;
    int depth = 0;
    NodeRec rec = t.nodes[0];
    while (rec.b == kInteriorTag) {  // :674-701
        uint32_t idx = rec.a;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const bool up = p[a] >= c[a];
            idx += up ? (1u << a) : 0u;
            c[a] = up ? c[a] + q : c[a] - q;
        }
        q = q * 0.5;
        ++depth;
        rec = t.nodes[idx];
    }
    const double s = (double)(2 << depth);
    const double u[3] = {(p[0] - c[0]) * s, (p[1] - c[1]) * s, (p[2] - c[2]) * s};  // :907
    if (rec.b == 4u)
        leafGradFixed<4>(t, t.coeffs + rec.a, u, depth, i, out, grad, sNl, sRec);
    else if (rec.b == 5u)
        leafGradFixed<5>(t, t.coeffs + rec.a, u, depth, i, out, grad, sNl, sRec);
    else
        queryPointWithGradient(t, i, xyz, out, grad, sNl, sRec);
}

// the second pass of the gradient query for the points query_general_kernel<.., GRAD> deferred (leaves of degree > 3).
// Dense over the concatenation of the per-workgroup lists, like query_deep_kernel.  FIXED45: the tree's degrees stop at 5.
template <bool FIXED45>
__global__ __launch_bounds__(256) void query_grad_deep_kernel(TreeDev t, const DeviceTables* __restrict__ T,
                                                              const double* __restrict__ xyz, double* __restrict__ out,
                                                              double* __restrict__ grad, uint32_t tilesPerWg, uint32_t nWg,
                                                              const uint32_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ deferIdx) {
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    const uint32_t total = offsets[nWg];
    if ((uint32_t)(blockIdx.x * blockDim.x) >= total) return;  // workgroup-uniform
    stageQueryTables(T, sNl, sRec);
    __syncthreads();
    for (uint32_t jj = blockIdx.x * blockDim.x + threadIdx.x; jj < total; jj += gridDim.x * blockDim.x) {
        const size_t i = deferredPoint(jj, offsets, nWg, tilesPerWg, deferIdx);
        if constexpr (FIXED45)
            queryPointWithGradient45(t, i, xyz, out, grad, sNl, sRec);
        else
            queryPointWithGradient(t, i, xyz, out, grad, sNl, sRec);
    }
}

// Octree::QueryRay (Octree.cpp:705-746) over Ray / Ray::IntersectAABB (Source/HP/Ray.cpp:5-68): sphere tracing,
// at most 200 Query steps per ray, one lane per ray.  The reference's behaviour is kept statement by statement,
// including what looks unintended: the origin is mapped to the unit cube but the direction is not (:711); for an
// origin outside the root `intMin` is what IntersectAABB leaves in its first output -- the entry parameter in x,
// per-axis slab parameters in y and z -- not a point (:717); Query maps its argument through the root transform
// again (:726 -> :665); and on a hit t_ receives the field value (:730).  t of a miss is left untouched.
template <int MAXP>
__global__ __launch_bounds__(256) void query_ray_kernel(TreeDev t, const DeviceTables* __restrict__ T,
                                                        const double* __restrict__ origins,
                                                        const double* __restrict__ dirs, const double* __restrict__ tMax,
                                                        size_t n, uint8_t* __restrict__ hit, double* __restrict__ tOut) {
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    stageQueryTables(T, sNl, sRec);
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double o[3], d[3], inv[3];
        int sgn[3];
        for (int a = 0; a < 3; ++a) {
            o[a] = (origins[3 * i + a] - t.rootCentre[a]) * t.rootInvSizes[a];  // :711
            d[a] = dirs[3 * i + a];
            inv[a] = 1.0 / d[a];  // Ray.cpp:10 cwiseInverse
            sgn[a] = inv[a] < 0.0 ? 1 : 0;
        }
        double im[3] = {o[0], o[1], o[2]};  // intMin
        const float fx = (float)o[0], fy = (float)o[1], fz = (float)o[2];
        const bool inside = fx >= -0.5f && fx <= 0.5f && fy >= -0.5f && fy <= 0.5f && fz >= -0.5f && fz <= 0.5f;
        bool miss = false;
        if (!inside) {  // Ray::IntersectAABB on [-0.5,0.5]^3, Ray.cpp:18-68
            double a0 = ((sgn[0] ? 0.5 : -0.5) - o[0]) * inv[0], b0 = ((sgn[0] ? -0.5 : 0.5) - o[0]) * inv[0];
            const double a1 = ((sgn[1] ? 0.5 : -0.5) - o[1]) * inv[1], b1 = ((sgn[1] ? -0.5 : 0.5) - o[1]) * inv[1];
            if ((a0 > b1) || (a1 > b0)) {
                miss = true;
            } else {
                if (a1 > a0) a0 = a1;
                if (b1 < b0) b0 = b1;
                const double a2 = ((sgn[2] ? 0.5 : -0.5) - o[2]) * inv[2], b2 = ((sgn[2] ? -0.5 : 0.5) - o[2]) * inv[2];
                if ((a0 > b2) || (a2 > b0)) {
                    miss = true;
                } else {
                    if (a2 > a0) a0 = a2;
                    im[0] = a0, im[1] = a1, im[2] = a2;
                }
            }
        }
        uint8_t h = 0;
        if (!miss) {
            const double eps = 0.0001, minStep = 0.0001, lim = tMax[i];
            double dist = 0.0;
            for (int s = 0; s < 200; ++s) {
                const double v = queryPoint<MAXP>(t, im[0] + dist * d[0], im[1] + dist * d[1], im[2] + dist * d[2], sNl, sRec);
                if (v < eps) {
                    tOut[i] = v;  // :730
                    h = 1;
                    break;
                }
                dist = dist + (v * 0.95 + minStep);  // :736
                if (dist > lim) break;
            }
        }
        hit[i] = h;
    }
}

// Sample points of Octree::OutputFunctionSlice (Octree.cpp:1144-1170): pixel (i, j) queries
// (min.x + (f32)j*step, min.y + (f32)i*step, c), step = (max.x - min.x) / nSamples in f32.  The points then go
// through the batched Query kernels (a row of pixels is a coherent run of points).
__global__ __launch_bounds__(256) void slice_points_kernel(double c, float minX, float minY, float step, uint32_t nSamples,
                                                           double* __restrict__ xyz) {
    const size_t total = (size_t)nSamples * nSamples, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const uint32_t i = (uint32_t)(p / nSamples), j = (uint32_t)(p - (size_t)i * nSamples);
        xyz[3 * p] = (double)minX + (double)((float)j * step);
        xyz[3 * p + 1] = (double)minY + (double)((float)i * step);
        xyz[3 * p + 2] = c;
    }
}

// the CSG wrapper of Octree.cpp:355-400 around an inner field value v
template <bool CSG>
__device__ __forceinline__ double applyCsg(const FieldDev& f, double v, double x, double y, double z, const double* sNl,
                                           const double* sRec) {
    if constexpr (CSG) {
        const double o = queryPoint<12>(f.oldTree, x, y, z, sNl, sRec);
        switch (f.csgOp) {
            case HPSDF_OP_UNION: v = o < v ? o : v; break;                   // std::min(old, F)
            case HPSDF_OP_SUBTRACT: v = (o * -1.0) < v ? v : (o * -1.0); break;  // std::max(-old, F)
            default: v = o < v ? v : o; break;                               // std::max(old, F)
        }
    }
    return v;
}

// F at a world-space point, with the optional CSG wrapper of Octree.cpp:355-400
template <int KIND, bool CSG, bool LEFT>
__device__ __forceinline__ double fieldEvalWorld(const FieldDev& f, double x, double y, double z, uint64_t sampleIdx,
                                                 const double* sNl, const double* sRec, uint32_t& meshHint) {
    double v;
    if constexpr (KIND == kFieldAnalytic)
        v = analyticEval<LEFT>(f, x, y, z);
    else if constexpr (KIND == kFieldSamples)
        v = f.samples[sampleIdx];
    else  // SURVEY 3.4 user glue: (f64) mesh.SignedDistanceAtPt(p.cast<f32>())
        v = (double)meshSignedDistance(f.mesh, V3{(float)x, (float)y, (float)z}, meshHint);
    return applyCsg<CSG>(f, v, x, y, z, sNl, sRec);
}

template <int KIND, bool CSG, bool LEFT>
__global__ __launch_bounds__(256) void field_kernel(FieldDev f, const DeviceTables* __restrict__ T,
                                                    const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    stageQueryTables(T, sNl, sRec);
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t hint = 0xFFFFFFFFu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = fieldEvalWorld<KIND, CSG, LEFT>(f, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], i, sNl, sRec, hint);
}

// ---------------------------------------------------------------------------
// fit: Octree::FitPolynomial, Octree.cpp:1007-1093
// ---------------------------------------------------------------------------
//
// One workgroup fits blk.nTasks cells of identical shape.  The reference loops
// samples (i,j,k) outermost and coefficients innermost; here each thread owns
// one (cell, coefficient row) accumulator and walks the samples in the same
// (i,j,k) order, so every coefficient is the same left-to-right sum.  The
// samples are processed in chunks of whole i-planes (as many as LDS holds):
// phase 1 evaluates F on the chunk's samples of every cell into LDS (all 256
// threads), phase 2 accumulates.  The per-term product
//   Lp = 1 * P_i0(x) * N_i0 * P_i1(y) * N_i1 * P_i2(z) * N_i2      (:1045-1050)
// is hoisted by loop level without changing its association.
//
// DEG > 0 fixes the degree at compile time (nq = 4*DEG+1): the innermost loop
// unrolls, the thread's P_i2 row lives in registers and the LDS reads of a row
// are issued together instead of one dependent read per term; DEG == 0 is the
// any-degree version (rows walked in groups of four, nq = 4p+1).

constexpr int kFitThreads = 256;
constexpr int kFitPiece = 8;  // terms of a row a degree-6..8 body forms at a time
#ifndef HPSDF_FIT_MIN_WAVES
#define HPSDF_FIT_MIN_WAVES 4  // waves per SIMD the register allocation must leave room for: <= 128 VGPRs.  Left alone the
                               // degree-2 kernel takes 240 (two waves per SIMD); held to 128 it spills 448 bytes and is 37 % faster
                               // (65 536 cells: 1.59 -> 2.18 TFLOP/s with the union3 field), degrees 4-5 gain 3-4 %
#endif

// Mesh fields: the order in which the np x nq x nq samples of a chunk are handed to the lanes.  A wave answers its 64
// closest-triangle queries with ONE traversal whose cost is the union of what its lanes need, so the 64 samples should
// sit close together: the chunk is cut into 4 x 4 x 4 blocks (smaller at the upper edges), blocks in (i, j, k) order,
// samples inside a block likewise -- in SPACE, not in index: the Gauss-Legendre tables list their roots as 0, -a, +a, ...
// (Legendre.h), so posI / posJK map a position along the axis (ascending coordinate) to the root's index (posI: among
// the chunk's np planes).  Returns the sample (il * nq + j) * nq + k that position r of that order holds.
// Every sample's value is independent of its companions (pruning is per lane), so this is a pure scheduling choice.
__device__ __forceinline__ int meshSampleOrder(int r, int np, int nq, const unsigned char* posI, const unsigned char* posJK) {
    const int nq2 = nq * nq;
    const int nbi = (np + 3) >> 2, nbj = (nq + 3) >> 2;
    int bi = min(r / (4 * nq2), nbi - 1);
    r -= bi * 4 * nq2;
    const int di = min(4, np - 4 * bi);
    const int strip = di * 4 * nq;
    int bj = min(r / strip, nbj - 1);
    r -= bj * strip;
    const int dj = min(4, nq - 4 * bj);
    const int blk = di * dj * 4;
    int bk = min(r / blk, nbj - 1);
    r -= bk * blk;
    const int dk = min(4, nq - 4 * bk);
    const int a = r / (dj * dk), rest = r - a * (dj * dk), b = rest / dk, c = rest - b * dk;
    return ((int)posI[4 * bi + a] * nq + (int)posJK[4 * bj + b]) * nq + (int)posJK[4 * bk + c];
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// acc += sum_k ((a1 * tk[k]) * n2) * F[k], k ascending
__device__ __forceinline__ double fitRowAny(double acc, double a1, const double* __restrict__ tk, double n2,
                                            const double* __restrict__ F, int nq) {
    int k = 0;
    for (; k + 4 <= nq; k += 4) {
        const double f0 = F[k], f1 = F[k + 1], f2 = F[k + 2], f3 = F[k + 3];
        const double t0 = tk[k], t1 = tk[k + 1], t2 = tk[k + 2], t3 = tk[k + 3];
        acc = acc + (a1 * t0 * n2) * f0;
        acc = acc + (a1 * t1 * n2) * f1;
        acc = acc + (a1 * t2 * n2) * f2;
        acc = acc + (a1 * t3 * n2) * f3;
    }
    for (; k < nq; ++k) acc = acc + (a1 * tk[k] * n2) * F[k];
    return acc;
}

// R > 1 (DEG > 0 only): a thread owns one row of R cells.  The product Lp of a sample does not depend on
// the cell (all cells of a workgroup share degree and depth), so it is formed once per sample and used
// for R accumulators: 2 + 3/R multiply/add instructions per (cell, sample, row) instead of 5.
template <int KIND, bool CSG, int DEG, int R, bool LEFT>
__device__ __forceinline__ void fitBlockBody(const FitBlock blk, const FitTask* __restrict__ tasks, double* __restrict__ arena,
                                             double* __restrict__ errs, double* __restrict__ mirror,
                                             const DeviceTables* __restrict__ T, const FieldDev& field, const RootMap& rm, double* lds) {
    static_assert(R == 1 || DEG > 0, "cell blocking needs a compile-time degree");
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    __shared__ int32_t sMeshStack[kFitThreads / 64][kMeshStack];  // per-wave traversal stacks (mesh fields)
    __shared__ unsigned char sPosI[KIND == kFieldMesh ? 64 : 4], sPosJK[KIND == kFieldMesh ? 64 : 4];  // meshSampleOrder
    const int tid = threadIdx.x;
    const int deg = DEG > 0 ? DEG : (int)blk.degree;
    const int nq = 4 * deg + 1, nq2 = nq * nq, G = blk.nTasks;
    const int rowStart = blk.rowStart, rowEnd = blk.rowEnd, nrows = rowEnd - rowStart;
    const int gl = nq * (nq - 1) / 2;  // Legendre.h: rule n starts at n(n-1)/2 (:1016-1017)
    const int planes = blk.planesPerChunk;  // i-planes per chunk
    const int depth = blk.depth;            // every cell of the workgroup has this depth
    const bool split = blk.split != 0;       // the top-degree rows of a from-scratch fit (device_types.hpp): outOff addresses row 0

    double* sT = lds;               // [deg+1][nq]  LpX(p, root_q)
    double* sR = sT + (deg + 1) * nq;  // [nq] roots
    double* sW = sR + nq;           // [nq] weights
    double* sC = sW + nq;           // [G][8]  scale xyz, centre xyz, scale product, sample offset (bits)
    double* sF = sC + 8 * G;        // [G][planes][nq2] weighted samples of the current chunk

    stageQueryTables(T, sNl, sRec);
    for (int q = tid; q < nq; q += kFitThreads) {
        const double x = T->roots[gl + q];
        sR[q] = x;
        sW[q] = T->weights[gl + q];
        // Octree::LpX, :988-1004
        double m2 = 0.0, m1 = 1.0;
        sT[q] = 1.0;
        for (int i = 1; i <= deg; ++i) {
            const double l = T->rec[i][0] * x * m1 - T->rec[i][1] * m2;
            m2 = m1, m1 = l;
            sT[i * nq + q] = l;
        }
    }
    for (int g = tid; g < G; g += kFitThreads) {
        const FitTask& tk = tasks[blk.firstTask + g];
        double sc[3];
        for (int a = 0; a < 3; ++a) {
            sc[a] = (double)(tk.bmax[a] - tk.bmin[a]) * 0.5;               // :1020 sizes() in f32
            sC[8 * g + 3 + a] = (double)((tk.bmin[a] + tk.bmax[a]) / 2.0f);  // :1021 center() in f32
            sC[8 * g + a] = sc[a];
        }
        sC[8 * g + 6] = prod3<LEFT>(sc[0], sc[1], sc[2]);  // :1022 Eigen prod()
        sC[8 * g + 7] = __longlong_as_double((long long)tk.sampleOff);
    }
    __syncthreads();

    // phase-2 ownership: thread -> (cell slot, row); a slot is R consecutive cells; cells with > 256 rows
    // (any-degree kernel only) use two rows per thread
    // (two rows per thread only exist where a cell has more than 256 rows, i.e. beyond degree 9: the any-degree kernel.  Saying so
    // at compile time removes the second row's code from the degree-specialised kernels -- it was where the degree-2 kernel spilled
    // 448 bytes per lane: a fully unrolled plane of LDS reads for a branch that never runs)
    const bool wide = DEG == 0 && nrows > kFitThreads;
    const int slot = wide ? 0 : tid / nrows;
    const int g2 = slot * R;  // first cell of this thread
    const int r0 = rowStart + (wide ? tid : tid % nrows);
    const int r1 = r0 + kFitThreads;
    const bool act0 = wide ? (r0 < rowEnd) : (tid < ((G + R - 1) / R) * nrows);
    const bool act1 = wide && r1 < rowEnd;
    int i0a = 0, i1a = 0, i2a = 0, i0b = 0, i1b = 0, i2b = 0;
    double n0a = 0, n1a = 0, n2a = 0, n0b = 0, n1b = 0, n2b = 0;
    if (act0) {
        i0a = T->bidx[r0][0], i1a = T->bidx[r0][1], i2a = T->bidx[r0][2];
        n0a = sNl[i0a * 11 + depth], n1a = sNl[i1a * 11 + depth], n2a = sNl[i2a * 11 + depth];
    }
    if (act1) {
        i0b = T->bidx[r1][0], i1b = T->bidx[r1][1], i2b = T->bidx[r1][2];
        n0b = sNl[i0b * 11 + depth], n1b = sNl[i1b * 11 + depth], n2b = sNl[i2b * 11 + depth];
    }
    constexpr int NQF = DEG > 0 ? 4 * DEG + 1 : 1;
    double tkReg[NQF];  // this thread's P_i2 row (DEG > 0)
    if (DEG > 0) {
#pragma unroll
        for (int k = 0; k < NQF; ++k) tkReg[k] = sT[i2a * nq + k];
    }
    uint32_t meshHint = 0xFFFFFFFFu;  // closest triangle of this thread's previous sample (mesh fields)
    double acc[R];  // :1025
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    double acc1 = 0.0;

    const int cellStride = planes * nq2;  // doubles per cell in sF
    for (int iBase = 0; iBase < nq; iBase += planes) {
        const int np = min(planes, nq - iBase);
        // ---- phase 1: F on planes [iBase, iBase+np) of every cell (:1035-1040)
        if constexpr (KIND == kFieldMesh) {  // position -> root index by rank (roots are distinct), nq <= 49
            if (tid < nq) {
                int rank = 0;
                for (int b = 0; b < nq; ++b) rank += sR[b] < sR[tid] ? 1 : 0;
                sPosJK[rank] = (unsigned char)tid;
            }
            if (tid >= 64 && tid < 64 + np) {
                const int a = tid - 64;
                int rank = 0;
                for (int b = 0; b < np; ++b) rank += sR[iBase + b] < sR[iBase + a] ? 1 : 0;
                sPosI[rank] = (unsigned char)a;
            }
            __syncthreads();
        }
        if constexpr (KIND != kFieldMesh) {
            // A thread takes (cell, j, k) COLUMNS of the chunk and walks the chunk's planes i: the column's index arithmetic, its y and z
            // coordinates and (a . (b . c) order) the product w_j w_k are formed once per column instead of once per sample -- ~30 of a
            // sample's ~330 instructions at degree 2.  The sample's own statements are those of the sample-major loop below, so are the bits.
            const int cols = G * nq2;
            for (int c0 = 0; c0 < cols; c0 += kFitThreads) {
                const int cc = c0 + tid;
                const bool activeS = cc < cols;
                const int ccl = activeS ? cc : cols - 1;
                const int g = ccl / nq2, jk = ccl - g * nq2, j = jk / nq, k = jk - j * nq;
                const double* c = sC + 8 * g;
                const double uy = sR[j] * c[1] + c[4], uz = sR[k] * c[2] + c[5];
                const double wy = uy * rm.bounds[1] + rm.centre[1];  // Octree.cpp:327
                const double wz = uz * rm.bounds[2] + rm.centre[2];
                const double wj = sW[j], wk = sW[k], c6 = c[6];
                const uint64_t sbase = (uint64_t)__double_as_longlong(c[7]) + (uint64_t)(j * nq + k);
                for (int il = 0; il < np; ++il) {
                    const int i = iBase + il;
                    const double ux = sR[i] * c[0] + c[3];
                    const double wx = ux * rm.bounds[0] + rm.centre[0];
                    const uint64_t sidx = sbase + (uint64_t)(i * nq2);
                    const double fv = activeS ? fieldEvalWorld<KIND, CSG, LEFT>(field, wx, wy, wz, sidx, sNl, sRec, meshHint) : 0.0;
                    if (activeS) {
                        sF[g * cellStride + il * nq2 + jk] = c6 * prod3<LEFT>(sW[i], wj, wk) * fv;  // :1040
                        // a split fit (FitBlock::split): the field's value goes (back) into the sample buffer, from where
                        // fit_low_kernel computes the rows below the top degree
                        if (split) const_cast<double*>(field.samples)[sidx] = fv;
                    }
                }
            }
        } else {
            const int chunkSamples = np * nq2, total = G * chunkSamples;
            const float invChunk = 1.0f / (float)chunkSamples;
            for (int s0 = 0; s0 < total; s0 += kFitThreads) {  // every lane iterates (the mesh path works wave-wide)
                const int s = s0 + tid;
                const bool activeS = s < total;
                const int sc = activeS ? s : total - 1;
                // s -> (cell g, sample rem) without an integer division by the run-time chunk size
                int g = (int)(((float)sc + 0.5f) * invChunk);
                int rem = sc - g * chunkSamples;
                if (rem < 0) {
                    --g;
                    rem += chunkSamples;
                } else if (rem >= chunkSamples) {
                    ++g;
                    rem -= chunkSamples;
                }
                if constexpr (KIND == kFieldMesh) rem = meshSampleOrder(rem, np, nq, sPosI, sPosJK);
                const double* c = sC + 8 * g;
                const int il = rem / nq2, jk = rem - il * nq2, j = jk / nq, k = jk - j * nq, i = iBase + il;
                const double ux = sR[i] * c[0] + c[3], uy = sR[j] * c[1] + c[4], uz = sR[k] * c[2] + c[5];
                const double wx = ux * rm.bounds[0] + rm.centre[0];  // Octree.cpp:327
                const double wy = uy * rm.bounds[1] + rm.centre[1];
                const double wz = uz * rm.bounds[2] + rm.centre[2];
                const uint64_t sidx = (uint64_t)__double_as_longlong(c[7]) + (uint64_t)((i * nq + j) * nq + k);
                double fv;
                if constexpr (KIND == kFieldMesh) {
                    // SURVEY 3.4 user glue: (f64) mesh.SignedDistanceAtPt(p.cast<f32>()) -- one traversal per wave
                    const double mv = (double)meshSignedDistanceWave(field.mesh, V3{(float)wx, (float)wy, (float)wz}, activeS, meshHint,
                                                                     sMeshStack[tid >> 6]);
                    fv = activeS ? applyCsg<CSG>(field, mv, wx, wy, wz, sNl, sRec) : 0.0;
                } else {
                    fv = activeS ? fieldEvalWorld<KIND, CSG, LEFT>(field, wx, wy, wz, sidx, sNl, sRec, meshHint) : 0.0;
                }
                if (activeS) {
                    sF[g * cellStride + rem] = c[6] * prod3<LEFT>(sW[i], sW[j], sW[k]) * fv;  // :1040
                    // a split fit (FitBlock::split): the field's value goes (back) into the sample buffer, from where
                    // fit_mfma_low_kernel contracts the rows below the top degree
                    if constexpr (KIND != kFieldMesh)
                        if (split) const_cast<double*>(field.samples)[sidx] = fv;
                }
            }
        }
        __syncthreads();
        // ---- phase 2: accumulate the chunk, planes ascending (:1043-1053)
        if (act0) {
            const double* tj = sT + i1a * nq;
            const double* tk = sT + i2a * nq;
            for (int il = 0; il < np; ++il) {
                const double* F = sF + g2 * cellStride + il * nq2;
                const double a0 = sT[i0a * nq + iBase + il] * n0a;
#ifdef HPSDF_FIT_ROW_UNROLL
#pragma unroll HPSDF_FIT_ROW_UNROLL
#endif
                for (int j = 0; j < nq; ++j) {
                    const double a1 = a0 * tj[j] * n1a;
                    if constexpr (DEG > 0 && DEG <= 5) {
                        double lp[NQF];
#pragma unroll
                        for (int k = 0; k < NQF; ++k) lp[k] = a1 * tkReg[k] * n2a;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const double* Fr = F + r * cellStride + j * nq;
                            double f[NQF];
#pragma unroll
                            for (int k = 0; k < NQF; ++k) f[k] = Fr[k];
#pragma unroll
                            for (int k = 0; k < NQF; ++k) acc[r] = acc[r] + lp[k] * f[k];
                        }
                    } else if constexpr (DEG > 5) {
                        // degrees 6..8: the thread's P_i2 row still lives in registers (25 / 29 / 33 doubles), the row of samples is
                        // taken in pieces of kFitPiece so that factors and samples in flight stay within the 128 registers the four
                        // waves a SIMD leave a lane (terms in the same order: k ascending)
                        const double* Fr = F + j * nq;
#pragma unroll
                        for (int k0 = 0; k0 < NQF; k0 += kFitPiece) {
                            double lp[kFitPiece], f[kFitPiece];
#pragma unroll
                            for (int k = 0; k < kFitPiece; ++k)
                                if (k0 + k < NQF) f[k] = Fr[k0 + k];
#pragma unroll
                            for (int k = 0; k < kFitPiece; ++k)
                                if (k0 + k < NQF) lp[k] = a1 * tkReg[k0 + k] * n2a;
#pragma unroll
                            for (int k = 0; k < kFitPiece; ++k)
                                if (k0 + k < NQF) acc[0] = acc[0] + lp[k] * f[k];
                        }
                    } else {
                        acc[0] = fitRowAny(acc[0], a1, tk, n2a, F + j * nq, nq);
                    }
                }
            }
        }
        if (act1) {
            const double* tj = sT + i1b * nq;
            const double* tk = sT + i2b * nq;
            for (int il = 0; il < np; ++il) {
                const double* F = sF + il * nq2;
                const double a0 = sT[i0b * nq + iBase + il] * n0b;
                for (int j = 0; j < nq; ++j) {
                    const double a1 = a0 * tj[j] * n1b;
                    acc1 = fitRowAny(acc1, a1, tk, n2b, F + j * nq, nq);
                }
            }
        }
        __syncthreads();
    }

    // coefficients out; the new rows are also stashed in LDS for the error sum.  Plain fits write only their
    // rows (outOff addresses row rowStart; an incremental fit, :847/:1012, leaves rows [0,rowStart) where the
    // earlier fit of the cell put them).  Weighted fits keep one full array per cell: outOff addresses row 0.
    const bool weighted = blk.weighted != 0;
    const int stashStride = weighted ? rowEnd : nrows, stashBase = weighted ? 0 : rowStart;
    const int outBase = split ? 0 : stashBase;  // (a split fit's array starts at row 0: the rows below rowStart come from the matrix cores)
    if (act0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (g2 + r < G) {
                const uint64_t at = tasks[blk.firstTask + g2 + r].outOff + (r0 - outBase);
                arena[at] = acc[r];
                // mirror (round 0 of the device-side frontier, one rank): host memory the device writes straight into -- a build that
                // stops after that round has its packed store there when the round closes, without a copy to wait for
                if (mirror != nullptr) mirror[at] = acc[r];
                sF[(g2 + r) * stashStride + (r0 - stashBase)] = acc[r];
            }
    }
    if (act1) {
        arena[tasks[blk.firstTask].outOff + (r1 - outBase)] = acc1;
        sF[r1 - stashBase] = acc1;
    }
    if (weighted && rowStart > 0) {  // carry the old rows over (:847)
        for (int s = tid; s < G * rowStart; s += kFitThreads) {
            const int g = s / rowStart, r = s - g * rowStart;
            const FitTask& tk = tasks[blk.firstTask + g];
            const double v = arena[tk.copyOff + r];
            arena[tk.outOff + r] = v;
            sF[g * stashStride + r] = v;
        }
    }
    __syncthreads();
    // :1062-1069  error = sum of squares of the rows of total degree == deg, in row order
    for (int g = tid; g < G; g += kFitThreads) {
        const int first = deg > 0 ? (int)T->count[deg - 1] : 0;
        double e = 0.0;
        for (int r = first > rowStart ? first : rowStart; r < rowEnd; ++r)
            if (T->bidx[r][3] == deg) {
                const double c = sF[g * stashStride + (r - stashBase)];
                e = e + c * c;
            }
        errs[tasks[blk.firstTask + g].errSlot] = e;
    }
}

// All the fits of a round in ONE launch (the device-side frontier, frontier.hip): the round's blocks lie degree by degree in
// one array and carry their degree, so a workgroup picks the compile-time-specialised body its block needs.  The blocks are
// handed out from the END of the array -- highest degree, longest fits first -- and workgroups of every degree share the chip
// at once, which is what the per-degree launches on side streams were for (their fork / join events cost 20-50 us a round).
// count: the round's number of blocks, written by the device; the grid is an upper bound.
template <int KIND, bool CSG, bool LEFT>
__global__ __launch_bounds__(kFitThreads, HPSDF_FIT_MIN_WAVES) void fit_multi_kernel(const FitBlock* __restrict__ blocks,
                                                                const FitTask* __restrict__ tasks, double* __restrict__ arena,
                                                                double* __restrict__ errs, const DeviceTables* __restrict__ T,
                                                                FieldDev field, RootMap rm, const uint32_t* __restrict__ count,
                                                                uint32_t countValue) {
    extern __shared__ double lds[];
    const uint32_t n = count ? *count : countValue;  // (the host scheduler knows the number, the device-side frontier writes it)
    if (blockIdx.x >= n) return;
    const FitBlock blk = blocks[n - 1u - blockIdx.x];
    switch (blk.degree) {
        case 2: fitBlockBody<KIND, CSG, 2, 1, LEFT>(blk, tasks, arena, errs, nullptr, T, field, rm, lds); break;
        case 3: fitBlockBody<KIND, CSG, 3, 1, LEFT>(blk, tasks, arena, errs, nullptr, T, field, rm, lds); break;
        case 4: fitBlockBody<KIND, CSG, 4, 1, LEFT>(blk, tasks, arena, errs, nullptr, T, field, rm, lds); break;
        case 5: fitBlockBody<KIND, CSG, 5, 1, LEFT>(blk, tasks, arena, errs, nullptr, T, field, rm, lds); break;
        case 6: fitBlockBody<KIND, CSG, 6, 1, LEFT>(blk, tasks, arena, errs, nullptr, T, field, rm, lds); break;
        case 7: fitBlockBody<KIND, CSG, 7, 1, LEFT>(blk, tasks, arena, errs, nullptr, T, field, rm, lds); break;
        case 8: fitBlockBody<KIND, CSG, 8, 1, LEFT>(blk, tasks, arena, errs, nullptr, T, field, rm, lds); break;
        default: fitBlockBody<KIND, CSG, 0, 1, LEFT>(blk, tasks, arena, errs, nullptr, T, field, rm, lds); break;
    }
}

// Nearness weighting, Octree.cpp:1209-1247: |mean of FApprox over 100 points of the cell| for every fit of a weighted
// build, from the full coefficient arrays fit_kernel has just written (a kernel of its own: its any-degree evaluation
// keeps tables in private memory, and inside fit_kernel that put 320 bytes of scratch on every fit launch, weighted or
// not).  The points come from a hash of (cell, sample, axis) instead of std::rand; the weight itself
// (pow / exp) is applied by the host so that it matches the CPU path bit for bit.
__global__ __launch_bounds__(kFitThreads) void fit_weight_kernel(const FitBlock* __restrict__ blocks, const FitTask* __restrict__ tasks,
                                                                 const double* __restrict__ arena, double* __restrict__ means,
                                                                 const DeviceTables* __restrict__ T, const uint32_t* __restrict__ count) {
    extern __shared__ double lds[];
    __shared__ double sNl[13 * 11];
    __shared__ double sRec[26];
    if (count != nullptr && blockIdx.x >= *count) return;  // (the device-side frontier writes the round's block count; the grid is an upper bound)
    const FitBlock blk = blocks[blockIdx.x];
    const int tid = threadIdx.x, deg = blk.degree, depth = blk.depth, G = blk.nTasks, nc = blk.rowEnd;
    stageQueryTables(T, sNl, sRec);
    double* sCo = lds;           // [G][nc] coefficients
    double* sV = lds + G * nc;   // [G][100]
    for (int s = tid; s < G * nc; s += kFitThreads) {
        const int g = s / nc, r = s - g * nc;
        sCo[s] = arena[tasks[blk.firstTask + g].outOff + r];
    }
    __syncthreads();
    for (int s = tid; s < G * 100; s += kFitThreads) {
        const int g = s / 100, n = s - g * 100;
        const FitTask& tk = tasks[blk.firstTask + g];
        const uint64_t ka = (uint64_t)__float_as_uint(tk.bmin[0]) | ((uint64_t)__float_as_uint(tk.bmin[1]) << 32);
        const uint64_t kb = (uint64_t)__float_as_uint(tk.bmin[2]) | ((uint64_t)(unsigned)depth << 32) |
                            ((uint64_t)(unsigned)deg << 40);
        const uint64_t key = splitmix64(ka) ^ splitmix64(kb ^ 0xD1B54A32D192ED03ull);
        double u[3];
        for (int a = 0; a < 3; ++a) {
            const float r = (float)(splitmix64(key + ((uint64_t)n * 3 + (uint64_t)a) * 0x9E3779B97F4A7C15ull) >> 40) *
                            (1.0f / 16777216.0f);
            const double pt = (double)(tk.bmin[a] + (tk.bmax[a] - tk.bmin[a]) * r);  // AlignedBox3f::sample()
            const double centre = (double)((tk.bmin[a] + tk.bmax[a]) / 2.0f);        // :1021 center() in f32
            u[a] = (pt - centre) * (double)(2 << depth);                             // :862
        }
        sV[s] = evalLeafGeneric(sCo + g * nc, deg, u[0], u[1], u[2], depth, sNl, sRec);
    }
    __syncthreads();
    for (int g = tid; g < G; g += kFitThreads) {
        double fIntegral = 0.0;
        for (int n = 0; n < 100; ++n) fIntegral = fIntegral + sV[g * 100 + n];
        fIntegral = fIntegral / 100.0;
        means[tasks[blk.firstTask + g].errSlot] = fabs(fIntegral);
    }
}

// range == nullptr: workgroup b fits blocks[b] (the host sized the grid).  Otherwise the grid is an upper bound and
// workgroup b fits blocks[range[0] + b] if b < range[1]: the device-side frontier (frontier.hip) writes the round's block
// list and its per-degree ranges itself, so the host never learns how many blocks a round has before it launches the fits
// (a grid-stride loop over the range instead cost 37 more VGPRs at degree 3-4: one wave per SIMD less).
template <int KIND, bool CSG, int DEG, int R, bool LEFT>
__global__ __launch_bounds__(kFitThreads, HPSDF_FIT_MIN_WAVES) void fit_kernel(const FitBlock* __restrict__ blocks,
                                                          const FitTask* __restrict__ tasks, double* __restrict__ arena,
                                                          double* __restrict__ errs, double* __restrict__ mirror,
                                                          const DeviceTables* __restrict__ T, FieldDev field, RootMap rm,
                                                          const uint32_t* __restrict__ range) {
    extern __shared__ double lds[];
    uint32_t b = blockIdx.x;
    if (range != nullptr) {
        if (b >= range[1]) return;
        b += range[0];
    }
    fitBlockBody<KIND, CSG, DEG, R, LEFT>(blocks[b], tasks, arena, errs, mirror, T, field, rm, lds);
}

// ---------------------------------------------------------------------------
// pack: Octree::ReallocCoeffs gather (Octree.cpp:510-552)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const PackItem* __restrict__ items, uint32_t nItems,
                                                   const double* __restrict__ arena, double* __restrict__ out) {
    // one wave per leaf
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= nItems) return;
    const PackItem it = items[wave];
    for (uint32_t i = lane; i < it.count; i += 64) out[it.dst + i] = arena[it.src + i];
}

// ---------------------------------------------------------------------------
// launch wrappers (host)
// ---------------------------------------------------------------------------
size_t fitLdsBytes(int degree, int nTasks, int planes) {
    const size_t nq = 4 * (size_t)degree + 1;
    // sT + roots + weights + per-cell constants + `planes` sample planes per cell (also holds the new rows at the end)
    return ((size_t)(degree + 1) * nq + 2 * nq + 8 * (size_t)nTasks + (size_t)nTasks * planes * nq * nq) * sizeof(double);
}

// Shape of the workgroups of one class: `count` fits of `nrows` coefficient rows at `degree`.
FitShape fitShape(int degree, int nrows, uint32_t count, bool weighted, bool latencyBound) {
    FitShape sh;
    const int slots = nrows > kFitThreads ? 1 : kFitThreads / nrows;
    // Cell blocking (4 cells per thread sharing each basis product) is implemented in the kernel but measured
    // slower on MI355X (p=2, 65536 cells: 296 us vs 240 us): its 64 KB of LDS per workgroup leaves two workgroups
    // per CU to hide the phase barriers, and the kernel is not VALU-bound (52 % VALU-active).  Kept off.
    sh.cellsPerThread = 1;
    (void)count;
    int gmax = slots * sh.cellsPerThread;
    while (gmax > sh.cellsPerThread && fitLdsBytes(degree, gmax, 1) > kFitMaxLdsBytes) gmax -= sh.cellsPerThread;
    if (fitLdsBytes(degree, gmax, 1) > kFitMaxLdsBytes) {  // blocking does not fit: fall back
        sh.cellsPerThread = 1;
        gmax = slots;
        while (gmax > 1 && fitLdsBytes(degree, gmax, 1) > kFitMaxLdsBytes) --gmax;
    }
    // enough workgroups to cover the chip twice before cells are stacked into one workgroup
    int g = sh.cellsPerThread > 1 ? gmax
                                  : (int)std::min<uint32_t>((uint32_t)gmax, std::max<uint32_t>(1, (count + 511) / 512));
    // degree 2 (25 cells fit a workgroup): a round-0-sized launch is fastest at 4 cells per workgroup, big ones at 16
    // (tools/fit_shape_sweep.py: 4096 cells 53 -> 45 us; 32 768 cells 251 -> 211 us)
    if (degree == 2 && sh.cellsPerThread == 1)
        g = (int)std::min<uint32_t>(std::min(gmax, 16), std::max<uint32_t>(1, count <= 4096 ? (count + 1023) / 1024 : (count + 511) / 512));
    // Mesh fields: phase 1 is a chain of dependent BVH-node gathers per sample (measured on a 1.3 M-triangle mesh,
    // 4096 coarse cells: 206 ms with 8 cells per workgroup, 176 / 151 / 128 ms with 4 / 2 / 1) -- many small
    // workgroups keep more waves in flight and shorten the wait for the slowest lane of a chunk.
    if (latencyBound) g = 1;
    if (const char* e = std::getenv("HPSDF_FIT_G")) {  // tuning knobs
        sh.cellsPerThread = 1;
        g = std::max(1, std::min(slots, std::atoi(e)));
        while (g > 1 && fitLdsBytes(degree, g, 1) > kFitMaxLdsBytes) --g;
    }
    sh.cells = g;
    const int nq = 4 * degree + 1;
    const size_t budget = sh.cellsPerThread > 1 ? kFitMaxLdsBytes : kFitChunkLdsBytes;
    sh.planes = nq;
    while (sh.planes > 1 && fitLdsBytes(degree, g, sh.planes) > budget) --sh.planes;
    sh.ldsBytes = fitLdsBytes(degree, g, sh.planes);
    if (weighted) {
        // the sample region is reused for the full coefficient array + 100 FApprox values of every cell
        const int need = coeffCount(degree) + 100;
        const int minPlanes = (need + nq * nq - 1) / (nq * nq);
        sh.planes = std::max(sh.planes, std::min(nq, minPlanes));
        while (sh.cells > 1 && fitLdsBytes(degree, sh.cells, sh.planes) > kFitMaxLdsBytes) --sh.cells;
        sh.ldsBytes = fitLdsBytes(degree, sh.cells, sh.planes);
    }
    return sh;
}

// Mesh fields, first half of a round: F at every sample of every fit, written where fit_kernel<kFieldSamples> reads it
// (FitTask::sampleOff + (i nq + j) nq + k).  A closest-triangle traversal costs anything between a few dozen and tens
// of thousands of steps depending on where the cell lies, so sampling inside the fit kernel (one workgroup per cell)
// left the chip a quarter full behind the expensive cells; here the unit of work is one wave = 64 samples that sit
// next to each other (meshSampleOrder over the whole grid), a workgroup is four of them, the hardware deals them out,
// and without the fit's accumulators twice as many waves fit on a CU.  grid = (ceil(nq^3 / 256), tasks of one degree).
// range == nullptr: grid = (chunks of 256 samples, tasks).  Otherwise grid.y is an upper bound and row y samples task
// range[0] + y if y < range[1] -- the device-side frontier's rounds (frontier.hip), whose task counts the host does not know.
#ifndef HPSDF_MESH_WG
#define HPSDF_MESH_WG 64  // threads of a sampling workgroup (a multiple of 64): the hardware deals out workgroups, so this is the grain of its load balancing
#endif
constexpr int kMeshWg = HPSDF_MESH_WG;
constexpr uint32_t kMeshXcdRun = 128;  // workgroups of 64 samples: runs of 64-256 measure alike; 1 (plain order) and >= 1024 lose 3-7 %
static uint32_t meshXcdRun() {  // HPSDF_MESH_XCD_RUN overrides (experiments); 1 = plain dispatch order
    static const uint32_t v = [] {
        const char* e = std::getenv("HPSDF_MESH_XCD_RUN");
        const long x = e ? std::atol(e) : (long)kMeshXcdRun;
        return (uint32_t)(x < 1 ? 1 : (x > 4096 ? 4096 : x));
    }();
    return v;
}
#ifndef HPSDF_MESH_MIN_WAVES
#define HPSDF_MESH_MIN_WAVES 6  // <= 80 registers, six waves a SIMD (round 6; 88 registers and five waves until then: the kernel is bound by instruction issue with a third of the slots empty, and a sixth wave fills some of them -- Create on the 2.1 M-triangle torus at 1e-6 32.1 -> 30.7 ms, 1.3 M-triangle icosphere 11.4 -> 10.9, at the price of 48 more bytes of scratch)
#endif
__global__ __launch_bounds__(kMeshWg, HPSDF_MESH_MIN_WAVES) void mesh_sample_kernel(const FitTask* __restrict__ tasks, int degree,
                                                          const DeviceTables* __restrict__ T, MeshDev mesh, RootMap rm,
                                                          double* __restrict__ samples, const uint32_t* __restrict__ range,
                                                          uint32_t nTasksArg, uint32_t xcdRun) {
    __shared__ MeshWaveLds sWave[kMeshWg / 64];
    __shared__ double sR[64];
    __shared__ unsigned char sPos[64];
    // Which (task, chunk) this workgroup samples.  Workgroups are dealt round-robin over the 8 XCDs in dispatch order
    // (blocks b and b + 8 share an XCD and its 4 MB L2: MI355X_MICROARCH.md, observed, a speed matter only), and the
    // tasks lie in node order, i.e. along the octree's space-filling curve.  The (task, chunk) list is cut into runs of
    // kMeshXcdRun consecutive entries -- a few neighbouring cells -- and the runs are dealt round-robin over the XCDs:
    // what an XCD has in flight at any time is a handful of compact regions, whose part of the BVH and of the triangle
    // records is all its L2 has to hold (plain blockIdx order: every eighth chunk of an eight times longer stretch;
    // one contiguous eighth of the list per XCD instead keeps the locality but not the balance -- cells far from the
    // surface cost several times more than cells on it).
    const uint32_t gx = gridDim.x, nTasks = range != nullptr ? range[1] : nTasksArg;
    const uint32_t nwg = nTasks * gx, orig = blockIdx.y * gx + blockIdx.x;
    const uint32_t inXcd = orig >> 3;  // position in the sequence of the workgroups that share this one's XCD
    const uint32_t wgid = (((inXcd / xcdRun) << 3) + (orig & 7u)) * xcdRun + inXcd % xcdRun;
    if (wgid >= nwg) return;  // (the grid is rounded up to whole groups of 8 runs, or an upper bound)
    const uint32_t chunk = wgid % gx;
    uint32_t task = wgid / gx;
    if (range != nullptr) task += range[0];
    const int tid = threadIdx.x, nq = 4 * degree + 1, gl = nq * (nq - 1) / 2, total = nq * nq * nq;
    if (tid < nq) sR[tid] = T->roots[gl + tid];
    __syncthreads();
    if (tid < nq) {
        int rank = 0;
        for (int b = 0; b < nq; ++b) rank += sR[b] < sR[tid] ? 1 : 0;
        sPos[rank] = (unsigned char)tid;
    }
    __syncthreads();
    const FitTask& tk = tasks[task];
    const int r = (int)chunk * kMeshWg + tid;
    const bool active = r < total;
    const int rem = meshSampleOrder(active ? r : total - 1, nq, nq, sPos, sPos);
    const int i = rem / (nq * nq), jk = rem - i * nq * nq, j = jk / nq, k = jk - j * nq;
    double w[3];
    const int idx[3] = {i, j, k};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double sc = (double)(tk.bmax[a] - tk.bmin[a]) * 0.5;       // Octree.cpp:1020 sizes() in f32
        const double ce = (double)((tk.bmin[a] + tk.bmax[a]) / 2.0f);     // :1021 center() in f32
        const double u = sR[idx[a]] * sc + ce;                            // :1035-1037
        w[a] = u * rm.bounds[a] + rm.centre[a];                           // :327
    }
    const float mv = meshSignedDistanceWaveQ(mesh, V3{(float)w[0], (float)w[1], (float)w[2]}, active, sWave[tid >> 6]);
    if (active) samples[tk.sampleOff + (uint64_t)rem] = (double)mv;
}

// Mesh::SignedDistanceAtPt(pt) WITHOUT a BVH (Mesh.cpp:42-51 over the linear scan Mesh::ClosestTriangleToPt, :134-159).
// A wave takes one point and one SLICE of the triangles: lane l tests triangles first + l, first + l + 64, ... of the slice,
// keeping the first strictly smaller squared distance (so the lowest index among its own equals); the lanes fold to the
// smallest distance, ties to the lower triangle index, and the wave's winner goes into the point's 64-bit key
// (distance bits << 32 | triangle) by atomicMin -- over all slices that leaves the smallest distance and, among equals, the
// lowest triangle: what the reference's `<` scan from triangle 0 upwards keeps.  The key lives in the point's slot of the
// OUTPUT array (8 bytes, preset to all ones) -- or in scratch of the caller's when the output is host memory mapped into the
// device, where an atomic is a PCIe transaction --; mesh_naive_finish_kernel repeats the winner's closest-point test and
// writes the signed distance.  The slices let a handful of points use the whole chip (one point: 8 ms -> 0.1 ms on 1 M
// triangles); with thousands of points there is one slice and the atomic is one per wave.
// It is the checker of the BVH path on the device (TestBVHQuerying, MeshingUnitTests.cpp:110-138) and O(n) per point.
__global__ __launch_bounds__(256) void mesh_naive_kernel(MeshDev m, const double* __restrict__ xyz, size_t n, unsigned long long* __restrict__ keys,
                                                         uint32_t slices) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // wave = (point, slice), the slices of a point adjacent
    const size_t i = w / slices;
    if (i >= n) return;  // wave-uniform
    const uint32_t slice = (uint32_t)(w % slices);
    const uint32_t per = ((m.nTris + slices - 1u) / slices + 63u) & ~63u;
    const uint32_t first = slice * per, last = first + per < m.nTris ? first + per : m.nTris;
    const int lane = threadIdx.x & 63;
    const V3 pt = {(float)xyz[3 * i], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2]};
    float best = FLT_MAX;
    uint32_t bestTri = 0xFFFFFFFFu;
    const float slack = meshSlack(loadNodeUniform(m.bvh, 0));  // (the face-case tolerance of closestSimplex: the same on every path)
    for (uint32_t t = first + (uint32_t)lane; t < last; t += 64u) {
        V3 q;
        const float4 tp[3] = {m.triPos[3 * (size_t)t], m.triPos[3 * (size_t)t + 1], m.triPos[3 * (size_t)t + 2]};
        closestSimplex(pt, V3{tp[0].x, tp[0].y, tp[0].z}, V3{tp[0].w, tp[1].x, tp[1].y}, V3{tp[1].z, tp[1].w, tp[2].x}, V3{tp[2].y, tp[2].z, tp[2].w}, m.faceTolOfSlack * slack, best, q);
        const float d = sqnorm(pt - q);
        if (d < best) best = d, bestTri = t;
    }
    float wd = best;
    uint32_t wt = bestTri;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(wd, off, 64);
        const uint32_t ot = __shfl_xor(wt, off, 64);
        if (od < wd || (od == wd && ot < wt)) wd = od, wt = ot;
    }
    // (a squared distance that won a `<` against FLT_MAX is a non-negative finite float: its bits order like its value)
    if (lane == 0 && wt != 0xFFFFFFFFu) atomicMin(&keys[i], ((unsigned long long)__float_as_uint(wd) << 32) | wt);
}
__global__ __launch_bounds__(256) void mesh_naive_finish_kernel(MeshDev m, const double* __restrict__ xyz, size_t n, const unsigned long long* keys,
                                                                double* out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = keys[i];
    if (key == ~0ull) {  // no triangle came closer than FLT_MAX (a point that is not finite): the NaN of the other paths
        out[i] = (double)meshNoTriangle();
        return;
    }
    const uint32_t t = (uint32_t)key;
    const V3 pt = {(float)xyz[3 * i], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2]};
    const float slack = meshSlack(m.bvh[0]);
    V3 q;
    const float4 tp[3] = {m.triPos[3 * (size_t)t], m.triPos[3 * (size_t)t + 1], m.triPos[3 * (size_t)t + 2]};
    const int code = closestSimplex(pt, V3{tp[0].x, tp[0].y, tp[0].z}, V3{tp[0].w, tp[1].x, tp[1].y}, V3{tp[1].z, tp[1].w, tp[2].x}, V3{tp[2].y, tp[2].z, tp[2].w}, m.faceTolOfSlack * slack, __builtin_inff(), q);
    const V3 nrm = pseudoNormal(m, t, code);
    const V3 d = pt - q;
    const float sign = dot(nrm, d) > 0.0f ? 1.0f : -1.0f;
    out[i] = (double)(sign * sqrtf(sqnorm(d)));
}

// Mesh::SignedDistanceAtPt(pt, bvh) through the traversal the sampler uses: 64 consecutive points share one walk
// (meshSignedDistanceWaveQ).  Correct for any points -- a wave visits the union of what its lanes need -- and fast when
// neighbours in the array are neighbours in space.
__global__ __launch_bounds__(256) void mesh_eval_wave_kernel(MeshDev m, const double* __restrict__ xyz, size_t n, double* __restrict__ out) {
    __shared__ MeshWaveLds sWave[4];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n;
    const size_t j = active ? i : n - 1;
    const float v = meshSignedDistanceWaveQ(m, V3{(float)xyz[3 * j], (float)xyz[3 * j + 1], (float)xyz[3 * j + 2]}, active, sWave[threadIdx.x >> 6]);
    if (active) out[i] = (double)v;
}

// The same for points in ANY order: the caller's points are visited along a Morton curve over the mesh's surroundings (30-bit
// keys, an index sort), so that the 64 points of a wave are neighbours in space and share most of their walk -- 1 M random points
// of a root box: 6.9 -> 3 ms on a 2.1 M-triangle mesh; every point's value is its own, whatever the order
// (test_full_size_hierarchy_equals_linear_scan_bitwise).  Sets below kMeshEvalSortMin are not worth the sort's launches.
constexpr size_t kMeshEvalSortMin = 4096;
__device__ __forceinline__ uint32_t mortonSpread10(uint32_t v) {  // 10 bits -> every third bit
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(256) void mesh_eval_keys_kernel(MeshDev m, const double* __restrict__ xyz, uint32_t n, uint32_t* __restrict__ keys,
                                                             uint32_t* __restrict__ ids) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const BvhNode& root = m.bvh[0];  // (its two child boxes: the mesh's box; a leaf root keeps the second empty)
    uint32_t key = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const bool two = root.lo1[a] <= root.hi1[a];
        const float lo = two ? fminf(root.lo0[a], root.lo1[a]) : root.lo0[a], hi = two ? fmaxf(root.hi0[a], root.hi1[a]) : root.hi0[a];
        const float ext = fmaxf(hi - lo, 1e-30f);
        // the grid spans the box and as much again on either side: points of a root box around the mesh keep their order too
        const float t = ((float)xyz[3 * (size_t)i + a] - (lo - ext)) / (3.0f * ext);
        const float q = fminf(fmaxf(t, 0.0f), 1.0f) * 1023.0f;  // (NaN -> 0 through fmaxf)
        key |= mortonSpread10((uint32_t)q) << a;
    }
    keys[i] = key;
    ids[i] = i;
}
__global__ __launch_bounds__(256) void mesh_eval_wave_sorted_kernel(MeshDev m, const double* __restrict__ xyz, const uint32_t* __restrict__ ids, size_t n,
                                                                    double* __restrict__ out) {
    __shared__ MeshWaveLds sWave[4];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = i < n;
    const size_t j = ids[active ? i : n - 1];
    const float v = meshSignedDistanceWaveQ(m, V3{(float)xyz[3 * j], (float)xyz[3 * j + 1], (float)xyz[3 * j + 2]}, active, sWave[threadIdx.x >> 6]);
    if (active) out[j] = (double)v;
}

hipError_t launchMeshEvalWave(hipStream_t stream, const FieldDev& f, const double* dXyz, size_t n, double* dOut) {
    if (n == 0) return hipSuccess;
    if (f.kind != kFieldMesh || f.csgOp >= 0) return hipErrorInvalidValue;
    static const bool noSort = std::getenv("HPSDF_MESH_EVAL_NO_SORT") != nullptr;  // measurement knob
    for (size_t first = 0; first < n; first += (size_t)1 << 30) {  // a part's indices fit 32 bits
        const size_t m = std::min<size_t>((size_t)1 << 30, n - first);
        char* block = nullptr;
        size_t tmpBytes = 0;
        bool sorted = false;
        if (!noSort && m >= kMeshEvalSortMin && sortPairsU32(stream, nullptr, tmpBytes, nullptr, nullptr, nullptr, nullptr, m, 30) == hipSuccess) {
            const size_t arr = (m * sizeof(uint32_t) + 255) & ~(size_t)255;
            // stream-ordered scratch: four index arrays and the sort's own; if the pool declines, the points go as they are
            int dev = 0;
            hipMemPool_t pool = hipGetDevice(&dev) == hipSuccess ? meshPool(dev) : nullptr;  // the library's own pool, not the default one
            if (pool && hipMallocFromPoolAsync((void**)&block, 4 * arr + tmpBytes, pool, stream) == hipSuccess) {
                uint32_t *keys = (uint32_t*)block, *keysOut = (uint32_t*)(block + arr), *ids = (uint32_t*)(block + 2 * arr), *idsOut = (uint32_t*)(block + 3 * arr);
                hipLaunchKernelGGL(mesh_eval_keys_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, f.mesh, dXyz + 3 * first, (uint32_t)m, keys, ids);
                if (sortPairsU32(stream, block + 4 * arr, tmpBytes, keys, keysOut, ids, idsOut, m, 30) == hipSuccess) {
                    hipLaunchKernelGGL(mesh_eval_wave_sorted_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, f.mesh, dXyz + 3 * first, idsOut, m,
                                       dOut + first);
                    sorted = true;
                }
                (void)hipFreeAsync(block, stream);
            } else {
                (void)hipGetLastError();
            }
        }
        if (!sorted) hipLaunchKernelGGL(mesh_eval_wave_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, f.mesh, dXyz + 3 * first, m, dOut + first);
    }
    return hipGetLastError();
}

// hpsdfAcosf of the floats whose bit patterns are first, first + stride, ...: the device half of the acosf parity test
__global__ __launch_bounds__(256) void acosf_selftest_kernel(uint32_t first, uint32_t stride, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = hpsdfAcosf(__uint_as_float(first + (uint32_t)i * stride));
}
hipError_t launchAcosfSelftest(hipStream_t stream, uint32_t first, uint32_t stride, size_t n, float* dOut) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(acosf_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, first, stride, n, dOut);
    return hipGetLastError();
}

hipError_t launchMeshNaive(hipStream_t stream, const FieldDev& f, const double* dXyz, size_t n, double* dOut, unsigned long long* dKeys) {
    if (n == 0) return hipSuccess;
    if (f.kind != kFieldMesh || f.csgOp >= 0) return hipErrorInvalidValue;
    for (size_t first = 0; first < n; first += (size_t)1 << 28) {  // grid.x stays below 2^31
        const size_t m = std::min<size_t>((size_t)1 << 28, n - first);
        // enough waves to fill the chip: 256 CUs x 32 wave slots; a slice keeps at least 1024 triangles
        const uint64_t byWaves = (8192 + m - 1) / m, byTris = std::max<uint64_t>(1, f.mesh.nTris / 1024);
        const uint32_t slices = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(byWaves, byTris), 4096));
        unsigned long long* keys = dKeys ? dKeys + first : reinterpret_cast<unsigned long long*>(dOut + first);
        hipError_t e = hipMemsetAsync(keys, 0xFF, m * sizeof(double), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(mesh_naive_kernel, dim3((unsigned)((m * slices + 3) / 4)), dim3(256), 0, stream, f.mesh, dXyz + 3 * first, m, keys, slices);
        hipLaunchKernelGGL(mesh_naive_finish_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, f.mesh, dXyz + 3 * first, m, keys,
                           dOut + first);
    }
    return hipGetLastError();
}

// MeshDev::triPos: per triangle one 48-byte record -- the nine vertex coordinates and the unnormalised normal
// cross(b - a, c - a) -- gathered once per mesh: a closest-point test is three 16-byte loads instead of index -> vertex chains
__global__ __launch_bounds__(256) void mesh_tripos_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ tris,
                                                          uint64_t nTris, float4* __restrict__ triPos) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nTris) return;
    const uint32_t ia = tris[3 * t], ib = tris[3 * t + 1], ic = tris[3 * t + 2];
    const V3 a = {verts[3 * (size_t)ia], verts[3 * (size_t)ia + 1], verts[3 * (size_t)ia + 2]};
    const V3 b = {verts[3 * (size_t)ib], verts[3 * (size_t)ib + 1], verts[3 * (size_t)ib + 2]};
    const V3 c = {verts[3 * (size_t)ic], verts[3 * (size_t)ic + 1], verts[3 * (size_t)ic + 2]};
    const V3 n = cross(b - a, c - a);
    triPos[3 * t] = make_float4(a.x, a.y, a.z, b.x);
    triPos[3 * t + 1] = make_float4(b.y, b.z, c.x, c.y);
    triPos[3 * t + 2] = make_float4(c.z, n.x, n.y, n.z);
}

// MeshDev::triPre: per leaf slot the data of the lower-bound test (triLowerBound2) and the triangle's index: the unit normal n,
// a unit vector u along the longest edge, the centre g of the triangle's bounding rectangle in the (u, n x u) frame and the
// rectangle's half-extents.  The bound is valid for ANY orthonormal n, u as long as every point x of the triangle has
// |n . (x - g)| <= e, |u . (x - g)| <= hu and sqrt(|x - g|^2 - (n . (x - g))^2 - (u . (x - g))^2) <= hv -- all three are convex in x,
// so the vertices decide, and all three are MEASURED here against the g that is stored, with the arithmetic of the test.  When e
// is not negligible (slivers, whose cross product cancels) or the frame is not orthonormal to 1e-6, n and u are set to zero and hv
// to the largest distance of a vertex from g, which turns the test into the ball's bound |p - g| - rho.  What is left of e
// (<= 4e-7 of the mesh's scale) and of the frame's rounding is covered by the caller's slack (2e-6 of that scale: meshSlack).
__global__ __launch_bounds__(256) void mesh_tripre_kernel(const float* __restrict__ verts, const uint32_t* __restrict__ tris,
                                                          const uint32_t* __restrict__ slotTri, uint64_t nTris, float4* __restrict__ triPre) {
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= nTris) return;
    const uint32_t t = slotTri ? slotTri[s] : (uint32_t)s;
    const uint32_t ia = tris[3 * (size_t)t], ib = tris[3 * (size_t)t + 1], ic = tris[3 * (size_t)t + 2];
    const V3 a = {verts[3 * (size_t)ia], verts[3 * (size_t)ia + 1], verts[3 * (size_t)ia + 2]};
    const V3 b = {verts[3 * (size_t)ib], verts[3 * (size_t)ib + 1], verts[3 * (size_t)ib + 2]};
    const V3 c = {verts[3 * (size_t)ic], verts[3 * (size_t)ic + 1], verts[3 * (size_t)ic + 2]};
    const V3 ab = b - a, ac = c - a, bc = c - b;
    const float lab = sqnorm(ab), lac = sqnorm(ac), lbc = sqnorm(bc);
    const V3 n = cross(ab, ac);
    const float inf = __builtin_inff();
    // the frame: n, u along the longest edge, v = n x u; the rectangle's centre from the vertices' (u, v) ranges about a
    const V3 le = lab >= lac && lab >= lbc ? ab : (lac >= lbc ? ac : bc);
    const float len = sqrtf(sqnorm(n)), ll = sqrtf(sqnorm(le));
    V3 nh = {0.0f, 0.0f, 0.0f}, uh = {0.0f, 0.0f, 0.0f};
    const float third = 1.0f / 3.0f;
    V3 g = third * (a + (b + c));
    bool framed = len > 0.0f && len < inf && ll > 0.0f && ll < inf;
    if (framed) {
        nh = (1.0f / len) * n;
        uh = (1.0f / ll) * le;
        const V3 vh = cross(nh, uh);
        const float ub = dot(uh, ab), uc = dot(uh, ac), vb = dot(vh, ab), vc = dot(vh, ac);  // (vertex a sits at (0, 0))
        const float um = 0.5f * (fminf(0.0f, fminf(ub, uc)) + fmaxf(0.0f, fmaxf(ub, uc)));
        const float vm = 0.5f * (fminf(0.0f, fminf(vb, vc)) + fmaxf(0.0f, fmaxf(vb, vc)));
        g = a + (um * uh + vm * vh);
        framed = fabsf(sqnorm(nh) - 1.0f) <= 1e-6f && fabsf(sqnorm(uh) - 1.0f) <= 1e-6f && fabsf(dot(nh, uh)) <= 1e-6f;
    }
    const V3 da = a - g, db = b - g, dc = c - g;
    const float ra = sqnorm(da), rb = sqnorm(db), rcq = sqnorm(dc);
    const float rho = sqrtf(fmaxf(ra, fmaxf(rb, rcq))) * 1.00001f + 1e-30f;
    // the scale the caller's slack is proportional to is at least this (the mesh's extent or its largest coordinate)
    const float scale = fmaxf(rho, fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fabsf(g.z)));
    float hu = 0.0f, hv = rho;
    if (framed) {
        const float sa = dot(nh, da), sb = dot(nh, db), sc = dot(nh, dc);
        const float ua = dot(uh, da), ub = dot(uh, db), uc = dot(uh, dc);
        const float e = fmaxf(fabsf(sa), fmaxf(fabsf(sb), fabsf(sc)));
        const float wa = sqrtf(fmaxf(ra - sa * sa - ua * ua, 0.0f)), wb = sqrtf(fmaxf(rb - sb * sb - ub * ub, 0.0f)),
                    wc = sqrtf(fmaxf(rcq - sc * sc - uc * uc, 0.0f));
        // The rectangle holds the TRIANGLE (measured above, with allowances for its own rounding).  What the closest-point routine
        // returns for the triangle lies within a quarter of the traversal's slack of it: closestSimplex does not take a face-case
        // point farther outside than that (until round 4 the rectangle was widened by a "play" of 1e-6 longest edge / sin(smallest
        // angle) instead, an estimate of how far the reference's barycentric quotients can throw q: it did not hold on needles).
        hu = fmaxf(fabsf(ua), fmaxf(fabsf(ub), fabsf(uc))) * 1.00001f + 4e-7f * scale;
        // (hv also takes 1e-3 hu: the test forms the in-plane distance across u as sqrt(|d|^2 - s^2 - a^2), whose cancellation leaves up to
        // sqrt(2 ulp) |d| = 3.5e-4 |d| where the true value is nearly zero -- beside a needle that is more than its width; past ~3 hu from
        // g the excess is below 1e-6 of the bound itself, which rejectBound's factor covers)
        hv = fmaxf(wa, fmaxf(wb, wc)) * 1.00001f + 4e-7f * scale + 1e-3f * hu;
        framed = e <= 4e-7f * scale && hu < inf && hv < inf;
    }
    if (!framed || !(rho < inf)) {  // (non-finite input: the bound degenerates to "always passes" via NaN)
        nh = V3{0.0f, 0.0f, 0.0f}, uh = V3{0.0f, 0.0f, 0.0f};
        hu = 0.0f, hv = rho;
    }
    triPre[3 * s] = make_float4(g.x, g.y, g.z, hu);
    triPre[3 * s + 1] = make_float4(nh.x, nh.y, nh.z, hv);
    triPre[3 * s + 2] = make_float4(uh.x, uh.y, uh.z, __uint_as_float(t));
}

hipError_t launchMeshTriPos(hipStream_t stream, const float* dVerts, const uint32_t* dTris, uint64_t nTris, float* dTriPos,
                            const uint32_t* dSlotTri, float* dTriPre) {
    if (nTris == 0) return hipSuccess;
    if (dTriPos)
        hipLaunchKernelGGL(mesh_tripos_kernel, dim3((unsigned)((nTris + 255) / 256)), dim3(256), 0, stream, dVerts, dTris, nTris,
                           reinterpret_cast<float4*>(dTriPos));
    if (dTriPre)
        hipLaunchKernelGGL(mesh_tripre_kernel, dim3((unsigned)((nTris + 255) / 256)), dim3(256), 0, stream, dVerts, dTris, dSlotTri, nTris,
                           reinterpret_cast<float4*>(dTriPre));
    return hipGetLastError();
}

hipError_t launchMeshSample(hipStream_t stream, const FitTask* dTasks, uint32_t nTasks, int degree, const DeviceTables* dTables,
                            const FieldDev& field, const RootMap& rm, double* dSamples) {
    if (nTasks == 0) return hipSuccess;
    if (degree < 1 || degree > 12 || field.kind != kFieldMesh) return hipErrorInvalidValue;
    const int nq = 4 * degree + 1;
    const unsigned gx = (unsigned)((nq * nq * nq + kMeshWg - 1) / kMeshWg);
    for (uint32_t first = 0; first < nTasks; first += 65535u) {
        const uint32_t n = nTasks - first < 65535u ? nTasks - first : 65535u;
        // (grid rounded up to whole groups of 8 runs of the XCD interleave)
        const uint32_t run = meshXcdRun(), wgs = (n * gx + 8u * run - 1u) / (8u * run) * (8u * run);
        hipLaunchKernelGGL(mesh_sample_kernel, dim3(gx, (wgs + gx - 1u) / gx), dim3(kMeshWg), 0, stream, dTasks + first, degree, dTables, field.mesh, rm,
                           dSamples, (const uint32_t*)nullptr, n, run);
    }
    return hipGetLastError();
}

// the same over the device-written task range dRange = {first task, count} of dTasks; maxTasks bounds the count
hipError_t launchMeshSampleRange(hipStream_t stream, const FitTask* dTasks, const uint32_t* dRange, uint32_t maxTasks, int degree,
                                 const DeviceTables* dTables, const FieldDev& field, const RootMap& rm, double* dSamples) {
    if (degree < 1 || degree > 12 || field.kind != kFieldMesh || maxTasks == 0 || maxTasks > 65535u) return hipErrorInvalidValue;
    const int nq = 4 * degree + 1;
    const unsigned gx = (unsigned)((nq * nq * nq + kMeshWg - 1) / kMeshWg);
    const uint32_t run = meshXcdRun();
    const unsigned gy = maxTasks + (8u * run + gx - 1u) / gx;  // whole groups of 8 runs of the XCD interleave past the last task
    if (gy > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mesh_sample_kernel, dim3(gx, gy), dim3(kMeshWg), 0, stream, dTasks, degree, dTables, field.mesh, rm, dSamples, dRange, 0u, run);
    return hipGetLastError();
}

// weighted builds: |mean FApprox| of every fit of the blocks (full coefficient arrays at FitTask::outOff)
hipError_t launchFitWeight(hipStream_t stream, const FitBlock* dBlocks, uint32_t nBlocks, size_t ldsBytes, const FitTask* dTasks,
                           const double* dArena, double* dMeans, const DeviceTables* dTables, const uint32_t* dCount) {
    if (nBlocks == 0) return hipSuccess;
    hipLaunchKernelGGL(fit_weight_kernel, dim3(nBlocks), dim3(kFitThreads), ldsBytes, stream, dBlocks, dTasks, dArena, dMeans, dTables, dCount);
    return hipGetLastError();
}

template <int KIND, bool CSG, bool LEFT>
static void launchFitT(hipStream_t stream, int degree, int cellsPerThread, const FitBlock* dBlocks, uint32_t nBlocks,
                       size_t ldsBytes, const FitTask* dTasks, double* dArena, double* dErrs, double* dMeans,
                       const DeviceTables* dTables, const FieldDev& field, const RootMap& rm, const uint32_t* dRange) {
#define HPSDF_FIT_LAUNCH(D, RR)                                                                                       \
    hipLaunchKernelGGL((fit_kernel<KIND, CSG, D, RR, LEFT>), dim3(nBlocks), dim3(kFitThreads), ldsBytes, stream, dBlocks, \
                       dTasks, dArena, dErrs, dMeans, dTables, field, rm, dRange)
#define HPSDF_FIT_CASE(D)       \
    case D:                     \
        HPSDF_FIT_LAUNCH(D, 1); \
        break;
    if (cellsPerThread != 1 && cellsPerThread != 4) cellsPerThread = 1;
    switch (degree) {
        HPSDF_FIT_CASE(2)
        HPSDF_FIT_CASE(3)
        HPSDF_FIT_CASE(4)
        HPSDF_FIT_CASE(5)
        HPSDF_FIT_CASE(6)
        HPSDF_FIT_CASE(7)
        HPSDF_FIT_CASE(8)
        default:
            HPSDF_FIT_LAUNCH(0, 1);
    }
#undef HPSDF_FIT_CASE
#undef HPSDF_FIT_LAUNCH
}

// FN<kind, csg wrapper, reduction order>(...): the instantiation for a FieldDev
#define HPSDF_DISPATCH_FIELD_ORDER(FN, K, C, field, ...)  \
    do {                                                  \
        if ((field).leftAssoc) FN<K, C, true>(__VA_ARGS__); \
        else FN<K, C, false>(__VA_ARGS__);                \
    } while (0)
#define HPSDF_DISPATCH_FIELD(FN, field, ...)                                                        \
    do {                                                                                            \
        const bool csg_ = (field).csgOp >= 0;                                                       \
        switch ((field).kind) {                                                                     \
            case kFieldAnalytic:                                                                    \
                if (csg_) HPSDF_DISPATCH_FIELD_ORDER(FN, kFieldAnalytic, true, field, __VA_ARGS__); \
                else HPSDF_DISPATCH_FIELD_ORDER(FN, kFieldAnalytic, false, field, __VA_ARGS__);     \
                break;                                                                              \
            case kFieldSamples:                                                                     \
                if (csg_) HPSDF_DISPATCH_FIELD_ORDER(FN, kFieldSamples, true, field, __VA_ARGS__);  \
                else HPSDF_DISPATCH_FIELD_ORDER(FN, kFieldSamples, false, field, __VA_ARGS__);      \
                break;                                                                              \
            default:                                                                                \
                if (csg_) HPSDF_DISPATCH_FIELD_ORDER(FN, kFieldMesh, true, field, __VA_ARGS__);     \
                else HPSDF_DISPATCH_FIELD_ORDER(FN, kFieldMesh, false, field, __VA_ARGS__);         \
                break;                                                                              \
        }                                                                                           \
    } while (0)

// One launch per degree: `degree` selects the compile-time-specialised kernel (0 = any; the blocks then carry
// their own degree).
template <int KIND, bool CSG, bool LEFT>
static void launchFitMultiT(hipStream_t stream, const FitBlock* dBlocks, uint32_t maxBlocks, size_t ldsBytes, const FitTask* dTasks,
                            double* dArena, double* dErrs, const DeviceTables* dTables, const FieldDev& field, const RootMap& rm,
                            const uint32_t* dCount) {
    hipLaunchKernelGGL((fit_multi_kernel<KIND, CSG, LEFT>), dim3(maxBlocks), dim3(kFitThreads), ldsBytes, stream, dBlocks, dTasks, dArena, dErrs,
                       dTables, field, rm, dCount, maxBlocks);
}
// every block of dBlocks[0 .. *dCount) -- or [0 .. maxBlocks) when dCount is null --, whatever its degree, in one launch;
// ldsBytes: the largest any of them needs
hipError_t launchFitMulti(hipStream_t stream, const FitBlock* dBlocks, uint32_t maxBlocks, size_t ldsBytes, const FitTask* dTasks,
                          double* dArena, double* dErrs, const DeviceTables* dTables, const FieldDev& field, const RootMap& rm,
                          const uint32_t* dCount) {
    if (maxBlocks == 0) return hipSuccess;
    if (ldsBytes > kFitMaxLdsBytes) return hipErrorInvalidValue;
    HPSDF_DISPATCH_FIELD(launchFitMultiT, field, stream, dBlocks, maxBlocks, ldsBytes, dTasks, dArena, dErrs, dTables, field, rm, dCount);
    return hipGetLastError();
}

hipError_t launchFit(hipStream_t stream, int degree, int cellsPerThread, const FitBlock* dBlocks, uint32_t nBlocks,
                     size_t ldsBytes, const FitTask* dTasks, double* dArena, double* dErrs, double* dMirror,
                     const DeviceTables* dTables, const FieldDev& field, const RootMap& rm, const uint32_t* dRange) {
    if (nBlocks == 0) return hipSuccess;
    if (ldsBytes > kFitMaxLdsBytes) return hipErrorInvalidValue;
    HPSDF_DISPATCH_FIELD(launchFitT, field, stream, degree, cellsPerThread, dBlocks, nBlocks, ldsBytes, dTasks, dArena,
                         dErrs, dMirror, dTables, field, rm, dRange);
    return hipGetLastError();
}

static unsigned gridFor(size_t n) {
    size_t blocks = (n + 255) / 256;
    static const size_t capEnv = [] { const char* e = std::getenv("HPSDF_QUERY_GRID"); return e ? (size_t)std::max(1, std::atoi(e)) : (size_t)0; }();  // tuning knob
    const size_t cap = capEnv ? capEnv : kQueryMaxGrid;  // grid-stride beyond this (the headline kernel: 512 ... 39 063 workgroups for 10 M points measured, flat from 4096 up)
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

// Query (dGrad == nullptr) or QueryWithGradient.  dDeferCount: 2 * kQueryMaxGrid + 1 words (counts, then their scan); dDeferIdx: n + 256 *
// kQueryMaxGrid slots (both only touched for trees with leaves of degree > 3).  n < 2^32 (the caller splits larger
// batches).
hipError_t launchQuery(hipStream_t stream, const TreeDev& t, const DeviceTables* dTables, const double* dXyz, size_t n,
                       double* dOut, double* dGrad, bool allInline, uint32_t* dDeferCount, uint32_t* dDeferIdx) {
    if (n == 0) return hipSuccess;
    if (dGrad && n <= kQueryFewPoints) {
        hipLaunchKernelGGL(query_grad_few_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, t, dTables, dXyz, (uint32_t)n, dOut,
                           dGrad);
        return hipGetLastError();
    }
    if (!dGrad && n <= kQueryFewPoints) {
        const dim3 fgrid((unsigned)((n + 63) / 64)), fblock(64);
        if (t.maxDegree <= 3)
            hipLaunchKernelGGL((query_few_kernel<3>), fgrid, fblock, 0, stream, t, dTables, dXyz, (uint32_t)n, dOut);
        else if (t.maxDegree <= 5)
            hipLaunchKernelGGL((query_few_kernel<5>), fgrid, fblock, 0, stream, t, dTables, dXyz, (uint32_t)n, dOut);
        else
            hipLaunchKernelGGL((query_few_kernel<12>), fgrid, fblock, 0, stream, t, dTables, dXyz, (uint32_t)n, dOut);
        return hipGetLastError();
    }
    const dim3 grid(gridFor(n)), block(256);
    if (allInline && dGrad && std::getenv("HPSDF_QUERY_GRAD_GENERAL") == nullptr) {
        if (t.topDepth == 4)
            hipLaunchKernelGGL((query_grad_kernel<4>), grid, block, 0, stream, t, dTables, dXyz, n, dOut, dGrad);
        else
            hipLaunchKernelGGL((query_grad_kernel<0>), grid, block, 0, stream, t, dTables, dXyz, n, dOut, dGrad);
        return hipGetLastError();
    }
    if (allInline && !dGrad) {
        const char* e = std::getenv("HPSDF_QUERY_DEDUPE");  // tuning knob; default on
        const bool dedupe = !(e && e[0] == '0');
        if (t.topDepth == 4) {
            if (dedupe)
                hipLaunchKernelGGL((query_kernel<4, true>), grid, block, 0, stream, t, dXyz, n, dOut);
            else
                hipLaunchKernelGGL((query_kernel<4, false>), grid, block, 0, stream, t, dXyz, n, dOut);
        } else {
            if (dedupe)
                hipLaunchKernelGGL((query_kernel<0, true>), grid, block, 0, stream, t, dXyz, n, dOut);
            else
                hipLaunchKernelGGL((query_kernel<0, false>), grid, block, 0, stream, t, dXyz, n, dOut);
        }
        return hipGetLastError();
    }
    bool defer = t.maxDegree > 3;
    const bool big = !dGrad && t.topDepth == 4 && std::getenv("HPSDF_QUERY_NO_LDSTOP") == nullptr;
    const unsigned tile = big ? 1024u : 256u;
    const size_t nTiles = (n + tile - 1) / tile;
    // The 16-wave workgroups stage the 32 KB thin table (and the basis tables) into LDS before their first tile: with 2048 of them -- eight
    // generations on 256 CUs, five tiles each -- that start was a tenth of the kernel (207 -> 194 us for 10 M points on union3 @ 1e-7 with
    // one workgroup a CU, 39 tiles each).  HPSDF_QUERY_GENERAL_WGS / _GRID: tuning knobs.
    static const unsigned bigWgs = [] { const char* e = std::getenv("HPSDF_QUERY_GENERAL_WGS"); return e ? (unsigned)std::max(1, std::atoi(e)) : 256u; }();
    static const unsigned smallWgs = [] { const char* e = std::getenv("HPSDF_QUERY_GENERAL_GRID"); return e ? (unsigned)std::max(1, std::min((int)kQueryMaxGrid, std::atoi(e))) : kQueryMaxGrid; }();
    const unsigned nWg = (unsigned)std::min<size_t>(nTiles, big ? bigWgs : smallWgs);
    const uint32_t tilesPerWg = (uint32_t)((nTiles + nWg - 1) / nWg);  // tiles b, b + G, ... of workgroup b
    const dim3 ggrid(nWg);
    const size_t lds = queryGeneralLdsBytes(big ? 16 : 4, big);
    if (big) {
        // > 64 KB of dynamic LDS is opt-in, per function and device
        const void* fn = defer ? (const void*)query_general_lds_kernel<true> : (const void*)query_general_lds_kernel<false>;
        const hipError_t ae = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ae != hipSuccess) return ae;
#ifdef HPSDF_QUERY_LAB_BUILD
        // lib/libhpsdf_lab.so only (build.py --lab; tools/query_general_floor.py): a link of the chain taken out (queryGeneralBody's LAB).
        // The values are then NOT the tree's, which is why the production library neither reads the variable nor holds these kernels.
        const char* labEnv = std::getenv("HPSDF_QUERY_LAB");
        const int lab = labEnv && dDeferCount && dDeferIdx ? std::atoi(labEnv) : 0;  // (the LAB kernels are DEFER instantiations: they need the lists)
#else
        constexpr int lab = 0;
#endif
        if (lab >= 1 && lab <= 4) {
#ifdef HPSDF_QUERY_LAB_BUILD
            const void* lf = lab == 1 ? (const void*)query_general_lds_kernel<true, 1> : lab == 2 ? (const void*)query_general_lds_kernel<true, 2>
                           : lab == 3 ? (const void*)query_general_lds_kernel<true, 3> : (const void*)query_general_lds_kernel<true, 4>;
            const hipError_t le = hipFuncSetAttribute(lf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (le != hipSuccess) return le;
            switch (lab) {
                case 1: hipLaunchKernelGGL((query_general_lds_kernel<true, 1>), ggrid, dim3(1024), lds, stream, t, dTables, dXyz, n, dOut, tilesPerWg, dDeferCount, dDeferIdx); break;
                case 2: hipLaunchKernelGGL((query_general_lds_kernel<true, 2>), ggrid, dim3(1024), lds, stream, t, dTables, dXyz, n, dOut, tilesPerWg, dDeferCount, dDeferIdx); break;
                case 3: hipLaunchKernelGGL((query_general_lds_kernel<true, 3>), ggrid, dim3(1024), lds, stream, t, dTables, dXyz, n, dOut, tilesPerWg, dDeferCount, dDeferIdx); break;
                default: hipLaunchKernelGGL((query_general_lds_kernel<true, 4>), ggrid, dim3(1024), lds, stream, t, dTables, dXyz, n, dOut, tilesPerWg, dDeferCount, dDeferIdx); break;
            }
#endif
        } else if (defer && t.maxDegree <= 5 && std::getenv("HPSDF_QUERY_TWO_PASS") == nullptr) {
            // degrees 4 and 5 are finished by the workgroup that met them (DEEP): one launch
            const hipError_t fe = hipFuncSetAttribute((const void*)query_general_lds_kernel<true, 0, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (fe != hipSuccess) return fe;
            hipLaunchKernelGGL((query_general_lds_kernel<true, 0, 5>), ggrid, dim3(1024), lds, stream, t, dTables, dXyz, n, dOut, tilesPerWg, dDeferCount, dDeferIdx);
            defer = false;
        } else if (defer)
            hipLaunchKernelGGL((query_general_lds_kernel<true>), ggrid, dim3(1024), lds, stream, t, dTables, dXyz, n, dOut, tilesPerWg, dDeferCount, dDeferIdx);
        else
            hipLaunchKernelGGL((query_general_lds_kernel<false>), ggrid, dim3(1024), lds, stream, t, dTables, dXyz, n, dOut, tilesPerWg, dDeferCount, dDeferIdx);
    } else {
#ifdef HPSDF_QUERY_LAB_BUILD  // (lib/libhpsdf_lab.so only: the gradient kernel with a link taken out -- values NOT the tree's)
        {
            const char* labEnv = std::getenv("HPSDF_QUERY_LAB");
            const int glab = labEnv && dGrad && dDeferCount && dDeferIdx && t.topDepth == 4 ? std::atoi(labEnv) : 0;
#define HPSDF_GRAD_LAB(L) case L: hipLaunchKernelGGL((query_general_grad_kernel<4, true, L>), ggrid, block, lds, stream, t, dTables, dXyz, n, dOut, dGrad, tilesPerWg, dDeferCount, dDeferIdx); return hipGetLastError();
            switch (glab) {
                HPSDF_GRAD_LAB(1) HPSDF_GRAD_LAB(2) HPSDF_GRAD_LAB(3) HPSDF_GRAD_LAB(4) HPSDF_GRAD_LAB(5) HPSDF_GRAD_LAB(6)
                default: break;
            }
#undef HPSDF_GRAD_LAB
        }
#endif
#define HPSDF_QUERY_GENERAL(TOPD, DF)                                                                               \
    do {                                                                                                            \
        if (dGrad)                                                                                                  \
            hipLaunchKernelGGL((query_general_grad_kernel<TOPD, DF>), ggrid, block, lds, stream, t, dTables, dXyz, n, dOut, \
                               dGrad, tilesPerWg, dDeferCount, dDeferIdx);                                          \
        else                                                                                                        \
            hipLaunchKernelGGL((query_general_kernel<TOPD, DF>), ggrid, block, lds, stream, t, dTables, dXyz, n, dOut,  \
                               tilesPerWg, dDeferCount, dDeferIdx);                                                 \
    } while (0)
#define HPSDF_QUERY_GENERAL_T(TOPD)         \
    do {                                    \
        if (defer)                          \
            HPSDF_QUERY_GENERAL(TOPD, true);  \
        else                                \
            HPSDF_QUERY_GENERAL(TOPD, false); \
    } while (0)
        if (t.topDepth == 4)
            HPSDF_QUERY_GENERAL_T(4);
        else
            HPSDF_QUERY_GENERAL_T(0);
#undef HPSDF_QUERY_GENERAL_T
#undef HPSDF_QUERY_GENERAL
    }
    if (defer) {
        // the per-workgroup lists are short and ragged: scan their lengths, then walk their concatenation densely
        uint32_t* dOffsets = dDeferCount + kQueryMaxGrid;
        hipLaunchKernelGGL(defer_scan_kernel, dim3(1), dim3(1024), 0, stream, dDeferCount, nWg, dOffsets);
        const dim3 dgrid(1024u);  // (grid-stride over the scanned lists: the lists' number, nWg, does not bound it)
        const uint32_t slotsPerWg = tilesPerWg * (tile / 256u);  // deferredPoint() counts in runs of 256
        if (dGrad && t.maxDegree <= 5)
            hipLaunchKernelGGL((query_grad_deep_kernel<true>), dgrid, block, 0, stream, t, dTables, dXyz, dOut, dGrad, slotsPerWg, nWg,
                               dOffsets, dDeferIdx);
        else if (dGrad)
            hipLaunchKernelGGL((query_grad_deep_kernel<false>), dgrid, block, 0, stream, t, dTables, dXyz, dOut, dGrad, slotsPerWg, nWg,
                               dOffsets, dDeferIdx);
        else if (t.maxDegree <= 5)
            hipLaunchKernelGGL((query_deep_kernel<5>), dgrid, block, 0, stream, t, dTables, dXyz, dOut, slotsPerWg, nWg, dOffsets, dDeferIdx);
        else
            hipLaunchKernelGGL((query_deep_kernel<12>), dgrid, block, 0, stream, t, dTables, dXyz, dOut, slotsPerWg, nWg, dOffsets, dDeferIdx);
    }
    return hipGetLastError();
}

hipError_t launchQueryRay(hipStream_t stream, const TreeDev& t, const DeviceTables* dTables, const double* dOrigins,
                          const double* dDirs, const double* dTMax, size_t n, uint8_t* dHit, double* dT) {
    if (n == 0) return hipSuccess;
    // the evaluation code for the degrees the tree does not contain is left out (registers, no scratch)
    if (t.maxDegree <= 3)
        hipLaunchKernelGGL((query_ray_kernel<3>), dim3(gridFor(n)), dim3(256), 0, stream, t, dTables, dOrigins, dDirs, dTMax, n, dHit, dT);
    else if (t.maxDegree <= 5)
        hipLaunchKernelGGL((query_ray_kernel<5>), dim3(gridFor(n)), dim3(256), 0, stream, t, dTables, dOrigins, dDirs, dTMax, n, dHit, dT);
    else
        hipLaunchKernelGGL((query_ray_kernel<12>), dim3(gridFor(n)), dim3(256), 0, stream, t, dTables, dOrigins, dDirs, dTMax, n, dHit, dT);
    return hipGetLastError();
}

hipError_t launchSlicePoints(hipStream_t stream, double c, float minX, float minY, float step, uint32_t nSamples,
                             double* dXyz) {
    if (nSamples == 0) return hipSuccess;
    hipLaunchKernelGGL(slice_points_kernel, dim3(gridFor((size_t)nSamples * nSamples)), dim3(256), 0, stream, c, minX, minY,
                       step, nSamples, dXyz);
    return hipGetLastError();
}

template <int KIND, bool CSG, bool LEFT>
static void launchFieldT(hipStream_t stream, const FieldDev& f, const DeviceTables* dTables, const double* dXyz,
                         size_t n, double* dOut) {
    hipLaunchKernelGGL((field_kernel<KIND, CSG, LEFT>), dim3(gridFor(n)), dim3(256), 0, stream, f, dTables, dXyz, n, dOut);
}

hipError_t launchFieldEval(hipStream_t stream, const FieldDev& f, const DeviceTables* dTables, const double* dXyz,
                           size_t n, double* dOut) {
    if (n == 0) return hipSuccess;
    HPSDF_DISPATCH_FIELD(launchFieldT, f, stream, f, dTables, dXyz, n, dOut);
    return hipGetLastError();
}

hipError_t launchPack(hipStream_t stream, const PackItem* dItems, uint32_t nItems, const double* dArena, double* dOut) {
    if (nItems == 0) return hipSuccess;
    const unsigned blocks = (nItems + 3) / 4;  // 4 waves per block
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, stream, dItems, nItems, dArena, dOut);
    return hipGetLastError();
}

// Mesh::SignedDistanceAtPt(pt, bvh) for a few points, on the calling thread: `hm` holds HOST copies of the field's arrays (capi.cpp
// keeps them with the field after the first such call).  The per-point stack traversal of meshSignedDistance, compiled for the host
// from the very statements the device runs: the same bits as every device path (tests/test_gpu_parity.py).
void meshEvalHostPoints(const MeshDev& hm, const double* xyz, size_t n, double* out) {
    // The previous answer of this thread is tried first (the `hint` of meshSignedDistance: a search that starts with a tight bound visits
    // a fraction of the nodes; the answer does not depend on it).  Successive one-point calls of a thread are what a per-sample SDF lambda
    // makes: neighbouring samples of one cell.
    static thread_local uint32_t hint = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i)
        out[i] = (double)meshSignedDistance(hm, V3{(float)xyz[3 * i], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2]}, hint);
}

}  // namespace hpsdf
