/*
 * hp_oracle_mesh.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the reference's mesh signed-distance field, the callback
 * F of configs 3-5.  The path restated is the reference's own
 * O(n) cross-check, Mesh::SignedDistanceAtPt(pt) (Source/Meshing/Mesh.cpp:42-51)
 * which the reference test compares against the BVH path
 * (Source/Tests/MeshingUnitTests.cpp:110-138): closest triangle by linear scan
 * with a strict '<' on squared distance, Ericson closest-simplex
 * classification with EPSILON_F32 guards, angle-weighted pseudo-normal sign.
 * All arithmetic is f32.  Eigen's fixed-size-3 reductions associate as
 * a0 + (a1 + a2); that order is kept (Eigen itself is absent here and
 * unpinned upstream: parity of this file is pinned only by the reference's
 * 1e-6 BVH-vs-naive tolerance and by closed-form meshes in tests/).
 */
#include "hp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPSILON_F32 0.000001f /* Include/Utility/Literals.h:13 */

struct ora_mesh {
    uint64_t nverts, ntris;
    float* verts;        /* Mesh.h:77 vertices */
    uint64_t* tris;      /* Mesh.h:75 triIndices */
    uint64_t* halfEdges; /* Mesh.h:74 */
};

typedef struct {
    float x, y, z;
} v3;
static v3 v3_sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static v3 v3_add(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static v3 v3_scale(float s, v3 a) { v3 r = {s * a.x, s * a.y, s * a.z}; return r; }
static float v3_dot(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static v3 v3_cross(v3 a, v3 b) {
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static float v3_sqnorm(v3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
static v3 v3_normalized(v3 a) { /* Eigen normalized(): divide by sqrt(squaredNorm) when > 0 */
    float z = v3_sqnorm(a);
    if (z > 0.0f) {
        float n = sqrtf(z);
        v3 r = {a.x / n, a.y / n, a.z / n};
        return r;
    }
    return a;
}
static v3 vert(const ora_mesh* m, uint64_t i) {
    v3 r = {m->verts[3 * i], m->verts[3 * i + 1], m->verts[3 * i + 2]};
    return r;
}

/* ---- Mesh::CreateHalfEdges, Mesh.cpp:87-131 ------------------------------ */
typedef struct {
    uint64_t a, b, idx;
    int used;
} edge_slot;
static uint64_t edge_hash(uint64_t a, uint64_t b) {
    uint64_t h = a * 0x9E3779B97F4A7C15ull ^ (b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    return h ^ (h >> 29);
}
static int create_half_edges(ora_mesh* m) {
    const uint64_t ne = m->ntris * 3;
    uint64_t cap = 16;
    while (cap < ne * 2) cap *= 2;
    edge_slot* tab = (edge_slot*)calloc(cap, sizeof(edge_slot));
    for (uint64_t i = 0; i < ne; ++i) m->halfEdges[i] = (uint64_t)-1;
    for (uint64_t i = 0; i < ne; ++i) {
        uint64_t ea = m->tris[i], eb = (i % 3 == 2) ? m->tris[i - 2] : m->tris[i + 1];
        /* look for the reversed edge */
        uint64_t s = edge_hash(eb, ea) & (cap - 1);
        int found = 0;
        while (tab[s].used) {
            if (tab[s].a == eb && tab[s].b == ea) {
                m->halfEdges[tab[s].idx] = i;
                m->halfEdges[i] = tab[s].idx;
                found = 1;
                break;
            }
            s = (s + 1) & (cap - 1);
        }
        if (!found) { /* std::map::insert keeps the first entry for a key */
            s = edge_hash(ea, eb) & (cap - 1);
            int dup = 0;
            while (tab[s].used) {
                if (tab[s].a == ea && tab[s].b == eb) {
                    dup = 1;
                    break;
                }
                s = (s + 1) & (cap - 1);
            }
            if (!dup) {
                tab[s].used = 1;
                tab[s].a = ea;
                tab[s].b = eb;
                tab[s].idx = i;
            }
        }
    }
    free(tab);
    for (uint64_t i = 0; i < ne; ++i)
        if (m->halfEdges[i] == (uint64_t)-1) return 0;
    return 1;
}

ora_mesh* ora_mesh_create(const float* verts, uint64_t nverts, const uint64_t* tris, uint64_t ntris) {
    ora_mesh* m = (ora_mesh*)calloc(1, sizeof(ora_mesh));
    m->nverts = nverts;
    m->ntris = ntris;
    m->verts = (float*)malloc(sizeof(float) * 3 * nverts);
    m->tris = (uint64_t*)malloc(sizeof(uint64_t) * 3 * ntris);
    m->halfEdges = (uint64_t*)malloc(sizeof(uint64_t) * 3 * ntris);
    memcpy(m->verts, verts, sizeof(float) * 3 * nverts);
    memcpy(m->tris, tris, sizeof(uint64_t) * 3 * ntris);
    if (!create_half_edges(m)) { /* Mesh.cpp:121-128: open meshes are rejected */
        ora_mesh_free(m);
        return NULL;
    }
    return m;
}
void ora_mesh_free(ora_mesh* m) {
    if (!m) return;
    free(m->verts);
    free(m->tris);
    free(m->halfEdges);
    free(m);
}

/* ---- ClosestSimplexToPt, Source/Meshing/Utility.cpp:5-97 ------------------ */
enum { SIMPLEX_VERTEX = 0, SIMPLEX_EDGE = 1, SIMPLEX_FACE = 2 };
typedef struct {
    int simplex, simplexIdx;
    v3 closestPt;
} simplex_info;

static simplex_info closest_simplex(v3 pt, v3 a, v3 b, v3 c) {
    simplex_info s;
    const v3 ab = v3_sub(b, a), ac = v3_sub(c, a), bc = v3_sub(c, b);
    const float snom = v3_dot(v3_sub(pt, a), ab);
    const float sdenom = v3_dot(v3_sub(pt, b), v3_sub(a, b));
    const float tnom = v3_dot(v3_sub(pt, a), ac);
    const float tdenom = v3_dot(v3_sub(pt, c), v3_sub(a, c));
    if (snom < EPSILON_F32 && tnom < EPSILON_F32) {
        s.closestPt = a, s.simplex = SIMPLEX_VERTEX, s.simplexIdx = 0;
        return s;
    }
    const float unom = v3_dot(v3_sub(pt, b), bc);
    const float udenom = v3_dot(v3_sub(pt, c), v3_sub(b, c));
    if (sdenom < EPSILON_F32 && unom < EPSILON_F32) {
        s.closestPt = b, s.simplex = SIMPLEX_VERTEX, s.simplexIdx = 1;
        return s;
    }
    if (tdenom < EPSILON_F32 && udenom < EPSILON_F32) {
        s.closestPt = c, s.simplex = SIMPLEX_VERTEX, s.simplexIdx = 2;
        return s;
    }
    const v3 n = v3_cross(v3_sub(b, a), v3_sub(c, a));
    const float vc = v3_dot(n, v3_cross(v3_sub(a, pt), v3_sub(b, pt)));
    if (vc < EPSILON_F32 && snom > EPSILON_F32 && sdenom > EPSILON_F32) {
        s.closestPt = v3_add(a, v3_scale(snom / (snom + sdenom), ab));
        s.simplex = SIMPLEX_EDGE, s.simplexIdx = 0;
        return s;
    }
    const float va = v3_dot(n, v3_cross(v3_sub(b, pt), v3_sub(c, pt)));
    if (va < EPSILON_F32 && unom > EPSILON_F32 && udenom > EPSILON_F32) {
        s.closestPt = v3_add(b, v3_scale(unom / (unom + udenom), bc));
        s.simplex = SIMPLEX_EDGE, s.simplexIdx = 1;
        return s;
    }
    const float vb = v3_dot(n, v3_cross(v3_sub(c, pt), v3_sub(a, pt)));
    if (vb < EPSILON_F32 && tnom > EPSILON_F32 && tdenom > EPSILON_F32) {
        s.closestPt = v3_add(a, v3_scale(tnom / (tnom + tdenom), ac));
        s.simplex = SIMPLEX_EDGE, s.simplexIdx = 2;
        return s;
    }
    const float u = va / (va + vb + vc);
    const float v = vb / (va + vb + vc);
    const float w = 1.0f - u - v;
    s.closestPt = v3_add(v3_add(v3_scale(u, a), v3_scale(v, b)), v3_scale(w, c));
    s.simplex = SIMPLEX_FACE, s.simplexIdx = 0;
    return s;
}

/* ---- pseudo-normals, Mesh.cpp:162-242 ------------------------------------- */
static v3 pn_face(const ora_mesh* m, uint64_t t) { /* :190-198 */
    v3 a = vert(m, m->tris[3 * t]), b = vert(m, m->tris[3 * t + 1]), c = vert(m, m->tris[3 * t + 2]);
    return v3_normalized(v3_cross(v3_sub(b, a), v3_sub(c, a)));
}
static v3 pn_edge(const ora_mesh* m, uint64_t t, int sidx) { /* :201-215 */
    const uint64_t adjEdge = m->halfEdges[3 * t + (uint64_t)sidx];
    const uint64_t adjTri = (adjEdge - (adjEdge % 3)) / 3;
    const double PI = 3.14159265359;
    v3 nA = pn_face(m, t), nB = pn_face(m, adjTri);
    /* nA * PI: Vector3f times f64 scalar -> the scalar is cast to f32 */
    const float pif = (float)PI;
    return v3_normalized(v3_add(v3_scale(pif, nA), v3_scale(pif, nB)));
}
static v3 pn_vertex(const ora_mesh* m, uint64_t t, int sidx) { /* :218-242 */
    v3 n = {0, 0, 0};
    uint64_t he = 3 * t + (uint64_t)sidx, cur = t;
    uint64_t guard = 0;
    do {
        v3 tri[3] = {vert(m, m->tris[3 * cur]), vert(m, m->tris[3 * cur + 1]), vert(m, m->tris[3 * cur + 2])};
        v3 ab = v3_sub(tri[(he + 1) % 3], tri[he % 3]);
        v3 ac = v3_sub(tri[(he + 2) % 3], tri[he % 3]);
        float ang = acosf(v3_dot(v3_normalized(ab), v3_normalized(ac)));
        n = v3_add(n, v3_scale(ang, pn_face(m, cur)));
        he = m->halfEdges[he];
        he = ((he % 3) == 2) ? (he - 2) : (he + 1);
        cur = (he - (he % 3)) / 3;
    } while (cur != t && ++guard < 100000);
    return v3_normalized(n);
}

/* Mesh::SignedDistanceAtPt(pt) (naive), Mesh.cpp:42-51 with
 * Mesh::ClosestTriangleToPt, Mesh.cpp:134-159 */
float ora_mesh_signed_distance(const ora_mesh* m, const float p[3], uint64_t* tri_out, int* simplex_out) {
    const v3 pt = {p[0], p[1], p[2]};
    simplex_info best;
    memset(&best, 0, sizeof best);
    uint64_t bestTri = (uint64_t)-1;
    float bestDist = FLT_MAX;
    for (uint64_t i = 0; i < m->ntris; ++i) {
        simplex_info c = closest_simplex(pt, vert(m, m->tris[3 * i]), vert(m, m->tris[3 * i + 1]), vert(m, m->tris[3 * i + 2]));
        float d = v3_sqnorm(v3_sub(pt, c.closestPt));
        if (d < bestDist) {
            best = c;
            bestTri = i;
            bestDist = d;
        }
    }
    v3 pn;
    switch (best.simplex) {
        case SIMPLEX_VERTEX: pn = pn_vertex(m, bestTri, best.simplexIdx); break;
        case SIMPLEX_EDGE: pn = pn_edge(m, bestTri, best.simplexIdx); break;
        default: pn = pn_face(m, bestTri); break;
    }
    const v3 d = v3_sub(pt, best.closestPt);
    const float sign = v3_dot(pn, d) > 0.0f ? 1.0f : -1.0f;
    if (tri_out) *tri_out = bestTri;
    if (simplex_out) *simplex_out = best.simplex * 4 + best.simplexIdx;
    return sign * sqrtf(v3_sqnorm(d));
}

void ora_acosf_batch(uint32_t first, uint32_t stride, size_t n, float* out) {
    for (size_t i = 0; i < n; ++i) {
        const uint32_t b = first + (uint32_t)i * stride;
        float x;
        memcpy(&x, &b, 4);
        out[i] = acosf(x);
    }
}

void ora_scalars(unsigned long long out[8]) {
    const float eps = EPSILON_F32;
    uint32_t bits;
    memcpy(&bits, &eps, 4);
    out[0] = ORA_BASIS_MAX_DEGREE, out[1] = ORA_TREE_MAX_DEPTH;
    out[2] = 4, out[3] = 8, out[4] = 8, out[5] = 8; /* int, long, unsigned long, size_t: "u32" is 8 bytes (a quirk the block layout keeps) */
    out[6] = 16;                                  /* { usize size; void* ptr; } */
    out[7] = bits;
}
