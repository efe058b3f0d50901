// Scalar-sized Query / QueryWithGradient calls, answered where the caller is.
//
// The reference's Octree::Query(pt) (Source/HP/Octree.cpp:662-702) costs ~75 ns and its own tests and benchmarks call it in loops
// of 1 M - 8 M points (Source/Tests/HPUnitTests.cpp:64-75, Source/Benchmarks/HPBenchmarks.cpp:105-109).  Through a kernel launch
// a one-point call is ~15 us whatever the kernel does, so calls of up to kHostQueryPoints points are evaluated here, on the
// calling thread, from the copy of the block's node array and coefficients the tree handle keeps (hpsdf_tree_upload) -- the
// same statements in the same order as queryPoint / queryPointWithGradient of kernels.hip (and this file is compiled with
// -ffp-contract=off like everything else), so the values are the kernels' bit for bit (tests/test_gpu_parity.py compares them
// on the edge-point set).  It is not a CPU build of the library: a tree handle only exists on a device context, Create, the
// fields and every batched call are GPU code, and there is no entry point that works without a GPU.
#include <cfloat>
#include <cmath>
#include <cstdint>

#include "runtime.hpp"
#include "tables.hpp"

namespace hpsdf {

namespace {

struct Leaf {
    const double* co;
    int degree, depth;
    double u[3];  // (pt - centre) * (2 << depth), Octree.cpp:862
};

// Octree.cpp:665-701: root remap, f32 containment (both ends inclusive, NaN fails), mid-plane descent (>= takes the upper child)
inline bool descend(const hpsdf_tree& t, const double* xyz, Leaf& L) {
    const double p[3] = {(xyz[0] - t.dev.rootCentre[0]) * t.dev.rootInvSizes[0], (xyz[1] - t.dev.rootCentre[1]) * t.dev.rootInvSizes[1],
                         (xyz[2] - t.dev.rootCentre[2]) * t.dev.rootInvSizes[2]};
    const float fx = (float)p[0], fy = (float)p[1], fz = (float)p[2];
    if (!(fx >= -0.5f && fx <= 0.5f && fy >= -0.5f && fy <= 0.5f && fz >= -0.5f && fz <= 0.5f)) return false;
    double c[3] = {0.0, 0.0, 0.0}, q = 0.25;  // cell centres are exact dyadics: the mid-planes of the f32 boxes
    int depth = 0;
    const NodeRec* nodes = t.hRecs.data();  // (interior: a = first child; leaf: a = offset in the line-aligned coefficient mirror, b = degree)
    uint64_t idx = 0;
    while (nodes[idx].b == kInteriorTag) {
        uint64_t next = nodes[idx].a;
        for (int a = 0; a < 3; ++a) {
            const bool up = p[a] >= c[a];
            next += up ? (1ull << a) : 0ull;
            c[a] = up ? c[a] + q : c[a] - q;
        }
        q = q * 0.5;
        ++depth;
        idx = next;
    }
    const double s = (double)(2 << depth);
    L.co = t.hPadded.data() + nodes[idx].a;
    L.degree = (int)nodes[idx].b;
    L.depth = depth;
    for (int a = 0; a < 3; ++a) L.u[a] = (p[a] - c[a]) * s;
    return true;
}

}  // namespace

// Octree::Query + FApprox (Octree.cpp:662-702, 859-901)
double hostQueryPoint(const hpsdf_tree& t, const double* xyz) {
    Leaf L;
    if (!descend(t, xyz, L)) return DBL_MAX;
    const Tables& T = tables();
    double tab[3][13];
    for (int a = 0; a < 3; ++a) {
        tab[a][0] = T.normalisedLengths[0][L.depth];
        double m2 = 0.0, m1 = 1.0;
        for (int j = 1; j <= L.degree; ++j) {
            const double l = T.recurrence[j][0] * L.u[a] * m1 - T.recurrence[j][1] * m2;
            m2 = m1, m1 = l;
            tab[a][j] = l * T.normalisedLengths[j][L.depth];
        }
    }
    double f = 0.0;
    const int n = (int)T.coeffCount[L.degree];
    for (int i = 0; i < n; ++i) {
        double lp = tab[0][T.basisIndex[i][0]];
        lp = lp * tab[1][T.basisIndex[i][1]];
        lp = lp * tab[2][T.basisIndex[i][2]];
        f = f + L.co[i] * lp;
    }
    return f;
}

// Octree::QueryWithGradient + FApproxWithGradient (Octree.cpp:749-789, 904-985).  Outside the root: *out = DBL_MAX, grad untouched.
void hostQueryPointWithGradient(const hpsdf_tree& t, const double* xyz, double* out, double* grad, int leftAssoc) {
    Leaf L;
    if (!descend(t, xyz, L)) {
        *out = DBL_MAX;
        return;
    }
    const Tables& T = tables();
    const double eps = 0.0001;
    double Lg[13][3][3];
    for (int a = 0; a < 3; ++a) {
        const double u = L.u[a];  // :907
        Lg[0][a][0] = Lg[0][a][1] = Lg[0][a][2] = T.normalisedLengths[0][L.depth];
        double a2 = 0.0, a1 = 1.0, b2 = 0.0, b1 = 1.0, c2 = 0.0, c1 = 1.0;
        for (int j = 1; j <= L.degree; ++j) {
            const double r0 = T.recurrence[j][0], r1 = T.recurrence[j][1], nl = T.normalisedLengths[j][L.depth];
            const double a0 = r0 * u * a1 - r1 * a2;          // :937
            const double b0 = r0 * (u + eps) * b1 - r1 * b2;  // :941
            const double c0 = r0 * (u - eps) * c1 - r1 * c2;  // :945
            a2 = a1, a1 = a0, b2 = b1, b1 = b0, c2 = c1, c1 = c0;
            Lg[j][a][0] = a0 * nl, Lg[j][a][1] = b0 * nl, Lg[j][a][2] = c0 * nl;
        }
    }
    const int nc = (int)T.coeffCount[L.degree];
    double g[3];
    for (int k = 0; k < 3; ++k) {  // :956-968
        double p1 = 0.0, m1 = 0.0;
        for (int r = 0; r < nc; ++r) {
            p1 = p1 + L.co[r] * Lg[T.basisIndex[r][k]][k][1];
            m1 = m1 + L.co[r] * Lg[T.basisIndex[r][k]][k][2];
        }
        g[k] = (p1 - m1) / (2.0 * eps);
    }
    const double g2[3] = {g[0] * g[0], g[1] * g[1], g[2] * g[2]};
    const double z = leftAssoc ? (g2[0] + g2[1]) + g2[2] : g2[0] + (g2[1] + g2[2]);  // Eigen normalize()
    if (z > 0.0) {
        const double nrm = std::sqrt(z);
        g[0] = g[0] / nrm, g[1] = g[1] / nrm, g[2] = g[2] / nrm;
    }
    double f = 0.0;  // :972-984
    for (int r = 0; r < nc; ++r) {
        double lp = Lg[T.basisIndex[r][0]][0][0];
        lp = lp * Lg[T.basisIndex[r][1]][1][0];
        lp = lp * Lg[T.basisIndex[r][2]][2][0];
        f = f + L.co[r] * lp;
    }
    *out = f;
    grad[0] = g[0], grad[1] = g[1], grad[2] = g[2];
}

// Octree::QueryRay (Octree.cpp:705-746; Ray::IntersectAABB, Source/Utility/Ray.cpp:18-68): the statements of query_ray_kernel
// (kernels.hip) -- the origin moved to the unit cube, the direction left as it is, the first intersection with [-0.5, 0.5]^3 unless the
// origin lies inside, then at most 200 steps of Query at the stepped point (Query maps to the unit cube AGAIN, as the reference's
// call does).  *tOut is written on a hit only.
bool hostQueryRay(const hpsdf_tree& t, const double* origin, const double* dir, double tMax, double* tOut) {
    double o[3], d[3], inv[3];
    int sgn[3];
    for (int a = 0; a < 3; ++a) {
        o[a] = (origin[a] - t.dev.rootCentre[a]) * t.dev.rootInvSizes[a];  // :711
        d[a] = dir[a];
        inv[a] = 1.0 / d[a];  // Ray.cpp:10 cwiseInverse
        sgn[a] = inv[a] < 0.0 ? 1 : 0;
    }
    double im[3] = {o[0], o[1], o[2]};  // intMin
    const float fx = (float)o[0], fy = (float)o[1], fz = (float)o[2];
    const bool inside = fx >= -0.5f && fx <= 0.5f && fy >= -0.5f && fy <= 0.5f && fz >= -0.5f && fz <= 0.5f;
    if (!inside) {
        double a0 = ((sgn[0] ? 0.5 : -0.5) - o[0]) * inv[0], b0 = ((sgn[0] ? -0.5 : 0.5) - o[0]) * inv[0];
        const double a1 = ((sgn[1] ? 0.5 : -0.5) - o[1]) * inv[1], b1 = ((sgn[1] ? -0.5 : 0.5) - o[1]) * inv[1];
        if ((a0 > b1) || (a1 > b0)) return false;
        if (a1 > a0) a0 = a1;
        if (b1 < b0) b0 = b1;
        const double a2 = ((sgn[2] ? 0.5 : -0.5) - o[2]) * inv[2], b2 = ((sgn[2] ? -0.5 : 0.5) - o[2]) * inv[2];
        if ((a0 > b2) || (a2 > b0)) return false;
        if (a2 > a0) a0 = a2;
        im[0] = a0, im[1] = a1, im[2] = a2;
    }
    const double eps = 0.0001, minStep = 0.0001;
    double dist = 0.0;
    for (int s = 0; s < 200; ++s) {
        const double p[3] = {im[0] + dist * d[0], im[1] + dist * d[1], im[2] + dist * d[2]};
        const double v = hostQueryPoint(t, p);
        if (v < eps) {
            *tOut = v;  // :730
            return true;
        }
        dist = dist + (v * 0.95 + minStep);  // :736
        if (dist > tMax) break;
    }
    return false;
}

}  // namespace hpsdf
