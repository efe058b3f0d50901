/*
 * hp_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 * See hp_oracle.h for scope and pinning status.  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off: the reference is built for baseline x86-64,
 * which has no FMA, so no multiply-add is ever contracted).
 */
#include "hp_oracle.h"

#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>

/* ======================================================================== */
/* Tables                                                                   */
/* ======================================================================== */

static int g_tables_ready = 0;
static double g_roots[2080], g_weights[2080];
static double g_nl[13][11];
static double g_rec[13][2];
static uint64_t g_count[13];
static uint64_t g_bidx[455][3];
static uint64_t g_sumton[50];

/* --- double-double helpers, used only to produce correctly rounded
 *     Gauss-Legendre nodes/weights.  Include/HP/Legendre.h:7-2089,2091-4173
 *     holds ~290-digit decimal literals, i.e. the correctly rounded doubles. */
typedef struct {
    double hi, lo;
} dd;
static dd dd_qts(double a, double b) {
    double s = a + b;
    dd r = {s, b - (s - a)};
    return r;
}
static dd dd_ts(double a, double b) {
    double s = a + b, bb = s - a;
    dd r = {s, (a - (s - bb)) + (b - bb)};
    return r;
}
static dd dd_tp(double a, double b) {
    double p = a * b;
    dd r = {p, fma(a, b, -p)};
    return r;
}
static dd dd_add(dd a, dd b) {
    dd s = dd_ts(a.hi, b.hi), t = dd_ts(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_qts(s.hi, s.lo);
    s.lo += t.lo;
    return dd_qts(s.hi, s.lo);
}
static dd dd_from(double a) {
    dd r = {a, 0.0};
    return r;
}
static dd dd_sub(dd a, dd b) {
    dd nb = {-b.hi, -b.lo};
    return dd_add(a, nb);
}
static dd dd_mul(dd a, dd b) {
    dd p = dd_tp(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_qts(p.hi, p.lo);
}
static dd dd_div(dd a, dd b) {
    double q1 = a.hi / b.hi;
    dd r = dd_sub(a, dd_mul(b, dd_from(q1)));
    double q2 = r.hi / b.hi;
    r = dd_sub(r, dd_mul(b, dd_from(q2)));
    double q3 = r.hi / b.hi;
    return dd_add(dd_qts(q1, q2), dd_from(q3));
}
/* P_n(x) and P_n'(x) by the Bonnet recurrence */
static void dd_legendre(int n, dd x, dd* pn, dd* dpn) {
    dd p0 = dd_from(1.0), p1 = x;
    for (int k = 2; k <= n; ++k) {
        dd a = dd_mul(dd_mul(dd_from(2.0 * k - 1.0), x), p1);
        dd b = dd_mul(dd_from(k - 1.0), p0);
        dd pk = dd_div(dd_sub(a, b), dd_from((double)k));
        p0 = p1;
        p1 = pk;
    }
    *pn = p1;
    *dpn = dd_div(dd_mul(dd_from((double)n), dd_sub(dd_mul(x, p1), p0)), dd_sub(dd_mul(x, x), dd_from(1.0)));
}

/* One n-point rule in the order Legendre.h stores it: 0 first for odd n, then
 * (-x,+x) pairs by ascending |x| -- except n = 6 and n = 9, whose pairs are
 * stored in the order (2nd,1st,3rd) and (3rd,4th,1st,2nd) smallest.  n = 9 is
 * the rule of every degree-2 fit (Octree.cpp:1016-1017), so the quirk changes
 * the summation order of the coarse pass. */
static void gl_rule(int n, double* x, double* w) {
    int m = n / 2, o = n & 1;
    double ax[32], aw[32];
    if (o) {
        dd pn, dp;
        dd_legendre(n, dd_from(0.0), &pn, &dp);
        x[0] = 0.0;
        w[0] = dd_div(dd_from(2.0), dd_mul(dp, dp)).hi;
    }
    for (int k = 0; k < m; ++k) {
        dd xx = dd_from(cos(M_PI * ((m - k) - 0.25) / (n + 0.5)));
        dd pn, dp;
        for (int it = 0; it < 8; ++it) {
            dd_legendre(n, xx, &pn, &dp);
            xx = dd_sub(xx, dd_div(pn, dp));
        }
        dd_legendre(n, xx, &pn, &dp);
        dd one_m = dd_sub(dd_from(1.0), dd_mul(xx, xx));
        ax[k] = xx.hi;
        aw[k] = dd_div(dd_from(2.0), dd_mul(one_m, dd_mul(dp, dp))).hi;
    }
    static const int perm6[3] = {1, 0, 2};
    static const int perm9[4] = {2, 3, 0, 1};
    for (int k = 0; k < m; ++k) {
        int src = (n == 6) ? perm6[k] : (n == 9) ? perm9[k] : k;
        x[o + 2 * k] = -ax[src];
        x[o + 2 * k + 1] = ax[src];
        w[o + 2 * k] = aw[src];
        w[o + 2 * k + 1] = aw[src];
    }
}

/* Include/HP/Utility.h:25-35 SqrtConst: 100 Newton steps from guess = x */
static double sqrt_const(double x) {
    double guess = x;
    for (int i = 0; i < 100; ++i) guess = 0.5 * (guess + x / guess);
    return guess;
}
/* Include/HP/Utility.h:14-24 PowConst */
static double pow_const(double base, unsigned n) { return n == 0 ? 1.0 : base * pow_const(base, n - 1); }

void ora_tables_init(void) {
    if (g_tables_ready) return;
    /* Legendre.h: rule n occupies [n(n-1)/2, n(n+1)/2) */
    for (int n = 1; n <= 64; ++n) gl_rule(n, g_roots + n * (n - 1) / 2, g_weights + n * (n - 1) / 2);
    /* Utility.h:40-57 SumToN */
    for (uint64_t i = 0; i <= 49; ++i) {
        uint64_t s = 0;
        for (uint64_t j = 0; j <= i; ++j) s += j;
        g_sumton[i] = s;
    }
    /* Utility.h:63-78 NormalisedLengths[i][j] = SqrtConst((2i+1) * 2^j) */
    for (unsigned i = 0; i <= 12; ++i)
        for (unsigned j = 0; j <= 10; ++j) g_nl[i][j] = sqrt_const((2.0 * i + 1.0) * pow_const(2.0, j));
    /* Utility.h:87-106: (u32)(1/6.0 * (i+1) * (i+2) * (i+3)) in f64 => count[6] = 83 */
    {
        const double f = 1.0 / 6.0;
        for (uint64_t i = 0; i <= 12; ++i) g_count[i] = (uint64_t)(f * (i + 1) * (i + 2) * (i + 3));
    }
    /* Utility.h:112-127 recurrence constants */
    g_rec[0][0] = 0.0;
    g_rec[0][1] = 0.0;
    for (unsigned i = 1; i <= 12; ++i) {
        g_rec[i][0] = (2.0 * i - 1.0) / i;
        g_rec[i][1] = (i - 1.0) / i;
    }
    /* Utility.h:133-160 BasisIndexValues: by total degree p, then i, then j ascending */
    {
        unsigned v = 0;
        for (unsigned p = 0; p <= 12; ++p)
            for (unsigned i = 0; i <= p; ++i)
                for (unsigned j = 0; j <= p - i; ++j)
                    for (unsigned k = 0; k <= p - i - j; ++k)
                        if (i + j + k == p) {
                            g_bidx[v][0] = i;
                            g_bidx[v][1] = j;
                            g_bidx[v][2] = k;
                            ++v;
                        }
    }
    g_tables_ready = 1;
}
const double* ora_gl_roots(void) { ora_tables_init(); return g_roots; }
const double* ora_gl_weights(void) { ora_tables_init(); return g_weights; }
const double* ora_normalised_lengths(void) { ora_tables_init(); return &g_nl[0][0]; }
const double* ora_recurrence(void) { ora_tables_init(); return &g_rec[0][0]; }
const uint64_t* ora_coeff_count(void) { ora_tables_init(); return g_count; }
const uint64_t* ora_basis_index(void) { ora_tables_init(); return &g_bidx[0][0]; }
const uint64_t* ora_sum_to_n(void) { ora_tables_init(); return g_sumton; }

/* ======================================================================== */
/* Config                                                                   */
/* ======================================================================== */

/* Source/HP/Config.cpp:5-14.  threadCount is left at 1 (the oracle is serial);
 * padding is zeroed so that serialised bytes are reproducible. */
void ora_config_default(ora_config* c) {
    memset(c, 0, sizeof(*c));
    c->target_error_threshold = pow(10, -10);
    c->weighting_type = 0;
    c->continuity_enforce = 1;
    c->continuity_strength = 8.0;
    c->thread_count = 1;
    for (int a = 0; a < 3; ++a) {
        c->root_min[a] = -0.5f;
        c->root_max[a] = 0.5f;
    }
    c->enable_logging = 0;
}

/* ======================================================================== */
/* Fields                                                                   */
/* ======================================================================== */

/* Eigen's Vector3d::norm() reduces as x^2 + (y^2 + z^2); the reference test
 * fields are written with it (Source/Tests/HPUnitTests.cpp:48-51). */
/* Which way a 3-vector reduction (prod / squaredNorm / dot) associates is Eigen's choice, not the reference's: a + (b + c)
 * (the default here and in the product) or (a + b) + c (ora_set_reduction_order(1) / hpsdf_set_reduction_order(1)).
 * What Eigen does, from memory of Core/Redux.h (Eigen is absent from this image, so this is a reading, not a test): Eigen >= 3.3 --
 * the reference clones Eigen's HEAD (Build.sh:3-11) and sets no vectorisation macro (CMakeLists.txt) -- reduces a fixed-size vector by
 * redux_impl<LinearVectorizedTraversal, CompleteUnrolling>: the first (Size / PacketSize) * PacketSize elements in packets (predux),
 * then func(res, <the rest>).  find_best_packet<double, 3> is Packet2d on every vectorised target (SSE2 is the x86-64 baseline; with
 * AVX Packet4d is halved because 3 % 4 != 0; aarch64 NEON has a Packet2d), so a Vector3d would reduce as (a . b) . c; a Vector3f gets
 * Packet4f, of which no whole packet fits three floats, so redux_novec_unroller's halving remains: a . (b . c), what the f32 mesh path
 * (hp_oracle_mesh.c) restates.  An unvectorised build (EIGEN_DONT_VECTORIZE, Eigen 3.2) reduces doubles as a . (b . c) too.
 * The DEFAULT stays a . (b . c) because that is the arithmetic every reference-derived number this repository holds was produced with:
 * the survey's known answers (tests/golden/kats.json, 13 digits) are reproduced under it and missed under the other order
 * (tests/test_oracle_fit.py::test_kats_tell_the_reduction_orders_apart) -- the surveyor's Eigen stand-in reduced that way.  What the
 * choice moves is a measured number (tools/assoc_sensitivity.py), and both settings are tested against the
 * product bit for bit (tests/test_gpu_parity.py::test_reduction_order_switch_matches_the_oracle).  Sites: Vector3d::norm() in the test
 * fields, aabbScale.prod() (:1022), unitWeights.prod() (:1040), grad.normalize() (:970). */
static int g_leftAssoc = 0;
void ora_set_reduction_order(int left_assoc) { g_leftAssoc = left_assoc != 0; }
int ora_get_reduction_order(void) { return g_leftAssoc; }
static double sum3(double a, double b, double c) { return g_leftAssoc ? (a + b) + c : a + (b + c); }
static double prod3(double a, double b, double c) { return g_leftAssoc ? (a * b) * c : a * (b * c); }
static double norm3(double x, double y, double z) { return sqrt(sum3(x * x, y * y, z * z)); }

static double prim_eval(const ora_prim* pr, const double pt[3]) {
    const double* p = pr->p;
    switch (pr->kind) {
        case ORA_PRIM_SPHERE: /* centre p[0..2], radius p[3] */
            return norm3(pt[0] - p[0], pt[1] - p[1], pt[2] - p[2]) - p[3];
        case ORA_PRIM_BOX: { /* centre p[0..2], half extents p[3..5]; exact box SDF */
            double qx = fabs(pt[0] - p[0]) - p[3];
            double qy = fabs(pt[1] - p[1]) - p[4];
            double qz = fabs(pt[2] - p[2]) - p[5];
            double mx = fmax(qx, 0.0), my = fmax(qy, 0.0), mz = fmax(qz, 0.0);
            double outside = norm3(mx, my, mz);
            double inside = fmin(fmax(qx, fmax(qy, qz)), 0.0);
            return outside + inside;
        }
        case ORA_PRIM_TORUS_Y: { /* centre p[0..2], major R p[3], minor r p[4], axis y */
            double dx = pt[0] - p[0], dy = pt[1] - p[1], dz = pt[2] - p[2];
            double l = sqrt(dx * dx + dz * dz) - p[3];
            return sqrt(l * l + dy * dy) - p[4];
        }
        case ORA_PRIM_PLANE: /* normal p[0..2], offset p[3] */
            return (p[0] * pt[0] + (p[1] * pt[1] + p[2] * pt[2])) + p[3];
        default:
            return 0.0;
    }
}

static double combine(int op, double acc, double d) {
    switch (op) {
        case ORA_OP_UNION: return fmin(acc, d);
        case ORA_OP_INTERSECT: return fmax(acc, d);
        case ORA_OP_SUBTRACT: return fmax(acc, -d);
        default: return acc;
    }
}

double ora_field_eval(const ora_field* f, const double pt[3]) {
    switch (f->kind) {
        case ORA_FIELD_ANALYTIC: {
            double acc = prim_eval(&f->prims[0], pt);
            for (int i = 1; i < f->nprims; ++i) acc = combine(f->prims[i].op, acc, prim_eval(&f->prims[i], pt));
            return acc;
        }
        case ORA_FIELD_CALLBACK:
            return f->cb(pt, 0, f->user);
        case ORA_FIELD_MESH: {
            /* user glue of SURVEY 3.4: (f64) mesh.SignedDistanceAtPt(p.cast<f32>()) */
            float p32[3] = {(float)pt[0], (float)pt[1], (float)pt[2]};
            return (double)ora_mesh_signed_distance(f->mesh, p32, NULL, NULL);
        }
        case ORA_FIELD_TREE_CSG: {
            /* Octree.cpp:355-400: Union min(old,F), Subtract max(-old,F), Intersect max(old,F) */
            double o = ora_query(f->tree, pt);
            double n = ora_field_eval(f->inner, pt);
            switch (f->csg_op) {
                case ORA_OP_UNION: return o < n ? o : n;          /* std::min(old, F) */
                case ORA_OP_SUBTRACT: return (o * -1.0) < n ? n : (o * -1.0); /* std::max(-old, F) */
                default: return o < n ? n : o;                     /* std::max(old, F) */
            }
        }
        default:
            return 0.0;
    }
}

/* ======================================================================== */
/* Per-node numerics                                                        */
/* ======================================================================== */

/* Octree::LpX, Octree.cpp:988-1004 */
double ora_lpx(uint64_t p, double x) {
    ora_tables_init();
    double LiMinus2 = 0.0, LiMinus1 = 1.0, Li = 1.0;
    for (uint64_t i = 1; i <= p; ++i) {
        Li = g_rec[i][0] * x * LiMinus1 - g_rec[i][1] * LiMinus2;
        LiMinus2 = LiMinus1;
        LiMinus1 = Li;
    }
    return Li;
}

/* Octree::CornerAABB, Octree.cpp:1096-1112 (f32 midpoint) */
void ora_corner_aabb(const float bmin[3], const float bmax[3], unsigned i, float omin[3], float omax[3]) {
    for (int d = 0; d < 3; ++d) {
        omin[d] = bmin[d];
        omax[d] = bmax[d];
        float mid = (bmax[d] + bmin[d]) * 0.5f;
        if (i & (1u << d))
            omin[d] = mid;
        else
            omax[d] = mid;
    }
}

/* ---- nearness weighting, Octree.cpp:1209-1247 ---------------------------------
 * The reference averages FApprox over 100 points drawn with aabb.sample(), i.e. std::rand(): run-to-run
 * different and unpinnable.  The restatement keeps the arithmetic and replaces the generator by a pure
 * function of (cell, sample, axis): SplitMix64 -> 24 bits -> f32 in [0,1).  PARITY UNPINNED against the
 * reference beyond the statistics of the mean; product and oracle agree bit-for-bit with each other. */
static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint32_t f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
uint64_t ora_weight_key(const float bmin[3], int depth, int degree) {
    const uint64_t a = (uint64_t)f32_bits(bmin[0]) | ((uint64_t)f32_bits(bmin[1]) << 32);
    const uint64_t b = (uint64_t)f32_bits(bmin[2]) | ((uint64_t)(unsigned)depth << 32) | ((uint64_t)(unsigned)degree << 40);
    return splitmix64(a) ^ splitmix64(b ^ 0xD1B54A32D192ED03ull);
}
/* |mean of FApprox over the 100 sample points| (:1215-1222 / :1236-1243) */
double ora_weight_mean(const double* coeffs, int degree, const float bmin[3], const float bmax[3], int depth) {
    const uint64_t key = ora_weight_key(bmin, depth, degree);
    const uint64_t nSamples = 100;
    double fIntegral = 0.0;
    for (uint64_t i = 0; i < nSamples; ++i) {
        double pt[3];
        for (int a = 0; a < 3; ++a) {
            /* AlignedBox3f::sample(): min + (max - min) * r in f32, then cast<f64>() */
            const float r = (float)(splitmix64(key + (i * 3 + (uint64_t)a) * 0x9E3779B97F4A7C15ull) >> 40) * (1.0f / 16777216.0f);
            pt[a] = (double)(bmin[a] + (bmax[a] - bmin[a]) * r);
        }
        fIntegral += ora_fapprox(coeffs, degree, bmin, bmax, pt, depth);
    }
    fIntegral /= (double)nSamples;
    return fabs(fIntegral);
}
double ora_weight_from_mean(int type, double strength, double mean) {
    const double d = sqrt(3.0);
    if (type == 1) { /* CalculatePolyWeighting :1224-1226 */
        const double k = pow(1.0 - mean / d, strength);
        const double m = (k < 0.0) ? 0.0 : k;   /* std::max<f64>(k, 0.0) */
        return (m < 1.0) ? m : 1.0;             /* std::min<f64>(1.0, .) */
    }
    if (type == 2) return exp(-1.0 * strength * mean / d); /* CalculateExpWeighting :1246 */
    return 1.0;
}
double ora_poly_weighting(const double* coeffs, int degree, const float bmin[3], const float bmax[3], int depth, double strength) {
    return ora_weight_from_mean(1, strength, ora_weight_mean(coeffs, degree, bmin, bmax, depth));
}
double ora_exp_weighting(const double* coeffs, int degree, const float bmin[3], const float bmax[3], int depth, double strength) {
    return ora_weight_from_mean(2, strength, ora_weight_mean(coeffs, degree, bmin, bmax, depth));
}

/* Octree::FitPolynomial, Octree.cpp:1007-1093 with the callback wrapper of
 * Octree.cpp:322-328 folded in (F = F_(pt * rootBounds + centre)). */
double ora_fit_polynomial(const ora_field* f, const ora_config* cfg, double* coeffs, int basis_degree,
                          const float bmin[3], const float bmax[3], int degree, int depth, int literal) {
    ora_tables_init();
    /* :1012-1013 */
    const uint64_t startingIdx = basis_degree > 0 ? g_count[basis_degree] : 0;
    const uint64_t endingIdx = g_count[degree];
    /* :1016-1017 */
    const uint64_t GQStart = g_sumton[4 * degree];
    const uint64_t GQEnd = g_sumton[4 * degree + 1];
    const uint64_t nq = GQEnd - GQStart;
    /* :1020-1022  sizes()/center() are f32 ops (Eigen AlignedBox3f), then cast */
    double scale[3], centre[3];
    for (int a = 0; a < 3; ++a) {
        scale[a] = (double)(bmax[a] - bmin[a]) * 0.5;
        centre[a] = (double)((bmin[a] + bmax[a]) / 2.0f);
    }
    const double scalesMult = prod3(scale[0], scale[1], scale[2]); /* Eigen prod(): a*(b*c) unless ora_set_reduction_order(1) */
    /* Octree.cpp:322,324 */
    double rootBounds[3], rootCentre[3];
    for (int a = 0; a < 3; ++a) {
        rootBounds[a] = (double)(cfg->root_max[a] - cfg->root_min[a]);
        rootCentre[a] = (double)((cfg->root_min[a] + cfg->root_max[a]) / 2.0f);
    }
    /* :1025 */
    memset(coeffs + startingIdx, 0, (endingIdx - startingIdx) * sizeof(double));

    /* per-axis cache of LpX(idx, root) -- same values the literal path recomputes */
    double T[13][49];
    if (!literal)
        for (int p = 0; p <= degree; ++p)
            for (uint64_t q = 0; q < nq; ++q) T[p][q] = ora_lpx((uint64_t)p, g_roots[GQStart + q]);

    /* :1028-1056 */
    for (uint64_t i = GQStart; i < GQEnd; ++i)
        for (uint64_t j = GQStart; j < GQEnd; ++j)
            for (uint64_t k = GQStart; k < GQEnd; ++k) {
                const double us[3] = {g_roots[i], g_roots[j], g_roots[k]};
                const double wprod = prod3(g_weights[i], g_weights[j], g_weights[k]);
                double world[3];
                for (int a = 0; a < 3; ++a) {
                    double s = us[a] * scale[a] + centre[a]; /* :1039 */
                    world[a] = s * rootBounds[a] + rootCentre[a]; /* :327 */
                }
                const double FaabSample = scalesMult * wprod * ora_field_eval(f, world); /* :1040 */
                const uint64_t qi[3] = {i - GQStart, j - GQStart, k - GQStart};
                for (uint64_t c = startingIdx; c < endingIdx; ++c) {
                    double Lp = 1.0;
                    for (int p = 0; p < 3; ++p) {
                        const uint64_t bi = g_bidx[c][p];
                        Lp *= literal ? ora_lpx(bi, us[p]) : T[bi][qi[p]];
                        Lp *= g_nl[bi][depth];
                    }
                    coeffs[c] += Lp * FaabSample;
                }
            }
    /* :1062-1069 */
    double newError = 0.0;
    for (uint64_t i = 0; i < endingIdx; ++i)
        if ((g_bidx[i][0] + g_bidx[i][1] + g_bidx[i][2]) == (uint64_t)degree) newError += coeffs[i] * coeffs[i];
    /* :1071-1092 */
    switch (cfg->weighting_type) {
        case 1: return newError * ora_poly_weighting(coeffs, degree, bmin, bmax, depth, cfg->weighting_strength);
        case 2: return newError * ora_exp_weighting(coeffs, degree, bmin, bmax, depth, cfg->weighting_strength);
        default: return newError;
    }
}

/* EstimateHImprovement (Octree.cpp:804-826), EstimatePImprovement (:829-856),
 * decision (:594-601). */
void ora_job(const ora_field* f, const ora_config* cfg, const float bmin[3], const float bmax[3], int depth,
             int degree, double err, const double* coeffs, double* p_coeffs, double* h_coeffs,
             ora_job_result* out, int literal) {
    ora_tables_init();
    memset(out, 0, sizeof(*out));
    const int coarse = fabs(err - ORA_INITIAL_NODE_ERR) < DBL_EPSILON; /* :806,:831 */
    out->coarse = coarse;
    /* H */
    if (coarse || depth >= ORA_TREE_MAX_DEPTH) {
        /* :806-810 coarse -> 0.  depth == 10: the reference fits children at
         * depth 11 (out-of-range NormalisedLengths read) and never uses the
         * result (:601); skipped here. */
        out->h_imp = 0.0;
    } else {
        double maxNewErr = 0.0;
        for (unsigned i = 0; i < 8; ++i) {
            float cmin[3], cmax[3];
            ora_corner_aabb(bmin, bmax, i, cmin, cmax);
            out->h_err[i] = ora_fit_polynomial(f, cfg, h_coeffs + (uint64_t)i * g_count[degree], 0, cmin, cmax, degree,
                                               depth + 1, literal);
            maxNewErr = maxNewErr > out->h_err[i] ? maxNewErr : out->h_err[i]; /* std::max */
        }
        out->h_imp = (1.0 / (7.0 * (double)g_count[degree])) * (err - 8.0 * maxNewErr); /* :825 */
    }
    /* P */
    if (coarse) {
        out->p_err = ora_fit_polynomial(f, cfg, p_coeffs, 0, bmin, bmax, 2, depth, literal); /* :838-842 */
        out->p_imp = out->p_err;
    } else if (degree >= ORA_BASIS_MAX_DEGREE - 1) {
        /* degree 11: the reference fits 11->12 and never uses it (:600); skipped */
        out->p_imp = 0.0;
    } else {
        memcpy(p_coeffs, coeffs, sizeof(double) * g_count[degree]); /* :847 */
        out->p_err = ora_fit_polynomial(f, cfg, p_coeffs, degree, bmin, bmax, degree + 1, depth, literal);
        out->p_imp = (1.0 / (double)(g_count[degree + 1] - g_count[degree])) * (err - 8.0 * out->p_err); /* :854 */
    }
    /* :600-601 */
    out->refine_p = degree < (ORA_BASIS_MAX_DEGREE - 1) && (depth == ORA_TREE_MAX_DEPTH || out->p_imp > out->h_imp);
    /* A coarse job whose P error is exactly 0 would fall into the H branch
     * with null child bases in the reference (UB): canonical rule:
     * coarse jobs always P-refine. */
    if (coarse) out->refine_p = 1;
    out->refine_h = depth < ORA_TREE_MAX_DEPTH && !out->refine_p;
}

/* Octree::FApprox, Octree.cpp:859-901 */
double ora_fapprox(const double* coeffs, int degree, const float bmin[3], const float bmax[3],
                   const double pt[3], int depth) {
    ora_tables_init();
    double unitPt[3];
    for (int a = 0; a < 3; ++a)
        unitPt[a] = (pt[a] - (double)((bmin[a] + bmax[a]) / 2.0f)) * (double)(2 << depth); /* :862 */
    double LpXLookup[13][3];
    for (int i = 0; i < 3; ++i) {
        LpXLookup[0][i] = g_nl[0][depth];
        double LjMinus2 = 0.0, LjMinus1 = 1.0, Lj = 1.0;
        for (int j = 1; j <= degree; ++j) {
            Lj = g_rec[j][0] * unitPt[i] * LjMinus1 - g_rec[j][1] * LjMinus2; /* :879 */
            LjMinus2 = LjMinus1;
            LjMinus1 = Lj;
            LpXLookup[j][i] = Lj * g_nl[j][depth];
        }
    }
    double fApprox = 0.0;
    for (uint64_t i = 0; i < g_count[degree]; ++i) {
        double Lp = 1.0;
        for (int j = 0; j < 3; ++j) Lp *= LpXLookup[g_bidx[i][j]][j];
        fApprox += coeffs[i] * Lp;
    }
    return fApprox;
}

/* Octree::FApproxWithGradient, Octree.cpp:904-985: value as FApprox; "gradient" by central differences of
 * the per-axis basis factor only (the reference's own shortcut), then normalised. */
double ora_fapprox_with_gradient(const double* coeffs, int degree, const float bmin[3], const float bmax[3],
                                 const double pt[3], int depth, double grad[3]) {
    ora_tables_init();
    double unitPt[3];
    for (int a = 0; a < 3; ++a)
        unitPt[a] = (pt[a] - (double)((bmin[a] + bmax[a]) / 2.0f)) * (double)(2 << depth); /* :907 */
    const double eps = 0.0001;
    double L[13][3][3];
    for (int i = 0; i < 3; ++i) {
        L[0][i][0] = L[0][i][1] = L[0][i][2] = g_nl[0][depth];
        double a2 = 0.0, a1 = 1.0, a0 = 1.0, b2 = 0.0, b1 = 1.0, b0 = 1.0, c2 = 0.0, c1 = 1.0, c0 = 1.0;
        for (int j = 1; j <= degree; ++j) {
            a0 = g_rec[j][0] * unitPt[i] * a1 - g_rec[j][1] * a2; /* :937 */
            a2 = a1;
            a1 = a0;
            b0 = g_rec[j][0] * (unitPt[i] + eps) * b1 - g_rec[j][1] * b2; /* :941 */
            b2 = b1;
            b1 = b0;
            c0 = g_rec[j][0] * (unitPt[i] - eps) * c1 - g_rec[j][1] * c2; /* :945 */
            c2 = c1;
            c1 = c0;
            L[j][i][0] = a0 * g_nl[j][depth];
            L[j][i][1] = b0 * g_nl[j][depth];
            L[j][i][2] = c0 * g_nl[j][depth];
        }
    }
    for (int k = 0; k < 3; ++k) { /* :956-968 */
        double p1 = 0.0, m1 = 0.0;
        for (uint64_t i = 0; i < g_count[degree]; ++i) {
            p1 += coeffs[i] * L[g_bidx[i][k]][k][1];
            m1 += coeffs[i] * L[g_bidx[i][k]][k][2];
        }
        grad[k] = (p1 - m1) / (2.0 * eps);
    }
    { /* Eigen normalize(): divide by sqrt(squaredNorm) when > 0 */
        const double z = sum3(grad[0] * grad[0], grad[1] * grad[1], grad[2] * grad[2]);
        if (z > 0.0) {
            const double n = sqrt(z);
            grad[0] /= n, grad[1] /= n, grad[2] /= n;
        }
    }
    double f = 0.0; /* :972-984 */
    for (uint64_t i = 0; i < g_count[degree]; ++i) {
        double Lp = 1.0;
        for (int j = 0; j < 3; ++j) Lp *= L[g_bidx[i][j]][j][0];
        f += coeffs[i] * Lp;
    }
    return f;
}

/* ======================================================================== */
/* Tree build under the canonical round schedule                             */
/* ======================================================================== */

typedef struct {
    uint64_t idx;
    double err;
} heap_ent;
typedef struct {
    heap_ent* e;
    uint64_t n, cap;
} heap_t;
/* strict total order: larger error first, then smaller node index */
static int ent_before(heap_ent a, heap_ent b) { return a.err > b.err || (a.err == b.err && a.idx < b.idx); }
static void heap_push(heap_t* h, heap_ent v) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 1024;
        h->e = (heap_ent*)realloc(h->e, h->cap * sizeof(heap_ent));
    }
    uint64_t i = h->n++;
    h->e[i] = v;
    while (i > 0) {
        uint64_t p = (i - 1) / 2;
        if (!ent_before(h->e[i], h->e[p])) break;
        heap_ent t = h->e[i];
        h->e[i] = h->e[p];
        h->e[p] = t;
        i = p;
    }
}
static heap_ent heap_pop(heap_t* h) {
    heap_ent top = h->e[0];
    h->e[0] = h->e[--h->n];
    uint64_t i = 0;
    for (;;) {
        uint64_t l = 2 * i + 1, r = l + 1, b = i;
        if (l < h->n && ent_before(h->e[l], h->e[b])) b = l;
        if (r < h->n && ent_before(h->e[r], h->e[b])) b = r;
        if (b == i) break;
        heap_ent t = h->e[i];
        h->e[i] = h->e[b];
        h->e[b] = t;
        i = b;
    }
    return top;
}

typedef struct {
    ora_tree* t;
    double** cptr; /* per-node coefficient pointer during the build (Node::Basis::coeffs) */
    uint64_t cap;
} builder;

static void node_init(ora_node* n) { /* Source/HP/Node.cpp:5-15 (+ zeroed padding) */
    memset(n, 0, sizeof(*n));
    n->childIdx = (uint64_t)-1;
    for (int a = 0; a < 3; ++a) { /* AlignedBox::setEmpty() */
        n->aabb_min[a] = FLT_MAX;
        n->aabb_max[a] = -FLT_MAX;
    }
    n->degree = ORA_INTERIOR_DEGREE;
    n->depth = ORA_TREE_MAX_DEPTH + 1;
}
static void ensure_nodes(builder* b, uint64_t want) {
    ora_tree* t = b->t;
    if (want <= t->cap_nodes) return;
    uint64_t nc = t->cap_nodes ? t->cap_nodes : 8192;
    while (nc < want) nc *= 2;
    t->nodes = (ora_node*)realloc(t->nodes, nc * sizeof(ora_node));
    b->cptr = (double**)realloc(b->cptr, nc * sizeof(double*));
    for (uint64_t i = t->cap_nodes; i < nc; ++i) b->cptr[i] = NULL;
    t->cap_nodes = nc;
}
/* Octree::Subdivide, Octree.cpp:1115-1128 */
static void subdivide(builder* b, uint64_t idx) {
    ora_tree* t = b->t;
    ensure_nodes(b, t->n_nodes + 8);
    t->nodes[idx].childIdx = t->n_nodes;
    for (unsigned i = 0; i < 8; ++i) {
        ora_node* n = &t->nodes[t->n_nodes];
        node_init(n);
        ora_corner_aabb(t->nodes[idx].aabb_min, t->nodes[idx].aabb_max, i, n->aabb_min, n->aabb_max);
        n->depth = (uint8_t)(t->nodes[idx].depth + 1);
        b->cptr[t->n_nodes] = NULL;
        t->n_nodes++;
    }
}
/* Octree::UniformlyRefine, Octree.cpp:112-191: pre-order DFS, subdividing on
 * first visit, to depth 4; leaves get degree 0 and error 100. */
static void uniformly_refine(builder* b, uint64_t idx, unsigned depth, heap_t* h) {
    if (depth < 4) {
        subdivide(b, idx);
        uint64_t c = b->t->nodes[idx].childIdx;
        for (unsigned i = 0; i < 8; ++i) uniformly_refine(b, c + i, depth + 1, h);
    } else {
        b->cptr[idx] = (double*)calloc(g_count[2], sizeof(double)); /* :172 (malloc; zeroed here) */
        b->t->nodes[idx].degree = 0;                               /* :173 */
        heap_ent e = {idx, ORA_INITIAL_NODE_ERR};                  /* :176-177 */
        heap_push(h, e);
    }
}

static int cmp_idx(const void* a, const void* b) {
    uint64_t x = ((const heap_ent*)a)->idx, y = ((const heap_ent*)b)->idx;
    return x < y ? -1 : x > y;
}

static void set_root_vectors(ora_tree* t) {
    for (int a = 0; a < 3; ++a) {
        /* Octree.cpp:322-323 / :419-420: f32 centre, f32 reciprocal of f32 size */
        t->root_centre[a] = (double)((t->config.root_min[a] + t->config.root_max[a]) / 2.0f);
        t->root_inv_sizes[a] = (double)(1.0f / (t->config.root_max[a] - t->config.root_min[a]));
    }
}

/* A round's jobs are pure functions of (cell, degree, error, coefficients, field) (Octree.cpp:594-601, 804-856): with
 * threads > 1 they are evaluated by a pool of pthreads, each job into buffers of its own, and applied afterwards in node
 * order exactly as the single-threaded loop applies them -- same tree, same bytes, for any thread count.  (The all-cores CPU
 * baseline of bench.py; fields given as Python callbacks must stay with threads = 1.) */
typedef struct {
    const ora_field* f;
    const ora_config* cfg;
    const ora_tree* t;
    double** cptr;
    const heap_ent* batch;
    uint64_t want;
    ora_job_result* res;
    double** pb;
    double** hb;
    int literal;
    uint64_t next; /* atomic */
} round_work;
static void* round_worker(void* arg) {
    round_work* w = (round_work*)arg;
    for (;;) {
        const uint64_t bi = __atomic_fetch_add(&w->next, 1, __ATOMIC_RELAXED);
        if (bi >= w->want) break;
        const uint64_t idx = w->batch[bi].idx;
        const ora_node* nd = &w->t->nodes[idx];
        const int p = nd->degree, d = nd->depth;
        const int coarse = fabs(w->batch[bi].err - ORA_INITIAL_NODE_ERR) < DBL_EPSILON;
        const int np = coarse ? 2 : (p + 1 < ORA_BASIS_MAX_DEGREE ? p + 1 : p);
        w->pb[bi] = (double*)malloc(sizeof(double) * g_count[np > p ? np : p]);
        w->hb[bi] = (double*)malloc(sizeof(double) * 8 * g_count[p > 2 ? p : 2]);
        ora_job(w->f, w->cfg, nd->aabb_min, nd->aabb_max, d, p, w->batch[bi].err, w->cptr[idx], w->pb[bi], w->hb[bi], &w->res[bi],
                w->literal);
    }
    return NULL;
}

ora_tree* ora_create(const ora_config* cfg, const ora_field* f, uint64_t K, int literal, ora_build_stats* stats) {
    return ora_create_mt(cfg, f, K, literal, stats, 1);
}

ora_tree* ora_create_mt(const ora_config* cfg, const ora_field* f, uint64_t K, int literal, ora_build_stats* stats, int threads) {
    ora_tables_init();
    ora_build_stats st;
    memset(&st, 0, sizeof(st));
    builder b;
    memset(&b, 0, sizeof(b));
    b.t = (ora_tree*)calloc(1, sizeof(ora_tree));
    ora_tree* t = b.t;
    t->config = *cfg;
    set_root_vectors(t);
    heap_t h = {0, 0, 0};

    /* CreateRoot, Octree.cpp:792-801 */
    ensure_nodes(&b, 1);
    node_init(&t->nodes[0]);
    t->n_nodes = 1;
    t->nodes[0].depth = 0;
    for (int a = 0; a < 3; ++a) {
        t->nodes[0].aabb_min[a] = -0.5f;
        t->nodes[0].aabb_max[a] = 0.5f;
    }
    subdivide(&b, 0);
    /* UniformlyRefine: the reference walks children 0..7 of the root */
    for (unsigned i = 0; i < 8; ++i) uniformly_refine(&b, t->nodes[0].childIdx + i, 1, &h);

    /* RunBuildThreadPool, Octree.cpp:194-309, canonical rounds */
    double total = pow(8, 4) * ORA_INITIAL_NODE_ERR; /* :212 */
    heap_ent* batch = NULL;
    uint64_t batch_cap = 0;
    ora_job_result* res = NULL;
    double **pbs = NULL, **hbs = NULL;
    if (threads < 1) threads = 1;
    pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    for (;;) {
        if (total < cfg->target_error_threshold || h.n == 0) break; /* :216 */
        uint64_t want = (st.rounds == 0) ? h.n : (K < h.n ? K : h.n);
        if (want > batch_cap) {
            batch_cap = want;
            batch = (heap_ent*)realloc(batch, batch_cap * sizeof(heap_ent));
            res = (ora_job_result*)realloc(res, batch_cap * sizeof(ora_job_result));
            pbs = (double**)realloc(pbs, batch_cap * sizeof(double*));
            hbs = (double**)realloc(hbs, batch_cap * sizeof(double*));
        }
        for (uint64_t i = 0; i < want; ++i) batch[i] = heap_pop(&h);
        qsort(batch, want, sizeof(heap_ent), cmp_idx); /* apply in nodeIdx order */
        {   /* evaluate (in parallel), then apply (in order) */
            round_work w = {f, cfg, t, b.cptr, batch, want, res, pbs, hbs, literal, 0};
            const int nt = (uint64_t)threads < want ? threads : (int)want;
            for (int k = 1; k < nt; ++k) pthread_create(&tids[k], NULL, round_worker, &w);
            round_worker(&w);
            for (int k = 1; k < nt; ++k) pthread_join(tids[k], NULL);
        }
        for (uint64_t bi = 0; bi < want; ++bi) {
            const uint64_t idx = batch[bi].idx;
            const double err = batch[bi].err;
            ora_node nd = t->nodes[idx]; /* copy: subdivide() may realloc */
            const int p = nd.degree, d = nd.depth;
            const ora_job_result r = res[bi];
            double *pbuf = pbs[bi], *hbuf = hbs[bi];
            st.jobs++;
            st.fits += r.coarse ? 1 : ((d < ORA_TREE_MAX_DEPTH ? 8 : 0) + (p < ORA_BASIS_MAX_DEGREE - 1 ? 1 : 0));
            if (r.refine_p) { /* :253-260, :286-290 */
                const int np = r.coarse ? 2 : p + 1;
                free(b.cptr[idx]);
                b.cptr[idx] = (double*)malloc(sizeof(double) * g_count[np]);
                memcpy(b.cptr[idx], pbuf, sizeof(double) * g_count[np]);
                t->nodes[idx].degree = (uint8_t)np;
                total += (r.p_err - err);
                heap_ent e = {idx, r.p_err};
                heap_push(&h, e);
                st.p_refines++;
            } else if (r.refine_h) { /* :262-279, :286-290 */
                free(b.cptr[idx]);
                b.cptr[idx] = NULL;
                t->nodes[idx].degree = ORA_INTERIOR_DEGREE;
                subdivide(&b, idx);
                total -= err;
                const uint64_t c0 = t->nodes[idx].childIdx;
                for (unsigned i = 0; i < 8; ++i) {
                    b.cptr[c0 + i] = (double*)malloc(sizeof(double) * g_count[p]);
                    memcpy(b.cptr[c0 + i], hbuf + (uint64_t)i * g_count[p], sizeof(double) * g_count[p]);
                    t->nodes[c0 + i].degree = (uint8_t)p;
                    total += r.h_err[i];
                    heap_ent e = {c0 + i, r.h_err[i]};
                    heap_push(&h, e);
                }
                st.h_refines++;
            } else {
                st.dropped++; /* :643-655 */
            }
            free(pbuf);
            free(hbuf);
        }
        st.rounds++;
    }
    free(batch);
    free(res);
    free(pbs);
    free(hbs);
    free(tids);
    free(h.e);
    st.total_error = total;

    /* ReallocCoeffs, Octree.cpp:474-555: DFS children 0..7, leaves packed in visit order */
    uint64_t nCoeffs = 0;
    for (uint64_t i = 0; i < t->n_nodes; ++i)
        if (t->nodes[i].degree != ORA_INTERIOR_DEGREE) nCoeffs += g_count[t->nodes[i].degree];
    t->n_coeffs = nCoeffs;
    t->coeff_store = (double*)malloc(sizeof(double) * (nCoeffs ? nCoeffs : 1));
    {
        uint64_t cur = 0;
        uint64_t stack_node[ORA_TREE_MAX_DEPTH + 2];
        int stack_child[ORA_TREE_MAX_DEPTH + 2];
        int sp = 0;
        stack_node[0] = 0;
        stack_child[0] = 0;
        while (sp >= 0) {
            if (stack_child[sp] == 8) {
                --sp;
                continue;
            }
            uint64_t n = t->nodes[stack_node[sp]].childIdx + (uint64_t)stack_child[sp]++;
            if (t->nodes[n].childIdx == (uint64_t)-1) {
                uint64_t cnt = g_count[t->nodes[n].degree];
                memcpy(t->coeff_store + cur, b.cptr[n], sizeof(double) * cnt);
                free(b.cptr[n]);
                b.cptr[n] = NULL;
                t->nodes[n].coeffsStart = cur;
                cur += cnt;
            } else {
                ++sp;
                stack_node[sp] = n;
                stack_child[sp] = 0;
            }
        }
    }
    free(b.cptr);
    if (stats) *stats = st;
    return t;
}

void ora_tree_free(ora_tree* t) {
    if (!t) return;
    free(t->nodes);
    free(t->coeff_store);
    free(t);
}

/* ======================================================================== */
/* Serialisation: [u64 nCoeffs][f64 x nCoeffs][u64 nNodes][Node x nNodes][Config] */
/* Octree.cpp:424-456 / 403-421                                              */
/* ======================================================================== */

size_t ora_tree_block_size(const ora_tree* t) {
    return 8 + 8 * (size_t)t->n_coeffs + 8 + sizeof(ora_node) * (size_t)t->n_nodes + sizeof(ora_config);
}
void ora_tree_to_block(const ora_tree* t, void* out) {
    uint8_t* p = (uint8_t*)out;
    memcpy(p, &t->n_coeffs, 8);
    p += 8;
    memcpy(p, t->coeff_store, 8 * (size_t)t->n_coeffs);
    p += 8 * (size_t)t->n_coeffs;
    memcpy(p, &t->n_nodes, 8);
    p += 8;
    for (uint64_t i = 0; i < t->n_nodes; ++i) {
        ora_node n = t->nodes[i];
        memset(n.pad0, 0, sizeof n.pad0);
        memset(n.pad1, 0, sizeof n.pad1);
        if (n.degree == ORA_INTERIOR_DEGREE) n.coeffsStart = 0; /* stale pointer in the reference */
        memcpy(p, &n, sizeof n);
        p += sizeof n;
    }
    ora_config c = t->config;
    memset(c.pad0, 0, sizeof c.pad0);
    memset(c.pad1, 0, sizeof c.pad1);
    memset(c.pad2, 0, sizeof c.pad2);
    memcpy(p, &c, sizeof c);
}
ora_tree* ora_tree_from_block(const void* block, size_t size) {
    if (!block || size < 16 + sizeof(ora_config)) return NULL;
    const uint8_t* p = (const uint8_t*)block;
    ora_tree* t = (ora_tree*)calloc(1, sizeof(ora_tree));
    memcpy(&t->n_coeffs, p, 8);
    p += 8;
    if (8 + 8 * (size_t)t->n_coeffs + 8 > size) {
        free(t);
        return NULL;
    }
    t->coeff_store = (double*)malloc(8 * (size_t)(t->n_coeffs ? t->n_coeffs : 1));
    memcpy(t->coeff_store, p, 8 * (size_t)t->n_coeffs);
    p += 8 * (size_t)t->n_coeffs;
    memcpy(&t->n_nodes, p, 8);
    p += 8;
    if (8 + 8 * (size_t)t->n_coeffs + 8 + sizeof(ora_node) * (size_t)t->n_nodes + sizeof(ora_config) > size) {
        free(t->coeff_store);
        free(t);
        return NULL;
    }
    t->cap_nodes = t->n_nodes;
    t->nodes = (ora_node*)malloc(sizeof(ora_node) * (size_t)(t->n_nodes ? t->n_nodes : 1));
    memcpy(t->nodes, p, sizeof(ora_node) * (size_t)t->n_nodes);
    p += sizeof(ora_node) * (size_t)t->n_nodes;
    memcpy(&t->config, p, sizeof(ora_config));
    set_root_vectors(t); /* :419-420 */
    return t;
}

/* ======================================================================== */
/* Query, Octree.cpp:662-702                                                 */
/* ======================================================================== */

double ora_query(const ora_tree* t, const double pt_[3]) {
    double pt[3];
    for (int a = 0; a < 3; ++a) pt[a] = (pt_[a] - t->root_centre[a]) * t->root_inv_sizes[a]; /* :665 */
    /* :668  contains() on the f32 cast, inclusive */
    for (int a = 0; a < 3; ++a) {
        float pf = (float)pt[a];
        if (!(t->nodes[0].aabb_min[a] <= pf && pf <= t->nodes[0].aabb_max[a])) return DBL_MAX;
    }
    uint64_t cur = 0;
    for (;;) {
        const ora_node* n = &t->nodes[cur];
        const float half = (n->aabb_max[0] - n->aabb_min[0]) * 0.5f; /* :679 x extent for every axis */
        const uint64_t xIdx = (pt[0] >= (double)(n->aabb_min[0] + half));
        const uint64_t yIdx = (uint64_t)(pt[1] >= (double)(n->aabb_min[1] + half)) << 1;
        const uint64_t zIdx = (uint64_t)(pt[2] >= (double)(n->aabb_min[2] + half)) << 2;
        const uint64_t childIdx = n->childIdx + xIdx + yIdx + zIdx;
        const ora_node* c = &t->nodes[childIdx];
        if (c->degree != ORA_INTERIOR_DEGREE)
            return ora_fapprox(t->coeff_store + c->coeffsStart, c->degree, c->aabb_min, c->aabb_max, pt, c->depth);
        cur = childIdx;
    }
}
/* Octree::QueryWithGradient, Octree.cpp:749-789 (grad is left untouched outside the root) */
double ora_query_with_gradient(const ora_tree* t, const double pt_[3], double grad[3]) {
    double pt[3];
    for (int a = 0; a < 3; ++a) pt[a] = (pt_[a] - t->root_centre[a]) * t->root_inv_sizes[a];
    for (int a = 0; a < 3; ++a) {
        float pf = (float)pt[a];
        if (!(t->nodes[0].aabb_min[a] <= pf && pf <= t->nodes[0].aabb_max[a])) return DBL_MAX;
    }
    uint64_t cur = 0;
    for (;;) {
        const ora_node* n = &t->nodes[cur];
        const float half = (n->aabb_max[0] - n->aabb_min[0]) * 0.5f;
        const uint64_t childIdx = n->childIdx + (uint64_t)(pt[0] >= (double)(n->aabb_min[0] + half)) +
                                  ((uint64_t)(pt[1] >= (double)(n->aabb_min[1] + half)) << 1) +
                                  ((uint64_t)(pt[2] >= (double)(n->aabb_min[2] + half)) << 2);
        const ora_node* c = &t->nodes[childIdx];
        if (c->degree != ORA_INTERIOR_DEGREE)
            return ora_fapprox_with_gradient(t->coeff_store + c->coeffsStart, c->degree, c->aabb_min, c->aabb_max, pt,
                                             c->depth, grad);
        cur = childIdx;
    }
}
void ora_query_gradient_batch(const ora_tree* t, const double* xyz, size_t n, double* out, double* grad) {
    for (size_t i = 0; i < n; ++i) out[i] = ora_query_with_gradient(t, xyz + 3 * i, grad + 3 * i);
}
void ora_query_batch(const ora_tree* t, const double* xyz, size_t n, double* out) {
    for (size_t i = 0; i < n; ++i) out[i] = ora_query(t, xyz + 3 * i);
}
/* Octree::Query is const (Octree.h:71): the reference's own parallel use.  The points are cut into `threads` contiguous
 * parts, one pthread each -- the loop of HPBenchmarks.cpp:105-109 on every core, with no interpreter in between. */
typedef struct {
    const ora_tree* t;
    const double* xyz;
    size_t n;
    double* out;
    int passes;
} query_part;
static void* query_worker(void* arg) {
    const query_part* q = (const query_part*)arg;
    for (int p = 0; p < q->passes; ++p) ora_query_batch(q->t, q->xyz, q->n, q->out);
    return NULL;
}
void ora_query_batch_mt(const ora_tree* t, const double* xyz, size_t n, double* out, int threads) {
    ora_query_batch_mt_passes(t, xyz, n, out, threads, 1);
}
/* the same, every thread going over its part `passes` times (a timing loop that starts its threads once: starting and
 * joining 256 threads costs more than one pass over 10 M points) */
void ora_query_batch_mt_passes(const ora_tree* t, const double* xyz, size_t n, double* out, int threads, int passes) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    query_part* parts = (query_part*)malloc(sizeof(query_part) * (size_t)threads);
    for (int k = 0; k < threads; ++k) {
        const size_t a = n * (size_t)k / (size_t)threads, b = n * (size_t)(k + 1) / (size_t)threads;
        parts[k].t = t, parts[k].xyz = xyz + 3 * a, parts[k].n = b - a, parts[k].out = out + a, parts[k].passes = passes;
        if (k) pthread_create(&tids[k], NULL, query_worker, &parts[k]);
    }
    query_worker(&parts[0]);
    for (int k = 1; k < threads; ++k) pthread_join(tids[k], NULL);
    free(tids);
    free(parts);
}
