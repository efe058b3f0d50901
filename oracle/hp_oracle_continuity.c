/*
 * hp_oracle_continuity.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the continuity post-process:
 *   NodeProc / FaceProc                       Octree.cpp:1549-1612
 *   EvaluateSharedFaceIntegralNumerically     Octree.cpp:1250-1456
 *   EvaluateSharedFaceIntegralAnalytically    Octree.cpp:1459-1546
 *   RunContinuityThreadPool                   Octree.cpp:1663-1714
 *   PerformContinuityPostProcess              Octree.cpp:1717-1762
 * single thread, statement order of the reference (the LpX values of a face sample are tabulated per axis;
 * they are the values the reference recomputes inside its innermost loop).
 *
 * PARITY UNPINNED for the solve: the reference hands (M + strength I) x = strength c to Eigen's
 * ConjugateGradient with an IncompleteCholesky preconditioner (Octree.cpp:1751-1755); Eigen is unpinned
 * (git HEAD, Build.sh:5), absent from this image and not restated.  The CG below is the textbook
 * preconditioned iteration with Eigen's stopping rule (|r|^2 < tol^2 |b|^2, initial guess = rhs) and a Jacobi
 * preconditioner; any convergent preconditioner reaches the same solution to O(tol).  The reference's tests pin
 * this path only end to end (|Query - true| <= 1e-2, Source/Tests/HPUnitTests.cpp:80-112,285-316).
 * The matrix assembly is plain reference arithmetic.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "hp_oracle.h"

#define EPSILON_F32 0.000001f /* Include/Utility/Literals.h:14 */

typedef struct {
    uint64_t r, c;
    double v;
    uint64_t seq; /* emission order (ties keep it, so duplicates sum in the order a 1-thread run emits them) */
} triplet;

typedef struct {
    triplet* t;
    uint64_t n, cap;
} tripvec;

static void push(tripvec* v, uint64_t r, uint64_t c, double val) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : (1u << 16);
        v->t = (triplet*)realloc(v->t, v->cap * sizeof(triplet));
    }
    v->t[v->n].r = r;
    v->t[v->n].c = c;
    v->t[v->n].v = val;
    v->t[v->n].seq = v->n;
    v->n++;
}

/* Include/HP/Utility.h:166-196 */
static void shared_face_lookup(unsigned L[3][4][2]) {
    for (unsigned i = 0; i < 3; ++i) {
        unsigned idx0 = 0, idx1 = 0;
        const unsigned modVal1 = 1u << i, modVal = 1u << (i + 1), valsPerMod = 1u << (3 - 1 - i);
        for (unsigned j = 0; j < valsPerMod; ++j) {
            for (unsigned k = 0; k < modVal1; ++k) L[i][idx0++][0] = k + j * modVal;
            for (unsigned k = modVal1; k < modVal; ++k) L[i][idx1++][1] = k + j * modVal;
        }
    }
}

typedef struct {
    uint64_t a, b;
    uint8_t dim;
} facejob;
typedef struct {
    facejob* j;
    uint64_t n, cap;
} jobvec;

static void face_proc(const ora_tree* t, unsigned L[3][4][2], uint64_t A, uint64_t B, unsigned dim, jobvec* out) {
    const ora_node* nA = &t->nodes[A];
    const ora_node* nB = &t->nodes[B];
    const int aHas = nA->childIdx != (uint64_t)-1, bHas = nB->childIdx != (uint64_t)-1;
    if (aHas || bHas) { /* :1583-1588 */
        for (unsigned i = 0; i < 4; ++i)
            face_proc(t, L, aHas ? nA->childIdx + L[dim][i][1] : A, bHas ? nB->childIdx + L[dim][i][0] : B, dim, out);
        return;
    }
    /* :1593-1594 order the pair along dim.  :1597-1604 procMap: a leaf pair shares one face and is reached by
     * one path from the root, and NodeProc(0) visits every node, so the later NodeProc(i) calls of :1677-1680
     * only meet pairs already in the map -- one traversal from the root yields the reference's job list. */
    const uint64_t first = nA->aabb_min[dim] < nB->aabb_min[dim] ? A : B;
    const uint64_t second = nA->aabb_min[dim] < nB->aabb_min[dim] ? B : A;
    if (out->n == out->cap) {
        out->cap = out->cap ? out->cap * 2 : 4096;
        out->j = (facejob*)realloc(out->j, out->cap * sizeof(facejob));
    }
    out->j[out->n].a = first;
    out->j[out->n].b = second;
    out->j[out->n].dim = (uint8_t)dim;
    out->n++;
}

static void node_proc(const ora_tree* t, unsigned L[3][4][2], uint64_t idx, jobvec* out) { /* :1549-1571 */
    const ora_node* n = &t->nodes[idx];
    if (n->childIdx == (uint64_t)-1) return;
    for (unsigned i = 0; i < 8; ++i) node_proc(t, L, n->childIdx + i, out);
    for (unsigned i = 0; i < 3; ++i)
        for (unsigned j = 0; j < 4; ++j) face_proc(t, L, n->childIdx + L[i][j][0], n->childIdx + L[i][j][1], i, out);
}

/* :1459-1546 */
static void integral_analytic(const ora_tree* t, uint64_t A, uint64_t B, unsigned dim, tripvec* out) {
    const uint64_t* cnt = ora_coeff_count();
    const uint64_t(*bi)[3] = (const uint64_t(*)[3])ora_basis_index();
    const double(*nl)[11] = (const double(*)[11])ora_normalised_lengths();
    const ora_node* nA = &t->nodes[A];
    const ora_node* nB = &t->nodes[B];
    const unsigned m1 = (dim + 1) % 3, m2 = (dim + 2) % 3;
    const uint64_t na = cnt[nA->degree], nb = cnt[nB->degree];
    for (uint64_t i = 0; i < na; ++i)
        for (uint64_t j = 0; j < na; ++j) {
            if (bi[i][m1] != bi[j][m1] || bi[i][m2] != bi[j][m2]) continue;
            double integral = 1.0;
            integral *= ora_lpx(bi[i][dim], 1.0);
            integral *= nl[bi[i][dim]][nA->depth];
            integral *= ora_lpx(bi[j][dim], 1.0);
            integral *= nl[bi[j][dim]][nA->depth];
            push(out, nA->coeffsStart + i, nA->coeffsStart + j, integral);
        }
    for (uint64_t i = 0; i < na; ++i)
        for (uint64_t j = 0; j < nb; ++j) {
            if (bi[i][m1] != bi[j][m1] || bi[i][m2] != bi[j][m2]) continue;
            double integral = -1.0;
            integral *= ora_lpx(bi[i][dim], 1.0);
            integral *= nl[bi[i][dim]][nA->depth];
            integral *= ora_lpx(bi[j][dim], -1.0);
            integral *= nl[bi[j][dim]][nB->depth];
            push(out, nA->coeffsStart + i, nB->coeffsStart + j, integral);
            push(out, nB->coeffsStart + j, nA->coeffsStart + i, integral);
        }
    for (uint64_t i = 0; i < nb; ++i)
        for (uint64_t j = 0; j < nb; ++j) {
            if (bi[i][m1] != bi[j][m1] || bi[i][m2] != bi[j][m2]) continue;
            double integral = 1.0;
            integral *= ora_lpx(bi[i][dim], -1.0);
            integral *= nl[bi[i][dim]][nB->depth];
            integral *= ora_lpx(bi[j][dim], -1.0);
            integral *= nl[bi[j][dim]][nB->depth];
            push(out, nB->coeffsStart + i, nB->coeffsStart + j, integral);
        }
}

/* LpX(p, u) for p = 0..12 at every face sample coordinate of one axis (values as Octree::LpX) */
static void lpx_table(double* tab /* [13][n] */, unsigned n, const double* u) {
    for (unsigned p = 0; p <= ORA_BASIS_MAX_DEGREE; ++p)
        for (unsigned q = 0; q < n; ++q) tab[p * n + q] = ora_lpx(p, u[q]);
}

/* :1250-1456 */
static void integral_numeric(const ora_tree* t, uint64_t A, uint64_t B, unsigned dim, tripvec* out) {
    const uint64_t* cnt = ora_coeff_count();
    const uint64_t(*bi)[3] = (const uint64_t(*)[3])ora_basis_index();
    const double(*nl)[11] = (const double(*)[11])ora_normalised_lengths();
    const uint64_t* sumToN = ora_sum_to_n();
    const double* roots = ora_gl_roots();
    const double* weights = ora_gl_weights();
    const ora_node* nA = &t->nodes[A];
    const ora_node* nB = &t->nodes[B];
    const unsigned m1 = (dim + 1) % 3, m2 = (dim + 2) % 3;

    /* :1264-1266 sharedFace = A.clamp(B) (intersection, f32); scale = sizes * 0.5 in f64 */
    double scale[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = nA->aabb_min[a] > nB->aabb_min[a] ? nA->aabb_min[a] : nB->aabb_min[a];
        const float hi = nA->aabb_max[a] < nB->aabb_max[a] ? nA->aabb_max[a] : nB->aabb_max[a];
        scale[a] = (double)(hi - lo) * 0.5;
    }
    const unsigned maxDegree = nA->degree > nB->degree ? nA->degree : nB->degree; /* :1269 */
    const uint64_t gqStart = sumToN[maxDegree], gqEnd = sumToN[maxDegree + 1];
    const unsigned n = (unsigned)(gqEnd - gqStart);
    const unsigned depthDiff = nA->depth > nB->depth ? (unsigned)(nA->depth - nB->depth) : (unsigned)(nB->depth - nA->depth);
    const double invDist = 1.0 / pow(2.0, (double)depthDiff); /* :1275 */
    double invT[3] = {0.0, 0.0, 0.0};                           /* :1278-1289 */
    {
        const ora_node* s = nA->depth > nB->depth ? nA : nB; /* the deeper (smaller) cell */
        const ora_node* l = nA->depth > nB->depth ? nB : nA;
        const unsigned ms[2] = {m1, m2};
        for (int q = 0; q < 2; ++q) {
            const unsigned m = ms[q];
            const float cs = (s->aabb_min[m] + s->aabb_max[m]) / 2.0f, cl = (l->aabb_min[m] + l->aabb_max[m]) / 2.0f;
            invT[m] = (double)(cs - cl) / ((double)(s->aabb_max[m] - s->aabb_min[m]) * 0.5);
        }
        for (int a = 0; a < 3; ++a) invT[a] *= invDist; /* :1290 */
    }
    /* sample coordinates on each side, per in-face axis (m1 <- x, m2 <- y), as :1304-1314 / :1353-1372 */
    double ua1[64], ua2[64], ub1[64], ub2[64];
    for (unsigned q = 0; q < n; ++q) {
        const double r = roots[gqStart + q];
        ua1[q] = ua2[q] = ub1[q] = ub2[q] = r;
        if (nB->depth > nA->depth) {
            ua1[q] = r * invDist + invT[m1];
            ua2[q] = r * invDist + invT[m2];
        } else if (nA->depth > nB->depth) {
            ub1[q] = r * invDist + invT[m1];
            ub2[q] = r * invDist + invT[m2];
        }
    }
    double* tab = (double*)malloc(sizeof(double) * 13 * n * 4);
    double *Ta1 = tab, *Ta2 = tab + 13 * n, *Tb1 = tab + 26 * n, *Tb2 = tab + 39 * n;
    lpx_table(Ta1, n, ua1);
    lpx_table(Ta2, n, ua2);
    lpx_table(Tb1, n, ub1);
    lpx_table(Tb2, n, ub2);
    double faceA[13], faceB[13]; /* LpX(p, +1) on A's side, LpX(p, -1) on B's side */
    for (unsigned p = 0; p <= ORA_BASIS_MAX_DEGREE; ++p) {
        faceA[p] = ora_lpx(p, 1.0);
        faceB[p] = ora_lpx(p, -1.0);
    }
    const uint64_t na = cnt[nA->degree], nb = cnt[nB->degree];

    /* side: 0 = A (dim coordinate +1), 1 = B (-1).  The k loop of :1318-1322 multiplies, for k = 0,1,2 in
     * turn, first the i factor then the j factor of axis k. */
#define FACTOR(side, row, k, x, y)                                                                      \
    ((k) == dim ? ((side) ? faceB[bi[row][k]] : faceA[bi[row][k]])                                      \
                : ((k) == m1 ? ((side) ? Tb1 : Ta1)[bi[row][k] * n + (x)] : ((side) ? Tb2 : Ta2)[bi[row][k] * n + (y)]))
    for (int block = 0; block < 3; ++block) {
        const int si = block == 2 ? 1 : 0, sj = block == 0 ? 0 : 1;
        const uint64_t ni = si ? nb : na, nj = sj ? nb : na;
        const ora_node* ndI = si ? nB : nA;
        const ora_node* ndJ = sj ? nB : nA;
        for (uint64_t i = 0; i < ni; ++i)
            for (uint64_t j = 0; j < nj; ++j) {
                double integral = 0.0;
                for (unsigned x = 0; x < n; ++x)
                    for (unsigned y = 0; y < n; ++y) {
                        double areaVal = weights[gqStart + x] * weights[gqStart + y];
                        for (unsigned k = 0; k < 3; ++k) {
                            areaVal *= FACTOR(si, i, k, x, y);
                            areaVal *= FACTOR(sj, j, k, x, y);
                        }
                        integral += areaVal;
                    }
                double basisWeights = 1.0;
                for (unsigned k = 0; k < 3; ++k) {
                    basisWeights *= nl[bi[i][k]][ndI->depth];
                    basisWeights *= nl[bi[j][k]][ndJ->depth];
                }
                if (block == 1)
                    integral *= scale[m1] * scale[m2] * basisWeights * -1.0; /* :1389 */
                else
                    integral *= scale[m1] * scale[m2] * basisWeights; /* :1334, :1446 */
                if (fabsf((float)integral) > EPSILON_F32) {
                    if (block == 1) {
                        push(out, nA->coeffsStart + i, nB->coeffsStart + j, integral);
                        push(out, nB->coeffsStart + j, nA->coeffsStart + i, integral);
                    } else {
                        push(out, ndI->coeffsStart + i, ndJ->coeffsStart + j, integral);
                    }
                }
            }
    }
#undef FACTOR
    free(tab);
}

static int cmp_triplet(const void* a, const void* b) {
    const triplet* x = (const triplet*)a;
    const triplet* y = (const triplet*)b;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

/* RunContinuityThreadPool (:1663-1714) with one worker, then setFromTriplets (:1732-1735; duplicates summed) */
int ora_continuity_matrix(const ora_tree* t, uint64_t** row_ptr, uint64_t** col, double** val,
                          ora_continuity_stats* stats) {
    ora_tables_init();
    unsigned L[3][4][2];
    shared_face_lookup(L);
    jobvec jobs = {0, 0, 0};
    if (t->n_nodes) node_proc(t, L, 0, &jobs);
    tripvec tv = {0, 0, 0};
    uint64_t nAna = 0, nNum = 0;
    for (uint64_t q = 0; q < jobs.n; ++q) {
        const facejob* j = &jobs.j[q];
        if (t->nodes[j->a].depth == t->nodes[j->b].depth) { /* :1648-1655 */
            integral_analytic(t, j->a, j->b, j->dim, &tv);
            ++nAna;
        } else {
            integral_numeric(t, j->a, j->b, j->dim, &tv);
            ++nNum;
        }
    }
    const uint64_t n = t->n_coeffs;
    qsort(tv.t, tv.n, sizeof(triplet), cmp_triplet);
    uint64_t* rp = (uint64_t*)calloc(n + 1, sizeof(uint64_t));
    uint64_t* ci = (uint64_t*)malloc(sizeof(uint64_t) * (tv.n ? tv.n : 1));
    double* vv = (double*)malloc(sizeof(double) * (tv.n ? tv.n : 1));
    uint64_t nnz = 0;
    for (uint64_t q = 0; q < tv.n;) {
        const uint64_t r = tv.t[q].r, c = tv.t[q].c;
        double s = 0.0;
        while (q < tv.n && tv.t[q].r == r && tv.t[q].c == c) s += tv.t[q++].v;
        ci[nnz] = c;
        vv[nnz] = s;
        ++nnz;
        rp[r + 1]++;
    }
    for (uint64_t r = 0; r < n; ++r) rp[r + 1] += rp[r];
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_pairs = jobs.n;
        stats->n_pairs_analytic = nAna;
        stats->n_pairs_numeric = nNum;
        stats->nnz = nnz;
    }
    free(jobs.j);
    free(tv.t);
    *row_ptr = rp;
    *col = ci;
    *val = vv;
    return 0;
}

static void spmv(uint64_t n, const uint64_t* rp, const uint64_t* ci, const double* v, double shift, const double* x,
                 double* y) {
    for (uint64_t r = 0; r < n; ++r) {
        double s = shift * x[r];
        for (uint64_t q = rp[r]; q < rp[r + 1]; ++q) s += v[q] * x[ci[q]];
        y[r] = s;
    }
}
static double dot(uint64_t n, const double* a, const double* b) {
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* PerformContinuityPostProcess, :1717-1762 */
int ora_continuity_post_process(ora_tree* t, double tol, int max_iter, ora_continuity_stats* stats) {
    uint64_t *rp, *ci;
    double* v;
    ora_continuity_stats st;
    ora_continuity_matrix(t, &rp, &ci, &v, &st);
    const uint64_t n = t->n_coeffs;
    const double lambda = t->config.continuity_strength; /* :1724-1729 regularisation on the diagonal */
    double* b = (double*)malloc(sizeof(double) * (n ? n : 1));
    double* x = (double*)malloc(sizeof(double) * (n ? n : 1));
    double* r = (double*)malloc(sizeof(double) * (n ? n : 1));
    double* p = (double*)malloc(sizeof(double) * (n ? n : 1));
    double* z = (double*)malloc(sizeof(double) * (n ? n : 1));
    double* tmp = (double*)malloc(sizeof(double) * (n ? n : 1));
    double* dinv = (double*)malloc(sizeof(double) * (n ? n : 1));
    for (uint64_t i = 0; i < n; ++i) {
        b[i] = t->coeff_store[i] * lambda; /* :1738-1741 */
        x[i] = b[i];                        /* solveWithGuess(oldCoeffs, oldCoeffs), :1755 */
        double d = lambda;
        for (uint64_t q = rp[i]; q < rp[i + 1]; ++q)
            if (ci[q] == i) d += v[q];
        dinv[i] = 1.0 / d;
    }
    spmv(n, rp, ci, v, 0.0, t->coeff_store, tmp);
    st.jump_before = dot(n, t->coeff_store, tmp);
    if (max_iter <= 0) max_iter = (int)(2 * n); /* Eigen default */
    /* Eigen's conjugate_gradient() with a diagonal preconditioner */
    spmv(n, rp, ci, v, lambda, x, tmp);
    for (uint64_t i = 0; i < n; ++i) r[i] = b[i] - tmp[i];
    const double rhsNorm2 = dot(n, b, b);
    int it = 0;
    double resNorm2 = dot(n, r, r);
    if (rhsNorm2 == 0.0) {
        memset(x, 0, sizeof(double) * n);
        resNorm2 = 0.0;
    } else {
        const double threshold = fmax(tol * tol * rhsNorm2, DBL_MIN);
        if (!(resNorm2 < threshold)) {
            for (uint64_t i = 0; i < n; ++i) p[i] = dinv[i] * r[i];
            double absNew = dot(n, r, p);
            while (it < max_iter) {
                spmv(n, rp, ci, v, lambda, p, tmp);
                const double alpha = absNew / dot(n, p, tmp);
                for (uint64_t i = 0; i < n; ++i) x[i] += alpha * p[i];
                for (uint64_t i = 0; i < n; ++i) r[i] -= alpha * tmp[i];
                resNorm2 = dot(n, r, r);
                if (resNorm2 < threshold) break;
                for (uint64_t i = 0; i < n; ++i) z[i] = dinv[i] * r[i];
                const double absOld = absNew;
                absNew = dot(n, r, z);
                const double beta = absNew / absOld;
                for (uint64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
                ++it;
            }
        }
    }
    memcpy(t->coeff_store, x, sizeof(double) * n); /* :1756 */
    spmv(n, rp, ci, v, 0.0, x, tmp);
    st.jump_after = dot(n, x, tmp);
    st.iterations = (uint64_t)it;
    st.residual = rhsNorm2 > 0.0 ? sqrt(resNorm2 / rhsNorm2) : 0.0;
    if (stats) *stats = st;
    free(b), free(x), free(r), free(p), free(z), free(tmp), free(dinv);
    free(rp), free(ci), free(v);
    return it;
}
