/*
 * hp_oracle_ray.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the query-side helpers that sit on top of Octree::Query
 * (SURVEY.md 8f-4): Ray / Ray::IntersectAABB, Octree::QueryRay and the pixel
 * arithmetic of Octree::OutputFunctionSlice.  Citations are file:line of the
 * reference checkout.
 *
 * Pinning: the reference holds no test of QueryRay (its header marks it
 * untested, Include/HP/Octree.h:73-75) and OutputFunctionSlice needs stb, which
 * is absent: PARITY UNPINNED beyond "same arithmetic, statement by statement".
 * The reference's own quirks are kept on purpose:
 *   - the ray origin is moved to the unit cube, the direction is not (:711);
 *   - when the origin is outside the root, `intMin` holds the slab parameters
 *     Ray::IntersectAABB leaves in its first output (x = entry parameter after
 *     the clamps, y/z = per-axis entry parameters), not a point (:717-720);
 *   - Query() is then called with that unit-cube value and maps it through the
 *     root transform a second time (:726, :665);
 *   - on a hit t_ receives the field value, not the ray parameter (:730).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "hp_oracle.h"

/* Ray::Ray, Source/HP/Ray.cpp:5-15 */
typedef struct {
    double origin[3], direction[3], inv[3];
    int sign[3];
} ray_t;

static void ray_init(ray_t* r, const double o[3], const double d[3]) {
    for (int a = 0; a < 3; ++a) {
        r->origin[a] = o[a];
        r->direction[a] = d[a];
        r->inv[a] = 1.0 / d[a]; /* cwiseInverse */
        r->sign[a] = r->inv[a] < 0.0;
    }
}

/* Ray::IntersectAABB, Source/HP/Ray.cpp:18-68 */
static int ray_intersect_aabb(const ray_t* r, const double bmin[3], const double bmax[3], double a[3], double b[3]) {
    const double* bounds[2] = {bmin, bmax};
    a[0] = (bounds[r->sign[0]][0] - r->origin[0]) * r->inv[0];
    b[0] = (bounds[1 - r->sign[0]][0] - r->origin[0]) * r->inv[0];
    a[1] = (bounds[r->sign[1]][1] - r->origin[1]) * r->inv[1];
    b[1] = (bounds[1 - r->sign[1]][1] - r->origin[1]) * r->inv[1];
    if ((a[0] > b[1]) || (a[1] > b[0])) return 0;
    if (a[1] > a[0]) a[0] = a[1];
    if (b[1] < b[0]) b[0] = b[1];
    a[2] = (bounds[r->sign[2]][2] - r->origin[2]) * r->inv[2];
    b[2] = (bounds[1 - r->sign[2]][2] - r->origin[2]) * r->inv[2];
    if ((a[0] > b[2]) || (a[2] > b[0])) return 0;
    if (a[2] > a[0]) a[0] = a[2];
    if (b[2] < b[0]) b[0] = b[2];
    return 1;
}

/* Octree::QueryRay, Octree.cpp:705-746.  Returns 1 on a hit (t_out written), 0 otherwise (t_out untouched). */
int ora_query_ray(const ora_tree* t, const double origin[3], const double direction[3], double t_max, double* t_out) {
    const unsigned MAX_STEPS = 200;
    const double eps = 0.0001, minStep = 0.0001;
    double o[3];
    for (int a = 0; a < 3; ++a) o[a] = (origin[a] - t->root_centre[a]) * t->root_inv_sizes[a]; /* :711 */
    ray_t ray;
    ray_init(&ray, o, direction);
    double intMin[3] = {ray.origin[0], ray.origin[1], ray.origin[2]}, intMax[3];
    /* :717  !contains(origin.cast<f32>()) && !IntersectAABB(nodes[0].aabb.cast<f64>(), ...) */
    int inside = 1;
    for (int a = 0; a < 3; ++a) {
        const float pf = (float)ray.origin[a];
        if (!(t->nodes[0].aabb_min[a] <= pf && pf <= t->nodes[0].aabb_max[a])) inside = 0;
    }
    if (!inside) {
        const double bmin[3] = {(double)t->nodes[0].aabb_min[0], (double)t->nodes[0].aabb_min[1], (double)t->nodes[0].aabb_min[2]};
        const double bmax[3] = {(double)t->nodes[0].aabb_max[0], (double)t->nodes[0].aabb_max[1], (double)t->nodes[0].aabb_max[2]};
        if (!ray_intersect_aabb(&ray, bmin, bmax, intMin, intMax)) return 0;
    }
    double d = 0.0;
    for (unsigned i = 0; i < MAX_STEPS; ++i) {
        const double pt[3] = {intMin[0] + d * ray.direction[0], intMin[1] + d * ray.direction[1],
                              intMin[2] + d * ray.direction[2]};
        const double v = ora_query(t, pt); /* :726 */
        if (v < eps) {
            *t_out = v; /* :730 */
            return 1;
        }
        d += v * 0.95 + minStep; /* :736 */
        if (d > t_max) return 0;
    }
    return 0;
}

void ora_query_ray_batch(const ora_tree* t, const double* origins, const double* directions, const double* t_max,
                         size_t n, uint8_t* hit, double* t_out) {
    for (size_t i = 0; i < n; ++i) hit[i] = (uint8_t)ora_query_ray(t, origins + 3 * i, directions + 3 * i, t_max[i], t_out + i);
}

/* Octree::OutputFunctionSlice, Octree.cpp:1131-1206, up to the byte image (stb's BMP writer is not restated).
 * n_samples is 2048 in the reference; rgb: n*n*3 bytes, values: n*n doubles (may be NULL). */
void ora_function_slice(const ora_tree* t, double c, const float view_min[3], const float view_max[3],
                        uint64_t n_samples, uint8_t* rgb, double* values) {
    const uint64_t n = n_samples;
    double* sdf = values ? values : (double*)malloc(sizeof(double) * n * n);
    double posFirst = DBL_MAX, posSecond = 0.0;  /* minMaxPosVals :1140 */
    double negFirst = 0.0, negSecond = DBL_MAX * -1.0; /* minMaxNegVals :1141 */
    for (uint64_t i = 0; i < n; ++i)
        for (uint64_t j = 0; j < n; ++j) {
            double s[3] = {(double)view_min[0], (double)view_min[1], (double)view_min[2]};
            const float step = (view_max[0] - view_min[0]) / (float)n; /* :1149: f32 / u32 */
            s[0] += (double)((float)j * step);                         /* u32 * f32 -> f32 */
            s[1] += (double)((float)i * step);
            s[2] = c;
            const double v = ora_query(t, s);
            if (v > (double)0.000001f) { /* EPSILON_F32 */
                posFirst = v < posFirst ? v : posFirst;
                posSecond = posSecond < v ? v : posSecond;
            } else {
                negFirst = v < negFirst ? v : negFirst;
                negSecond = negSecond < v ? v : negSecond;
            }
            sdf[i * n + j] = v;
        }
    for (uint64_t i = 0; i < n; ++i)
        for (uint64_t j = 0; j < n; ++j) {
            const float u = (float)sdf[i * n + j];
            uint8_t* px = rgb + 3 * (i * n + j);
            if (u > 0.0f) {
                /* (u8)(255 * (f32 - f64) / (f64 - f64)): double arithmetic, :1181 */
                const double q = 255 * ((double)u - posSecond) / (posFirst - posSecond);
                px[0] = 0, px[1] = ora_f64_to_u8(q), px[2] = 0;
            } else {
                const double q = 255 * ((double)u - negFirst) / (negSecond - negFirst);
                px[0] = 0, px[1] = 0, px[2] = ora_f64_to_u8(q);
            }
        }
    if (!values) free(sdf);
}

/* (u8)double as x86-64 GCC compiles it: cvttsd2si to a 32-bit int (out of range / NaN -> INT_MIN), low byte kept.
 * Strictly the cast is undefined outside [0,255]; the min/max normalisation keeps it inside except for NaN
 * (0/0 when all values of one sign are equal) and DBL_MAX pixels outside the root. */
uint8_t ora_f64_to_u8(double q) {
    int32_t v;
    if (!(q > -2147483649.0 && q < 2147483648.0)) v = INT32_MIN; /* also NaN */
    else v = (int32_t)q;
    return (uint8_t)(v & 0xFF);
}
