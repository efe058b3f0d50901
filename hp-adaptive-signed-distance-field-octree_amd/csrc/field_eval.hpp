// Analytic fields on the device: CSG of primitives, evaluated in f64 with the reference's operation order (no fused
// multiply-add; Eigen's 3-vector reductions as a + (b + c), or (a + b) + c under hpsdf_set_reduction_order(1)).  Shared by kernels.hip and fit_mfma.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace hpsdf {

// Eigen's 3-vector reductions: a . (b . c) (a scalar build; the default) or (a . b) . c (LEFT: a Packet2d build).  A template
// parameter of the kernels, not a run-time select: the select cost the fit kernels 2-4 % (profiles/r04_fit_ab_builds.txt),
// so the launchers pick the instantiation from FieldDev::leftAssoc and the default order's code is what it was without the switch.
template <bool LEFT>
__device__ __forceinline__ double sum3(double a, double b, double c) {
    if constexpr (LEFT) return (a + b) + c;
    else return a + (b + c);
}
template <bool LEFT>
__device__ __forceinline__ double prod3(double a, double b, double c) {
    if constexpr (LEFT) return (a * b) * c;
    else return a * (b * c);
}
// sqrt(x), correctly rounded like the compiler's own expansion (which it restates: v_rsq_f64, one Goldschmidt step on g ~ sqrt x and
// h ~ 1 / (2 sqrt x), two residual corrections), without that expansion's input scaling (x < 2^-767 is multiplied by 2^256 first) and
// its special-value selects (0, inf, NaN pass through) when no lane of the wave needs them -- which a field's squared distances never
// do in practice: 12 instructions instead of 18, four square roots a sample in the union3 field.  Any lane with such an input sends
// the whole wave through the compiler's version.
__device__ __forceinline__ double sqrtExact(double x) {
    const uint32_t hi = (uint32_t)__double2hiint(x);
    const bool plain = hi - 0x10000000u < 0x7FF00000u - 0x10000000u;  // 2^-767 <= x < inf (and not NaN, not negative)
    if (__builtin_amdgcn_ballot_w64(!plain) != 0ull) return sqrt(x);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// Eigen's Vector3d::norm()
template <bool LEFT>
__device__ __forceinline__ double norm3(double x, double y, double z) { return sqrtExact(sum3<LEFT>(x * x, y * y, z * z)); }

template <bool LEFT>
__device__ __forceinline__ double primEval(const hpsdf_prim& pr, double x, double y, double z) {
    const double* p = pr.p;
    switch (pr.kind) {
        case HPSDF_PRIM_SPHERE:
            return norm3<LEFT>(x - p[0], y - p[1], z - p[2]) - p[3];
        case HPSDF_PRIM_BOX: {
            const double qx = fabs(x - p[0]) - p[3];
            const double qy = fabs(y - p[1]) - p[4];
            const double qz = fabs(z - p[2]) - p[5];
            const double outside = norm3<LEFT>(fmax(qx, 0.0), fmax(qy, 0.0), fmax(qz, 0.0));
            const double inside = fmin(fmax(qx, fmax(qy, qz)), 0.0);
            return outside + inside;
        }
        case HPSDF_PRIM_TORUS_Y: {
            const double dx = x - p[0], dy = y - p[1], dz = z - p[2];
            const double l = sqrtExact(dx * dx + dz * dz) - p[3];
            return sqrtExact(l * l + dy * dy) - p[4];
        }
        case HPSDF_PRIM_PLANE:
            return (p[0] * x + (p[1] * y + p[2] * z)) + p[3];
        default:
            return 0.0;
    }
}

template <bool LEFT>
__device__ __forceinline__ double analyticEval(const FieldDev& f, double x, double y, double z) {
    double acc = primEval<LEFT>(f.prims[0], x, y, z);
    for (int i = 1; i < f.nPrims; ++i) {
        const double d = primEval<LEFT>(f.prims[i], x, y, z);
        switch (f.prims[i].op) {
            case HPSDF_OP_UNION: acc = fmin(acc, d); break;
            case HPSDF_OP_INTERSECT: acc = fmax(acc, d); break;
            default: acc = fmax(acc, -d); break;
        }
    }
    return acc;
}

}  // namespace hpsdf
