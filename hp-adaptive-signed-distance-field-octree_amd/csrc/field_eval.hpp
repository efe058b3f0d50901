// Analytic fields on the device: CSG of primitives, evaluated in f64 with the reference's operation order (no fused
// multiply-add; Eigen's 3-vector reductions as a + (b + c), or (a + b) + c under hpsdf_set_reduction_order(1)).  Shared by kernels.hip and fit_mfma.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace hpsdf {

// Eigen's 3-vector reductions: a . (b . c) (a scalar build; the default) or (a . b) . c (LEFT: a Packet2d build).  A template
// parameter of the kernels, not a run-time select: the select cost the fit kernels 2-4 % (profiles/r04_reduction_order_switch.txt),
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
// Eigen's Vector3d::norm()
template <bool LEFT>
__device__ __forceinline__ double norm3(double x, double y, double z) { return sqrt(sum3<LEFT>(x * x, y * y, z * z)); }

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
            const double l = sqrt(dx * dx + dz * dz) - p[3];
            return sqrt(l * l + dy * dy) - p[4];
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
