// Analytic fields on the device: CSG of primitives, evaluated in f64 with the reference's operation order (no fused
// multiply-add; Eigen's 3-vector reductions as a + (b + c)).  Shared by kernels.hip and fit_mfma.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace hpsdf {

// Eigen's Vector3d::norm(): sqrt(x^2 + (y^2 + z^2))
__device__ __forceinline__ double norm3(double x, double y, double z) { return sqrt(x * x + (y * y + z * z)); }

__device__ __forceinline__ double primEval(const hpsdf_prim& pr, double x, double y, double z) {
    const double* p = pr.p;
    switch (pr.kind) {
        case HPSDF_PRIM_SPHERE:
            return norm3(x - p[0], y - p[1], z - p[2]) - p[3];
        case HPSDF_PRIM_BOX: {
            const double qx = fabs(x - p[0]) - p[3];
            const double qy = fabs(y - p[1]) - p[4];
            const double qz = fabs(z - p[2]) - p[5];
            const double outside = norm3(fmax(qx, 0.0), fmax(qy, 0.0), fmax(qz, 0.0));
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

__device__ __forceinline__ double analyticEval(const FieldDev& f, double x, double y, double z) {
    double acc = primEval(f.prims[0], x, y, z);
    for (int i = 1; i < f.nPrims; ++i) {
        const double d = primEval(f.prims[i], x, y, z);
        switch (f.prims[i].op) {
            case HPSDF_OP_UNION: acc = fmin(acc, d); break;
            case HPSDF_OP_INTERSECT: acc = fmax(acc, d); break;
            default: acc = fmax(acc, -d); break;
        }
    }
    return acc;
}

}  // namespace hpsdf
