// acosf as the HOST's libm computes it, for the device.
//
// Mesh::SignedDistance weighs a vertex's incident face normals by their angles, std::acos of a float
// (Source/Meshing/Mesh.cpp:226-231); the sign of the distance follows from that sum.  A device libm's acosf differs
// from the host's in the last place on part of its domain, so a distance whose pseudo-normal is nearly perpendicular
// to (point - closest point) could change sign between the CPU build and this one.  The reference runs on glibc, whose
// float acos (2.35, sysdeps/ieee754/flt-32/e_acosf.c) is the fdlibm rational approximation below: R(z) = p(z) / q(z) on
// |x| < 0.5, and the half-angle forms 2 asin(sqrt((1 -+ x) / 2)) beyond, all in float operations.  Restated here it gives
// the host's bits on the device as long as +, *, /, sqrt are the IEEE ones (they are: -ffp-contract=off, correctly
// rounded f32 divide and square root are hipcc's defaults, denormals are kept).
// tests/native/acosf_exhaustive.c compiles this file with gcc and compares it with the libm of the machine for every float
// in [-1, 1]; tests/test_gpu_parity.py compares the device's values.
#pragma once
#include <stdint.h>
#if !defined(__HIPCC__)
#include <math.h>
#endif

#if defined(__HIPCC__)
#define HPSDF_ACOS_FN __host__ __device__ inline
#else
#define HPSDF_ACOS_FN static inline
#endif

HPSDF_ACOS_FN float hpsdfAcosfBits(uint32_t u) {
    union {
        uint32_t u;
        float f;
    } c;
    c.u = u;
    return c.f;
}
HPSDF_ACOS_FN uint32_t hpsdfAcosfWord(float f) {
    union {
        uint32_t u;
        float f;
    } c;
    c.f = f;
    return c.u;
}
// (sqrtf: on the device v_sqrt_f32 plus the last-place correction, i.e. correctly rounded like the host's sqrtss.  HIP's
// __fsqrt_rn is the bare 1-ulp instruction in this toolchain: 5 results in 100 000 then differ from the host's)
#define HPSDF_ACOS_SQRT(z) sqrtf(z)

HPSDF_ACOS_FN float hpsdfAcosf(float x) {
    const float pi = hpsdfAcosfBits(0x40490fdau), pio2Hi = hpsdfAcosfBits(0x3fc90fdau), pio2Lo = hpsdfAcosfBits(0x33a22168u);
    const float p0 = hpsdfAcosfBits(0x3e2aaaabu), p1 = hpsdfAcosfBits(0xbea6b090u), p2 = hpsdfAcosfBits(0x3e4e0aa8u),
                p3 = hpsdfAcosfBits(0xbd241146u), p4 = hpsdfAcosfBits(0x3a4f7f04u), p5 = hpsdfAcosfBits(0x3811ef08u);
    const float q1 = hpsdfAcosfBits(0xc019d139u), q2 = hpsdfAcosfBits(0x4001572du), q3 = hpsdfAcosfBits(0xbf303361u),
                q4 = hpsdfAcosfBits(0x3d9dc62eu);
    const uint32_t hx = hpsdfAcosfWord(x), ix = hx & 0x7fffffffu;
    const int neg = (int)(hx >> 31);
    if (ix == 0x3f800000u) return neg ? pi + 2.0f * pio2Lo : 0.0f;
    if (ix > 0x3f800000u) return (x - x) / (x - x);  // NaN outside [-1, 1]
    if (ix < 0x3f000000u) {
        if (ix <= 0x23000000u) return pio2Hi + pio2Lo;
        const float z = x * x;
        const float p = z * (p0 + z * (p1 + z * (p2 + z * (p3 + z * (p4 + z * p5)))));
        const float q = 1.0f + z * (q1 + z * (q2 + z * (q3 + z * q4)));
        const float r = p / q;
        return pio2Hi - (x - (pio2Lo - x * r));
    }
    const float z = (neg ? 1.0f + x : 1.0f - x) * 0.5f;
    const float p = z * (p0 + z * (p1 + z * (p2 + z * (p3 + z * (p4 + z * p5)))));
    const float q = 1.0f + z * (q1 + z * (q2 + z * (q3 + z * q4)));
    const float s = HPSDF_ACOS_SQRT(z);
    const float r = p / q;
    if (neg) {
        const float w = r * s - pio2Lo;
        return pi - 2.0f * (s + w);
    }
    const float df = hpsdfAcosfBits(hpsdfAcosfWord(s) & 0xfffff000u);  // the square root's leading 12 bits: df * df is exact
    const float c = (z - df * df) / (s + df);
    const float w = r * s + c;
    return 2.0f * (df + w);
}
