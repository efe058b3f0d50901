// Constant tables of the hp-adaptive SDF octree, generated at start-up.
//
// What the reference holds as constexpr tables and a 4000-line literal header
// (Include/HP/Utility.h:40-160, Include/HP/Legendre.h:7-4173) is produced here
// by construction: double-double Newton for the Gauss-Legendre rules (correctly
// rounded, so identical to the reference's ~290-digit literals), the same
// 100-step Newton square root for the normalisation table, and the same
// floating-point expression for the coefficient counts (which yields 83, not 84,
// for degree 6).  tests/test_tables.py pins all of them bit-for-bit.
#pragma once
#include <cstdint>

namespace hpsdf {

constexpr int kMaxDegree = 12;       // Include/HP/Consts.h:7
constexpr int kMaxDepth = 10;        // Include/HP/Consts.h:8
constexpr int kInteriorDegree = 13;  // BASIS_MAX_DEGREE + 1
constexpr int kMaxCoeffs = 455;
constexpr int kGLTotal = 2080;       // rules n = 1..64 concatenated
constexpr int kMaxGL1D = 49;         // n = 4 * 12 + 1

struct Tables {
    double roots[kGLTotal];
    double weights[kGLTotal];
    double normalisedLengths[kMaxDegree + 1][kMaxDepth + 1];  // sqrt((2i+1) * 2^j)
    double recurrence[kMaxDegree + 1][2];                      // (2i-1)/i, (i-1)/i
    uint64_t coeffCount[kMaxDegree + 1];                       // {1,4,10,20,35,56,83,120,...}
    uint64_t basisIndex[kMaxCoeffs][3];                        // graded order
    uint64_t sumToN[4 * kMaxDegree + 2];
};

const Tables& tables();  // thread-safe lazy singleton

inline int glOffset(int n) { return n * (n - 1) / 2; }

}  // namespace hpsdf
