// TEST INFRASTRUCTURE ONLY -- not part of the product.
//
// Exposes the reference's *own* constant tables through a C ABI so that the
// oracle restatement (oracle/hp_oracle.c) and the product tables can be pinned
// against them.  This translation unit contains no reference code: it only
// #includes the two reference headers that compile without Eigen, from where
// they lie under /root/reference (see oracle/Makefile; output goes to
// oracle/_ref/, which is git-ignored).
//
//   Include/HP/Utility.h:40-57   SumToN
//   Include/HP/Utility.h:63-78   NormalisedLengths
//   Include/HP/Utility.h:87-106  LegendreCoeffientCount
//   Include/HP/Utility.h:112-127 LegendreCoefficent
//   Include/HP/Utility.h:133-160 BasisIndexValues
//   Include/HP/Legendre.h:7,2091 LegendreRoots / LegendreWeights
//   Include/HP/Consts.h:7-8, Include/Utility/Literals.h:3-13, Include/Utility/MemoryBlock.h:5-9   scalars
#include <cstddef>
#include <cstring>
#include "HP/Utility.h"
#include "HP/Legendre.h"
#include "Utility/MemoryBlock.h"

extern "C" {

int ref_basis_max_degree(void) { return (int)SDF::BASIS_MAX_DEGREE; }
int ref_tree_max_depth(void) { return (int)SDF::TREE_MAX_DEPTH; }
int ref_sizeof_u32(void) { return (int)sizeof(u32); }
// out[8]: BASIS_MAX_DEGREE, TREE_MAX_DEPTH, sizeof i16 / i32 / u32 / usize / MemoryBlock, the bits of EPSILON_F32
void ref_scalars(unsigned long long* out) {
    const float eps = EPSILON_F32;
    unsigned int bits;
    std::memcpy(&bits, &eps, 4);
    out[0] = SDF::BASIS_MAX_DEGREE, out[1] = SDF::TREE_MAX_DEPTH;
    out[2] = sizeof(i16), out[3] = sizeof(i32), out[4] = sizeof(u32), out[5] = sizeof(usize), out[6] = sizeof(MemoryBlock);
    out[7] = bits;
}

// out[50]
void ref_sum_to_n(unsigned long long* out) {
    for (int i = 0; i < 50; ++i) out[i] = SDF::SumToN[i];
}
// out[13*11]
void ref_normalised_lengths(double* out) {
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 11; ++j) out[i * 11 + j] = SDF::NormalisedLengths[i][j];
}
// out[13]
void ref_coeff_count(unsigned long long* out) {
    for (int i = 0; i < 13; ++i) out[i] = SDF::LegendreCoeffientCount[i];
}
// out[13*2]
void ref_recurrence(double* out) {
    for (int i = 0; i < 13; ++i) {
        out[2 * i] = SDF::LegendreCoefficent[i][0];
        out[2 * i + 1] = SDF::LegendreCoefficent[i][1];
    }
}
// out[455*3]
void ref_basis_index(unsigned long long* out) {
    for (int i = 0; i < 455; ++i)
        for (int j = 0; j < 3; ++j) out[i * 3 + j] = SDF::BasisIndexValues[i][j];
}
// out[2080] each
void ref_gl(double* roots, double* weights) {
    std::memcpy(roots, SDF::LegendreRoots, sizeof(double) * 2080);
    std::memcpy(weights, SDF::LegendreWeights, sizeof(double) * 2080);
}
}
