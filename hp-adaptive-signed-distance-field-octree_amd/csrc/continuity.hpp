// Host-side continuity post-process (continuity.cpp); see include/hpsdf.h for the C entry points.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/hpsdf.h"

namespace hpsdf {

// In place on a serialised MemoryBlock.  tol <= 0: EPSILON_F32; maxIter <= 0: 2n; threads 0: config.threadCount.
// ctx != nullptr: the conjugate-gradient loop runs on that context's device (cg.hip) -- same arithmetic, same bits.
int continuityPostProcess(void* block, size_t size, double tol, int maxIter, uint64_t threads,
                          hpsdf_continuity_stats* stats, std::string& err, hpsdf_ctx* ctx = nullptr);
// The jump-energy matrix M (no regularisation) as CSR in malloc'd arrays (caller frees).
int continuityMatrix(const void* block, size_t size, uint64_t threads, uint64_t** rowPtr, uint64_t** col, double** val,
                     hpsdf_continuity_stats* stats, std::string& err);

}  // namespace hpsdf
