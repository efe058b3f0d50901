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

// The same matrix assembled on ctx's device (continuity_asm.hip) and copied back: the host assembly's arrays bit for bit.
// HPSDF_ERR_UNSUPPORTED when the tree is one the device assembly leaves to the host.
int continuityMatrixDevice(hpsdf_ctx* ctx, const void* block, size_t size, uint64_t** rowPtr, uint64_t** col, double** val,
                           hpsdf_continuity_stats* stats, std::string& err);

// M in HBM, CSR with 32-bit columns, and the buffers behind it (kept by the context between post-processes)
struct ContinuityDeviceMatrix {
    char* base = nullptr;  // tree, incidents, own blocks, row pointer
    uint64_t cap = 0;
    char* entries = nullptr;  // columns and values
    uint64_t entryCap = 0;
    void* scanTmp = nullptr;
    size_t scanCap = 0;
    int device = -1;
    uint64_t n = 0, nnz = 0, maxRow = 0;  // maxRow: entries of the longest row
    const uint64_t* dRowPtr = nullptr;
    const uint32_t* dCol = nullptr;
    const double* dVal = nullptr;
    ~ContinuityDeviceMatrix();
};
// *fallback = 1: this tree is left to the host assembler (nothing else is reported then)
int continuityAssembleDevice(hpsdf_ctx* ctx, const hpsdf_node* nodes, uint64_t nNodes, uint64_t nCoeffs, ContinuityDeviceMatrix& out,
                             hpsdf_continuity_stats& st, int* fallback, std::string& err);

}  // namespace hpsdf
