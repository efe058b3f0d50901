// Host-side launch wrappers of kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "device_types.hpp"

namespace hpsdf {

constexpr size_t kFitMaxLdsBytes = 60 * 1024;  // stays under the 64 KiB default dynamic-LDS limit
constexpr int kFitBlockThreads = 256;
constexpr unsigned kQueryMaxGrid = 256 * 8 * 4;  // workgroups of one Query launch (grid-stride beyond)

constexpr size_t kFitChunkLdsBytes = 40 * 1024;  // sample planes staged per chunk (keeps >= 3 workgroups per CU)
size_t fitLdsBytes(int degree, int nTasks, int planes);
struct FitShape {
    int cells;           // fits per workgroup
    int cellsPerThread;  // 1, or 4 with cell blocking
    int planes;          // i-planes staged per chunk
    size_t ldsBytes;
};
// latencyBound: the field is a BVH traversal (mesh): one cell per workgroup, occupancy hides the gathers
FitShape fitShape(int degree, int nrows, uint32_t count, bool weighted, bool latencyBound = false);


hipError_t launchFit(hipStream_t stream, int degree, int cellsPerThread, const FitBlock* dBlocks, uint32_t nBlocks, size_t ldsBytes,
                     const FitTask* dTasks, double* dArena, double* dErrs, double* dMeans,
                     const DeviceTables* dTables, const FieldDev& field, const RootMap& rm);
hipError_t launchQuery(hipStream_t stream, const TreeDev& t, const DeviceTables* dTables, const double* dXyz, size_t n,
                       double* dOut, double* dGrad, bool allInline, uint32_t* dDeferCount, uint32_t* dDeferIdx);
hipError_t launchQueryRay(hipStream_t stream, const TreeDev& t, const DeviceTables* dTables, const double* dOrigins,
                          const double* dDirs, const double* dTMax, size_t n, uint8_t* dHit, double* dT);
hipError_t launchSlicePoints(hipStream_t stream, double c, float minX, float minY, float step, uint32_t nSamples,
                             double* dXyz);
hipError_t launchFieldEval(hipStream_t stream, const FieldDev& f, const DeviceTables* dTables, const double* dXyz,
                           size_t n, double* dOut);
// mesh fields: F at every sample of nTasks fits of one degree -> dSamples[FitTask::sampleOff + sample]
hipError_t launchMeshSample(hipStream_t stream, const FitTask* dTasks, uint32_t nTasks, int degree, const DeviceTables* dTables,
                            const FieldDev& field, const RootMap& rm, double* dSamples);
hipError_t launchPack(hipStream_t stream, const PackItem* dItems, uint32_t nItems, const double* dArena, double* dOut);

}  // namespace hpsdf
