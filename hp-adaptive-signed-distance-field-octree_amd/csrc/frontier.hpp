// Octree::Create with the frontier on the device (frontier.hip): what hpsdf_create runs for fields the GPU evaluates
// itself (analytic primitives, meshes, tree-CSG of those) without nearness weighting and K <= 4096.  Everything else --
// host callbacks, weighted builds, the stepwise hpsdf_build_* API -- runs the host scheduler of builder.cpp.
#pragma once
#include <cstddef>
#include <cstdint>

#include "runtime.hpp"

namespace hpsdf {

bool frontierEligible(const hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, uint64_t K);
// the whole build up to the serialised block (malloc'd); the continuity post-process is the caller's
// rank / world / gather: this rank's part of a build sharded over `world` ranks (hpsdf_create_distributed)
int frontierCreate(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, uint64_t K, void** block, size_t* size,
                   hpsdf_build_stats* stats, int rank = 0, int world = 1, hpsdf_allgather_fn gather = nullptr, void* gatherUser = nullptr);

}  // namespace hpsdf
