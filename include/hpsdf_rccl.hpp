// RCCL adaptor for hpsdf_create_distributed (include/hpsdf.h): the in-place all-gather the sharded build asks for is one
// ncclAllGather over xGMI.  Header-only on purpose -- libhpsdf.so does not link RCCL; a program that shards a build does.
//
//   hpsdf_rccl::Comm comm{ncclComm, rank};           // one per process / thread, from ncclCommInitRank / ncclCommInitAll
//   SDF::Octree tree;  tree.SetDevice(device);  tree.SetRanks(rank, world, hpsdf_rccl::AllGather, &comm);
//   tree.Create(config, field);                      // every rank; every rank ends with the identical tree
// or, on the C ABI:
//   hpsdf_create_distributed(ctx, &cfg, field, K, rank, world, hpsdf_rccl::AllGather, &comm, &block, &size, &stats);
//
// Exchange points of a build: per round the 9 errors of every job (K * 72 bytes per rank, padded to equal
// parts), at the end the ranks' packed coefficients.  xGMI is point-to-point; both messages are small (latency-bound), so
// the default ring/tree choice of RCCL is left alone.
#pragma once
#include <rccl/rccl.h>

#include "hpsdf.h"

namespace hpsdf_rccl {

struct Comm {
    ncclComm_t comm;
    int rank;
    ncclResult_t last = ncclSuccess;  // the last RCCL status, for diagnostics when AllGather returns non-zero
};

/// hpsdf_allgather_fn: rank r's part sits at d_buf + r * bytes_per_rank; in place, asynchronous on `stream`.
inline int AllGather(void* user, void* d_buf, size_t bytes_per_rank, void* stream) {
    Comm* c = static_cast<Comm*>(user);
    c->last = ncclAllGather(static_cast<const char*>(d_buf) + (size_t)c->rank * bytes_per_rank, d_buf, bytes_per_rank, ncclChar, c->comm,
                            static_cast<hipStream_t>(stream));
    return c->last == ncclSuccess ? 0 : 1;
}

}  // namespace hpsdf_rccl
