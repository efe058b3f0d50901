// Host side of Octree::Create: tree bookkeeping and the canonical round schedule.
//
// The reference drives its fits from a polling thread pool whose result depends
// on timing (Octree.cpp:194-309, 558-659; two identical runs give different
// trees).  Here the same bookkeeping runs in deterministic rounds:
//
//   round:  stop if total < target or heap empty                     (:216)
//           round 0 pops every (coarse) entry, later rounds the top K by
//           (error desc, node index asc); jobs are evaluated as pure
//           functions (on the GPU, sharded by cost over the ranks);
//           results are applied in node-index order with the arithmetic
//           of :253-290.
//
// Coefficients never come back to the host during the build: a node owns a
// chain of segments (rows [rowStart,rowEnd) living in the arena of the rank
// that computed them).  A P-refinement appends a segment, so nothing is copied
// when the degree rises; ReallocCoeffs (:474-555) becomes one gather at the end.
#pragma once
#include <cstdint>
#include <vector>

#include "runtime.hpp"

struct hpsdf_build {
    // ---- configuration
    hpsdf_config cfg{};
    uint64_t K = HPSDF_DEFAULT_JOBS_PER_ROUND;
    int rank = 0, world = 1;
    // nearness weighting (Octree.cpp:1071-1092): every fit also evaluates its own polynomial, so each node
    // keeps ONE full coefficient array (an incremental fit copies the old rows) instead of a segment chain
    bool weighted = false;

    // ---- tree state (identical on every rank)
    struct Seg {
        uint64_t off;  // offset in the owner's store (valid on the owner only)
        uint32_t rowStart, rowEnd;
        int32_t owner;
        int32_t hostStore;  // 1: off indexes hostStore (injected), 0: device arena
        int32_t local = 0;  // a copy of another rank's rows lives HERE at off / hostStore (weighted builds on N ranks)
        int64_t next;       // next segment of the same node, -1 = end
    };
    std::vector<hpsdf_node> nodes;
    std::vector<int64_t> segHead, segTail;  // per node, -1 = none
    std::vector<Seg> segs;
    struct HeapEnt {
        uint64_t idx;
        double err;
    };
    std::vector<HeapEnt> heap;
    double total = 0.0;
    hpsdf_build_stats stats{};
    bool finished = false, laidOut = false;

    // ---- current round
    std::vector<HeapEnt> batch;  // sorted by node index
    struct Slice {
        uint64_t first, count;
    };
    std::vector<Slice> slices;  // per rank
    struct JobOut {             // where this rank put job results (my slice only)
        uint64_t pOff = ~0ull, hOff = ~0ull;
        int32_t pHost = 0, hHost = 0;
    };
    std::vector<JobOut> jobOut;  // indexed by job - slices[rank].first
    bool roundOpen = false, computed = false;
    // weighted builds on N ranks: the arrays the round just applied has accepted, in job order (children 0..7 within
    // a job) -- what the ranks hand each other after every round, because the next incremental fit of a node copies
    // its previous rows (Octree.cpp:847) and may run on any rank
    struct RowItem {
        int64_t seg;
        int32_t owner;
        uint32_t count;
    };
    std::vector<RowItem> rowItems;

    // ---- device state of this rank: buffers borrowed from the context's workspace (or private ones
    //      when several builds share a context); the context must outlive the build
    hpsdf::Workspace* ws = nullptr;
    bool ownsWs = false;
    uint64_t arenaUsed = 0;
    uint64_t measuredLimit = 0;  // hpsdf_ctx_set_build_limits' default bound on bytes, measured at most once per build (checkBuildLimits)
    std::vector<double> hostStore;  // injected coefficients

    // ---- layout (after the last round)
    struct LeafSeg {
        uint64_t src, dst;
        uint32_t count;
        int32_t owner, hostStore;
    };
    std::vector<LeafSeg> layout;  // DFS leaf order, segments in row order
    std::vector<uint64_t> packCounts;
    uint64_t nCoeffsTotal = 0;

    ~hpsdf_build();
};

namespace hpsdf {

uint64_t jobCost(int degree, int depth, bool coarse);

int builderBegin(hpsdf_build* b, const hpsdf_config* cfg, const hpsdf_build_opts* opts);
int builderSelect(hpsdf_build* b, uint64_t* nJobs);
int builderJobs(const hpsdf_build* b, hpsdf_job* out);
int builderCompute(hpsdf_build* b, hpsdf_ctx* ctx, const hpsdf_field* field);
int builderInject(hpsdf_build* b, uint64_t job, const double* pCoeffs, const double* hCoeffs);
int builderApply(hpsdf_build* b, const double* headers);
int builderLayout(hpsdf_build* b);
int builderPackDevice(hpsdf_build* b, hpsdf_ctx* ctx, double** dPack, uint64_t* n);
int builderPackHost(hpsdf_build* b, hpsdf_ctx* ctx, double* out);
int builderAssemble(hpsdf_build* b, const double* const* packs, void** block, size_t* size);
int builderRowsCounts(const hpsdf_build* b, uint64_t* counts);
int builderRowsPackHost(hpsdf_build* b, hpsdf_ctx* ctx, double* out);
int builderRowsUnpackHost(hpsdf_build* b, hpsdf_ctx* ctx, const double* const* parts);
int builderNodeRowsHost(hpsdf_build* b, hpsdf_ctx* ctx, uint64_t node, double* out, uint64_t* n);

}  // namespace hpsdf
