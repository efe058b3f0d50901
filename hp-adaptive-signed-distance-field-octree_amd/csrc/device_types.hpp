// Structures shared between the host runtime and the gfx950 kernels.
#pragma once
#include <hip/hip_vector_types.h>

#include <cstdint>

#include "../../include/hpsdf.h"

namespace hpsdf {

// Constant tables as the kernels read them (one copy in HBM per context; the
// kernels stage what they need into LDS).
struct DeviceTables {
    double roots[2080];
    double weights[2080];
    double nl[13][11];   // NormalisedLengths[degree][depth]
    double rec[13][2];   // Legendre recurrence constants
    uint8_t bidx[456][4];  // basis index (i, j, k) and i+j+k per coefficient row
    uint32_t count[16];    // coefficient count per degree (count[6] == 83)
};

// 8-byte node record of the query-side tree mirror.  The 56-byte serialised
// node (Include/HP/Node.h:10-33) is not needed on the device: cell boxes are
// exact dyadics recomputed during the descent, the depth is the descent count.
//   interior: a = index of first child, b = 0xFFFFFFFF
//   leaf:     a = offset of its coefficients in the device mirror (doubles; a multiple of 16: every leaf's
//             block starts on a 128-byte line), b = degree
struct NodeRec {
    uint32_t a, b;
};
constexpr uint32_t kInteriorTag = 0xFFFFFFFFu;

// Entry of the dense top-level table: one 128-byte line.  a/b as in NodeRec; a leaf of degree <= 2 also
// carries its coefficients inline (c[0..9]); larger leaves and interior nodes go through nodes/coeffs.
struct alignas(128) TopEntry {
    uint32_t a, b;
    uint32_t pad[2];
    double c[14];
};
static_assert(sizeof(TopEntry) == 128, "one line per entry");

struct TreeDev {
    const NodeRec* nodes;
    const TopEntry* top;     // dense table of the nodes at depth topDepth, indexed by cell: x + side*(y + side*z)
    const NodeRec* topRec;   // the same table, records only (8 bytes per cell)
    const double* coeffs;    // per-leaf blocks, each padded to whole 128-byte lines
    int32_t topDepth;        // every node above this depth is interior (1..5)
    int32_t maxDegree;
    int32_t leftAssoc;       // hpsdf_set_reduction_order() at the time of the launch (the gradient's normalize())
    int32_t pad;
    double nlTop[3];         // NormalisedLengths[j][topDepth], j <= 2 (degrees of the inline leaves)
    double rootCentre[3];    // Octree.cpp:322 (f32 centre widened)
    double rootInvSizes[3];  // Octree.cpp:323 (f32 reciprocal widened)
};

enum FieldKind : int32_t { kFieldAnalytic = 0, kFieldSamples = 1, kFieldMesh = 2 };

// Binary BVH node, one 64-byte line: the boxes of BOTH children (so a visit decides about both subtrees from
// one load, and a leaf child is tested against its triangle's box before the triangle itself) and the child
// references: >= 0 an inner node, < 0 a leaf: ~c = first slot << kMeshLeafShift | (count - 1), i.e. 1..kMeshLeafMax
// consecutive entries of MeshDev::triPre (the host build makes one-triangle leaves with slot = triangle; the device build
// sorts the triangles along a Morton curve and closes every subtree of <= leafTris triangles into one leaf).
constexpr uint32_t kMeshLeafShift = 4, kMeshLeafMax = 1u << kMeshLeafShift;
constexpr uint64_t kMeshMaxTris = (1ull << (31 - kMeshLeafShift)) - 1;
struct alignas(64) BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t c0, c1;
    uint32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "one line per node");

// What a BVH child's triangles are known to lie in besides its box: the slab |n . (x - g)| <= e cut by the ball |x - g| <= rho
// (mesh_build.hip, mb_slab_kernel).  A box says nothing about WHERE in it the surface runs: for a sample at distance D from a
// surface patch of size H that is tilted against the axes, the patch's box reaches ~H/2 towards the sample, so every patch
// within ~sqrt(D H) of the foot point passes the box test; the slab of a smooth patch is thin (e ~ H^2 / 8R) and leaves the
// ones within ~rho.  n = 0 turns the bound into the ball's; e < 0 marks "no slab" (the walk skips the test; g and rho still hold a ball).
struct alignas(64) NodeSlab {
    float4 g0, n0;  // child 0: g.xyz rho | n.xyz e
    float4 g1, n1;  // child 1
};
static_assert(sizeof(NodeSlab) == 64, "one line per node");

struct MeshDev {
    const float* verts;        // xyz per vertex
    const uint32_t* tris;      // 3 vertex ids per triangle
    const float4* triPos;      // 3 x float4 per triangle: its 9 vertex coordinates and its normal, gathered: a closest-point test is ONE
                               // fetch instead of a chain of index -> vertex fetches (same values, same arithmetic)
    const float4* triPre;      // 3 x float4 per leaf SLOT: the centre g and the half-extents of the triangle's bounding rectangle in its
                               // own plane, the unit normal, a unit vector along the longest edge, and the triangle's index (bits
                               // in .w of the third): a lower bound of the distance in ~24 instructions, and the slot -> triangle map
    const uint32_t* halfEdges; // twin half-edge per half-edge (Mesh.h:74)
    const BvhNode* bvh;        // node 0 is the root
    const NodeSlab* slabs;     // per BVH node, for each child: its triangles lie within e of the plane through g across the unit
                               // vector n and within rho of g (NodeSlab above); nullptr: boxes only (host-built trees)
    uint32_t nTris, nNodes;
    uint32_t leafLog2;         // every leaf holds at most 1 << leafLog2 triangles (how the sampler cuts its lower-bound batches)
    uint32_t poolCap;          // (lane, node) pairs the sampler's pool may hold, <= kMeshPoolCap (tests lower it to reach the overflow path)
    float faceTolOfSlack;      // closestSimplex's face-case tolerance as a fraction of the traversal's slack: 0.25, or +inf under
                               // hpsdf_set_mesh_face_rule(1) -- the reference's point whatever its weights (Utility.cpp:5-97)
    // traversal statistics (diagnostic builds, -DHPSDF_MESH_STATS_BUILD, and the field created under HPSDF_MESH_STATS=1):
    // [0] wave-wide queries, [1] nodes visited by them, [2] triangle tests issued (wave level), [3] lanes that ran one
    unsigned long long* stats;
};

// What the fit kernel evaluates at a sample point (world coordinates).
struct FieldDev {
    int32_t kind;      // FieldKind of the innermost field
    int32_t nPrims;
    int32_t csgOp;     // -1: none; else HPSDF_OP_* combining oldTree.Query with the inner field
    int32_t leftAssoc;  // hpsdf_set_reduction_order(): Eigen's 3-vector reductions as (a . b) . c instead of a . (b . c)
    hpsdf_prim prims[HPSDF_MAX_PRIMS];
    const double* samples;  // kFieldSamples: F values, indexed by FitTask::sampleOff + sample number
    MeshDev mesh;
    TreeDev oldTree;
};

// unit cube -> world: pt * rootBounds + rootCentre (Octree.cpp:324-328)
struct RootMap {
    double bounds[3];
    double centre[3];
};

// One cell fit (one call of Octree::FitPolynomial).
struct FitTask {
    float bmin[3], bmax[3];  // cell box in unit-cube coordinates (exact dyadics)
    uint64_t outOff;         // arena offset (doubles) where row FitBlock::rowStart of this fit goes
                             // (row 0 when FitBlock::weighted: the fit then owns a full coefficient array)
    uint64_t copyOff;        // weighted incremental fit: arena offset of the cell's previous full array
    uint64_t sampleOff;      // kFieldSamples: offset of this fit's first sample value
    uint32_t errSlot;        // where the returned error goes
    uint8_t depth;
    uint8_t pad[3];
};

// One workgroup of the fit kernel: nTasks fits of identical shape.
struct FitBlock {
    uint32_t firstTask;
    uint16_t nTasks;
    uint8_t degree;  // target degree -> (4*degree+1)^3 samples
    uint8_t planesPerChunk;  // i-planes of samples staged in LDS at a time
    uint16_t rowStart, rowEnd;  // coefficient rows computed: [rowStart,rowEnd)
    uint8_t depth;              // depth of every cell of the workgroup
    uint8_t weighted;           // also produce |mean FApprox| over 100 sample points (nearness weighting)
    uint8_t split;              // a from-scratch fit cut in two (HPSDF_FIT_SPLIT, the default mode, from degree 6 on: hpsdf_ctx_set_split_min_degree): this block is its rows of TOP degree
                                // [rowStart, rowEnd), fitted bit-exactly -- they alone enter the error, Octree.cpp:1062-1069 --; the
                                // task's outOff addresses row 0 of the whole array, the field values are written back to the sample
                                // buffer, and fit_mfma_low_kernel contracts rows [0, rowStart) from them on the matrix cores
    uint8_t pad1[1];
};

struct PackItem {  // gather of one leaf's coefficients into the packed store
    uint64_t src, dst;
    uint32_t count, pad;
};

}  // namespace hpsdf
