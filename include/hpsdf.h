/*
 * hpsdf.h -- C ABI of the MI355X-native hp-adaptive SDF octree hot path.
 *
 * The reference (jw007123/hp-Adaptive-Signed-Distance-Field-Octree) has no FFI
 * or plugin layer: its hot path sits behind the C++ class SDF::Octree
 * (Include/HP/Octree.h:37-86).  This header is the boundary a maintainer binds
 * to replace the bodies of that class's Create / Query / ToMemoryBlock /
 * FromMemoryBlock with gfx950 kernels (the C++ drop-in that
 * does exactly this is include/hpsdf_octree.hpp).
 *
 * Conventions: plain pointers and sizes, POD structs, caller-allocated outputs,
 * no exceptions cross the boundary, every entry point returns an hpsdf_status
 * (0 = ok) and hpsdf_last_error() gives the thread-local message.  Pointers
 * named d_* are device (HBM) pointers on the context's GPU; everything else is
 * host memory.  "stream" is a hipStream_t passed as void*.
 *
 * All file:line citations are relative to the reference checkout.
 */
#ifndef HPSDF_H
#define HPSDF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define HPSDF_API __attribute__((visibility("default")))
#else
#define HPSDF_API
#endif

typedef enum hpsdf_status {
    HPSDF_OK = 0,
    HPSDF_ERR_INVALID_ARGUMENT = 1,
    HPSDF_ERR_NO_DEVICE = 2,      /* no usable gfx950 device / HIP runtime error */
    HPSDF_ERR_HIP = 3,
    HPSDF_ERR_BAD_BLOCK = 4,      /* malformed MemoryBlock */
    HPSDF_ERR_UNSUPPORTED = 5,
    HPSDF_ERR_STATE = 6,          /* call sequence violated */
    HPSDF_ERR_OUT_OF_MEMORY = 7,
    HPSDF_ERR_OPEN_MESH = 8,      /* Mesh::CreateHalfEdges would return false (Mesh.cpp:121-128) */
    HPSDF_ERR_BUILD_LIMIT = 9     /* Create stopped: the tree outgrew hpsdf_ctx_set_build_limits (ABI 4) */
} hpsdf_status;

/* ---- ABI version ------------------------------------------------------------
 * Bumped whenever a struct of this header changes size or an entry point changes meaning; additions of entry points alone do not
 * bump it.  hpsdf_abi_version() is what the loaded library was built with: a binding compares it with the HPSDF_ABI_VERSION it was
 * compiled against before it hands the library a struct (include/hpsdf_octree.hpp does, and throws on a mismatch).
 *   3 (round 5): hpsdf_build_stats grew from 88 to 112 bytes (fit_mode, split_fits, device_frontier) -- hpsdf_create,
 *                hpsdf_create_distributed and hpsdf_build_get_stats write all 112; hpsdf_ctx_set_fast_fit(ctx, 0) selects
 *                HPSDF_FIT_EXACT (it left HPSDF_FIT_SPLIT on in ABI 2).
 *   4 (round 6): HPSDF_ERR_BUILD_LIMIT; no struct changed.  hpsdf_build_stats is FROZEN at 112 bytes from here on: anything new a
 *                build has to report comes through a getter of its own (hpsdf_ctx_get_build_limits, ...). */
#define HPSDF_ABI_VERSION 4
HPSDF_API int hpsdf_abi_version(void);

/* ---- constants: Include/HP/Consts.h:7-8, Include/HP/Octree.h:89 ----------- */
#define HPSDF_BASIS_MAX_DEGREE 12
#define HPSDF_TREE_MAX_DEPTH 10
#define HPSDF_INTERIOR_DEGREE 13
#define HPSDF_INITIAL_NODE_ERR 100.0
#define HPSDF_DEFAULT_JOBS_PER_ROUND 1024

/* ---- serialised PODs -------------------------------------------------------
 * Byte-identical to the reference structs on Linux LP64, where the reference's
 * "u32" is an 8-byte unsigned long (Include/Utility/Literals.h:9). */

/* SDF::Node, Include/HP/Node.h:10-33 (56 bytes) */
typedef struct hpsdf_node {
    uint64_t child_idx;    /* @0  first of 8 children; all-ones = leaf */
    float aabb_min[3];     /* @8  Eigen::AlignedBox3f min */
    float aabb_max[3];     /* @20 Eigen::AlignedBox3f max */
    uint64_t coeffs_start; /* @32 offset into the coefficient store (doubles) */
    uint8_t degree;        /* @40 13 = interior */
    uint8_t pad0[7];
    uint8_t depth;         /* @48 */
    uint8_t pad1[7];
} hpsdf_node;

/* SDF::Config, Include/HP/Config.h:12-43 (80 bytes) */
typedef struct hpsdf_config {
    uint8_t weighting_type; /* @0  0 None, 1 Polynomial, 2 Exponential */
    uint8_t pad0[7];
    double weighting_strength;  /* @8  */
    uint8_t continuity_enforce; /* @16 */
    uint8_t pad1[7];
    double continuity_strength; /* @24 */
    uint8_t enable_logging;     /* @32 */
    uint8_t pad2[7];
    double target_error_threshold; /* @40 */
    uint64_t thread_count;         /* @48 */
    float root_min[3];             /* @56 */
    float root_max[3];             /* @68 */
} hpsdf_config;

/* Source/HP/Config.cpp:5-14 (thread_count = hardware concurrency) */
HPSDF_API int hpsdf_config_default(hpsdf_config* out);

HPSDF_API const char* hpsdf_last_error(void);
HPSDF_API const char* hpsdf_version(void);

/* ---- constant tables (Include/HP/Utility.h:40-160, Include/HP/Legendre.h) --
 * Copies the host tables the kernels are fed with.  Any pointer may be NULL.
 * roots/weights: 2080 (rule n at n(n-1)/2); normalised_lengths: 13*11;
 * recurrence: 13*2; coeff_count: 13; basis_index: 455*3; sum_to_n: 50. */
HPSDF_API int hpsdf_tables_get(double* roots, double* weights, double* normalised_lengths, double* recurrence,
                               uint64_t* coeff_count, uint64_t* basis_index, uint64_t* sum_to_n);

/* ---- device context --------------------------------------------------------- */
typedef struct hpsdf_ctx hpsdf_ctx;
/* device: HIP ordinal; stream: hipStream_t to launch on (NULL = a stream owned by the context).
 * A context owns the build's workspace (coefficient arena, per-round staging, the continuity post-process's matrix,
 * vectors and worker threads -- all kept between calls): hpsdf_create, the hpsdf_build_* round calls and
 * hpsdf_continuity_post_process_device use one context from one thread at a time.  The *_host query entry points
 * serialise themselves per context (Octree::Query* is const and callable from many threads in the reference);
 * use one context per thread for concurrent builds. */
HPSDF_API int hpsdf_ctx_create(int device, void* stream, hpsdf_ctx** out);
HPSDF_API int hpsdf_ctx_destroy(hpsdf_ctx* ctx);
HPSDF_API int hpsdf_ctx_set_stream(hpsdf_ctx* ctx, void* stream);
HPSDF_API int hpsdf_ctx_synchronize(hpsdf_ctx* ctx);
/* How cell fits (Octree::FitPolynomial, Octree.cpp:1007-1093) may leave the reference's term-by-term summation where that changes no
 * decision (csrc/fit_low.hip, csrc/fit_mfma.hip).  A fit's returned error is the sum of squares of its rows of TOP total degree alone (:1062-1069), and the
 * errors are all that selection, the P/H decision (:600-601) and the stop rule (:216) ever read.
 *   HPSDF_FIT_SPLIT (default): a from-scratch fit of degree p >= 6 (hpsdf_ctx_set_split_min_degree) is cut in two -- the rows of total degree p by the term-by-term
 *     kernel that reproduces the reference's summation order bit for bit, the rows below p from the same samples by sum factorisation
 *     (three one-axis contractions with fused multiply-adds: 5-13 x fewer operations than the direct contraction; csrc/fit_low.hip.
 *     HPSDF_LOW_KERNEL=mfma: the direct contraction as a GEMM on the matrix cores instead, v_mfma_f64_16x16x4_f64, degrees >= 4).  Incremental fits (all their rows are top-degree rows) stay exact.  Errors, topology, node array: identical
 *     to HPSDF_FIT_EXACT by construction, no guard band; coefficients of rows below the top degree of leaves that were created by
 *     a split at degree >= 6 agree with it to ~1e-17 absolute (not bit for bit).  Trees whose leaves never exceed degree 5 --
 *     the BASELINE configs stop at 3 -- are byte-identical in both modes.  Weighted builds (the weight reads every row, :1209-1247) and
 *     mesh fits that sample inside the fit kernel keep the exact kernel.
 *   HPSDF_FIT_EXACT: every row by the bit-exact kernel: the canonical bytes (what the CPU oracle produces).
 *   HPSDF_FIT_FAST: every row of every fit of degree >= 4 on the matrix cores.  Errors then agree to ~1e-15 relative only, and
 *     refinement decisions that are exact ties in the reference's arithmetic may fall the other way.
 * hpsdf_ctx_set_fast_fit(ctx, on) = set_fit_mode(on ? HPSDF_FIT_FAST : HPSDF_FIT_EXACT): switching the fast fit off gives the canonical
 * bytes, as it did before the split mode existed (round 5; round 4 sent it to HPSDF_FIT_SPLIT).  The environment variable
 * HPSDF_FIT_MODE=exact|split|fast sets the mode a new context starts with.  hpsdf_build_stats::fit_mode / split_fits say which bytes a
 * build returned: split_fits == 0 means the block is the canonical one whatever the mode. */
enum { HPSDF_FIT_EXACT = 0, HPSDF_FIT_SPLIT = 1, HPSDF_FIT_FAST = 2 };
HPSDF_API int hpsdf_ctx_set_fit_mode(hpsdf_ctx* ctx, int mode);
HPSDF_API int hpsdf_ctx_get_fit_mode(hpsdf_ctx* ctx, int* mode);
/* HPSDF_FIT_SPLIT splits from-scratch fits from this degree on (2..12, 12 = never; default 6, or HPSDF_SPLIT_MIN_DEGREE when the
 * context is created): the two-kernel form pays from degree 5 (-9 %; -27 % / -36 % / -47 % at 6 / 7 / 8), below that the samples' trip
 * through memory costs more than the rows saved (csrc/fit_mfma.hip, fitSplitDefaultMinDegree).  Round 0's coarse fits are never split. */
HPSDF_API int hpsdf_ctx_set_split_min_degree(hpsdf_ctx* ctx, int degree);
HPSDF_API int hpsdf_ctx_set_fast_fit(hpsdf_ctx* ctx, int on);
/* Which way Eigen's 3-vector reductions associate is a property of the reference's BUILD, not of its source: a . (b . c) when Eigen
 * does not vectorise the reduction (EIGEN_DONT_VECTORIZE, non-SSE targets; the default here and in the oracle), (a . b) . c when it
 * reduces a Vector3d through a Packet2d first (what an SSE2 build of Eigen 3.4 appears to do).  Four statements of the path depend on
 * it: aabbScale.prod() (Source/HP/Octree.cpp:1022), unitWeights.prod() (:1040), grad.normalize() (:970) and Vector3d::norm() in the
 * analytic test fields (Source/Tests/HPUnitTests.cpp:48-51; the HPSDF_PRIM_* primitives here).  left_assoc != 0 switches all of them,
 * in every kernel and in the host-answered scalar calls, process-wide and for launches prepared afterwards (call it before Create /
 * Query, not during).  Under either setting the blocks are byte-identical to the oracle's under ora_set_reduction_order() of the same
 * value (tests/test_gpu_parity.py::test_reduction_order_switch_matches_the_oracle).  HPSDF_REDUCTION_ORDER=left sets it at load. */
HPSDF_API void hpsdf_set_reduction_order(int left_assoc);
HPSDF_API int hpsdf_get_reduction_order(void);
/* The same switch for ONE context (round 6): -1 = follow the process-wide setting above (the default), 0 / 1 = this context's own,
 * for every launch it prepares afterwards -- fits, field evaluation, gradient queries, the host-answered scalar calls made through it.
 * Two Octrees of one process can then differ (say, one per reference build being compared against).  _get returns the value in effect. */
HPSDF_API int hpsdf_ctx_set_reduction_order(hpsdf_ctx* ctx, int left_assoc);
HPSDF_API int hpsdf_ctx_get_reduction_order(const hpsdf_ctx* ctx, int* left_assoc);
/* Runaway builds.  The reference's loop (Octree.cpp:212-216) ends when the total error falls below the threshold or the queue is
 * empty; a threshold below what the error estimate can reach on a field (the default Config()'s 1e-10 on most fields, Config.cpp:7)
 * refines towards depth 10 and degree 12 everywhere, and the reference grows its heap until host memory ends.  Here such a build used to
 * end with HPSDF_ERR_OUT_OF_MEMORY when a device allocation finally failed -- minutes later on a 288 GB device.  Limits, checked when
 * a round opens, by both schedulers and by every rank of a sharded build:
 *   max_nodes: the tree's node count (identical on every rank, so every rank stops in the same round);
 *   max_bytes: the device memory this rank's build state needs for the round about to open (node arrays, coefficient arena, sample
 *              buffer -- capacities grow by doubling, so the allocation can reach twice this).
 * 0 = the default: no bound on nodes; bytes = 1/64 of the device memory that is free when the build first needs more than
 * 256 MiB, at least 1 GiB (measured then, once per Create; nothing is measured for builds that stay below, i.e. for every BASELINE
 * config; 4.5 GiB on an idle MI355X -- a tree of one to ten million nodes, which the runaway builds on file reach within 3-15 seconds:
 * profiles/r06_default_config.txt; the first version, 1/256, refused a mesh build at 1e-8 that was most of the way to its threshold).  The hand-over buffer of split fits (HPSDF_FIT_SPLIT) does not count: it is bounded by 2^31 samples
 * (16 GiB) whatever the tree's size.  The DEFAULT bound applies to what grows with the tree -- node arrays and coefficient arena -- and
 * leaves a mesh field's sample buffer out as well (same bound; an ordinary mesh build at 4096 jobs a round needs 1.5-3 GiB of it at
 * degrees 3-4 with a tree of 25 000 nodes); a max_bytes the caller sets bounds all of it.
 * UINT64_MAX = no limit.  A build that crosses a limit returns HPSDF_ERR_BUILD_LIMIT; hpsdf_last_error() names rounds, nodes,
 * bytes, both limits, and the total error against the threshold.  No block is returned.  On several ranks a rank that stops on its
 * bytes alone takes the others out through the FAILURES protocol of hpsdf_create_distributed. */
HPSDF_API int hpsdf_ctx_set_build_limits(hpsdf_ctx* ctx, uint64_t max_nodes, uint64_t max_bytes);
HPSDF_API int hpsdf_ctx_get_build_limits(const hpsdf_ctx* ctx, uint64_t* max_nodes, uint64_t* max_bytes);
HPSDF_API void* hpsdf_ctx_stream(hpsdf_ctx* ctx);

/* ---- fields: the callback F of Octree::Create (Include/HP/Octree.h:50) ------ */
enum { HPSDF_PRIM_SPHERE = 0, HPSDF_PRIM_BOX = 1, HPSDF_PRIM_TORUS_Y = 2, HPSDF_PRIM_PLANE = 3 };
enum { HPSDF_OP_UNION = 0, HPSDF_OP_INTERSECT = 1, HPSDF_OP_SUBTRACT = 2 };
#define HPSDF_MAX_PRIMS 16
typedef struct hpsdf_prim {
    int32_t kind; /* HPSDF_PRIM_* */
    int32_t op;   /* HPSDF_OP_*: how this primitive combines with the running value (ignored for the first) */
    double p[8];  /* sphere: c,r | box: c,half | torus(y axis): c,R,r | plane: n,offset */
} hpsdf_prim;

/* f64 F(const Vector3d& pt, u32 threadIdx): called concurrently from
 * config.thread_count host threads, thread_idx in [0, thread_count)
 * (Source/HP/BuildThreadPool.cpp:10-14). */
typedef double (*hpsdf_callback)(const double* pt, uint64_t thread_idx, void* user);

typedef struct hpsdf_field hpsdf_field;
typedef struct hpsdf_tree hpsdf_tree;

/* evaluated on the GPU inside the fit kernel */
HPSDF_API int hpsdf_field_create_analytic(const hpsdf_prim* prims, int n_prims, hpsdf_field** out);
/* evaluated by host threads per round, values shipped to HBM (keeps Create(config, std::function) working) */
HPSDF_API int hpsdf_field_create_callback(hpsdf_callback cb, void* user, hpsdf_field** out);
/* closed triangle mesh; f32 signed distance with angle-weighted pseudo-normals
 * (Source/Meshing/Mesh.cpp:54-63,162-242; Source/Meshing/Utility.cpp:5-97), evaluated on the GPU
 * over a device BVH.  tris: 3 vertex indices per triangle, CCW. */
HPSDF_API int hpsdf_field_create_mesh(hpsdf_ctx* ctx, const float* verts, uint64_t n_verts, const uint64_t* tris,
                                      uint64_t n_tris, hpsdf_field** out);
/* Meshing::ObjParser::Load (Source/Meshing/ObjParser.cpp:11-164): `v` records and triangular `f` records in
 * the spellings a, a/t, a//n, a/t/n.  *verts (3 floats per vertex) and *tris (3 zero-based indices per
 * triangle) are malloc'd; the caller frees them.  Host-only: needs no device. */
HPSDF_API int hpsdf_obj_load(const char* path, float** verts, uint64_t* n_verts, uint64_t** tris, uint64_t* n_tris);
/* F'(p) = op(old.Query(p), inner(p)): Octree::UnionSDF/SubtractSDF/IntersectSDF, Octree.cpp:355-400.
 * op: HPSDF_OP_UNION -> min(old,F); HPSDF_OP_SUBTRACT -> max(-old,F); HPSDF_OP_INTERSECT -> max(old,F). */
HPSDF_API int hpsdf_field_create_tree_csg(const hpsdf_tree* old_tree, int op, const hpsdf_field* inner,
                                          hpsdf_field** out);
HPSDF_API int hpsdf_field_destroy(hpsdf_field* f);
/* F at n world-space points (device pointers, xyz interleaved); analytic/mesh/tree fields only. */
HPSDF_API int hpsdf_field_eval_device(hpsdf_ctx* ctx, const hpsdf_field* f, const double* d_xyz, size_t n,
                                      double* d_out);
HPSDF_API int hpsdf_field_eval_host(hpsdf_ctx* ctx, const hpsdf_field* f, const double* xyz, size_t n, double* out);
/* Calls of one or two points on a plain mesh field (Mesh::SignedDistanceAtPt(pt, bvh) from a user's SDF lambda) are answered on the
 * calling thread from host copies of the field's device-built arrays.  The first such call makes them: one device-wide
 * synchronisation and a download of the whole mesh (vertices, triangles, records, BVH: ~160 bytes a triangle), kept until the field is
 * destroyed or this call drops them.  Meshes whose copies would exceed HPSDF_HOST_MESH_MIRROR_MB (default 512) are never mirrored:
 * their small calls are launches. */
HPSDF_API int hpsdf_field_release_host_copies(hpsdf_field* f);
/* Mesh fields, the one stated deviation from the reference's closest-point arithmetic (Source/Meshing/Utility.cpp:5-97): its face
 * case returns q = u a + v b + w c whatever the barycentric weights are, and beside the short edges of needle-shaped triangles
 * (its 1e-6 guards are absolute) a weight can be negative enough to put q well outside the triangle -- a distance below the
 * triangle's own, which a search reports or not depending on what it has pruned (the reference's BVH and its linear scan
 * disagree there too).  Here a face-case point farther outside its triangle than 5e-7 of the mesh's scale (a quarter of the
 * traversal's slack) is not taken by a search it could win: the closest point of the triangle's boundary (the nearest of its three
 * edges' closest points, f32) takes its place.  Consequences: the four evaluation paths
 * below (naive scan, per-lane traversal, shared traversal, hpsdf_field_eval_*) agree BIT FOR BIT on every mesh; against the
 * reference's arithmetic they differ exactly at the points where its own value is such an artefact (6 points in 8 000 random
 * meshes x 5 301 points, all on meshes squashed 100 : 1 or more; tests/test_gpu_configs.py::test_needle_meshes_one_answer_on_every_path).
 * At such a point the value is the distance to the triangle's boundary: at least the true distance, and above it by a fraction of the
 * needle's width when the true closest point lies inside the needle (largest seen in the sweeps: 2.9e-4 of the mesh's extent, where the
 * reference's value was 4e-3 of the extent BELOW the distance; tools/fuzz_mesh_bvh.py, seed 910968). */
/* ... and the way back to the reference's values: hpsdf_set_mesh_face_rule(1) (process-wide, for launches prepared afterwards; the
 * environment variable HPSDF_MESH_FACE_RULE=reference sets it from the start) makes the closest-point routine return Utility.cpp:5-97's
 * face-case point unconditionally.  The O(n) scan (hpsdf_field_eval_naive_host) then equals the reference's Mesh::SignedDistanceAtPt(pt),
 * Mesh.cpp:134-159, operation by operation on every mesh, needles included; hpsdf_field_eval_* and mesh builds use the per-point
 * traversal, which prunes by boxes only, as the reference's BVH does -- and, like it, may report such an artefact or not depending
 * on what it pruned.  The shared traversal (hpsdf_field_eval_wave_host, and the sampler of the device-side frontier) bounds distances
 * to TRIANGLES, which an artefact can undercut: it returns HPSDF_ERR_UNSUPPORTED under this rule, and Create with a mesh field takes
 * the host scheduler with the per-point traversal inside the fit (what HPSDF_MESH_FUSED=1 selects). */
HPSDF_API void hpsdf_set_mesh_face_rule(int reference);
HPSDF_API int hpsdf_get_mesh_face_rule(void);
/* ... per context (round 6): -1 = follow the process-wide rule (default), 0 / 1 = this context's own, for the launches it prepares. */
HPSDF_API int hpsdf_ctx_set_mesh_face_rule(hpsdf_ctx* ctx, int reference);
HPSDF_API int hpsdf_ctx_get_mesh_face_rule(const hpsdf_ctx* ctx, int* reference);
/* (Mesh fields: a point with a coordinate that is not a finite number has no closest triangle -- the reference's search ends
 * with bestTri = -1 there and reads out of bounds, Mesh.cpp:139,157 -- and evaluates to a NaN on every entry point below.) */
/* Mesh::SignedDistanceAtPt(pt) without a BVH (Source/Meshing/Mesh.cpp:42-51 over the O(n) scan :134-159), mesh fields
 * only: every triangle is tested for every point (one wave per point).  What the reference's TestBVHQuerying
 * (Source/Tests/MeshingUnitTests.cpp:110-138) compares the BVH answer with; same tie rule, so the two agree bit for bit. */
HPSDF_API int hpsdf_field_eval_naive_host(hpsdf_ctx* ctx, const hpsdf_field* f, const double* xyz, size_t n, double* out);
/* Mesh::SignedDistanceAtPt(pt, bvh, threadIdx) (Source/Meshing/Mesh.cpp:54-84) through the traversal Create's sampler
 * uses: 64 points share one walk of the BVH.  Same values as hpsdf_field_eval_host bit for bit, whatever the order of the
 * points: sets of 4096 points or more are visited along a Morton curve (an index sort on the device), so that a wave's
 * points are neighbours in space even when the array's are not.  Mesh fields only. */
HPSDF_API int hpsdf_field_eval_wave_host(hpsdf_ctx* ctx, const hpsdf_field* f, const double* xyz, size_t n, double* out);
/* The same through the per-point stack traversal (what a mesh field under a tree-CSG wrapper and the fused mesh fit run;
 * hpsdf_field_eval_* itself takes the faster shared traversal for plain mesh fields): same bits.  Diagnostics. */
HPSDF_API int hpsdf_field_eval_lane_host(hpsdf_ctx* ctx, const hpsdf_field* f, const double* xyz, size_t n, double* out);
/* Diagnostics: the device's acosf -- the angle weights of a vertex pseudo-normal, std::acos in Source/Meshing/Mesh.cpp:226-231
 * -- for the n floats whose bit patterns are first_bits, first_bits + stride, ...  It is glibc's float acos as shipped up to glibc
 * 2.40 (the fdlibm e_acosf algorithm, restated in csrc/acosf_host_libm.hpp): out[] equals acosf() of such a host bit for bit.
 * glibc >= 2.41 (correctly rounded CORE-MATH acosf), musl and other libms may differ from it in the last place; the tests
 * detect that (tests/test_product_cpu.py compares with the machine's libm and says which one it found). */
HPSDF_API int hpsdf_selftest_acosf(hpsdf_ctx* ctx, uint32_t first_bits, uint32_t stride, size_t n, float* out);
/* Diagnostics (no reference counterpart; HPSDF_ERR_UNSUPPORTED unless the library was built with
 * -DHPSDF_MESH_STATS_BUILD): BVH traversal counters of a mesh field created while the environment
 * variable HPSDF_MESH_STATS was set -- out[0] wave-wide closest-triangle queries (64 points each), [1] BVH nodes they visited,
 * [2] (point, triangle) pairs that went through the lower-bound test, [3] pairs that went on to the closest-point test,
 * [4] (point, leaf) pairs queued, [5] wave-wide batches of lower-bound tests, [6] of closest-point tests, [7] points whose
 * seed (the leaf their own descent ended in) already held the answer.  Synchronises the device; reset != 0 zeroes the counters. */
HPSDF_API int hpsdf_field_mesh_stats(const hpsdf_field* f, uint64_t out[8], int reset);

/* ---- Query: Octree::FromMemoryBlock + Octree::Query (Octree.cpp:403-421, 662-702, 859-901) */
/* block layout: [u64 nCoeffs][f64 x nCoeffs][u64 nNodes][hpsdf_node x nNodes][hpsdf_config] */
HPSDF_API int hpsdf_tree_upload(hpsdf_ctx* ctx, const void* block, size_t size, hpsdf_tree** out);
HPSDF_API int hpsdf_tree_destroy(hpsdf_tree* t);
HPSDF_API int hpsdf_tree_info(const hpsdf_tree* t, uint64_t* n_nodes, uint64_t* n_coeffs, uint64_t* n_leaves,
                              int* max_degree, int* max_depth);
/* out[i] = Query(xyz[3i..3i+2]); DBL_MAX outside the root (Octree.cpp:668-671).
 * Asynchronous on the context stream; no host synchronisation. */
HPSDF_API int hpsdf_query_device(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* d_xyz, size_t n, double* d_out);
/* host buffers: H2D + kernel + D2H, synchronous (PCIe-inclusive) */
HPSDF_API int hpsdf_query_host(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* xyz, size_t n, double* out);

/* Octree::QueryWithGradient (Octree.cpp:749-789, 904-985): out[i] as Query; grad[3i..] = the reference's
 * normalised central-difference "gradient".  Rows of points outside the root are left untouched. */
HPSDF_API int hpsdf_query_gradient_device(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* d_xyz, size_t n,
                                          double* d_out, double* d_grad);
HPSDF_API int hpsdf_query_gradient_host(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* xyz, size_t n, double* out,
                                        double* grad);

/* Octree::QueryRay (Octree.cpp:705-746; Ray: Include/HP/Ray.h, Source/HP/Ray.cpp:5-68) for n rays: sphere
 * tracing, <= 200 Query steps each.  hit[i] = 1/0; t[i] is written only on a hit (the reference leaves t_
 * untouched otherwise) and receives what the reference stores there -- the field value at the stopping point
 * (Octree.cpp:730).  origins/dirs: xyz interleaved, world coordinates; dirs are used as given.  A _host call of up to 32 rays -- the
 * scalar QueryRay(ray, tMax, t) -- is stepped on the calling thread like the scalar Query (csrc/host_query.cpp): same bits, no launch. */
HPSDF_API int hpsdf_query_ray_device(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* d_origins,
                                     const double* d_dirs, const double* d_tmax, size_t n, uint8_t* d_hit,
                                     double* d_t);
HPSDF_API int hpsdf_query_ray_host(hpsdf_ctx* ctx, const hpsdf_tree* t, const double* origins, const double* dirs,
                                   const double* tmax, size_t n, uint8_t* hit, double* t_out);

/* Octree::OutputFunctionSlice (Octree.cpp:1131-1206) up to the byte image: n_samples^2 Query() calls on the
 * plane z = c over [view_min.xy, view_min.xy + (view_max.x - view_min.x)) (the reference steps both axes by
 * the x extent), then its green/blue normalisation.  rgb: n_samples^2 * 3 bytes, row i = y index (host);
 * values: the n_samples^2 Query results (host, may be NULL).  The reference uses n_samples = 2048 and hands
 * the bytes to stb_image_write (not a dependency here; include/hpsdf_octree.hpp writes the BMP itself). */
HPSDF_API int hpsdf_function_slice(hpsdf_ctx* ctx, const hpsdf_tree* t, double c, const float* view_min,
                                   const float* view_max, uint64_t n_samples, uint8_t* rgb, double* values);

/* ---- Create: Octree::Create under the canonical round schedule ---------------
 * (Octree.cpp:312-352, 194-309, 558-659, 804-856, 1007-1093; schedule: DESIGN.md)
 *
 * One round = select jobs -> compute this rank's slice on the GPU -> (exchange
 * the 9 errors per job across ranks) -> apply.  The exchange is the caller's:
 * world == 1 needs none; N ranks all-gather the slice buffers over RCCL
 * (hp-adaptive-..._amd/distributed.py).  Every rank applies the same headers in
 * the same order, so all ranks hold identical trees. */
typedef struct hpsdf_build hpsdf_build;

typedef struct hpsdf_build_opts {
    uint64_t max_jobs_per_round; /* K of the canonical schedule; 0 = HPSDF_DEFAULT_JOBS_PER_ROUND */
    int32_t rank, world;         /* this process's shard of every round */
    int32_t reserved[2];
} hpsdf_build_opts;

typedef struct hpsdf_job { /* one popped heap entry (BuildThreadPool::Input, BuildThreadPool.h:24-28) */
    uint64_t node_idx;
    float aabb_min[3], aabb_max[3];
    double err;
    uint8_t degree, depth, coarse, pad[5];
} hpsdf_job;

#define HPSDF_JOB_HEADER_DOUBLES 9 /* p_err, h_err[0..7] */

typedef struct hpsdf_build_stats { /* 112 bytes since ABI 3, frozen (see HPSDF_ABI_VERSION) */
    uint64_t rounds, jobs, p_refines, h_refines, dropped, fits, samples;
    uint64_t n_nodes, n_leaves, n_coeffs;
    double total_error;
    uint64_t fit_mode;   /* HPSDF_FIT_* the build ran in */
    uint64_t split_fits; /* from-scratch fits whose rows below the top degree came from the sum-factorised / matrix-core kernel (HPSDF_FIT_SPLIT,
                            degree >= split_min_degree); 0: every coefficient is the bit-exact kernel's */
    uint64_t device_frontier; /* 1: selection, decision and bookkeeping ran on the device (csrc/frontier.hip); 0: the host scheduler's rounds */
} hpsdf_build_stats;

HPSDF_API int hpsdf_build_begin(const hpsdf_config* cfg, const hpsdf_build_opts* opts, hpsdf_build** out);
HPSDF_API int hpsdf_build_destroy(hpsdf_build* b);
/* pops the next batch; *n_jobs == 0 means the stop rule fired (Octree.cpp:216) */
HPSDF_API int hpsdf_build_round_select(hpsdf_build* b, uint64_t* n_jobs);
HPSDF_API int hpsdf_build_round_jobs(const hpsdf_build* b, hpsdf_job* out);
/* contiguous cost-balanced job range of `rank`; identical on every rank */
HPSDF_API int hpsdf_build_round_slice(const hpsdf_build* b, int rank, uint64_t* first_job, uint64_t* n_jobs);
/* largest slice over ranks (padding unit of the all-gather) */
HPSDF_API int hpsdf_build_round_max_slice(const hpsdf_build* b, uint64_t* n_jobs);
/* sample + fit + error kernels for this rank's slice, asynchronous on the ctx stream */
HPSDF_API int hpsdf_build_round_compute(hpsdf_build* b, hpsdf_ctx* ctx, const hpsdf_field* field);
/* this rank's [slice][9] header buffer in HBM (valid until the next round_select) */
HPSDF_API int hpsdf_build_round_results_device(hpsdf_build* b, double** d_headers, uint64_t* n_doubles);
/* D2H of the same, synchronous */
HPSDF_API int hpsdf_build_round_results_host(hpsdf_build* b, hpsdf_ctx* ctx, double* out);
/* headers of ALL jobs of the round, [n_jobs][9], host memory */
HPSDF_API int hpsdf_build_round_apply(hpsdf_build* b, const double* headers);
/* integration/test hook: supply one job's coefficients from the host instead of the GPU arena
 * (p_coeffs: ncoef(p+1) or 10 for a coarse job; h_coeffs: 8*ncoef(p); either may be NULL if unused) */
HPSDF_API int hpsdf_build_round_inject(hpsdf_build* b, uint64_t job, const double* p_coeffs,
                                       const double* h_coeffs);

/* Nearness-weighted builds on N ranks (config.weighting_type != 0, opts.world > 1).  A weighted fit keeps ONE full
 * coefficient array per node and an incremental fit copies the node's previous rows (Octree.cpp:847), so the rank
 * that fits a node next needs what another rank accepted for it: after every hpsdf_build_round_apply the ranks hand
 * each other the arrays that round accepted (a P refinement: the node's new array; an H refinement: its 8
 * children's), in job order.  hpsdf_build_rows_counts: doubles each rank contributes (all zero for unweighted or
 * single-rank builds); hpsdf_build_rows_pack_host: this rank's part, contiguous; after an all-gather,
 * hpsdf_build_rows_unpack_host(parts[world]) stores the other ranks' parts in this rank's arena (ctx != NULL) or host
 * store (ctx == NULL, CPU tests).  parts[own rank] is not read.  hpsdf_build_node_rows_host: the rows a node holds right
 * now, if this rank has them (out may be NULL to ask for the count). */
HPSDF_API int hpsdf_build_rows_counts(const hpsdf_build* b, uint64_t* counts_per_rank);
HPSDF_API int hpsdf_build_rows_pack_host(hpsdf_build* b, hpsdf_ctx* ctx, double* out);
HPSDF_API int hpsdf_build_rows_unpack_host(hpsdf_build* b, hpsdf_ctx* ctx, const double* const* parts);
HPSDF_API int hpsdf_build_node_rows_host(hpsdf_build* b, hpsdf_ctx* ctx, uint64_t node_idx, double* out, uint64_t* n_rows);

/* ReallocCoeffs (Octree.cpp:474-555): DFS layout, then gather.  counts[r] = doubles owned by rank r. */
HPSDF_API int hpsdf_build_layout(hpsdf_build* b, uint64_t* n_coeffs_total, uint64_t* counts_per_rank);
/* this rank's leaves' coefficients, in DFS order */
HPSDF_API int hpsdf_build_pack_device(hpsdf_build* b, hpsdf_ctx* ctx, double** d_pack, uint64_t* n_doubles);
HPSDF_API int hpsdf_build_pack_host(hpsdf_build* b, hpsdf_ctx* ctx, double* out);
/* packs[r] = rank r's buffer (host).  *block is malloc'd; the caller frees it with free()
 * (ToMemoryBlock ownership, Octree.cpp:445; README.md:41). */
HPSDF_API int hpsdf_build_assemble(hpsdf_build* b, const double* const* packs, void** block, size_t* size);
HPSDF_API int hpsdf_build_get_stats(const hpsdf_build* b, hpsdf_build_stats* out);

/* ---- continuity post-process: Octree::PerformContinuityPostProcess (Octree.cpp:1717-1762) ------------
 * Host side, as the reference's (which hands the system to Eigen): enumerate the face-adjacent leaf pairs
 * (NodeProc/FaceProc, :1549-1612), assemble the jump-energy matrix M (analytic integrals for equal depths
 * :1459-1546, Gauss-Legendre ones otherwise :1250-1456), solve (M + strength I) x = strength c by
 * preconditioned CG from the guess strength c until |r| < tol |b| (the reference: tol = EPSILON_F32 with
 * Eigen's IncompleteCholesky; here Jacobi -- same solution to solver tolerance), overwrite the coefficients.
 * Deterministic for any thread count. */
typedef struct hpsdf_continuity_stats {
    uint64_t n_pairs, n_pairs_analytic, n_pairs_numeric, nnz, iterations;
    double residual;    /* |b - A x| / |b| at exit */
    double jump_before; /* c^T M c: jump energy of the fitted coefficients */
    double jump_after;  /* x^T M x */
    double assemble_ms, solve_ms;
} hpsdf_continuity_stats;
/* In place on a serialised block in host memory (the layout hpsdf_tree_upload takes); strength is read from
 * the block's Config.  tol <= 0: EPSILON_F32 (1e-6, Octree.cpp:1754); max_iter <= 0: 2n (Eigen's default);
 * threads 0: the block's config.thread_count (capped to the machine). */
HPSDF_API int hpsdf_continuity_post_process(void* block, size_t size, double tol, int max_iter, uint64_t threads,
                                            hpsdf_continuity_stats* stats);
/* The same on ctx's device (what hpsdf_create runs): assembly and conjugate-gradient loop in HBM, every entry and every
 * sum formed in the host path's order -- the block comes back bit-identical to hpsdf_continuity_post_process's.
 * (HPSDF_CONTINUITY_HOST_ASSEMBLY=1 in the environment keeps the assembly on the host.) */
HPSDF_API int hpsdf_continuity_post_process_device(hpsdf_ctx* ctx, void* block, size_t size, double tol, int max_iter,
                                                   uint64_t threads, hpsdf_continuity_stats* stats);
/* M itself (without the regularisation), CSR with duplicates summed; the three arrays are malloc'd, the
 * caller frees them.  Test / diagnostic hook. */
HPSDF_API int hpsdf_continuity_matrix(const void* block, size_t size, uint64_t threads, uint64_t** row_ptr,
                                      uint64_t** col, double** val, hpsdf_continuity_stats* stats);
/* The same matrix assembled on ctx's device (what hpsdf_create and hpsdf_continuity_post_process_device use) and copied
 * back: identical to hpsdf_continuity_matrix's arrays bit for bit.  HPSDF_ERR_UNSUPPORTED for trees the device assembly
 * leaves to the host (a leaf with more than 1024 face neighbours, own blocks beyond 2 GB). */
HPSDF_API int hpsdf_continuity_matrix_device(hpsdf_ctx* ctx, const void* block, size_t size, uint64_t** row_ptr, uint64_t** col,
                                             double** val, hpsdf_continuity_stats* stats);
/* stats of the last post-process hpsdf_create ran on this thread (zeros if it ran none) */
HPSDF_API int hpsdf_continuity_last_stats(hpsdf_continuity_stats* out);

/* Where the memory blocks of hpsdf_create / hpsdf_create_distributed come from.  By default they are malloc'd and the caller frees
 * them with free() -- MemoryBlock's contract (Utility.h: { size, ptr }, freed by the caller).  A host binding whose own block type
 * cannot adopt a malloc'd pointer (a Python bytes object, a Go slice, a JVM direct buffer) pays a second copy of the whole block for
 * that; with an allocator the library writes the block straight into the binding's memory: alloc(size, user) is called in place of
 * malloc(size) and the returned pointer comes back as *block (NULL: the build fails with HPSDF_ERR_OUT_OF_MEMORY).  A block the
 * build has begun but will not return (it allocates the block of a build that stops after the first round before it knows that the
 * build stops there; failures) goes to release(ptr, user).  Both are called on the thread that called Create, during the call.
 * alloc == NULL restores malloc / free.  (hpsdf_build_assemble always mallocs.) */
typedef void* (*hpsdf_block_alloc_fn)(size_t size, void* user);
typedef void (*hpsdf_block_release_fn)(void* block, void* user);
HPSDF_API void hpsdf_ctx_set_block_allocator(hpsdf_ctx* ctx, hpsdf_block_alloc_fn alloc, hpsdf_block_release_fn release, void* user);

/* whole Create on one GPU: begin .. assemble, then the continuity post-process when
 * cfg->continuity_enforce is set (Octree.cpp:341-344).  *block is malloc'd (caller frees) unless the context has a block allocator.
 * HPSDF_ERR_INVALID_ARGUMENT after the first round if the field is NaN or infinite at a sample point: the total error is then
 * NaN / infinite for good and the reference's loop (Octree.cpp:216) would refine until memory ends. */
HPSDF_API int hpsdf_create(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field,
                           uint64_t max_jobs_per_round, void** block, size_t* size, hpsdf_build_stats* stats);

/* ---- Create sharded over the GPUs of one node ---------------------------------------------------------------
 * One process (or thread) per GPU, each with its own context, all calling this with the same config, field and K.
 * Tree, frontier and field are replicated; every round's jobs are cut into `world` contiguous cost-balanced slices and
 * rank r fits slice r on its GPU; the ranks then exchange the 9 errors per job -- ONE all-gather per round -- and every
 * rank applies identical results in identical order; one more all-gather at the end reassembles the packed coefficient
 * store.  The block is byte-identical on every rank and identical to what world = 1 builds.
 * (Reference: there is no multi-process build; this is Octree::Create, Octree.cpp:312-352, with the job loop :194-309
 * sharded.)
 *
 * gather: in-place all-gather of equal parts on device memory -- rank r's contribution sits at
 * d_buf + r * bytes_per_rank when it is called; when the work it enqueues on `stream` (a hipStream_t) has run, every
 * rank's d_buf must hold all `world` parts.  It returns 0 on success.  With RCCL this is one call,
 *     ncclAllGather((char*)d_buf + rank * n, d_buf, n, ncclChar, comm, stream)      (include/hpsdf_rccl.hpp);
 * the library itself does not link RCCL.
 * Fields the device evaluates itself run the device-side frontier on up to 8 ranks (slices cut, errors decided and the tree
 * updated on every rank's GPU).  A nearness-weighted config does too (round 5): a weighted fit reads the cell's earlier rows, so
 * the round's part of the coefficient arena is laid out alike on every rank and exchanged as well -- one more call of `gather`
 * a round, in front of the errors' -- and nothing is exchanged at the end.  Host callbacks, logging, K > 4096 and worlds beyond 8
 * run the host scheduler's rounds, sharded the same way (same bytes): the exchanges are staged through a device buffer for the
 * same `gather`, and a weighted build hands the arrays each round accepted to every rank (hpsdf_build_rows_*).
 * The stepwise hpsdf_build_* calls expose the same loop to callers with a transport of their own.
 * stats: everything describes the whole build and is the same on every rank, except `fits` and `samples`, which count the rank's
 * own share (their sums over the ranks are the single-rank figures).
 * FAILURES: a rank whose share of a round fails on its own (device memory one GPU cannot serve) still enters the exchange the
 * other ranks are heading for, with a status word set in its part of the exchanged buffer, and then returns its error; every
 * other rank finds the status after the exchange and returns HPSDF_ERR_STATE naming the rank -- nobody is left waiting in a
 * collective (tests: test_a_failing_rank_takes_the_others_out_with_it).  What this cannot cover: a failure of the exchange
 * callback itself, a sticky device error (after which no further call succeeds), and the last exchange of the device-side
 * frontier (the packed coefficients, whose only local failure is the allocation of the packed store): there the caller must
 * abort the communicator when a rank reports an error, as for any collective program. */
typedef int (*hpsdf_allgather_fn)(void* user, void* d_buf, size_t bytes_per_rank, void* stream);
HPSDF_API int hpsdf_create_distributed(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field,
                                       uint64_t max_jobs_per_round, int rank, int world, hpsdf_allgather_fn gather,
                                       void* user, void** block, size_t* size, hpsdf_build_stats* stats);

/* ---- steady-state micro-benchmark hook (bench.py / profiles) ------------------
 * n_cells from-scratch fits of `degree` at `depth` over a lattice of cells, results discarded. */
HPSDF_API int hpsdf_bench_fit(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, int degree,
                              int depth, uint64_t n_cells, int repeats, double* ms_per_launch);
/* The same fits with their results: Octree::FitPolynomial (Octree.cpp:1007-1093) from scratch at `degree` for the first
 * n_cells cells of the depth-`depth` lattice over [-1/2, 1/2]^3 (x fastest): coeffs[n_cells][ncoef(degree)], errs[n_cells].
 * The kernels are the ones a build would take in the context's fit mode (hpsdf_ctx_set_fit_mode).
 * What the fit known-answer tests compare with the oracle's FitPolynomial, degree by degree. */
HPSDF_API int hpsdf_fit_cells(hpsdf_ctx* ctx, const hpsdf_config* cfg, const hpsdf_field* field, int degree, int depth,
                              uint64_t n_cells, double* coeffs, double* errs);

#ifdef __cplusplus
}
#endif
#endif /* HPSDF_H */
