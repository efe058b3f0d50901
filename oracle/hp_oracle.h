/*
 * hp_oracle.h -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * Plain-C, single-thread CPU restatement of the reference's hot path
 * (jw007123/hp-Adaptive-Signed-Distance-Field-Octree): per-node Legendre
 * fit (Octree::Create) and Octree::Query, plus the MemoryBlock layout.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libhpsdf.so) never links or calls it.
 *
 * Pinning status (see oracle/README.md and DESIGN.md "Oracle"):
 *   - constant tables and scalars (limits, typedef widths, EPSILON_F32): PINNED bit-for-bit against the reference's own
 *     headers (Include/HP/Utility.h, Legendre.h, Consts.h, Include/Utility/Literals.h, MemoryBlock.h compiled from where they
 *     lie into oracle/_ref/libref_tables.so; tests/test_oracle_tables.py).
 *   - fit / query numerics: checked against the known-answer values of SURVEY.md
 *     Appendix C (10-13 significant digits; the survey produced them with a stand-in for Eigen, so they are a consistency
 *     check, not a pin) and against the reference's own end-to-end tests (Source/Tests/HPUnitTests.cpp:46-77,115-154,285-316:
 *     |Query - true| <= 1e-2).  The reference holds no golden coefficient,
 *     topology or serialisation vectors, and Source/HP/Octree.cpp itself is
 *     UNBUILDABLE here (needs Eigen, which is neither vendored nor installed),
 *     so coefficient-level parity against the reference binary is
 *     "parity unpinned" beyond those values.  Nearness weighting (Octree.cpp:1209-1247) draws from
 *     std::rand() in the reference: restated with a hashed generator, parity unpinned by construction.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).
 */
#ifndef HP_ORACLE_H
#define HP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Include/HP/Consts.h:7-8 */
#define ORA_BASIS_MAX_DEGREE 12
#define ORA_TREE_MAX_DEPTH 10
#define ORA_INTERIOR_DEGREE 13 /* BASIS_MAX_DEGREE + 1 marks interior nodes */
#define ORA_NCOEF_MAX 455
#define ORA_INITIAL_NODE_ERR 100.0 /* Include/HP/Octree.h:89 */

/* ---- tables (Include/HP/Utility.h, Include/HP/Legendre.h) ---------------- */
void ora_tables_init(void);
const double* ora_gl_roots(void);           /* [2080], rule n at n(n-1)/2 */
const double* ora_gl_weights(void);         /* [2080] */
const double* ora_normalised_lengths(void); /* [13][11] */
const double* ora_recurrence(void);         /* [13][2] */
const uint64_t* ora_coeff_count(void);      /* [13]  (count[6] == 83) */
const uint64_t* ora_basis_index(void);      /* [455][3] */
const uint64_t* ora_sum_to_n(void);         /* [50] */

/* ---- serialised structs (Include/HP/Node.h:10-33, Config.h:12-43) -------- */
typedef struct {
    uint64_t childIdx; /* @0  all-ones = leaf */
    float aabb_min[3]; /* @8  */
    float aabb_max[3]; /* @20 */
    uint64_t coeffsStart; /* @32 */
    uint8_t degree;       /* @40 13 = interior */
    uint8_t pad0[7];
    uint8_t depth; /* @48 */
    uint8_t pad1[7];
} ora_node; /* 56 bytes */

typedef struct {
    uint8_t weighting_type; /* @0 */
    uint8_t pad0[7];
    double weighting_strength; /* @8 */
    uint8_t continuity_enforce; /* @16 */
    uint8_t pad1[7];
    double continuity_strength; /* @24 */
    uint8_t enable_logging;     /* @32 */
    uint8_t pad2[7];
    double target_error_threshold; /* @40 */
    uint64_t thread_count;         /* @48 */
    float root_min[3];             /* @56 */
    float root_max[3];             /* @68 */
} ora_config; /* 80 bytes */

void ora_config_default(ora_config* c); /* Source/HP/Config.cpp:5-14 */

/* ---- fields (the callback F of Include/HP/Octree.h:50) ------------------- */
enum { ORA_PRIM_SPHERE = 0, ORA_PRIM_BOX = 1, ORA_PRIM_TORUS_Y = 2, ORA_PRIM_PLANE = 3 };
enum { ORA_OP_UNION = 0, ORA_OP_INTERSECT = 1, ORA_OP_SUBTRACT = 2 };
typedef struct {
    int32_t kind;
    int32_t op; /* how this primitive combines with the running value */
    double p[8];
} ora_prim;

typedef double (*ora_callback)(const double* pt, uint64_t thread_idx, void* user);

enum { ORA_FIELD_ANALYTIC = 0, ORA_FIELD_CALLBACK = 1, ORA_FIELD_MESH = 2, ORA_FIELD_TREE_CSG = 3 };
struct ora_tree;
struct ora_mesh;
typedef struct ora_field {
    int32_t kind;
    int32_t nprims;
    const ora_prim* prims;
    ora_callback cb;
    void* user;
    const struct ora_mesh* mesh;   /* ORA_FIELD_MESH */
    const struct ora_tree* tree;   /* ORA_FIELD_TREE_CSG: old tree */
    const struct ora_field* inner; /* ORA_FIELD_TREE_CSG: the new F_ */
    int32_t csg_op;                /* ORA_OP_* applied as in Octree.cpp:355-400 */
} ora_field;

double ora_field_eval(const ora_field* f, const double pt[3]);
/* 0 (default): 3-vector reductions associate as a + (b + c); 1: (a + b) + c -- sensitivity study only, see hp_oracle.c */
void ora_set_reduction_order(int left_assoc);
int ora_get_reduction_order(void);

/* ---- per-node numerics --------------------------------------------------- */
/* Octree::LpX, Octree.cpp:988-1004 */
double ora_lpx(uint64_t p, double x);

/* Octree::FitPolynomial, Octree.cpp:1007-1093 (error times the nearness weight when cfg->weighting_type != 0).
 * coeffs holds ncoef(degree) doubles; rows < ncoef(basis_degree) are kept when
 * basis_degree > 0.  `literal` != 0 re-runs LpX in the innermost loop exactly
 * as the reference does; 0 uses per-axis tables of the same values (bitwise
 * identical result, tests/test_oracle_fit.py). */
double ora_fit_polynomial(const ora_field* f, const ora_config* cfg, double* coeffs, int basis_degree,
                          const float bmin[3], const float bmax[3], int degree, int depth, int literal);

/* Nearness weighting, Octree.cpp:1209-1247, with a deterministic stand-in for aabb.sample() (std::rand in the
 * reference): parity unpinned, see hp_oracle.c */
uint64_t ora_weight_key(const float bmin[3], int depth, int degree);
double ora_weight_mean(const double* coeffs, int degree, const float bmin[3], const float bmax[3], int depth);
double ora_weight_from_mean(int type, double strength, double mean);
double ora_poly_weighting(const double* coeffs, int degree, const float bmin[3], const float bmax[3], int depth, double strength);
double ora_exp_weighting(const double* coeffs, int degree, const float bmin[3], const float bmax[3], int depth, double strength);

/* Octree::CornerAABB, Octree.cpp:1096-1112 */
void ora_corner_aabb(const float bmin[3], const float bmax[3], unsigned i, float omin[3], float omax[3]);

typedef struct {
    double p_err, h_err[8];
    double p_imp, h_imp;
    int32_t refine_p, refine_h, coarse;
} ora_job_result;

/* EstimateHImprovement + EstimatePImprovement + decision,
 * Octree.cpp:804-856, 594-601.  p_coeffs: ncoef(p+1) (or 10 if coarse);
 * h_coeffs: 8*ncoef(p). */
void ora_job(const ora_field* f, const ora_config* cfg, const float bmin[3], const float bmax[3], int depth,
             int degree, double err, const double* coeffs, double* p_coeffs, double* h_coeffs,
             ora_job_result* out, int literal);

/* Octree::FApprox, Octree.cpp:859-901 */
double ora_fapprox(const double* coeffs, int degree, const float bmin[3], const float bmax[3],
                   const double pt[3], int depth);

/* ---- tree ---------------------------------------------------------------- */
typedef struct ora_tree {
    ora_config config;
    double root_centre[3];   /* Octree.cpp:322 */
    double root_inv_sizes[3]; /* Octree.cpp:323 */
    uint64_t n_nodes, cap_nodes;
    ora_node* nodes;
    uint64_t n_coeffs;
    double* coeff_store;
} ora_tree;

typedef struct {
    uint64_t rounds, jobs, p_refines, h_refines, dropped, fits;
    double total_error;
} ora_build_stats;

/* Octree::Create, Octree.cpp:312-352, under the canonical round schedule of
 * SURVEY.md Appendix B (round 0 = all coarse jobs, then top-K by (err desc,
 * nodeIdx asc)).  max_jobs_per_round = K. */
ora_tree* ora_create(const ora_config* cfg, const ora_field* f, uint64_t max_jobs_per_round, int literal,
                     ora_build_stats* stats);
/* the same with a round's jobs evaluated by `threads` pthreads (same tree for any thread count; C fields only) */
ora_tree* ora_create_mt(const ora_config* cfg, const ora_field* f, uint64_t max_jobs_per_round, int literal,
                        ora_build_stats* stats, int threads);
void ora_tree_free(ora_tree* t);

/* Octree::ToMemoryBlock / FromMemoryBlock, Octree.cpp:424-456, 403-421.
 * Padding bytes and interior coeffsStart are written as zero. */
size_t ora_tree_block_size(const ora_tree* t);
void ora_tree_to_block(const ora_tree* t, void* out);
ora_tree* ora_tree_from_block(const void* block, size_t size);

/* Octree::Query, Octree.cpp:662-702 */
double ora_query(const ora_tree* t, const double pt[3]);
void ora_query_batch(const ora_tree* t, const double* xyz, size_t n, double* out);
void ora_query_batch_mt(const ora_tree* t, const double* xyz, size_t n, double* out, int threads);
void ora_query_batch_mt_passes(const ora_tree* t, const double* xyz, size_t n, double* out, int threads, int passes);
/* Octree::QueryWithGradient / FApproxWithGradient, Octree.cpp:749-789, 904-985 */
double ora_fapprox_with_gradient(const double* coeffs, int degree, const float bmin[3], const float bmax[3],
                                 const double pt[3], int depth, double grad[3]);
double ora_query_with_gradient(const ora_tree* t, const double pt[3], double grad[3]);
void ora_query_gradient_batch(const ora_tree* t, const double* xyz, size_t n, double* out, double* grad);

/* ---- query-side helpers on top of Query -- see hp_oracle_ray.c.  Parity unpinned: the
 * reference has no test of either and OutputFunctionSlice needs stb (absent). */
/* Octree::QueryRay, Octree.cpp:705-746 (Ray, Source/HP/Ray.cpp).  1 = hit (*t_out written) */
int ora_query_ray(const ora_tree* t, const double origin[3], const double direction[3], double t_max, double* t_out);
void ora_query_ray_batch(const ora_tree* t, const double* origins, const double* directions, const double* t_max,
                         size_t n, uint8_t* hit, double* t_out);
/* Octree::OutputFunctionSlice, Octree.cpp:1131-1206, up to the RGB byte image */
void ora_function_slice(const ora_tree* t, double c, const float view_min[3], const float view_max[3],
                        uint64_t n_samples, uint8_t* rgb, double* values);
uint8_t ora_f64_to_u8(double q);

/* ---- continuity post-process -- see hp_oracle_continuity.c ---
 * Octree::PerformContinuityPostProcess, Octree.cpp:1250-1762: face-pair enumeration, jump-energy matrix,
 * (M + strength I) x = strength c solved by preconditioned CG to relative residual `tol` (reference:
 * EPSILON_F32 = 1e-6 with Eigen's IncompleteCholesky; here Jacobi -- the solver is unpinned Eigen arithmetic,
 * so parity is to solver tolerance only).  Overwrites t->coeff_store.  Returns CG iterations, < 0 on error. */
typedef struct {
    uint64_t n_pairs, n_pairs_analytic, n_pairs_numeric, nnz, iterations;
    double residual;     /* |b - A x| / |b| at exit */
    double jump_before;  /* c^T M c   (jump energy of the input coefficients) */
    double jump_after;   /* x^T M x */
} ora_continuity_stats;
int ora_continuity_post_process(ora_tree* t, double tol, int max_iter, ora_continuity_stats* stats);
/* the assembled matrix (without the regularisation) in CSR, duplicates summed; caller frees the three arrays */
int ora_continuity_matrix(const ora_tree* t, uint64_t** row_ptr, uint64_t** col, double** val,
                          ora_continuity_stats* stats);

/* ---- mesh field (Source/Meshing) -- see hp_oracle_mesh.c ---------------- */
typedef struct ora_mesh ora_mesh;
ora_mesh* ora_mesh_create(const float* verts, uint64_t nverts, const uint64_t* tris, uint64_t ntris);
void ora_mesh_free(ora_mesh* m);
/* naive O(n) signed distance, the reference test's own cross-check
 * (Source/Tests/MeshingUnitTests.cpp:110-138) */
float ora_mesh_signed_distance(const ora_mesh* m, const float pt[3], uint64_t* tri_out, int* simplex_out);
/* acosf of this machine's libm (what Mesh.cpp:226-231's std::acos is) for the floats with bit patterns first + i * stride */
void ora_acosf_batch(uint32_t first, uint32_t stride, size_t n, float* out);
/* the restatement's scalars in the order of ref_tables.cpp's ref_scalars: max degree, max depth, sizeof i16 / i32 / u32 / usize /
 * MemoryBlock as the reference's typedefs make them on LP64 (Literals.h:3-11), the bits of EPSILON_F32 */
void ora_scalars(unsigned long long out[8]);

#ifdef __cplusplus
}
#endif
#endif
