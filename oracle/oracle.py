"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

NCOEF = [1, 4, 10, 20, 35, 56, 83, 120, 165, 220, 286, 364, 455]

PRIM_SPHERE, PRIM_BOX, PRIM_TORUS_Y, PRIM_PLANE = 0, 1, 2, 3
OP_UNION, OP_INTERSECT, OP_SUBTRACT = 0, 1, 2
FIELD_ANALYTIC, FIELD_CALLBACK, FIELD_MESH, FIELD_TREE_CSG = 0, 1, 2, 3


class Prim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("op", C.c_int32), ("p", C.c_double * 8)]


class Config(C.Structure):
    """80-byte SDF::Config (Include/HP/Config.h:12-43)."""
    _fields_ = [
        ("weighting_type", C.c_uint8), ("pad0", C.c_uint8 * 7),
        ("weighting_strength", C.c_double),
        ("continuity_enforce", C.c_uint8), ("pad1", C.c_uint8 * 7),
        ("continuity_strength", C.c_double),
        ("enable_logging", C.c_uint8), ("pad2", C.c_uint8 * 7),
        ("target_error_threshold", C.c_double),
        ("thread_count", C.c_uint64),
        ("root_min", C.c_float * 3),
        ("root_max", C.c_float * 3),
    ]


CALLBACK = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_uint64, C.c_void_p)


class Field(C.Structure):
    pass


Field._fields_ = [
    ("kind", C.c_int32), ("nprims", C.c_int32),
    ("prims", C.POINTER(Prim)),
    ("cb", CALLBACK), ("user", C.c_void_p),
    ("mesh", C.c_void_p), ("tree", C.c_void_p), ("inner", C.POINTER(Field)),
    ("csg_op", C.c_int32),
]


class JobResult(C.Structure):
    _fields_ = [("p_err", C.c_double), ("h_err", C.c_double * 8), ("p_imp", C.c_double), ("h_imp", C.c_double),
                ("refine_p", C.c_int32), ("refine_h", C.c_int32), ("coarse", C.c_int32)]


class ContinuityStats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_pairs_analytic", C.c_uint64), ("n_pairs_numeric", C.c_uint64),
                ("nnz", C.c_uint64), ("iterations", C.c_uint64), ("residual", C.c_double),
                ("jump_before", C.c_double), ("jump_after", C.c_double)]


class BuildStats(C.Structure):
    _fields_ = [("rounds", C.c_uint64), ("jobs", C.c_uint64), ("p_refines", C.c_uint64), ("h_refines", C.c_uint64),
                ("dropped", C.c_uint64), ("fits", C.c_uint64), ("total_error", C.c_double)]


def build(force=False):
    """Compile oracle/liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("hp_oracle.c", "hp_oracle_mesh.c", "hp_oracle_ray.c", "hp_oracle_continuity.c",
                                            "hp_oracle.h", "Makefile")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    ref_so = os.path.join(_HERE, "_ref", "libref_tables.so")
    if os.path.exists("/root/reference/Include/HP/Utility.h") and (force or not os.path.exists(ref_so)):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = build()
    L = C.CDLL(so)
    dp, u64p, fp = C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_float)
    for name, res in (("ora_gl_roots", dp), ("ora_gl_weights", dp), ("ora_normalised_lengths", dp),
                      ("ora_recurrence", dp), ("ora_coeff_count", u64p), ("ora_basis_index", u64p),
                      ("ora_sum_to_n", u64p)):
        getattr(L, name).restype = res
    L.ora_config_default.argtypes = [C.POINTER(Config)]
    L.ora_lpx.restype = C.c_double
    L.ora_lpx.argtypes = [C.c_uint64, C.c_double]
    L.ora_field_eval.restype = C.c_double
    L.ora_field_eval.argtypes = [C.POINTER(Field), dp]
    L.ora_fit_polynomial.restype = C.c_double
    L.ora_fit_polynomial.argtypes = [C.POINTER(Field), C.POINTER(Config), dp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int]
    L.ora_job.argtypes = [C.POINTER(Field), C.POINTER(Config), fp, fp, C.c_int, C.c_int, C.c_double, dp, dp, dp,
                          C.POINTER(JobResult), C.c_int]
    L.ora_fapprox.restype = C.c_double
    L.ora_fapprox.argtypes = [dp, C.c_int, fp, fp, dp, C.c_int]
    L.ora_create.restype = C.c_void_p
    L.ora_create.argtypes = [C.POINTER(Config), C.POINTER(Field), C.c_uint64, C.c_int, C.POINTER(BuildStats)]
    L.ora_create_mt.restype = C.c_void_p
    L.ora_create_mt.argtypes = [C.POINTER(Config), C.POINTER(Field), C.c_uint64, C.c_int, C.POINTER(BuildStats), C.c_int]
    L.ora_query_batch_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    L.ora_query_batch_mt_passes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    L.ora_tree_free.argtypes = [C.c_void_p]
    L.ora_tree_block_size.restype = C.c_size_t
    L.ora_tree_block_size.argtypes = [C.c_void_p]
    L.ora_tree_to_block.argtypes = [C.c_void_p, C.c_void_p]
    L.ora_tree_from_block.restype = C.c_void_p
    L.ora_tree_from_block.argtypes = [C.c_void_p, C.c_size_t]
    L.ora_query.restype = C.c_double
    L.ora_query.argtypes = [C.c_void_p, dp]
    L.ora_query_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.ora_query_gradient_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.ora_query_ray_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.ora_function_slice.argtypes = [C.c_void_p, C.c_double, fp, fp, C.c_uint64, C.c_void_p, C.c_void_p]
    L.ora_continuity_post_process.restype = C.c_int
    L.ora_continuity_post_process.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(ContinuityStats)]
    L.ora_continuity_matrix.restype = C.c_int
    L.ora_continuity_matrix.argtypes = [C.c_void_p, C.POINTER(u64p), C.POINTER(u64p), C.POINTER(dp),
                                        C.POINTER(ContinuityStats)]
    L.ora_mesh_create.restype = C.c_void_p
    L.ora_mesh_create.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.ora_mesh_free.argtypes = [C.c_void_p]
    L.ora_acosf_batch.argtypes = [C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p]
    L.ora_mesh_signed_distance.restype = C.c_float
    L.ora_mesh_signed_distance.argtypes = [C.c_void_p, fp, u64p, C.POINTER(C.c_int)]
    L.ora_set_reduction_order.argtypes = [C.c_int]
    L.ora_get_reduction_order.restype = C.c_int
    _LIB = L
    return L


def set_reduction_order(left_assoc):
    """hp_oracle.c: Eigen's 3-vector reductions as (a . b) . c (1) instead of a . (b . c) (0, the default).  Process-wide."""
    lib().ora_set_reduction_order(1 if left_assoc else 0)


def reduction_order():
    return int(lib().ora_get_reduction_order())


def ref_tables_lib():
    """The reference's own tables (oracle/_ref/libref_tables.so) or None."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_tables.so")
        if not os.path.exists(p):
            return None
        _REF = C.CDLL(p)
    return _REF


def _arr(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).copy()


def tables():
    L = lib()
    return {
        "roots": _arr(L.ora_gl_roots(), 2080, np.float64),
        "weights": _arr(L.ora_gl_weights(), 2080, np.float64),
        "normalised_lengths": _arr(L.ora_normalised_lengths(), 13 * 11, np.float64).reshape(13, 11),
        "recurrence": _arr(L.ora_recurrence(), 26, np.float64).reshape(13, 2),
        "coeff_count": _arr(L.ora_coeff_count(), 13, np.uint64),
        "basis_index": _arr(L.ora_basis_index(), 455 * 3, np.uint64).reshape(455, 3),
        "sum_to_n": _arr(L.ora_sum_to_n(), 50, np.uint64),
    }


def ref_tables():
    R = ref_tables_lib()
    if R is None:
        return None
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    out = {"roots": np.zeros(2080), "weights": np.zeros(2080), "normalised_lengths": np.zeros((13, 11)),
           "recurrence": np.zeros((13, 2)), "coeff_count": np.zeros(13, np.uint64),
           "basis_index": np.zeros((455, 3), np.uint64), "sum_to_n": np.zeros(50, np.uint64)}
    R.ref_gl(vp(out["roots"]), vp(out["weights"]))
    R.ref_normalised_lengths(vp(out["normalised_lengths"]))
    R.ref_recurrence(vp(out["recurrence"]))
    R.ref_coeff_count(vp(out["coeff_count"]))
    R.ref_basis_index(vp(out["basis_index"]))
    R.ref_sum_to_n(vp(out["sum_to_n"]))
    return out


# ----------------------------------------------------------------------------
# convenience wrappers
# ----------------------------------------------------------------------------
def default_config(target=1e-10, root_min=(-0.5, -0.5, -0.5), root_max=(0.5, 0.5, 0.5), continuity=False):
    c = Config()
    lib().ora_config_default(C.byref(c))
    c.target_error_threshold = target
    c.continuity_enforce = 1 if continuity else 0
    for a in range(3):
        c.root_min[a] = root_min[a]
        c.root_max[a] = root_max[a]
    return c


def make_prims(spec):
    """spec: list of (kind, op, params) -> ctypes Prim array."""
    arr = (Prim * len(spec))()
    for i, (kind, op, params) in enumerate(spec):
        arr[i].kind, arr[i].op = kind, op
        for j, v in enumerate(params):
            arr[i].p[j] = float(v)
    return arr


class AnalyticField:
    def __init__(self, spec):
        self.spec = list(spec)
        self.prims = make_prims(self.spec)
        self.f = Field()
        self.f.kind = FIELD_ANALYTIC
        self.f.nprims = len(self.spec)
        self.f.prims = C.cast(self.prims, C.POINTER(Prim))

    def eval(self, pts):
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        L = lib()
        out = np.empty(len(pts))
        for i in range(len(pts)):
            out[i] = L.ora_field_eval(C.byref(self.f), pts[i].ctypes.data_as(C.POINTER(C.c_double)))
        return out


class MeshField:
    def __init__(self, verts, tris):
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        self.tris = np.ascontiguousarray(tris, np.uint64).reshape(-1, 3)
        L = lib()
        self.handle = L.ora_mesh_create(self.verts.ctypes.data_as(C.c_void_p), len(self.verts),
                                        self.tris.ctypes.data_as(C.c_void_p), len(self.tris))
        if not self.handle:
            raise ValueError("mesh is not closed (Mesh::CreateHalfEdges would fail)")
        self.f = Field()
        self.f.kind = FIELD_MESH
        self.f.mesh = self.handle

    def signed_distance(self, pts):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
        L = lib()
        out = np.empty(len(pts), np.float32)
        tri = np.empty(len(pts), np.uint64)
        simp = np.empty(len(pts), np.int32)
        t, s = C.c_uint64(), C.c_int()
        for i in range(len(pts)):
            out[i] = L.ora_mesh_signed_distance(self.handle, pts[i].ctypes.data_as(C.POINTER(C.c_float)),
                                                C.byref(t), C.byref(s))
            tri[i], simp[i] = t.value, s.value
        return out, tri, simp

    def __del__(self):
        if getattr(self, "handle", None):
            lib().ora_mesh_free(self.handle)
            self.handle = None


def acosf_batch(first_bits, stride, n):
    """acosf of the host libm for the floats with bit patterns first_bits + i * stride (numpy's arccos is not libm's)."""
    out = np.empty(n, np.float32)
    lib().ora_acosf_batch(first_bits, stride, n, out.ctypes.data_as(C.c_void_p))
    return out


class TreeCsgField:
    """F' = op(oldTree.Query, inner) as in Octree.cpp:355-400."""

    def __init__(self, tree, inner, op):
        self.tree, self.inner = tree, inner
        self.f = Field()
        self.f.kind = FIELD_TREE_CSG
        self.f.tree = tree.handle
        self.f.inner = C.pointer(inner.f)
        self.f.csg_op = op


def sphere_field(centre=(0.25, 0.0, 0.0), radius=0.5):
    """Source/Tests/HPUnitTests.cpp:48-51."""
    return AnalyticField([(PRIM_SPHERE, OP_UNION, list(centre) + [radius])])


def union3_field():
    """BASELINE config[1] / SURVEY 8(d) C2."""
    return AnalyticField([
        (PRIM_SPHERE, OP_UNION, [-0.2, -0.15, 0.1, 0.18]),
        (PRIM_BOX, OP_UNION, [0.15, 0.2, -0.1, 0.12, 0.10, 0.15]),
        (PRIM_TORUS_Y, OP_UNION, [0.0, -0.2, -0.2, 0.15, 0.05]),
    ])


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def fit_polynomial(field, cfg, bmin, bmax, degree, depth, coeffs_in=None, basis_degree=0, literal=False):
    L = lib()
    co = np.zeros(NCOEF[degree])
    if coeffs_in is not None:
        co[:len(coeffs_in)] = coeffs_in
    err = L.ora_fit_polynomial(C.byref(field.f), C.byref(cfg), co.ctypes.data_as(C.POINTER(C.c_double)),
                               basis_degree, _f3(bmin), _f3(bmax), degree, depth, 1 if literal else 0)
    return co, err


def job(field, cfg, bmin, bmax, depth, degree, err, coeffs, literal=False):
    L = lib()
    coarse = abs(err - 100.0) < np.finfo(np.float64).eps
    pc = np.zeros(NCOEF[2 if coarse else min(degree + 1, 12)])
    hc = np.zeros(8 * NCOEF[degree])
    r = JobResult()
    cin = np.ascontiguousarray(coeffs if coeffs is not None else np.zeros(1), np.float64)
    dp = C.POINTER(C.c_double)
    L.ora_job(C.byref(field.f), C.byref(cfg), _f3(bmin), _f3(bmax), depth, degree, err,
              cin.ctypes.data_as(dp), pc.ctypes.data_as(dp), hc.ctypes.data_as(dp), C.byref(r), 1 if literal else 0)
    return r, pc, hc.reshape(8, -1)


class Tree:
    def __init__(self, handle):
        self.handle = handle

    @staticmethod
    def create(cfg, field, K=1024, literal=False, threads=1):
        """threads > 1: a round's jobs on that many pthreads (same tree; fields implemented in C only)."""
        st = BuildStats()
        if threads > 1:
            h = lib().ora_create_mt(C.byref(cfg), C.byref(field.f), K, 1 if literal else 0, C.byref(st), threads)
        else:
            h = lib().ora_create(C.byref(cfg), C.byref(field.f), K, 1 if literal else 0, C.byref(st))
        t = Tree(h)
        t.stats = {k: getattr(st, k) for k, _ in BuildStats._fields_}
        return t

    @staticmethod
    def from_block(block):
        b = bytes(block)
        h = lib().ora_tree_from_block(b, len(b))
        if not h:
            raise ValueError("bad block")
        return Tree(h)

    def to_block(self):
        L = lib()
        n = L.ora_tree_block_size(self.handle)
        buf = C.create_string_buffer(n)
        L.ora_tree_to_block(self.handle, buf)
        return buf.raw

    def query(self, pts, threads=1, passes=1):
        """threads > 1: the points cut into that many contiguous parts, one pthread each, every thread going over its part
        `passes` times (timing loops: the threads are started once)."""
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty(len(pts))
        if threads > 1:
            lib().ora_query_batch_mt_passes(self.handle, pts.ctypes.data_as(C.c_void_p), len(pts), out.ctypes.data_as(C.c_void_p), threads,
                                            passes)
        else:
            lib().ora_query_batch(self.handle, pts.ctypes.data_as(C.c_void_p), len(pts), out.ctypes.data_as(C.c_void_p))
        return out

    def query_with_gradient(self, pts, grad_init=None):
        """grad rows of points outside the root keep grad_init (the reference leaves the output untouched)."""
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty(len(pts))
        grad = np.zeros((len(pts), 3)) if grad_init is None else np.array(grad_init, np.float64).reshape(-1, 3).copy()
        lib().ora_query_gradient_batch(self.handle, pts.ctypes.data_as(C.c_void_p), len(pts),
                                       out.ctypes.data_as(C.c_void_p), grad.ctypes.data_as(C.c_void_p))
        return out, grad

    def query_ray(self, origins, directions, t_max, t_init=None):
        """Octree::QueryRay per row: (hit u8, t) -- t rows of misses keep t_init."""
        o = np.ascontiguousarray(origins, np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float64).reshape(-1, 3)
        n = len(o)
        tm = np.ascontiguousarray(np.broadcast_to(np.asarray(t_max, np.float64), (n,)))
        hit = np.zeros(n, np.uint8)
        t = np.zeros(n) if t_init is None else np.array(t_init, np.float64).reshape(n).copy()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        lib().ora_query_ray_batch(self.handle, vp(o), vp(d), vp(tm), n, vp(hit), vp(t))
        return hit, t

    def function_slice(self, c, view_min, view_max, n_samples=2048):
        """Octree::OutputFunctionSlice up to the byte image: (rgb [n,n,3] u8, values [n,n] f64)."""
        n = int(n_samples)
        rgb = np.zeros((n, n, 3), np.uint8)
        vals = np.zeros((n, n))
        lib().ora_function_slice(self.handle, float(c), _f3(view_min), _f3(view_max), n,
                                 rgb.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p))
        return rgb, vals

    def continuity_post_process(self, tol=1e-6, max_iter=0):
        """Octree::PerformContinuityPostProcess in place; returns the stats dict."""
        st = ContinuityStats()
        rc = lib().ora_continuity_post_process(self.handle, tol, max_iter, C.byref(st))
        if rc < 0:
            raise RuntimeError("continuity failed")
        return {k: getattr(st, k) for k, _ in ContinuityStats._fields_}

    def continuity_matrix(self):
        """(row_ptr, col, val) CSR of the jump-energy matrix M (no regularisation), duplicates summed."""
        L = lib()
        dp, u64p = C.POINTER(C.c_double), C.POINTER(C.c_uint64)
        rp, ci, v = u64p(), u64p(), dp()
        st = ContinuityStats()
        L.ora_continuity_matrix(self.handle, C.byref(rp), C.byref(ci), C.byref(v), C.byref(st))
        n = len(parse_block(self.to_block())["coeffs"])
        row_ptr = np.ctypeslib.as_array(rp, shape=(n + 1,)).copy()
        nnz = int(row_ptr[-1])
        col = np.ctypeslib.as_array(ci, shape=(max(nnz, 1),))[:nnz].copy()
        val = np.ctypeslib.as_array(v, shape=(max(nnz, 1),))[:nnz].copy()
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        for p in (rp, ci, v):
            libc.free(C.cast(p, C.c_void_p))
        return row_ptr, col, val, {k: getattr(st, k) for k, _ in ContinuityStats._fields_}

    def __del__(self):
        if getattr(self, "handle", None):
            lib().ora_tree_free(self.handle)
            self.handle = None


def parse_block(block):
    """Layout of SURVEY 8(a-D): returns dict of numpy views (copies)."""
    b = np.frombuffer(bytes(block), np.uint8)
    ncoef = int(b[:8].view(np.uint64)[0])
    coeffs = b[8:8 + 8 * ncoef].view(np.float64).copy()
    o = 8 + 8 * ncoef
    nnodes = int(b[o:o + 8].view(np.uint64)[0])
    o += 8
    nodes = b[o:o + 56 * nnodes].reshape(nnodes, 56)
    out = {
        "n_coeffs": ncoef, "coeffs": coeffs, "n_nodes": nnodes,
        "childIdx": nodes[:, 0:8].copy().view(np.uint64).reshape(-1),
        "aabb": nodes[:, 8:32].copy().view(np.float32).reshape(-1, 6),
        "coeffsStart": nodes[:, 32:40].copy().view(np.uint64).reshape(-1),
        "degree": nodes[:, 40].copy(), "depth": nodes[:, 48].copy(),
        "config": bytes(b[o + 56 * nnodes:]),
    }
    return out


def splitmix64_points(n, seed=12345):
    """SURVEY 8(d) C2 query set: SplitMix64(seed) -> (u >> 11) * 2^-53 - 0.5, xyz interleaved."""
    m = 3 * n
    idx = np.arange(1, m + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53 - 0.5).reshape(n, 3)
