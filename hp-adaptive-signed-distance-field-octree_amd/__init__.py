"""MI355X-native hp-adaptive SDF octree: Python host mirror of the C ABI (include/hpsdf.h).

The directory name is the one the build contract prescribes and is not a Python
identifier; load it with ``hpsdf_loader.load()`` (repo root) or importlib.  All numerics
run in ``lib/libhpsdf.so`` (HIP, gfx950).  There is no CPU fallback: without the built
library ``lib()`` raises, and without a GPU every compute entry point returns
HPSDF_ERR_NO_DEVICE, which surfaces here as ``HpsdfError``.

Reference API mirrored (file:line in the reference checkout):
  Config            Include/HP/Config.h:12-43, Source/HP/Config.cpp:5-32
  Octree.Create     Include/HP/Octree.h:50      Source/HP/Octree.cpp:312-352
  Octree.Query      Include/HP/Octree.h:71      Source/HP/Octree.cpp:662-702
  To/FromMemoryBlock Include/HP/Octree.h:65-68  Source/HP/Octree.cpp:403-456
  Union/Subtract/IntersectSDF Include/HP/Octree.h:53-59, Octree.cpp:355-400
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HPSDF_LIBRARY=hooks: lib/libhpsdf_hooks.so, the same library with the tests' fault-injection hook compiled in -- tests/ only;
#  HPSDF_LIBRARY=lab: lib/libhpsdf_lab.so, made on demand by `build.py --lab` for tools/query_general_floor.py -- its Query lab kernels
#  return values that are not the tree's)
#  (any other name: lib/libhpsdf_<name>.so, a measurement variant made by `build.py --variant=<name>:<flags>`)
_WHICH = os.environ.get("HPSDF_LIBRARY", "")
LIB_PATH = os.path.join(_HERE, "lib", "libhpsdf_%s.so" % _WHICH if _WHICH else "libhpsdf.so")
_LIB = None

OK = 0
ERR_INVALID_ARGUMENT = 1
ERR_NO_DEVICE = 2
ERR_HIP = 3
ERR_BAD_BLOCK = 4
ERR_UNSUPPORTED = 5
ERR_STATE = 6
ERR_OUT_OF_MEMORY = 7
ERR_OPEN_MESH = 8
ERR_BUILD_LIMIT = 9
ABI_VERSION = 4  # HPSDF_ABI_VERSION this binding was written against (lib() refuses a library built with another)
PRIM_SPHERE, PRIM_BOX, PRIM_TORUS_Y, PRIM_PLANE = 0, 1, 2, 3
OP_UNION, OP_INTERSECT, OP_SUBTRACT = 0, 1, 2
JOB_HEADER_DOUBLES = 9
DEFAULT_JOBS_PER_ROUND = 1024
NCOEF = [1, 4, 10, 20, 35, 56, 83, 120, 165, 220, 286, 364, 455]
DBL_MAX = float(np.finfo(np.float64).max)


class HpsdfError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("hpsdf status %d: %s" % (status, msg))
        self.status = status


class PodConfig(C.Structure):
    """hpsdf_config == SDF::Config, 80 bytes."""
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


class Prim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("op", C.c_int32), ("p", C.c_double * 8)]


class BuildOpts(C.Structure):
    _fields_ = [("max_jobs_per_round", C.c_uint64), ("rank", C.c_int32), ("world", C.c_int32),
                ("reserved", C.c_int32 * 2)]


class Job(C.Structure):
    _fields_ = [("node_idx", C.c_uint64), ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3),
                ("err", C.c_double), ("degree", C.c_uint8), ("depth", C.c_uint8), ("coarse", C.c_uint8),
                ("pad", C.c_uint8 * 5)]


class BuildStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rounds", "jobs", "p_refines", "h_refines", "dropped", "fits", "samples",
                                           "n_nodes", "n_leaves", "n_coeffs")] + [("total_error", C.c_double), ("fit_mode", C.c_uint64),
                                                                                    ("split_fits", C.c_uint64), ("device_frontier", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ContinuityStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_pairs", "n_pairs_analytic", "n_pairs_numeric", "nnz", "iterations")] + \
               [(n, C.c_double) for n in ("residual", "jump_before", "jump_after", "assemble_ms", "solve_ms")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


CALLBACK = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_uint64, C.c_void_p)

# every symbol include/hpsdf.h declares (tests/test_capi_symbols.py checks the list against the header)
_SIGNATURES = {
    "hpsdf_config_default": (C.c_int, [C.POINTER(PodConfig)]),
    "hpsdf_last_error": (C.c_char_p, []),
    "hpsdf_version": (C.c_char_p, []),
    "hpsdf_tables_get": (C.c_int, [C.c_void_p] * 7),
    "hpsdf_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "hpsdf_ctx_destroy": (C.c_int, [C.c_void_p]),
    "hpsdf_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hpsdf_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "hpsdf_ctx_set_fast_fit": (C.c_int, [C.c_void_p, C.c_int]),
    "hpsdf_ctx_set_fit_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "hpsdf_ctx_get_fit_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "hpsdf_ctx_set_split_min_degree": (C.c_int, [C.c_void_p, C.c_int]),
    "hpsdf_ctx_set_block_allocator": (None, [C.c_void_p] * 4),
    "hpsdf_field_release_host_copies": (C.c_int, [C.c_void_p]),
    "hpsdf_set_mesh_face_rule": (None, [C.c_int]),
    "hpsdf_get_mesh_face_rule": (C.c_int, []),
    "hpsdf_set_reduction_order": (None, [C.c_int]),
    "hpsdf_get_reduction_order": (C.c_int, []),
    "hpsdf_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "hpsdf_abi_version": (C.c_int, []),
    "hpsdf_ctx_set_reduction_order": (C.c_int, [C.c_void_p, C.c_int]),
    "hpsdf_ctx_get_reduction_order": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "hpsdf_ctx_set_mesh_face_rule": (C.c_int, [C.c_void_p, C.c_int]),
    "hpsdf_ctx_get_mesh_face_rule": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "hpsdf_ctx_set_build_limits": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "hpsdf_ctx_get_build_limits": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hpsdf_field_create_analytic": (C.c_int, [C.POINTER(Prim), C.c_int, C.POINTER(C.c_void_p)]),
    "hpsdf_field_create_callback": (C.c_int, [CALLBACK, C.c_void_p, C.POINTER(C.c_void_p)]),
    "hpsdf_field_create_mesh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                          C.POINTER(C.c_void_p)]),
    "hpsdf_obj_load": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint64),
                                 C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]),
    "hpsdf_field_create_tree_csg": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "hpsdf_field_destroy": (C.c_int, [C.c_void_p]),
    "hpsdf_field_mesh_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    "hpsdf_field_eval_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hpsdf_field_eval_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hpsdf_field_eval_naive_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hpsdf_field_eval_wave_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hpsdf_field_eval_lane_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hpsdf_selftest_acosf": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p]),
    "hpsdf_tree_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "hpsdf_tree_destroy": (C.c_int, [C.c_void_p]),
    "hpsdf_tree_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hpsdf_query_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hpsdf_query_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hpsdf_query_gradient_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "hpsdf_query_gradient_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "hpsdf_query_ray_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_void_p]),
    "hpsdf_query_ray_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_void_p]),
    "hpsdf_function_slice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       C.c_uint64, C.c_void_p, C.c_void_p]),
    "hpsdf_build_begin": (C.c_int, [C.POINTER(PodConfig), C.POINTER(BuildOpts), C.POINTER(C.c_void_p)]),
    "hpsdf_build_destroy": (C.c_int, [C.c_void_p]),
    "hpsdf_build_round_select": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "hpsdf_build_round_jobs": (C.c_int, [C.c_void_p, C.POINTER(Job)]),
    "hpsdf_build_round_slice": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hpsdf_build_round_max_slice": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "hpsdf_build_round_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpsdf_build_round_results_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "hpsdf_build_round_results_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpsdf_build_round_apply": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hpsdf_build_round_inject": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "hpsdf_build_rows_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "hpsdf_build_rows_pack_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpsdf_build_rows_unpack_host": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "hpsdf_build_node_rows_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]),
    "hpsdf_build_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "hpsdf_build_pack_device": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "hpsdf_build_pack_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hpsdf_build_assemble": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t)]),
    "hpsdf_build_get_stats": (C.c_int, [C.c_void_p, C.POINTER(BuildStats)]),
    "hpsdf_continuity_post_process": (C.c_int, [C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_uint64,
                                                C.POINTER(ContinuityStats)]),
    "hpsdf_continuity_post_process_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_int, C.c_uint64,
                                                       C.POINTER(ContinuityStats)]),
    "hpsdf_continuity_matrix": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(C.POINTER(C.c_uint64)),
                                          C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_double)),
                                          C.POINTER(ContinuityStats)]),
    "hpsdf_continuity_matrix_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint64)),
                                          C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_double)),
                                          C.POINTER(ContinuityStats)]),
    "hpsdf_continuity_last_stats": (C.c_int, [C.POINTER(ContinuityStats)]),
    "hpsdf_create": (C.c_int, [C.c_void_p, C.POINTER(PodConfig), C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p),
                               C.POINTER(C.c_size_t), C.POINTER(BuildStats)]),
    "hpsdf_create_distributed": (C.c_int, [C.c_void_p, C.POINTER(PodConfig), C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(BuildStats)]),
    "hpsdf_fit_cells": (C.c_int, [C.c_void_p, C.POINTER(PodConfig), C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]),
    "hpsdf_bench_fit": (C.c_int, [C.c_void_p, C.POINTER(PodConfig), C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                  C.POINTER(C.c_double)]),
}


def lib():
    """The HIP extension.  Raises if it has not been built: there is no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libhpsdf.so is not built (run __graft_entry__.build() or "
                              "python hp-adaptive-signed-distance-field-octree_amd/build.py); "
                              "there is no CPU fallback for the hot path")
        try:
            # torch-rocm bundles its own HIP/HSA runtime under the same SONAME; loading it first makes
            # this library bind to that one copy (two HSA runtimes in one process cannot both see the GPU)
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.hpsdf_abi_version() != ABI_VERSION:  # (structs are handed over by size: a library of another ABI would write past them)
            raise ImportError("libhpsdf.so has ABI version %d, this binding was written against %d: rebuild the library" % (L.hpsdf_abi_version(), ABI_VERSION))
        L._libc = C.CDLL(None)
        L._libc.free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def check(rc):
    if rc != OK:
        raise HpsdfError(rc, lib().hpsdf_last_error().decode("utf-8", "replace"))


def tables():
    L = lib()
    out = {"roots": np.zeros(2080), "weights": np.zeros(2080), "normalised_lengths": np.zeros((13, 11)),
           "recurrence": np.zeros((13, 2)), "coeff_count": np.zeros(13, np.uint64),
           "basis_index": np.zeros((455, 3), np.uint64), "sum_to_n": np.zeros(50, np.uint64)}
    check(L.hpsdf_tables_get(*[out[k].ctypes.data_as(C.c_void_p) for k in
                               ("roots", "weights", "normalised_lengths", "recurrence", "coeff_count", "basis_index",
                                "sum_to_n")]))
    return out


# ------------------------------------------------------------------------------------------------
class Config:
    """SDF::Config with the reference's field names and defaults (Config.cpp:5-14)."""

    def __init__(self):
        pod = PodConfig()
        check(lib().hpsdf_config_default(C.byref(pod)))
        self.nearnessWeighting_type = 0
        self.nearnessWeighting_strength = 0.0
        self.continuity_enforce = bool(pod.continuity_enforce)
        self.continuity_strength = pod.continuity_strength
        self.enableLogging = False
        self.targetErrorThreshold = pod.target_error_threshold
        self.threadCount = int(pod.thread_count)
        self.root_min = (-0.5, -0.5, -0.5)
        self.root_max = (0.5, 0.5, 0.5)

    def to_pod(self):
        pod = PodConfig()
        pod.weighting_type = self.nearnessWeighting_type
        pod.weighting_strength = self.nearnessWeighting_strength
        pod.continuity_enforce = 1 if self.continuity_enforce else 0
        pod.continuity_strength = self.continuity_strength
        pod.enable_logging = 1 if self.enableLogging else 0
        pod.target_error_threshold = self.targetErrorThreshold
        pod.thread_count = self.threadCount
        for a in range(3):
            pod.root_min[a] = self.root_min[a]
            pod.root_max[a] = self.root_max[a]
        return pod

    @staticmethod
    def from_pod(pod):
        c = Config.__new__(Config)
        c.nearnessWeighting_type = pod.weighting_type
        c.nearnessWeighting_strength = pod.weighting_strength
        c.continuity_enforce = bool(pod.continuity_enforce)
        c.continuity_strength = pod.continuity_strength
        c.enableLogging = bool(pod.enable_logging)
        c.targetErrorThreshold = pod.target_error_threshold
        c.threadCount = int(pod.thread_count)
        c.root_min = tuple(pod.root_min)
        c.root_max = tuple(pod.root_max)
        return c


FIT_EXACT, FIT_SPLIT, FIT_FAST = 0, 1, 2


def set_reduction_order(left_assoc):
    """Eigen's 3-vector reductions (prod / norm / normalize) as (a . b) . c (True) instead of a . (b . c) (False, the default): which one
    the reference computes depends on how its Eigen was built (include/hpsdf.h).  Process-wide; set it between calls, not during."""
    lib().hpsdf_set_reduction_order(1 if left_assoc else 0)


def reduction_order():
    return int(lib().hpsdf_get_reduction_order())


def set_mesh_face_rule(reference):
    """Mesh fields: True = the reference's closest-point routine to the letter (its face-case point whatever the weights,
    Utility.cpp:5-97); False (default) = a face-case point that has left its triangle is replaced by the boundary's closest point, so
    that every evaluation path gives one answer (include/hpsdf.h).  Process-wide; set it between calls, not during."""
    lib().hpsdf_set_mesh_face_rule(1 if reference else 0)


def mesh_face_rule():
    return int(lib().hpsdf_get_mesh_face_rule())


def make_config(target=1e-10, root_min=(-0.5, -0.5, -0.5), root_max=(0.5, 0.5, 0.5), threads=1, continuity=False):
    c = Config()
    c.targetErrorThreshold = target
    c.root_min, c.root_max = tuple(root_min), tuple(root_max)
    c.threadCount = threads
    c.continuity_enforce = continuity
    return c


BLOCK_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)
BLOCK_RELEASE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_bytes_new = C.pythonapi.PyBytes_FromStringAndSize  # (NULL, n): an uninitialised bytes object of n bytes, ours alone until returned
_bytes_new.restype, _bytes_new.argtypes = C.py_object, [C.c_void_p, C.c_ssize_t]
_bytes_ptr = C.pythonapi.PyBytes_AsString
_bytes_ptr.restype, _bytes_ptr.argtypes = C.c_void_p, [C.py_object]


class Context:
    """One GPU + one HIP stream.  stream: raw hipStream_t (int) or None for a library-owned one."""

    def __init__(self, device=0, stream=None):
        self.handle = C.c_void_p()
        check(lib().hpsdf_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self.handle)))
        self.device = device
        # Create writes its block straight into a bytes object (hpsdf_ctx_set_block_allocator): no second copy of the block
        self._blocks = {}

        def _alloc(size, _user, held=self._blocks):
            try:
                obj = _bytes_new(None, size)
                ptr = _bytes_ptr(obj)
                held[ptr] = obj
                return ptr
            except BaseException:  # noqa: BLE001 -- must not propagate through the C frames: NULL fails the build
                return None

        def _release(ptr, _user, held=self._blocks):
            held.pop(ptr, None)

        self._alloc_cb, self._release_cb = BLOCK_ALLOC(_alloc), BLOCK_RELEASE(_release)
        lib().hpsdf_ctx_set_block_allocator(self.handle, C.cast(self._alloc_cb, C.c_void_p), C.cast(self._release_cb, C.c_void_p), None)

    def _take_block(self, blk, size):
        """The bytes object behind a block pointer Create returned (or a copy of a malloc'd block)."""
        obj = self._blocks.pop(blk.value, None)
        if obj is not None and len(obj) == size:
            return obj
        data = C.string_at(blk, size)
        if obj is None:
            lib()._libc.free(blk)
        return data

    def set_stream(self, stream):
        check(lib().hpsdf_ctx_set_stream(self.handle, C.c_void_p(stream) if stream else None))

    def set_reduction_order(self, left_assoc):
        """This context's own reduction order (True / False), or None to follow the process-wide setting again (hpsdf.h)."""
        check(lib().hpsdf_ctx_set_reduction_order(self.handle, -1 if left_assoc is None else (1 if left_assoc else 0)))

    def reduction_order(self):
        v = C.c_int()
        check(lib().hpsdf_ctx_get_reduction_order(self.handle, C.byref(v)))
        return v.value

    def set_mesh_face_rule(self, reference):
        """This context's own mesh face rule (True = the reference's), or None to follow the process-wide rule again."""
        check(lib().hpsdf_ctx_set_mesh_face_rule(self.handle, -1 if reference is None else (1 if reference else 0)))

    def mesh_face_rule(self):
        v = C.c_int()
        check(lib().hpsdf_ctx_get_mesh_face_rule(self.handle, C.byref(v)))
        return v.value

    def set_build_limits(self, max_nodes=0, max_bytes=0):
        """Bounds on a Create (hpsdf_ctx_set_build_limits): 0 = the default (no bound on nodes; bytes: 1/64 of the free device
        memory, at least 1 GiB), None = no limit.  A build that crosses one raises HpsdfError with status ERR_BUILD_LIMIT."""
        none = (1 << 64) - 1
        check(lib().hpsdf_ctx_set_build_limits(self.handle, none if max_nodes is None else int(max_nodes), none if max_bytes is None else int(max_bytes)))

    def build_limits(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(lib().hpsdf_ctx_get_build_limits(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_fast_fit(self, on=True):
        """Every row of every fit of degree >= 4 on the matrix cores (FIT_FAST; errors then only agree to ~1e-15)."""
        check(lib().hpsdf_ctx_set_fast_fit(self.handle, 1 if on else 0))

    def set_fit_mode(self, mode):
        """FIT_EXACT (0): every row bit-exact, the canonical bytes; FIT_SPLIT (1, default): top-degree rows of from-scratch fits of
        degree >= split_min_degree (default 6) exact, the rows below them by sum factorisation from the same samples -- errors and topology
        canonical; FIT_FAST (2): every row of every fit of degree >= 4 on the matrix cores."""
        check(lib().hpsdf_ctx_set_fit_mode(self.handle, int(mode)))

    def set_split_min_degree(self, degree):
        """FIT_SPLIT splits from-scratch fits from this degree on (2..12, 12 = never; default 6: it pays from degree 5)."""
        check(lib().hpsdf_ctx_set_split_min_degree(self.handle, int(degree)))

    def fit_mode(self):
        m = C.c_int(0)
        check(lib().hpsdf_ctx_get_fit_mode(self.handle, C.byref(m)))
        return m.value

    def synchronize(self):
        check(lib().hpsdf_ctx_synchronize(self.handle))

    def close(self):
        if self.handle:
            lib().hpsdf_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Field:
    def __init__(self, handle, keep=(), kind=None):
        self.handle = handle
        self._keep = keep
        self.kind = kind  # "analytic" | "callback" | "mesh" | "tree_csg"

    @staticmethod
    def analytic(spec):
        """spec: list of (kind, op, params)."""
        arr = (Prim * len(spec))()
        for i, (kind, op, params) in enumerate(spec):
            arr[i].kind, arr[i].op = kind, op
            for j, v in enumerate(params):
                arr[i].p[j] = float(v)
        h = C.c_void_p()
        check(lib().hpsdf_field_create_analytic(arr, len(spec), C.byref(h)))
        return Field(h, kind="analytic")

    @staticmethod
    def sphere(centre=(0.25, 0.0, 0.0), radius=0.5):
        return Field.analytic([(PRIM_SPHERE, OP_UNION, list(centre) + [radius])])

    @staticmethod
    def union3():
        """BASELINE config[1]: union of sphere, box, torus."""
        return Field.analytic([
            (PRIM_SPHERE, OP_UNION, [-0.2, -0.15, 0.1, 0.18]),
            (PRIM_BOX, OP_UNION, [0.15, 0.2, -0.1, 0.12, 0.10, 0.15]),
            (PRIM_TORUS_Y, OP_UNION, [0.0, -0.2, -0.2, 0.15, 0.05]),
        ])

    @staticmethod
    def callback(fn):
        """fn(pt: (x,y,z), thread_idx) -> float, called from host threads."""
        def tramp(p, t, _u):
            return float(fn((p[0], p[1], p[2]), t))
        cb = CALLBACK(tramp)
        h = C.c_void_p()
        check(lib().hpsdf_field_create_callback(cb, None, C.byref(h)))
        return Field(h, keep=(cb, fn), kind="callback")

    @staticmethod
    def mesh(ctx, verts, tris):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, np.uint64).reshape(-1, 3)
        h = C.c_void_p()
        check(lib().hpsdf_field_create_mesh(ctx.handle, v.ctypes.data_as(C.c_void_p), len(v),
                                            t.ctypes.data_as(C.c_void_p), len(t), C.byref(h)))
        return Field(h, keep=(ctx,), kind="mesh")

    @staticmethod
    def tree_csg(tree, op, inner):
        h = C.c_void_p()
        check(lib().hpsdf_field_create_tree_csg(tree.handle, op, inner.handle, C.byref(h)))
        return Field(h, keep=(tree, inner), kind="tree_csg")

    def eval(self, ctx, pts):
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty(len(pts))
        check(lib().hpsdf_field_eval_host(ctx.handle, self.handle, pts.ctypes.data_as(C.c_void_p), len(pts),
                                          out.ctypes.data_as(C.c_void_p)))
        return out

    def eval_naive(self, ctx, pts):
        """Mesh fields: Mesh::SignedDistanceAtPt(pt) without the BVH -- the O(n) scan, on the GPU."""
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty(len(pts))
        check(lib().hpsdf_field_eval_naive_host(ctx.handle, self.handle, pts.ctypes.data_as(C.c_void_p), len(pts),
                                                out.ctypes.data_as(C.c_void_p)))
        return out

    def eval_wave(self, ctx, pts):
        """Mesh fields: the BVH query as Create's sampler runs it (64 consecutive points share one traversal)."""
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty(len(pts))
        check(lib().hpsdf_field_eval_wave_host(ctx.handle, self.handle, pts.ctypes.data_as(C.c_void_p), len(pts),
                                               out.ctypes.data_as(C.c_void_p)))
        return out

    def eval_lane(self, ctx, pts):
        """Mesh fields: the per-point stack traversal (what the fused mesh fit and a mesh under a tree-CSG wrapper run;
        eval() itself takes the shared traversal for plain mesh fields): same bits."""
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty(len(pts))
        check(lib().hpsdf_field_eval_lane_host(ctx.handle, self.handle, pts.ctypes.data_as(C.c_void_p), len(pts),
                                               out.ctypes.data_as(C.c_void_p)))
        return out

    def mesh_stats(self, reset=True):
        """BVH traversal counters (mesh fields created under HPSDF_MESH_STATS=1, diagnostic builds): wave queries, nodes
        visited, pairs through the lower-bound test ("tri_tests"), pairs through the closest-point test ("tri_test_lanes")."""
        out = (C.c_uint64 * 8)()
        check(lib().hpsdf_field_mesh_stats(self.handle, out, 1 if reset else 0))
        return dict(zip(("wave_queries", "node_visits", "tri_tests", "tri_test_lanes", "leaf_pairs", "bound_batches", "closest_batches",
                         "seed_exact"), (int(v) for v in out)))

    def release_host_copies(self):
        """Drops the host-side copies of a mesh field's arrays that calls of one or two points are answered from (they come back with
        the next such call)."""
        check(lib().hpsdf_field_release_host_copies(self.handle))

    def close(self):
        if self.handle:
            lib().hpsdf_field_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceTree:
    """A serialised octree resident in HBM (FromMemoryBlock on the device side)."""

    def __init__(self, ctx, block):
        self.ctx = ctx
        self.block = bytes(block)
        self.handle = C.c_void_p()
        check(lib().hpsdf_tree_upload(ctx.handle, self.block, len(self.block), C.byref(self.handle)))

    def info(self):
        nn, nc, nl = C.c_uint64(), C.c_uint64(), C.c_uint64()
        md, mdep = C.c_int(), C.c_int()
        check(lib().hpsdf_tree_info(self.handle, C.byref(nn), C.byref(nc), C.byref(nl), C.byref(md), C.byref(mdep)))
        return {"n_nodes": nn.value, "n_coeffs": nc.value, "n_leaves": nl.value, "max_degree": md.value,
                "max_depth": mdep.value}

    def query(self, pts):
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty(len(pts))
        check(lib().hpsdf_query_host(self.ctx.handle, self.handle, pts.ctypes.data_as(C.c_void_p), len(pts),
                                     out.ctypes.data_as(C.c_void_p)))
        return out

    def query_with_gradient(self, pts, grad_init=None):
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 3)
        out = np.empty(len(pts))
        grad = np.zeros((len(pts), 3)) if grad_init is None else np.array(grad_init, np.float64).reshape(-1, 3).copy()
        check(lib().hpsdf_query_gradient_host(self.ctx.handle, self.handle, pts.ctypes.data_as(C.c_void_p), len(pts),
                                              out.ctypes.data_as(C.c_void_p), grad.ctypes.data_as(C.c_void_p)))
        return out, grad

    def query_ray(self, origins, directions, t_max, t_init=None):
        """Octree::QueryRay per row -> (hit u8 [n], t f64 [n]); t rows of misses keep t_init."""
        o = np.ascontiguousarray(origins, np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float64).reshape(-1, 3)
        n = len(o)
        tm = np.ascontiguousarray(np.broadcast_to(np.asarray(t_max, np.float64), (n,)))
        hit = np.zeros(n, np.uint8)
        t = np.zeros(n) if t_init is None else np.array(t_init, np.float64).reshape(n).copy()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        check(lib().hpsdf_query_ray_host(self.ctx.handle, self.handle, vp(o), vp(d), vp(tm), n, vp(hit), vp(t)))
        return hit, t

    def function_slice(self, c, view_min, view_max, n_samples=2048):
        """Octree::OutputFunctionSlice up to the byte image -> (rgb u8 [n,n,3], values f64 [n,n])."""
        n = int(n_samples)
        rgb = np.zeros((n, n, 3), np.uint8)
        vals = np.zeros((n, n))
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
        check(lib().hpsdf_function_slice(self.ctx.handle, self.handle, float(c), f3(view_min), f3(view_max), n,
                                         rgb.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p)))
        return rgb, vals

    def query_device(self, d_xyz_ptr, n, d_out_ptr):
        """Raw device pointers (ints); asynchronous on the context stream."""
        check(lib().hpsdf_query_device(self.ctx.handle, self.handle, C.c_void_p(d_xyz_ptr), n, C.c_void_p(d_out_ptr)))

    def close(self):
        if self.handle:
            lib().hpsdf_tree_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Build:
    """Stepwise Create (one canonical round at a time); see include/hpsdf.h."""

    def __init__(self, config, K=0, rank=0, world=1):
        self.pod = config.to_pod() if isinstance(config, Config) else config
        self.opts = BuildOpts(K, rank, world)
        self.handle = C.c_void_p()
        self.rank, self.world = rank, world
        check(lib().hpsdf_build_begin(C.byref(self.pod), C.byref(self.opts), C.byref(self.handle)))

    def select(self):
        n = C.c_uint64()
        check(lib().hpsdf_build_round_select(self.handle, C.byref(n)))
        return n.value

    def jobs(self, n):
        arr = (Job * n)()
        check(lib().hpsdf_build_round_jobs(self.handle, arr))
        return arr

    def slice(self, rank=None):
        f, c = C.c_uint64(), C.c_uint64()
        check(lib().hpsdf_build_round_slice(self.handle, self.rank if rank is None else rank, C.byref(f), C.byref(c)))
        return f.value, c.value

    def max_slice(self):
        n = C.c_uint64()
        check(lib().hpsdf_build_round_max_slice(self.handle, C.byref(n)))
        return n.value

    def compute(self, ctx, field):
        check(lib().hpsdf_build_round_compute(self.handle, ctx.handle if ctx is not None else None, field.handle))

    def results_device(self):
        p, n = C.c_void_p(), C.c_uint64()
        check(lib().hpsdf_build_round_results_device(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def results_host(self, ctx):
        _, cnt = self.slice()
        out = np.zeros(cnt * JOB_HEADER_DOUBLES)
        check(lib().hpsdf_build_round_results_host(self.handle, ctx.handle, out.ctypes.data_as(C.c_void_p)))
        return out

    def apply(self, headers):
        h = np.ascontiguousarray(headers, np.float64)
        check(lib().hpsdf_build_round_apply(self.handle, h.ctypes.data_as(C.c_void_p)))

    def inject(self, job, p_coeffs, h_coeffs):
        p = np.ascontiguousarray(p_coeffs, np.float64) if p_coeffs is not None else None
        h = np.ascontiguousarray(h_coeffs, np.float64) if h_coeffs is not None else None
        check(lib().hpsdf_build_round_inject(self.handle, job, p.ctypes.data_as(C.c_void_p) if p is not None else None,
                                             h.ctypes.data_as(C.c_void_p) if h is not None else None))

    def rows_counts(self):
        """Weighted builds on N ranks: doubles every rank hands over after the round just applied (hpsdf.h)."""
        counts = (C.c_uint64 * self.world)()
        check(lib().hpsdf_build_rows_counts(self.handle, counts))
        return list(counts)

    def rows_pack_host(self, ctx, count):
        out = np.zeros(max(1, count))
        check(lib().hpsdf_build_rows_pack_host(self.handle, ctx.handle if ctx is not None else None, out.ctypes.data_as(C.c_void_p)))
        return out[:count]

    def rows_unpack_host(self, ctx, parts):
        keep = [np.ascontiguousarray(p, np.float64) if len(p) else np.zeros(1) for p in parts]
        arr = (C.c_void_p * len(keep))(*[k.ctypes.data_as(C.c_void_p).value for k in keep])
        check(lib().hpsdf_build_rows_unpack_host(self.handle, ctx.handle if ctx is not None else None, arr))

    def node_rows(self, ctx, node_idx):
        """The coefficient rows node ``node_idx`` holds right now (its full array in a weighted build), if on this rank."""
        n = C.c_uint64()
        check(lib().hpsdf_build_node_rows_host(self.handle, ctx.handle if ctx is not None else None, node_idx, None, C.byref(n)))
        out = np.zeros(max(1, n.value))
        check(lib().hpsdf_build_node_rows_host(self.handle, ctx.handle if ctx is not None else None, node_idx,
                                               out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out[:n.value]

    def layout(self):
        tot = C.c_uint64()
        counts = (C.c_uint64 * self.world)()
        check(lib().hpsdf_build_layout(self.handle, C.byref(tot), counts))
        return tot.value, list(counts)

    def pack_host(self, ctx, count):
        out = np.zeros(max(1, count))
        check(lib().hpsdf_build_pack_host(self.handle, ctx.handle if ctx is not None else None,
                                          out.ctypes.data_as(C.c_void_p)))
        return out[:count]

    def pack_device(self, ctx):
        p, n = C.c_void_p(), C.c_uint64()
        check(lib().hpsdf_build_pack_device(self.handle, ctx.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def assemble(self, packs):
        keep = [np.ascontiguousarray(p, np.float64) if len(p) else np.zeros(1) for p in packs]
        arr = (C.c_void_p * len(keep))(*[k.ctypes.data_as(C.c_void_p).value for k in keep])
        blk, sz = C.c_void_p(), C.c_size_t()
        check(lib().hpsdf_build_assemble(self.handle, arr, C.byref(blk), C.byref(sz)))
        data = C.string_at(blk, sz.value)
        lib()._libc.free(blk)
        return data

    def stats(self):
        st = BuildStats()
        check(lib().hpsdf_build_get_stats(self.handle, C.byref(st)))
        return st.as_dict()

    def close(self):
        if self.handle:
            lib().hpsdf_build_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_block(ctx, config, field, K=0):
    """Octree::Create on one GPU -> (serialised block bytes, stats dict)."""
    pod = config.to_pod() if isinstance(config, Config) else config
    blk, sz, st = C.c_void_p(), C.c_size_t(), BuildStats()
    rc = lib().hpsdf_create(ctx.handle if ctx is not None else None, C.byref(pod), field.handle, K, C.byref(blk), C.byref(sz), C.byref(st))
    if rc and ctx is not None:
        ctx._blocks.clear()  # (a failed build gave its blocks back; nothing of it is kept)
    check(rc)
    if ctx is None:  # (not reached: Create needs a device)
        data = C.string_at(blk, sz.value)
        lib()._libc.free(blk)
    else:
        data = ctx._take_block(blk, sz.value)
    return data, st.as_dict()


ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


def create_block_distributed(ctx, config, field, K, rank, world, gather):
    """hpsdf_create_distributed: this rank's part of a Create sharded over `world` GPUs, frontier on the device.
    gather(d_buf_ptr, bytes_per_rank, stream_ptr) -> None performs the in-place all-gather (rank r's part at
    d_buf_ptr + r * bytes_per_rank); exceptions it raises fail the build.  Returns (block bytes, stats)."""
    pod = config.to_pod() if hasattr(config, "to_pod") else config
    failure = []

    def _cb(_user, d_buf, nbytes, stream):
        try:
            gather(d_buf, nbytes, stream)
            return 0
        except BaseException as e:  # noqa: BLE001 -- must not propagate through the C frames
            failure.append(e)
            return 1

    cb = ALLGATHER(_cb)
    blk, size, st = C.c_void_p(), C.c_size_t(), BuildStats()
    rc = lib().hpsdf_create_distributed(ctx.handle, C.byref(pod), field.handle, K, rank, world, C.cast(cb, C.c_void_p), None,
                                        C.byref(blk), C.byref(size), C.byref(st))
    if failure or rc:
        ctx._blocks.clear()
    if failure:
        raise failure[0]
    check(rc)
    return ctx._take_block(blk, size.value), st.as_dict()


def continuity_post_process(block, tol=0.0, max_iter=0, threads=0, ctx=None):
    """Octree::PerformContinuityPostProcess (Octree.cpp:1717-1762) on a serialised block.  ctx=None: everything on
    the host; with a Context the conjugate-gradient loop runs on its device (bit-identical result).
    Returns (new block bytes, stats dict).  tol 0 = the reference's EPSILON_F32."""
    buf = C.create_string_buffer(bytes(block), len(block))
    st = ContinuityStats()
    if ctx is None:
        check(lib().hpsdf_continuity_post_process(buf, len(block), tol, max_iter, threads, C.byref(st)))
    else:
        check(lib().hpsdf_continuity_post_process_device(ctx.handle, buf, len(block), tol, max_iter, threads, C.byref(st)))
    return buf.raw, st.as_dict()


def continuity_matrix(block, threads=0):
    """(row_ptr, col, val, stats) of the jump-energy matrix M (no regularisation), CSR."""
    b = bytes(block)
    rp, ci, v = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_double)()
    st = ContinuityStats()
    check(lib().hpsdf_continuity_matrix(b, len(b), threads, C.byref(rp), C.byref(ci), C.byref(v), C.byref(st)))
    n = int(np.frombuffer(b[:8], np.uint64)[0])
    row_ptr = np.ctypeslib.as_array(rp, shape=(n + 1,)).copy()
    nnz = int(row_ptr[-1])
    col = np.ctypeslib.as_array(ci, shape=(max(nnz, 1),))[:nnz].copy()
    val = np.ctypeslib.as_array(v, shape=(max(nnz, 1),))[:nnz].copy()
    for p in (rp, ci, v):
        lib()._libc.free(C.cast(p, C.c_void_p))
    return row_ptr, col, val, st.as_dict()


def continuity_matrix_device(ctx, block):
    """The same matrix assembled on the device (what Create uses) and copied back: (row_ptr, col, val, stats)."""
    b = bytes(block)
    rp, ci, v = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_double)()
    st = ContinuityStats()
    check(lib().hpsdf_continuity_matrix_device(ctx.handle, b, len(b), C.byref(rp), C.byref(ci), C.byref(v), C.byref(st)))
    n = int(np.frombuffer(b[:8], np.uint64)[0])
    row_ptr = np.ctypeslib.as_array(rp, shape=(n + 1,)).copy()
    nnz = int(row_ptr[-1])
    col = np.ctypeslib.as_array(ci, shape=(max(nnz, 1),))[:nnz].copy()
    val = np.ctypeslib.as_array(v, shape=(max(nnz, 1),))[:nnz].copy()
    for p in (rp, ci, v):
        lib()._libc.free(C.cast(p, C.c_void_p))
    return row_ptr, col, val, st.as_dict()


def continuity_last_stats():
    st = ContinuityStats()
    check(lib().hpsdf_continuity_last_stats(C.byref(st)))
    return st.as_dict()


def selftest_acosf(ctx, first_bits, stride, n):
    """The device's acosf (the angle weights of vertex pseudo-normals) of the floats with bit patterns first_bits + i * stride."""
    out = np.empty(n, np.float32)
    check(lib().hpsdf_selftest_acosf(ctx.handle, first_bits, stride, n, out.ctypes.data_as(C.c_void_p)))
    return out


def bench_fit(ctx, config, field, degree, depth, n_cells, repeats=5):
    pod = config.to_pod()
    ms = C.c_double()
    check(lib().hpsdf_bench_fit(ctx.handle, C.byref(pod), field.handle, degree, depth, n_cells, repeats, C.byref(ms)))
    return ms.value


def fit_cells(ctx, config, field, degree, depth, n_cells):
    """From-scratch fits of `degree` for the first n_cells cells of the depth-`depth` lattice (x fastest) -> (coeffs[n, ncoef], errs[n])."""
    pod = config.to_pod()
    nc = int(NCOEF[degree])
    coeffs, errs = np.empty((n_cells, nc)), np.empty(n_cells)
    check(lib().hpsdf_fit_cells(ctx.handle, C.byref(pod), field.handle, degree, depth, n_cells, coeffs.ctypes.data_as(C.c_void_p),
                                errs.ctypes.data_as(C.c_void_p)))
    return coeffs, errs


class Octree:
    """SDF::Octree mirror: Create / Query / ToMemoryBlock / FromMemoryBlock / CSG rebuilds."""

    def __init__(self, device=0, stream=None, jobs_per_round=0):
        self._device, self._stream, self.K = device, stream, jobs_per_round
        self._ctx = None
        self._tree = None
        self.block = None
        self.config = None
        self.stats = None

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context(self._device, self._stream)
        return self._ctx

    def Create(self, config, F):
        """F: a Field, or a Python callable f(pt, threadIdx) like the reference's std::function."""
        field = F if isinstance(F, Field) else Field.callback(F)
        block, stats = create_block(self.ctx, config, field, self.K)
        self.Clear()
        self.block, self.stats, self.config = block, stats, config
        self._tree = DeviceTree(self.ctx, block)

    def _csg(self, op, F):
        if self._tree is None:
            raise HpsdfError(6, "CSG on an empty octree")
        inner = F if isinstance(F, Field) else Field.callback(F)
        field = Field.tree_csg(self._tree, op, inner)
        old = self._tree
        block, stats = create_block(self.ctx, self.config, field, self.K)
        field.close()
        old.close()
        self.block, self.stats = block, stats
        self._tree = DeviceTree(self.ctx, block)

    def UnionSDF(self, F):
        self._csg(OP_UNION, F)

    def SubtractSDF(self, F):
        self._csg(OP_SUBTRACT, F)

    def IntersectSDF(self, F):
        self._csg(OP_INTERSECT, F)

    def Clear(self):
        if self._tree is not None:
            self._tree.close()
        self._tree, self.block = None, None

    def FromMemoryBlock(self, block):
        if not block:
            raise HpsdfError(4, "empty MemoryBlock")
        if len(block) < 16 + C.sizeof(PodConfig):
            raise HpsdfError(4, "MemoryBlock too small")
        self.Clear()
        self.block = bytes(block)
        pod = PodConfig.from_buffer_copy(self.block[-C.sizeof(PodConfig):])
        self.config = Config.from_pod(pod)
        self._tree = DeviceTree(self.ctx, self.block)

    def ToMemoryBlock(self):
        return bytes(self.block) if self.block is not None else b""

    def Query(self, pts):
        """One point (3,) -> float, or (n,3) -> ndarray; DBL_MAX outside the root."""
        if self._tree is None:
            raise HpsdfError(6, "Query on an empty octree")
        a = np.asarray(pts, np.float64)
        out = self._tree.query(a)
        return float(out[0]) if a.ndim == 1 else out

    def QueryWithGradient(self, pts):
        """(n,3) -> (values, unit 'gradients' as the reference's central-difference shortcut computes them)."""
        if self._tree is None:
            raise HpsdfError(6, "Query on an empty octree")
        return self._tree.query_with_gradient(pts)

    def QueryRay(self, origins, directions, t_max):
        """Octree::QueryRay (Octree.h:75) for one ray -> (hit, t) or (n,3) arrays -> (hit[n], t[n])."""
        if self._tree is None:
            raise HpsdfError(6, "Query on an empty octree")
        o = np.asarray(origins, np.float64)
        hit, t = self._tree.query_ray(o, directions, t_max)
        return (bool(hit[0]), float(t[0])) if o.ndim == 1 else (hit, t)

    def OutputFunctionSlice(self, fname, c, view_min, view_max, n_samples=2048):
        """Octree::OutputFunctionSlice (Octree.h:83-86): writes <fname>.bmp (24-bit, bottom-up rows as
        stb_image_write lays them out) and returns the (n,n,3) RGB array."""
        if self._tree is None:
            raise HpsdfError(6, "Query on an empty octree")
        rgb, _ = self._tree.function_slice(c, view_min, view_max, n_samples)
        write_bmp(fname + ".bmp", rgb)
        return rgb

    def GetRootAABB(self):
        return self.config.root_min, self.config.root_max

    def copy(self):
        o = Octree(self._device, self._stream, self.K)
        if self.block is not None:
            o.FromMemoryBlock(self.block)
            o.stats = self.stats
        return o


def write_bmp(path, rgb):
    """24-bit uncompressed BMP of an (h, w, 3) RGB array: 14-byte file header, 40-byte BITMAPINFOHEADER, rows
    bottom-up, BGR, padded to 4 bytes -- the layout stbi_write_bmp produces for comp = 3."""
    import struct
    h, w, _ = rgb.shape
    pad = (-3 * w) % 4
    rows = np.zeros((h, 3 * w + pad), np.uint8)
    rows[:, :3 * w] = rgb[::-1, :, ::-1].reshape(h, 3 * w)
    with open(path, "wb") as fh:
        fh.write(struct.pack("<2sIHHI", b"BM", 14 + 40 + rows.size, 0, 0, 14 + 40))
        fh.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, 0, 0, 0, 0, 0))
        fh.write(rows.tobytes())


def load_obj(path):
    """Meshing::ObjParser::Load (Source/Meshing/ObjParser.cpp:11-164) through the native reader:
    (verts f32 [n,3], tris u64 [m,3]), zero-based."""
    v, t = C.POINTER(C.c_float)(), C.POINTER(C.c_uint64)()
    nv, nt = C.c_uint64(), C.c_uint64()
    check(lib().hpsdf_obj_load(os.fsencode(path), C.byref(v), C.byref(nv), C.byref(t), C.byref(nt)))
    verts = np.ctypeslib.as_array(v, shape=(nv.value, 3)).copy()
    tris = np.ctypeslib.as_array(t, shape=(nt.value, 3)).copy()
    lib()._libc.free(C.cast(v, C.c_void_p))
    lib()._libc.free(C.cast(t, C.c_void_p))
    return verts, tris
