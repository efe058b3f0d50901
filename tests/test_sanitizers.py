"""The host-side native code of libhpsdf.so under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU
sanitizers are not available on the pool).  tests/native/sanitizer_harness.cpp drives the continuity post-process
(1/3/8 threads, truncated block), the OBJ reader (good and malformed files), the mesh preparation (closed and open
mesh), the round scheduler (two simulated ranks through the injection hook) and the scalar-call path of Query /
QueryWithGradient (csrc/host_query.cpp: 5 000 points incl. the edge cases, compared with the oracle bit for bit) in one
binary built with g++."""
import os
import subprocess

import numpy as np

from conftest import ROOT
from helpers import icosphere, oracle_field

CSRC = os.path.join(ROOT, "hp-adaptive-signed-distance-field-octree_amd", "csrc")


def test_host_native_code_is_clean_under_asan_ubsan(O, tmp_path):
    exe = str(tmp_path / "harness")
    srcs = [os.path.join(ROOT, "tests", "native", "sanitizer_harness.cpp")] + \
           [os.path.join(CSRC, f) for f in ("continuity.cpp", "tables.cpp", "obj.cpp", "mesh.cpp", "builder.cpp", "host_query.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=all",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", CSRC, "-I", os.path.join(ROOT, "include")] + srcs + \
          ["-o", exe, "-pthread", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    cfg = O.default_config(1e-7, continuity=True)
    blk = O.Tree.create(cfg, oracle_field(O, "union3"), 1024).to_block()   # depths 4-6: analytic and numeric face integrals
    (tmp_path / "blk.bin").write_bytes(blk)
    v, t = icosphere(3, 0.35)
    with open(tmp_path / "ico.obj", "w") as fh:
        for p in v:
            fh.write("v %.9g %.9g %.9g\n" % tuple(p))
        fh.write("vt 0 0\nvn 0 0 1\n")
        for a, b, c in t:
            fh.write("f %d/1/1 %d/1/1 %d/1/1\n" % (a + 1, b + 1, c + 1))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    # the scalar-call path of Query / QueryWithGradient (csrc/host_query.cpp) on the same block: edge points, outside points, NaN
    from helpers import edge_points
    pts = np.concatenate([O.splitmix64_points(3000, seed=4), edge_points(np.random.default_rng(2), 2000)])
    pts.astype(np.float64).tofile(tmp_path / "pts.bin")
    rng = np.random.default_rng(11)
    nr = 3000
    ro = rng.uniform(-0.6, 0.6, (nr, 3))
    ro[:400] = rng.uniform(-3.0, 3.0, (400, 3))          # far outside: Ray::IntersectAABB
    rd = rng.standard_normal((nr, 3))
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd[400:416, 1] = 0.0                                  # axis-parallel components: infinite slab parameters
    rd[416:432] = [1.0, 0.0, 0.0]
    rt = rng.uniform(0.05, 3.0, nr)
    np.concatenate([ro, rd, rt[:, None]], axis=1).astype(np.float64).tofile(tmp_path / "rays.bin")
    r = subprocess.run([exe, str(tmp_path / "blk.bin"), str(tmp_path / "ico.obj"), str(tmp_path / "bad.obj"), str(tmp_path / "pts.bin"),
                        str(tmp_path / "res.bin"), str(tmp_path / "rays.bin"), str(tmp_path / "rays_out.bin")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    res = np.fromfile(tmp_path / "res.bin", np.float64)
    n = len(pts)
    otree = O.Tree.from_block(blk)
    wv, wg = otree.query_with_gradient(pts)
    assert np.array_equal(res[:n].view(np.uint64), otree.query(pts).view(np.uint64))      # Octree::Query, bit for bit
    inside = wv < 1e300
    got_g = res[n:].reshape(n, 3)
    assert np.array_equal(got_g[inside].view(np.uint64), wg[inside].view(np.uint64)) and np.all(got_g[~inside] == 7.0)
    # Octree::QueryRay on the calling thread (hostQueryRay), bit for bit the oracle's
    rr = np.fromfile(tmp_path / "rays_out.bin", np.float64).reshape(nr, 2)
    whit, wt = otree.query_ray(ro, rd, rt, t_init=np.full(nr, -123.0))
    assert np.array_equal(rr[:, 0] != 0.0, whit != 0) and np.array_equal(rr[:, 1].view(np.uint64), wt.view(np.uint64)) and 0 < whit.sum() < nr
    # the other reading of Eigen's normalize() (hpsdf_set_reduction_order(1) / ora_set_reduction_order(1)): same bits again
    r = subprocess.run([exe, str(tmp_path / "blk.bin"), str(tmp_path / "ico.obj"), str(tmp_path / "bad.obj"), str(tmp_path / "pts.bin"),
                        str(tmp_path / "res_left.bin")], capture_output=True, text=True, timeout=600, env=dict(env, HPSDF_REDUCTION_ORDER="left"))
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-4000:]
    O.set_reduction_order(1)
    try:
        _, wgl = otree.query_with_gradient(pts)
    finally:
        O.set_reduction_order(0)
    got_l = np.fromfile(tmp_path / "res_left.bin", np.float64)[n:].reshape(n, 3)
    assert np.array_equal(got_l[inside].view(np.uint64), wgl[inside].view(np.uint64))
    assert not np.array_equal(wgl[inside].view(np.uint64), wg[inside].view(np.uint64))  # the switch is not a no-op on these points


def test_host_threads_are_clean_under_tsan(O, tmp_path):
    """The same harness under ThreadSanitizer: the continuity post-process on 1, 3 and 8 threads (pair enumeration, assembly, the CG's
    chunked dot products), the host round scheduler for two ranks, mesh preparation -- no data race reported."""
    exe = str(tmp_path / "harness_tsan")
    srcs = [os.path.join(ROOT, "tests", "native", "sanitizer_harness.cpp")] + \
           [os.path.join(CSRC, f) for f in ("continuity.cpp", "tables.cpp", "obj.cpp", "mesh.cpp", "builder.cpp", "host_query.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", CSRC, "-I", os.path.join(ROOT, "include")] + srcs + \
          ["-o", exe, "-pthread", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    cfg = O.default_config(1e-6, continuity=True)
    (tmp_path / "blk.bin").write_bytes(O.Tree.create(cfg, oracle_field(O, "union3"), 1024).to_block())
    v, t = icosphere(2, 0.35)
    with open(tmp_path / "ico.obj", "w") as fh:
        for p in v:
            fh.write("v %.9g %.9g %.9g\n" % tuple(p))
        for a, b, c in t:
            fh.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0", HPSDF_HARNESS_QUICK="1")
    r = subprocess.run([exe, str(tmp_path / "blk.bin"), str(tmp_path / "ico.obj"), str(tmp_path / "bad.obj")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-4000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
