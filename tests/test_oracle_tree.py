"""Oracle Create/Query/serialisation: schedule KATs from SURVEY section 7 H1 (reference numerics driven by
the canonical loop), committed block hashes, the reference's own end-to-end tolerance, Query edge cases."""
import hashlib

import numpy as np
import pytest

from helpers import oracle_field, query_points, edge_points, synthetic_block, sha

DBL_MAX = np.finfo(np.float64).max


def hist(pb, key):
    leaves = pb["degree"] != 13
    return {str(int(d)): int((pb[key][leaves] == d).sum()) for d in np.unique(pb[key][leaves])}


@pytest.mark.parametrize("name,fname,target,K", [
    ("sphere_1e-4_K4096", "sphere", 1e-4, 4096), ("union3_1e-5_K4096", "union3", 1e-5, 4096),
    ("sphere_1e-8_K1024", "sphere", 1e-8, 1024),
])
def test_schedule_kats_from_survey(O, golden, name, fname, target, K):
    want = golden["kats"]["schedule_plainK"][name]
    t = O.Tree.create(O.default_config(target), oracle_field(O, fname), K)
    pb = O.parse_block(t.to_block())
    assert pb["n_nodes"] == want["n_nodes"]
    assert hist(pb, "degree") == want["degree_hist"]


@pytest.mark.parametrize("K", [1024, 256])
def test_schedule_kat_union3_1e7(O, golden, K):
    want = golden["kats"]["schedule_plainK"]["union3_1e-7_K%d" % K]
    g = golden["blocks"]["A1_union3_1e-7_K%d" % K]
    assert g["n_nodes"] == want["n_nodes"] and g["degree_hist"] == want["degree_hist"] and g["depth_hist"] == want["depth_hist"]


@pytest.mark.parametrize("case", ["C1_sphere_1e-4", "C2_union3_1e-5", "A2_sphere_1e-8_K1024", "D1_sphere075_customroot_1e-6"])
def test_oracle_blocks_reproduce_committed_hashes(O, golden, case):
    g = golden["blocks"][case]
    cfg = O.default_config(g["target"], g["root_min"], g["root_max"])
    t = O.Tree.create(cfg, oracle_field(O, g["field"]), g["K"])
    blk = t.to_block()
    assert len(blk) == g["block_bytes"] and hashlib.sha256(blk).hexdigest() == g["block_sha256"]
    q = t.query(query_points(O, g["root_min"], g["root_max"]))
    assert sha(q) == g["query_sha256"] and int((q > 1e300).sum()) == g["query_n_outside"]
    # run-to-run deterministic, literal inner loop included
    if case == "C1_sphere_1e-4":
        assert O.Tree.create(cfg, oracle_field(O, g["field"]), g["K"], literal=True).to_block() == blk


def test_block_layout_and_roundtrip(O):
    t = O.Tree.create(O.default_config(1e-4), O.sphere_field(), 1024)
    blk = t.to_block()
    pb = O.parse_block(blk)
    # [u64 nCoeffs][f64..][u64 nNodes][56-byte nodes][80-byte config]   (Octree.cpp:424-456)
    assert len(blk) == 8 + 8 * pb["n_coeffs"] + 8 + 56 * pb["n_nodes"] + 80
    assert pb["n_nodes"] == 4681 and pb["n_coeffs"] == 40960
    assert pb["childIdx"][0] == 1 and pb["degree"][0] == 13 and pb["depth"][0] == 0
    assert np.array_equal(pb["aabb"][0], np.array([-0.5] * 3 + [0.5] * 3, np.float32))
    leaves = pb["degree"] != 13
    assert np.all(pb["childIdx"][leaves] == np.uint64(0xFFFFFFFFFFFFFFFF))
    # leaves packed depth-first: coeffsStart strictly increasing in DFS order = 10 apart here
    assert sorted(pb["coeffsStart"][leaves].tolist()) == list(range(0, 40960, 10))
    t2 = O.Tree.from_block(blk)
    assert t2.to_block() == blk
    p = O.splitmix64_points(2000)
    assert np.array_equal(t.query(p), t2.query(p))


def test_reference_end_to_end_tolerance(O):
    """Source/Tests/HPUnitTests.cpp:46-77: |Query - true| <= 0.01 for the sphere at target 1e-8
    (weighting None here; the reference test uses Polynomial weighting, which only relaxes the stop rule)."""
    f = O.sphere_field()
    t = O.Tree.create(O.default_config(1e-8), f, 1024)
    p = O.splitmix64_points(100000, seed=99)
    assert np.abs(t.query(p) - f.eval(p)).max() <= 0.01


def test_custom_domain_tolerance(O, golden):
    """Source/Tests/HPUnitTests.cpp:285-316: root [-0.25,5]^3, sphere r 0.75."""
    g = golden["blocks"]["D1_sphere075_customroot_1e-6"]
    f = oracle_field(O, "sphere075")
    t = O.Tree.create(O.default_config(g["target"], g["root_min"], g["root_max"]), f, g["K"])
    p = (O.splitmix64_points(50000, seed=5) + 0.5) * 5.25 - 0.25
    assert np.abs(t.query(p) - f.eval(p)).max() <= 0.01


def test_query_edge_cases(O):
    t = O.Tree.create(O.default_config(1e-4), O.sphere_field(), 1024)
    q = t.query(edge_points(np.random.default_rng(3)))
    p = edge_points(np.random.default_rng(3))
    inside = np.all(np.abs(p.astype(np.float32)) <= np.float32(0.5), axis=1)
    assert np.all(q[~inside] == DBL_MAX) and np.all(q[inside] < 10.0)
    # the containment test is on the f32 cast (Octree.cpp:668): 0.5 + 1e-9 is inside, nextafter_f32(0.5) is not
    assert t.query(np.array([[0.0, 0.0, 0.5 + 1e-9]]))[0] < 10.0
    assert t.query(np.array([[0.50000006, 0.0, 0.0]]))[0] == DBL_MAX
    assert t.query(np.array([[np.nan, 0.0, 0.0]]))[0] == DBL_MAX
    # a point on a mid-plane belongs to the upper child (>=, Octree.cpp:681-683)
    eps = 1e-12
    a, b = t.query(np.array([[0.0, 0.1, 0.1]]))[0], t.query(np.array([[eps, 0.1, 0.1]]))[0]
    c = t.query(np.array([[-eps, 0.1, 0.1]]))[0]
    assert abs(a - b) < 1e-9 and abs(a - c) < 1e-2


def test_synthetic_block_all_degrees(O):
    rng = np.random.default_rng(11)
    for degs in ([0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11, 12, 6, 7, 2]):
        blk = synthetic_block(rng, degs, depth=2)
        t = O.Tree.from_block(blk)
        assert t.to_block() == blk
        assert np.all(np.isfinite(t.query(rng.uniform(-0.5, 0.5, (500, 3)))))


def test_from_block_rejects_garbage(O):
    with pytest.raises(ValueError):
        O.Tree.from_block(b"\x00" * 10)
    with pytest.raises(ValueError):
        O.Tree.from_block(np.array([10 ** 9], np.uint64).tobytes() + b"\x00" * 200)


@pytest.mark.parametrize("wtype", [1, 2])
def test_weighted_build_is_deterministic_and_accurate(O, wtype):
    """Nearness weighting (Octree.cpp:1209-1247) with the hashed stand-in for std::rand: the weight only scales the
    error estimate, so the reference test's accuracy bar (HPUnitTests.cpp:46-77) still has to hold."""
    cfg = O.default_config(1e-8)
    cfg.weighting_type, cfg.weighting_strength = wtype, 3.0
    a = O.Tree.create(cfg, O.sphere_field(), 1024)
    b = O.Tree.create(cfg, O.sphere_field(), 1024)
    assert a.to_block() == b.to_block()
    p = O.splitmix64_points(50000, seed=2)
    assert np.abs(a.query(p) - O.sphere_field().eval(p)).max() <= 0.01
    L = O.lib()
    import ctypes as C
    L.ora_weight_from_mean.restype = C.c_double
    L.ora_weight_from_mean.argtypes = [C.c_int, C.c_double, C.c_double]
    # :1224-1226 clamp to [0,1]; :1246 exp(-s m / sqrt 3)
    assert L.ora_weight_from_mean(1, 3.0, 0.0) == 1.0 and 0.0 < L.ora_weight_from_mean(1, 3.0, 0.5) < 1.0
    assert abs(L.ora_weight_from_mean(2, 3.0, 0.5) - np.exp(-1.5 / np.sqrt(3.0))) < 1e-15


def test_query_with_gradient_oracle(O):
    """Octree.cpp:749-789, 904-985: same value as Query; unit-length output aligned with the true normal."""
    t = O.Tree.create(O.default_config(1e-8), O.sphere_field(), 1024)
    p = O.splitmix64_points(5000, seed=3) * 0.8
    v, g = t.query_with_gradient(p)
    assert np.array_equal(v, t.query(p))
    true = p - np.array([0.25, 0.0, 0.0])
    true /= np.linalg.norm(true, axis=1, keepdims=True)
    cos = (g * true).sum(1)
    assert np.allclose(np.linalg.norm(g, axis=1), 1.0, atol=1e-12) and cos.mean() > 0.999 and cos.min() > 0.9
    v2, g2 = t.query_with_gradient(np.array([[3.0, 0.0, 0.0]]), np.array([[5.0, 6.0, 7.0]]))
    assert v2[0] == DBL_MAX and g2.tolist() == [[5.0, 6.0, 7.0]]


def test_oracle_threads_do_not_change_results(O):
    """bench.py's all-cores CPU baseline: a round's jobs on several pthreads (ora_create_mt) and Query cut into contiguous
    parts (ora_query_batch_mt) give the bytes of the single-threaded loops, for an analytic and for a weighted build."""
    cfg = O.default_config(1e-6)
    a = O.Tree.create(cfg, O.union3_field(), 256)
    b = O.Tree.create(cfg, O.union3_field(), 256, threads=5)
    assert a.to_block() == b.to_block() and a.stats == b.stats
    w = O.default_config(1e-6)
    w.weighting_type, w.weighting_strength = 2, 3.0
    assert O.Tree.create(w, O.sphere_field(), 512).to_block() == O.Tree.create(w, O.sphere_field(), 512, threads=3).to_block()
    p = O.splitmix64_points(30001, seed=9)
    p[:7] *= 3.0
    assert np.array_equal(a.query(p), a.query(p, threads=4)) and np.array_equal(a.query(p[:3]), a.query(p[:3], threads=8))
