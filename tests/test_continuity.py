"""Continuity post-process (Octree.cpp:1250-1762): the oracle restatement against analytic
properties, and the product's host implementation (libhpsdf.so, no GPU involved) against the oracle.

The reference pins this path only end to end (Source/Tests/HPUnitTests.cpp:80-112, 285-316: |Query - true| <= 1e-2
with continuity on); its solver is Eigen's CG + IncompleteCholesky (unpinned, absent).  Oracle and product both
solve the same system to a relative residual; the tests compare them at a tolerance far below the north_star's
1e-6 by tightening the solver tolerance."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import oracle_field

CASES = {
    "sphere_uniform": ("sphere", 1e-6, (-0.5,) * 3, (0.5,) * 3, 1024),       # analytic integrals only
    "union3_adaptive": ("union3", 1e-7, (-0.5,) * 3, (0.5,) * 3, 1024),     # depths 4-6, degrees 2-4: numeric ones too
    "custom_root": ("sphere075", 1e-6, (-0.25,) * 3, (5.0,) * 3, 1024),     # HPUnitTests.cpp:285-316
}


@pytest.fixture(scope="module")
def trees(O):
    out = {}
    for name, (field, target, rmin, rmax, K) in CASES.items():
        cfg = O.default_config(target, rmin, rmax, continuity=True)
        cfg.continuity_strength = 8.0
        out[name] = O.Tree.create(cfg, oracle_field(O, field), K).to_block()
    return out


def csr(rp, ci, v):
    n = len(rp) - 1
    return sp.csr_matrix((v, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n))


def test_oracle_matrix_is_symmetric_psd_jump_energy(O, trees):
    t = O.Tree.from_block(trees["union3_adaptive"])
    rp, ci, v, st = t.continuity_matrix()
    M = csr(rp, ci, v)
    assert st["n_pairs_numeric"] > 1000 and st["n_pairs_analytic"] > 1000
    assert abs(M - M.T).max() <= 1e-9 * abs(M).max()
    rng = np.random.default_rng(2)
    for _ in range(5):  # x^T M x is a sum of squared face jumps (up to the reference's dropped |v| <= 1e-6 entries)
        x = rng.standard_normal(M.shape[0])
        assert x @ (M @ x) > -1e-3
    # the uniform tree: a constant field has no jump.  Leaf coefficient of the constant c is c * sqrt(|cell|),
    # other rows zero -> M applied to it vanishes
    t2 = O.Tree.from_block(trees["sphere_uniform"])
    rp, ci, v, _ = t2.continuity_matrix()
    M2 = csr(rp, ci, v)
    pb = O.parse_block(trees["sphere_uniform"])
    x = np.zeros(M2.shape[0])
    leaf = pb["degree"] != 13
    x[pb["coeffsStart"][leaf]] = (0.5 ** pb["depth"][leaf].astype(float)) ** 1.5
    assert np.abs(M2 @ x).max() < 1e-9


@pytest.mark.parametrize("name", list(CASES))
def test_product_matrix_equals_oracle_matrix(O, H, trees, name):
    blk = trees[name]
    rp, ci, v, st = O.Tree.from_block(blk).continuity_matrix()
    rp2, ci2, v2, st2 = H.continuity_matrix(blk, 4)
    for k in ("n_pairs", "n_pairs_analytic", "n_pairs_numeric"):
        assert st[k] == st2[k]
    Mo, Mp = csr(rp, ci, v), csr(rp2, ci2, v2)
    scale = abs(Mo).max()
    diff = abs(Mo - Mp)
    # the product evaluates the non-conforming face integrals as products of 1-D quadratures: rounding-level
    # differences, plus possibly entries on either side of the reference's |v| > 1e-6 keep/drop threshold
    assert diff.max() <= 2e-6
    assert (diff > 1e-12 * scale).sum() <= 8
    if st["n_pairs_numeric"] == 0:
        assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2) and np.array_equal(v, v2)  # same arithmetic


@pytest.mark.parametrize("name", list(CASES))
def test_product_solution_equals_oracle_solution(O, H, trees, name):
    blk = trees[name]
    t = O.Tree.from_block(blk)
    so = t.continuity_post_process(1e-11)
    want = O.parse_block(t.to_block())["coeffs"]
    got_blk, sp_ = H.continuity_post_process(blk, 1e-11, 0, 4)
    got = O.parse_block(got_blk)["coeffs"]
    assert sp_["residual"] < 1e-10 and so["residual"] < 1e-10
    assert np.abs(got - want).max() <= 1e-9          # north_star bar: 1e-6
    assert got_blk[8 + 8 * len(got):] == blk[8 + 8 * len(got):]  # nodes and config untouched
    assert sp_["jump_after"] < 0.5 * sp_["jump_before"]
    # at the reference's own tolerance both stay within the north_star's 1e-6 of the converged solution
    t1 = O.Tree.from_block(blk)
    t1.continuity_post_process(1e-6)
    loose_o = O.parse_block(t1.to_block())["coeffs"]
    loose_p = O.parse_block(H.continuity_post_process(blk)[0])["coeffs"]
    assert np.abs(loose_o - want).max() <= 1e-6 and np.abs(loose_p - want).max() <= 1e-6


def test_product_is_thread_count_independent(H, trees):
    blk = trees["union3_adaptive"]
    a, _ = H.continuity_post_process(blk, 0.0, 0, 1)
    b, _ = H.continuity_post_process(blk, 0.0, 0, 3)
    c, _ = H.continuity_post_process(blk, 0.0, 0, 8)
    assert a == b == c


def test_end_to_end_accuracy_as_the_reference_tests(O, H):
    """HPUnitTests.cpp:80-112 (sphere r 0.5 at (0.25,0,0)) and :285-316 (root [-0.25,5]^3, r 0.75): target 1e-8,
    continuity on with strength 8, |Query - true| <= 1e-2 -- for the oracle's and for the product's post-process."""
    for field, radius, rmin, rmax in (("sphere", 0.5, (-0.5,) * 3, (0.5,) * 3), ("sphere075", 0.75, (-0.25,) * 3, (5.0,) * 3)):
        cfg = O.default_config(1e-8, rmin, rmax, continuity=True)
        cfg.continuity_strength = 8.0
        t = O.Tree.create(cfg, oracle_field(O, field), 1024)
        blk = t.to_block()
        t.continuity_post_process(1e-6)
        pts = (O.splitmix64_points(200000, seed=4) + 0.5) * (np.array(rmax) - np.array(rmin)) + np.array(rmin)
        true = np.linalg.norm(pts - np.array((0.25, 0, 0)), axis=1) - radius
        assert np.abs(t.query(pts) - true).max() <= 1e-2
        t2 = O.Tree.from_block(H.continuity_post_process(blk)[0])
        assert np.abs(t2.query(pts) - true).max() <= 1e-2
        # two CG runs stopped at the reference's relative 1e-6 agree to about that, relative to the field's size
        assert np.abs(t2.query(pts) - t.query(pts)).max() <= 1e-5 * np.abs(true).max()


def test_bad_blocks_are_rejected(H, trees):
    blk = trees["sphere_uniform"]
    with pytest.raises(H.HpsdfError):
        H.continuity_post_process(blk[:-8])
    bad = bytearray(blk)
    bad[-80 + 24:-80 + 32] = np.array([0.0]).tobytes()  # continuity.strength = 0 (Config.cpp:28-31 asserts > 0)
    with pytest.raises(H.HpsdfError):
        H.continuity_post_process(bytes(bad))


def test_malformed_blocks_return_bad_block_not_crash(H, trees):
    """Deserialisation of untrusted bytes: child indices that wrap or cycle, coefficient ranges that wrap,
    overlap or leave the store, a 1-node interior root -- every one is HPSDF_ERR_BAD_BLOCK, never a fault."""
    blk = trees["sphere_uniform"]
    nc = int(np.frombuffer(blk[:8], np.uint64)[0])
    nn = int(np.frombuffer(blk[8 + 8 * nc:16 + 8 * nc], np.uint64)[0])
    base = 16 + 8 * nc

    def edit(node, off, value, dtype=np.uint64):
        b = bytearray(blk)
        raw = np.array([value], dtype).tobytes()
        b[base + 56 * node + off:base + 56 * node + off + len(raw)] = raw
        return bytes(b)

    nodes = np.frombuffer(blk[base:base + 56 * nn], np.uint8).reshape(nn, 56)
    child = nodes[:, :8].copy().view(np.uint64).ravel()
    leaf = int(np.nonzero(child == np.uint64(0xFFFFFFFFFFFFFFFF))[0][0])
    leaf2 = int(np.nonzero(child == np.uint64(0xFFFFFFFFFFFFFFFF))[0][1])
    start1 = int(nodes[leaf, 32:40].copy().view(np.uint64)[0])
    cases = {
        "child index 1<<40": edit(0, 0, 1 << 40),
        "children run past the end": edit(0, 0, nn - 3),
        "child is the root": edit(1, 0, 0),
        "child is an ancestor": edit(int(child[1]), 0, 1),
        "coefficient start wraps": edit(leaf, 32, 0xFFFFFFFFFFFFFFFC),
        "coefficient start past the store": edit(leaf, 32, nc - 1),
        "leaves share coefficients": edit(leaf2, 32, start1),
        "leaf degree 200": edit(leaf, 40, 200, np.uint8),
        "leaf depth lies": edit(leaf, 48, 1, np.uint8),
    }
    tiny = bytearray(16 + 56 + 80)
    tiny[8:16] = np.array([1], np.uint64).tobytes()
    tiny[16:24] = np.array([1 << 40], np.uint64).tobytes()
    tiny[16 + 40] = 13
    cases["1-node interior root"] = bytes(tiny) [:16 + 56] + blk[-80:]
    for name, b in cases.items():
        with pytest.raises(H.HpsdfError) as e:
            H.continuity_post_process(b)
        assert e.value.status == H.ERR_BAD_BLOCK, name
