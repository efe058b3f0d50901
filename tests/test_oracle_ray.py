"""CPU checks of the query-side helpers of the oracle (QueryRay, OutputFunctionSlice).
The reference has no test of either (QueryRay is marked untested in Include/HP/Octree.h:73-75;
OutputFunctionSlice needs stb): these pin the restatement to its own statement-level semantics."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def tree(O):
    return O.Tree.create(O.default_config(1e-4), O.sphere_field(), 1024)


def ray_reference_python(tree, o, d, t_max):
    """Octree.cpp:705-746 written out again in Python (origin inside the root only)."""
    d_ = 0.0
    for _ in range(200):
        v = tree.query(np.array([o + d_ * d]))[0]
        if v < 0.0001:
            return True, v
        d_ += v * 0.95 + 0.0001
        if d_ > t_max:
            return False, None
    return False, None


def test_query_ray_inside_origin_matches_step_loop(O, tree):
    rng = np.random.default_rng(5)
    o = rng.uniform(-0.45, 0.45, (200, 3))
    d = rng.standard_normal((200, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hit, t = tree.query_ray(o, d, 2.0, t_init=np.full(200, -7.0))
    for i in range(200):
        h, v = ray_reference_python(tree, o[i], d[i], 2.0)
        assert bool(hit[i]) == h
        if h:
            assert t[i] == v and v < 1e-4
        else:
            assert t[i] == -7.0  # untouched on a miss


def test_query_ray_hits_the_sphere_from_inside_the_root(O, tree):
    # from the root centre towards +x the sphere surface (centre 0.25, r 0.5) is not reached inside the root:
    # the origin is inside the sphere (negative distance) -> immediate hit with the field value
    hit, t = tree.query_ray([[0.0, 0.0, 0.0]], [[1.0, 0.0, 0.0]], 5.0)
    assert hit[0] == 1 and abs(t[0] - (-0.25)) < 5e-3
    # a ray starting outside the sphere, inside the root, marching towards it
    hit, t = tree.query_ray([[-0.45, 0.0, 0.0]], [[1.0, 0.0, 0.0]], 5.0)
    assert hit[0] == 1 and 0.0 <= t[0] < 1e-4
    # same origin, marching away: leaves the root (Query -> DBL_MAX) -> miss
    hit, t = tree.query_ray([[-0.45, 0.0, 0.0]], [[-1.0, 0.0, 0.0]], 5.0, t_init=[3.0])
    assert hit[0] == 0 and t[0] == 3.0


def test_query_ray_outside_origin_follows_the_reference_quirk(O, tree):
    # origin outside the root: a ray that misses the box returns False ...
    hit, _ = tree.query_ray([[2.0, 2.0, 2.0]], [[1.0, 0.0, 0.0]], 10.0)
    assert hit[0] == 0
    # ... and one that crosses it marches from the slab-parameter vector IntersectAABB leaves behind
    # (Octree.cpp:717): for origin (-2,0,0), dir +x that vector is (1.5, -inf, -inf) -> Query outside -> miss
    hit, _ = tree.query_ray([[-2.0, 0.0, 0.0]], [[1.0, 0.0, 0.0]], 10.0)
    assert hit[0] == 0


def test_function_slice_bytes(O, tree):
    n = 64
    rgb, vals = tree.function_slice(0.0, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), n)
    step = np.float32(1.0) / np.float32(n)
    xs = -0.5 + (np.arange(n, dtype=np.float32) * step).astype(np.float64)
    gx, gy = np.meshgrid(xs, xs, indexing="xy")
    pts = np.stack([gx, gy, np.zeros((n, n))], axis=-1).reshape(-1, 3)
    want = tree.query(pts).reshape(n, n)   # row i = y index, column j = x index
    assert np.array_equal(vals, want)
    pos = vals > 1e-6
    assert rgb[..., 0].max() == 0
    assert np.all(rgb[pos][:, 2] == 0) and np.all(rgb[~pos][:, 1] == 0)
    # the normalisation: the smallest positive value maps to 255 green, the largest to 0
    u = vals.astype(np.float32).astype(np.float64)
    g = (255 * (u - vals[pos].max()) / (vals[pos].min() - vals[pos].max()))
    assert np.array_equal(rgb[..., 1][pos], np.trunc(g[pos]).astype(np.int64) & 0xFF)
    b = (255 * (u - vals[~pos].min()) / (vals[~pos].max() - vals[~pos].min()))
    assert np.array_equal(rgb[..., 2][~pos], np.trunc(b[~pos]).astype(np.int64) & 0xFF)
