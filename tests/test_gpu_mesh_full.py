"""GPU tests at the full size of BASELINE configs[2..4]: the 2 097 152-triangle displaced torus (SURVEY 8(d)'s stand-in for the
2 M-triangle mesh of the north_star; dragon.obj / Ramesses.obj are not in the mount) at targetError 1e-6, root = mesh box.
The small-mesh tests elsewhere compare with the oracle; here the oracle cannot follow (it scans every triangle for every
sample), so the checks are the ones that hold at any size: the hierarchy against the O(n) scan, the sharded build against the
single-rank build, the continuity build against the post-process of the plain build, and Query against the field."""
import numpy as np
import pytest

from conftest import bits
from helpers import displaced_torus
from test_gpu_configs import _create_on_simulated_ranks

pytestmark = pytest.mark.gpu
K = 1024


@pytest.fixture(scope="module")
def torus():
    verts, tris = displaced_torus()
    assert len(tris) == 2_097_152
    lo, hi = verts.min(0) - 0.02, verts.max(0) + 0.02
    return verts, tris, tuple(lo), tuple(hi)


@pytest.fixture(scope="module")
def torus_build(H, ctx, torus):
    verts, tris, lo, hi = torus
    f = H.Field.mesh(ctx, verts, tris)
    blk, st = H.create_block(ctx, H.make_config(1e-6, lo, hi), f, K)
    return f, blk, st


def _probe_points(verts, tris, lo, hi, n, seed):
    """Random points of the root box, points just off the surface (a few triangle sizes and a few f32 ulps away), points on
    vertices and face centres, and points on the torus' centre circle, where a whole ring of triangles is nearly equidistant."""
    rng = np.random.default_rng(seed)
    lo, hi = np.array(lo), np.array(hi)
    box = rng.uniform(lo, hi, (n // 2, 3))
    t = rng.integers(0, len(tris), n // 4)
    a, b, c = (verts[tris[t, k].astype(np.int64)].astype(np.float64) for k in range(3))
    w = rng.dirichlet((1.0, 1.0, 1.0), len(t))
    on = w[:, :1] * a + w[:, 1:2] * b + w[:, 2:] * c
    nrm = np.cross(b - a, c - a)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    off = on + nrm * (10.0 ** rng.uniform(-7, -2, (len(t), 1))) * rng.choice([-1.0, 1.0], (len(t), 1))
    u = rng.uniform(0, 2 * np.pi, n // 8)
    ring = np.stack([0.3 * np.cos(u), 0.3 * np.sin(u), np.zeros_like(u)], -1) + rng.normal(0, 1e-3, (len(u), 3))
    return np.concatenate([box, off, a[: n // 16], on[: n // 16], ring])


def test_full_size_hierarchy_equals_linear_scan_bitwise(H, ctx, torus, torus_build):
    """TestBVHQuerying (MeshingUnitTests.cpp:110-138) on 2 M triangles: the per-point traversal and the sampler's traversal
    (box + slab bounds, lower-bound filter, pooled sparse subtrees) return the bits of the O(n) scan of Mesh.cpp:134-159."""
    verts, tris, lo, hi = torus
    f = torus_build[0]
    pts = _probe_points(verts, tris, lo, hi, 2048, 11)
    want = f.eval_naive(ctx, pts)
    assert np.array_equal(bits(f.eval_lane(ctx, pts)), bits(want))
    assert np.array_equal(bits(f.eval_wave(ctx, pts)), bits(want)) and np.array_equal(bits(f.eval(ctx, pts)), bits(want))
    # neighbours in the array are neighbours in space for the sampler; shuffled they are not: same values either way
    perm = np.random.default_rng(1).permutation(len(pts))
    assert np.array_equal(bits(f.eval_wave(ctx, pts[perm])), bits(want[perm]))
    assert (want < 0).any() and (want > 0).any()


def test_full_size_tree_approximates_the_field(H, ctx, torus, torus_build):
    """The reference's own acceptance test (HPUnitTests.cpp:46-77: |Query - F| on random points) at this size, and the build
    did refine in both ways."""
    verts, tris, lo, hi = torus
    f, blk, st = torus_build
    assert st["rounds"] >= 3 and st["h_refines"] > 0 and st["p_refines"] > 4096 and st["n_nodes"] > 10000
    pts = np.random.default_rng(7).uniform(lo, hi, (200_000, 3))
    q = H.DeviceTree(ctx, blk).query(pts)
    d = f.eval_wave(ctx, pts)
    # (a mesh's distance field has creases -- the medial surface inside the tube and around the hole -- that no polynomial
    # follows: the reference's 1e-2 bar, set for a sphere at 1e-8, holds here for all but a few points beside them)
    err = np.abs(q - d)
    assert err.max() <= 5e-2 and np.quantile(err, 0.999) <= 1e-2 and np.median(err) <= 5e-4


def test_full_size_sampling_paths_build_identical_trees(H, ctx, torus, torus_build, monkeypatch):
    """The sampler's overflow path (HPSDF_MESH_POOL_CAP: lanes whose pairs find the pool full walk the tree again) and the
    boxes-only hierarchy (HPSDF_MESH_NO_SLABS) sample the same field bits: the blocks are byte-identical."""
    verts, tris, lo, hi = torus
    _, blk, _ = torus_build
    cfg = H.make_config(1e-6, lo, hi)
    monkeypatch.setenv("HPSDF_MESH_POOL_CAP", "128")
    f2 = H.Field.mesh(ctx, verts, tris)
    assert H.create_block(ctx, cfg, f2, K)[0] == blk
    f2.close()
    monkeypatch.delenv("HPSDF_MESH_POOL_CAP")
    monkeypatch.setenv("HPSDF_MESH_NO_SLABS", "1")
    f3 = H.Field.mesh(ctx, verts, tris)
    assert H.create_block(ctx, H.make_config(1e-5, lo, hi), f3, K)[0] == H.create_block(ctx, H.make_config(1e-5, lo, hi), torus_build[0], K)[0]
    f3.close()


def test_full_size_eight_simulated_ranks_byte_identical(H, ctx, torus, torus_build):
    """BASELINE configs[3] at full size: the frontier sharded over 8 ranks (hpsdf_create_distributed on eight threads of this
    process, one context and one copy of the mesh each, the all-gather a barrier plus device copies): every rank's block
    equals the single-rank block."""
    verts, tris, lo, hi = torus
    _, one, st = torus_build
    cfg = H.make_config(1e-6, lo, hi)
    for blk, s in _create_on_simulated_ranks(H, 8, cfg, lambda c: H.Field.mesh(c, verts, tris), K):
        assert blk == one
        assert s["rounds"] == st["rounds"] and s["n_nodes"] == st["n_nodes"] and s["jobs"] == st["jobs"]


def test_full_size_create_with_continuity(H, ctx, torus, torus_build):
    """BASELINE configs[4] at full size and at the tighter target: continuity.enforce on the 1e-6 tree.  The block equals the
    host post-process (hpsdf_continuity_post_process: what the north_star keeps on the host) of the plain build."""
    verts, tris, lo, hi = torus
    f, plain, _ = torus_build
    cfg = H.make_config(1e-6, lo, hi, continuity=True)
    blk, st = H.create_block(ctx, cfg, f, K)
    cs = H.continuity_last_stats()
    assert cs["residual"] < 1e-6 and cs["jump_after"] < cs["jump_before"]
    b0 = bytearray(plain)
    b0[-80 + 16] = 1
    assert H.continuity_post_process(bytes(b0))[0] == blk


def test_full_size_sharded_create_with_continuity(H, ctx, torus, torus_build):
    """BASELINE configs[4] as written -- 2 M-triangle mesh, targetError 1e-5, continuity.enforce, frontier sharded over 8 ranks --
    through hpsdf_create_distributed (eight threads, one context and one copy of the mesh each): every rank's block equals the
    single-rank Create with continuity, which equals the host post-process of the plain build."""
    verts, tris, lo, hi = torus
    f = torus_build[0]
    cfg = H.make_config(1e-5, lo, hi, continuity=True)
    one, st = H.create_block(ctx, cfg, f, K)
    plain, _ = H.create_block(ctx, H.make_config(1e-5, lo, hi), f, K)
    b0 = bytearray(plain)
    b0[-80 + 16] = 1
    assert H.continuity_post_process(bytes(b0))[0] == one
    for blk, s in _create_on_simulated_ranks(H, 8, cfg, lambda c: H.Field.mesh(c, verts, tris), K):
        assert blk == one
        assert s["n_nodes"] == st["n_nodes"] and s["jobs"] == st["jobs"]


def test_destroyed_mesh_fields_give_their_memory_back(H, ctx, torus):
    """the mesh field's block and the build's temporaries come from a stream-ordered pool of the library's OWN
    (mesh_build.hip meshPool -- the device's default pool and its attributes are the application's), which holds on to at most
    1 GiB of freed blocks and is trimmed to that when a field is destroyed: creating and destroying the 2 M-triangle field over and
    over leaves the device's free memory where it was, up to that bound (each field is ~400 MB; the temporaries ~250 MB)."""
    import torch
    verts, tris, lo, hi = torus
    torch.cuda.synchronize()
    H.Field.mesh(ctx, verts, tris).close()  # (first use: the pool itself, rocPRIM's code objects)
    ctx.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    lows = []
    for _ in range(6):
        f = H.Field.mesh(ctx, verts, tris)
        lows.append(torch.cuda.mem_get_info()[0])
        f.close()
    ctx.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 <= (1 << 30) + (64 << 20), (free0, free1)            # nothing accumulates beyond the pool's bound
    assert max(lows) - min(lows) <= (1 << 30), lows                             # ... nor while fields come and go
