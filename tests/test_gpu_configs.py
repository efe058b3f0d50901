"""GPU tests for BASELINE configs[3] (mesh field at targetError 1e-6 through the sharded frontier) and for the
deserialiser against blocks shaped like the reference's own, all through the C ABI."""
import hashlib

import os

import numpy as np
import pytest

from conftest import bits
from helpers import displaced_torus, icosphere, synthetic_block

pytestmark = pytest.mark.gpu
TOL = 1e-6  # north_star: coefficients and Query() values within 1e-6 abs

MESH_ROOT = ((-0.45, -0.45, -0.2), (0.45, 0.45, 0.2))  # anisotropic, like "root = mesh AABB" (Octree.cpp:322-328)
MESH_K = 256                                            # several rounds, so the frontier exchange happens at all


def _mesh():
    return displaced_torus(12, 8)  # 192 triangles: the oracle scans every triangle for every sample


@pytest.fixture(scope="module")
def mesh_oracle_tree(O):
    verts, tris = _mesh()
    t = O.Tree.create(O.default_config(1e-6, *MESH_ROOT), O.MeshField(verts, tris), MESH_K)
    return t


def test_mesh_field_at_1e6_matches_oracle(H, O, ctx, mesh_oracle_tree):
    """configs[3] in shape: mesh field, targetError 1e-6, root = (anisotropic) mesh box; several rounds with H- and
    P-refinement.  The block equals the oracle's byte for byte (the mesh path is the reference's f32 operations in the
    reference's order, acosf included: test_device_acosf_is_the_host_libms)."""
    verts, tris = _mesh()
    blk, st = H.create_block(ctx, H.make_config(1e-6, *MESH_ROOT), H.Field.mesh(ctx, verts, tris), MESH_K)
    assert st["rounds"] >= 3 and st["h_refines"] > 0 and st["p_refines"] > 4096
    a, b = O.parse_block(blk), O.parse_block(mesh_oracle_tree.to_block())
    assert len(a["degree"]) == len(b["degree"])
    assert np.array_equal(a["degree"], b["degree"]) and np.array_equal(a["childIdx"], b["childIdx"])
    assert np.array_equal(a["depth"], b["depth"])
    leaf = a["degree"][a["degree"] != 13]
    assert leaf.max() >= 3 and a["depth"].max() >= 5  # both kinds of refinement happened
    assert blk == mesh_oracle_tree.to_block()
    lo, hi = np.array(MESH_ROOT[0]), np.array(MESH_ROOT[1])
    pts = (O.splitmix64_points(50000, seed=31) + 0.5) * (hi - lo) + lo
    assert np.array_equal(bits(H.DeviceTree(ctx, blk).query(pts)), bits(mesh_oracle_tree.query(pts)))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_mesh_field_through_sharded_frontier_is_byte_identical(H, ctx, world):
    """The same mesh build with the frontier sharded over `world` ranks (each rank's slice computed on this GPU, the
    headers exchanged as the all-gather would): the block equals the single-rank block byte for byte."""
    verts, tris = _mesh()
    cfg = H.make_config(1e-6, *MESH_ROOT)
    f = H.Field.mesh(ctx, verts, tris)
    one, st1 = H.create_block(ctx, cfg, f, MESH_K)
    builds = [H.Build(cfg, MESH_K, r, world) for r in range(world)]
    rounds = 0
    while True:
        n = builds[0].select()
        for b in builds[1:]:
            assert b.select() == n
        if n == 0:
            break
        rounds += 1
        hdr = np.zeros((n, 9))
        covered = 0
        for b in builds:
            b.compute(ctx, f)
            first, count = b.slice()
            assert first == covered
            covered += count
            hdr[first:first + count] = b.results_host(ctx).reshape(count, 9)
        assert covered == n
        for b in builds:
            b.apply(hdr)
    assert rounds == st1["rounds"]
    lays = [b.layout() for b in builds]
    packs = [builds[r].pack_host(ctx, lays[r][1][r]) for r in range(world)]
    for b in builds:
        assert b.assemble(packs) == one


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("wtype,target", [(1, 1e-8), (2, 1e-9)])
def test_weighted_build_sharded_is_byte_identical(H, ctx, world, wtype, target):
    """Nearness-weighted builds (what the reference's own tests and benchmarks switch on, HPUnitTests.cpp:53-58) with
    the frontier sharded over `world` ranks: a weighted incremental fit copies the node's previous rows (Octree.cpp:847),
    which another rank may have fitted -- after every round the ranks hand each other the arrays that round accepted
    (hpsdf_build_rows_*).  Every rank ends with the single-rank block, byte for byte."""
    cfg = H.make_config(target)
    cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = wtype, 3.0
    f = H.Field.sphere()
    one, st1 = H.create_block(ctx, cfg, f, 256)
    assert st1["p_refines"] > 4096 and st1["rounds"] >= 3  # incremental fits of nodes fitted in earlier rounds
    builds = [H.Build(cfg, 256, r, world) for r in range(world)]
    handed = 0
    while True:
        n = builds[0].select()
        for b in builds[1:]:
            assert b.select() == n
        if n == 0:
            break
        hdr = np.zeros((n, 9))
        for b in builds:
            b.compute(ctx, f)
            first, count = b.slice()
            hdr[first:first + count] = b.results_host(ctx).reshape(count, 9)
        for b in builds:
            b.apply(hdr)
        rc = builds[0].rows_counts()
        assert all(b.rows_counts() == rc for b in builds)
        parts = [builds[r].rows_pack_host(ctx, rc[r]) for r in range(world)]
        handed += sum(rc)
        for b in builds:
            b.rows_unpack_host(ctx, parts)
    assert handed > 0
    lays = [b.layout() for b in builds]
    packs = [builds[r].pack_host(ctx, lays[r][1][r]) for r in range(world)]
    for b in builds:
        assert b.assemble(packs) == one
        assert b.stats()["jobs"] == st1["jobs"]


def test_mesh_bvh_equals_linear_scan_bitwise(H, O, ctx):
    """TestBVHQuerying (MeshingUnitTests.cpp:110-138) on the device: the BVH traversal returns what the O(n) scan of
    Mesh::ClosestTriangleToPt (Mesh.cpp:134-159) returns -- same triangle, same bits -- and both match the oracle."""
    for verts, tris in (icosphere(3, 0.35, (0.02, 0.0, -0.01)), displaced_torus(48, 32)):
        f = H.Field.mesh(ctx, verts, tris)
        pts = O.splitmix64_points(4000, seed=3)
        a, b = f.eval(ctx, pts), f.eval_naive(ctx, pts)
        assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(f.eval_lane(ctx, pts)), bits(b))
        want, _, _ = O.MeshField(verts, tris).signed_distance(pts[:500])
        assert np.array_equal(bits(b[:500]), bits(want.astype(np.float64)))


def test_linear_scan_is_the_same_however_it_is_cut(H, O, ctx):
    """The O(n) scan cuts the triangles into as many slices as fill the chip (one point: hundreds; thousands of points: one)
    and merges the slices' winners by (distance, triangle index): the answer does not depend on the cut, and the tie rule
    is the reference's -- the first triangle of the scan among equals (Mesh.cpp:134-159) -- so points ON vertices and edges,
    where up to six triangles tie exactly, return the oracle's triangle and bits."""
    verts, tris = icosphere(4, 0.35, (0.02, 0.0, -0.01))  # 5120 triangles
    f = H.Field.mesh(ctx, verts, tris)
    rng = np.random.default_rng(5)
    t = rng.integers(0, len(tris), 300)
    a, b = verts[tris[t, 0].astype(np.int64)].astype(np.float64), verts[tris[t, 1].astype(np.int64)].astype(np.float64)
    pts = np.concatenate([rng.uniform(-0.5, 0.5, (400, 3)), a, 0.5 * (a + b), a + 1e-3 * rng.normal(size=a.shape)])
    whole = f.eval_naive(ctx, pts)
    for n in (1, 2, 7, 100):
        assert np.array_equal(bits(f.eval_naive(ctx, pts[:n])), bits(whole[:n]))
        assert np.array_equal(bits(f.eval_naive(ctx, pts[-n:])), bits(whole[-n:]))
    one_by_one = np.array([f.eval_naive(ctx, pts[i:i + 1])[0] for i in range(380, 440)])  # random points, then points on vertices
    assert np.array_equal(bits(one_by_one), bits(whole[380:440]))
    want, _, _ = O.MeshField(verts, tris).signed_distance(pts)
    assert np.array_equal(bits(whole), bits(want.astype(np.float64)))
    assert np.array_equal(bits(f.eval(ctx, pts)), bits(whole))


def test_mesh_evaluation_order_does_not_matter(H, ctx):
    """hpsdf_field_eval_* visits sets of 4096 points or more along a Morton curve (keys from the mesh's surroundings, an index
    sort on the device); smaller sets go as they are.  Same bits either way -- also for points far outside the key grid and
    repeated points.  A point with a coordinate that is not a finite number has no closest triangle (the reference's search
    ends with bestTri = -1 and reads out of bounds, Mesh.cpp:139,157): here its value is a NaN, the same on every path, and
    it costs its wave nothing."""
    verts, tris = displaced_torus(96, 64)
    f = H.Field.mesh(ctx, verts, tris)
    rng = np.random.default_rng(3)
    pts = rng.uniform(-0.6, 0.6, (9000, 3))
    pts[100:200] = pts[0]                      # repeated
    pts[200:260] *= 1e4                        # far outside the grid: all in the corner cells
    pts[260:264, 1] = np.nan
    pts[264, 0], pts[265, 2] = np.inf, -np.inf
    whole = f.eval(ctx, pts)                   # sorted
    parts = np.concatenate([f.eval(ctx, pts[i:i + 3000]) for i in range(0, 9000, 3000)])  # three calls below the threshold
    assert np.array_equal(bits(whole), bits(parts))
    assert np.array_equal(bits(f.eval(ctx, pts[:4096])), bits(parts[:4096])) and np.array_equal(bits(f.eval(ctx, pts[:4097])), bits(parts[:4097]))
    assert np.isnan(whole[260:266]).all() and not np.isnan(np.delete(whole, np.arange(260, 266))).any()
    assert np.array_equal(bits(whole), bits(f.eval_lane(ctx, pts))) and np.array_equal(bits(whole), bits(f.eval_wave(ctx, pts)))
    assert np.array_equal(bits(whole[:600]), bits(f.eval_naive(ctx, pts[:600])))


def test_field_that_is_not_finite_fails_the_build(H, ctx):
    """A field that is NaN or infinite at a sample makes the running total NaN / infinite for good; the reference's loop
    (`finished = total < threshold || queue.empty()`, Octree.cpp:216) would then refine until memory ends.  Here the build
    fails after the first round, on the device frontier and on the host scheduler alike."""
    for field in (H.Field.sphere((0.1, 0.0, 0.0), float("nan")), H.Field.sphere((float("inf"), 0.0, 0.0), 0.3),
                  H.Field.callback(lambda p, t: float("nan") if p[0] > 0.25 else float(np.linalg.norm(p) - 0.3))):
        with pytest.raises(H.HpsdfError) as e:
            H.create_block(ctx, H.make_config(1e-6), field, 1024)
        assert e.value.status == H.ERR_INVALID_ARGUMENT and "not a finite number" in str(e.value)
    # and the context is usable afterwards
    blk, st = H.create_block(ctx, H.make_config(1e-4), H.Field.sphere((0.25, 0, 0), 0.5), 1024)
    assert st["n_nodes"] == 4681


def _ranks_collect(H, world, cfg, make_field, K, setup=None):
    """As _create_on_simulated_ranks, but every rank's outcome (block or exception) is returned and a rank left waiting in an
    exchange shows as a broken barrier after a minute instead of hanging the test.  setup(rank, ctx): per-rank context settings."""
    import threading
    import torch
    ctxs = [H.Context(0) for _ in range(world)]
    if setup:
        for r, c in enumerate(ctxs):
            setup(r, c)
    fields = [make_field(c) for c in ctxs]
    barrier = threading.Barrier(world, timeout=60)
    bufs, out = [None] * world, [None] * world

    def gather_for(rank):
        def gather(d_buf, nbytes, stream):
            ctxs[rank].synchronize()
            bufs[rank] = d_buf
            barrier.wait()
            mine = torch.as_tensor(_DevBytes(d_buf, nbytes * world), device="cuda")
            for r in range(world):
                if r != rank:
                    other = torch.as_tensor(_DevBytes(bufs[r], nbytes * world), device="cuda")
                    mine[r * nbytes:(r + 1) * nbytes].copy_(other[r * nbytes:(r + 1) * nbytes])
            torch.cuda.synchronize()
            barrier.wait()
        return gather

    def worker(rank):
        try:
            out[rank] = H.create_block_distributed(ctxs[rank], cfg, fields[rank], K, rank, world, gather_for(rank))
        except BaseException as e:  # noqa: BLE001
            out[rank] = e

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return out


def _failing_rank_case(weighted, fail_at, expect_device_frontier):
    """(runs in a process of its own, on lib/libhpsdf_hooks.so: HPSDF_LIBRARY=hooks; the scheduler / selection path under test comes
    with the environment: HPSDF_FRONTIER_INLINE_NODES=0, HPSDF_HOST_FRONTIER=1)"""
    import hpsdf_loader
    H = hpsdf_loader.load()
    world, bad = 4, 2
    cfg = H.make_config(1e-7)
    if weighted:
        cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = 2, 3.0
    os.environ["HPSDF_TEST_FAIL_RANK"] = "%d:%d" % (bad, fail_at)
    out = _ranks_collect(H, world, cfg, lambda c: H.Field.union3(), 256)
    del os.environ["HPSDF_TEST_FAIL_RANK"]
    for r in range(world):
        assert isinstance(out[r], H.HpsdfError), (r, out[r])
        if r == bad:
            assert out[r].status == H.ERR_OUT_OF_MEMORY and "injected failure" in str(out[r])
        else:
            assert out[r].status == H.ERR_STATE and ("rank %d failed" % bad) in str(out[r])
    good = _create_on_simulated_ranks(H, world, cfg, lambda c: H.Field.union3(), 256)
    one, _ = H.create_block(H.Context(0), cfg, H.Field.union3(), 256)
    assert all(blk == one for blk, _ in good)
    assert all(st["device_frontier"] == expect_device_frontier for _, st in good), [st["device_frontier"] for _, st in good]
    print("failing-rank case ok")


@pytest.mark.parametrize("weighted,fail_at,env", [
    (False, 0, {}), (False, 3, {}), (True, 0, {}), (True, 2, {}),
    # the grid selection (trees beyond HPSDF_FRONTIER_INLINE_NODES): the round's part of the arena is the batch kernel's decision
    (False, 3, {"HPSDF_FRONTIER_INLINE_NODES": "0"}), (True, 0, {"HPSDF_FRONTIER_INLINE_NODES": "0"}),
    (True, 1, {"HPSDF_FRONTIER_INLINE_NODES": "0"}), (True, 3, {"HPSDF_FRONTIER_INLINE_NODES": "0"}),
    # the host scheduler's sharded rounds (what callbacks, logging builds and more than 8 ranks run)
    (False, 2, {"HPSDF_HOST_FRONTIER": "1"}), (True, 2, {"HPSDF_HOST_FRONTIER": "1"})])
def test_a_failing_rank_takes_the_others_out_with_it(weighted, fail_at, env):
    """One rank's share of a round fails on its own (HPSDF_TEST_FAIL_RANK stands in for a device allocation that one GPU cannot
    serve): it still enters the exchange(s) the others are heading for, with its status set; it returns its own error, every other
    rank HPSDF_ERR_STATE naming it -- nobody is left waiting in a collective.  Unweighted and weighted builds on the device-side
    frontier (the weighted ones in its replica mode: the round's rows are exchanged in front of the errors, and the failing rank has
    to enter that exchange too, with the size the others use), both with the inline selection and with the grid selection
    (HPSDF_FRONTIER_INLINE_NODES=0, where that size is only known once fr_batch_kernel has run), and the host scheduler's sharded
    rounds (HPSDF_HOST_FRONTIER=1); the contexts build normally afterwards.  The hook that makes a rank fail is compiled into
    lib/libhpsdf_hooks.so only (-DHPSDF_TEST_HOOKS): the case runs in a process of its own that loads that library, and the
    production library is shown not to look at the variable."""
    import subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_configs as T; T._failing_rank_case(%r, %r, %r)"
            % (ROOT, os.path.join(ROOT, "tests"), weighted, fail_at, 0 if env.get("HPSDF_HOST_FRONTIER") else 1))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HPSDF_LIBRARY="hooks", **env))
    assert r.returncode == 0 and "failing-rank case ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("wtype,strength,target", [(1, 3.0, 1e-8), (2, 3.0, 1e-10)])
def test_weighted_build_sharded_through_the_device_frontier(H, ctx, wtype, strength, target):
    """Nearness-weighted builds (the reference's own test and benchmark configuration, HPUnitTests.cpp:53-58, HPBenchmarks.cpp:34-39) on
    several ranks run the device-side frontier too (round 5): an incremental weighted fit carries the cell's previous rows over
    (Octree.cpp:847), which any rank may have fitted, so the ranks' arenas are replicas -- a round's fits laid out owner by owner at
    the same offsets everywhere, one in-place all-gather of the round's part beside the errors' (FrDev::replica).  Every rank's block
    is the single-rank block; the statistics say which scheduler ran."""
    import time
    cfg = H.make_config(target)
    cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = wtype, strength
    t0 = time.perf_counter()
    one, st1 = H.create_block(ctx, cfg, H.Field.sphere(), 1024)
    t1 = time.perf_counter()
    assert st1["device_frontier"] == 1
    for world in (2, 4):
        ta = time.perf_counter()
        out = _create_on_simulated_ranks(H, world, cfg, lambda c: H.Field.sphere(), 1024)
        tb = time.perf_counter()
        for blk, st in out:
            assert blk == one and st["device_frontier"] == 1 and st["rounds"] == st1["rounds"] and st["jobs"] == st1["jobs"]
        print("weighted sphere @ %g: one rank %.1f ms, %d simulated ranks (threads on one GPU, contexts and fields included) %.1f ms"
              % (target, (t1 - t0) * 1e3, world, (tb - ta) * 1e3))


@pytest.mark.parametrize("world", [2, 3, 4])
def test_weighted_build_sharded_with_the_grid_selection(H, ctx, monkeypatch, world):
    """The same replica mode when the selection runs as a grid (trees beyond HPSDF_FRONTIER_INLINE_NODES; forced here with 0): the
    round's part of the arena (FrHdr::roundBase / partStride) is then fr_batch_kernel's decision and the host fetches it -- behind a
    wait for the context's stream, which does not order itself against the null stream (without the wait the copy
    returned the previous round's part and the ranks gathered the wrong slice).  Polynomial weighting on union3 @ 1e-7 with 256 jobs a
    round: seventeen rounds whose parts all differ."""
    cfg = H.make_config(1e-7)
    cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = 1, 2.0
    one, st1 = H.create_block(ctx, cfg, H.Field.union3(), 256)
    monkeypatch.setenv("HPSDF_FRONTIER_INLINE_NODES", "0")
    again, st2 = H.create_block(ctx, cfg, H.Field.union3(), 256)
    assert again == one and st2["device_frontier"] == 1
    out = _create_on_simulated_ranks(H, world, cfg, lambda c: H.Field.union3(), 256)
    for blk, st in out:
        assert blk == one and st["device_frontier"] == 1 and st["rounds"] == st1["rounds"] and st["jobs"] == st1["jobs"]


def test_the_production_library_ignores_the_fault_injection_variable(H, ctx, monkeypatch):
    monkeypatch.setenv("HPSDF_TEST_FAIL_RANK", "1:0")
    cfg = H.make_config(1e-6)
    blocks = _create_on_simulated_ranks(H, 2, cfg, lambda c: H.Field.union3(), 256)
    one, _ = H.create_block(ctx, cfg, H.Field.union3(), 256)
    assert all(b == one for b, _ in blocks)


def _hard_meshes():
    yield "icosphere L5", icosphere(5, 0.4)
    yield "displaced torus", displaced_torus(160, 96)
    v, t = icosphere(4, 0.3)
    yield "far from the origin", ((v + np.float32([50.0, -30.0, 20.0])).astype(np.float32), t)
    v, t = icosphere(4, 0.4)
    yield "flattened (thin triangles)", ((v * np.float32([1.0, 0.01, 1.0])).astype(np.float32), t)
    v, t = icosphere(3, 0.35)
    v = v.copy()
    for k in range(0, len(t), 7):  # slivers: a corner pulled to within 1e-6 of its neighbour (still closed and manifold)
        a, b = int(t[k, 0]), int(t[k, 1])
        v[b] = v[a] + np.float32(1e-6) * (v[b] - v[a])
    yield "slivers", (v, t)


def test_tiny_meshes_through_the_device_build(H, ctx):
    """Four and eight triangles: fewer than one leaf holds -- the root's children are leaves, the LBVH has one or seven inner
    nodes -- through the device build, the host build and the O(n) scan."""
    tetra_v = np.float32([[0.3, 0.3, 0.3], [-0.3, -0.3, 0.3], [-0.3, 0.3, -0.3], [0.3, -0.3, -0.3]])
    tetra_t = np.uint64([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]])
    octa_v = np.float32([[0.3, 0, 0], [-0.3, 0, 0], [0, 0.25, 0], [0, -0.25, 0], [0, 0, 0.2], [0, 0, -0.2]])
    octa_t = np.uint64([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
    pts = np.random.default_rng(3).uniform(-0.5, 0.5, (3000, 3))
    for verts, tris in ((tetra_v, tetra_t), (octa_v, octa_t)):
        f = H.Field.mesh(ctx, verts, tris)
        want = f.eval_naive(ctx, pts)
        assert np.array_equal(bits(f.eval_lane(ctx, pts)), bits(want)) and np.array_equal(bits(f.eval_wave(ctx, pts)), bits(want))
        assert np.array_equal(bits(f.eval(ctx, pts)), bits(want))
        assert (want < 0).any() and (want > 0).any()
        blk, st = H.create_block(ctx, H.make_config(1e-3), f, 1024)
        assert st["n_nodes"] >= 4681
        f.close()


def _membrane_bipyramid(n=48, levels=2):
    """A bipyramid over an n-gon with a membrane across its equator: every equator edge has THREE faces (bottom, top, membrane),
    so one directed edge occurs twice -- not a manifold.  The reference's one-pass pairing (Mesh.cpp:87-131) accepts it because
    the direction that occurs once (the bottom's) comes first in index order: bottom, top, membrane."""
    ang = np.arange(n) * (2.0 * np.pi / n)
    eq = np.stack([0.33 * np.cos(ang), 0.33 * np.sin(ang), 0.02 * np.sin(3 * ang)], -1)
    v = [tuple(p) for p in eq] + [(0.0, 0.0, 0.3), (0.0, 0.0, -0.27), (0.0, 0.0, 0.01)]
    T, B, C = n, n + 1, n + 2
    f = [(B, (i + 1) % n, i) for i in range(n)] + [(T, i, (i + 1) % n) for i in range(n)] + [(C, i, (i + 1) % n) for i in range(n)]
    for _ in range(levels):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                v.append(tuple(0.5 * (np.array(v[a]) + np.array(v[b]))))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float32), np.array(f, np.uint64)


def test_non_manifold_mesh_keeps_the_device_hierarchy(H, O, ctx, monkeypatch):
    """A closed complex in which a directed edge occurs twice: the device's hash pairing cannot reproduce the reference's
    sequential std::map pass for it, so the twins come from the host -- while BVH, slabs and triangle records stay the device's
    (round 2 rebuilt everything on the host).  Same bits as the all-host preparation and as the O(n) scan, same Create block."""
    verts, tris = _membrane_bipyramid()
    pts = _hard_points(O, verts, tris, 5)
    f = H.Field.mesh(ctx, verts, tris)
    want = f.eval_naive(ctx, pts)
    assert (want > 0).any() and (want < 0).any()
    assert np.array_equal(bits(f.eval(ctx, pts)), bits(want)) and np.array_equal(bits(f.eval_lane(ctx, pts)), bits(want))
    cfg = H.make_config(1e-5, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
    blk, _ = H.create_block(ctx, cfg, f, 1024)
    monkeypatch.setenv("HPSDF_MESH_HOST_BUILD", "1")
    g = H.Field.mesh(ctx, verts, tris)
    assert np.array_equal(bits(g.eval(ctx, pts)), bits(want))
    assert H.create_block(ctx, cfg, g, 1024)[0] == blk
    monkeypatch.delenv("HPSDF_MESH_HOST_BUILD")
    # with the membrane FIRST the doubled direction meets an empty map and one half-edge stays without a twin: the reference
    # returns false (Mesh.cpp:121-128), and so do both preparations here
    n = len(tris) // 3
    bad = np.vstack([tris[2 * n:], tris[:2 * n]])
    for host in (False, True):
        if host:
            monkeypatch.setenv("HPSDF_MESH_HOST_BUILD", "1")
        with pytest.raises(H.HpsdfError) as e:
            H.Field.mesh(ctx, verts, bad)
        assert e.value.status == H.ERR_OPEN_MESH


def _hard_points(O, verts, tris, seed):
    from helpers import hard_points
    return hard_points(verts, tris, seed)


@pytest.mark.parametrize("host_build", [False, True])
def test_mesh_lower_bound_filter_keeps_the_scan_winner(H, O, ctx, monkeypatch, host_build):
    """Leaves of several triangles and the plane-and-circle lower bound in front of the closest-point test (kernels.hip,
    triLowerBound2) only skip work: per-lane traversal, the sampler's shared traversal and the O(n) scan return the same
    bits -- on thin triangles, slivers, meshes far from the origin, points on the surface and in the medial region, with
    the device-built LBVH and with the host-built tree."""
    if host_build:
        monkeypatch.setenv("HPSDF_MESH_HOST_BUILD", "1")
    for seed, (name, (verts, tris)) in enumerate(_hard_meshes()):
        f = H.Field.mesh(ctx, verts, tris)
        pts = _hard_points(O, verts, tris, seed)
        want = f.eval_naive(ctx, pts)
        assert np.array_equal(bits(f.eval_lane(ctx, pts)), bits(want)), name
        assert np.array_equal(bits(f.eval_wave(ctx, pts)), bits(want)), name
        perm = np.random.default_rng(seed).permutation(len(pts))
        assert np.array_equal(bits(f.eval_wave(ctx, pts[perm])), bits(want[perm])), name
        f.close()


@pytest.mark.parametrize("seed", [228, 489, 3624, 4851, 100758, 501177, 202581, 202707, 910968])
def test_needle_meshes_one_answer_on_every_path(H, O, ctx, monkeypatch, seed):
    """The random meshes on which four rounds of tools/fuzz_mesh_bvh.py soaks saw the exhaustive scan and the hierarchy
    disagree (spheres and tori squashed up to 1000 : 1: every triangle a needle).  The reference's closest-point routine
    (Utility.cpp:5-97) forms its face-case point from barycentric quotients and returns it even when the weights put it outside
    the triangle -- its absolute 1e-6 guards let that happen beside short edges -- i.e. a distance BELOW the triangle's, which a
    search sees or not depending on what it prunes (BVH.cpp:263-342 would too).  The product's rule (kernels.hip, closestSimplex):
    a face-case point farther outside its triangle than a quarter of the traversal's slack is not taken by a search it could win;
    the closest point of the triangle's boundary takes its place.  So: (1) the O(n) scan kernel, the per-lane traversal and the shared traversal agree BIT FOR BIT, always; (2) wherever
    they differ from the oracle's scan (= the reference's arithmetic), the oracle's value is such an artefact -- below the
    float64 brute-force distance -- and the product's is that distance."""
    from helpers import fuzz_mesh_case, hard_points, true_distance_f64
    verts, tris, leaf, host, scale, shift = fuzz_mesh_case(seed)
    monkeypatch.setenv("HPSDF_MESH_LEAF_TRIS", str(leaf))
    if host:
        monkeypatch.setenv("HPSDF_MESH_HOST_BUILD", "1")
    f = H.Field.mesh(ctx, verts, tris)
    pts = hard_points(verts, tris, seed)
    scan, lane, wave = f.eval_naive(ctx, pts), f.eval_lane(ctx, pts), f.eval_wave(ctx, pts)
    assert np.array_equal(bits(lane), bits(scan)) and np.array_equal(bits(wave), bits(scan))
    ref = O.MeshField(verts, tris).signed_distance(pts)[0].astype(np.float64)
    diff = np.nonzero(bits(ref) != bits(scan))[0]
    assert 1 <= len(diff) <= 8, len(diff)  # (the soaks found one point each)
    ext = max(float(np.linalg.norm(verts.max(0) - verts.min(0))), float(np.abs(verts).max()))  # the scale the slack is proportional to
    for i in diff:
        d = true_distance_f64(verts, tris, pts[i])
        assert abs(ref[i]) < d - 1e-6 * ext, (i, ref[i], d)                   # the reference's value: below the true distance
        # the product's: not below it (beyond the slack), above it by no more than the f32 routine's conditioning on needles
        assert -1e-5 * ext <= abs(scan[i]) - d <= 5e-4 * max(ext, d), (i, scan[i], d)
    f.close()


@pytest.mark.parametrize("seed", [100758, 501177, 200123])
def test_mesh_face_rule_switch_returns_the_reference_values(H, O, ctx, monkeypatch, seed):
    """hpsdf_set_mesh_face_rule(1): the closest-point routine returns the reference's face-case point whatever its weights
    (Utility.cpp:5-97).  On the needle meshes where the default rule differs from the oracle (the test above), the O(n) scan
    (Mesh.cpp:134-159) is then the oracle's scan BIT FOR BIT; the shared traversal, whose bounds assume the default rule, refuses; the
    batched evaluation takes the per-point traversal (boxes only, like the reference's BVH: where it differs from the scan it has
    pruned an artefact, i.e. its value lies ABOVE the scan's); and switching back restores the default's bytes."""
    from helpers import fuzz_mesh_case, hard_points
    verts, tris, leaf, host, scale, shift = fuzz_mesh_case(seed)
    monkeypatch.setenv("HPSDF_MESH_LEAF_TRIS", str(leaf))
    if host:
        monkeypatch.setenv("HPSDF_MESH_HOST_BUILD", "1")
    f = H.Field.mesh(ctx, verts, tris)
    pts = hard_points(verts, tris, seed)
    ref = O.MeshField(verts, tris).signed_distance(pts)[0].astype(np.float64)
    default_scan = f.eval_naive(ctx, pts)
    assert H.mesh_face_rule() == 0
    try:
        H.set_mesh_face_rule(True)
        assert H.mesh_face_rule() == 1
        scan = f.eval_naive(ctx, pts)
        assert np.array_equal(bits(scan), bits(ref))                    # the reference's arithmetic, needles included
        if seed != 200123:
            assert not np.array_equal(bits(scan), bits(default_scan))  # (these two meshes are where the rules differ)
        with pytest.raises(H.HpsdfError):
            f.eval_wave(ctx, pts)
        lane, batched = f.eval_lane(ctx, pts), f.eval(ctx, pts)
        assert np.array_equal(bits(batched), bits(lane))
        off = np.nonzero(bits(lane) != bits(scan))[0]
        assert len(off) <= 8 and np.all(np.abs(lane[off]) >= np.abs(scan[off]))
        few = f.eval(ctx, pts[:2])                                      # (answered on the calling thread: the same traversal)
        assert np.array_equal(bits(few), bits(lane[:2]))
    finally:
        H.set_mesh_face_rule(False)
    assert np.array_equal(bits(f.eval_naive(ctx, pts)), bits(default_scan))
    f.close()


def test_mesh_create_under_the_reference_face_rule(H, O, ctx):
    """Create with a mesh field under hpsdf_set_mesh_face_rule(1) runs the host scheduler with the per-point traversal inside the fit
    (the device-side frontier's sampler assumes the default rule).  On an ordinary mesh the two rules give the same values, so the
    block is the default's, byte for byte."""
    from helpers import icosphere
    v, t = icosphere(3, 0.4)
    lo, hi = v.min(0) - 0.02, v.max(0) + 0.02
    f = H.Field.mesh(ctx, v, t)
    cfg = H.make_config(1e-5, tuple(lo), tuple(hi))
    want, _ = H.create_block(ctx, cfg, f, 256)
    try:
        H.set_mesh_face_rule(True)
        got, _ = H.create_block(ctx, cfg, f, 256)
    finally:
        H.set_mesh_face_rule(False)
    assert got == want
    f.close()


# ------------------------------------------------------------------ blocks shaped like the reference's
def _reference_shaped(blk, rng):
    """What Octree::ToMemoryBlock really emits: interior nodes keep a stale heap pointer in basis.coeffs
    (Octree.cpp:267-268 frees but never clears the union), Node padding bytes 41-47 / 49-55 and Config padding bytes
    1-7, 17-23, 33-39 are indeterminate."""
    b = bytearray(blk)
    nc = int(np.frombuffer(blk[:8], np.uint64)[0])
    nn = int(np.frombuffer(blk[8 + 8 * nc:16 + 8 * nc], np.uint64)[0])
    base = 16 + 8 * nc
    for i in range(nn):
        o = base + 56 * i
        if b[o + 40] == 13:
            b[o + 32:o + 40] = (0x00007F0000000000 + int(rng.integers(1 << 30)) * 16).to_bytes(8, "little")
        b[o + 41:o + 48] = rng.integers(0, 256, 7, dtype=np.uint8).tobytes()
        b[o + 49:o + 56] = rng.integers(0, 256, 7, dtype=np.uint8).tobytes()
    c = len(b) - 80
    for lo, hi in ((1, 8), (17, 24), (33, 40)):
        b[c + lo:c + hi] = rng.integers(0, 256, hi - lo, dtype=np.uint8).tobytes()
    return bytes(b)


def test_upload_accepts_reference_shaped_block(H, O, ctx, golden):
    rng = np.random.default_rng(5)
    g = golden["blocks"]["A1_union3_1e-7_K1024"]
    clean, _ = H.create_block(ctx, H.make_config(g["target"]), H.Field.union3(), g["K"])
    dirty = _reference_shaped(clean, rng)
    assert dirty != clean
    pts = O.splitmix64_points(50000, seed=8)
    a, b = H.DeviceTree(ctx, clean), H.DeviceTree(ctx, dirty)
    assert a.info() == b.info()
    assert np.array_equal(bits(a.query(pts)), bits(b.query(pts)))
    va, ga = a.query_with_gradient(pts[:5000])
    vb, gb = b.query_with_gradient(pts[:5000])
    assert np.array_equal(bits(va), bits(vb)) and np.array_equal(bits(ga), bits(gb))
    # the Python Octree mirror round-trips the dirty bytes untouched (FromMemoryBlock copies, ToMemoryBlock returns them)
    t = H.Octree()
    t.FromMemoryBlock(dirty)
    assert t.ToMemoryBlock() == dirty
    # the continuity post-process reads the same dirty block (host path) and changes only coefficients
    cdirty = bytearray(dirty)
    cdirty[-80 + 16] = 1
    cdirty[-80 + 24:-80 + 32] = np.array([8.0]).tobytes()
    cclean = bytearray(clean)
    cclean[-80 + 16] = 1
    cclean[-80 + 24:-80 + 32] = np.array([8.0]).tobytes()
    pd, _ = H.continuity_post_process(bytes(cdirty))
    pc, _ = H.continuity_post_process(bytes(cclean))
    assert np.array_equal(O.parse_block(pd)["coeffs"], O.parse_block(pc)["coeffs"])


def test_upload_rejects_malformed_blocks(H, ctx):
    """hpsdf_tree_upload on untrusted bytes: wrap-around child indices, cycles, wrapping coefficient ranges,
    an interior root without children -- HPSDF_ERR_BAD_BLOCK each, no fault."""
    rng = np.random.default_rng(1)
    blk = synthetic_block(rng, [2] * 8, depth=2)
    nc = int(np.frombuffer(blk[:8], np.uint64)[0])
    nn = int(np.frombuffer(blk[8 + 8 * nc:16 + 8 * nc], np.uint64)[0])
    base = 16 + 8 * nc

    def edit(node, off, value, dtype=np.uint64):
        b = bytearray(blk)
        raw = np.array([value], dtype).tobytes()
        b[base + 56 * node + off:base + 56 * node + off + len(raw)] = raw
        return bytes(b)

    H.DeviceTree(ctx, blk)  # the unedited block is fine
    cases = {
        "child index 1<<40": edit(0, 0, 1 << 40),
        "children run past the end": edit(0, 0, nn - 3),
        "child is the root": edit(1, 0, 0),
        "child is its own parent block": edit(1, 0, 1),
        "coefficient start wraps": edit(2, 32, 0xFFFFFFFFFFFFFFFC),
        "coefficient start past the store": edit(2, 32, nc - 1),
        "leaf degree 200": edit(2, 40, 200, np.uint8),
        "leaf depth lies": edit(2, 48, 3, np.uint8),
    }
    tiny = bytearray(16 + 56 + 80)
    tiny[8:16] = np.array([1], np.uint64).tobytes()
    tiny[16:24] = np.array([1 << 40], np.uint64).tobytes()
    tiny[16 + 8:16 + 32] = np.array([-0.5] * 3 + [0.5] * 3, np.float32).tobytes()
    tiny[16 + 40] = 13
    cases["1-node interior root"] = bytes(tiny[:16 + 56]) + blk[-80:]
    for name, b in cases.items():
        with pytest.raises(H.HpsdfError) as e:
            H.DeviceTree(ctx, b)
        assert e.value.status == H.ERR_BAD_BLOCK, name
    t = H.Octree()
    with pytest.raises(H.HpsdfError):
        t.FromMemoryBlock(b"\x00" * 40)  # smaller than two counts + Config


# ------------------------------------------------------------------ the device-side frontier (csrc/frontier.hip)
@pytest.mark.parametrize("name,target,K", [("union3", 1e-5, 1024), ("union3", 1e-7, 1024), ("union3", 1e-7, 256),
                                           ("sphere", 1e-8, 1024), ("union3", 1e-8, 4096), ("sphere075", 1e-6, 64)])
def test_device_frontier_equals_host_scheduler(H, ctx, monkeypatch, name, target, K):
    """hpsdf_create runs selection, decision and bookkeeping on the device; HPSDF_HOST_FRONTIER=1 runs the host scheduler
    (csrc/builder.cpp).  Same schedule, same arithmetic: the MemoryBlock and the statistics are identical."""
    from helpers import product_field
    root = ((-0.25,) * 3, (5.0,) * 3) if name == "sphere075" else ((-0.5,) * 3, (0.5,) * 3)
    cfg = H.make_config(target, *root)
    f = product_field(H, name)
    monkeypatch.setenv("HPSDF_HOST_FRONTIER", "1")
    want, swant = H.create_block(ctx, cfg, f, K)
    monkeypatch.setenv("HPSDF_HOST_FRONTIER", "0")
    got, sgot = H.create_block(ctx, cfg, f, K)
    assert got == want
    for k in ("rounds", "jobs", "p_refines", "h_refines", "dropped", "fits", "samples", "n_nodes", "n_leaves", "n_coeffs", "total_error"):
        assert sgot[k] == swant[k], k


@pytest.mark.gpu
@pytest.mark.parametrize("host", ["0", "1"])
def test_block_allocator_receives_every_block(H, monkeypatch, host):
    """hpsdf_ctx_set_block_allocator (hpsdf.h): Create writes its MemoryBlock into memory the caller's allocator hands out -- the
    returned pointer is the allocator's, the bytes are those of a malloc'd block, a block begun and not returned (the one a build
    that goes on past round 0 prepared) is given back, and with the allocator taken out again blocks are malloc'd as before."""
    import ctypes as C
    monkeypatch.setenv("HPSDF_HOST_FRONTIER", host)
    L = H.lib()
    ctx = H.Context(0)
    L.hpsdf_ctx_set_block_allocator(ctx.handle, None, None, None)  # (the Python wrapper installs its own)
    f = H.Field.union3()

    def create(target):
        pod = H.make_config(target).to_pod()
        blk, sz, st = C.c_void_p(), C.c_size_t(), H.BuildStats()
        H.check(L.hpsdf_create(ctx.handle, C.byref(pod), f.handle, 1024, C.byref(blk), C.byref(sz), C.byref(st)))
        return blk, sz.value

    want = {}
    for tg in (1e-5, 1e-7):
        blk, n = create(tg)
        want[tg] = C.string_at(blk, n)
        L._libc.free(blk)
    live, log = {}, []

    def alloc(size, _user):
        buf = C.create_string_buffer(size)
        live[C.addressof(buf)] = buf
        log.append(("alloc", size))
        return C.addressof(buf)

    def release(ptr, _user):
        log.append(("release", len(live.pop(ptr))))

    a, r = H.BLOCK_ALLOC(alloc), H.BLOCK_RELEASE(release)
    L.hpsdf_ctx_set_block_allocator(ctx.handle, C.cast(a, C.c_void_p), C.cast(r, C.c_void_p), None)
    for tg in (1e-5, 1e-7):
        del log[:]
        blk, n = create(tg)
        assert blk.value in live and len(live[blk.value]) == n and live.pop(blk.value).raw == want[tg]
        assert not live, "a block that was not returned was not given back either"
        assert [e for e in log if e[0] == "alloc"][-1] == ("alloc", n)
        assert len([e for e in log if e[0] == "alloc"]) == 1 + len([e for e in log if e[0] == "release"])
    # an allocator that has no memory fails the build, and the library says so
    none = H.BLOCK_ALLOC(lambda size, _user: None)
    L.hpsdf_ctx_set_block_allocator(ctx.handle, C.cast(none, C.c_void_p), None, None)
    pod = H.make_config(1e-7).to_pod()
    blk, sz = C.c_void_p(), C.c_size_t()
    rc = L.hpsdf_create(ctx.handle, C.byref(pod), f.handle, 1024, C.byref(blk), C.byref(sz), None)
    assert rc == H.ERR_OUT_OF_MEMORY and not blk.value
    L.hpsdf_ctx_set_block_allocator(ctx.handle, None, None, None)
    blk, n = create(1e-5)
    assert C.string_at(blk, n) == want[1e-5]
    L._libc.free(blk)
    # ... and the wrapper's own allocator: create_block's bytes object is the block itself
    ctx2 = H.Context(0)
    got, _ = H.create_block(ctx2, H.make_config(1e-7), f, 1024)
    assert type(got) is bytes and got == want[1e-7] and not ctx2._blocks


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"HPSDF_FRONTIER_INLINE_NODES": "0"}, {"HPSDF_FRONTIER_NO_BLIND": "1"},
                                 {"HPSDF_FRONTIER_INLINE_NODES": "0", "HPSDF_FRONTIER_NO_BLIND": "1"}])
@pytest.mark.parametrize("target,K", [(1e-7, 256), (1e-8, 4096)])
def test_device_frontier_paths_agree(H, ctx, monkeypatch, env, target, K):
    """The round kernel's leader selects the next batch itself on trees of up to 131 072 nodes and leaves larger ones to the grid
    selection (fr_select / fr_batch / fr_tasks_kernel); fits are launched without waiting for the header from the second round on.
    Either switch thrown the other way gives the same bytes -- and, on two simulated ranks, so does the grid selection."""
    f = H.Field.union3()
    want, swant = H.create_block(ctx, H.make_config(target), f, K)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got, sgot = H.create_block(ctx, H.make_config(target), f, K)
    assert got == want and sgot == swant
    if "HPSDF_FRONTIER_INLINE_NODES" in env and K == 256:
        blocks = _create_on_simulated_ranks(H, 2, H.make_config(target), lambda c: H.Field.union3(), K)
        assert all(b[0] == want for b in blocks)


@pytest.mark.gpu
def test_round_one_launched_behind_round_zero_on_a_guess(H, O, monkeypatch):
    """When a context's last build went on past round 0, the next build's second-round lists and fits are enqueued right behind round
    0's closing launch (round 6).  The guess can be wrong either way: a build that stops after round 0 behind one that went on (two
    empty launches), a build that goes on behind one that stopped (the ordinary wait), and builds that go on one after the other --
    every block equals the one built with the guess switched off, on a fresh context, and the oracle's."""
    f = H.Field.union3()
    seq = [(1e-7, 1024), (1e-5, 1024), (1e-7, 256), (1e-6, 1024), (1e-5, 1024), (1e-5, 1024), (1e-7, 1024)]
    monkeypatch.setenv("HPSDF_FRONTIER_NO_BLIND", "1")
    c0 = H.Context(0)
    want = {tk: H.create_block(c0, H.make_config(tk[0]), f, tk[1]) for tk in set(seq)}
    c0.close()
    monkeypatch.delenv("HPSDF_FRONTIER_NO_BLIND")
    c = H.Context(0)
    for tk in seq:
        got, st = H.create_block(c, H.make_config(tk[0]), f, tk[1])
        assert got == want[tk][0] and st == want[tk][1], tk
        assert st["device_frontier"] == 1
    c.close()
    ot = O.Tree.create(O.default_config(1e-7), O.union3_field(), 1024)
    assert want[(1e-7, 1024)][0] == ot.to_block()
    # ... and with a mesh field (the second round's samples are then launched on the guess too)
    verts, tris = _mesh()
    mseq = [1e-6, 1e-3, 1e-6, 1e-5]
    monkeypatch.setenv("HPSDF_FRONTIER_NO_BLIND", "1")
    c0 = H.Context(0)
    m0 = H.Field.mesh(c0, verts, tris)
    mwant = {t: H.create_block(c0, H.make_config(t, *MESH_ROOT), m0, MESH_K) for t in set(mseq)}
    m0.close(), c0.close()
    monkeypatch.delenv("HPSDF_FRONTIER_NO_BLIND")
    c = H.Context(0)
    m = H.Field.mesh(c, verts, tris)
    rounds = []
    for t in mseq:
        got, st = H.create_block(c, H.make_config(t, *MESH_ROOT), m, MESH_K)
        assert got == mwant[t][0] and st == mwant[t][1], t
        rounds.append(st["rounds"])
    assert rounds[0] > 1 and rounds[1] == 1, rounds  # (a build that goes on, then one that stops after round 0)
    m.close(), c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("target,K", [(1e-7, 256), (1e-8, 1024)])
def test_split_mode_schedulers_agree_byte_for_byte(H, monkeypatch, target, K):
    """HPSDF_FIT_SPLIT with the split forced from degree 2 (every from-scratch fit of these trees is split): whether a round splits
    is one rule -- some class can be split and the round's samples number at most 2^31 -- applied by the host scheduler
    (builderCompute) and, on the device, by the batch of the round kernel: same bytes, same split_fits.  Switching the fast fit off
    gives the canonical mode again (round 4 sent it to the split mode)."""
    c = H.Context(0)
    c.set_fit_mode(H.FIT_SPLIT)
    c.set_split_min_degree(2)
    f = H.Field.union3()
    monkeypatch.setenv("HPSDF_HOST_FRONTIER", "1")
    want, swant = H.create_block(c, H.make_config(target), f, K)
    monkeypatch.setenv("HPSDF_HOST_FRONTIER", "0")
    got, sgot = H.create_block(c, H.make_config(target), f, K)
    assert got == want and swant["device_frontier"] == 0 and sgot["device_frontier"] == 1
    assert all(sgot[k] == swant[k] for k in sgot if k != "device_frontier")
    assert sgot["fit_mode"] == H.FIT_SPLIT and sgot["split_fits"] > 0
    c.set_fast_fit(False)
    assert c.fit_mode() == H.FIT_EXACT
    exact, sexact = H.create_block(c, H.make_config(target), f, K)
    assert sexact["split_fits"] == 0 and sexact["fit_mode"] == H.FIT_EXACT
    assert exact != got and len(exact) == len(got)  # the lower rows differ in their last bits; errors, topology, statistics do not
    for k in ("rounds", "jobs", "p_refines", "h_refines", "dropped", "fits", "samples", "n_nodes", "n_leaves", "n_coeffs", "total_error"):
        assert sexact[k] == sgot[k], k
    c.close()


class _DevBytes:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _create_on_simulated_ranks(H, world, cfg, make_field, K):
    """hpsdf_create_distributed on `world` threads of this process, one context each, all on this GPU; the all-gather is
    a thread barrier plus device-to-device copies between the ranks' buffers."""
    import threading
    import torch
    ctxs = [H.Context(0) for _ in range(world)]
    fields = [make_field(c) for c in ctxs]
    barrier = threading.Barrier(world)
    bufs, out, errs = [None] * world, [None] * world, []

    def gather_for(rank):
        def gather(d_buf, nbytes, stream):
            ctxs[rank].synchronize()
            bufs[rank] = d_buf
            barrier.wait()
            mine = torch.as_tensor(_DevBytes(d_buf, nbytes * world), device="cuda")
            for r in range(world):
                if r != rank:
                    other = torch.as_tensor(_DevBytes(bufs[r], nbytes * world), device="cuda")
                    mine[r * nbytes:(r + 1) * nbytes].copy_(other[r * nbytes:(r + 1) * nbytes])
            torch.cuda.synchronize()
            barrier.wait()
        return gather

    def worker(rank):
        try:
            out[rank] = H.create_block_distributed(ctxs[rank], cfg, fields[rank], K, rank, world, gather_for(rank))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            barrier.abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    return out


@pytest.mark.parametrize("seed", [3, 11, 29, 1007, 1044, 5132])
def test_random_builds_on_simulated_ranks(H, ctx, seed):
    """tools/fuzz_ranks.py's sweep in small: a random CSG field, root box, threshold, round size, weighting (none / Polynomial /
    Exponential) and world size 2..4 -- every rank's block equals the single-rank block, the statistics too (a rank counts its own
    fits and samples: their sums are the single-rank figures).  Weighted builds take the device frontier's replica mode."""
    rng = np.random.default_rng(seed)
    spec = []
    for k in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 4))
        c = rng.uniform(-0.3, 0.3, 3)
        if kind == H.PRIM_SPHERE:
            par = list(c) + [float(rng.uniform(0.08, 0.35))]
        elif kind == H.PRIM_BOX:
            par = list(c) + list(rng.uniform(0.05, 0.25, 3))
        elif kind == H.PRIM_TORUS_Y:
            par = list(c) + [float(rng.uniform(0.1, 0.25)), float(rng.uniform(0.03, 0.08))]
        else:
            nrm = rng.normal(size=3)
            nrm /= np.linalg.norm(nrm)
            par = list(nrm) + [float(rng.uniform(-0.2, 0.2))]
        spec.append((kind, H.OP_UNION if k == 0 else int(rng.integers(0, 3)), [float(x) for x in par]))
    lo = tuple(float(x) for x in (-0.5 + rng.uniform(-0.2, 0.2, 3)).astype(np.float32))
    hi = tuple(float(x) for x in (0.5 + rng.uniform(-0.2, 0.3, 3)).astype(np.float32))
    target = float(rng.choice([1e-5, 1e-6, 3e-7, 1e-7, 3e-8]))
    K = int(rng.choice([64, 256, 1024, 4096]))
    wtype = int(rng.choice([0, 1, 2, 2]))
    world = int(rng.integers(2, 5))
    cfg = H.make_config(target, lo, hi)
    if wtype:
        cfg.nearnessWeighting_type, cfg.nearnessWeighting_strength = wtype, float(rng.choice([2.0, 3.0]))
    one, st = H.create_block(ctx, cfg, H.Field.analytic(spec), K)
    res = _create_on_simulated_ranks(H, world, cfg, lambda c: H.Field.analytic(spec), K)
    keys = ("rounds", "jobs", "p_refines", "h_refines", "dropped", "n_nodes", "n_leaves", "n_coeffs", "total_error")
    for blk, s in res:
        assert blk == one
        assert all(s[k] == st[k] for k in keys) and s["device_frontier"] == 1
    assert sum(s["fits"] for _, s in res) == st["fits"] and sum(s["samples"] for _, s in res) == st["samples"]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_create_distributed_device_frontier_byte_identical(H, ctx, world):
    """The sharded build of the C ABI (hpsdf_create_distributed: slices cut on the device, one all-gather per round, one
    for the packed coefficients) with 2 / 4 / 8 ranks: every rank's block equals the single-rank block.  configs[3] in
    shape (mesh field, 1e-6, anisotropic root) and the refined analytic tree at two round sizes."""
    verts, tris = _mesh()
    cases = [(H.make_config(1e-6, *MESH_ROOT), lambda c: H.Field.mesh(c, verts, tris), MESH_K),
             (H.make_config(1e-7), lambda c: H.Field.union3(), 256),
             (H.make_config(1e-5), lambda c: H.Field.union3(), 1024)]
    for cfg, make_field, K in cases:
        one, st = H.create_block(ctx, cfg, make_field(ctx), K)
        for blk, s in _create_on_simulated_ranks(H, world, cfg, make_field, K):
            assert blk == one
            assert s["rounds"] == st["rounds"] and s["n_nodes"] == st["n_nodes"] and s["jobs"] == st["jobs"]


@pytest.mark.parametrize("world", [2, 8])
def test_create_distributed_with_continuity_byte_identical(H, ctx, world):
    """BASELINE configs[4]'s actual combination -- mesh field x continuity.enforce x sharded ranks -- through
    hpsdf_create_distributed: every rank runs the post-process (Octree.cpp:341-344) on its identical copy of the gathered tree,
    so every rank's block equals the single-rank block, which equals the host post-process of the plain sharded build.  And the
    same for the refined analytic tree."""
    verts, tris = _mesh()
    cases = [(1e-6, MESH_ROOT, lambda c: H.Field.mesh(c, verts, tris), MESH_K),
             (1e-7, ((-0.5,) * 3, (0.5,) * 3), lambda c: H.Field.union3(), 256)]
    for target, root, make_field, K in cases:
        cfg = H.make_config(target, *root, continuity=True)
        one, st = H.create_block(ctx, cfg, make_field(ctx), K)
        plain, _ = H.create_block(ctx, H.make_config(target, *root), make_field(ctx), K)
        b0 = bytearray(plain)
        b0[-80 + 16] = 1  # Config::continuity.enforce of the serialised block (Config.h:12-43)
        assert H.continuity_post_process(bytes(b0))[0] == one
        assert one != bytes(b0)  # (the post-process did move coefficients)
        for blk, s in _create_on_simulated_ranks(H, world, cfg, make_field, K):
            assert blk == one
            assert s["rounds"] == st["rounds"] and s["n_nodes"] == st["n_nodes"] and s["jobs"] == st["jobs"]


@pytest.mark.parametrize("world", [2, 3])
def test_create_distributed_shards_what_the_host_scheduler_owns(H, ctx, world):
    """hpsdf_create_distributed on builds the device-side frontier does not take -- nearness weighting (the reference's
    own test configuration, HPUnitTests.cpp:53-58) and fields given as host callbacks: the host scheduler's rounds,
    sharded over the caller's all-gather (errors every round, the accepted rows of weighted builds, the packed
    coefficients at the end).  Every rank ends with the single-rank block."""
    import math
    w = H.make_config(1e-8)
    w.nearnessWeighting_type, w.nearnessWeighting_strength = 1, 3.0
    one, st = H.create_block(ctx, w, H.Field.sphere(), 256)
    for blk, s in _create_on_simulated_ranks(H, world, w, lambda c: H.Field.sphere(), 256):
        assert blk == one and s["jobs"] == st["jobs"] and s["rounds"] == st["rounds"]
    if world == 2:
        def sphere(pt, thread_idx):
            return math.sqrt((pt[0] - 0.1) ** 2 + pt[1] ** 2 + pt[2] ** 2) - 0.3
        cfg = H.make_config(1e-3)
        one, st = H.create_block(ctx, cfg, H.Field.callback(sphere), 1024)
        for blk, s in _create_on_simulated_ranks(H, world, cfg, lambda c: H.Field.callback(sphere), 1024):
            assert blk == one and s["jobs"] == st["jobs"]


def test_bench_two_ranks_sharing_this_gpu():
    """bench.py's N > 1 path end to end: two torch.distributed processes (gloo, both on this GPU: HPSDF_BENCH_SHARE_GPU=1)
    -- sharded Create of the analytic field (asserted equal to the replicated one inside bench.py), the sharded mesh
    leg, the weak-scaling Query -- and one JSON line with the contract's keys from rank 0."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, HPSDF_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--points", "400000",
           "--mesh", "5", "--no-fit-bench", "--no-cpu-baseline", "--no-refined", "--no-sorted-ceiling"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["create_sharded_ms"] is not None and out["create_sharded_ms"] > 0
    assert out["mesh_create"]["n_gpus"] == 2 and out["mesh_create"]["create_ms_1e-6"] > 0
    for k in ("roofline", "config", "metric", "unit", "ms_per_step", "dtype"):
        assert k in out
    # the timed steps walk distinct batches (points from HBM); the one-batch loop of rounds 1-4 is reported beside them, not as `value`
    assert out["config"]["point_batches"] == 4 and out["repeated_batch"]["avg_launch_ms"] > 0
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1


    assert len(out["create_sharded"]["ms_per_rank"]) == 2 and out["create_sharded"]["exchanges_per_create"] >= 2
    assert len(out["mesh_create"]["create_ms_per_rank_1e-6"]) == 2 and out["mesh_create"]["exchanges_per_create_1e-6"] >= 3


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher (how the driver starts the N = 1 run): the parent spawns the two ranks as
    child processes before it has made any GPU call, relays rank 0's line and the exit code.  Octree::Create sharded over the
    ranks is Octree.cpp:312-352 run by N processes."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, HPSDF_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--points", "400000",
           "--mesh", "5", "--no-fit-bench", "--no-cpu-baseline", "--no-refined", "--no-sorted-ceiling"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world"] == 2 and out["backend"] == "gloo" and out["value"] > 0
    assert out["exchanges_per_create"] >= 2 and out["create_sharded_ms"] > 0


def test_bench_four_ranks_sharing_this_gpu():
    """The driver's scaling run is `bench.py --gpus N` for N up to 8 on a node this build never sees.  What can be rehearsed on a
    one-GPU box: the same command with the ranks sharing the card over gloo -- four of them, because the pool allows six processes on a
    GPU at once and this test runner and the launcher count (five ranks: "7 processes had the GPU open (limit 6)"; eight ranks run as
    threads in test_create_distributed_*).  bench.py itself asserts that the sharded block equals the replicated one."""
    import json, os, subprocess, sys, time
    from conftest import ROOT
    env = dict(os.environ, HPSDF_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--points", "1000000",
           "--mesh", "7", "--no-fit-bench", "--no-cpu-baseline", "--no-refined", "--no-sorted-ceiling"]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    wall = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["world"] == 4 and out["backend"] == "gloo" and out["value"] > 0
    assert isinstance(out["exchanges_per_create"], int) and out["exchanges_per_create"] >= 2
    assert "error" not in out["create_sharded"] and len(out["create_sharded"]["ms_per_rank"]) == 4
    m = out["mesh_create"]
    assert m["triangles"] == 327680 and len(m["create_ms_per_rank_1e-5"]) == 4 and len(m["create_ms_per_rank_1e-6"]) == 4
    assert isinstance(m["exchanges_per_create_1e-6"], int)
    print("bench.py --gpus 4 on one GPU: %.0f s of wall clock" % wall)
    assert wall < 600  # (the driver's lease: N real GPUs do each of these legs N times faster than four ranks on one)


_NCCL_GATHER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import hpsdf_loader, importlib
H = hpsdf_loader.load(); D = importlib.import_module("hpsdf_amd.distributed")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29547", rank=0, world_size=1, device_id=torch.device("cuda", 0))
stream = torch.cuda.Stream()
ctx = H.Context(0, stream.cuda_stream)
gather = D.device_allgather(ctx)
with torch.cuda.stream(stream):
    buf = torch.arange(4096, dtype=torch.uint8, device="cuda").repeat(4)   # written on the context's stream ...
    gather(buf.data_ptr(), buf.numel(), stream.cuda_stream)                 # ... gathered on it, no host wait in between
    after = buf + 1                                                         # ... and read on it
stream.synchronize()
assert gather.calls == 1 and torch.equal(after.cpu(), (torch.arange(4096, dtype=torch.uint8).repeat(4) + 1))
# the same through the whole sharded entry point is world > 1 only; here: the RCCL path accepts the in-place aliasing
print("ok")
dist.destroy_process_group()
'''


def test_nccl_allgather_runs_on_the_context_stream(tmp_path):
    """distributed.device_allgather with backend nccl (= RCCL) on the one GPU present: the in-place all_gather_into_tensor
    issued with the context's stream current -- what every rank of a sharded Create does twice per round on real multi-GPU
    hardware (the 2- and 8-rank cases need as many GPUs; the driver's scaling run is their first execution)."""
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "nccl_gather.py"
    script.write_text(_NCCL_GATHER)
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_create_distributed_argument_checks(H, ctx):
    with pytest.raises(H.HpsdfError) as e:
        H.create_block_distributed(ctx, H.make_config(1e-6), H.Field.sphere(), 1024, 2, 2, lambda *a: None)
    assert e.value.status == H.ERR_INVALID_ARGUMENT


def test_create_distributed_on_more_than_eight_ranks(H, ctx):
    """The device-side frontier holds a segment's owner rank in three bits; nine ranks take the host scheduler's rounds
    through the same entry point and end with the single-rank block."""
    cfg = H.make_config(1e-7)
    one, st = H.create_block(ctx, cfg, H.Field.union3(), 1024)
    for blk, s in _create_on_simulated_ranks(H, 9, cfg, lambda c: H.Field.union3(), 1024):
        assert blk == one and s["jobs"] == st["jobs"] and s["rounds"] == st["rounds"]


# ------------------------------------------------------------------ the opt-in matrix-core fit (csrc/fit_mfma.hip)
@pytest.mark.parametrize("name,target", [("union3", 1e-7), ("sphere", 1e-8), ("union3", 1e-8)])
def test_fast_fit_within_tolerance_of_oracle(H, O, name, target):
    """hpsdf_ctx_set_fast_fit: fits of degree >= 4 as a GEMM on v_mfma_f64_16x16x4_f64.  Not bit-identical by design;
    the gate is the north_star's: topology identical to the CPU oracle's, coefficients and Query() within 1e-6."""
    from helpers import oracle_field, product_field
    fast = H.Context(0)
    fast.set_fast_fit(True)
    blk, st = H.create_block(fast, H.make_config(target), product_field(H, name), 1024)
    want = O.Tree.create(O.default_config(target), oracle_field(O, name), 1024)
    a, b = O.parse_block(blk), O.parse_block(want.to_block())
    assert len(a["degree"]) == len(b["degree"])
    assert np.array_equal(a["degree"], b["degree"]) and np.array_equal(a["childIdx"], b["childIdx"])
    leaf = a["degree"][a["degree"] != 13]
    assert leaf.max() >= 4  # the matrix-core kernel has actually produced leaves
    assert np.abs(a["coeffs"] - b["coeffs"]).max() <= TOL
    assert not np.array_equal(a["coeffs"], b["coeffs"])  # (and it is a different arithmetic: not the default kernel by accident)
    pts = O.splitmix64_points(50000, seed=13)
    assert np.abs(H.DeviceTree(fast, blk).query(pts) - want.query(pts)).max() <= TOL
    fast.close()


def _same_stats(x, y):
    """build statistics but for the two that name the fit mode (hpsdf_build_stats::fit_mode, split_fits)"""
    return all(x[k] == y[k] for k in x if k not in ("fit_mode", "split_fits", "device_frontier"))


@pytest.mark.parametrize("name,target,K", [("union3", 1e-8, 1024), ("sphere", 1e-9, 1024), ("union3", 1e-7, 256)])
def test_split_fit_keeps_errors_and_topology_canonical(H, O, ctx, name, target, K, monkeypatch):
    """The default fit mode (HPSDF_FIT_SPLIT), made to split from degree 2 so that EVERY from-scratch fit of these trees is split (the
    default threshold is 6): a from-scratch fit's rows of top degree come from the bit-exact kernel, the rows below them from the
    sum-factorised kernel (csrc/fit_low.hip) -- and, second pass, from the direct contraction on the matrix cores (HPSDF_LOW_KERNEL=mfma,
    degrees >= 4).  Only the top rows enter a fit's error (Octree.cpp:1062-1069) and only errors are read by selection, the P/H decision
    (:600-601) and the stop rule (:216) -- so the node array, the statistics (total error included) and every coefficient of top
    degree equal the ORACLE's bit for bit with no guard band, and the other coefficients agree to 1e-12.  The host scheduler, the
    device-side frontier and two sharded ranks give the same bytes as each other."""
    from helpers import oracle_field, product_field
    split = H.Context(0)
    split.set_split_min_degree(2)
    blk, st = H.create_block(split, H.make_config(target), product_field(H, name), K)
    want = O.Tree.create(O.default_config(target), oracle_field(O, name), K).to_block()
    a, b = O.parse_block(blk), O.parse_block(want)
    nc = len(a["coeffs"])
    assert len(blk) == len(want) and blk[8 + 8 * nc:] == want[8 + 8 * nc:]       # node array and Config: byte for byte
    assert np.abs(a["coeffs"] - b["coeffs"]).max() <= 1e-12
    exact = H.Context(0)
    exact.set_fit_mode(H.FIT_EXACT)
    eb, est = H.create_block(exact, H.make_config(target), product_field(H, name), K)
    same = _same_stats
    assert eb == want and same(est, st)                                          # (statistics: jobs, rounds, fits, total error)
    assert est["split_fits"] == 0 and st["split_fits"] > 0 and st["fit_mode"] == H.FIT_SPLIT
    if name == "union3" and target == 1e-8:
        assert blk != want and a["degree"][a["degree"] != 13].max() >= 4         # the matrix cores did produce rows
    monkeypatch.setenv("HPSDF_HOST_FRONTIER", "1")
    assert H.create_block(split, H.make_config(target), product_field(H, name), K)[0] == blk
    monkeypatch.setenv("HPSDF_HOST_FRONTIER", "0")
    pts = O.splitmix64_points(50000, seed=13)
    assert np.abs(H.DeviceTree(split, blk).query(pts) - H.DeviceTree(split, want).query(pts)).max() <= 1e-12
    # the direct contraction on the matrix cores (fit_mfma_low_kernel, degrees >= 4), which the sum-factorised kernel replaced as the default
    monkeypatch.setenv("HPSDF_LOW_KERNEL", "mfma")
    split.set_split_min_degree(4)
    mb, mst = H.create_block(split, H.make_config(target), product_field(H, name), K)
    monkeypatch.delenv("HPSDF_LOW_KERNEL")
    m = O.parse_block(mb)
    assert mb[8 + 8 * nc:] == want[8 + 8 * nc:] and same(mst, st) and np.abs(m["coeffs"] - b["coeffs"]).max() <= 1e-12
    split.close(), exact.close()


def test_split_fit_sharded_and_sampled_fields(H, O, ctx):
    """Split fits where the samples come from somewhere else: a mesh field (the BVH sampler's buffer), and two simulated ranks
    (every rank splits its own slice): same bytes as one rank in the same mode; node arrays equal the exact mode's."""
    verts, tris = _mesh()
    cfg = H.make_config(1e-7, *MESH_ROOT)

    def mk(mode_ctx):
        mode_ctx.set_split_min_degree(2)
        return H.Field.mesh(mode_ctx, verts, tris)
    split, exact = H.Context(0), H.Context(0)
    exact.set_fit_mode(H.FIT_EXACT)
    sb, sst = H.create_block(split, cfg, mk(split), 256)
    eb, est = H.create_block(exact, cfg, H.Field.mesh(exact, verts, tris), 256)
    a, b = O.parse_block(sb), O.parse_block(eb)
    nc = len(a["coeffs"])
    assert sb[8 + 8 * nc:] == eb[8 + 8 * nc:] and _same_stats(sst, est) and a["degree"][a["degree"] != 13].max() >= 4
    assert np.abs(a["coeffs"] - b["coeffs"]).max() <= 1e-12 and sb != eb

    import threading  # two ranks, each a context in split mode from degree 4
    ucfg = H.make_config(1e-8)
    one, _ = H.create_block(split, ucfg, H.Field.union3(), 1024)

    def make_field(c):
        c.set_split_min_degree(2)
        return H.Field.union3()
    for blk, s in _create_on_simulated_ranks(H, 2, ucfg, make_field, 1024):
        assert blk == one
    split.close(), exact.close()


def test_split_fit_under_a_csg_wrapper(H, O, ctx):
    """UnionSDF (Octree.cpp:355-400: the old tree queried inside the new build's field) with split fits forced from degree 4: the
    exact kernel writes the COMBINED field value (old tree min new field) back to the sample buffer, so the matrix-core rows see
    what the exact rows saw.  Same operand tree for both builds; node array and statistics identical to the all-exact rebuild,
    coefficients to 1e-12, the device frontier and the host scheduler agree byte for byte."""
    import os
    exact, split = H.Context(0), H.Context(0)
    exact.set_fit_mode(H.FIT_EXACT)
    split.set_split_min_degree(2)
    cfg = H.make_config(1e-8)
    old_blk, _ = H.create_block(exact, cfg, H.Field.union3(), 1024)  # (leaves up to degree 5: the rebuild's jobs include from-scratch fits at 4 and 5)
    out = {}
    for name, c in (("exact", exact), ("split", split)):
        old = H.DeviceTree(c, old_blk)
        f = H.Field.tree_csg(old, H.OP_UNION, H.Field.sphere((0.3, 0.3, 0.3), 0.15))
        out[name] = H.create_block(c, cfg, f, 1024)
        if name == "split":
            os.environ["HPSDF_HOST_FRONTIER"] = "1"
            try:
                assert H.create_block(c, cfg, f, 1024)[0] == out[name][0]
            finally:
                os.environ["HPSDF_HOST_FRONTIER"] = "0"
    (eb, est), (sb, sst) = out["exact"], out["split"]
    a, b = O.parse_block(eb), O.parse_block(sb)
    nc = len(a["coeffs"])
    assert sb[8 + 8 * nc:] == eb[8 + 8 * nc:] and _same_stats(sst, est)
    assert a["degree"][a["degree"] != 13].max() >= 4
    assert np.abs(a["coeffs"] - b["coeffs"]).max() <= 1e-12
    exact.close(), split.close()


def test_fast_fit_on_a_sampled_field(H, O, ctx):
    """The fast fit on a sampled field (mesh: samples from the BVH kernel; host callbacks take the same kernel)."""
    fast = H.Context(0)
    fast.set_fast_fit(True)
    verts, tris = _mesh()
    cfg = H.make_config(1e-7, *MESH_ROOT)
    want, _ = H.create_block(ctx, cfg, H.Field.mesh(ctx, verts, tris), 256)
    got, st = H.create_block(fast, cfg, H.Field.mesh(fast, verts, tris), 256)
    a, b = O.parse_block(got), O.parse_block(want)
    assert np.array_equal(a["degree"], b["degree"]) and a["degree"][a["degree"] != 13].max() >= 4
    assert np.abs(a["coeffs"] - b["coeffs"]).max() <= TOL

    fast.close()


@pytest.mark.parametrize("host", ["0", "1"])
def test_build_limits_stop_a_runaway_build(H, ctx, golden, monkeypatch, host):
    """hpsdf_ctx_set_build_limits (round 6).  The reference's loop has no bound (Octree.cpp:212-216): a threshold below what the error
    estimate reaches -- the default Config()'s 1e-10 on most fields -- refines until memory ends, and until round 6 such a build ended
    here minutes later in a failed hipMalloc.  With a limit on nodes or on bytes the build is refused when the round that crosses it
    opens, with HPSDF_ERR_BUILD_LIMIT and a message that says how far it got; no block comes back; the context builds normally
    afterwards.  Device-side frontier and host scheduler alike."""
    monkeypatch.setenv("HPSDF_HOST_FRONTIER", host)
    c = H.Context(0)
    assert c.build_limits() == (0, 0)
    cfg = H.make_config(1e-8)  # union3 @ 1e-8, K = 1024: 35 545 nodes in 11 rounds
    c.set_build_limits(max_nodes=20000)
    with pytest.raises(H.HpsdfError) as e:
        H.create_block(c, cfg, H.Field.union3(), 1024)
    assert e.value.status == H.ERR_BUILD_LIMIT and "nodes (limit 20000)" in str(e.value) and "rounds" in str(e.value), str(e.value)
    c.set_build_limits(max_bytes=3 << 20)
    with pytest.raises(H.HpsdfError) as e:
        H.create_block(c, cfg, H.Field.union3(), 1024)
    assert e.value.status == H.ERR_BUILD_LIMIT and "GiB of device memory" in str(e.value) and "hpsdf_ctx_set_build_limits" in str(e.value), str(e.value)
    # a limit the build stays under changes nothing; nor does "none"
    g = golden["blocks"]["A1_union3_1e-7_K1024"]
    for limits in ((50000, 1 << 30), (None, None), (0, 0)):
        c.set_build_limits(*limits)
        blk, st = H.create_block(c, H.make_config(1e-7), H.Field.union3(), 1024)
        assert hashlib.sha256(blk).hexdigest() == g["block_sha256"] and st["device_frontier"] == (0 if host == "1" else 1)
    c.close()


@pytest.mark.parametrize("host", ["0", "1"])
def test_default_limits_leave_a_mesh_builds_sample_buffer_alone(H, monkeypatch, host):
    """The default limit bounds what grows with the tree (nodes, coefficient arena): a mesh build at 4096 jobs a round needs 1.5 GiB of
    SAMPLE buffer for its third round with a tree of 25 000 nodes -- above 1/256 of the device, the first default -- and is an ordinary build (the first
    version of the limits refused it).  A max_bytes the caller sets counts the sample buffer too.  Both schedulers."""
    monkeypatch.setenv("HPSDF_HOST_FRONTIER", host)
    verts, tris = displaced_torus(48, 32)
    c = H.Context(0)
    f = H.Field.mesh(c, verts, tris)
    cfg = H.make_config(1e-7, *MESH_ROOT)
    blk, st = H.create_block(c, cfg, f, 4096)
    assert st["n_nodes"] > 25000 and st["rounds"] >= 3 and st["device_frontier"] == (0 if host == "1" else 1), st
    # (the host scheduler samples a mesh inside its fits: no sample buffer, and a tree of this size needs far less than a GiB)
    c.set_build_limits(max_bytes=(1 << 30) if host == "0" else (16 << 20))
    with pytest.raises(H.HpsdfError) as e:
        H.create_block(c, cfg, f, 4096)
    assert e.value.status == H.ERR_BUILD_LIMIT and "hpsdf_ctx_set_build_limits" in str(e.value), str(e.value)
    c.set_build_limits(max_bytes=8 << 30)
    blk2, st2 = H.create_block(c, cfg, f, 4096)
    assert blk2 == blk and st2 == st
    f.close(), c.close()


def test_build_limits_on_simulated_ranks(H):
    """The same on two ranks: a bound on nodes stops every rank in the same round (the tree is replicated) -- each returns
    HPSDF_ERR_BUILD_LIMIT itself; a bound on bytes that only ONE rank has takes the other out through the failing-rank protocol
    (HPSDF_ERR_STATE naming it).  Nobody waits in a collective (the barrier of _ranks_collect would break after a minute)."""
    cfg = H.make_config(1e-8)
    out = _ranks_collect(H, 2, cfg, lambda c: H.Field.union3(), 1024, setup=lambda r, c: c.set_build_limits(max_nodes=20000))
    for r in range(2):
        assert isinstance(out[r], H.HpsdfError) and out[r].status == H.ERR_BUILD_LIMIT, (r, out[r])
    out = _ranks_collect(H, 2, cfg, lambda c: H.Field.union3(), 1024, setup=lambda r, c: c.set_build_limits(max_bytes=(3 << 20) if r == 1 else 0))
    assert isinstance(out[1], H.HpsdfError) and out[1].status == H.ERR_BUILD_LIMIT, out[1]
    assert isinstance(out[0], H.HpsdfError) and out[0].status == H.ERR_STATE and "rank 1 failed" in str(out[0]), out[0]
    # weighted (the device frontier's replica mode: two exchanges a round)
    w = H.make_config(1e-8)
    w.nearnessWeighting_type, w.nearnessWeighting_strength = 2, 3.0
    out = _ranks_collect(H, 2, w, lambda c: H.Field.union3(), 1024, setup=lambda r, c: c.set_build_limits(max_bytes=(3 << 20) if r == 1 else 0))
    assert isinstance(out[1], H.HpsdfError) and out[1].status == H.ERR_BUILD_LIMIT, out[1]
    assert isinstance(out[0], H.HpsdfError) and out[0].status == H.ERR_STATE and "rank 1 failed" in str(out[0]), out[0]


def test_two_contexts_differ_in_reduction_order_and_face_rule(H, O, golden):
    """hpsdf_ctx_set_reduction_order / hpsdf_ctx_set_mesh_face_rule (round 6): the two semantic switches per context -- the process-wide
    setters stay as the default a context follows until it is given its own.  Two contexts of ONE process build the C2 block under
    different reduction orders, interleaved: each equals the oracle under its own order, call after call; gradients likewise; and two
    contexts evaluate one needle mesh under the two face rules."""
    from helpers import fuzz_mesh_case, hard_points
    g = golden["blocks"]["C2_union3_1e-5"]
    a, b = H.Context(0), H.Context(0)
    for c in (a, b):
        c.set_fit_mode(H.FIT_EXACT)
    assert H.reduction_order() == 0 and a.reduction_order() == 0 and b.reduction_order() == 0
    b.set_reduction_order(True)
    assert b.reduction_order() == 1 and a.reduction_order() == 0 and H.reduction_order() == 0
    cfg = H.make_config(g["target"], g["root_min"], g["root_max"])
    O.set_reduction_order(1)
    try:
        ot_left = O.Tree.create(O.default_config(g["target"], g["root_min"], g["root_max"]), O.union3_field(), g["K"])
        left = ot_left.to_block()
        qp = O.splitmix64_points(20000, seed=5)
        init = np.full((len(qp), 3), 7.0)
        lv, lg = ot_left.query_with_gradient(qp, init)
    finally:
        O.set_reduction_order(0)
    assert hashlib.sha256(left).hexdigest() != g["block_sha256"]
    for _ in range(2):
        for host in ("0", "1"):
            os.environ["HPSDF_HOST_FRONTIER"] = host
            try:
                assert hashlib.sha256(H.create_block(a, cfg, H.Field.union3(), g["K"])[0]).hexdigest() == g["block_sha256"]
                assert H.create_block(b, cfg, H.Field.union3(), g["K"])[0] == left
            finally:
                del os.environ["HPSDF_HOST_FRONTIER"]
    # the gradient's normalize(): one block, two contexts (kernels, and the calls of <= 32 points answered on the calling thread)
    ta, tb = H.DeviceTree(a, left), H.DeviceTree(b, left)
    bv, bg = tb.query_with_gradient(qp, init)
    av, ag = ta.query_with_gradient(qp, init)
    assert np.array_equal(bits(bv), bits(lv)) and np.array_equal(bits(bg), bits(lg))
    assert np.array_equal(bits(av), bits(lv)) and not np.array_equal(bits(ag), bits(lg))
    sv, sg = tb.query_with_gradient(qp[:32], init[:32])
    assert np.array_equal(bits(sg), bits(lg[:32]))
    # following the process-wide setting again
    b.set_reduction_order(None)
    assert b.reduction_order() == 0
    assert hashlib.sha256(H.create_block(b, cfg, H.Field.union3(), g["K"])[0]).hexdigest() == g["block_sha256"]
    # the mesh face rule: one needle mesh, one field, two contexts
    verts, tris, leaf, host, scale, shift = fuzz_mesh_case(100758)
    os.environ["HPSDF_MESH_LEAF_TRIS"] = str(leaf)
    if host:
        os.environ["HPSDF_MESH_HOST_BUILD"] = "1"
    try:
        f = H.Field.mesh(a, verts, tris)
    finally:
        del os.environ["HPSDF_MESH_LEAF_TRIS"]
        os.environ.pop("HPSDF_MESH_HOST_BUILD", None)
    pts = hard_points(verts, tris, 100758)
    ref = O.MeshField(verts, tris).signed_distance(pts)[0].astype(np.float64)
    b.set_mesh_face_rule(True)
    assert b.mesh_face_rule() == 1 and a.mesh_face_rule() == 0 and H.mesh_face_rule() == 0
    da, db = f.eval_naive(a, pts), f.eval_naive(b, pts)
    assert np.array_equal(bits(db), bits(ref)) and not np.array_equal(bits(da), bits(ref))
    assert np.array_equal(bits(f.eval_naive(a, pts)), bits(da))
    with pytest.raises(H.HpsdfError):
        f.eval_wave(b, pts)
    f.eval_wave(a, pts)
    f.close()
    a.close(); b.close()
