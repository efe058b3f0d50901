"""Shared test helpers: fields by name, point sets, synthetic blocks, meshes."""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def oracle_field(O, name):
    if name == "sphere":
        return O.sphere_field()
    if name == "union3":
        return O.union3_field()
    if name == "sphere075":
        return O.sphere_field((0.25, 0.0, 0.0), 0.75)
    raise KeyError(name)


def product_field(H, name):
    if name == "sphere":
        return H.Field.sphere()
    if name == "union3":
        return H.Field.union3()
    if name == "sphere075":
        return H.Field.sphere((0.25, 0.0, 0.0), 0.75)
    raise KeyError(name)


def query_points(O, root_min, root_max, n=10000):
    """Same construction as tests/golden/make_golden.py."""
    p = O.splitmix64_points(n)
    lo, hi = np.array(root_min), np.array(root_max)
    p = (p + 0.5) * (hi - lo) + lo
    p[:64] = (p[:64] - lo) * 3.0 + lo - (hi - lo)
    return p


def edge_points(rng, n=4000):
    """Points on cell mid-planes, root faces, corners, just outside, NaN -- the descent's edge cases."""
    p = rng.uniform(-0.5, 0.5, (n, 3))
    k = n // 8
    grid = np.arange(-8, 9) / 16.0  # depth-4 mid-planes and faces
    p[:k, 0] = rng.choice(grid, k)
    p[k:2 * k, 1] = rng.choice(grid, k)
    p[2 * k:3 * k] = rng.choice(grid, (k, 3))
    p[3 * k:4 * k] = rng.choice([-0.5, 0.5], (k, 3))
    p[4 * k:5 * k, 2] = 0.5 + 1e-9          # f32 cast rounds back onto the face: still inside
    p[5 * k:6 * k, 0] = np.nextafter(np.float32(0.5), np.float32(1.0)).astype(np.float64)  # first f32 outside
    p[6 * k:6 * k + 8] = np.nan
    p[6 * k + 8:6 * k + 16] = np.inf
    p[6 * k + 16:6 * k + 24, 1] = -0.5 - 1e-3
    return p


NCOEF = [1, 4, 10, 20, 35, 56, 83, 120, 165, 220, 286, 364, 455]


def synthetic_block(rng, degrees, depth=1, root_min=(-0.5,) * 3, root_max=(0.5,) * 3):
    """A hand-made MemoryBlock: root + 8 children; child i is a leaf of degree degrees[i]
    (or, with depth=2, child 0 is split again into 8 leaves of degrees[0]).  Random coefficients."""
    nodes = []

    def node(child, bmin, bmax, start, degree, dep):
        b = np.zeros(56, np.uint8)
        b[0:8] = np.array([child], np.uint64).view(np.uint8)
        b[8:32] = np.array(list(bmin) + list(bmax), np.float32).view(np.uint8)
        b[32:40] = np.array([start], np.uint64).view(np.uint8)
        b[40] = degree
        b[48] = dep
        return b

    def corner(bmin, bmax, i):
        lo, hi = list(bmin), list(bmax)
        for d in range(3):
            mid = (np.float32(bmax[d]) + np.float32(bmin[d])) * np.float32(0.5)
            if (i >> d) & 1:
                lo[d] = mid
            else:
                hi[d] = mid
        return lo, hi

    coeffs = []
    leaf = np.uint64(0xFFFFFFFFFFFFFFFF)
    rmin, rmax = [-0.5] * 3, [0.5] * 3
    nodes.append(node(1, rmin, rmax, 0, 13, 0))
    level1 = []
    for i in range(8):
        lo, hi = corner(rmin, rmax, i)
        level1.append((lo, hi))
    cur = 0
    extra = []
    for i, (lo, hi) in enumerate(level1):
        if depth == 2 and i == 0:
            nodes.append(node(9, lo, hi, 0, 13, 1))
            for j in range(8):
                l2, h2 = corner(lo, hi, j)
                extra.append(node(leaf, l2, h2, cur, degrees[0], 2))
                c = rng.standard_normal(NCOEF[degrees[0]])
                coeffs.append(c)
                cur += len(c)
        else:
            nodes.append(node(leaf, lo, hi, cur, degrees[i], 1))
            c = rng.standard_normal(NCOEF[degrees[i]])
            coeffs.append(c)
            cur += len(c)
    nodes += extra
    coeffs = np.concatenate(coeffs)
    cfg = np.zeros(80, np.uint8)
    cfg[40:48] = np.array([1e-10], np.float64).view(np.uint8)
    cfg[48:56] = np.array([1], np.uint64).view(np.uint8)
    cfg[56:80] = np.array(list(root_min) + list(root_max), np.float32).view(np.uint8)
    blk = (np.array([len(coeffs)], np.uint64).tobytes() + coeffs.tobytes() + np.array([len(nodes)], np.uint64).tobytes()
           + np.concatenate(nodes).tobytes() + cfg.tobytes())
    return blk


def icosphere(level=1, radius=0.35, centre=(0.0, 0.0, 0.0)):
    """Closed, consistently CCW (outward) triangle mesh: 20 * 4^level triangles."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(x, np.float64) / np.linalg.norm(x) for x in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    verts = (np.array(v) * radius + np.array(centre)).astype(np.float32)
    return verts, np.array(f, np.uint64)


def deep_chain_block(rng, max_depth=10, degrees=(2, 3, 1, 0, 4, 3, 2, 5)):
    """A MemoryBlock whose octant 7 is split again and again down to `max_depth` (TREE_MAX_DEPTH = 10): at every level
    seven leaves (degrees cycling through `degrees`) and one interior child; the last level holds eight leaves."""
    nodes, coeffs = [], []
    leaf = np.uint64(0xFFFFFFFFFFFFFFFF)
    cur = [0]

    def node(child, bmin, bmax, start, degree, dep):
        b = np.zeros(56, np.uint8)
        b[0:8] = np.array([child], np.uint64).view(np.uint8)
        b[8:32] = np.array(list(bmin) + list(bmax), np.float32).view(np.uint8)
        b[32:40] = np.array([start], np.uint64).view(np.uint8)
        b[40] = degree
        b[48] = dep
        return b

    def corner(bmin, bmax, i):
        lo, hi = list(bmin), list(bmax)
        for d in range(3):
            mid = (np.float32(bmax[d]) + np.float32(bmin[d])) * np.float32(0.5)
            if (i >> d) & 1:
                lo[d] = mid
            else:
                hi[d] = mid
        return lo, hi

    box = ([-0.5] * 3, [0.5] * 3)
    nodes.append(node(1, box[0], box[1], 0, 13, 0))
    k = 0
    for dep in range(1, max_depth + 1):
        first = len(nodes)
        nxt = None
        for i in range(8):
            lo, hi = corner(box[0], box[1], i)
            if i == 7 and dep < max_depth:
                nodes.append(node(first + 8, lo, hi, 0, 13, dep))
                nxt = (lo, hi)
            else:
                deg = degrees[k % len(degrees)]
                k += 1
                c = rng.standard_normal(NCOEF[deg])
                nodes.append(node(leaf, lo, hi, cur[0], deg, dep))
                coeffs.append(c)
                cur[0] += len(c)
        box = nxt
    coeffs = np.concatenate(coeffs)
    cfg = np.zeros(80, np.uint8)
    cfg[40:48] = np.array([1e-10], np.float64).view(np.uint8)
    cfg[48:56] = np.array([1], np.uint64).view(np.uint8)
    cfg[56:80] = np.array([-0.5] * 3 + [0.5] * 3, np.float32).view(np.uint8)
    return (np.array([len(coeffs)], np.uint64).tobytes() + coeffs.tobytes() + np.array([len(nodes)], np.uint64).tobytes()
            + np.concatenate(nodes).tobytes() + cfg.tobytes())


def displaced_torus(nu=1024, nv=1024, R=0.3, r=0.1, amp=0.02):
    """Closed, consistently outward-oriented torus grid with a smooth radial displacement: 2 * nu * nv triangles
    (1024 x 1024 -> 2 097 152, SURVEY 8(d)'s stand-in for the 2 M-triangle mesh of the north_star)."""
    u = np.arange(nu) * (2.0 * np.pi / nu)
    v = np.arange(nv) * (2.0 * np.pi / nv)
    U, V = np.meshgrid(u, v, indexing="ij")
    rr = r * (1.0 + amp / r * np.sin(7 * U) * np.cos(5 * V) + 0.5 * amp / r * np.sin(11 * V + 3 * U))
    x = (R + rr * np.cos(V)) * np.cos(U)
    y = (R + rr * np.cos(V)) * np.sin(U)
    z = rr * np.sin(V)
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * nv + j).ravel()
    b = (((i + 1) % nu) * nv + j).ravel()
    c = (((i + 1) % nu) * nv + (j + 1) % nv).ravel()
    d = (i * nv + (j + 1) % nv).ravel()
    tris = np.concatenate([np.stack([a, b, c], -1), np.stack([a, c, d], -1)]).astype(np.uint64)
    return verts, tris


def hard_points(verts, tris, seed):
    """Points where a BVH's bounds and leaf grouping could bite: around the mesh, on vertices / edges / faces, just off the surface,
    the medial region (near-ties everywhere), far away."""
    rng = np.random.default_rng(seed)
    lo, hi = verts.min(0).astype(np.float64), verts.max(0).astype(np.float64)
    ext = (hi - lo).max()
    c = 0.5 * (lo + hi)
    tri = verts[tris[rng.integers(0, len(tris), 600)]].astype(np.float64)
    w = rng.dirichlet((1.0, 1.0, 1.0), 600)
    return np.concatenate([
        lo - 0.1 * ext + rng.random((3000, 3)) * (hi - lo + 0.2 * ext),        # around the mesh
        verts[rng.integers(0, len(verts), 300)].astype(np.float64),             # on vertices
        0.5 * (tri[:300, 0] + tri[:300, 1]),                                    # on edges
        (tri * w[:, :, None]).sum(1),                                           # on faces
        (tri * w[:, :, None]).sum(1) + 1e-4 * ext * rng.standard_normal((600, 3)),  # just off the surface
        c + 1e-3 * ext * rng.standard_normal((300, 3)), c[None, :],             # the medial region: near-ties everywhere
        c + 10.0 * ext * rng.standard_normal((200, 3)),                         # far away
    ])


def fuzz_mesh_case(seed):
    """The random closed mesh of tools/fuzz_mesh_bvh.py for a seed: a bumpy icosphere or a displaced torus under a random affine map
    (anisotropic scales down to 1e-3 -- needles --, translations up to 100 extents).  Returns verts, tris, leaf size, host build, scale."""
    rng = np.random.default_rng(seed)
    if seed % 2 == 0:
        verts, tris = icosphere(int(rng.integers(2, 6)), 0.3)
        d = verts / np.linalg.norm(verts, axis=1, keepdims=True)
        verts = verts * (1 + rng.uniform(0, 0.3) * np.sin(rng.integers(2, 9) * d[:, 0] + seed) * np.cos(rng.integers(2, 9) * d[:, 1]))[:, None]
    else:
        verts, tris = displaced_torus(int(rng.integers(10, 120)), int(rng.integers(8, 90)), 0.28, 0.09, float(rng.uniform(0, 0.03)))
    scale = 10.0 ** rng.uniform(-3, 0, 3) if seed % 3 == 0 else np.ones(3)
    shift = rng.uniform(-1, 1, 3) * (100.0 if seed % 5 == 0 else 0.1)
    verts = (verts * scale + shift).astype(np.float32)
    leaf = int(rng.choice([1, 2, 4, 8, 16]))
    return verts, tris, leaf, seed % 7 == 3, scale, shift


def closest_point_on_triangle_f64(p, a, b, c):
    """Ericson's closest point of a triangle, vectorised over triangles, in float64 (the brute-force truth of the mesh tests)."""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(1), (ac * ap).sum(1)
    bp = p - b
    d3, d4 = (ab * bp).sum(1), (ac * bp).sum(1)
    cp = p - c
    d5, d6 = (ab * cp).sum(1), (ac * cp).sum(1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    with np.errstate(all="ignore"):
        den = 1.0 / (va + vb + vc)
        q = a + ab * (vb * den)[:, None] + ac * (vc * den)[:, None]
        m = (va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0)
        q[m] = (b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[:, None])[m]
        m = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
        q[m] = (a + ac * (d2 / (d2 - d6))[:, None])[m]
        m = (d6 >= 0) & (d5 <= d6)
        q[m] = c[m]
        m = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
        q[m] = (a + ab * (d1 / (d1 - d3))[:, None])[m]
        m = (d3 >= 0) & (d4 <= d3)
        q[m] = b[m]
        m = (d1 <= 0) & (d2 <= 0)
        q[m] = a[m]
    return q


def true_distance_f64(verts, tris, p):
    """Unsigned distance of one point from the mesh by an exhaustive float64 scan."""
    A, B, Cc = (verts[tris[:, k]].astype(np.float64) for k in range(3))
    q = closest_point_on_triangle_f64(np.asarray(p, np.float64)[None, :], A, B, Cc)
    return float(np.sqrt(((q - p) ** 2).sum(1)).min())
