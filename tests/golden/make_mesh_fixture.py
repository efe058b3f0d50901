"""tests/golden/halfedge_fail_mesh.npz: the vertex and index arrays of the one mesh the reference ships
(Resources/halfedge_fail.obj, 11 422 vertices / 22 840 triangles, closed) -- a data file of the reference,
stored as arrays.  Needs /root/reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import hpsdf_loader  # noqa: E402

H = hpsdf_loader.load()
v, t = H.load_obj("/root/reference/Resources/halfedge_fail.obj")
assert v.shape == (11422, 3) and t.shape == (22840, 3)
np.savez_compressed(os.path.join(HERE, "halfedge_fail_mesh.npz"), verts=v, tris=t.astype(np.uint32))
print(v.min(0), v.max(0))
