#!/usr/bin/env python3
"""How much does the unverified choice of Eigen's 3-vector reduction order matter?  

The oracle and the product associate Vector3d prod()/norm()/normalize() as a op (b op c); an SSE2 build of Eigen 3.4 may
reduce as (a op b) op c.  Eigen is absent here, so this tool builds the canonical trees both ways with the oracle
(ora_set_reduction_order) and reports: topology equality, the largest coefficient difference, the largest Query()
difference on 100 k points, and how close any refinement decision came to flipping (the smallest relative margin
|pImp - hImp| / max(|pImp|, |hImp|) is not available from outside, so the decisive evidence is topology + values).
CPU only; writes profiles/r02_assoc_sensitivity.txt when run with --write."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle as O  # noqa: E402

CASES = [("C1 sphere 1e-4", "sphere", 1e-4, 1024), ("C2 union3 1e-5", "union3", 1e-5, 1024),
         ("A1 union3 1e-7 K=1024", "union3", 1e-7, 1024), ("A1 union3 1e-7 K=256", "union3", 1e-7, 256),
         ("A2 sphere 1e-8", "sphere", 1e-8, 1024)]


def main():
    L = O.lib()
    L.ora_set_reduction_order.argtypes = [__import__("ctypes").c_int]
    lines = ["reduction-order sensitivity: a op (b op c) [default] vs (a op b) op c, oracle builds, 100 000 query points",
             "%-24s %9s %9s %12s %12s %12s" % ("case", "nodes", "topology", "max|dcoef|", "max|dQuery|", "coef bits =")]
    pts = O.splitmix64_points(100000, seed=5)
    for name, fld, target, K in CASES:
        field = O.sphere_field() if fld == "sphere" else O.union3_field()
        out = []
        for order in (0, 1):
            L.ora_set_reduction_order(order)
            t = O.Tree.create(O.default_config(target), field, K)
            out.append((O.parse_block(t.to_block()), t.query(pts)))
        L.ora_set_reduction_order(0)
        (a, qa), (b, qb) = out
        same = len(a["degree"]) == len(b["degree"]) and np.array_equal(a["degree"], b["degree"]) and \
            np.array_equal(a["childIdx"], b["childIdx"])
        if same:
            dc = float(np.abs(a["coeffs"] - b["coeffs"]).max())
            eq = float(np.mean(a["coeffs"].view(np.uint64) == b["coeffs"].view(np.uint64)))
        else:
            dc, eq = float("nan"), float("nan")
        dq = float(np.abs(qa - qb).max())
        lines.append("%-24s %9d %9s %12.3e %12.3e %11.1f%%" % (name, len(a["degree"]), "same" if same else "DIFFERS", dc, dq, 100 * eq))
        if not same:
            true = field.eval(pts)
            la, lb = a["degree"][a["degree"] != 13], b["degree"][b["degree"] != 13]
            lines.append("    default : %d nodes, leaf degrees %s, max|Query - F| %.3e" % (len(a["degree"]), np.bincount(la).tolist(), np.abs(qa - true).max()))
            lines.append("    (a op b): %d nodes, leaf degrees %s, max|Query - F| %.3e" % (len(b["degree"]), np.bincount(lb).tolist(), np.abs(qb - true).max()))
            lines.append("    points whose Query differs by more than 1e-6: %d of %d" % (int((np.abs(qa - qb) > 1e-6).sum()), len(pts)))
    txt = "\n".join(lines)
    print(txt)
    if "--write" in sys.argv:
        open(os.path.join(ROOT, "profiles", "r02_assoc_sensitivity.txt"), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
