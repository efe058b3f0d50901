"""Regenerates tests/golden/*.json.  Run in the build container (needs /root/reference for the
reference-table hashes; everything else comes from the oracle).

  tables.json   SHA-256 of the reference's own constant tables, obtained by compiling the
                reference headers Include/HP/Utility.h + Include/HP/Legendre.h where they lie
                (oracle/_ref/libref_tables.so) -- values, not source text.
  blocks.json   SHA-256 + shape statistics of the MemoryBlocks the oracle builds under the
                canonical schedule, and SHA-256 of its Query values on the SplitMix64 point set.
  kats.json     is NOT generated: it holds the known-answer values printed in SURVEY.md
                (Appendix C, section 7 H1), which the surveyor obtained from the reference's own
                FitPolynomial / EstimateH/PImprovement.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def tables():
    r = O.ref_tables()
    assert r is not None, "needs oracle/_ref (reference checkout)"
    out = {k: {"sha256": sha(v), "shape": list(v.shape), "dtype": str(v.dtype)} for k, v in r.items()}
    out["coeff_count"]["values"] = [int(x) for x in r["coeff_count"]]
    out["spot"] = {
        "basis_index_rows_82_85": r["basis_index"][82:86].astype(int).tolist(),
        "gl9_abs_order": np.abs(r["roots"][36:45]).tolist(),
        "nl_2_4": float(r["normalised_lengths"][2, 4]),
    }
    return out


CASES = {
    # name: (field, target, K, root_min, root_max)
    "C1_sphere_1e-4": ("sphere", 1e-4, 1024, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)),
    "C2_union3_1e-5": ("union3", 1e-5, 1024, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)),
    "A1_union3_1e-7_K1024": ("union3", 1e-7, 1024, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)),
    "A1_union3_1e-7_K256": ("union3", 1e-7, 256, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)),
    "A2_sphere_1e-8_K1024": ("sphere", 1e-8, 1024, (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)),
    "D1_sphere075_customroot_1e-6": ("sphere075", 1e-6, 512, (-0.25, -0.25, -0.25), (5.0, 5.0, 5.0)),
}


def field_of(name):
    if name == "sphere":
        return O.sphere_field()
    if name == "union3":
        return O.union3_field()
    if name == "sphere075":  # Source/Tests/HPUnitTests.cpp:287-290
        return O.sphere_field((0.25, 0.0, 0.0), 0.75)
    raise KeyError(name)


def query_points(root_min, root_max, n=10000):
    p = O.splitmix64_points(n)  # in [-0.5,0.5)^3
    lo, hi = np.array(root_min), np.array(root_max)
    p = (p + 0.5) * (hi - lo) + lo
    p[:64] = (p[:64] - lo) * 3.0 + lo - (hi - lo)  # a few outside the root
    return p


def blocks():
    out = {}
    for name, (fname, target, K, rmin, rmax) in CASES.items():
        cfg = O.default_config(target, rmin, rmax)
        t = O.Tree.create(cfg, field_of(fname), K)
        blk = t.to_block()
        pb = O.parse_block(blk)
        leaves = pb["degree"] != 13
        q = t.query(query_points(rmin, rmax))
        out[name] = {
            "field": fname, "target": target, "K": K, "root_min": list(rmin), "root_max": list(rmax),
            "block_sha256": hashlib.sha256(blk).hexdigest(), "block_bytes": len(blk),
            "n_nodes": pb["n_nodes"], "n_coeffs": pb["n_coeffs"],
            "degree_hist": {str(int(d)): int((pb["degree"][leaves] == d).sum()) for d in np.unique(pb["degree"][leaves])},
            "depth_hist": {str(int(d)): int((pb["depth"][leaves] == d).sum()) for d in np.unique(pb["depth"][leaves])},
            "stats": t.stats, "query_sha256": sha(q), "query_sum_finite": float(q[q < 1e300].sum()),
            "query_n_outside": int((q > 1e300).sum()),
        }
        print(name, out[name]["n_nodes"], out[name]["degree_hist"])
    return out


if __name__ == "__main__":
    json.dump(tables(), open(os.path.join(HERE, "tables.json"), "w"), indent=1)
    json.dump(blocks(), open(os.path.join(HERE, "blocks.json"), "w"), indent=1)
