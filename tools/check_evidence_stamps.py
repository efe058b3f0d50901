#!/usr/bin/env python3
"""Are profiles/query_pmc.json, mesh_pmc.json and fit_pmc.json still the counters of the kernels in the tree?  (bench.py drops a
record whose source hash is stale to null; the hashes of mesh_pmc.json / fit_pmc.json cover the whole of kernels.hip, so ANY edit
there calls for tools/mesh_pmc.sh + tools/fit_pmc_all.sh again before the round's last bench.)  No GPU needed.
usage: python tools/check_evidence_stamps.py   (exit code 1 if anything is stale)"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hp-adaptive-signed-distance-field-octree_amd", "csrc")
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (counter_record is the code bench.py itself uses)

bad = 0
for name, keys in (("mesh_pmc.json", ("frac_valu_issue",)), ("fit_pmc.json", ("degrees",)), ("fit_mfma_pmc.json", ("degrees",))):
    m = bench.counter_record(name, keys)["measured_from"]
    state = "MISSING" if m is None else ("STALE" if m["stale"] else "current")
    bad += state != "current"
    print("%-16s %s %s" % (name, state, "" if m is None else "(recorded %s, tree %s)" % (m["source_sha16"], m["current_source_sha16"])))
text = open(os.path.join(CSRC, "kernels.hip")).read()
a, b = text.index("template <int TOPD, bool DEDUPE, bool GRAD>"), text.index("// 16-byte chunks a leaf of degree d occupies")
cur = hashlib.sha256(text[a:b].encode()).hexdigest()[:16]
try:
    rec = json.load(open(os.path.join(ROOT, "profiles", "query_pmc.json"))).get("query_kernel_sha16")
except Exception:  # noqa: BLE001
    rec = None
state = "MISSING" if rec is None else ("current" if rec == cur else "STALE")
bad += state != "current"
print("%-16s %s (recorded %s, tree %s)" % ("query_pmc.json", state, rec, cur))
sys.exit(1 if bad else 0)
