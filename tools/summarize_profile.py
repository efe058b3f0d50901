"""Condenses a rocprofv3 output directory (tools/profile.sh) into a small text + JSON summary for profiles/."""
import csv, glob, json, os, sys

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
stats = list(rows("trace/**/*kernel_stats.csv"))
for r in sorted(stats, key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))[:12]:
    print("%-90s calls %6s  avg %10.1f ns  total %12s ns  %6s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]),
                                                                    r["TotalDurationNs"], r.get("Percentage", "")))
summary = {"kernels": {r["Name"]: {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                   "total_ns": float(r["TotalDurationNs"])} for r in stats}}
# per-dispatch trace: resources of our kernels
seen = set()
for r in rows("trace/**/*kernel_trace.csv"):
    n = r["Kernel_Name"]
    if n in seen or "hpsdf" not in n:
        continue
    seen.add(n)
    print("dispatch %-70s grid %s wg %s vgpr %s sgpr %s lds %s scratch %s" % (
        n[:70], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")),
        r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    vals = {}
    for r in rows(sub + "/**/*counter_collection.csv"):
        if r.get("Counter_Name") != name:
            continue
        vals.setdefault(r["Kernel_Name"], []).append((int(r.get("Dispatch_Id", 0) or 0), float(r["Counter_Value"])))
    for k, v in vals.items():
        v = [x for _, x in sorted(v, key=lambda t: t[0])]
        if "query_kernel" in k or "fit_kernel" in k:
            if k.startswith("void hpsdf::query_kernel<4, true>") and len(v) > 6:
                v = v[1:6]  # the timed launches of `bench.py --steps 5 --warmup 1` (tools/profile.sh): the ones that walk the distinct batches
            avg = sum(v) / len(v)
            print("%s %-60s dispatches %d  avg %.1f (KB units)" % (name, k[:60], len(v), avg))
            summary.setdefault("pmc", {}).setdefault(k, {})[name] = avg
# HBM bytes per query launch, MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950
# FETCH_SIZE under-reports wide coalesced reads by 2x -> the doubled figure is an upper bound for this
# kernel's mixed 8-byte strided / gather pattern (uncalibrated widths), so both are recorded.
for k, d in summary.get("pmc", {}).items():
    if "query_kernel" in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        raw = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        corrected = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        summary["query"] = {"hbm_bytes_per_launch_raw": raw, "hbm_bytes_per_launch": corrected,
                            "fetch_kb": d["FETCH_SIZE"], "write_kb": d["WRITE_SIZE"]}
        print("query_kernel HBM bytes/launch: raw %.0f, with the gfx950 x2 read correction %.0f" % (raw, corrected))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
# the headline kernel's launches inside bench.py's timed region (after the warm-up launches, before the cell-sorted ones),
# from the trace's own timestamps: the figure bench.py's HIP-event average (roofline.avg_launch_ms) has to agree with
try:
    import csv, glob
    line = [l for l in open(os.path.join(out, "bench_trace.log")) if l.startswith("{")][0]
    b = json.loads(line)
    tr = glob.glob(os.path.join(out, "trace", "*", "*_kernel_trace.csv"))[0]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr))
         if r["Kernel_Name"].startswith("void hpsdf::query_kernel<4, true>")]
    timed = d[b["warmup"]:b["warmup"] + b["steps"]]
    summary.setdefault("query", {}).update(timed_launches=len(timed), timed_avg_us=sum(timed) / len(timed),
                                            bench_hip_event_avg_us=b["roofline"]["avg_launch_ms"] * 1e3)
    print("query_kernel<4, true>: %d timed launches, trace average %.1f us; bench.py HIP events %.1f us"
          % (len(timed), sum(timed) / len(timed), b["roofline"]["avg_launch_ms"] * 1e3))
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
except Exception as e:
    print("timed-region average not computed:", e)

# profiles/query_pmc.json: what bench.py reports as roofline.traffic, stamped with the source of the kernel it was measured on
if "query" in summary:
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "hp-adaptive-signed-distance-field-octree_amd", "csrc", "kernels.hip")).read()
    a, b = text.index("template <int TOPD, bool DEDUPE, bool GRAD>"), text.index("// 16-byte chunks a leaf of degree d occupies")
    rec = dict(summary["query"], kernel="query_kernel<4, true>", profile=os.path.basename(out.rstrip("/")),
               query_kernel_sha16=hashlib.sha256(text[a:b].encode()).hexdigest()[:16],
               note="FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B read requests at 64 B); WRITE_SIZE exact; "
                    "separate --pmc passes of bench.py --steps 5 (tools/profile.sh)")
    json.dump(rec, open(os.path.join(out, "query_pmc.json"), "w"), indent=1)
    print("wrote", os.path.join(out, "query_pmc.json"))
