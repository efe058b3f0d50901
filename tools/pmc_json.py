#!/usr/bin/env python3
"""Counter figures for bench.py's secondary rooflines, from the rocprofv3 --pmc passes of tools/mesh_pmc.sh / tools/fit_pmc_all.sh:
writes profiles-ready JSON stamped with a hash of the kernels' source, so that bench.py can tell a stale record from a live one
(it reports null instead of yesterday's counters).
  pmc_json.py mesh <pmc dir> <out.json> <tag>
  pmc_json.py fit <out.json> <tag> <degree>=<pmc dir> ...
  pmc_json.py mfma <out.json> <tag> <degree>:<field>=<pmc dir> ...      (tools/fit_mfma_pmc.sh: the fast fit's matrix-core counters)"""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hp-adaptive-signed-distance-field-octree_amd", "csrc")


def source_sha16(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


MESH_FILES = ["kernels.hip", "device_types.hpp", "acosf_host_libm.hpp"]
FIT_FILES = ["kernels.hip", "fit_low.hip", "field_eval.hpp", "device_types.hpp"]
MFMA_FILES = ["fit_mfma.hip", "field_eval.hpp", "device_types.hpp"]


def counters(pmc_dir, want):
    acc = {}
    for f in glob.glob(os.path.join(pmc_dir, "**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if not want(r["Kernel_Name"]):
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def valu_issue(c):
    """VALU issue slots in use: SQ_INSTS_VALU x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) (MI355X_MICROARCH.md, PMC notes)"""
    if "SQ_INSTS_VALU" in c and c.get("GRBM_GUI_ACTIVE"):
        return c["SQ_INSTS_VALU"] * 4.0 / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    return None


def main():
    if sys.argv[1] == "mesh":
        pmc_dir, out, tag = sys.argv[2:5]
        c, n = counters(pmc_dir, lambda k: "mesh_sample_kernel" in k)
        w = c.get("SQ_WAVES")
        rec = {"kernel": "mesh_sample_kernel", "profile": tag, "source_sha16": source_sha16(MESH_FILES), "launches": n.get("SQ_INSTS_VALU"),
               "frac_valu_issue": valu_issue(c),
               "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64.0) if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_INSTS_VALU") else None,
               "valu_insts_per_64_samples": c["SQ_INSTS_VALU"] / w if w and c.get("SQ_INSTS_VALU") else None,
               # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 counts a 128-byte read request as 64 bytes (MI355X_MICROARCH.md): reads doubled
               "fetched_bytes_per_sample": 2 * c["FETCH_SIZE"] * 1024 / (w * 64) if w and c.get("FETCH_SIZE") else None,
               "written_bytes_per_sample": c["WRITE_SIZE"] * 1024 / (w * 64) if w and c.get("WRITE_SIZE") else None,
               "note": "rocprofv3 --pmc passes of tools/mesh_probe.py (tools/mesh_pmc.sh); averages over the kernel's launches"}
    elif sys.argv[1] == "mfma":
        out, tag = sys.argv[2:4]
        rec = {"profile": tag, "source_sha16": source_sha16(MFMA_FILES), "degrees": {},
               "note": "rocprofv3 --pmc passes of tools/fit_one.py <field> <degree> 16384 fast (tools/fit_mfma_pmc.sh), kernel fit_mfma_kernel: mfma_busy = "
                       "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) -- the fraction of the chip's matrix-pipe cycles in use while the kernel "
                       "ran; union3 = fused with the headline field, plane = contraction only"}
        for arg in sys.argv[4:]:
            key, pmc_dir = arg.split("=", 1)
            deg, fld = key.split(":")
            c, n = counters(pmc_dir, lambda k: "fit_mfma_kernel" in k)
            d = rec["degrees"].setdefault("p" + deg, {})
            if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
                d[fld] = {"mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), "frac_valu_issue": valu_issue(c),
                          "mfma_mops_f64": c.get("SQ_INSTS_VALU_MFMA_MOPS_F64"), "gui_active_cycles": c["GRBM_GUI_ACTIVE"], "launches": n.get("GRBM_GUI_ACTIVE")}
    else:
        out, tag = sys.argv[2:4]
        rec = {"profile": tag, "source_sha16": source_sha16(FIT_FILES), "field": "union3", "degrees": {},
               "note": "rocprofv3 --pmc passes of tools/fit_one.py union3 <degree> (tools/fit_pmc_all.sh): VALU issue = SQ_INSTS_VALU x 4 / "
                       "(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), per kernel of the default mode's launch set"}
        for arg in sys.argv[4:]:
            deg, pmc_dir = arg.split("=", 1)
            d = {}
            for kname, want in (("fit_kernel", lambda k: "fit_kernel<" in k), ("fit_low_kernel", lambda k: "fit_low_kernel" in k)):
                c, n = counters(pmc_dir, want)
                if c.get("SQ_INSTS_VALU"):
                    d[kname] = {"frac_valu_issue": valu_issue(c), "valu_insts": c["SQ_INSTS_VALU"], "gui_active_cycles": c.get("GRBM_GUI_ACTIVE"),
                                "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c else None,
                                "launches": n.get("SQ_INSTS_VALU")}
            rec["degrees"]["p" + deg] = d
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
