#!/bin/bash
# SQ counters of the mesh sampling kernel (and of the fused mesh fit kernel under HPSDF_MESH_FUSED=1) (tools/mesh_probe.py <level>).  Usage: bash tools/mesh_pmc.sh <tag> <level>
TAG=${1:-mesh}; LEVEL=${2:-8}
OUT=$PWD/gpurun_out/pmcm_$TAG
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 $REPO/tools/mesh_probe.py $LEVEL > $OUT/log$i.txt 2>&1
done
cd $REPO && python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "fit_kernel<" not in r["Kernel_Name"] and "mesh_sample_kernel" not in r["Kernel_Name"]:
            continue
        acc.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-42s %-26s n=%d avg %.4g max %.4g" % (k[0], k[1], len(v), sum(v) / len(v), max(v)))
def avg(name):
    for k in acc:
        if k[1] == name and "mesh_sample" in k[0]:
            return sum(acc[k]) / len(acc[k])
    return None
tc, iv, w = avg("SQ_THREAD_CYCLES_VALU"), avg("SQ_INSTS_VALU"), avg("SQ_WAVES")
if tc and iv:
    print("mesh_sample_kernel SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = %.1f active lanes per VALU instruction" % (tc / iv))
    print("mesh_sample_kernel lane utilisation SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU * 64) = %.3f   [/ (... * 256) = %.3f]" % (tc / (iv * 64), tc / (iv * 256)))
    # calibration of the counter's full scale: the fit kernel of the same run has every lane active in all but a few instructions
    for k in sorted(acc):
        if k[1] == "SQ_THREAD_CYCLES_VALU" and "fit_kernel<" in k[0]:
            fi = acc.get((k[0], "SQ_INSTS_VALU"))
            if fi:
                print("calibration: %s SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = %.1f (all lanes active: full scale is 64, not 256)"
                      % (k[0], (sum(acc[k]) / len(acc[k])) / (sum(fi) / len(fi))))
    print("mesh_sample_kernel VALU instructions per wave (64 samples): %.0f" % (iv / w))
fs, ws = avg("FETCH_SIZE"), avg("WRITE_SIZE")
if fs and w:
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 counts a 128-byte read request as 64 bytes (MI355X_MICROARCH.md): reads doubled
    print("mesh_sample_kernel HBM traffic per sample: fetched %.1f B (counter x 2), written %.1f B" % (2 * fs * 1024 / (w * 64), (ws or 0) * 1024 / (w * 64)))
PY
python3 tools/pmc_json.py mesh "$OUT" "$OUT/mesh_pmc.json" "$TAG" > /dev/null && echo "wrote $OUT/mesh_pmc.json (copy to profiles/mesh_pmc.json)"
tail -2 $OUT/log1.txt
