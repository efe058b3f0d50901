#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel of the built HIP objects (from the code objects' metadata notes).
usage: python tools/kernel_resources.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
objdir = os.path.join(ROOT, "hp-adaptive-signed-distance-field-octree_amd", "build")
notes = ""
with tempfile.TemporaryDirectory() as td:
    for obj in sorted(f for f in os.listdir(objdir) if f.endswith(".hip.o")):
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, os.path.join(objdir, obj), os.path.join(td, "copy.o")],  # (no output named = rewritten in place)
                       check=True, capture_output=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
        notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
rows = []
for blk in notes.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    rows.append((name, blk.split("\n")[0].strip(), g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print("%-6s %-6s %-6s %-8s %-8s %s" % ("vgpr", "agpr", "sgpr", "scratch", "lds", "kernel"))
for name, agpr, v, sg, scr, lds in sorted(rows):
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]):
        continue
    print("%-6s %-6s %-6s %-8s %-8s %s" % (v, agpr, sg, scr, lds, name[:150]))
