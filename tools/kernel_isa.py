#!/usr/bin/env python3
"""Disassembly of one kernel of the built HIP objects.  usage: python tools/kernel_isa.py <object stem, e.g. kernels> <mangled-name substring>"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.join(ROOT, "hp-adaptive-signed-distance-field-octree_amd", "build", sys.argv[1] + ".hip.o")
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
    # (an output file is named: with the input alone llvm-objcopy rewrites it in place, and the fresh time stamp hides later header edits from build.py)
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(td, "copy.o")], check=True, capture_output=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
for blk in re.split(r"\n(?=[0-9a-f]+ <)", dis):
    head = blk.split("\n", 1)[0]
    if sys.argv[2] in head and ".kd" not in head:
        print(blk)
