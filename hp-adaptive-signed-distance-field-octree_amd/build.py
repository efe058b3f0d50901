"""Build libhpsdf.so (HIP kernels + host runtime + C ABI + C++ drop-in) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the built .so sits next to this file in lib/ so that it
travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhpsdf.so")
HOOKS_LIB = os.path.join(LIBDIR, "libhpsdf_hooks.so")
HOOK_SOURCES = ["frontier.hip", "capi.cpp"]
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))

SOURCES = ["kernels.hip", "frontier.hip", "fit_mfma.hip", "fit_low.hip", "mesh_build.hip", "cg.hip", "continuity_asm.hip", "tables.cpp", "builder.cpp", "mesh.cpp", "obj.cpp", "continuity.cpp", "host_query.cpp", "capi.cpp"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp"))  # every header: an edit to any of them rebuilds every object
PUBLIC_HEADERS = ["hpsdf.h", "hpsdf_octree.hpp"]

# -ffp-contract=off: no multiply-add is fused anywhere (bit parity with the x86-64 reference path)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-I", INCLUDE]
# diagnostic builds: HPSDF_EXTRA_FLAGS="-DHPSDF_MESH_STATS_BUILD" python build.py --force
FLAGS += os.environ.get("HPSDF_EXTRA_FLAGS", "").split()
# per-file flags.  fit_mfma.hip: keep the matrix instructions' accumulators in VGPRs -- left to its heuristics the
# compiler gives them AGPR results and copies all of them to VGPRs and back around every step of the k-loop
# (2 x 8 moves per tile and step, and a wait for each result)
FILE_FLAGS = {"fit_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def stale():
    if not os.path.exists(LIB) or not os.path.exists(HOOKS_LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(INCLUDE, h) for h in PUBLIC_HEADERS]
    deps.append(os.path.abspath(__file__))
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    cc = hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and all(os.path.getmtime(obj) > os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
                and all(os.path.getmtime(obj) > os.path.getmtime(os.path.join(INCLUDE, h)) for h in PUBLIC_HEADERS)):
            continue
        cmd = [cc] + FLAGS + FILE_FLAGS.get(s, []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out))
        if verbose and out.strip():
            print(out)
    cmd = [cc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    build_hooks(cc, objs, verbose)
    return LIB


def build_hooks(cc, objs, verbose=False):
    """lib/libhpsdf_hooks.so: the same library with the fault-injection hook of the multi-rank tests compiled in (-DHPSDF_TEST_HOOKS:
    HPSDF_TEST_FAIL_RANK makes one rank's share of a round fail).  Only the two sources that hold the hook are compiled again; the
    production library never parses the variable."""
    hookdir = os.path.join(HERE, "build", "hooks")
    os.makedirs(hookdir, exist_ok=True)
    procs, swapped = [], list(objs)
    for s in HOOK_SOURCES:
        obj = os.path.join(hookdir, s + ".o")
        swapped[SOURCES.index(s)] = obj
        cmd = [cc] + FLAGS + ["-DHPSDF_TEST_HOOKS"] + FILE_FLAGS.get(s, []) + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s (hooks):\n%s" % (s, out))
    r = subprocess.run([cc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", HOOKS_LIB] + swapped, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed (hooks):\n" + r.stdout)
    return HOOKS_LIB


LAB_LIB = os.path.join(LIBDIR, "libhpsdf_lab.so")


def build_variant(name, flags, sources=("kernels.hip",), verbose=False):
    """lib/libhpsdf_<name>.so: the library with `sources` compiled again under extra `flags` -- lab and measurement builds, made ON
    DEMAND only (neither build() nor the tests make or load them; HPSDF_LIBRARY=<name> selects one in the Python loader)."""
    lib = build()
    cc = hipcc()
    out = os.path.join(LIBDIR, "libhpsdf_%s.so" % name)
    if os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(lib):
        return out
    vdir = os.path.join(HERE, "build", "variant_" + name)
    os.makedirs(vdir, exist_ok=True)
    objs = [os.path.join(HERE, "build", s + ".o") for s in SOURCES]
    procs = []
    for s in sources:
        obj = os.path.join(vdir, s + ".o")
        objs[SOURCES.index(s)] = obj
        cmd = [cc] + FLAGS + list(flags) + FILE_FLAGS.get(s, []) + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        o, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s (%s):\n%s" % (s, name, o))
    r = subprocess.run([cc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out] + objs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed (%s):\n%s" % (name, r.stdout))
    return out


def build_lab(verbose=False):
    """lib/libhpsdf_lab.so (python build.py --lab; tools/query_general_floor.py): the library with the Query lab kernels compiled in
    (-DHPSDF_QUERY_LAB_BUILD: HPSDF_QUERY_LAB=1..4 takes one link out of query_general_lds_kernel's chain -- the values it returns are
    then not the tree's).  The production library holds no such kernel and does not read the variable (tests/test_product_cpu.py)."""
    return build_variant("lab", ["-DHPSDF_QUERY_LAB_BUILD"], verbose=verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--lab" in sys.argv:
        print(build_lab(verbose=True))
    for a in sys.argv[1:]:  # --variant=name:-DFLAG,-DFLAG2[:source.hip,source2.hip]   (default: kernels.hip compiled again under the flags)
        if a.startswith("--variant="):
            name, _, rest = a[len("--variant="):].partition(":")
            fl, _, srcs = rest.partition(":")
            print(build_variant(name, [f for f in fl.split(",") if f], tuple(x for x in srcs.split(",") if x) or ("kernels.hip",), verbose=True))
