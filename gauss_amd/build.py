"""Build libgauss_hip.so (HIP kernels + C ABI) and libgauss_host.so in-tree with hipcc / g++.

The shared objects land in gauss_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
hipcc cross-compiles gfx950 code objects without a GPU present.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
ARCH = "gfx950"

# translation unit -> extra flags
HIP_UNITS = {
    "gauss_hip.cpp": [],                         # host only: planner, jobs, C ABI
    "k_gram.hip": [],
    "k_pack_epilogue.hip": ["-ffp-contract=off"],   # reference-order fp64 tails: no fused multiply-add
    "k_solve.hip": [],
    "k_misc.hip": [],
}


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libgauss_hip.so cannot be built")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, "gauss_internal.h"), os.path.join(CSRC, "k_gram_common.h"), os.path.join(HERE, "..", "include", "gauss_hip.h")]
    objs = []
    for unit, extra in HIP_UNITS.items():
        src = os.path.join(CSRC, unit)
        obj = os.path.join(OBJDIR, os.path.splitext(unit)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
                   "-Wno-unused-result", "-Wno-unused-value"] + extra
            if unit.endswith(".hip"):
                cmd.insert(3, f"--offload-arch={ARCH}")
            else:   # plain host C++ against the HIP runtime API (no device pass)
                cmd[1:1] = ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    so = os.path.join(LIBDIR, "libgauss_hip.so")
    if force or _newer(so, objs):
        subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", so] + objs)
    return so


def build_host(force=False, verbose=False):
    """libgauss_host.so: host data layer + the five reference entry points (plain g++, zlib)."""
    os.makedirs(OBJDIR, exist_ok=True)
    hip_so = build_hip(force=False, verbose=verbose)
    hdir = os.path.join(CSRC, "host")
    srcs = [os.path.join(hdir, f) for f in ("gauss_host.cpp", "bgzf_io.cpp", "packed_panel.cpp")]
    deps = srcs + [os.path.join(hdir, "bgzf_io.h"), os.path.join(hdir, "packed_panel.h"), os.path.join(HERE, "..", "include", "gauss_host.h"),
                   os.path.join(HERE, "..", "include", "gauss_hip.h"), hip_so]
    so = os.path.join(LIBDIR, "libgauss_host.so")
    if force or _newer(so, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", so] + srcs + [
            "-L" + LIBDIR, "-lgauss_hip", "-Wl,-rpath,$ORIGIN", "-lz", "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
