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
    "k_solve_lite.hip": [],
    "k_misc.hip": [],
}


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libgauss_hip.so cannot be built")


def source_hash():
    """sha256 (first 16 hex digits) over the sources of libgauss_hip.so, names and contents: the identity that
    profiles/*_provenance.json records and that bench.py compares before it quotes a profile-derived number."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")))
    for f in names + ["../../include/gauss_hip.h"]:
        p = os.path.join(CSRC, f)
        h.update(os.path.basename(f).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def git_head():
    try:
        head = subprocess.check_output(["git", "-C", os.path.join(HERE, ".."), "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        dirty = bool(subprocess.check_output(["git", "-C", os.path.join(HERE, ".."), "status", "--porcelain", "--untracked-files=no"],
                                             stderr=subprocess.DEVNULL).decode().strip())
        return head, dirty
    except Exception:
        return None, None


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, "gauss_internal.h"), os.path.join(CSRC, "k_gram_common.h"), os.path.join(CSRC, "k_solve_common.h"), os.path.join(HERE, "..", "include", "gauss_hip.h")]
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
    # the source hash is compiled into the library (gauss_hip_source_hash): one tiny unit, rebuilt when the hash changes
    import json
    sh = source_hash()
    stamp_path = os.path.join(LIBDIR, "build_stamp.json")
    try:
        with open(stamp_path) as fh:
            old = json.load(fh)
    except Exception:
        old = {}
    vobj = os.path.join(OBJDIR, "gauss_version.o")
    objs.append(vobj)
    if force or old.get("csrc_hash") != sh or not os.path.exists(vobj):
        vsrc = os.path.join(OBJDIR, "gauss_version.cpp")
        with open(vsrc, "w") as fh:
            fh.write('extern "C" const char* gauss_hip_source_hash(void) { return "%s"; }\n' % sh)
        subprocess.check_call(["g++", "-O1", "-fPIC", "-c", vsrc, "-o", vobj])
    so = os.path.join(LIBDIR, "libgauss_hip.so")
    if force or _newer(so, objs):
        subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", so] + objs)
    head, dirty = git_head()
    if head is None:            # no git here (the GPU box): keep what the development container wrote for these sources
        head, dirty = (old.get("git_head"), old.get("git_dirty")) if old.get("csrc_hash") == sh else (None, None)
    with open(stamp_path, "w") as fh:
        json.dump({"csrc_hash": sh, "git_head": head, "git_dirty": dirty}, fh)
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
