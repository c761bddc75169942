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
EXPORTS_MAP = os.path.join(CSRC, "exports.map")     # both libraries export `gauss_*` only

# translation unit -> extra flags
HIP_UNITS = {
    # host only: contexts + queue registry, planner, runs, row stores, C ABI
    "gauss_ctx.cpp": [],
    "gauss_plan.cpp": [],
    "gauss_run.cpp": [],
    "gauss_store.cpp": [],
    "gauss_abi.cpp": [],
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


_HIPCC_VERSION = None


def hipcc_version():
    """First line of `hipcc --version` that names the compiler (part of the build identity)."""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            out = subprocess.check_output([_hipcc(), "--version"], stderr=subprocess.STDOUT).decode()
            _HIPCC_VERSION = " ".join(l.strip() for l in out.splitlines() if "version" in l.lower())[:200]
        except Exception:
            _HIPCC_VERSION = "unknown"
    return _HIPCC_VERSION


def _unit_cmd(hipcc, unit, src, obj):
    # -fvisibility=hidden: the library exports the C ABI of include/gauss_hip.h and nothing else (gauss_job.h brackets the header with
    # a visibility pragma); loaded into an R process beside other packages, un-namespaced internals (fail, trace_on) would collide
    cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-c", src, "-o", obj,
           "-Wno-unused-result", "-Wno-unused-value"] + HIP_UNITS[unit] + EXTRA_FLAGS
    if unit.endswith(".hip"):
        cmd.insert(3, f"--offload-arch={ARCH}")
    else:   # plain host C++ against the HIP runtime API (no device pass)
        cmd[1:1] = ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
    return cmd


# -D overrides for experiments (GAUSS_HIPCC_FLAGS="-DGAUSS_GRAM_EDGE16=0"): part of every unit's command line AND of the hash
EXTRA_FLAGS = os.environ.get("GAUSS_HIPCC_FLAGS", "").split()


def source_hash():
    """sha256 (first 16 hex digits) over everything that decides what libgauss_hip.so contains: the sources (names and
    contents), every unit's compile flags (HIP_UNITS, GAUSS_HIPCC_FLAGS), the target architecture and the compiler's
    version.  profiles/*_provenance.json records it and bench.py compares it before it quotes a profile-derived number:
    a library built from the same sources with other flags is a different library."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h", ".map")))
    for f in names + ["../../include/gauss_hip.h"]:
        p = os.path.join(CSRC, f)
        h.update(os.path.basename(f).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
    for unit in sorted(HIP_UNITS):
        h.update(("|" + unit + ":" + " ".join(_unit_cmd("hipcc", unit, unit, unit + ".o"))).encode())
    h.update(("|arch=" + ARCH + "|cc=" + hipcc_version()).encode())
    return h.hexdigest()[:16]


def _unit_hash(unit, hdrs):
    """Identity of one object file: its source, the shared headers, its command line, the compiler."""
    import hashlib
    h = hashlib.sha256()
    for p in [os.path.join(CSRC, unit)] + hdrs:
        with open(p, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(_unit_cmd("hipcc", unit, unit, unit + ".o")).encode())
    h.update(hipcc_version().encode())
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


def _run_all(jobs, verbose=False):
    """[(name, command)] side by side on the container's cores; the first failure is raised after the others have ended."""
    if not jobs:
        return
    from concurrent.futures import ThreadPoolExecutor
    def one(job):
        name, cmd = job
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return name
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), (os.cpu_count() or 2), 8))) as pool:
        list(pool.map(one, jobs))


def build_hip(force=False, verbose=False):
    import json
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, "gauss_internal.h"), os.path.join(CSRC, "gauss_job.h"), os.path.join(CSRC, "k_gram_common.h"), os.path.join(CSRC, "k_solve_common.h"), os.path.join(HERE, "..", "include", "gauss_hip.h")]
    # an object is rebuilt when its recorded identity (source + headers + command line + compiler) differs -- not by mtime: a
    # flag change or a checkout that restores old timestamps must not link stale objects under a fresh source hash
    ids_path = os.path.join(OBJDIR, "unit_ids.json")
    try:
        with open(ids_path) as fh:
            ids = json.load(fh)
    except Exception:
        ids = {}
    objs = []
    todo = []
    for unit in HIP_UNITS:
        src = os.path.join(CSRC, unit)
        obj = os.path.join(OBJDIR, os.path.splitext(unit)[0] + ".o")
        objs.append(obj)
        uid = _unit_hash(unit, hdrs)
        if force or not os.path.exists(obj) or ids.get(unit) != uid:
            todo.append((unit, uid, _unit_cmd(hipcc, unit, src, obj)))
    relink = bool(todo)
    # the units are independent: compiled side by side (a fresh tree: 38 s one after the other), each recorded as it completes
    _run_all([(unit, cmd) for unit, _, cmd in todo], verbose)
    for unit, uid, _ in todo:
        ids[unit] = uid
    if todo:
        with open(ids_path, "w") as fh:
            json.dump(ids, fh)
    # the source hash is compiled into the library (gauss_hip_source_hash): one tiny unit, rebuilt when the hash changes
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
            fh.write('extern "C" __attribute__((visibility("default"))) const char* gauss_hip_source_hash(void) { return "%s"; }\n' % sh)
        subprocess.check_call(["g++", "-O1", "-fPIC", "-c", vsrc, "-o", vobj])
    so = os.path.join(LIBDIR, "libgauss_hip.so")
    if force or relink or _newer(so, objs):
        subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,--version-script=" + EXPORTS_MAP, "-o", so] + objs)
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
    # (host_internal.h: what the four host_*.cpp share; -fvisibility=hidden: only the C ABI of include/gauss_host.h is exported)
    srcs = [os.path.join(hdir, f) for f in ("host_feeder.cpp", "host_tables.cpp", "host_calls.cpp", "host_chrom.cpp", "bgzf_io.cpp", "packed_panel.cpp")]
    deps = srcs + [EXPORTS_MAP, os.path.join(hdir, "host_internal.h"), os.path.join(hdir, "bgzf_io.h"), os.path.join(hdir, "packed_panel.h"), os.path.join(HERE, "..", "include", "gauss_host.h"),
                   os.path.join(HERE, "..", "include", "gauss_hip.h"), hip_so]
    so = os.path.join(LIBDIR, "libgauss_host.so")
    if force or _newer(so, deps):
        flags = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fvisibility-inlines-hidden", "-Wall"]
        objs = [os.path.join(OBJDIR, "host_" + os.path.splitext(os.path.basename(f))[0] + ".o") for f in srcs]
        _run_all([(os.path.basename(f), ["g++"] + flags + ["-c", f, "-o", o]) for f, o in zip(srcs, objs)], verbose)
        cmd = ["g++", "-shared", "-Wl,--version-script=" + EXPORTS_MAP, "-o", so] + objs + ["-L" + LIBDIR, "-lgauss_hip", "-Wl,-rpath,$ORIGIN", "-lz", "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
