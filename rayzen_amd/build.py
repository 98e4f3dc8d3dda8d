"""Builds the two in-tree shared libraries of the product:

  rayzen_amd/lib/librayzen_hip.so   HIP kernels + the C-ABI of include/rayzen_hip.h
                                    (hipcc, --offload-arch=gfx950, cross-compiles without a GPU)
  rayzen_amd/lib/librayzen_host.so  host-side scene / BVH builders, include/rayzen_host.h (g++)

`python -m rayzen_amd.build` or rayzen_amd.build.build_all().  Both land in
the source tree so they travel to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rayzen_amd")
LIB = os.path.join(PKG, "lib")
INC = os.path.join(ROOT, "include")
HIP_DIR = os.path.join(PKG, "csrc", "hip")
HOST_DIR = os.path.join(PKG, "csrc", "host")

HIP_SO = os.path.join(LIB, "librayzen_hip.so")
HOST_SO = os.path.join(LIB, "librayzen_host.so")

# -ffp-contract=off is part of the numerics contract (rz_device_math.h), not a tuning flag.
# -fno-slp-vectorize: left on, the SLP vectoriser pairs neighbouring f32 adds/multiplies into v_pk_*_f32 on 64-bit
# register tuples; in the render kernel that cost 25 VGPRs (167 vs 142) for no gain in issue slots.  Where packed math
# does pay (the two-plane slab test) it is written explicitly (rz_trace.h).
# -mllvm -greedy-regclass-priority-trumps-globalness (round 4): of sixteen register-allocator / scheduler switches tried on the
# render kernels (profiles/r04_regs/) the only one that moved anything the right way: 39 -> 34 spilled VGPRs in the C2 kernel,
# C2 10.88 -> 10.79 ms, C4 neutral.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-fast-math", "-fno-slp-vectorize", "-mllvm", "-greedy-regclass-priority-trumps-globalness",
               "-Wall", "-Wno-unused-function"]
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _files(d, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


STAMP_MARK = b"RZSRCHASH:"      # rz_context.hip keeps "RZSRCHASH:<64 hex digits>" in the library's read-only data


def stamped_hash(so_path):
    """The source hash a librayzen_hip*.so was built from, read from the file (the library is not loaded); None if the
    file is missing or carries no stamp."""
    try:
        blob = open(so_path, "rb").read()
    except OSError:
        return None
    k = blob.find(STAMP_MARK)
    if k < 0:
        return None
    h = blob[k + len(STAMP_MARK):k + len(STAMP_MARK) + 64]
    return h.decode() if len(h) == 64 and all(c in b"0123456789abcdef" for c in h) else None


def build_hip(force=False, verbose=False, extra_flags=()):
    """Builds librayzen_hip.so unless the one in the tree was built from exactly these sources and flags: the decision is
    taken on the source hash compiled into the library (rz_source_hash()), not on file times -- a stale library beside
    fresh sources is rebuilt whatever its mtime says."""
    os.makedirs(LIB, exist_ok=True)
    srcs = _files(HIP_DIR, (".hip",))
    want = source_hash(extra_flags)
    if not force and stamped_hash(HIP_SO) == want:
        return HIP_SO
    # one object per source, compiled side by side (rz_kernels.hip alone is 80 s of the 115 s a single hipcc command takes), then linked
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    common = [hipcc_path()] + HIPCC_FLAGS + list(extra_flags) + [f'-DRZ_SOURCE_HASH="{want}"', "-I", INC, "-I", HIP_DIR]
    with tempfile.TemporaryDirectory(prefix="rz_build_") as tmp:
        objs = [os.path.join(tmp, os.path.basename(f) + ".o") for f in srcs]

        def one(job):
            src, obj = job
            cmd = common + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)

        with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
            list(ex.map(one, zip(srcs, objs)))
        link = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", HIP_SO + ".tmp"] + objs + ["-ldl"]
        if verbose:
            print(" ".join(link), file=sys.stderr)
        subprocess.check_call(link)
        os.replace(HIP_SO + ".tmp", HIP_SO)      # (a reader never sees a half-written library)
    return HIP_SO


def build_host(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    srcs = _files(HOST_DIR, (".cpp",))
    deps = srcs + _files(HOST_DIR, (".h",)) + _files(INC, (".h",)) + [os.path.abspath(__file__)]
    if not force and not _newer(HOST_SO, deps):
        return HOST_SO
    cmd = [os.environ.get("CXX", "g++")] + CXX_FLAGS + ["-I", INC, "-I", HOST_DIR, "-shared", "-o", HOST_SO] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HOST_SO


def source_hash(extra_flags=()):
    """sha256 over everything librayzen_hip.so is built from (sources, headers, flags).  It is compiled into the library
    (-DRZ_SOURCE_HASH; rz_source_hash() returns it), so a committed rocprofv3 counter file (profiles/), the tree and the
    LOADED library can be tied to one another."""
    import hashlib
    h = hashlib.sha256()
    for f in _files(HIP_DIR, (".hip", ".h")) + _files(INC, (".h",)):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(" ".join(HIPCC_FLAGS + list(extra_flags)).encode())
    return h.hexdigest()


def build_all(force=False, verbose=False):
    return build_host(force, verbose), build_hip(force, verbose)


if __name__ == "__main__":
    if "--hash" in sys.argv:
        print(source_hash())
        sys.exit(0)
    build_all(force="--force" in sys.argv, verbose=True)
    print("built", HOST_SO, HIP_SO)
