"""Build libnesvor_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake.

    python -m nesvor_amd.csrc.build        # or: python nesvor_amd/csrc/build.py

The library is written to nesvor_amd/lib/libnesvor_hip.so (git-ignored, shipped to
the GPU box by gpurun).  Rebuilds only when a source is newer than the library.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
LIB = os.path.join(LIB_DIR, "libnesvor_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "nesvor_hip.h"))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
            os.path.getmtime(src),
            max(os.path.getmtime(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".h")),
            os.path.getmtime(os.path.join(ROOT, "include", "nesvor_hip.h")),
            os.path.getmtime(os.path.abspath(__file__)),
        ):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", *FLAGS, "-I", os.path.join(ROOT, "include"), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr, flush=True)  # stdout stays clean for bench.py's JSON line
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr, flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
