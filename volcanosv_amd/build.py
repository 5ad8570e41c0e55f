"""Builds the gfx950 shared library in-tree: volcanosv_amd/libvolcanosv_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU
box with the gpurun snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["cigar_scan.hip", "radix_sort.hip", "sig_stages.hip", "slim_path.hip", "bnd.hip", "support.hip", "cutesv.hip", "redundancy.hip", "inflate.hip", "bam_device.hip", "capi.hip", "bam_ingest.cc"]
LIB = os.path.join(HERE, "libvolcanosv_hip.so")
# -ffp-contract=off: the ONT/CLR split rule compares fp64 products exactly as CPython does
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; the HIP extension cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "volcanosv.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return LIB
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o").replace(".cc", ".o"))
        cmd = [hipcc()] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
