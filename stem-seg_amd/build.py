#!/usr/bin/env python3
"""Builds libstemseg_hip.so (gfx950 only) in-tree with one hipcc invocation per source file.

    python stem-seg_amd/build.py [--force]

The .so lands in stem-seg_amd/stemseg_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "stemseg_amd", "lib")
OBJDIR = os.path.join(CSRC, "build")
LIB = os.path.join(LIBDIR, "libstemseg_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# A/B builds (tools/): STEMSEG_BUILD_DEFINES="-DSS_F16_WPLANES=3" STEMSEG_BUILD_TAG=w3 -> lib/libstemseg_hip_w3.so (objects under build_w3/);
# select at run time with STEMSEG_HIP_LIB=<path>
TAG = os.environ.get("STEMSEG_BUILD_TAG", "")
FLAGS += os.environ.get("STEMSEG_BUILD_DEFINES", "").split()
if TAG:
    OBJDIR = os.path.join(CSRC, "build_" + TAG)
    LIB = os.path.join(LIBDIR, "libstemseg_hip_%s.so" % TAG)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "stemseg_hip.h")]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and (f.endswith((".hip", ".h"))):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(OBJDIR, "libstemseg_hip.stamp")      # csrc/build/ is git-ignored
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        if verbose:
            print("[build] %s is up to date" % os.path.relpath(LIB, os.path.dirname(HERE)))
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr.strip())
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("[build] built %s (%d kB)" % (os.path.relpath(LIB, os.path.dirname(HERE)), os.path.getsize(LIB) // 1024))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
