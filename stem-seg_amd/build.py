#!/usr/bin/env python3
"""Builds libstemseg_hip.so (gfx950 only) in-tree with one hipcc invocation per source file.

    python stem-seg_amd/build.py [--force]

The .so lands in stem-seg_amd/stemseg_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import hashlib
import json
import os
import re
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
# The product library contains NO packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  Round 5 reproduced the lane
# differences of rounds 3-4 with two bare kernels (tools/graph_corun_probe.py): a wave whose FMAs are v_pk_fma_f32
# gets wrong LOW halves in lanes 48..63 when it shares a CU with the f16x3 128 x 128 1x1 convolution -- 100 % of the launches on every
# box tried -- and the same kernel compiled with scalar v_fma_f32 does not (0 of 400).  No kernel of the product ever showed it, but the
# instruction class is cheap to do without: 0.5 % of the step (interleaved A/B, one box), results bit-identical (an fma is an fma).
# Experiment builds (-DSS_EXPERIMENTS: the VALU stem the probes need) keep the compiler's default.
NO_PACKED_FP32 = "-DSS_EXPERIMENTS" not in FLAGS and os.environ.get("STEMSEG_BUILD_PACKED_FP32", "0") != "1"
if NO_PACKED_FP32:
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS += ["-save-temps=obj"]           # keeps the device assembly next to the objects: the ISA report below reads it
PACKED_FP32 = re.compile(r"^\s+(v_pk_(?:fma|mul|add)_f32)\b")
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


def isa_report():
    """{source: {kernels, packed_fp32_valu_instructions, by_kernel}} from the device assembly -save-temps left in the object directory."""
    out = {}
    for src in _sources():
        path = os.path.join(OBJDIR, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        kernels, by_kernel, cur = 0, {}, None
        if os.path.exists(path):
            for line in open(path, errors="replace"):
                m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
                if m and not line.startswith("."):
                    cur = m.group(1)
                if ".amdhsa_kernel " in line:
                    kernels += 1
                m = PACKED_FP32.match(line)
                if m and cur:
                    by_kernel[cur] = by_kernel.get(cur, 0) + 1
        out[src] = {"kernels": kernels, "packed_fp32_valu_instructions": sum(by_kernel.values()), "kernels_with_packed_fp32": len(by_kernel),
                    "assembly_found": os.path.exists(path)}
    return {"arch": ARCH, "no_packed_fp32_flag": NO_PACKED_FP32, "sources": out}


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
        # (the feature flag reaches the host compilation too, which does not know it and says so: not a finding)
        err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l).strip()
        if verbose and err:
            print(err)
        stem = os.path.join(OBJDIR, src.replace(".hip", ""))
        for ext in ("-hip-amdgcn-amd-amdhsa-gfx950.hipi", "-hip-amdgcn-amd-amdhsa-gfx950.bc", "-host-x86_64-unknown-linux-gnu.hipi",
                    "-host-x86_64-unknown-linux-gnu.bc", "-host-x86_64-unknown-linux-gnu.s"):
            if os.path.exists(stem + ext):
                os.remove(stem + ext)          # (-save-temps leftovers nobody reads: ~10 MB per source)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    report = isa_report()
    with open(os.path.splitext(LIB)[0] + ".isa.json", "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    n_pk = sum(v["packed_fp32_valu_instructions"] for v in report["sources"].values())
    if NO_PACKED_FP32 and n_pk:
        raise RuntimeError("the product library must not contain packed-fp32 VALU instructions, found %d: %s" % (n_pk, report))
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("[build] built %s (%d kB)" % (os.path.relpath(LIB, os.path.dirname(HERE)), os.path.getsize(LIB) // 1024))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
