#!/usr/bin/env python3
"""Lists every kernel of the built library with its private-segment (scratch) size, spill counts and register use, from the code
objects' metadata (CPU only: works on the objects hipcc cross-compiled).  Exit code 1 when a kernel uses scratch memory.

    python tools/scratch_scan.py [--all]

Why it matters: spills are slow, and kernels that keep state in scratch memory were the first suspect for the run-to-run differences
of rounds 3 / 4 (they turned out not to be the cause).  The shipped modes' kernels use none; the listing
names the few non-default tiles that do."""
import glob
import os
import re
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def code_object(obj, tmp):
    out = os.path.join(tmp, os.path.basename(obj) + ".co")
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj, "--targets=" + TARGET, "--output=" + out],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        fb = out + ".fatbin"
        subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb], check=True)
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fb, "--targets=" + TARGET, "--output=" + out],
                           capture_output=True, text=True)
        if r.returncode != 0:
            return None
    return out


def kernels(co):
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    m = re.search(r"^\s*---\n(.*?)^\s*\.\.\.", notes, re.S | re.M)
    if not m:
        return []
    return yaml.safe_load(m.group(1)).get("amdhsa.kernels", [])


def scan(objs=None):
    objs = objs or sorted(glob.glob(os.path.join(ROOT, "stem-seg_amd", "csrc", "build", "*.o")))
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            co = code_object(o, tmp)
            if co is None:
                continue                                      # host-only object (no device code)
            for k in kernels(co):
                name = subprocess.run(["c++filt", k[".name"]], capture_output=True, text=True).stdout.strip()
                rows.append(dict(obj=os.path.basename(o), name=name, scratch=int(k.get(".private_segment_fixed_size", 0)),
                                 dynamic_stack=bool(k.get(".uses_dynamic_stack", False)),
                                 vgpr_spill=int(k.get(".vgpr_spill_count", 0)), sgpr_spill=int(k.get(".sgpr_spill_count", 0)),
                                 vgpr=int(k.get(".vgpr_count", 0)), agpr=int(k.get(".agpr_count", 0)), lds=int(k.get(".group_segment_fixed_size", 0))))
    return rows


if __name__ == "__main__":
    rows = scan()
    bad = [r for r in rows if r["scratch"] > 0 or r["dynamic_stack"]]
    show = rows if "--all" in sys.argv else bad
    for r in sorted(show, key=lambda r: (-r["scratch"], r["name"])):
        print("%-14s scratch %5d B  vgpr-spills %3d  sgpr-spills %3d  vgpr %3d agpr %3d lds %6d  %s" % (
            r["obj"], r["scratch"], r["vgpr_spill"], r["sgpr_spill"], r["vgpr"], r["agpr"], r["lds"], r["name"][:150]))
    print("%d kernels, %d with scratch memory" % (len(rows), len(bad)))
    sys.exit(1 if bad else 0)
