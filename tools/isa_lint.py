#!/usr/bin/env python3
"""ISA diagnostic for the split-staged convolution tiles -- NOT a correctness rule any more.  Round 3 blamed run-to-run differences under
several HIP streams on packed / 64-bit VALU writes inside the MFMA streams (a presumed VALU-write-after-MFMA-read hazard) and this
tool enforced "none of those".  Round 4 traced every one of those differences to the encoder's VALU stem kernel (tools/soak_probe.py;
with the stem on the matrix cores the build WITH v_pk_mul_f16 inside the streams is bit-stable over 2 400 lane-rounds), so the rule
is withdrawn; the tool stays as a way to see what the compiler puts between the MFMAs.  It compiles one tile of csrc/conv_igemm.hip to gfx950
assembly and reports, for every write into a VGPR inside the kernel's MFMA streams, how many MFMAs were issued since the last MFMA that
read that register as its A or B operand.  VALU writes are the dangerous class (they land within cycles of their issue); LDS returns
and global loads land one memory latency later.  Usage: tools/isa_lint.py "<ConvCfg template arguments>" [more configs ...]
With no arguments: every f16x3 and bf16x6 tile the launcher uses.  LINT_DEFS="-DSS_X6_SPREAD=1" etc. lints another build."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ["3, 3, 3, 4, 4, 2, 1, 8, 1", "3, 3, 3, 4, 2, 2, 2, 4, 1", "3, 3, 3, 4, 2, 1, 2, 4, 1", "1, 3, 3, 8, 4, 2, 1, 8, 1", "1, 3, 3, 8, 2, 2, 2, 2, 1",
          "1, 3, 3, 8, 2, 1, 2, 2, 1", "1, 3, 3, 8, 2, 2, 1, 4, 1", "1, 1, 1, 32, 4, 2, 1, 4, 8", "1, 1, 1, 32, 2, 2, 2, 2, 4", "1, 1, 1, 32, 2, 2, 1, 4, 8",
          "1, 1, 1, 32, 4, 2, 2, 4, 8"]


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def lint(cfg):
    src = open(os.path.join(ROOT, "stem-seg_amd", "csrc", "conv_igemm.hip")).read()
    head = src[:src.index("// split-K epilogue: out[c,t,y,x]")].replace('#include "common.h"', '#include "%s/stem-seg_amd/csrc/common.h"' % ROOT)
    with tempfile.TemporaryDirectory() as td:
        hipf, asm = os.path.join(td, "t.hip"), os.path.join(td, "t.s")
        open(hipf, "w").write(head + "\nusing YT = ConvCfg<%s>;\ntemplate __global__ void conv_igemm_kernel<YT>(const ConvKParams);\n}\n" % cfg)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value"] + os.environ.get("LINT_DEFS", "").split() + [hipf, "-o", asm],
                       check=True, capture_output=True)
        lines = [l.strip() for l in open(asm) if l.strip() and not l.strip().startswith(";")]
    # segments = maximal runs between barriers / labels / branches; a "stream" = first .. last MFMA of a segment with >= 6 MFMAs
    segs, cur = [], []
    for l in lines:
        op = l.split()[0]
        if op == "s_barrier" or op.endswith(":") or op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm"):
            segs.append(cur)
            cur = []
        else:
            cur.append(l)
    segs.append(cur)
    fam = lambda op: ("packed" if op.startswith("v_pk_") or "_mix" in op or op.startswith("v_dot") else
                      "wide" if re.search(r"_(u64|i64|f64|b64)", op) else "plain")
    worst, counts = {}, {}
    for sgm in segs:
        idx = [i for i, l in enumerate(sgm) if l.startswith("v_mfma")]
        if len(idx) < 6:
            continue
        counts["streams"] = counts.get("streams", 0) + 1
        n_mfma, last_read = 0, {}
        for l in sgm[idx[0]:idx[-1] + 1]:
            op = l.split()[0]
            args = l[len(op):].split(",")
            if op.startswith("v_mfma"):
                n_mfma += 1
                for a in args[1:3]:
                    for r in regs(a):
                        last_read[r] = n_mfma
                continue
            kind = ("valu-" + fam(op)) if op.startswith("v_") else "lds-return" if op.startswith("ds_read") else \
                   "vmem-return" if op.startswith(("global_load", "buffer_load")) else None
            if kind is None:
                continue
            counts[kind] = counts.get(kind, 0) + 1
            for r in regs(args[0]):
                if r in last_read:
                    d = n_mfma - last_read[r]
                    if d < worst.get(kind, (10 ** 9, ""))[0]:
                        worst[kind] = (d, l)
    return worst, counts


if __name__ == "__main__":
    cfgs = sys.argv[1:] or ["%s, false, %d" % (s, bf) for bf in (3, 2) for s in SHAPES]
    print("Inside the MFMA streams (first .. last MFMA between two barriers): instructions that write VGPRs, by class, and the smallest number of")
    print("MFMAs issued between an MFMA that read a register as its A / B operand and a write into that register (0 = the very next MFMA slot)")
    for c in cfgs:
        w, n = lint(c)
        print("ConvCfg<%s>" % c)
        print("   %d MFMA streams" % n.pop("streams", 0))
        for k in sorted(n):
            d, l = w.get(k, (None, ""))
            print("   %-12s %4d in the streams; closest write behind a reader: %s" % (k, n[k], "none" if d is None else "%d MFMAs   %s" % (d, l[:64])))
