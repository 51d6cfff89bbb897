#!/usr/bin/env python3
"""Builds a stand-alone timing probe of the bf16x6 conv kernel (big 3x3x3 tile on the block_4x shape) in four variants that
strip parts of the chunk loop, to see where the MFMA pipe idles: V0 as shipped; V1 no global prefetch / LDS staging inside the
loop (barriers kept); V2 = V1 without the barriers; V3 = V2 with the operand fragments loaded once (MFMA stream only).
Results are garbage by construction -- only the time matters.  Usage: python tools/microbench/mk_x6_probe.py  (writes
/tmp/x6probe_V*.hip and builds gpurun_out/x6probe_V*)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "stem-seg_amd", "csrc", "conv_igemm.hip")).read()
head = src[:src.index("// split-K epilogue: out[c,t,y,x]")]
head = head.replace('#include "common.h"', '#include "%s/stem-seg_amd/csrc/common.h"' % ROOT)
cfg = sys.argv[1] if len(sys.argv) > 1 else "3, 3, 3, 4, 4, 2, 1, 8, 1, false, 2"
main = r'''
using YT = ConvCfg<%s>;
}
namespace stemseg { void set_error(const char*, ...) {} }
int main() {
    using namespace stemseg;
    const int Cin = 256, Cout = 128, T = 8, H = 120, W = 216;
    PaddedGeom g(Cin, T, H, W);
    float *in, *out, *wp, *bias;
    hipMalloc(&in, g.total * 4); hipMemset(in, 0, g.total * 4);
    hipMalloc(&out, (size_t)Cout * T * H * W * 4);
    const size_t wbytes = (size_t)(Cin / 4) * 7 * 3 * 2 * Cout * 16;
    hipMalloc(&wp, wbytes); hipMemset(wp, 0, wbytes);
    hipMalloc(&bias, Cout * 4); hipMemset(bias, 0, Cout * 4);
    ConvKParams p;
    memset(&p, 0, sizeof(p));
    p.in = in; p.in_cs = g.cs; p.in_ts = g.ts; p.in_ys = g.pitch; p.in_limit = g.total; p.in_H = H + 2;
    p.wpk = wp; p.bias = bias; p.out = out; p.out_cs = (int64_t)T * H * W; p.out_ts = (int64_t)H * W; p.out_ys = W;
    p.Cin = Cin; p.Cout = Cout; p.T = T; p.H = H; p.W = W; p.vec4 = 1; p.vec_epi = 1; p.t_fastest = 1; p.n_co = 1;
    p.tiles_x = (W + 31) / 32; p.tiles_y = (H + YT::ROWS - 1) / YT::ROWS; p.chunks_per_split = Cin / YT::CK;
    dim3 grid(p.tiles_x * p.tiles_y * T * p.n_co);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(conv_igemm_kernel<YT>, grid, dim3(YT::NTHREADS), 0, 0, p);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(conv_igemm_kernel<YT>, grid, dim3(YT::NTHREADS), 0, 0, p);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("variant %%d: %%.1f us per launch, %%.1f TFLOP/s-equivalent (%%d workgroups)\n", X6_VARIANT, ms * 1e3, 2.0 * Cin * 27 * Cout * T * H * W / ms / 1e9, (int)grid.x);
    return 0;
}
''' % cfg
os.makedirs(os.path.join(ROOT, "tools", "microbench", "bin"), exist_ok=True)
for v in range(4):
    body = head
    if v >= 1:      # no prefetch / staging inside the loop
        body = body.replace("if (more) fetch_w6(c0 + C::CK, Q0{}, QA{});", "").replace("                fetch_w6(c0 + C::CK, Q0{}, QA{});\n", "")
        body = body.replace("store_w6(Q0{}, QA{});", "").replace("fetch_w6(c0 + C::CK, QA{}, QE{});", "").replace("store_w6(QA{}, QE{});", "")
        body = body.replace("for (int k = 0; k < IN_PT6; ++k) fetch_in6_fast(c0 + C::CK, k, rin[2 * k], rin[2 * k + 1]);", "for (int k = 0; k < 0; ++k) {}")
        body = body.replace("for (int k = 0; k < IN_PT6; ++k) { const int q = tid + k * C::NTHREADS; if (q < NQ6) store_in6(q, rin[2 * k], rin[2 * k + 1]); }", "for (int k = 0; k < 0; ++k) {}")
    if v >= 2:      # no barriers in the chunk loop
        a = body.index("        for (int c0 = c_begin; c0 < c_end; c0 += C::CK) {\n            const bool more = c0 + C::CK < c_end;\n            if (more) {")
        b = body.index("    } else if constexpr (C::DB && C::GL) {")
        body = body[:a] + body[a:b].replace("__syncthreads();", "") + body[b:]
    if v >= 3:      # fragments loaded once
        body = body.replace("            if (st + 1 < NSTEP) ld_a(g0 + (st + 1) / C::MI, (st + 1) % C::MI, a1);", "            if (st == 0) ld_a(g0, 1, a1);")
        body = body.replace("                if ((st + 1) % C::MI == 0) ld_b(g0 + (st + 1) / C::MI, bfr);\n", "")
        body = body.replace("                if (st + 2 < NSTEP) ld_a(g0 + (st + 2) / C::MI, (st + 2) % C::MI, a0);\n", "")
        body = body.replace("                if (st + 2 < NSTEP && (st + 2) % C::MI == 0) ld_b(g0 + (st + 2) / C::MI, bfr);\n", "")
    path = "/tmp/x6probe_V%d.hip" % v
    open(path, "w").write("#define X6_VARIANT %d\n" % v + body + main)
    exe = os.path.join(ROOT, "tools", "microbench", "bin", "x6probe_V%d" % v)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-unused-value", path, "-o", exe],
                       capture_output=True, text=True)
    print("V%d build %s" % (v, "ok" if r.returncode == 0 else "FAILED\n" + r.stderr[-1500:]))
