#!/usr/bin/env python3
"""Builds a stand-alone timing probe of the bf16x6 conv kernel (big 3x3x3 tile on the block_4x shape) in four variants that
strip parts of the chunk loop, to see where the MFMA pipe idles: V0 as shipped; V1 no global prefetch / LDS staging inside the
loop (barriers kept); V2 = V1 without the barriers; V3 = V2 with the operand fragments loaded once (MFMA stream only).
Results are garbage by construction -- only the time matters.  Usage: python tools/microbench/mk_x6_probe.py  (writes
/tmp/<tag>_V*.hip and builds tools/microbench/bin/<tag>_V*; the text surgery of V1..V3 targets the two-phase loop,
PROBE_VARIANTS=0 PROBE_DEFS=-DSS_X6_WMODE_SMALLG=<0|1|2> times the shipped kernel under another weight-staging mode)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "stem-seg_amd", "csrc", "conv_igemm.hip")).read()
head = src[:src.index("// split-K epilogue: out[c,t,y,x]")]
head = head.replace('#include "common.h"', '#include "%s/stem-seg_amd/csrc/common.h"' % ROOT)
cfg = sys.argv[1] if len(sys.argv) > 1 else "3, 3, 3, 4, 4, 2, 1, 8, 1, false, 2"
shape = sys.argv[2] if len(sys.argv) > 2 else "256, 128, 8, 120, 216"          # Cin, Cout, T, H, W
tag = sys.argv[3] if len(sys.argv) > 3 else "x6probe"
main = r'''
using YT = ConvCfg<%s>;
}
namespace stemseg { void set_error(const char*, ...) {} }
int main() {
    using namespace stemseg;
    const int shp[5] = {%s};
    const int Cin = shp[0], Cout = shp[1], T = shp[2], H = shp[3], W = shp[4];
    constexpr int KT = YT::KT, TAPS = YT::TAPS;
    struct { int64_t pitch, ts, cs, total; } g;
    g.pitch = (W + 2 + 3) / 4 * 4; g.ts = (int64_t)(H + 2) * g.pitch; g.cs = (int64_t)(T + KT - 1) * g.ts; g.total = (int64_t)Cin * g.cs + 64;
    float *in, *out, *wp, *bias;
    hipMalloc(&in, g.total * 4); hipMemset(in, 0, g.total * 4);
    hipMalloc(&out, (size_t)Cout * T * H * W * 4);
    const size_t wbytes = (size_t)((Cin + YT::CK - 1) / YT::CK) * YT::G * 3 * 2 * Cout * 16;
    hipMalloc(&wp, wbytes); hipMemset(wp, 0, wbytes);
    hipMalloc(&bias, Cout * 4); hipMemset(bias, 0, Cout * 4);
    ConvKParams p;
    memset(&p, 0, sizeof(p));
    p.in = in; p.in_cs = g.cs; p.in_ts = g.ts; p.in_ys = g.pitch; p.in_limit = g.total; p.in_H = H + 2;
    if (getenv("PROBE_CS")) p.in_cs = atoll(getenv("PROBE_CS"));      // channel stride override (floats): a small one makes the input L2-resident
    p.wpk = wp; p.bias = bias; p.out = out; p.out_cs = (int64_t)T * H * W; p.out_ts = (int64_t)H * W; p.out_ys = W;
    p.Cin = Cin; p.Cout = Cout; p.T = T; p.H = H; p.W = W; p.vec4 = 1; p.vec_epi = 1; p.t_fastest = 1; p.n_co = (Cout + YT::MT - 1) / YT::MT;
    p.tiles_x = (W + YT::COLS * 32 - 1) / (YT::COLS * 32); p.tiles_y = (H + YT::ROWS - 1) / YT::ROWS; p.chunks_per_split = Cin / YT::CK;
    dim3 grid(p.tiles_x * p.tiles_y * T * p.n_co);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(conv_igemm_kernel<YT>, grid, dim3(YT::NTHREADS), 0, 0, p);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(conv_igemm_kernel<YT>, grid, dim3(YT::NTHREADS), 0, 0, p);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("variant %%d: %%.1f us per launch, %%.1f TFLOP/s-equivalent (%%d workgroups)\n", X6_VARIANT, ms * 1e3, 2.0 * Cin * TAPS * Cout * T * H * W / ms / 1e9, (int)grid.x);
    return 0;
}
''' % (cfg, shape)
os.makedirs(os.path.join(ROOT, "tools", "microbench", "bin"), exist_ok=True)
for v in [int(x) for x in os.environ.get("PROBE_VARIANTS", "0,1,2,3").split(",")]:
    body = head
    if 1 <= v <= 3:      # no prefetch / staging inside the loop
        body = body.replace("                fetch_w6(c0 + C::CK, Q0{}, QA{}, rw6);\n", "")
        body = body.replace("store_w6(Q0{}, QA{}, rw6);", "").replace("fetch_w6(c0 + C::CK, QA{}, QE{}, rw6);", "").replace("store_w6(QA{}, QE{}, rw6);", "")
        body = body.replace("for (int k = 0; k < IN_PT6; ++k) fetch_in6_fast(c0 + C::CK, k, rin[2 * k], rin[2 * k + 1]);", "for (int k = 0; k < 0; ++k) {}")
        body = body.replace("for (int k = 0; k < IN_PT6; ++k) { const int q = tid + k * C::NTHREADS; if (q < NQ6) store_in6(q, rin[2 * k], rin[2 * k + 1]); }", "for (int k = 0; k < 0; ++k) {}")
    if 2 <= v <= 3:      # no barriers in the chunk loop
        a = body.index("        for (int c0 = c_begin; c0 < c_end; c0 += C::CK) {\n            const bool more = c0 + C::CK < c_end;\n            if (more) {")
        b = body.index("    } else if constexpr (C::DB && C::GL) {")
        body = body[:a] + body[a:b].replace("__syncthreads();", "") + body[b:]
    if v == 3:      # fragments loaded once
        body = body.replace("            if (st + 1 < NSTEP) ld_a(g0 + (st + 1) / C::MI, (st + 1) % C::MI, a1);", "            if (st == 0) ld_a(g0, 1, a1);")
        body = body.replace("                if ((st + 1) % C::MI == 0) ld_b(g0 + (st + 1) / C::MI, bfr);\n", "")
        body = body.replace("                if (st + 2 < NSTEP) ld_a(g0 + (st + 2) / C::MI, (st + 2) % C::MI, a0);\n", "")
        body = body.replace("                if (st + 2 < NSTEP && (st + 2) % C::MI == 0) ld_b(g0 + (st + 2) / C::MI, bfr);\n", "")
    # staging dissection (two-phase loop, SS_X6_WMODE_SMALLG=0): 4 = no hi/mid/lo split (raw bits stored), 5 = weights only,
    # 6 = input tile only, 7 = LDS stores kept but nothing fetched from global memory
    IN_F = "for (int k = 0; k < IN_PT6; ++k) fetch_in6_fast(c0 + C::CK, k, rin[2 * k], rin[2 * k + 1]);"
    IN_S = "for (int k = 0; k < IN_PT6; ++k) { const int q = tid + k * C::NTHREADS; if (q < NQ6) store_in6(q, rin[2 * k], rin[2 * k + 1]); }"
    if v == 4:
        a = body.index("        split3(v0.x, h0, m0, l0); split3(v1.x")
        b = body.index("        unsigned int* d = reinterpret_cast<unsigned int*>(in_lds) + q * 4;")
        body = body[:a] + "        ph = *reinterpret_cast<const uint4*>(&v0); pm = *reinterpret_cast<const uint4*>(&v1); pl = ph;\n" + body[b:]
    if v == 5:
        body = body.replace(IN_F, "for (int k = 0; k < 0; ++k) {}").replace(IN_S, "for (int k = 0; k < 0; ++k) {}")
    if v == 6:
        body = body.replace("                fetch_w6(c0 + C::CK, Q0{}, QA{}, rw6);\n", "").replace("store_w6(Q0{}, QA{}, rw6);", "")
        body = body.replace("fetch_w6(c0 + C::CK, QA{}, QE{}, rw6);", "").replace("store_w6(QA{}, QE{}, rw6);", "")
    if v == 7:
        body = body.replace("                fetch_w6(c0 + C::CK, Q0{}, QA{}, rw6);\n", "").replace("fetch_w6(c0 + C::CK, QA{}, QE{}, rw6);", "")
        body = body.replace(IN_F, "for (int k = 0; k < 2 * IN_PT6; ++k) rin[k] = make_float4(1.f, 2.f, 3.f, 4.f);")
        body = body.replace("f32x4 rw6[W_PT6];", "f32x4 rw6[W_PT6] = {};")
    path = "/tmp/%s_V%d.hip" % (tag, v)
    open(path, "w").write("#define X6_VARIANT %d\n" % v + body + main)
    exe = os.path.join(ROOT, "tools", "microbench", "bin", "%s_V%d" % (tag, v))
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-unused-value"] + os.environ.get("PROBE_DEFS", "").split() + [path, "-o", exe],
                       capture_output=True, text=True)
    print("V%d build %s" % (v, "ok" if r.returncode == 0 else "FAILED\n" + r.stderr[-1500:]))
