#!/usr/bin/env python3
"""(round-2 tool, kept for reference: its embedded launches predate the round-3 kernel signatures; ablation builds are
made with tools/debug/build_variant.sh now)

Phase ablation of the fused float32 kernels (va_gauss_f32_fused.hip) on an MI355X.

Builds variants of the PRODUCT source by textual substitution (nothing here ships), times
ema_row_f32_kernel<3,true> and col_march_f32_kernel on 64 x 1080p x 3 float32 frames and prints
ms per 256 frames.  Usage (on the GPU box):  python tools/microbench/f32_phase_ablation.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "video-analysis_amd", "csrc", "va_gauss_f32_fused.hip")
OUT = os.path.join(ROOT, "gpurun_out", "f32_ablation")

MAIN = r'''
#include <stdarg.h>
namespace va { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
const char *get_error() { return ""; }
int gauss_taps_f32_host(double sigma, TapsF32 *t) {
    int ks = ((int)(sigma * 8 + 1 + 0.5)) | 1; t->ksize = ks; double s = 0, k[256];
    for (int i = 0; i < ks; i++) { double x = i - (ks - 1) * 0.5; k[i] = exp(-x * x / (2 * sigma * sigma)); s += k[i]; }
    for (int i = 0; i < ks; i++) t->t[i] = (float)(k[i] / s); return 0; } }
int main() {
    using namespace va;
    const int n = 64, h = 1080, w = 1920, c = 3; const size_t px = (size_t)h * w * c;
    float *src, *tmp, *dst, *bg;
    hipMalloc(&src, n * px * 4); hipMalloc(&tmp, n * px * 4); hipMalloc(&dst, n * px * 4); hipMalloc(&bg, px * 4);
    hipMemset(src, 0x3c, n * px * 4); hipMemset(bg, 0, px * 4); hipMemset(tmp, 0x3c, n * px * 4);
    TapsF32 taps; gauss_taps_f32_host(9.0, &taps);
    RowPlan plan; if (!plan_rows(h, w, c, taps, 1024, &plan)) { printf("plan failed\n"); return 1; }
    const size_t lds1 = 2 * (size_t)(kHalfCap + (taps.ksize - 1) * c + kP + 8) * sizeof(f2) + 2 * (size_t)kNV * kT * sizeof(unsigned short);
    const int r = taps.ksize / 2, rw = w * c, ncolt = cdiv(rw, kColCols);
    const size_t lds2 = (size_t)(kColRows + 2 * r) * kColStride * sizeof(float);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    float best1 = 1e9, best2 = 1e9;
    for (int it = 0; it < 4; it++) {
        hipEventRecord(e0);
        ema_row_f32_kernel<3, true><<<(plan.nchunks + 1) / 2, 2 * kT, lds1>>>(src, tmp, bg, 1, 0.02f, n, h, w, plan.L, taps);
        hipEventRecord(e1);
        col_march_f32_kernel<<<(unsigned)((size_t)ncolt * n), 256, lds2>>>(tmp, dst, h, rw, ncolt, taps);
        hipEventRecord(e2); hipEventSynchronize(e2);
        float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
        if (a < best1) best1 = a; if (b < best2) best2 = b;
    }
    printf("%s: chunks %d L %d | ema_row %.3f ms  col_march %.3f ms  (per 256 frames)  err=%s\n", VARIANT, plan.nchunks, plan.L,
           best1 * 256 / n, best2 * 256 / n, hipGetErrorString(hipGetLastError()));
    return 0;
}
'''

# name -> list of (old, new) substitutions on the product source
VARIANTS = {
    "baseline": [],
    "stamps": [("            float *sf = reinterpret_cast<float *>(buf);\n            int sbase = sub * kNV * kT + tid;",
                "            VA_STAMP(); asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); VA_STAMP();\n            float *sf = reinterpret_cast<float *>(buf);\n            int sbase = sub * kNV * kT + tid;"),
               ("                __builtin_amdgcn_sched_barrier(0);\n            }\n        };\n        // halos beyond a row end",
                "                __builtin_amdgcn_sched_barrier(0);\n            }\n            VA_STAMP();\n        };\n        // halos beyond a row end")],
    "no_row_pass": [("        if (i0 >= T.Hoff)\n            return;\n        f2 win[WIN];", "        for (int j = 0; j < kP; j++) acc[j] = f2{0.f, 0.f};\n        if (true)\n            return;\n        f2 win[WIN];")],
    "no_staging": [("                if (cls != 0) {", "                if (cls != 0 && ld[m].x == 1234.5f) {")],
    "col_lb4": [("__global__ void __launch_bounds__(256, 3)\ncol_march_f32_kernel", "__global__ void __launch_bounds__(256, 4)\ncol_march_f32_kernel")],
    "col_no_shift": [("        for (int c0 = 0; c0 < 2 * r; c0 += kColRows) {", "        for (int c0 = 0; c0 < 0; c0 += kColRows) {")],
    "no_prio": [("        __builtin_amdgcn_s_setprio(3);", "")],
    "col_no_loads": [("                pf[u] = load4(y0 + kColRows + r + (idx >> 4), (idx & 15) * 4);", "                pf[u] = f4{1.f, 2.f, 3.f, (float)idx};")],
    "no_col_compute": [("        if (col < rw) {\n            f2 up[8], dn[8], acc[8];", "        if (col < rw && tile[0] == 1234.5f) {\n            f2 up[8], dn[8], acc[8];")],
}


STAMP_PREFIX = r'''
#include <hip/hip_runtime.h>
__device__ unsigned long long g_dbg[8][1024][2];
__device__ int g_cnt[8];
#define VA_STAMP() do { if (blockIdx.x == 37 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) { int w_ = threadIdx.x >> 6; int i_ = g_cnt[w_]; if (i_ < 1020) { g_dbg[w_][i_][0] = __LINE__; g_dbg[w_][i_][1] = __builtin_amdgcn_s_memtime(); g_cnt[w_] = i_ + 1; } } } while (0)
#define __syncthreads() do { VA_STAMP(); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); VA_STAMP(); } while (0)
'''
STAMP_DUMP = r'''
    { static unsigned long long hd[8][1024][2]; int hc[8];
      hipMemcpyFromSymbol(hd, HIP_SYMBOL(g_dbg), sizeof(hd)); hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_cnt), sizeof(hc));
      for (int w = 0; w < 8; w += 4) { printf("wave %d (%s): %d stamps; (line: cycles since previous stamp)\\n", w, w < 4 ? "loader" : "compute", hc[w]);
        for (int i = 200; i < 200 + 44 && i < hc[w]; i++) printf("  L%llu:%llu", hd[w][i][0], hd[w][i][1] - hd[w][i - 1][1]); printf("\\n"); } }
'''


def main():
    os.makedirs(OUT, exist_ok=True)
    src = open(SRC).read()
    names = sys.argv[1:] or list(VARIANTS)
    for name in names:
        text = src
        for old, new in VARIANTS[name]:
            if old not in text:
                print("variant %s: pattern not found: %r" % (name, old[:50]))
                sys.exit(1)
            text = text.replace(old, new)
        text = text.replace('#include "va_common.h"', '#include "%s"\n#include <math.h>' %
                            os.path.join(ROOT, "video-analysis_amd", "csrc", "va_common.h"))
        # the kernels live in an anonymous namespace: append main() inside the same TU
        text = text.replace("}  // namespace va\n", "}  // namespace va\n", 1)
        path = os.path.join(OUT, name + ".hip")
        body = text.replace("namespace {\n", "inline namespace ablate {\n", 1)
        main_c = MAIN.replace("using namespace va;", "using namespace va; using namespace va::ablate;")
        prefix = ""
        if name.startswith("stamps"):
            prefix = STAMP_PREFIX
            main_c = main_c.replace("    printf(\"%s: chunks", STAMP_DUMP + "    printf(\"%s: chunks")
            main_c = main_c.replace("for (int it = 0; it < 4; it++)", "for (int it = 0; it < 1; it++)")
        open(path, "w").write('#define VARIANT "%s"\n' % name + prefix + body + main_c)
        exe = os.path.join(OUT, name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-Wno-unused-function", "-Wno-unused-result", path, "-o", exe])
        subprocess.call([exe])


if __name__ == "__main__":
    main()
