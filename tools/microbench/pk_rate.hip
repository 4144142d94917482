// micro-benchmark: issue rate of the packed fp32 VALU ops the float32 Gaussian is made of
// (v_pk_fma_f32 with VGPR / SGPR weights, v_pk_add_f32 + v_pk_fma_f32 pairs, plain v_fma_f32),
// at 8, 4 and 3 waves per SIMD, plus the in-kernel clock (s_memtime / s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 2048
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float a, float b, unsigned long long *clk)
{
    f2 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = f2{(float)threadIdx.x + i, (float)i};
    f2 x = f2{a + threadIdx.x, a}, y = f2{b, b};
    f2 z[8];
    for (int i = 0; i < 8; i++) z[i] = f2{(float)i * a, b + i};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) acc[i] = __builtin_elementwise_fma(x, y, acc[i]);                       // pk_fma, VGPR operands
            if (MODE == 1) acc[i] = __builtin_elementwise_fma(acc[i], f2{b, b}, x);                // pk_fma, SGPR weight
            if (MODE == 2) { f2 s = z[i & 7] + z[(i + 3) & 7]; acc[i] = __builtin_elementwise_fma(s, f2{b, b}, acc[i]); }   // add + fma
            if (MODE == 3) { acc[i].x = __builtin_fmaf(x.x, y.x, acc[i].x); acc[i].y = __builtin_fmaf(x.y, y.y, acc[i].y); }  // 2 scalar fma
            if (MODE == 4) acc[i] = acc[i] + x;                                                    // pk_add
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
        asm volatile("" : "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    f2 s = f2{0.f, 0.f};
    for (int i = 0; i < 16; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <int MODE> void run(const char *name, float *d, unsigned long long *clk, int waves_per_simd)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * waves_per_simd, threads = 256;
    k<MODE><<<blocks, threads>>>(d, 3.f, 5.f, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 20; r++) k<MODE><<<blocks, threads>>>(d, 3.f, 5.f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / (double)h[1] * 0.1;                  // s_memrealtime ticks at 100 MHz
    int per = (MODE == 2 || MODE == 3) ? 2 : 1;                       // VALU instructions per loop slot
    double waveinstr = (double)blocks * threads / 64 * ITERS * 16 * per;
    double per_simd = waveinstr / (256 * 4);
    printf("%-34s %d waves/SIMD  %.3f ms  %.3f ns/wave-instr/SIMD = %.2f cycles at the in-kernel clock %.2f GHz\n", name,
           waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * ghz, ghz);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long *clk; hipMalloc(&clk, 16);
    for (int w : {8, 4, 3, 1}) {
        run<0>("v_pk_fma_f32 (VGPR x VGPR)", d, clk, w);
        run<1>("v_pk_fma_f32 (SGPR weight)", d, clk, w);
        run<2>("v_pk_add_f32 + v_pk_fma_f32", d, clk, w);
        run<3>("2 x v_fma_f32", d, clk, w);
        run<4>("v_pk_add_f32", d, clk, w);
    }
    return 0;
}
