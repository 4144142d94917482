// micro-benchmark: issue rate of the float64 VALU ops of the running-mean kernel on gfx950
// build: hipcc --offload-arch=gfx950 -O3 f64_rate.hip -o f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
template <int MODE>
__global__ void k(double *out, double a, double b)
{
    double acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
    double x = a + threadIdx.x, y = b;
    uint32_t u = threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
            if (MODE == 1) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
            if (MODE == 2) asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
            if (MODE == 3) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(acc[i]) : "v"(u));
            if (MODE == 4) { uint32_t r; asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(acc[i])); u += r; }
            if (MODE == 5) { int r; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(acc[i])); u += r; }
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(u));
    }
    double s = u;
    for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, double *d, int extra)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int blocks = 256 * 8, threads = 256;   // 8 waves per SIMD
    k<MODE><<<blocks, threads>>>(d, 3, 5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(d, 3, 5);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double waveinstr = (double)blocks * threads / 64 * ITERS * 8 * (1 + extra);
    double per_simd = waveinstr / (256 * 4);
    printf("%-18s %.3f ms  -> %.2f cycles @2.4GHz per wave-instr per SIMD%s\n", name, ms, ms * 1e6 / per_simd * 2.4, extra ? " (incl. one v_add_u32 each)" : "");
}
int main()
{
    double *d; (void)hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("v_fma_f64", d, 0); run<1>("v_mul_f64", d, 0); run<2>("v_add_f64", d, 0); run<3>("v_cvt_f64_u32", d, 0);
    run<4>("v_cvt_u32_f64", d, 1); run<5>("v_cvt_i32_f64", d, 1);
    return 0;
}
