// micro-benchmark: issue rate of v_dot4_u32_u8 / v_dot2_u32_u16 / v_mad_u32_u24 / v_fma_f32 on gfx950
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ void k(uint32_t *out, uint32_t a, uint32_t b)
{
    uint32_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
    uint32_t x = a + threadIdx.x, y = b;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_udot4(x, y, acc[i], false);
            if (MODE == 1) acc[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, x), __builtin_bit_cast(us2, y), acc[i], false);
            if (MODE == 2) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
            if (MODE == 5) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
            if (MODE == 6) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
            if (MODE == 3) { float f = __builtin_bit_cast(float, acc[i]); f = __builtin_fmaf(__builtin_bit_cast(float, x), __builtin_bit_cast(float, y), f); acc[i] = __builtin_bit_cast(uint32_t, f); }
            if (MODE == 4) acc[i] = (acc[i] | x) + y;   // generic int ops (v_or + v_add or v_add3)
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, uint32_t *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * 8, threads = 256;   // 8 waves per SIMD
    k<MODE><<<blocks, threads>>>(d, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(d, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waveinstr = (double)blocks * threads / 64 * ITERS * 8;      // wave-instructions
    double per_simd = waveinstr / (256 * 4);
    printf("%-18s %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<3>("v_fma_f32", d); run<2>("v_mad_u32_u24", d); run<0>("v_dot4_u32_u8", d); run<1>("v_dot2_u32_u16", d); run<4>("int or+add (2 instr)", d); run<5>("v_pk_mad_u16", d); run<6>("v_add_u32", d);
    return 0;
}
