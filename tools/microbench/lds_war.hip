// Does an LDS store read its data VGPR at issue?  A wave issues three LDS stores and overwrites their data registers
// with a VALU move right behind them (what a compiler may schedule), while the other waves of the CU keep the LDS
// queue busy; the stored values are read back and compared.   hipcc --offload-arch=gfx950 -O2 lds_war.hip -o lds_war
// MI355X, 1024 workgroups x 16 waves x 2000 rounds: 0 differences with and without wait states -- the data is read at
// issue (checked while looking for the cause of the planar-layout failures; it is not this).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(1024) probe(uint32_t *bad_out, int iters, int gap)
{
    __shared__ uint32_t s[12288];
    const int t = threadIdx.x;
    uint32_t bad = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t d0 = 0x1000000u + it * 4u + t, d1 = d0 + 0x100000u, d2 = d0 + 0x200000u, d3 = d0 + 0x300000u;
        const uint32_t e0 = d0, e1 = d1, e2 = d2, e3 = d3;
        uint32_t a0 = (uint32_t)(4 * (t * 4)), a1 = a0 + 8, a2 = a0 + 12;      // four words per thread
        uint32_t junk = 0xFFFFB484u;
        if (gap == 0)
            asm volatile("ds_write2_b32 %4, %0, %3 offset1:1\n\tds_write_b32 %5, %1\n\tds_write_b32 %6, %2\n\t"
                         "v_mov_b32 %0, %7\n\tv_mov_b32 %1, %7\n\tv_mov_b32 %2, %7\n\tv_mov_b32 %3, %7"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(junk) : "memory");
        else
            asm volatile("ds_write2_b32 %4, %0, %3 offset1:1\n\tds_write_b32 %5, %1\n\tds_write_b32 %6, %2\n\t"
                         "s_nop 7\n\tv_mov_b32 %0, %7\n\tv_mov_b32 %1, %7\n\tv_mov_b32 %2, %7\n\tv_mov_b32 %3, %7"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(junk) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t r0 = s[t * 4], r3 = s[t * 4 + 1], r1 = s[t * 4 + 2], r2 = s[t * 4 + 3];
        bad += (r0 != e0) + (r1 != e1) + (r2 != e2) + (r3 != e3);
        bad += (d0 != junk);                       // (keeps the moves alive)
    }
    if (bad)
        atomicAdd(bad_out, bad);
}

int main()
{
    uint32_t *d, h = 0;
    hipMalloc(&d, 4);
    for (int gap = 0; gap < 2; gap++) {
        hipMemset(d, 0, 4);
        probe<<<1024, 1024>>>(d, 2000, gap);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("gap %d: %u stored words differ from the data their store was issued with\n", gap, h);
    }
    return 0;
}
