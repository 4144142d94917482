// A 16-byte store followed by vector writes to its data registers: how many wait states keep the stored float4 intact?
// LLVM puts `s_nop 1` (two wait states) behind buffer stores whose soffset is a constant and behind global stores, and
// nothing behind buffer stores whose soffset is a register (GCNHazardRecognizer::createsVALUHazard).  Here every form
// is issued by hand with 0 ... 4 wait states, while the whole chip streams 1 GiB of such stores, so that the stores
// have to wait for the memory pipeline.  MI355X (profiles/r03_store_hazard.txt): constant soffset / global: 23 % of
// the first dwords wrong with 0 wait states, 0.8 % with 1, none from 2 on (the compiler's choice is right); register
// soffset: 0.45 % wrong with 0 wait states (the compiler's choice), none from 1 on; 8-byte stores: never.
//   hipcc --offload-arch=gfx950 -O2 store_data_hazard.hip -o store_data_hazard && ./store_data_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define FILL "v_mov_b32 v20, %0\n\tv_add_u32 v21, 1, %0\n\tv_add_u32 v22, 2, %0\n\tv_add_u32 v23, 3, %0\n\ts_nop 4\n\t"
#define SMASH "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %4\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %4"
#define CLOB "v20", "v21", "v22", "v23", "memory"

// KIND 0: buffer_store_dwordx4, register soffset; 1: buffer_store_dwordx4, constant soffset (offset in the vector
// register); 2: global_store_dwordx4; 3: buffer_store_dwordx2, register soffset.  WS = wait states (0: none)
template <int KIND, int WS>
__global__ void __launch_bounds__(256) writer(uint32_t *out, int iters, uint32_t stride_bytes)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7FFFFFFF, 0x00027000);
    const uint32_t voff = t * 16u;
    for (int it = 0; it < iters; it++) {
        const uint32_t a = 0x10000000u + (uint32_t)it * 0x10000u + (t & 0xFFFFu), junk = 0xDEAD0000u + it;
        const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)it * stride_bytes);
        const uint32_t vo = voff + soff;
        const uint64_t gaddr = (uint64_t)out + vo;
#define GAP(n) ((n) == 0 ? "" : (n) == 1 ? "s_nop 0\n\t" : (n) == 2 ? "s_nop 1\n\t" : (n) == 3 ? "s_nop 2\n\t" : "s_nop 3\n\t")
        if (KIND == 0) {
            if (WS == 0) asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\t" SMASH :: "v"(a), "v"(voff), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 1) asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\ts_nop 0\n\t" SMASH :: "v"(a), "v"(voff), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 2) asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\ts_nop 1\n\t" SMASH :: "v"(a), "v"(voff), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 4) asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\ts_nop 3\n\t" SMASH :: "v"(a), "v"(voff), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
        } else if (KIND == 1) {
            if (WS == 0) asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n\t" SMASH :: "v"(a), "v"(vo), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 1) asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n\ts_nop 0\n\t" SMASH :: "v"(a), "v"(vo), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 2) asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n\ts_nop 1\n\t" SMASH :: "v"(a), "v"(vo), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 3) asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n\ts_nop 2\n\t" SMASH :: "v"(a), "v"(vo), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 4) asm volatile(FILL "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n\ts_nop 3\n\t" SMASH :: "v"(a), "v"(vo), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
        } else if (KIND == 2) {
            if (WS == 0) asm volatile(FILL "global_store_dwordx4 %1, v[20:23], off\n\t" SMASH :: "v"(a), "v"(gaddr), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 1) asm volatile(FILL "global_store_dwordx4 %1, v[20:23], off\n\ts_nop 0\n\t" SMASH :: "v"(a), "v"(gaddr), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 2) asm volatile(FILL "global_store_dwordx4 %1, v[20:23], off\n\ts_nop 1\n\t" SMASH :: "v"(a), "v"(gaddr), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
            if (WS == 3) asm volatile(FILL "global_store_dwordx4 %1, v[20:23], off\n\ts_nop 2\n\t" SMASH :: "v"(a), "v"(gaddr), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
        } else {
            if (WS == 0) asm volatile(FILL "buffer_store_dwordx2 v[20:21], %1, %2, %3 offen\n\t" SMASH :: "v"(a), "v"(voff), "s"(rsrc), "s"(soff), "v"(junk) : CLOB);
        }
    }
}

__global__ void checker(const uint32_t *out, int iters, uint32_t stride_words, uint32_t threads, unsigned long long *bad,
                        int nd)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= threads)
        return;
    unsigned b[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        const uint32_t a = 0x10000000u + (uint32_t)it * 0x10000u + (t & 0xFFFFu);
        const uint32_t *p = out + (size_t)it * stride_words + t * 4;
        for (int k = 0; k < nd; k++)
            b[k] += p[k] != a + k;
    }
    for (int k = 0; k < 4; k++)
        if (b[k])
            atomicAdd(bad + k, (unsigned long long)b[k]);
}

static uint32_t *d;
static unsigned long long *bad;
static const uint32_t threads = 256u * 1024u, iters = 256;

template <int KIND, int WS>
static void run(const char *name)
{
    unsigned long long h[4];
    (void)hipMemset(d, 0, (size_t)threads * 16 * iters);
    (void)hipMemset(bad, 0, 32);
    writer<KIND, WS><<<threads / 256, 256>>>(d, iters, threads * 16);
    checker<<<threads / 256, 256>>>(d, iters, threads * 4, threads, bad, KIND == 3 ? 2 : 4);
    (void)hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    printf("%-58s %d wait states: wrong dwords x/y/z/w %llu %llu %llu %llu of %u each\n", name, WS, h[0], h[1], h[2], h[3],
           threads * iters);
}

int main()
{
    if (hipMalloc(&d, (size_t)threads * 16 * iters) != hipSuccess || hipMalloc(&bad, 32) != hipSuccess)
        return 1;
    run<0, 0>("buffer_store_dwordx4, register soffset (compiler: 0)");
    run<0, 1>("buffer_store_dwordx4, register soffset");
    run<0, 2>("buffer_store_dwordx4, register soffset");
    run<0, 4>("buffer_store_dwordx4, register soffset");
    run<1, 0>("buffer_store_dwordx4, constant soffset");
    run<1, 1>("buffer_store_dwordx4, constant soffset");
    run<1, 2>("buffer_store_dwordx4, constant soffset (compiler: 2)");
    run<1, 3>("buffer_store_dwordx4, constant soffset");
    run<1, 4>("buffer_store_dwordx4, constant soffset");
    run<2, 0>("global_store_dwordx4");
    run<2, 1>("global_store_dwordx4");
    run<2, 2>("global_store_dwordx4 (compiler: 2)");
    run<2, 3>("global_store_dwordx4");
    run<3, 0>("buffer_store_dwordx2, register soffset");
    return 0;
}
