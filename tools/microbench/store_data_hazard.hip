// A 16-byte buffer store followed at once by vector writes to its data registers: is the stored float4 intact?
// LLVM inserts the wait state itself when soffset is a constant and none when soffset is a register
// (GCNHazardRecognizer::createsVALUHazard); here both forms are issued by hand, with and without wait states, while
// the whole chip streams 1 GiB of such stores (the stores have to wait for the memory pipeline).
//   hipcc --offload-arch=gfx950 -O2 store_data_hazard.hip -o store_data_hazard && ./store_data_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: register soffset, no gap; 1: register soffset, s_nop 3; 2: constant soffset 0, no gap;
                      // 3: an 8-byte store (dwordx2), register soffset, no gap
__global__ void __launch_bounds__(256) writer(uint32_t *out, int iters, uint32_t stride_bytes)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7FFFFFFF, 0x00027000);
    const uint32_t voff = t * 16u;
    for (int it = 0; it < iters; it++) {
        const uint32_t a = 0x10000000u + (uint32_t)it * 0x10000u + (t & 0xFFFFu), junk = 0xDEAD0000u + it;
        uint32_t soff = (uint32_t)it * stride_bytes;
        soff = __builtin_amdgcn_readfirstlane(soff);
        if (MODE == 0)
            asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v21, 1, %0\n\tv_add_u32 v22, 2, %0\n\tv_add_u32 v23, 3, %0\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\t"
                         "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %4\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %4"
                         :: "v"(a), "v"(voff), "s"(rsrc), "s"(soff), "v"(junk) : "v20", "v21", "v22", "v23", "memory");
        else if (MODE == 1)
            asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v21, 1, %0\n\tv_add_u32 v22, 2, %0\n\tv_add_u32 v23, 3, %0\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\ts_nop 3\n\t"
                         "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %4\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %4"
                         :: "v"(a), "v"(voff), "s"(rsrc), "s"(soff), "v"(junk) : "v20", "v21", "v22", "v23", "memory");
        else if (MODE == 3)
            asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v21, 1, %0\n\tv_add_u32 v22, 2, %0\n\tv_add_u32 v23, 3, %0\n\ts_nop 4\n\t"
                         "buffer_store_dwordx2 v[20:21], %1, %2, %3 offen\n\t"
                         "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %4\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %4"
                         :: "v"(a), "v"(voff), "s"(rsrc), "s"(soff), "v"(junk) : "v20", "v21", "v22", "v23", "memory");
        else {
            const uint32_t vo = voff + soff;
            asm volatile("v_mov_b32 v20, %0\n\tv_add_u32 v21, 1, %0\n\tv_add_u32 v22, 2, %0\n\tv_add_u32 v23, 3, %0\n\ts_nop 4\n\t"
                         "buffer_store_dwordx4 v[20:23], %1, %2, 0 offen\n\t"
                         "v_mov_b32 v20, %3\n\tv_mov_b32 v21, %3\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %3"
                         :: "v"(a), "v"(vo), "s"(rsrc), "v"(junk) : "v20", "v21", "v22", "v23", "memory");
        }
    }
}

__global__ void checker(const uint32_t *out, int iters, uint32_t stride_words, uint32_t threads, unsigned long long *bad,
                        int two)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= threads)
        return;
    unsigned b[4] = {0, 0, 0, 0};
    const int nd = two ? 2 : 4;
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

int main()
{
    const uint32_t threads = 256u * 1024u, iters = 256;
    const size_t bytes = (size_t)threads * 16 * iters;         // 1 GiB
    uint32_t *d;
    unsigned long long *bad, h[4];
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&bad, 32) != hipSuccess)
        return 1;
    const char *names[4] = {"register soffset, vector writes right behind the store", "register soffset, s_nop 3 in between",
                            "constant soffset (offset in the vector register), writes right behind",
                            "8-byte store, register soffset, writes right behind"};
    for (int mode = 0; mode < 4; mode++)
        for (int rep = 0; rep < 2; rep++) {
            (void)hipMemset(d, 0, bytes);
            (void)hipMemset(bad, 0, 32);
            if (mode == 0)
                writer<0><<<threads / 256, 256>>>(d, iters, threads * 16);
            else if (mode == 1)
                writer<1><<<threads / 256, 256>>>(d, iters, threads * 16);
            else if (mode == 2)
                writer<2><<<threads / 256, 256>>>(d, iters, threads * 16);
            else
                writer<3><<<threads / 256, 256>>>(d, iters, threads * 16);
            checker<<<threads / 256, 256>>>(d, iters, threads * 4, threads, bad, mode == 3);
            (void)hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
            printf("%-72s wrong dwords x/y/z/w: %llu %llu %llu %llu of %u each\n", names[mode], h[0], h[1], h[2], h[3],
                   threads * iters);
        }
    return 0;
}
