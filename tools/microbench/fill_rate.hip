// micro-benchmark: HBM write ceiling for a 2.1 GB stream (the label image of one batch)
// build: hipcc --offload-arch=gfx950 -O3 fill_rate.hip -o fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void fill(v4i *p, size_t n16, int v)
{
    const v4i val = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        if (NT)
            __builtin_nontemporal_store(val, p + i);
        else
            p[i] = val;
    }
}
// one 16-byte store per thread (what a memset does)
template <bool NT>
__global__ void fill_once(v4i *p, size_t n16, int v)
{
    const v4i val = {v, v, v, v};
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) {
        if (NT)
            __builtin_nontemporal_store(val, p + i);
        else
            p[i] = val;
    }
}
// the paint kernel's shape: a wave writes one 7680-byte row as 8 consecutive 1 KB pieces
template <bool NT>
__global__ void fill_rows(v4i *p, size_t rows, int v)
{
    const v4i val = {v, v, v, v};
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows)
        return;
    v4i *q = p + row * 480;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int e = i * 64 + lane;
        if (e < 480) {
            if (NT)
                __builtin_nontemporal_store(val, q + e);
            else
                q[e] = val;
        }
    }
}
// a 7680-byte row split over WPR waves, each writing 8 / WPR consecutive 1 KB pieces
template <int WPR>
__global__ void fill_rows_split(v4i *p, size_t rows, int v)
{
    const v4i val = {v, v, v, v};
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t row = wave / WPR;
    const int part = (int)(wave % WPR), lane = threadIdx.x & 63;
    if (row >= rows)
        return;
    v4i *q = p + row * 480;
#pragma unroll
    for (int i = 0; i < 8 / WPR; i++) {
        const int e = (part * (8 / WPR) + i) * 64 + lane;
        if (e < 480)
            q[e] = val;
    }
}
// a wave writes PIECES consecutive, 1 KB-aligned pieces (OFF = byte offset of the whole stream)
template <int PIECES, int OFF>
__global__ void fill_aligned(v4i *p, size_t n16, int v)
{
    const v4i val = {v, v, v, v};
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < PIECES; i++) {
        const size_t e = (wave * PIECES + i) * 64 + lane + OFF / 16;
        if (e < n16)
            p[e] = val;
    }
}
template <int PIECES, int OFF> void run4(const char *name, v4i *d, size_t bytes)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t n16 = bytes / 16, waves = (n16 + 64 * PIECES - 1) / (64 * PIECES);
    fill_aligned<PIECES, OFF><<<(unsigned)((waves + 3) / 4), 256>>>(d, n16, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++)
        fill_aligned<PIECES, OFF><<<(unsigned)((waves + 3) / 4), 256>>>(d, n16, r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s: %.3f ms per pass -> %.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}
// eight waves share eight 7680-byte rows: wave i writes piece i of every row (its consecutive
// stores are a row apart instead of adjacent)
__global__ void fill_rows_transposed(v4i *p, size_t rows, int v)
{
    const v4i val = {v, v, v, v};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t row0 = (size_t)blockIdx.x * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const size_t row = row0 + j;
        const int e = wave * 64 + lane;
        if (row < rows && e < 480)
            p[row * 480 + e] = val;
    }
}
void run5(const char *name, v4i *d, size_t bytes)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t rows = bytes / 7680;
    fill_rows_transposed<<<(unsigned)((rows + 7) / 8), 512>>>(d, rows, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++)
        fill_rows_transposed<<<(unsigned)((rows + 7) / 8), 512>>>(d, rows, r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s: %.3f ms per pass -> %.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}
// a wave writes its row's 8 pieces with a pause between the stores
template <int SLEEP>
__global__ void fill_rows_paced(v4i *p, size_t rows, int v)
{
    const v4i val = {v, v, v, v};
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows)
        return;
    v4i *q = p + row * 480;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int e = i * 64 + lane;
        if (e < 480)
            q[e] = val;
        __builtin_amdgcn_s_sleep(SLEEP);
    }
}
template <int SLEEP> void run6(const char *name, v4i *d, size_t bytes)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t rows = bytes / 7680;
    fill_rows_paced<SLEEP><<<(unsigned)((rows + 3) / 4), 256>>>(d, rows, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++)
        fill_rows_paced<SLEEP><<<(unsigned)((rows + 3) / 4), 256>>>(d, rows, r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s: %.3f ms per pass -> %.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}
template <int WPR> void run3(const char *name, v4i *d, size_t bytes)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t rows = bytes / 7680, waves = rows * WPR;
    fill_rows_split<WPR><<<(unsigned)((waves + 3) / 4), 256>>>(d, rows, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++)
        fill_rows_split<WPR><<<(unsigned)((waves + 3) / 4), 256>>>(d, rows, r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s: %.3f ms per pass -> %.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}
template <int MODE, bool NT> void run2(const char *name, v4i *d, size_t bytes)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t n16 = bytes / 16, rows = bytes / 7680;
    auto launch = [&](int v) {
        if (MODE == 0)
            fill_once<NT><<<(unsigned)((n16 + 255) / 256), 256>>>(d, n16, v);
        else
            fill_rows<NT><<<(unsigned)((rows + 3) / 4), 256>>>(d, rows, v);
    };
    launch(1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++)
        launch(r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s: %.3f ms per pass -> %.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}
template <bool NT> void run(const char *name, v4i *d, size_t bytes, int blocks)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    fill<NT><<<blocks, 256>>>(d, bytes / 16, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++)
        fill<NT><<<blocks, 256>>>(d, bytes / 16, r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %6d: %.3f ms per pass -> %.2f TB/s\n", name, blocks, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}
int main()
{
    const size_t bytes = (size_t)256 * 1080 * 1920 * 4;
    v4i *d; (void)hipMalloc(&d, bytes);
    run<false>("plain 16-B stores", d, bytes, 256 * 8);
    run<true>("non-temporal 16-B stores", d, bytes, 256 * 8);
    run<true>("non-temporal 16-B stores", d, bytes, 256 * 32);
    run<true>("non-temporal 16-B stores", d, bytes, 65536);
    run2<0, false>("one plain 16-B store per thread", d, bytes);
    run2<0, true>("one non-temporal 16-B store per thread", d, bytes);
    run2<1, false>("wave = 7680-B row, plain", d, bytes);
    run2<1, true>("wave = 7680-B row, non-temporal", d, bytes);
    run4<8, 0>("wave = 8 aligned KB", d, bytes);
    run4<8, 512>("wave = 8 KB at a 512-B offset", d, bytes);
    run4<1, 512>("wave = 1 KB at a 512-B offset", d, bytes);
    run4<2, 0>("wave = 2 aligned KB", d, bytes);
    run5("8 rows x 8 waves, wave = piece column", d, bytes);
    run6<2>("row per wave, s_sleep 2 between stores", d, bytes);
    run6<8>("row per wave, s_sleep 8 between stores", d, bytes);
    run6<32>("row per wave, s_sleep 32 between stores", d, bytes);
    run3<2>("row over 2 waves (4 x 1 KB each), plain", d, bytes);
    run3<4>("row over 4 waves (2 x 1 KB each), plain", d, bytes);
    run3<8>("row over 8 waves (1 x 1 KB each), plain", d, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemsetAsync(d, 0, bytes, 0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) (void)hipMemsetAsync(d, r, bytes, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s               : %.3f ms per pass -> %.2f TB/s\n", "hipMemsetAsync", ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    return 0;
}
