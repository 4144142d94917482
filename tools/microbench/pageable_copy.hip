// Large asynchronous copies between pageable host memory and the device on the NULL stream, the way a ctypes / NumPy
// host layer issues them: upload, a kernel, download into a fresh array, hipStreamSynchronize, compare.
//   hipcc --offload-arch=gfx950 -O2 pageable_copy.hip -o pageable_copy && ./pageable_copy
// MI355X: no wrong word in either mode -- the span of zeros seen through video.ops needs more than
// this pattern (there the call also leases scratch from the stream-ordered allocator before the download).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>

__global__ void plus_one(const uint32_t *in, uint32_t *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = in[i] + 1u;
}

int main()
{
    const size_t n = 64ull * 1080 * 1920, bytes = n * 4;          // 531 MB
    uint32_t *src = (uint32_t *)malloc(bytes), *a, *b;
    for (size_t i = 0; i < n; i++)
        src[i] = (uint32_t)(i * 2654435761u);
    if (hipMalloc(&a, 1ull << 30) != hipSuccess || hipMalloc(&b, 1ull << 30) != hipSuccess)
        return 1;
    for (int mode = 0; mode < 2; mode++) {
        unsigned long long total_bad = 0;
        for (int run = 0; run < 6; run++) {
            uint32_t *din = (run & 1) ? b : a, *dout = (run & 1) ? a : b;     // buffers swap roles like a pool's
            uint32_t *out = (uint32_t *)malloc(bytes);                          // fresh pages every run
            if (mode == 0) {
                (void)hipMemcpyAsync(din, src, bytes, hipMemcpyHostToDevice, nullptr);
                plus_one<<<(unsigned)((n + 255) / 256), 256>>>(din, dout, n);
                (void)hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, nullptr);
                (void)hipStreamSynchronize(nullptr);
            } else {
                (void)hipMemcpy(din, src, bytes, hipMemcpyHostToDevice);
                plus_one<<<(unsigned)((n + 255) / 256), 256>>>(din, dout, n);
                (void)hipStreamSynchronize(nullptr);
                (void)hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
            }
            unsigned long long bad = 0, zeros = 0;
            for (size_t i = 0; i < n; i++) {
                bad += out[i] != src[i] + 1u;
                zeros += out[i] == 0;
            }
            printf("%s run %d: %llu wrong words (%llu zeros)\n", mode == 0 ? "hipMemcpyAsync + sync" : "hipMemcpy          ", run, bad,
                   zeros);
            total_bad += bad;
            free(out);
        }
        printf("%s: %llu wrong words in all\n", mode == 0 ? "asynchronous pageable copies" : "blocking copies", total_bad);
    }
    return 0;
}
