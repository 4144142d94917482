// micro-benchmark: can 256 persistent workgroups (one per frame, 16 waves, as ccl_frame_kernel runs)
// write the batch's 2.1 GB label image at the HBM store ceiling?  Wave v of workgroup f writes
// CHUNK consecutive KB, then jumps 16 * CHUNK KB ahead (the shape a paint phase fused into the
// per-frame labelling kernel would have).  build: hipcc --offload-arch=gfx950 -O3 fill_persistent.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int CHUNK, bool NT>
__global__ void __launch_bounds__(1024) fill_frames(v4i *p, int kb_per_frame, int v)
{
    const v4i val = {v, v, v, v};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v4i *frame = p + (size_t)blockIdx.x * kb_per_frame * 64;
    for (int k0 = wave * CHUNK; k0 < kb_per_frame; k0 += 16 * CHUNK)
#pragma unroll 4
        for (int i = 0; i < CHUNK; i++) {
            const int kb = k0 + i;
            if (kb < kb_per_frame) {
                if (NT) __builtin_nontemporal_store(val, frame + (size_t)kb * 64 + lane);
                else frame[(size_t)kb * 64 + lane] = val;
            }
        }
}
template <int CHUNK, bool NT> void run(const char *name, v4i *d, int frames, int kbpf)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    fill_frames<CHUNK, NT><<<frames, 1024>>>(d, kbpf, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) fill_frames<CHUNK, NT><<<frames, 1024>>>(d, kbpf, r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)frames * kbpf * 1024;
    printf("%-44s: %.3f ms per pass -> %.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
}
int main()
{
    const int frames = 256, kbpf = 8100;            // 1080 x 1920 int32 labels = 8100 KB per frame
    v4i *d; (void)hipMalloc(&d, (size_t)frames * kbpf * 1024);
    run<60, false>("60 KB per wave step (8 rows)", d, frames, kbpf);
    run<60, true>("60 KB per wave step, nontemporal", d, frames, kbpf);
    run<15, true>("15 KB per wave step (2 rows), nontemporal", d, frames, kbpf);
    run<4, true>("4 KB per wave step, nontemporal", d, frames, kbpf);
    run<1, true>("1 KB per wave step, nontemporal", d, frames, kbpf);
    run<1, false>("1 KB per wave step", d, frames, kbpf);
    return 0;
}
