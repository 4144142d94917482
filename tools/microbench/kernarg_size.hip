// Probe: how large may a by-value kernel argument be on this ROCm / gfx950?  (HIP documents 4 KB.)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> struct Big { double v[N]; };
template <int N> __global__ void k(Big<N> b, double *out) { if (threadIdx.x == 0) { double s = 0; for (int i = 0; i < N; i++) s += b.v[i]; *out = s; } }
template <int N> void run(double *d) {
    Big<N> b; for (int i = 0; i < N; i++) b.v[i] = 1.0;
    hipLaunchKernelGGL(k<N>, 1, 64, 0, 0, b, d);
    hipError_t e = hipDeviceSynchronize(); double h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%5d bytes: %s, sum %.0f (launch err %s)\n", N * 8, hipGetErrorString(e), h, hipGetErrorString(hipGetLastError()));
}
int main() { double *d; hipMalloc(&d, 8); run<256>(d); run<500>(d); run<768>(d); run<1024>(d); run<2000>(d); return 0; }
