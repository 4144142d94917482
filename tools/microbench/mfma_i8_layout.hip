// Probe of v_mfma_i32_32x32x32_i8 operand / result lane maps with exact integer data (gfx950).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_i8_layout tools/microbench/mfma_i8_layout.hip
// Hypothesis checked: lane l (r = l & 31, h = l >> 5) holds A[r][16h + j] and B[16h + j][r] in byte
// j = 0..15 of its 4-VGPR fragment; D[row][col]: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const int8_t *A, const int8_t *B, int *D)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a, b;
    for (int g = 0; g < 4; g++) {
        uint32_t wa = 0, wb = 0;
        for (int j = 0; j < 4; j++) {
            int k = 16 * h + 4 * g + j;
            wa |= (uint32_t)(uint8_t)A[r * 32 + k] << (8 * j);
            wb |= (uint32_t)(uint8_t)B[k * 32 + r] << (8 * j);
        }
        a[g] = (int)wa;
        b[g] = (int)wb;
    }
    v16i c = {};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; i++) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        D[row * 32 + r] = c[i];
    }
}

int main()
{
    int8_t hA[1024], hB[1024];
    int hD[1024], ref[1024];
    srand(7);
    for (int i = 0; i < 1024; i++) {
        hA[i] = (int8_t)(rand() % 256 - 128);
        hB[i] = (int8_t)(rand() % 256 - 128);
    }
    for (int m = 0; m < 32; m++)
        for (int n = 0; n < 32; n++) {
            int s = 0;
            for (int k = 0; k < 32; k++)
                s += (int)hA[m * 32 + k] * (int)hB[k * 32 + n];
            ref[m * 32 + n] = s;
        }
    int8_t *dA, *dB;
    int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++)
        bad += hD[i] != ref[i];
    printf("mfma_i32_32x32x32_i8 layout hypothesis: %d mismatches of 1024\n", bad);
    return bad != 0;
}
