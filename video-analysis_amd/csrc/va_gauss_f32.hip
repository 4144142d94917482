// va_gauss_f32.hip -- float32 Gaussian blur, fast two-pass path (1 or 3 interleaved channels)
//
// replaces cv2.GaussianBlur(float image, (0, 0), sigma), video/analysis/active_contour.py:108
// (BASELINE.json configs[4]: 1080p x 3 channels, sigma = 9 -> 73 taps).
//
// Bit-exact with the oracle's definition: row pass = fmaf chain over the taps in order, column
// pass = centre tap then symmetric pairs (T[y+k] + T[y-k]) folded with fmaf.  Unlike the u8
// kernel there is no packed trick: ~146 FMA-class operations per sample, so both passes are
// organised to be VALU-bound, not load-bound:
//   row pass : one block per image row; the row (+ reflected halo) is staged once in LDS; a
//              thread owns P consecutive samples of the interleaved row, keeps a rolling window
//              of P + C inputs in registers and per tap reads only C new floats (register
//              indices are compile-time: the tap loop is unrolled over one window period)
//   col pass : a thread owns one sample column for 8 consecutive rows and streams outwards from
//              the centre: two 8-row register rings, two new (coalesced, prefetched) rows per tap
// The intermediate image goes through HBM once (write + read = 8 B per sample); a fully fused
// kernel would need a (2r+8)-row float ring in LDS (61 KB at r = 36 for a 64-pixel strip).
#include "va_common.h"

namespace va {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

template <int C, int P>
__global__ void __launch_bounds__(kBlock, 4)
gauss_row_f32_kernel(const float *__restrict__ src, float *__restrict__ tmp, int w, TapsF32 taps,
                     int vec4)
{
    constexpr int WIN = P + C;          // rolling window: P samples in use + C arriving
    constexpr int PERIOD = WIN / C;     // taps after which the window slots repeat
    static_assert(WIN % C == 0, "P must be a multiple of C");
    static_assert(P % 2 == 1, "an odd per-thread stride keeps the LDS reads of a wave conflict-free");
    extern __shared__ float s_row[];    // w*C + 2*r*C samples: the row with its reflected halo
    const int n = taps.ksize, r = n >> 1;
    const int rw = w * C, halo = r * C;
    const float *row = src + (size_t)blockIdx.x * rw;
    // interior: straight (vectorised when aligned) copy; only the 2*halo edge samples reflect
    if (vec4) {
        const float4 *row4 = reinterpret_cast<const float4 *>(row);
        for (int i = threadIdx.x; i < rw / 4; i += kBlock) {
            const float4 v = row4[i];
            float *d = s_row + halo + 4 * i;
            d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < rw; i += kBlock)
            s_row[halo + i] = row[i];
    }
    for (int i = threadIdx.x; i < 2 * halo; i += kBlock) {
        const int q = i < halo ? i - halo : rw + (i - halo);   // sample index relative to the row start
        const int px = q >= 0 ? q / C : -((-q + C - 1) / C);
        const int ch = q - px * C;
        s_row[q + halo] = row[reflect101(px, w) * C + ch];
    }
    __syncthreads();

    for (int i0 = threadIdx.x * P; i0 < rw; i0 += kBlock * P) {
        float win[WIN], acc[P];
#pragma unroll
        for (int j = 0; j < P; j++) {
            acc[j] = 0.0f;
            win[j] = s_row[i0 + j];
        }
        // full window periods: branch-free, the C samples of the next tap are fetched from LDS
        // before this tap's P fmas and installed after them (software pipeline, depth one tap)
        int k0 = 0;
        for (; k0 + PERIOD <= n; k0 += PERIOD) {
            float wk[PERIOD];               // this period's taps, fetched together (scalar loads)
#pragma unroll
            for (int t = 0; t < PERIOD; t++)
                wk[t] = taps.t[k0 + t];
#pragma unroll
            for (int t = 0; t < PERIOD; t++) {
                float nx[C];
#pragma unroll
                for (int c = 0; c < C; c++)     // (one tap past the end reads the LDS padding)
                    nx[c] = s_row[i0 + (k0 + t) * C + P + c];
#pragma unroll
                for (int j = 0; j < P; j++)
                    acc[j] = fmaf(win[(t * C + j) % WIN], wk[t], acc[j]);
#pragma unroll
                for (int c = 0; c < C; c++)
                    win[(t * C + P + c) % WIN] = nx[c];
                // keep the pipeline depth at one tap: without this the scheduler hoists the LDS
                // reads of many taps and triples the register footprint
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // remaining taps (fewer than one period)
#pragma unroll
        for (int t = 0; t < PERIOD; t++) {
            const int k = k0 + t;
            if (k < n) {                                // uniform
                const float wk = taps.t[k];
                float nx[C];
#pragma unroll
                for (int c = 0; c < C; c++)
                    nx[c] = s_row[i0 + k * C + P + c];
#pragma unroll
                for (int j = 0; j < P; j++)
                    acc[j] = fmaf(win[(t * C + j) % WIN], wk, acc[j]);
#pragma unroll
                for (int c = 0; c < C; c++)
                    win[(t * C + P + c) % WIN] = nx[c];
            }
        }
        float *out = tmp + (size_t)blockIdx.x * rw + i0;
#pragma unroll
        for (int j = 0; j < P; j++)
            if (i0 + j < rw)
                out[j] = acc[j];
    }
}

// Column pass.  For output rows y0..y0+7 of one sample column, tap k needs the two 8-row
// groups T[y0+k .. y0+7+k] and T[y0-k .. y0+7-k]; from one tap to the next each group slides by
// one row.  So the thread keeps just two 8-entry rings (slot = row mod 8, compile-time after
// unrolling 8 taps), fetches two new rows per tap -- prefetched kPF taps ahead -- and the whole
// pass needs ~45 VGPRs instead of a (2r+8)-row register window.
template <int R>
__global__ void __launch_bounds__(kBlock, 6)
gauss_col_f32_kernel(const float *__restrict__ tmp, float *__restrict__ dst, int h, int rw,
                     int nx, int nbands, TapsF32 taps)
{
    static_assert(R == 8, "ring arithmetic below assumes 8 rows per thread");
    constexpr int kPF = 8;                 // taps of prefetch distance (covers an L2 round trip)
    // XCD-aware tile order (blocks are dealt round-robin over the 8 XCDs): XCD x handles the
    // column strips x, x+8, ... and walks the row bands of a frame in order, so the rows that
    // consecutive bands share are re-read from that XCD's own L2 instead of the fabric
    const int nxg = (nx + 7) >> 3;
    const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
    const int bx = xcd + 8 * (m % nxg);
    const int band = (m / nxg) % nbands, fz = m / (nxg * nbands);
    const int i = bx * kBlock + threadIdx.x;
    if (bx >= nx || i >= rw)
        return;
    const int y0 = band * R;
    const int r = taps.ksize >> 1;
    const float *frame = tmp + (size_t)fz * h * rw + i;
    float *out = dst + (size_t)fz * h * rw + i;
    auto rowv = [&](int y) { return frame[(size_t)reflect101(y, h) * rw]; };

    float up[R], dn[R], acc[R];            // up[s]: row y with (y - y0) % 8 == s, same for dn
#pragma unroll
    for (int j = 0; j < R; j++) {
        up[j] = dn[j] = rowv(y0 + j);
        acc[j] = fmaf(up[j], taps.t[r], 0.0f);
    }
    float pu[kPF], pd[kPF];                // rows arriving for taps k .. k+kPF-1
#pragma unroll
    for (int q = 0; q < kPF; q++) {
        pu[q] = rowv(y0 + R - 1 + 1 + q);
        pd[q] = rowv(y0 - 1 - q);
    }
    for (int k0 = 1; k0 <= r; k0 += 8) {   // k0 = 1 (mod 8): every ring / queue slot below is static
        float wk8[8];                          // this period's taps, one batch of scalar loads
#pragma unroll
        for (int t = 0; t < 8; t++)
            wk8[t] = taps.t[min(r + k0 + t, kMaxTaps)];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int k = k0 + t;
            if (k <= r) {                               // uniform
                const float wk = wk8[t];
                up[t] = pu[t % kPF];                    // row y0+7+k replaces row y0+k-1
                dn[7 - t] = pd[t % kPF];                // row y0-k   replaces row y0+8-k
                if (k + kPF <= r) {                     // uniform: fetch the rows of tap k+kPF
                    pu[t % kPF] = rowv(y0 + R - 1 + k + kPF);
                    pd[t % kPF] = rowv(y0 - k - kPF);
                }
#pragma unroll
                for (int j = 0; j < R; j++)
                    acc[j] = fmaf(up[(j + 1 + t) % 8] + dn[(j + 7 - t) % 8], wk, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < R; j++)
        if (y0 + j < h)
            out[(size_t)(y0 + j) * rw] = acc[j];
}

}  // namespace

bool gauss_f32_fast_supported(int w, int c, const TapsF32 &taps)
{
    const int r = taps.ksize / 2;
    if (c != 1 && c != 3)
        return false;
    if (r > 127 || w < 1)
        return false;
    return ((size_t)w * c + 2 * (size_t)r * c + 32) * sizeof(float) <= 64 * 1024;   // staged row fits LDS
}

int launch_gauss_f32_fast(const float *src, float *dst, float *scratch, int n, int h, int w, int c,
                          const TapsF32 &taps, hipStream_t st)
{
    VA_REQUIRE(gauss_f32_fast_supported(w, c, taps), "float gaussian: unsupported shape/kernel");
    if (n == 0 || h == 0)
        return VA_OK;
    const int r = taps.ksize / 2;
    const int rw = w * c;
    // + 32 samples: the last (partial) segment of a row reads a little past the staged row
    const size_t lds = ((size_t)rw + 2 * (size_t)r * c + 32) * sizeof(float);
    const unsigned rows = (unsigned)((size_t)n * h);
    const int vec4 = (rw % 4 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0);
    if (c == 1)
        gauss_row_f32_kernel<1, 17><<<rows, kBlock, lds, st>>>(src, scratch, w, taps, vec4);
    else
        gauss_row_f32_kernel<3, 27><<<rows, kBlock, lds, st>>>(src, scratch, w, taps, vec4);
    VA_LAUNCH_CHECK("gauss_row_f32_kernel");
    const int nx = cdiv(rw, kBlock), nbands = cdiv(h, 8);
    const unsigned grid = (unsigned)(8 * ((nx + 7) / 8) * nbands * n);
    gauss_col_f32_kernel<8><<<grid, kBlock, 0, st>>>(scratch, dst, h, rw, nx, nbands, taps);
    VA_LAUNCH_CHECK("gauss_col_f32_kernel");
    return VA_OK;
}

}  // namespace va
