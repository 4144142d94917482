// va_gauss.hip -- Gaussian blur (A1)
//
// replaces cv2.GaussianBlur(frame.astype(np.uint8), (0, 0), sigma)
//          FilterBlur._process_frame, video/filters.py:388-392           (8-bit fixed point)
//      and cv2.GaussianBlur(float image, (0,0), sigma), video/analysis/active_contour.py:108
//
// Two implementations:
//   * generic  : any radius <= 127, any channel count; row pass -> scratch in HBM -> column
//                pass.  Fallback and cross-check.
//   * fused    : va_gauss_fused.hip -- single channel u8, radius <= 31, LDS-staged, one read
//                of the input and one write of the output (optionally thresholded + bit-packed).
#include <math.h>

#include "va_common.h"

namespace va {

// ---------------------------------------------------------------------------- host: taps
static int cv_round(double v) { return (int)lrint(v); }  // round-half-even like cvRound

int gauss_ksize(double sigma, bool is_u8)
{
    return cv_round(sigma * (is_u8 ? 3 : 4) * 2 + 1) | 1;
}

static int taps_f64(double sigma, int n, double *out)
{
    // OpenCV getGaussianKernelBitExact: x runs over half-integer steps, hence -0.125
    const int n2 = (n - 1) / 2;
    const double scale2x = -0.125 / (sigma * sigma);
    double sum = 0.0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        out[i] = exp((double)(x * x) * scale2x);
        sum += out[i];
    }
    sum *= 2.0;
    sum += 1.0;
    const double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; i++) {
        double t = out[i] * mul1;
        out[i] = t;
        out[n - 1 - i] = t;
    }
    out[n2] = 1.0 * mul1;
    return 0;
}

int gauss_taps_q8(double sigma, int *ksize, uint16_t *taps, int cap, int rule)
{
    VA_REQUIRE(sigma > 0 && sigma == sigma, "gaussian: sigma must be > 0 (got %g)", sigma);
    VA_REQUIRE(rule == VA_TAPS_CV4 || rule == VA_TAPS_CV3, "gaussian: unknown tap rule %d", rule);
    const int n = gauss_ksize(sigma, true);
    if (n > cap) {
        set_error("gaussian: sigma=%g needs %d taps, more than the supported %d", sigma, n, cap);
        return VA_ERR_RANGE;
    }
    *ksize = n;
    if (rule == VA_TAPS_CV3) {
        // OpenCV 2.4 / 3.x (the reference's era, video/analysis/regions.py:180-182): float32
        // getGaussianKernel -- exp values rounded to float, summed in double, scaled, rounded to float --
        // then every tap cvRound(k * 256) on its own; the sum is not forced to 256
        float cf[kMaxTaps + 1];
        const double scale2x = -0.5 / (sigma * sigma);
        double sum = 0.0;
        for (int i = 0; i < n; i++) {
            const double x = i - (n - 1) * 0.5;
            cf[i] = (float)exp(scale2x * x * x);
            sum += cf[i];
        }
        sum = 1.0 / sum;
        for (int i = 0; i < n; i++) {
            const float k = (float)(cf[i] * sum);
            taps[i] = (uint16_t)cv_round((double)(k * 256.0f));
        }
        return VA_OK;
    }
    double k[kMaxTaps + 1];
    taps_f64(sigma, n, k);
    // getGaussianKernelFixedPoint_ED: error diffusion from the tails, centre takes the rest
    const int n2 = n / 2;
    double err = 0.0;
    long long sum = 0;
    for (int i = 0; i < n2; i++) {
        double adj = k[i] * 256.0 + err;
        long long v0 = cv_round(adj);
        err = adj - (double)v0;
        taps[i] = (uint16_t)v0;
        taps[n - 1 - i] = (uint16_t)v0;
        sum += v0;
    }
    taps[n2] = (uint16_t)(256 - 2 * sum);
    return VA_OK;
}

int gauss_taps_f32(double sigma, int *ksize, float *taps, int cap)
{
    VA_REQUIRE(sigma > 0 && sigma == sigma, "gaussian: sigma must be > 0 (got %g)", sigma);
    const int n = gauss_ksize(sigma, false);
    if (n > cap) {
        set_error("gaussian: sigma=%g needs %d taps, more than the supported %d", sigma, n, cap);
        return VA_ERR_RANGE;
    }
    double k[kMaxTaps + 1];
    taps_f64(sigma, n, k);
    for (int i = 0; i < n; i++)
        taps[i] = (float)k[i];
    *ksize = n;
    return VA_OK;
}

// ------------------------------------------------------------------------ device: generic
namespace {

constexpr int kBlock = 256;

// BORDER_REFLECT_101 with repeated reflection (kernel wider than the image)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// T = uint16_t while the row sums fit (tap sum <= 257: every OpenCV >= 4 tap set), uint32_t otherwise
template <class T>
__global__ void __launch_bounds__(kBlock)
gauss_row_u8_generic(const uint8_t *__restrict__ src, T *__restrict__ tmp, int w, int c,
                     TapsQ8 taps, size_t total)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    int ch = (int)(e % c);
    size_t pix = e / c;
    int x = (int)(pix % w);
    size_t rowi = pix / w;
    const uint8_t *row = src + rowi * (size_t)w * c;
    const int r = taps.ksize >> 1;
    uint32_t acc = 0;
    for (int i = 0; i < taps.ksize; i++)
        acc += (uint32_t)taps.t[i] * row[(size_t)reflect101(x + i - r, w) * c + ch];
    tmp[e] = (T)acc;
}

template <class T>
__global__ void __launch_bounds__(kBlock)
gauss_col_u8_generic(const T *__restrict__ tmp, uint8_t *__restrict__ dst, int h, int w,
                     int c, TapsQ8 taps, size_t total)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    size_t wc = (size_t)w * c;
    size_t col = e % wc;
    size_t rowi = e / wc;
    int y = (int)(rowi % h);
    const T *frame = tmp + (rowi - y) * wc;
    const int r = taps.ksize >> 1;
    uint32_t acc = 0;
    for (int j = 0; j < taps.ksize; j++)
        acc += (uint32_t)taps.t[j] * frame[(size_t)reflect101(y + j - r, h) * wc + col];
    uint32_t v = (acc + 32768u) >> 16;
    dst[e] = (uint8_t)(v > 255u ? 255u : v);
}

__global__ void __launch_bounds__(kBlock)
gauss_row_f32_generic(const float *__restrict__ src, float *__restrict__ tmp, int w, int c,
                      TapsF32 taps, size_t total)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    int ch = (int)(e % c);
    size_t pix = e / c;
    int x = (int)(pix % w);
    size_t rowi = pix / w;
    const float *row = src + rowi * (size_t)w * c;
    const int r = taps.ksize >> 1;
    float acc = 0.0f;
    for (int i = 0; i < taps.ksize; i++)  // in-order fmaf chain (OpenCV RowVec_32f)
        acc = fmaf(row[(size_t)reflect101(x + i - r, w) * c + ch], taps.t[i], acc);
    tmp[e] = acc;
}

__global__ void __launch_bounds__(kBlock)
gauss_col_f32_generic(const float *__restrict__ tmp, float *__restrict__ dst, int h, int w, int c,
                      TapsF32 taps, size_t total)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    size_t wc = (size_t)w * c;
    size_t col = e % wc;
    size_t rowi = e / wc;
    int y = (int)(rowi % h);
    const float *frame = tmp + (rowi - y) * wc;
    const int r = taps.ksize >> 1;
    // centre tap, then symmetric pairs folded (OpenCV SymmColumnVec_32f)
    float acc = fmaf(frame[(size_t)y * wc + col], taps.t[r], 0.0f);
    for (int k = 1; k <= r; k++) {
        float a = frame[(size_t)reflect101(y + k, h) * wc + col];
        float b = frame[(size_t)reflect101(y - k, h) * wc + col];
        acc = fmaf(a + b, taps.t[r + k], acc);
    }
    dst[e] = acc;
}

}  // namespace

static int taps_sum(const TapsQ8 &taps)
{
    int sum = 0;
    for (int i = 0; i < taps.ksize; i++)
        sum += taps.t[i];
    return sum;
}

size_t gauss_generic_u8_scratch_bytes(size_t count, const TapsQ8 &taps)
{
    return count * (taps_sum(taps) <= 257 ? sizeof(uint16_t) : sizeof(uint32_t));     // row sums <= 255 * sum
}

int launch_gauss_generic_u8(const uint8_t *src, uint8_t *dst, void *scratch, int n, int h,
                            int w, int c, const TapsQ8 &taps, hipStream_t st)
{
    size_t total = (size_t)n * h * w * c;
    if (total == 0)
        return VA_OK;
    int grid = cdiv((long long)total, kBlock);
    if (taps_sum(taps) <= 257) {
        gauss_row_u8_generic<uint16_t><<<grid, kBlock, 0, st>>>(src, (uint16_t *)scratch, w, c, taps, total);
        VA_LAUNCH_CHECK("gauss_row_u8_generic");
        gauss_col_u8_generic<uint16_t><<<grid, kBlock, 0, st>>>((const uint16_t *)scratch, dst, h, w, c, taps, total);
    } else {
        gauss_row_u8_generic<uint32_t><<<grid, kBlock, 0, st>>>(src, (uint32_t *)scratch, w, c, taps, total);
        VA_LAUNCH_CHECK("gauss_row_u8_generic");
        gauss_col_u8_generic<uint32_t><<<grid, kBlock, 0, st>>>((const uint32_t *)scratch, dst, h, w, c, taps, total);
    }
    VA_LAUNCH_CHECK("gauss_col_u8_generic");
    return VA_OK;
}

int launch_gauss_generic_f32(const float *src, float *dst, float *scratch, int n, int h, int w,
                             int c, const TapsF32 &taps, hipStream_t st)
{
    size_t total = (size_t)n * h * w * c;
    if (total == 0)
        return VA_OK;
    int grid = cdiv((long long)total, kBlock);
    gauss_row_f32_generic<<<grid, kBlock, 0, st>>>(src, scratch, w, c, taps, total);
    VA_LAUNCH_CHECK("gauss_row_f32_generic");
    gauss_col_f32_generic<<<grid, kBlock, 0, st>>>(scratch, dst, h, w, c, taps, total);
    VA_LAUNCH_CHECK("gauss_col_f32_generic");
    return VA_OK;
}

}  // namespace va
