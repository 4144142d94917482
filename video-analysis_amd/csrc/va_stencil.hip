// va_stencil.hip -- the small stencils next to the hot path:
//   detect_peaks          video/analysis/image.py:267-306  (3x3 maximum filter + eroded background)
//   mask_thinning (python) video/analysis/image.py:243-258  (iterated 3x3-cross erode/dilate)
//   get_image_statistics  video/analysis/image.py:131-201  (window sums -> local mean / variance)
// One thread per pixel; the 3x3 / 5-point neighbourhoods are served by L1 (each byte is re-read
// by its 8 neighbours within the same wave or the next row's wave).
#include "va_common.h"

namespace va {

namespace {

constexpr int kBlock = 256;

// scipy.ndimage 'reflect' border (d c b a | a b c d | d c b a) for a 3x3 window == clamping
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// detect_peaks: include_plateaus -> (img == max over the 3x3 window) XOR (whole 3x3 window is
// background, pixels outside the frame counting as background); else img > max over the 8
// neighbours
__global__ void __launch_bounds__(kBlock)
detect_peaks_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int h, int w,
                    size_t total, int include_plateaus)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    const int x = (int)(e % w);
    const size_t rowi = e / w;
    const int y = (int)(rowi % h);
    const uint8_t *frame = src + (rowi - y) * (size_t)w;
    const int c = frame[(size_t)y * w + x];
    int mx_all = 0, mx_nb = 0, any_fg = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int yy = clampi(y + dy, 0, h - 1), xx = clampi(x + dx, 0, w - 1);
            const int v = frame[(size_t)yy * w + xx];
            mx_all = v > mx_all ? v : mx_all;
            if (dx != 0 || dy != 0)
                mx_nb = v > mx_nb ? v : mx_nb;
            // erosion of (img == 0) with border_value = 1: outside pixels are background
            const bool inside = (y + dy >= 0 && y + dy < h && x + dx >= 0 && x + dx < w);
            any_fg |= inside && frame[(size_t)(y + dy) * w + (x + dx)] != 0;
        }
    int out;
    if (include_plateaus)
        out = (c == mx_all) != (any_fg == 0);
    else
        out = c > mx_nb;
    dst[e] = (uint8_t)out;
}

// one iteration of the python mask_thinning loop (MORPH_CROSS 3x3, OpenCV border = never wins):
//   eroded = erode(img); temp = dilate(eroded); temp = sat(img - temp); skel |= temp; img = eroded
__device__ __forceinline__ int erode_cross(const uint8_t *frame, int x, int y, int h, int w)
{
    int v = frame[(size_t)y * w + x];
    if (x > 0) v = min(v, (int)frame[(size_t)y * w + x - 1]);
    if (x + 1 < w) v = min(v, (int)frame[(size_t)y * w + x + 1]);
    if (y > 0) v = min(v, (int)frame[(size_t)(y - 1) * w + x]);
    if (y + 1 < h) v = min(v, (int)frame[(size_t)(y + 1) * w + x]);
    return v;
}

__global__ void __launch_bounds__(kBlock)
thinning_step_kernel(const uint8_t *__restrict__ img, uint8_t *__restrict__ eroded_out,
                     uint8_t *__restrict__ skel, int h, int w, size_t total,
                     unsigned long long *__restrict__ nonzero)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    int nz = 0;
    if (e < total) {
        const int x = (int)(e % w);
        const size_t rowi = e / w;
        const int y = (int)(rowi % h);
        const uint8_t *frame = img + (rowi - y) * (size_t)w;
        const int ec = erode_cross(frame, x, y, h, w);
        int t = ec;
        if (x > 0) t = max(t, erode_cross(frame, x - 1, y, h, w));
        if (x + 1 < w) t = max(t, erode_cross(frame, x + 1, y, h, w));
        if (y > 0) t = max(t, erode_cross(frame, x, y - 1, h, w));
        if (y + 1 < h) t = max(t, erode_cross(frame, x, y + 1, h, w));
        const int c = frame[(size_t)y * w + x];
        const int d = c > t ? c - t : 0;          // cv2.subtract saturates
        skel[e] = (uint8_t)(skel[e] | d);         // cv2.bitwise_or
        eroded_out[e] = (uint8_t)ec;
        nz = ec != 0;
    }
    const unsigned long long b = __ballot(nz);
    if ((threadIdx.x & 63) == 0 && b)
        atomicAdd(nonzero, (unsigned long long)__popcll(b));
}

// get_image_statistics: window sums of (img - prior) and its square over a row-span element
// with zero border (cv2.boxFilter(normalize=False) / cv2.filter2D(ellipse), BORDER_CONSTANT):
//   mean = s1/count + prior ; var = (s2 - s1^2/count)/(count - 1)
__global__ void __launch_bounds__(kBlock)
image_statistics_kernel(const uint8_t *__restrict__ src, double *__restrict__ mean_out,
                        double *__restrict__ var_out, int h, int w, size_t total, RowSpans se,
                        double prior, int exclude_center, double count)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    const int x = (int)(e % w);
    const size_t rowi = e / w;
    const int y = (int)(rowi % h);
    const uint8_t *frame = src + (rowi - y) * (size_t)w;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < se.ksize; i++) {
        const int yy = y + i - se.anchor;
        if (yy < 0 || yy >= h)
            continue;
        int x0 = x + se.lo[i] - se.anchor, x1 = x + se.hi[i] - 1 - se.anchor;
        x0 = x0 < 0 ? 0 : x0;
        x1 = x1 >= w ? w - 1 : x1;
        const uint8_t *row = frame + (size_t)yy * w;
        for (int xx = x0; xx <= x1; xx++) {
            const double d = (double)row[xx] - prior;
            s1 += d;
            s2 += d * d;
        }
    }
    if (exclude_center) {
        const double d = (double)frame[(size_t)y * w + x] - prior;
        s1 = s1 - d;
        s2 = s2 - d * d;
    }
    mean_out[e] = s1 / count + prior;
    if (var_out)
        var_out[e] = (s2 - s1 * s1 / count) / (count - 1.0);
}

}  // namespace

int launch_detect_peaks(const uint8_t *src, uint8_t *dst, int n, int h, int w, int include_plateaus,
                        hipStream_t st)
{
    size_t total = (size_t)n * h * w;
    if (total == 0)
        return VA_OK;
    detect_peaks_kernel<<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(src, dst, h, w, total,
                                                                          include_plateaus);
    VA_LAUNCH_CHECK("detect_peaks_kernel");
    return VA_OK;
}

int launch_thinning_step(const uint8_t *img, uint8_t *eroded, uint8_t *skel, int n, int h, int w,
                         unsigned long long *nonzero, hipStream_t st)
{
    size_t total = (size_t)n * h * w;
    if (total == 0)
        return VA_OK;
    // `nonzero` is this iteration's own counter; the caller zeroes the counter array once
    thinning_step_kernel<<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(img, eroded, skel, h, w,
                                                                           total, nonzero);
    VA_LAUNCH_CHECK("thinning_step_kernel");
    return VA_OK;
}

int launch_image_statistics(const uint8_t *src, double *mean_out, double *var_out, int n, int h,
                            int w, const RowSpans &se, double prior, int exclude_center,
                            hipStream_t st)
{
    size_t total = (size_t)n * h * w;
    if (total == 0)
        return VA_OK;
    double count = 0;
    for (int i = 0; i < se.ksize; i++)
        count += se.hi[i] > se.lo[i] ? se.hi[i] - se.lo[i] : 0;
    if (exclude_center)
        count -= 1;
    image_statistics_kernel<<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(
        src, mean_out, var_out, h, w, total, se, prior, exclude_center, count);
    VA_LAUNCH_CHECK("image_statistics_kernel");
    return VA_OK;
}

}  // namespace va
