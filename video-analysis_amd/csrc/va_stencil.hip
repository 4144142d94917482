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
// (T = uint8_t or float: the reference feeds it float maps as well; the maximum starts from the centre
//  pixel, so negative values are handled)
template <class T>
__global__ void __launch_bounds__(kBlock)
detect_peaks_kernel(const T *__restrict__ src, uint8_t *__restrict__ dst, int h, int w,
                    size_t total, int include_plateaus)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    const int x = (int)(e % w);
    const size_t rowi = e / w;
    const int y = (int)(rowi % h);
    const T *frame = src + (rowi - y) * (size_t)w;
    const T c = frame[(size_t)y * w + x];
    T mx_all = c, mx_nb = c;
    bool nb_set = false;
    int any_fg = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int yy = clampi(y + dy, 0, h - 1), xx = clampi(x + dx, 0, w - 1);
            const T v = frame[(size_t)yy * w + xx];
            mx_all = v > mx_all ? v : mx_all;
            if (dx != 0 || dy != 0) {
                mx_nb = (!nb_set || v > mx_nb) ? v : mx_nb;
                nb_set = true;
            }
            // erosion of (img == 0) with border_value = 1: outside pixels are background
            const bool inside = (y + dy >= 0 && y + dy < h && x + dx >= 0 && x + dx < w);
            any_fg |= inside && frame[(size_t)(y + dy) * w + (x + dx)] != (T)0;
        }
    int out;
    if (include_plateaus)
        out = (c == mx_all) != (any_fg == 0);
    else
        out = c > mx_nb;
    dst[e] = (uint8_t)out;
}

// uint8 fast path: a thread owns FOUR consecutive pixels (w % 4 == 0, 4-byte aligned buffers): three dword loads
// + six edge bytes instead of eighteen byte loads per pixel, separable 3-maxima (v_max3_u32), one dword store.
// Edge replication never changes a 3x3 maximum, and for unsigned pixels "some pixel of the window is foreground"
// is "the window's maximum is not 0" (the replicated pixel is itself inside the window), so the eroded-background
// term of the reference costs nothing extra.
__global__ void __launch_bounds__(kBlock)
detect_peaks_u8x4_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int h, int w, size_t quads,
                         int include_plateaus)
{
    const size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= quads)
        return;
    const int wq = w >> 2;
    const int xq = (int)(q % wq);
    const size_t rowi = q / wq;
    const int y = (int)(rowi % h), x0 = 4 * xq;
    const uint8_t *frame = src + (rowi - y) * (size_t)w;
    uint32_t hmax[3][4], mid[6];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const uint8_t *row = frame + (size_t)clampi(y + r - 1, 0, h - 1) * w;
        const uint32_t wd = *reinterpret_cast<const uint32_t *>(row + x0);
        uint32_t v[6];
        v[0] = row[x0 > 0 ? x0 - 1 : 0];
        v[1] = wd & 255u;
        v[2] = (wd >> 8) & 255u;
        v[3] = (wd >> 16) & 255u;
        v[4] = wd >> 24;
        v[5] = row[x0 + 4 < w ? x0 + 4 : w - 1];
#pragma unroll
        for (int p = 0; p < 4; p++)
            hmax[r][p] = max(max(v[p], v[p + 1]), v[p + 2]);
        if (r == 1)
#pragma unroll
            for (int k = 0; k < 6; k++)
                mid[k] = v[k];
    }
    uint32_t out = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const uint32_t c = mid[p + 1];
        uint32_t bit;
        if (include_plateaus) {
            const uint32_t mx = max(max(hmax[0][p], hmax[1][p]), hmax[2][p]);
            bit = (uint32_t)((c == mx) != (mx == 0u));
        } else {
            const uint32_t nb = max(max(hmax[0][p], hmax[2][p]), max(mid[p], mid[p + 2]));
            bit = (uint32_t)(c > nb);
        }
        out |= bit << (8 * p);
    }
    *reinterpret_cast<uint32_t *>(dst + 4 * q) = out;
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
    if (__ballot(nz) && (threadIdx.x & 63) == 0)      // (a flag, not a count: see thinning_tail_x4_kernel)
        *nonzero = 1ull;
}

// The second half of a thinning step for rows of whole dwords, four samples per thread: `eroded` is already there
// (va_morph_u8's four-sample kernel); temp = dilate(eroded) from five dwords of it, sat(img - temp), skel |=, count.
// Bytewise extremes and the saturating difference run on pairs of 16-bit lanes (even bytes, odd bytes).
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(kBlock)
thinning_tail_x4_kernel(const uint8_t *__restrict__ img, const uint8_t *__restrict__ eroded, uint8_t *__restrict__ skel,
                        int h, int w, unsigned long long *__restrict__ nonzero)
{
    const int q = blockIdx.x * kBlock + threadIdx.x, wq = w >> 2;
    const int y = blockIdx.y;
    const size_t f = blockIdx.z;
    unsigned cnt = 0;
    if (q < wq) {
        const uint32_t *E = reinterpret_cast<const uint32_t *>(eroded + f * (size_t)h * w);
        const size_t at = (size_t)y * wq + q;
        const uint32_t c = E[at];
        const uint32_t l = q > 0 ? E[at - 1] : 0u, r = q + 1 < wq ? E[at + 1] : 0u;          // the border never wins
        const uint32_t u = y > 0 ? E[at - wq] : 0u, d = y + 1 < h ? E[at + wq] : 0u;
        const uint32_t left = __builtin_amdgcn_alignbyte(c, l, 3), right = __builtin_amdgcn_alignbyte(r, c, 1);
        auto even = [](uint32_t v) { return __builtin_bit_cast(us2, v & 0x00FF00FFu); };
        auto odd = [](uint32_t v) { return __builtin_bit_cast(us2, (v >> 8) & 0x00FF00FFu); };
        us2 te = even(c), to = odd(c);
        for (uint32_t v : {left, right, u, d}) {
            te = __builtin_elementwise_max(te, even(v));
            to = __builtin_elementwise_max(to, odd(v));
        }
        const uint32_t px = reinterpret_cast<const uint32_t *>(img + f * (size_t)h * w)[at];
        const us2 de = __builtin_elementwise_sub_sat(even(px), te), dd = __builtin_elementwise_sub_sat(odd(px), to);
        const uint32_t diff = __builtin_bit_cast(uint32_t, de) | (__builtin_bit_cast(uint32_t, dd) << 8);
        uint32_t *sk = reinterpret_cast<uint32_t *>(skel + f * (size_t)h * w);
        sk[at] |= diff;
        cnt = __popc((((c & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | c) & 0x80808080u);                  // non-zero bytes of c
    }
    // the host only asks whether anything survived: a plain store of 1 (8 k waves adding to one counter took most
    // of this kernel's 18 us)
    if (__ballot(cnt != 0) && (threadIdx.x & 63) == 0)
        *nonzero = 1ull;
}

// get_image_statistics: window sums of (img - prior) and its square over a row-span element
// with zero border (cv2.boxFilter(normalize=False) / cv2.filter2D(ellipse), BORDER_CONSTANT):
//   mean = s1/count + prior ; var = (s2 - s1^2/count)/(count - 1)
// Generic fallback (frames wider than 65536 pixels): O(window area) float64 additions per pixel.
// T = float: the reference's `img.astype(np.int) - prior` truncates float images towards zero first
template <class T>
__device__ __forceinline__ double as_int_value(T v) { return (double)v; }
template <>
__device__ __forceinline__ double as_int_value<float>(float v) { return (double)(long long)v; }

template <class T>
__global__ void __launch_bounds__(kBlock)
image_statistics_kernel(const T *__restrict__ src, double *__restrict__ mean_out,
                        double *__restrict__ var_out, int h, int w, size_t total, RowSpans se,
                        double prior, int exclude_center, double count)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    const int x = (int)(e % w);
    const size_t rowi = e / w;
    const int y = (int)(rowi % h);
    const T *frame = src + (rowi - y) * (size_t)w;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < se.ksize; i++) {
        const int yy = y + i - se.anchor;
        if (yy < 0 || yy >= h)
            continue;
        int x0 = x + se.lo[i] - se.anchor, x1 = x + se.hi[i] - 1 - se.anchor;
        x0 = x0 < 0 ? 0 : x0;
        x1 = x1 >= w ? w - 1 : x1;
        const T *row = frame + (size_t)yy * w;
        for (int xx = x0; xx <= x1; xx++) {
            const double d = as_int_value<T>(row[xx]) - prior;
            s1 += d;
            s2 += d * d;
        }
    }
    if (exclude_center) {
        const double d = as_int_value<T>(frame[(size_t)y * w + x]) - prior;
        s1 = s1 - d;
        s2 = s2 - d * d;
    }
    mean_out[e] = s1 / count + prior;
    if (var_out)
        var_out[e] = (s2 - s1 * s1 / count) / (count - 1.0);
}

// The fast path: the structuring element is a set of horizontal spans, so a window sum is a sum
// over the window's rows of (prefix[x1 + 1] - prefix[x0]).  Pass 1 writes, per image row, the
// exclusive prefix sums of img and img^2 (exact integers: uint32 holds 65025 * 65536); pass 2
// needs 4 loads per window row instead of one load and two float64 additions per window PIXEL,
// and forms  sum(img - prior) = S1 - N prior,  sum((img - prior)^2) = S2 - 2 prior S1 + N prior^2
// with N = the number of in-image pixels under the element (zero border).  For integer priors
// every quantity is an exact integer below 2^53, i.e. the results equal the direct sums bit for
// bit; for fractional priors they agree to rounding (1e-12 relative on the mean).
__global__ void __launch_bounds__(kBlock)
row_prefix_kernel(const uint8_t *__restrict__ src, uint32_t *__restrict__ p1, uint32_t *__restrict__ p2, int w,
                  size_t total_rows)
{
    const size_t row = (size_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (row >= total_rows)
        return;
    const int lane = threadIdx.x & 63;
    const uint8_t *s = src + row * (size_t)w;
    uint32_t *o1 = p1 + row * (size_t)(w + 1), *o2 = p2 + row * (size_t)(w + 1);
    const int per = (w + kWave - 1) / kWave;            // consecutive pixels per lane
    const int xa = min(lane * per, w), xb = min(xa + per, w);
    uint32_t a1 = 0, a2 = 0;
    for (int x = xa; x < xb; x++) {
        const uint32_t v = s[x];
        a1 += v;
        a2 += v * v;
    }
    uint32_t e1 = a1, e2 = a2;                          // inclusive scan over the lanes' totals
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t t1 = __shfl_up(e1, d, kWave), t2 = __shfl_up(e2, d, kWave);
        if (lane >= d) {
            e1 += t1;
            e2 += t2;
        }
    }
    uint32_t r1 = e1 - a1, r2 = e2 - a2;                // exclusive: sum of everything left of xa
    for (int x = xa; x < xb; x++) {
        o1[x] = r1;
        o2[x] = r2;
        const uint32_t v = s[x];
        r1 += v;
        r2 += v * v;
    }
    if (xb == w) {          // the lane(s) whose range ends at the row end hold the row total
        o1[w] = r1;
        o2[w] = r2;
    }
}

__global__ void __launch_bounds__(kBlock)
image_statistics_prefix_kernel(const uint8_t *__restrict__ src, const uint32_t *__restrict__ p1,
                               const uint32_t *__restrict__ p2, double *__restrict__ mean_out,
                               double *__restrict__ var_out, int h, int w, size_t total, RowSpans se,
                               double prior, int exclude_center, double count)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    const int x = (int)(e % w);
    const size_t rowi = e / w;
    const int y = (int)(rowi % h);
    const size_t frow0 = rowi - y;                      // first row of this frame
    long long S1 = 0, S2 = 0, N = 0;
    for (int i = 0; i < se.ksize; i++) {
        const int yy = y + i - se.anchor;
        if (yy < 0 || yy >= h)
            continue;
        int x0 = x + se.lo[i] - se.anchor, x1 = x + se.hi[i] - se.anchor;     // [x0, x1)
        x0 = x0 < 0 ? 0 : x0;
        x1 = x1 > w ? w : x1;
        if (x1 <= x0)
            continue;
        const size_t base = (frow0 + yy) * (size_t)(w + 1);
        S1 += (long long)(p1[base + x1] - p1[base + x0]);
        S2 += (long long)(p2[base + x1] - p2[base + x0]);
        N += x1 - x0;
    }
    if (exclude_center) {
        const long long c = src[e];
        S1 -= c;
        S2 -= c * c;
        N -= 1;
    }
    const double dS1 = (double)S1, dN = (double)N;
    const double s1 = dS1 - dN * prior;
    const double s2 = ((double)S2 - 2.0 * prior * dS1) + dN * prior * prior;
    mean_out[e] = s1 / count + prior;
    if (var_out)
        var_out[e] = (s2 - s1 * s1 / count) / (count - 1.0);
}

// Box windows up to 31 x 31 (ksize <= 15), uint8: one workgroup per 64 x 64 tile, everything in LDS -- the tile
// with its zero border, sliding horizontal window sums per row, sliding vertical sums per column -- so a pixel costs
// 1.3 bytes read and 16 written instead of 4 gathers per window row from the prefix arrays.  The sums are the same
// integers; the float64 arithmetic after them is the prefix kernel's.
constexpr int kStT = 64, kStMaxK = 31, kStR = kStT + kStMaxK - 1;
__global__ void __launch_bounds__(256)
image_statistics_box_kernel(const uint8_t *__restrict__ src, double *__restrict__ mean_out,
                            double *__restrict__ var_out, int h, int w, int tiles_x, int tiles_y, int K, int anchor,
                            double prior, int exclude_center, double count)
{
    __shared__ uint8_t s_px[kStR][kStR + 2];
    __shared__ uint32_t s_h1[kStR][kStT + 1], s_h2[kStR][kStT + 1];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y;
    const size_t f = blockIdx.x / ((size_t)tiles_x * tiles_y);
    const uint8_t *img = src + f * (size_t)h * w;
    const int X0 = tx * kStT, Y0 = ty * kStT, R = kStT + K - 1;
    for (int idx = tid; idx < R * R; idx += 256) {
        const int r = idx / R, c = idx - r * R;
        const int yy = Y0 - anchor + r, xx = X0 - anchor + c;
        s_px[r][c] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? img[(size_t)yy * w + xx] : (uint8_t)0;
    }
    __syncthreads();
    for (int t = tid; t < R * 4; t += 256) {             // row r, 16 output columns from c0
        const int r = t >> 2, c0 = (t & 3) * 16;
        uint32_t a1 = 0, a2 = 0;
        for (int i = 0; i < K; i++) {
            const uint32_t v = s_px[r][c0 + i];
            a1 += v;
            a2 += v * v;
        }
        for (int c = 0; c < 16; c++) {
            s_h1[r][c0 + c] = a1;
            s_h2[r][c0 + c] = a2;
            const uint32_t in = s_px[r][c0 + c + K], out = s_px[r][c0 + c];   // (column R .. R+1: padding, unused)
            a1 += in - out;
            a2 += in * in - out * out;
        }
    }
    __syncthreads();
    const int col = tid & 63, y0 = (tid >> 6) * 16;
    const int x = X0 + col;
    uint32_t v1 = 0, v2 = 0;
    for (int i = 0; i < K; i++) {
        v1 += s_h1[y0 + i][col];
        v2 += s_h2[y0 + i][col];
    }
    const int nx = min(w, x - anchor + K) - max(0, x - anchor);
    for (int j = 0; j < 16; j++) {
        const int y = Y0 + y0 + j;
        if (x < w && y < h) {
            long long S1 = v1, S2 = v2;
            long long N = (long long)nx * (min(h, y - anchor + K) - max(0, y - anchor));
            if (exclude_center) {
                const long long c = s_px[y0 + j + anchor][col + anchor];
                S1 -= c;
                S2 -= c * c;
                N -= 1;
            }
            const double dS1 = (double)S1, dN = (double)N;
            const double s1 = dS1 - dN * prior;
            const double s2 = ((double)S2 - 2.0 * prior * dS1) + dN * prior * prior;
            const size_t e = (f * h + y) * (size_t)w + x;
            mean_out[e] = s1 / count + prior;
            if (var_out)
                var_out[e] = (s2 - s1 * s1 / count) / (count - 1.0);
        }
        if (j < 15) {
            v1 += s_h1[y0 + j + K][col] - s_h1[y0 + j][col];
            v2 += s_h2[y0 + j + K][col] - s_h2[y0 + j][col];
        }
    }
}

}  // namespace

int launch_detect_peaks(const uint8_t *src, uint8_t *dst, int n, int h, int w, int include_plateaus,
                        hipStream_t st)
{
    size_t total = (size_t)n * h * w;
    if (total == 0)
        return VA_OK;
    if (w % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0) {
        detect_peaks_u8x4_kernel<<<cdiv((long long)(total / 4), kBlock), kBlock, 0, st>>>(src, dst, h, w, total / 4,
                                                                                         include_plateaus);
        VA_LAUNCH_CHECK("detect_peaks_u8x4_kernel");
        return VA_OK;
    }
    detect_peaks_kernel<uint8_t><<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(src, dst, h, w, total,
                                                                                   include_plateaus);
    VA_LAUNCH_CHECK("detect_peaks_kernel");
    return VA_OK;
}

int launch_detect_peaks_f32(const float *src, uint8_t *dst, int n, int h, int w, int include_plateaus,
                            hipStream_t st)
{
    size_t total = (size_t)n * h * w;
    if (total == 0)
        return VA_OK;
    detect_peaks_kernel<float><<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(src, dst, h, w, total,
                                                                                 include_plateaus);
    VA_LAUNCH_CHECK("detect_peaks_kernel");
    return VA_OK;
}

static double element_count(const RowSpans &se, int exclude_center)
{
    double count = 0;
    for (int i = 0; i < se.ksize; i++)
        count += se.hi[i] > se.lo[i] ? se.hi[i] - se.lo[i] : 0;
    return exclude_center ? count - 1 : count;
}

// float32 images: truncated to integers like the reference, then direct window sums in float64
int launch_image_statistics_f32(const float *src, double *mean_out, double *var_out, int n, int h, int w,
                                const RowSpans &se, double prior, int exclude_center, hipStream_t st)
{
    size_t total = (size_t)n * h * w;
    if (total == 0)
        return VA_OK;
    image_statistics_kernel<float><<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(
        src, mean_out, var_out, h, w, total, se, prior, exclude_center, element_count(se, exclude_center));
    VA_LAUNCH_CHECK("image_statistics_kernel");
    return VA_OK;
}

int launch_thinning_step(const uint8_t *img, uint8_t *eroded, uint8_t *skel, int n, int h, int w,
                         unsigned long long *nonzero, hipStream_t st)
{
    size_t total = (size_t)n * h * w;
    if (total == 0)
        return VA_OK;
    // `nonzero` is this iteration's own counter; the caller zeroes the counter array once
    // `nonzero` is this iteration's own counter; the caller zeroes the counter array once
    if (w % 4 == 0 && h <= 65535 && n <= 65535 && reinterpret_cast<uintptr_t>(img) % 4 == 0 &&
        reinterpret_cast<uintptr_t>(eroded) % 4 == 0 && reinterpret_cast<uintptr_t>(skel) % 4 == 0) {
        // two launches of four-sample kernels: the erosion (va_morph.hip), then the rest of the step
        RowSpans cross;
        int rc = make_row_spans(VA_SHAPE_CROSS, 3, &cross);
        if (rc)
            return rc;
        rc = launch_morph_u8(img, eroded, n, h, w, VA_MORPH_ERODE, cross, st, nullptr);
        if (rc)
            return rc;
        const dim3 grid((unsigned)cdiv(w / 4, kBlock), (unsigned)h, (unsigned)n);
        thinning_tail_x4_kernel<<<grid, kBlock, 0, st>>>(img, eroded, skel, h, w, nonzero);
        VA_LAUNCH_CHECK("thinning_tail_x4_kernel");
        return VA_OK;
    }
    thinning_step_kernel<<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(img, eroded, skel, h, w,
                                                                           total, nonzero);
    VA_LAUNCH_CHECK("thinning_step_kernel");
    return VA_OK;
}

size_t image_statistics_scratch_bytes(int n, int h, int w)
{
    if (w > 65536)
        return 256;                                     // (generic kernel: no scratch)
    return 2 * (size_t)n * h * ((size_t)w + 1) * sizeof(uint32_t) + 256;
}

int launch_image_statistics(const uint8_t *src, double *mean_out, double *var_out, int n, int h,
                            int w, const RowSpans &se, double prior, int exclude_center,
                            void *scratch, hipStream_t st)
{
    size_t total = (size_t)n * h * w;
    if (total == 0)
        return VA_OK;
    double count = 0;
    for (int i = 0; i < se.ksize; i++)
        count += se.hi[i] > se.lo[i] ? se.hi[i] - se.lo[i] : 0;
    if (exclude_center)
        count -= 1;
    bool box = se.ksize <= kStMaxK;                     // every row the same full span
    for (int i = 0; i < se.ksize; i++)
        box = box && se.lo[i] == 0 && se.hi[i] == se.ksize;
    const long long tiles = (long long)n * cdiv(h, kStT) * cdiv(w, kStT);
    if (box && tiles < (1ll << 31)) {
        image_statistics_box_kernel<<<(unsigned)tiles, 256, 0, st>>>(src, mean_out, var_out, h, w, cdiv(w, kStT),
                                                                      cdiv(h, kStT), se.ksize, se.anchor, prior,
                                                                      exclude_center, count);
        VA_LAUNCH_CHECK("image_statistics_box_kernel");
        return VA_OK;
    }
    if (w <= 65536 && scratch) {
        const size_t rows = (size_t)n * h;
        uint32_t *p1 = (uint32_t *)scratch, *p2 = p1 + rows * ((size_t)w + 1);
        row_prefix_kernel<<<cdiv((long long)rows, kBlock / kWave), kBlock, 0, st>>>(src, p1, p2, w, rows);
        VA_LAUNCH_CHECK("row_prefix_kernel");
        image_statistics_prefix_kernel<<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(
            src, p1, p2, mean_out, var_out, h, w, total, se, prior, exclude_center, count);
        VA_LAUNCH_CHECK("image_statistics_prefix_kernel");
        return VA_OK;
    }
    image_statistics_kernel<uint8_t><<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(
        src, mean_out, var_out, h, w, total, se, prior, exclude_center, count);
    VA_LAUNCH_CHECK("image_statistics_kernel");
    return VA_OK;
}

}  // namespace va
