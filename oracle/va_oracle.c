/*
 * va_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the reference's per-frame hot path
 * (david-zwicker/video-analysis, `video.filters` chain + `video.analysis` image ops).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (video-analysis_amd/) never does.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - labelling (A7)      : PINNED against scipy.ndimage.label, the exact call the
 *                           reference makes (video/analysis/regions.py:162).
 *   - running mean (A2)   : PINNED against the literal NumPy formula of
 *                           video/analysis/video.py:33 and :48-50.
 *   - morphology (A6)     : PINNED against scipy.ndimage grey_erosion/grey_dilation with
 *                           constant neutral border (OpenCV's documented default border).
 *   - raw moments (A9)    : PINNED against a direct NumPy sum of x^p y^q.
 *   - Gaussian blur (A1)  : PARITY UNPINNED against OpenCV itself (cv2 is not installed,
 *                           the reference holds no tests/vectors). It follows OpenCV's
 *                           published 8-bit fixed-point algorithm (see vao_gauss_taps_q8)
 *                           and is sanity-bounded against scipy.ndimage.gaussian_filter.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VAO_MAX_TAPS 1024

/* OpenCV cvRound(): round-half-to-even (lrint in the default rounding mode). */
static int cv_round(double v) { return (int)lrint(v); }

/* BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba), as cv::borderInterpolate does it,
 * including the repeated reflection needed when the kernel is wider than the image. */
static int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    }
    return p;
}

/* ------------------------------------------------------------------------------------
 * A1  FilterBlur._process_frame  (video/filters.py:388-392):
 *         cv2.GaussianBlur(frame.astype(np.uint8), (0, 0), sigma)
 * OpenCV (>=4.5) 8-bit path: ksize = cvRound(6*sigma+1)|1; taps = exp(-x^2/(2 sigma^2))
 * normalised in double, quantised to unsigned 8.8 fixed point with error diffusion from
 * the tails inwards, centre tap = 256 - sum(others); row pass u8*q8.8 -> u16; column pass
 * accumulates u32 and stores (acc + 2^15) >> 16.
 * ---------------------------------------------------------------------------------- */
int vao_gauss_ksize(double sigma, int is_u8)
{
    int k = cv_round(sigma * (is_u8 ? 3 : 4) * 2 + 1) | 1;
    return k;
}

/* normalised double taps: getGaussianKernelBitExact (IEEE double, libm exp) */
static int gauss_taps_f64(double sigma, int n, double *out)
{
    if (n < 1 || n > VAO_MAX_TAPS || !(sigma > 0))
        return -1;
    int n2 = (n - 1) / 2;
    double scale2x = -0.125 / (sigma * sigma);
    double sum = 0.0;
    double *values = (double *)malloc(sizeof(double) * (n2 + 1));
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        double t = exp((double)(x * x) * scale2x);
        values[i] = t;
        sum += t;
    }
    sum *= 2.0;
    sum += 1.0;
    double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; i++) {
        double t = values[i] * mul1;
        out[i] = t;
        out[n - 1 - i] = t;
    }
    out[n2] = 1.0 * mul1;
    free(values);
    return 0;
}

/* The reference-era definition (rule 1, "cv3").  The reference calls cv2.findContours(...)[1]
 * (video/analysis/regions.py:180-182), i.e. OpenCV 2.4 / 3.x, whose 8-bit GaussianBlur
 * (video/filters.py:392) ran sepFilter2D on float32 kernels scaled to integers:
 *   getGaussianKernel(n, sigma, CV_32F): t_i = (float)exp(-x_i^2 / (2 sigma^2)), sum over the float
 *   values in double, k_i = (float)(t_i * (1 / sum));
 *   createSeparableLinearFilter (8-bit source and destination, smooth symmetric kernels):
 *   kernel.convertTo(CV_32S, 256) -> tap_i = cvRound(k_i * 256), every tap on its own (no error
 *   diffusion, the sum is whatever it comes to: 253..259 for sigma = 5), int32 row sums,
 *   column pass (acc + 2^15) >> 16 saturated to 8 bits (FixedPtCastEx<int, uchar>, bits = 16).
 * Written from upstream knowledge like rule 0; unverifiable offline (no cv2 of either era). */
static int gauss_taps_q8_cv3(double sigma, int n, uint16_t *taps)
{
    if (n < 1 || n > VAO_MAX_TAPS || !(sigma > 0))
        return -1;
    float cf[VAO_MAX_TAPS];
    double scale2x = -0.5 / (sigma * sigma), sum = 0.0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        cf[i] = (float)exp(scale2x * x * x);
        sum += cf[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; i++) {
        float k = (float)(cf[i] * sum);
        taps[i] = (uint16_t)cv_round((double)(k * 256.0f));      /* (x 2^8 is exact in float) */
    }
    return 0;
}

int vao_gauss_taps_q8_rule(double sigma, int rule, int *ksize_out, uint16_t *taps)
{
    int n = vao_gauss_ksize(sigma, 1);
    if (rule == 1) {
        if (gauss_taps_q8_cv3(sigma, n, taps))
            return -1;
        *ksize_out = n;
        return 0;
    }
    if (rule != 0)
        return -1;
    double k[VAO_MAX_TAPS];
    if (gauss_taps_f64(sigma, n, k))
        return -1;
    int n2 = n / 2;
    double err = 0.0;
    int64_t sum = 0;
    for (int i = 0; i < n2; i++) {
        double adj = k[i] * 256.0 + err;
        int64_t v0 = cv_round(adj);
        err = adj - (double)v0;
        taps[i] = (uint16_t)v0;
        taps[n - 1 - i] = (uint16_t)v0;
        sum += v0;
    }
    sum *= 2;
    taps[n2] = (uint16_t)(256 - sum);
    *ksize_out = n;
    return 0;
}

int vao_gauss_taps_q8(double sigma, int *ksize_out, uint16_t *taps)
{
    return vao_gauss_taps_q8_rule(sigma, 0, ksize_out, taps);
}

int vao_gauss_taps_f32(double sigma, int *ksize_out, float *taps)
{
    int n = vao_gauss_ksize(sigma, 0);
    double k[VAO_MAX_TAPS];
    if (gauss_taps_f64(sigma, n, k))
        return -1;
    for (int i = 0; i < n; i++)
        taps[i] = (float)k[i];
    *ksize_out = n;
    return 0;
}

/* frames: (n, h, w, c) u8 contiguous; each channel independently.  rule: 0 = OpenCV >= 4.x taps
 * (sum 256: row sums fit 16 bits), 1 = the 2.4 / 3.x taps (any sum: int32 row sums, as there) */
int vao_gaussian_u8_rule(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c,
                         double sigma, int rule)
{
    uint16_t taps[VAO_MAX_TAPS];
    int ks;
    if (vao_gauss_taps_q8_rule(sigma, rule, &ks, taps))
        return -1;
    int r = ks / 2;
    size_t fsz = (size_t)h * w * c;
    uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * fsz);
    int *xi = (int *)malloc(sizeof(int) * (size_t)(w + 2 * r));
    int *yi = (int *)malloc(sizeof(int) * (size_t)(h + 2 * r));
    for (int x = -r; x < w + r; x++)
        xi[x + r] = reflect101(x, w);
    for (int y = -r; y < h + r; y++)
        yi[y + r] = reflect101(y, h);
    for (int f = 0; f < n; f++) {
        const uint8_t *s = src + f * fsz;
        uint8_t *d = dst + f * fsz;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                for (int ch = 0; ch < c; ch++) {
                    uint32_t acc = 0;
                    for (int i = 0; i < ks; i++)
                        acc += (uint32_t)taps[i] * s[((size_t)y * w + xi[x + i]) * c + ch];
                    tmp[((size_t)y * w + x) * c + ch] = acc;
                }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                for (int ch = 0; ch < c; ch++) {
                    uint32_t acc = 0;
                    for (int j = 0; j < ks; j++)
                        acc += (uint32_t)taps[j] * tmp[((size_t)yi[y + j] * w + x) * c + ch];
                    uint32_t v = (acc + 32768u) >> 16;
                    d[((size_t)y * w + x) * c + ch] = (uint8_t)(v > 255 ? 255 : v);
                }
    }
    free(tmp);
    free(xi);
    free(yi);
    return 0;
}

int vao_gaussian_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c, double sigma)
{
    return vao_gaussian_u8_rule(src, dst, n, h, w, c, sigma, 0);
}

/* Float path: cv2.GaussianBlur on a non-8-bit image, as at
 * video/analysis/active_contour.py:108. ksize = cvRound(8*sigma+1)|1, f32 taps; row pass =
 * in-order fmaf chain over the taps, column pass = centre tap then symmetric pairs
 * (S[+k]+S[-k]) folded with fmaf (OpenCV's RowVec_32f / SymmColumnVec_32f order). */
int vao_gaussian_f32(const float *src, float *dst, int n, int h, int w, int c, double sigma)
{
    float taps[VAO_MAX_TAPS];
    int ks;
    if (vao_gauss_taps_f32(sigma, &ks, taps))
        return -1;
    int r = ks / 2;
    size_t fsz = (size_t)h * w * c;
    float *tmp = (float *)malloc(sizeof(float) * fsz);
    int *xi = (int *)malloc(sizeof(int) * (size_t)(w + 2 * r));
    int *yi = (int *)malloc(sizeof(int) * (size_t)(h + 2 * r));
    for (int x = -r; x < w + r; x++)
        xi[x + r] = reflect101(x, w);
    for (int y = -r; y < h + r; y++)
        yi[y + r] = reflect101(y, h);
    for (int f = 0; f < n; f++) {
        const float *s = src + f * fsz;
        float *d = dst + f * fsz;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                for (int ch = 0; ch < c; ch++) {
                    float acc = 0.0f;
                    for (int i = 0; i < ks; i++)
                        acc = fmaf(s[((size_t)y * w + xi[x + i]) * c + ch], taps[i], acc);
                    tmp[((size_t)y * w + x) * c + ch] = acc;
                }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                for (int ch = 0; ch < c; ch++) {
                    float acc = fmaf(tmp[((size_t)yi[y + r] * w + x) * c + ch], taps[r], 0.0f);
                    for (int k = 1; k <= r; k++) {
                        float a = tmp[((size_t)yi[y + r + k] * w + x) * c + ch];
                        float b = tmp[((size_t)yi[y + r - k] * w + x) * c + ch];
                        acc = fmaf(a + b, taps[r + k], acc);
                    }
                    d[((size_t)y * w + x) * c + ch] = acc;
                }
    }
    free(tmp);
    free(xi);
    free(yi);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * A2  measure_mean (video/analysis/video.py:26-35):
 *         mean = mean*n/(n + 1) + frame/(n + 1)          (float64, this operation order)
 *     BUILD-DEFINED FilterBackground(mode='mean'): emit sat_u8(trunc(|frame - bg_prev|)),
 *     then update bg with the formula above. `n_seen` = frames folded into `mean` so far.
 * ---------------------------------------------------------------------------------- */
void vao_bg_mean_u8(const uint8_t *frames, uint8_t *diff_out, double *mean, int64_t n_seen,
                    int n, size_t px)
{
    for (int f = 0; f < n; f++) {
        double dn = (double)(n_seen + f), dn1 = (double)(n_seen + f + 1);
        const uint8_t *s = frames + (size_t)f * px;
        for (size_t i = 0; i < px; i++) {
            double fr = (double)s[i];
            if (diff_out) {
                double d = fabs(fr - mean[i]);
                d = trunc(d);
                diff_out[(size_t)f * px + i] = (uint8_t)(d > 255.0 ? 255.0 : d);
            }
            mean[i] = mean[i] * dn / dn1 + fr / dn1;
        }
    }
}

/* measure_mean / measure_mean_std for frames of dtype 0 = uint8, 3 = int16, 1 = float32 (the codes of the
 * product's header).  NumPy's promotions are part of the definition (video/analysis/video.py:33, 48-50):
 * `frame/(n + 1)` with a float32 frame and a Python int is a FLOAT32 quotient, which only then joins the
 * float64 mean; integer frames divide in float64; Welford's `frame - mean` promotes the frame exactly. */
static double vao_frame_value(const void *frames, int dtype, size_t i)
{
    return dtype == 0 ? (double)((const uint8_t *)frames)[i]
         : dtype == 3 ? (double)((const int16_t *)frames)[i] : (double)((const float *)frames)[i];
}
int vao_mean_any(const void *frames, int dtype, double *mean, int64_t n_seen, int n, size_t px)
{
    if (dtype != 0 && dtype != 1 && dtype != 3)
        return -1;
    for (int f = 0; f < n; f++) {
        double dn = (double)(n_seen + f), dn1 = (double)(n_seen + f + 1);
        for (size_t i = 0; i < px; i++) {
            double q;
            if (dtype == 1)
                q = (double)(((const float *)frames)[(size_t)f * px + i] / (float)dn1);
            else
                q = vao_frame_value(frames, dtype, (size_t)f * px + i) / dn1;
            mean[i] = mean[i] * dn / dn1 + q;
        }
    }
    return 0;
}
int vao_welford_any(const void *frames, int dtype, double *mean, double *m2, int64_t n_seen, int n, size_t px)
{
    if (dtype != 0 && dtype != 1 && dtype != 3)
        return -1;
    for (int f = 0; f < n; f++) {
        double dn1 = (double)(n_seen + f + 1);
        for (size_t i = 0; i < px; i++) {
            double fr = vao_frame_value(frames, dtype, (size_t)f * px + i);
            double delta = fr - mean[i];
            mean[i] = mean[i] + delta / dn1;
            m2[i] = m2[i] + delta * (fr - mean[i]);
        }
    }
    return 0;
}

/* measure_mean_std (video/analysis/video.py:39-55), Welford:
 *   delta = frame - mean; mean = mean + delta/(n+1); M2 = M2 + delta*(frame - mean) */
void vao_welford_u8(const uint8_t *frames, double *mean, double *m2, int64_t n_seen, int n,
                    size_t px)
{
    for (int f = 0; f < n; f++) {
        double dn1 = (double)(n_seen + f + 1);
        const uint8_t *s = frames + (size_t)f * px;
        for (size_t i = 0; i < px; i++) {
            double fr = (double)s[i];
            double delta = fr - mean[i];
            mean[i] = mean[i] + delta / dn1;
            m2[i] = m2[i] + delta * (fr - mean[i]);
        }
    }
}

/* BUILD-DEFINED FilterBackground(mode='ema') (no reference counterpart):
 *   out = |frame - bg|;  bg = bg + rate*(frame - bg)      all float32, no contraction.
 * The first frame ever seen (n_seen == 0) initialises bg = frame and emits 0. */
void vao_bg_ema_f32(const float *frames, float *diff_out, float *bg, int64_t n_seen, float rate,
                    int n, size_t px)
{
    for (int f = 0; f < n; f++) {
        const float *s = frames + (size_t)f * px;
        for (size_t i = 0; i < px; i++) {
            float fr = s[i];
            if (n_seen + f == 0)
                bg[i] = fr;
            float d = fr - bg[i];
            if (diff_out)
                diff_out[(size_t)f * px + i] = fabsf(d);
            float step = rate * d; /* built with -ffp-contract=off: no fma */
            bg[i] = bg[i] + step;
        }
    }
}

/* BUILD-DEFINED u8 variant of the EMA background: state float32,
 *   out = sat_u8(trunc(|frame - bg|)); bg = bg + rate*(frame - bg). */
void vao_bg_ema_u8(const uint8_t *frames, uint8_t *diff_out, float *bg, int64_t n_seen,
                   float rate, int n, size_t px)
{
    for (int f = 0; f < n; f++) {
        const uint8_t *s = frames + (size_t)f * px;
        for (size_t i = 0; i < px; i++) {
            float fr = (float)s[i];
            if (n_seen + f == 0)
                bg[i] = fr;
            float d = fr - bg[i];
            if (diff_out) {
                float a = truncf(fabsf(d));
                diff_out[(size_t)f * px + i] = (uint8_t)(a > 255.0f ? 255.0f : a);
            }
            float step = rate * d;
            bg[i] = bg[i] + step;
        }
    }
}

/* static background: out = sat_u8(trunc(|frame - bg|)), bg float64 unchanged */
void vao_bg_static_u8(const uint8_t *frames, uint8_t *diff_out, const double *bg, int n,
                      size_t px)
{
    for (int f = 0; f < n; f++)
        for (size_t i = 0; i < px; i++) {
            double d = trunc(fabs((double)frames[(size_t)f * px + i] - bg[i]));
            diff_out[(size_t)f * px + i] = (uint8_t)(d > 255.0 ? 255.0 : d);
        }
}

/* A3  FilterTimeDifference._compare_frames (video/filters.py:564-568):
 *         this_frame.astype(int16) - prev_frame      -> int16 */
void vao_time_difference_u8(const uint8_t *this_frame, const uint8_t *prev_frame, int16_t *out,
                            size_t count)
{
    for (size_t i = 0; i < count; i++)
        out[i] = (int16_t)((int16_t)this_frame[i] - (int16_t)prev_frame[i]);
}

/* A4  BUILD-DEFINED FilterThreshold: strict `>` like the reference's boolean-mask idiom
 * (video/analysis/image.py:282,288,304) == cv2.THRESH_BINARY. */
void vao_threshold_u8(const uint8_t *src, uint8_t *dst, size_t count, int thresh, int maxval)
{
    for (size_t i = 0; i < count; i++)
        dst[i] = (uint8_t)(src[i] > thresh ? maxval : 0);
}

/* A5  FilterMonochrome(mode='mean') (video/filters.py:365-366):
 *         np.mean(frame, axis=2).astype(frame.dtype)  -- f64 mean of 3 channels, truncation.
 * NumPy's mean adds the channels in f64 (pairwise == sequential for 3 items) then divides. */
void vao_mono_mean_u8(const uint8_t *src, uint8_t *dst, size_t pixels)
{
    for (size_t i = 0; i < pixels; i++) {
        double s = (double)src[3 * i] + (double)src[3 * i + 1] + (double)src[3 * i + 2];
        dst[i] = (uint8_t)(s / 3.0);
    }
}

/* ------------------------------------------------------------------------------------
 * A6  cv2.erode / cv2.dilate with cv2.getStructuringElement (video/analysis/image.py:248-251)
 * op: 0 = erode (min), 1 = dilate (max). shape: 0 = RECT, 1 = CROSS, 2 = ELLIPSE.
 * anchor = centre (k/2); pixels outside the image never win (OpenCV's default
 * morphologyDefaultBorderValue()).
 * ---------------------------------------------------------------------------------- */
int vao_structuring_element(int shape, int ksize, uint8_t *elem /* ksize*ksize */)
{
    if (ksize < 1)
        return -1;
    int r = ksize / 2, c = ksize / 2;
    double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < ksize; i++) {
        int j1 = 0, j2 = 0;
        if (shape == 0 || (shape == 1 && i == r))
            j2 = ksize;
        else if (shape == 1) {
            j1 = c;
            j2 = j1 + 1;
        } else {
            int dy = i - r;
            if (abs(dy) <= r) {
                int dx = cv_round(c * sqrt((r * r - dy * dy) * inv_r2));
                j1 = c - dx > 0 ? c - dx : 0;
                j2 = c + dx + 1 < ksize ? c + dx + 1 : ksize;
            }
        }
        for (int j = 0; j < ksize; j++)
            elem[i * ksize + j] = (uint8_t)(j >= j1 && j < j2);
    }
    return 0;
}

int vao_morph_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int op, int shape,
                 int ksize)
{
    uint8_t *elem = (uint8_t *)malloc((size_t)ksize * ksize);
    if (vao_structuring_element(shape, ksize, elem)) {
        free(elem);
        return -1;
    }
    int a = ksize / 2;
    if (shape == 0) {
        /* rectangle: separable (1 x k then k x 1), identical result */
        uint8_t *t = (uint8_t *)malloc((size_t)h * w);
        for (int f = 0; f < n; f++) {
            const uint8_t *s = src + (size_t)f * h * w;
            uint8_t *d = dst + (size_t)f * h * w;
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    int lo = x - a < 0 ? 0 : x - a, hi = x - a + ksize > w ? w : x - a + ksize;
                    int best = op ? 0 : 255;
                    for (int xx = lo; xx < hi; xx++) {
                        int v = s[(size_t)y * w + xx];
                        if (op ? v > best : v < best)
                            best = v;
                    }
                    t[(size_t)y * w + x] = (uint8_t)best;
                }
            for (int y = 0; y < h; y++) {
                int lo = y - a < 0 ? 0 : y - a, hi = y - a + ksize > h ? h : y - a + ksize;
                for (int x = 0; x < w; x++) {
                    int best = op ? 0 : 255;
                    for (int yy = lo; yy < hi; yy++) {
                        int v = t[(size_t)yy * w + x];
                        if (op ? v > best : v < best)
                            best = v;
                    }
                    d[(size_t)y * w + x] = (uint8_t)best;
                }
            }
        }
        free(t);
        free(elem);
        return 0;
    }
    for (int f = 0; f < n; f++) {
        const uint8_t *s = src + (size_t)f * h * w;
        uint8_t *d = dst + (size_t)f * h * w;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int best = op ? 0 : 255;
                for (int i = 0; i < ksize; i++) {
                    int yy = y + i - a;
                    if (yy < 0 || yy >= h)
                        continue;
                    for (int j = 0; j < ksize; j++) {
                        int xx = x + j - a;
                        if (xx < 0 || xx >= w || !elem[i * ksize + j])
                            continue;
                        int v = s[(size_t)yy * w + xx];
                        if (op ? v > best : v < best)
                            best = v;
                    }
                }
                d[(size_t)y * w + x] = (uint8_t)best;
            }
    }
    free(elem);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * A7  get_largest_region (video/analysis/regions.py:159-174):
 *         labels, num_features = ndimage.measurements.label(mask)
 * 4-connectivity by default (SciPy's default structure), any non-zero = foreground, int32
 * labels 1..L numbered in raster order of each component's first pixel.
 * Two-pass union-find with min-index roots, then raster renumbering.
 * ---------------------------------------------------------------------------------- */
static int32_t uf_find(int32_t *p, int32_t a)
{
    while (p[a] != a) {
        p[a] = p[p[a]];
        a = p[a];
    }
    return a;
}
static void uf_union(int32_t *p, int32_t a, int32_t b)
{
    a = uf_find(p, a);
    b = uf_find(p, b);
    if (a < b)
        p[b] = a;
    else if (b < a)
        p[a] = b;
}

int vao_label_i32(const uint8_t *mask, int32_t *labels, int h, int w, int connectivity)
{
    size_t px = (size_t)h * w;
    int32_t *p = (int32_t *)malloc(sizeof(int32_t) * (px ? px : 1));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t i = y * w + x;
            if (!mask[i]) {
                p[i] = -1;
                continue;
            }
            p[i] = i;
            if (x > 0 && mask[i - 1])
                uf_union(p, i, i - 1);
            if (y > 0) {
                if (mask[i - w])
                    uf_union(p, i, i - w);
                if (connectivity == 8) {
                    if (x > 0 && mask[i - w - 1])
                        uf_union(p, i, i - w - 1);
                    if (x + 1 < w && mask[i - w + 1])
                        uf_union(p, i, i - w + 1);
                }
            }
        }
    int32_t next = 0;
    for (size_t i = 0; i < px; i++) {
        if (p[i] < 0) {
            labels[i] = 0;
            continue;
        }
        int32_t r = uf_find(p, (int32_t)i);
        if (r == (int32_t)i)
            labels[i] = ++next; /* first pixel of a new component in raster order */
        else
            labels[i] = labels[r];
    }
    free(p);
    return next;
}

/* batch wrapper: counts[f] = number of components of frame f */
void vao_label_batch_i32(const uint8_t *mask, int32_t *labels, int32_t *counts, int n, int h,
                         int w, int connectivity)
{
    for (int f = 0; f < n; f++)
        counts[f] = vao_label_i32(mask + (size_t)f * h * w, labels + (size_t)f * h * w, h, w,
                                  connectivity);
}

/* ------------------------------------------------------------------------------------
 * A7/A8/A9 per-label statistics from a label image.
 * stats[(l-1)*16 + k], k = 0 area(m00) 1 m10 2 m01 3 m20 4 m11 5 m02 6 m30 7 m21 8 m12
 * 9 m03 10 xmin 11 ymin 12 xmax 13 ymax 14,15 reserved(0).
 * Follows np.sum(labels == label) (video/analysis/regions.py:165-166), find_bounding_box
 * (regions.py:113-149) and cv2.moments(mask.astype(np.uint8)) spatial moments
 * m_pq = sum x^p y^q (video/analysis/image.py:353) with unit intensity.
 * ---------------------------------------------------------------------------------- */
void vao_region_stats(const int32_t *labels, int h, int w, int count, int64_t *stats)
{
    memset(stats, 0, sizeof(int64_t) * 16 * (size_t)count);
    for (int l = 0; l < count; l++) {
        stats[l * 16 + 10] = w;
        stats[l * 16 + 11] = h;
        stats[l * 16 + 12] = -1;
        stats[l * 16 + 13] = -1;
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t l = labels[(size_t)y * w + x];
            if (l <= 0 || l > count)
                continue;
            int64_t *s = stats + (size_t)(l - 1) * 16;
            int64_t X = x, Y = y;
            s[0] += 1;
            s[1] += X;
            s[2] += Y;
            s[3] += X * X;
            s[4] += X * Y;
            s[5] += Y * Y;
            s[6] += X * X * X;
            s[7] += X * X * Y;
            s[8] += X * Y * Y;
            s[9] += Y * Y * Y;
            if (X < s[10]) s[10] = X;
            if (Y < s[11]) s[11] = Y;
            if (X > s[12]) s[12] = X;
            if (Y > s[13]) s[13] = Y;
        }
}

/* Central and normalised moments from the 10 spatial moments, in the operation order of
 * OpenCV's completeMomentState (the arithmetic behind cv2.moments used at
 * video/analysis/image.py:353 and video/analysis/shapes.py:533).
 * in : m[10] = m00 m10 m01 m20 m11 m02 m30 m21 m12 m03
 * out: o[14] = mu20 mu11 mu02 mu30 mu21 mu12 mu03 nu20 nu11 nu02 nu30 nu21 nu12 nu03 */
void vao_complete_moments(const double *m, double *o)
{
    double m00 = m[0], m10 = m[1], m01 = m[2], m20 = m[3], m11 = m[4], m02 = m[5];
    double m30 = m[6], m21 = m[7], m12 = m[8], m03 = m[9];
    double cx = 0, cy = 0, inv_m00 = 0.0;
    if (fabs(m00) > 2.220446049250313e-16) {
        inv_m00 = 1. / m00;
        cx = m10 * inv_m00;
        cy = m01 * inv_m00;
    }
    double mu20 = m20 - m10 * cx;
    double mu11 = m11 - m10 * cy;
    double mu02 = m02 - m01 * cy;
    o[0] = mu20;
    o[1] = mu11;
    o[2] = mu02;
    o[3] = m30 - cx * (3 * mu20 + cx * m10);
    mu11 += mu11;
    o[4] = m21 - cx * (mu11 + cx * m01) - cy * mu20;
    o[5] = m12 - cy * (mu11 + cy * m10) - cx * mu02;
    o[6] = m03 - cy * (3 * mu02 + cy * m01);
    double inv_sqrt_m00 = sqrt(fabs(inv_m00));
    double s2 = inv_m00 * inv_m00, s3 = s2 * inv_sqrt_m00;
    o[7] = o[0] * s2;
    o[8] = o[1] * s2;
    o[9] = o[2] * s2;
    o[10] = o[3] * s3;
    o[11] = o[4] * s3;
    o[12] = o[5] * s3;
    o[13] = o[6] * s3;
}

/* ------------------------------------------------------------------------------------
 * Whole chain for one batch (the BASELINE.json cfg#2/#3 workload), single-threaded:
 * bg(mean) -> gaussian(sigma) -> threshold -> [dilate k, erode k (rect)] -> label.
 * Used by tests for end-to-end parity and by bench.py's cpu_baseline leg.
 * Any of mask_out / labels_out may be NULL. `mean` is updated in place.
 * ---------------------------------------------------------------------------------- */
int vao_chain_u8(const uint8_t *frames, int n, int h, int w, double *mean, int64_t n_seen,
                 double sigma, int thresh, int morph_ksize, int connectivity,
                 uint8_t *mask_out, int32_t *labels_out, int32_t *counts_out)
{
    size_t px = (size_t)h * w;
    uint8_t *a = (uint8_t *)malloc(px), *b = (uint8_t *)malloc(px);
    int32_t *lab = labels_out ? NULL : (int32_t *)malloc(sizeof(int32_t) * px);
    for (int f = 0; f < n; f++) {
        vao_bg_mean_u8(frames + f * px, a, mean, n_seen + f, 1, px);
        if (sigma > 0) {
            if (vao_gaussian_u8(a, b, 1, h, w, 1, sigma))
                return -1;
        } else
            memcpy(b, a, px);
        vao_threshold_u8(b, a, px, thresh, 255);
        if (morph_ksize > 1) {
            vao_morph_u8(a, b, 1, h, w, 1, 0, morph_ksize);
            vao_morph_u8(b, a, 1, h, w, 0, 0, morph_ksize);
        }
        if (mask_out)
            memcpy(mask_out + f * px, a, px);
        if (connectivity) {
            int32_t *l = labels_out ? labels_out + f * px : lab;
            int cnt = vao_label_i32(a, l, h, w, connectivity);
            if (counts_out)
                counts_out[f] = cnt;
        }
    }
    free(a);
    free(b);
    free(lab);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * A8  get_contour_from_largest_region (video/analysis/regions.py:178-197):
 *         cv2.findContours(mask.astype(np.uint8), cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)[1]
 *         contour_areas = [cv2.contourArea(cnt) for cnt in contours]; np.argmax
 * Restatement of OpenCV's Suzuki-Abe scanner (cvFindNextContour / icvFetchContour, OpenCV >= 3.2
 * semantics: the image is padded with a zero border, nothing is clipped) for 8-bit input,
 * mode RETR_EXTERNAL, method CHAIN_APPROX_SIMPLE.  PARITY UNPINNED against cv2 itself (not
 * installed); property-checked in tests/ (points lie on region borders, polygon area vs pixel
 * count) and used as the independent checker of the GPU's label-root based tracing.
 * Contours are returned in OpenCV's list order (most recently found first).
 * out_points: flat (x,y) pairs of all contours in that order; out_sizes[k] = points of contour k.
 * returns the number of contours, or -1 when a capacity is exceeded.
 * ---------------------------------------------------------------------------------- */
static const int kCodeDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int kCodeDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

int vao_find_contours_external_simple(const uint8_t *mask, int h, int w, int32_t *out_points,
                                      int64_t cap_points, int32_t *out_sizes, int cap_contours)
{
    const int step = w + 2;
    signed char *img = (signed char *)calloc((size_t)(h + 2) * step, 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            img[(size_t)(y + 1) * step + x + 1] = mask[(size_t)y * w + x] ? 1 : 0;
    int deltas[16];
    for (int k = 0; k < 8; k++)
        deltas[k] = deltas[k + 8] = kCodeDy[k] * step + kCodeDx[k];
    /* discovery order first; reversed at the end */
    int ncont = 0;
    int64_t npts = 0;
    int64_t *starts = (int64_t *)malloc(sizeof(int64_t) * (size_t)(cap_contours + 1));
    const signed char nbd = 2;
    for (int y = 1; y <= h; y++) {
        int lnbd_x = 0;
        int prev = 0;
        signed char *row = img + (size_t)y * step;
        for (int x = 1; x <= w + 1; x++) {
            int p = row[x];
            if (p == prev)
                continue;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1)
                    goto resume_scan;
                if (prev & -2)
                    lnbd_x = x - 1;
                is_hole = 1;
            }
            if (is_hole || row[lnbd_x] > 0)   /* RETR_EXTERNAL */
                goto resume_scan;
            {
                /* icvFetchContour at (x, y), CHAIN_APPROX_SIMPLE */
                if (ncont >= cap_contours) {
                    free(img);
                    free(starts);
                    return -1;
                }
                starts[ncont] = npts;
                signed char *i0 = row + x, *i1, *i3, *i4 = 0;
                int ptx = x - 1, pty = y - 1;   /* un-padded coordinates */
                int s_end = 4, s = 4, prev_s;
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + deltas[s];
                } while (*i1 == 0 && s != s_end);
                if (s == s_end) { /* single pixel */
                    *i0 = (signed char)(nbd | -128);
                    if (npts + 1 > cap_points) { free(img); free(starts); return -1; }
                    out_points[2 * npts] = ptx;
                    out_points[2 * npts + 1] = pty;
                    npts++;
                } else {
                    i3 = i0;
                    prev_s = s ^ 4;
                    for (;;) {
                        s_end = s;
                        while (s < 15) {
                            i4 = i3 + deltas[++s];
                            if (*i4 != 0)
                                break;
                        }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end)
                            *i3 = (signed char)(nbd | -128);
                        else if (*i3 == 1)
                            *i3 = nbd;
                        if (s != prev_s) {
                            if (npts + 1 > cap_points) { free(img); free(starts); return -1; }
                            out_points[2 * npts] = ptx;
                            out_points[2 * npts + 1] = pty;
                            npts++;
                            prev_s = s;
                        }
                        ptx += kCodeDx[s];
                        pty += kCodeDy[s];
                        if (i4 == i0 && i3 == i1)
                            break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                ncont++;
                lnbd_x = x;
                p = row[x];
            }
        resume_scan:
            prev = p;
            if (prev & -2)
                lnbd_x = x;
        }
    }
    starts[ncont] = npts;
    /* reverse the contour order in place (OpenCV inserts each new contour at the list head) */
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(npts ? npts : 1));
    memcpy(tmp, out_points, sizeof(int32_t) * 2 * (size_t)npts);
    int64_t o = 0;
    for (int k = ncont - 1, j = 0; k >= 0; k--, j++) {
        int64_t n = starts[k + 1] - starts[k];
        memcpy(out_points + 2 * o, tmp + 2 * starts[k], sizeof(int32_t) * 2 * (size_t)n);
        out_sizes[j] = (int32_t)n;
        o += n;
    }
    free(tmp);
    free(img);
    free(starts);
    return ncont;
}

/* cv2.contourArea(contour) (oriented = false) on integer points */
double vao_contour_area(const int32_t *pts, int n)
{
    if (n == 0)
        return 0.0;
    double a00 = 0;
    double px = pts[2 * (n - 1)], py = pts[2 * (n - 1) + 1];
    for (int i = 0; i < n; i++) {
        double x = pts[2 * i], y = pts[2 * i + 1];
        a00 += px * y - py * x;
        px = x;
        py = y;
    }
    return fabs(a00 * 0.5);
}

/* ------------------------------------------------------------------------------------
 * A9  cv2.moments(contour)  -- regionprops(contour=...), video/analysis/image.py:355, and
 *     Polygon.moments, video/analysis/shapes.py:527-533 (float32 points there).
 * Restates OpenCV's contourMoments (modules/imgproc/src/moments.cpp): Green's theorem over the
 * closed polygon, ten double accumulators updated sequentially from point lpt-1 -> 0 -> 1 ...,
 * then the constant factors 1/2, 1/6, 1/12, 1/24, 1/20, 1/60 carrying the sign of a00 so that
 * m00 >= 0 for either orientation; |a00| <= FLT_EPSILON leaves every moment 0.
 * pts: n points (x, y), int32 when is_float == 0, float32 otherwise.  out[10] = m00 m10 m01 m20
 * m11 m02 m30 m21 m12 m03 (complete with vao_complete_moments).
 * ---------------------------------------------------------------------------------- */
void vao_contour_moments(const void *pts, int n, int is_float, double *out)
{
    const int32_t *pi = (const int32_t *)pts;
    const float *pf = (const float *)pts;
    for (int k = 0; k < 10; k++)
        out[k] = 0.0;
    if (n == 0)
        return;
    double a00 = 0, a10 = 0, a01 = 0, a20 = 0, a11 = 0, a02 = 0, a30 = 0, a21 = 0, a12 = 0, a03 = 0;
    double xi_1 = is_float ? (double)pf[2 * (n - 1)] : (double)pi[2 * (n - 1)];
    double yi_1 = is_float ? (double)pf[2 * (n - 1) + 1] : (double)pi[2 * (n - 1) + 1];
    double xi_12 = xi_1 * xi_1, yi_12 = yi_1 * yi_1;
    for (int i = 0; i < n; i++) {
        double xi = is_float ? (double)pf[2 * i] : (double)pi[2 * i];
        double yi = is_float ? (double)pf[2 * i + 1] : (double)pi[2 * i + 1];
        double xi2 = xi * xi, yi2 = yi * yi;
        double dxy = xi_1 * yi - xi * yi_1;
        double xii_1 = xi_1 + xi, yii_1 = yi_1 + yi;
        a00 += dxy;
        a10 += dxy * xii_1;
        a01 += dxy * yii_1;
        a20 += dxy * (xi_1 * xii_1 + xi2);
        a11 += dxy * (xi_1 * (yii_1 + yi_1) + xi * (yii_1 + yi));
        a02 += dxy * (yi_1 * yii_1 + yi2);
        a30 += dxy * xii_1 * (xi_12 + xi2);
        a03 += dxy * yii_1 * (yi_12 + yi2);
        a21 += dxy * (xi_12 * (3 * yi_1 + yi) + 2 * xi * xi_1 * yii_1 + xi2 * (yi_1 + 3 * yi));
        a12 += dxy * (yi_12 * (3 * xi_1 + xi) + 2 * yi * yi_1 * xii_1 + yi2 * (xi_1 + 3 * xi));
        xi_1 = xi;
        yi_1 = yi;
        xi_12 = xi2;
        yi_12 = yi2;
    }
    if (fabs(a00) > 1.1920928955078125e-07 /* FLT_EPSILON */) {
        double s = a00 > 0 ? 1.0 : -1.0;
        double db1_2 = s * 0.5, db1_6 = s * 0.16666666666666666666666666666667;
        double db1_12 = s * 0.083333333333333333333333333333333;
        double db1_24 = s * 0.041666666666666666666666666666667, db1_20 = s * 0.05;
        double db1_60 = s * 0.016666666666666666666666666666667;
        out[0] = a00 * db1_2;
        out[1] = a10 * db1_6;
        out[2] = a01 * db1_6;
        out[3] = a20 * db1_12;
        out[4] = a11 * db1_24;
        out[5] = a02 * db1_12;
        out[6] = a30 * db1_20;
        out[7] = a21 * db1_60;
        out[8] = a12 * db1_60;
        out[9] = a03 * db1_20;
    }
}

/* ------------------------------------------------------------------------------------
 * N4  cv2.resize(frame, size, interpolation=...)  -- FilterResize._process_frame,
 *     video/filters.py:310-314 (modes chosen at :274-293; 'auto' = INTER_AREA when the frame
 *     shrinks, INTER_CUBIC when it grows).  uint8 frames, c interleaved channels.
 * Restates OpenCV's resize (modules/imgproc/src/resize.cpp) as published for 8-bit images:
 *   NEAREST  sx = min(floor(dx * (1/inv_scale_x)), sw-1)
 *   LINEAR   fx = (dx+0.5)*scale_x - 0.5 in float; 11-bit fixed-point coefficients
 *            (saturate_cast<short>(c * 2048)); horizontal sums in int; vertical
 *            ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; an exact 2x2 shrink is AREA
 *   CUBIC    A = -0.75 coefficients in float, 11-bit fixed point, replicated border,
 *            (sum + 2^21) >> 22
 *   AREA     shrink by integer factors: block mean ((sum+2)>>2 for 2x2, else
 *            saturate_cast<uchar>(sum * (1.f/area))); other shrinks: float cell-overlap tables
 *            (computeResizeAreaTab order); growing: LINEAR with area-style positions
 *   LANCZOS4 8 x 8 taps (interpolateLanczos4: sines by angle addition, normalised in float), 11-bit
 *            fixed point like CUBIC, int32 accumulation, (sum + 2^21) >> 22, replicated border
 * mode: 0 nearest, 1 linear, 2 cubic, 3 area, 4 lanczos4.  Parity with real OpenCV: unpinned (no cv2).
 * vao_resize_f32: the float32 instantiations of the same templates (float coefficients, products
 * summed left to right in float without contraction, no rounding step at the end).
 * ---------------------------------------------------------------------------------- */
static int vao_round_half_even(double v) { return (int)nearbyint(v); }
static short vao_sat_short(float v)
{
    int r = (int)nearbyintf(v);
    return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r);
}
static uint8_t vao_sat_u8i(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static void vao_cubic_coeffs(float x, float *c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}
static void vao_lanczos4_coeffs(float x, float *coeffs)
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-07f) {          /* FLT_EPSILON */
        for (int i = 0; i < 8; i++)
            coeffs[i] = 0;
        coeffs[3] = 1;
        return;
    }
    float sum = 0;
    const double pi = 3.1415926535897932384626433832795;
    double y0 = -(double)(x + 3) * pi * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        double y = -(double)(x + 3 - i) * pi * 0.25;
        coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += coeffs[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++)
        coeffs[i] *= sum;
}
typedef struct { int si, di; float alpha; } vao_dec_alpha;
static int vao_area_tab(int ssize, int dsize, double scale, vao_dec_alpha *tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx, tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cell);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx, tab[k].si = sx;
            tab[k++].alpha = (float)(1.0 / cell);
        }
        if (fsx2 - sx2 > 1e-3) {
            double m = fsx2 - sx2 < 1. ? fsx2 - sx2 : 1.;
            m = m < cell ? m : cell;
            tab[k].di = dx, tab[k].si = sx2;
            tab[k++].alpha = (float)(m / cell);
        }
    }
    return k;
}

int vao_resize_u8(const uint8_t *src, uint8_t *dst, int n, int sh, int sw, int c, int dh, int dw,
                  int mode)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || c <= 0 || mode < 0 || mode > 4)
        return -1;
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
    const size_t sfs = (size_t)sh * sw * c, dfs = (size_t)dh * dw * c;
    if (mode == 0) {
        const double ifx = 1. / inv_sx, ify = 1. / inv_sy;
        for (int f = 0; f < n; f++)
            for (int y = 0; y < dh; y++) {
                int sy = (int)floor(y * ify);
                if (sy > sh - 1) sy = sh - 1;
                for (int x = 0; x < dw; x++) {
                    int sx = (int)floor(x * ifx);
                    if (sx > sw - 1) sx = sw - 1;
                    for (int k = 0; k < c; k++)
                        dst[f * dfs + ((size_t)y * dw + x) * c + k] = src[f * sfs + ((size_t)sy * sw + sx) * c + k];
                }
            }
        return 0;
    }
    int iscale_x = (int)nearbyint(scale_x), iscale_y = (int)nearbyint(scale_y);   /* saturate_cast<int> */
    const int area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (mode == 1 && area_fast && iscale_x == 2 && iscale_y == 2)
        mode = 3;
    if (mode == 3 && scale_x >= 1 && scale_y >= 1) {
        if (area_fast) {
            const int area = iscale_x * iscale_y;
            const float scale = 1.f / area;
            int wfull = (int)(sw / scale_x);
            if (wfull > dw) wfull = dw;
            for (int f = 0; f < n; f++)
                for (int y = 0; y < dh; y++) {
                    const int sy0 = y * iscale_y;
                    for (int x = 0; x < dw; x++)
                        for (int k = 0; k < c; k++) {
                            const int sx0 = x * iscale_x;
                            uint8_t out;
                            if (sy0 + iscale_y <= sh && x < wfull) {
                                int sum = 0;
                                for (int yy = 0; yy < iscale_y; yy++)
                                    for (int xx = 0; xx < iscale_x; xx++)
                                        sum += src[f * sfs + ((size_t)(sy0 + yy) * sw + sx0 + xx) * c + k];
                                if (iscale_x == 2 && iscale_y == 2)
                                    out = (uint8_t)((sum + 2) >> 2);
                                else
                                    out = vao_sat_u8i((int)nearbyintf(sum * scale));
                            } else {
                                int sum = 0, count = 0;
                                for (int yy = 0; yy < iscale_y && sy0 + yy < sh; yy++)
                                    for (int xx = 0; xx < iscale_x && sx0 + xx < sw; xx++) {
                                        sum += src[f * sfs + ((size_t)(sy0 + yy) * sw + sx0 + xx) * c + k];
                                        count++;
                                    }
                                out = vao_sat_u8i((int)nearbyintf((float)sum / count));
                            }
                            dst[f * dfs + ((size_t)y * dw + x) * c + k] = out;
                        }
                }
            return 0;
        }
        vao_dec_alpha *xt = (vao_dec_alpha *)malloc(sizeof(vao_dec_alpha) * (size_t)(sw * 2 + dw * 2 + 4));
        vao_dec_alpha *yt = (vao_dec_alpha *)malloc(sizeof(vao_dec_alpha) * (size_t)(sh * 2 + dh * 2 + 4));
        const int nx = vao_area_tab(sw, dw, scale_x, xt), ny = vao_area_tab(sh, dh, scale_y, yt);
        float *buf = (float *)malloc(sizeof(float) * (size_t)dw * c), *sum = (float *)malloc(sizeof(float) * (size_t)dw * c);
        for (int f = 0; f < n; f++) {
            int prev_dy = yt[0].di;
            for (int i = 0; i < dw * c; i++) sum[i] = 0.f;
            for (int j = 0; j < ny; j++) {
                const float beta = yt[j].alpha;
                const int dy = yt[j].di, sy = yt[j].si;
                const uint8_t *S = src + f * sfs + (size_t)sy * sw * c;
                for (int i = 0; i < dw * c; i++) buf[i] = 0.f;
                for (int k = 0; k < nx; k++)
                    for (int ch = 0; ch < c; ch++) {
                        const float p = S[xt[k].si * c + ch] * xt[k].alpha;
                        buf[xt[k].di * c + ch] += p;
                    }
                if (dy != prev_dy) {
                    uint8_t *D = dst + f * dfs + (size_t)prev_dy * dw * c;
                    for (int i = 0; i < dw * c; i++) {
                        D[i] = vao_sat_u8i((int)nearbyintf(sum[i]));
                        sum[i] = beta * buf[i];
                    }
                    prev_dy = dy;
                } else {
                    for (int i = 0; i < dw * c; i++) {
                        const float p = beta * buf[i];
                        sum[i] += p;
                    }
                }
            }
            uint8_t *D = dst + f * dfs + (size_t)prev_dy * dw * c;
            for (int i = 0; i < dw * c; i++)
                D[i] = vao_sat_u8i((int)nearbyintf(sum[i]));
        }
        free(xt); free(yt); free(buf); free(sum);
        return 0;
    }
    /* LINEAR / CUBIC (and AREA when growing = LINEAR with area-style positions) */
    const int area_mode = mode == 3;
    if (area_mode) mode = 1;
    const int ksize = mode == 1 ? 2 : mode == 2 ? 4 : 8, ksize2 = ksize / 2;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw), *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
    short *ialpha = (short *)malloc(sizeof(short) * (size_t)dw * ksize), *ibeta = (short *)malloc(sizeof(short) * (size_t)dh * ksize);
    int xmin = 0, xmax = dw;
    float cbuf[8];
    for (int dx = 0; dx < dw; dx++) {
        float fx;
        int sx;
        if (!area_mode) {
            fx = (float)((dx + 0.5) * scale_x - 0.5);
            sx = (int)floorf(fx);
            fx -= sx;
        } else {
            sx = (int)floor(dx * scale_x);
            fx = (float)((dx + 1) - (sx + 1) * inv_sx);
            fx = fx <= 0 ? 0.f : fx - floorf(fx);
        }
        if (sx < ksize2 - 1) {
            xmin = dx + 1;
            if (sx < 0 && mode == 1) fx = 0, sx = 0;
        }
        if (sx + ksize2 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1 && mode == 1) fx = 0, sx = sw - 1;
        }
        xofs[dx] = sx;
        if (mode == 1) cbuf[0] = 1.f - fx, cbuf[1] = fx;
        else if (mode == 2) vao_cubic_coeffs(fx, cbuf);
        else vao_lanczos4_coeffs(fx, cbuf);
        for (int k = 0; k < ksize; k++)
            ialpha[dx * ksize + k] = vao_sat_short(cbuf[k] * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy;
        int sy;
        if (!area_mode) {
            fy = (float)((dy + 0.5) * scale_y - 0.5);
            sy = (int)floorf(fy);
            fy -= sy;
        } else {
            sy = (int)floor(dy * scale_y);
            fy = (float)((dy + 1) - (sy + 1) * inv_sy);
            fy = fy <= 0 ? 0.f : fy - floorf(fy);
        }
        yofs[dy] = sy;
        if (mode == 1) cbuf[0] = 1.f - fy, cbuf[1] = fy;
        else if (mode == 2) vao_cubic_coeffs(fy, cbuf);
        else vao_lanczos4_coeffs(fy, cbuf);
        for (int k = 0; k < ksize; k++)
            ibeta[dy * ksize + k] = vao_sat_short(cbuf[k] * 2048);
    }
    (void)xmin;
    for (int f = 0; f < n; f++)
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int ch = 0; ch < c; ch++) {
                    int rows[8];
                    for (int k = 0; k < ksize; k++) {
                        int sy = yofs[dy] - ksize2 + 1 + k;
                        sy = sy < 0 ? 0 : sy > sh - 1 ? sh - 1 : sy;
                        const uint8_t *S = src + f * sfs + (size_t)sy * sw * c;
                        int v = 0;
                        if (mode == 1) {
                            if (dx < xmax)
                                v = S[xofs[dx] * c + ch] * ialpha[dx * 2] + S[(xofs[dx] + 1) * c + ch] * ialpha[dx * 2 + 1];
                            else
                                v = S[xofs[dx] * c + ch] * 2048;
                        } else {
                            for (int j = 0; j < ksize; j++) {
                                int sxj = xofs[dx] - ksize2 + 1 + j;
                                sxj = sxj < 0 ? 0 : sxj > sw - 1 ? sw - 1 : sxj;
                                v += S[sxj * c + ch] * ialpha[dx * ksize + j];
                            }
                        }
                        rows[k] = v;
                    }
                    int out;
                    if (mode == 1) {
                        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
                        out = (((b0 * (rows[0] >> 4)) >> 16) + ((b1 * (rows[1] >> 4)) >> 16) + 2) >> 2;
                    } else if (mode == 2) {
                        long long s = 0;
                        for (int k = 0; k < 4; k++)
                            s += (long long)ibeta[dy * 4 + k] * rows[k];
                        out = (int)((s + (1 << 21)) >> 22);
                    } else {                    /* LANCZOS4: OpenCV accumulates in int (wraps like its int does) */
                        uint32_t s = 0;
                        for (int k = 0; k < 8; k++)
                            s += (uint32_t)((int)ibeta[dy * 8 + k] * rows[k]);
                        out = (int)(s + (1u << 21)) >> 22;
                    }
                    dst[f * dfs + ((size_t)dy * dw + dx) * c + ch] = vao_sat_u8i(out);
                }
    free(xofs); free(yofs); free(ialpha); free(ibeta);
    (void)vao_round_half_even;
    return 0;
}

/* cv2.resize on float32 frames (FilterResize takes whatever dtype the video has, video/filters.py:310-314):
 * the <float, float, float> instantiations of OpenCV's resize templates -- float coefficient tables,
 * `S[sx]*a0 + S[sx+cn]*a1` / sums of products from the first tap to the last in float, no FMA
 * contraction, results stored as they are. */
int vao_resize_f32(const float *src, float *dst, int n, int sh, int sw, int c, int dh, int dw, int mode)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || c <= 0 || mode < 0 || mode > 4)
        return -1;
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
    const size_t sfs = (size_t)sh * sw * c, dfs = (size_t)dh * dw * c;
    if (mode == 0) {
        const double ifx = 1. / inv_sx, ify = 1. / inv_sy;
        for (int f = 0; f < n; f++)
            for (int y = 0; y < dh; y++) {
                int sy = (int)floor(y * ify);
                if (sy > sh - 1) sy = sh - 1;
                for (int x = 0; x < dw; x++) {
                    int sx = (int)floor(x * ifx);
                    if (sx > sw - 1) sx = sw - 1;
                    for (int k = 0; k < c; k++)
                        dst[f * dfs + ((size_t)y * dw + x) * c + k] = src[f * sfs + ((size_t)sy * sw + sx) * c + k];
                }
            }
        return 0;
    }
    int iscale_x = (int)nearbyint(scale_x), iscale_y = (int)nearbyint(scale_y);
    const int area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (mode == 1 && area_fast && iscale_x == 2 && iscale_y == 2)
        mode = 3;
    if (mode == 3 && scale_x >= 1 && scale_y >= 1) {
        if (area_fast) {
            const float scale = 1.f / (iscale_x * iscale_y);
            int wfull = (int)(sw / scale_x);
            if (wfull > dw) wfull = dw;
            for (int f = 0; f < n; f++)
                for (int y = 0; y < dh; y++) {
                    const int sy0 = y * iscale_y;
                    for (int x = 0; x < dw; x++)
                        for (int k = 0; k < c; k++) {
                            const int sx0 = x * iscale_x;
                            float sum = 0.f;
                            int count = 0;
                            for (int yy = 0; yy < iscale_y && sy0 + yy < sh; yy++)
                                for (int xx = 0; xx < iscale_x && sx0 + xx < sw; xx++) {
                                    sum += src[f * sfs + ((size_t)(sy0 + yy) * sw + sx0 + xx) * c + k];
                                    count++;
                                }
                            dst[f * dfs + ((size_t)y * dw + x) * c + k] =
                                (sy0 + iscale_y <= sh && x < wfull) ? sum * scale : sum / count;
                        }
                }
            return 0;
        }
        vao_dec_alpha *xt = (vao_dec_alpha *)malloc(sizeof(vao_dec_alpha) * (size_t)(sw * 2 + dw * 2 + 4));
        vao_dec_alpha *yt = (vao_dec_alpha *)malloc(sizeof(vao_dec_alpha) * (size_t)(sh * 2 + dh * 2 + 4));
        const int nx = vao_area_tab(sw, dw, scale_x, xt), ny = vao_area_tab(sh, dh, scale_y, yt);
        float *buf = (float *)malloc(sizeof(float) * (size_t)dw * c), *sum = (float *)malloc(sizeof(float) * (size_t)dw * c);
        for (int f = 0; f < n; f++) {
            int prev_dy = yt[0].di;
            for (int i = 0; i < dw * c; i++) sum[i] = 0.f;
            for (int j = 0; j < ny; j++) {
                const float beta = yt[j].alpha;
                const int dy = yt[j].di, sy = yt[j].si;
                const float *S = src + f * sfs + (size_t)sy * sw * c;
                for (int i = 0; i < dw * c; i++) buf[i] = 0.f;
                for (int k = 0; k < nx; k++)
                    for (int ch = 0; ch < c; ch++) {
                        const float p = S[xt[k].si * c + ch] * xt[k].alpha;
                        buf[xt[k].di * c + ch] += p;
                    }
                if (dy != prev_dy) {
                    float *D = dst + f * dfs + (size_t)prev_dy * dw * c;
                    for (int i = 0; i < dw * c; i++) {
                        D[i] = sum[i];
                        sum[i] = beta * buf[i];
                    }
                    prev_dy = dy;
                } else {
                    for (int i = 0; i < dw * c; i++) {
                        const float p = beta * buf[i];
                        sum[i] += p;
                    }
                }
            }
            float *D = dst + f * dfs + (size_t)prev_dy * dw * c;
            for (int i = 0; i < dw * c; i++)
                D[i] = sum[i];
        }
        free(xt); free(yt); free(buf); free(sum);
        return 0;
    }
    const int area_mode = mode == 3;
    if (area_mode) mode = 1;
    const int ksize = mode == 1 ? 2 : mode == 2 ? 4 : 8, ksize2 = ksize / 2;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw), *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
    float *alpha = (float *)malloc(sizeof(float) * (size_t)dw * ksize), *beta = (float *)malloc(sizeof(float) * (size_t)dh * ksize);
    int xmax = dw;
    for (int axis = 0; axis < 2; axis++) {
        const int dsize = axis ? dh : dw, ssize = axis ? sh : sw;
        const double scale = axis ? scale_y : scale_x, inv = axis ? inv_sy : inv_sx;
        for (int d = 0; d < dsize; d++) {
            float fx;
            int sx;
            if (!area_mode) {
                fx = (float)((d + 0.5) * scale - 0.5);
                sx = (int)floorf(fx);
                fx -= sx;
            } else {
                sx = (int)floor(d * scale);
                fx = (float)((d + 1) - (sx + 1) * inv);
                fx = fx <= 0 ? 0.f : fx - floorf(fx);
            }
            if (!axis) {
                if (sx < ksize2 - 1 && sx < 0 && mode == 1) fx = 0, sx = 0;
                if (sx + ksize2 >= ssize) {
                    if (d < xmax) xmax = d;
                    if (sx >= ssize - 1 && mode == 1) fx = 0, sx = ssize - 1;
                }
            }
            (axis ? yofs : xofs)[d] = sx;
            float *cf = (axis ? beta : alpha) + (size_t)d * ksize;
            if (mode == 1) cf[0] = 1.f - fx, cf[1] = fx;
            else if (mode == 2) vao_cubic_coeffs(fx, cf);
            else vao_lanczos4_coeffs(fx, cf);
        }
    }
    for (int f = 0; f < n; f++)
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int ch = 0; ch < c; ch++) {
                    float rows[8];
                    for (int k = 0; k < ksize; k++) {
                        int sy = yofs[dy] - ksize2 + 1 + k;
                        sy = sy < 0 ? 0 : sy > sh - 1 ? sh - 1 : sy;
                        const float *S = src + f * sfs + (size_t)sy * sw * c;
                        float v = 0.f;
                        if (mode == 1) {
                            if (dx < xmax) {
                                const float p0 = S[xofs[dx] * c + ch] * alpha[dx * 2], p1 = S[(xofs[dx] + 1) * c + ch] * alpha[dx * 2 + 1];
                                v = p0 + p1;
                            } else {
                                v = S[xofs[dx] * c + ch] * 1.f;
                            }
                        } else {
                            for (int j = 0; j < ksize; j++) {
                                int sxj = xofs[dx] - ksize2 + 1 + j;
                                sxj = sxj < 0 ? 0 : sxj > sw - 1 ? sw - 1 : sxj;
                                const float p = S[sxj * c + ch] * alpha[dx * ksize + j];
                                v += p;
                            }
                        }
                        rows[k] = v;
                    }
                    float out = rows[0] * beta[dy * ksize];
                    for (int k = 1; k < ksize; k++) {
                        const float p = rows[k] * beta[dy * ksize + k];
                        out += p;
                    }
                    dst[f * dfs + ((size_t)dy * dw + dx) * c + ch] = out;
                }
    free(xofs); free(yofs); free(alpha); free(beta);
    return 0;
}
