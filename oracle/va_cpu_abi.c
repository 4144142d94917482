/*
 * va_cpu_abi.c -- libvideoanalysis_cpu.so: the CPU restatement behind the PRODUCT's C ABI.
 *
 * TEST INFRASTRUCTURE ("same signatures exported by the CPU restatement library
 * so tests swap backends").  Every function here has the signature of the same-named entry point
 * of include/videoanalysis_hip.h and forwards to the oracle (va_oracle.c); "device" pointers are
 * host pointers, streams are ignored, everything is synchronous.  Only tests load it -- the
 * product never does, and has no CPU fallback.  Entry points without a CPU twin (the fused
 * pipeline handle, streams/events, RCCL, test hooks) are simply absent: a test asking for them
 * on this backend fails at symbol lookup.
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/videoanalysis_hip.h"

/* oracle entry points (va_oracle.c, linked into the same library) */
int vao_gauss_taps_q8(double sigma, int *ksize_out, uint16_t *taps);
int vao_gauss_taps_q8_rule(double sigma, int rule, int *ksize_out, uint16_t *taps);
int vao_gaussian_u8_rule(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c, double sigma, int rule);
int vao_gauss_taps_f32(double sigma, int *ksize_out, float *taps);
int vao_gaussian_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c, double sigma);
int vao_gaussian_f32(const float *src, float *dst, int n, int h, int w, int c, double sigma);
void vao_bg_mean_u8(const uint8_t *frames, uint8_t *diff_out, double *mean, int64_t n_seen, int n, size_t px);
void vao_welford_u8(const uint8_t *frames, double *mean, double *m2, int64_t n_seen, int n, size_t px);
void vao_bg_ema_f32(const float *frames, float *diff_out, float *bg, int64_t n_seen, float rate, int n, size_t px);
void vao_bg_ema_u8(const uint8_t *frames, uint8_t *diff_out, float *bg, int64_t n_seen, float rate, int n, size_t px);
void vao_bg_static_u8(const uint8_t *frames, uint8_t *diff_out, const double *bg, int n, size_t px);
void vao_time_difference_u8(const uint8_t *this_frame, const uint8_t *prev_frame, int16_t *out, size_t count);
void vao_threshold_u8(const uint8_t *src, uint8_t *dst, size_t count, int thresh, int maxval);
void vao_mono_mean_u8(const uint8_t *src, uint8_t *dst, size_t pixels);
int vao_morph_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int op, int shape, int ksize);
void vao_label_batch_i32(const uint8_t *mask, int32_t *labels, int32_t *counts, int n, int h, int w, int connectivity);
void vao_region_stats(const int32_t *labels, int h, int w, int count, int64_t *stats);
void vao_contour_moments(const void *pts, int n, int is_float, double *out);
int vao_resize_u8(const uint8_t *src, uint8_t *dst, int n, int sh, int sw, int c, int dh, int dw, int mode);
int vao_resize_f32(const float *src, float *dst, int n, int sh, int sw, int c, int dh, int dw, int mode);

static _Thread_local char g_err[256] = "";
#define FAIL(code, ...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return (code); } while (0)

int va_init(int device) { if (device != 0) FAIL(VA_ERR_INVALID, "va_init: the CPU twin has one device"); return VA_OK; }
int va_device_count(void) { return 1; }
const char *va_version(void) { return "videoanalysis_cpu 0.1 (oracle behind the product ABI; tests only)"; }
const char *va_last_error(void) { return g_err; }
int va_malloc(void **p, size_t bytes) { if (!p) FAIL(VA_ERR_INVALID, "va_malloc: NULL"); *p = bytes ? malloc(bytes) : NULL; if (bytes && !*p) FAIL(VA_ERR_NOMEM, "va_malloc: out of memory"); return VA_OK; }
int va_free(void *p) { free(p); return VA_OK; }
int va_host_alloc(void **p, size_t bytes) { return va_malloc(p, bytes); }
int va_host_free(void *p) { free(p); return VA_OK; }
int va_memcpy_h2d(void *d, const void *s, size_t n, void *st) { (void)st; if (n) memcpy(d, s, n); return VA_OK; }
int va_memcpy_d2h(void *d, const void *s, size_t n, void *st) { (void)st; if (n) memcpy(d, s, n); return VA_OK; }
int va_memcpy_d2d(void *d, const void *s, size_t n, void *st) { (void)st; if (n) memmove(d, s, n); return VA_OK; }
int va_memset(void *d, int v, size_t n, void *st) { (void)st; if (n) memset(d, v, n); return VA_OK; }
int va_stream_sync(void *st) { (void)st; return VA_OK; }

int va_gauss_taps_q8(double sigma, int *ks, uint16_t *taps, int cap)
{
    uint16_t t[512];
    int k;
    if (!ks || !taps || vao_gauss_taps_q8(sigma, &k, t)) FAIL(VA_ERR_INVALID, "va_gauss_taps_q8: bad argument");
    if (k > cap) FAIL(VA_ERR_RANGE, "va_gauss_taps_q8: %d taps > capacity %d", k, cap);
    memcpy(taps, t, sizeof(uint16_t) * (size_t)k);
    *ks = k;
    return VA_OK;
}
int va_gauss_taps_q8_rule(double sigma, int rule, int *ks, uint16_t *taps, int cap)
{
    uint16_t t[512];
    int k;
    if (!ks || !taps || vao_gauss_taps_q8_rule(sigma, rule, &k, t)) FAIL(VA_ERR_INVALID, "va_gauss_taps_q8_rule: bad argument");
    if (k > cap) FAIL(VA_ERR_RANGE, "va_gauss_taps_q8_rule: %d taps > capacity %d", k, cap);
    memcpy(taps, t, sizeof(uint16_t) * (size_t)k);
    *ks = k;
    return VA_OK;
}
int va_gaussian_u8_rule(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c, double sigma, int rule, void *st)
{
    (void)st;
    if (!src || !dst || src == dst) FAIL(VA_ERR_INVALID, "va_gaussian_u8_rule: src/dst must be distinct non-NULL");
    if (n < 0 || h <= 0 || w <= 0 || c <= 0) FAIL(VA_ERR_INVALID, "va_gaussian_u8_rule: bad shape");
    if (vao_gaussian_u8_rule(src, dst, n, h, w, c, sigma, rule)) FAIL(VA_ERR_INVALID, "va_gaussian_u8_rule: bad sigma / rule");
    return VA_OK;
}
int va_gauss_taps_f32(double sigma, int *ks, float *taps, int cap)
{
    float t[512];
    int k;
    if (!ks || !taps || vao_gauss_taps_f32(sigma, &k, t)) FAIL(VA_ERR_INVALID, "va_gauss_taps_f32: bad argument");
    if (k > cap) FAIL(VA_ERR_RANGE, "va_gauss_taps_f32: %d taps > capacity %d", k, cap);
    memcpy(taps, t, sizeof(float) * (size_t)k);
    *ks = k;
    return VA_OK;
}
int va_gaussian_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c, double sigma, void *st)
{
    (void)st;
    if (!src || !dst || src == dst) FAIL(VA_ERR_INVALID, "va_gaussian_u8: src/dst must be distinct non-NULL");
    if (n < 0 || h <= 0 || w <= 0 || c <= 0) FAIL(VA_ERR_INVALID, "va_gaussian_u8: bad shape");
    if (vao_gaussian_u8(src, dst, n, h, w, c, sigma)) FAIL(VA_ERR_INVALID, "va_gaussian_u8: bad sigma");
    return VA_OK;
}
int va_gaussian_f32(const float *src, float *dst, int n, int h, int w, int c, double sigma, void *st)
{
    (void)st;
    if (!src || !dst || src == dst) FAIL(VA_ERR_INVALID, "va_gaussian_f32: src/dst must be distinct non-NULL");
    if (n < 0 || h <= 0 || w <= 0 || c <= 0) FAIL(VA_ERR_INVALID, "va_gaussian_f32: bad shape");
    if (vao_gaussian_f32(src, dst, n, h, w, c, sigma)) FAIL(VA_ERR_INVALID, "va_gaussian_f32: bad sigma");
    return VA_OK;
}
int va_bg_update(int mode, int dtype, const void *frames, void *diff, void *state, int64_t n_seen, double rate,
                 int n, size_t px, void *st)
{
    (void)st;
    if (!frames || !state) FAIL(VA_ERR_INVALID, "va_bg_update: frames/state must not be NULL");
    if (dtype == VA_U8 && mode == VA_BG_MEAN) vao_bg_mean_u8(frames, diff, state, n_seen, n, px);
    else if (dtype == VA_U8 && mode == VA_BG_EMA) vao_bg_ema_u8(frames, diff, state, n_seen, (float)rate, n, px);
    else if (dtype == VA_U8 && mode == VA_BG_STATIC) { if (!diff) FAIL(VA_ERR_INVALID, "va_bg_update: static mode needs diff_out"); vao_bg_static_u8(frames, diff, state, n, px); }
    else if (dtype == VA_F32 && mode == VA_BG_EMA) vao_bg_ema_f32(frames, diff, state, n_seen, (float)rate, n, px);
    else FAIL(VA_ERR_INVALID, "va_bg_update: unsupported mode %d / dtype %d", mode, dtype);
    return VA_OK;
}
int va_welford_u8(const uint8_t *frames, double *mean, double *m2, int64_t n_seen, int n, size_t px, void *st)
{
    (void)st;
    if (!frames || !mean || !m2) FAIL(VA_ERR_INVALID, "va_welford_u8: NULL argument");
    vao_welford_u8(frames, mean, m2, n_seen, n, px);
    return VA_OK;
}
int va_time_difference_u8(const uint8_t *a, const uint8_t *b, int16_t *out, size_t count, void *st)
{
    (void)st;
    if (!a || !b || !out) FAIL(VA_ERR_INVALID, "va_time_difference_u8: NULL argument");
    vao_time_difference_u8(a, b, out, count);
    return VA_OK;
}
int va_threshold_u8(const uint8_t *src, uint8_t *dst, size_t count, int thresh, int maxval, void *st)
{
    (void)st;
    if (!src || !dst) FAIL(VA_ERR_INVALID, "va_threshold_u8: NULL argument");
    vao_threshold_u8(src, dst, count, thresh, maxval);
    return VA_OK;
}
int va_mono_mean_u8(const uint8_t *src, uint8_t *dst, size_t pixels, void *st)
{
    (void)st;
    if (!src || !dst) FAIL(VA_ERR_INVALID, "va_mono_mean_u8: NULL argument");
    vao_mono_mean_u8(src, dst, pixels);
    return VA_OK;
}
int va_normalize_u8(const uint8_t *src, uint8_t *dst, size_t count, double fmin, double fmax, double alpha,
                    double tmin, void *st)
{
    (void)st;
    if (!src || !dst) FAIL(VA_ERR_INVALID, "va_normalize_u8: NULL argument");
    for (size_t i = 0; i < count; i++) {            /* video/filters.py:126-132 */
        double f = src[i];
        f = f < fmin ? fmin : (f > fmax ? fmax : f);
        dst[i] = (uint8_t)(int)((f - fmin) * alpha + tmin);
    }
    return VA_OK;
}
int va_morph_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int op, int shape, int ksize, void *st)
{
    (void)st;
    if (!src || !dst || src == dst) FAIL(VA_ERR_INVALID, "va_morph_u8: src/dst must be distinct non-NULL");
    if (n < 0 || h <= 0 || w <= 0) FAIL(VA_ERR_INVALID, "va_morph_u8: bad shape");
    if (op != VA_MORPH_ERODE && op != VA_MORPH_DILATE) FAIL(VA_ERR_INVALID, "va_morph_u8: bad op %d", op);
    if (vao_morph_u8(src, dst, n, h, w, op, shape, ksize)) FAIL(VA_ERR_INVALID, "va_morph_u8: bad structuring element");
    return VA_OK;
}
size_t va_label_workspace_bytes(int n, int h, int w) { (void)n; (void)h; (void)w; return 256; }
int va_label_i32(const uint8_t *mask, int32_t *labels, int32_t *counts, int n, int h, int w, int conn, void *ws,
                 size_t ws_bytes, void *st)
{
    (void)st; (void)ws_bytes;
    if (!mask || !labels || !counts || !ws) FAIL(VA_ERR_INVALID, "va_label_i32: NULL argument");
    if (n < 0 || h <= 0 || w <= 0) FAIL(VA_ERR_INVALID, "va_label_i32: bad shape");
    if (conn != 4 && conn != 8) FAIL(VA_ERR_INVALID, "label: connectivity must be 4 or 8 (got %d)", conn);
    vao_label_batch_i32(mask, labels, counts, n, h, w, conn);
    return VA_OK;
}
int va_moments_i64(const int32_t *labels, int n, int h, int w, int max_labels, int64_t *stats, void *st)
{
    (void)st;
    if (!labels || !stats || max_labels <= 0) FAIL(VA_ERR_INVALID, "moments: NULL argument / max_labels <= 0");
    for (int f = 0; f < n; f++) {
        int64_t *s = stats + (size_t)f * max_labels * VA_STATS_STRIDE;
        const int32_t *L = labels + (size_t)f * h * w;
        int count = 0;
        for (size_t i = 0; i < (size_t)h * w; i++) count = L[i] > count ? L[i] : count;
        int64_t *tmp = (int64_t *)calloc((size_t)(count > 0 ? count : 1) * VA_STATS_STRIDE, sizeof(int64_t));
        vao_region_stats(L, h, w, count, tmp);
        for (int l = 0; l < max_labels; l++)
            for (int k = 0; k < VA_STATS_STRIDE; k++)
                s[(size_t)l * VA_STATS_STRIDE + k] = l < count ? tmp[(size_t)l * VA_STATS_STRIDE + k]
                                                               : (k == 10 ? w : k == 11 ? h : k == 12 || k == 13 ? -1 : 0);
        free(tmp);
    }
    return VA_OK;
}
int va_resize_u8(const uint8_t *src, uint8_t *dst, int n, int sh, int sw, int c, int dh, int dw, int mode, void *st)
{
    (void)st;
    if (!src || !dst || src == dst) FAIL(VA_ERR_INVALID, "va_resize_u8: src/dst must be distinct non-NULL");
    if (vao_resize_u8(src, dst, n, sh, sw, c, dh, dw, mode)) FAIL(VA_ERR_INVALID, "va_resize_u8: bad argument");
    return VA_OK;
}
int va_resize_f32(const float *src, float *dst, int n, int sh, int sw, int c, int dh, int dw, int mode, void *st)
{
    (void)st;
    if (!src || !dst || src == dst) FAIL(VA_ERR_INVALID, "va_resize_f32: src/dst must be distinct non-NULL");
    if (vao_resize_f32(src, dst, n, sh, sw, c, dh, dw, mode)) FAIL(VA_ERR_INVALID, "va_resize_f32: bad argument");
    return VA_OK;
}
int va_contour_moments(const void *points, const int32_t *npoints, int n, int max_points, int is_float,
                       double *out, void *st)
{
    (void)st;
    if (!points || !out) FAIL(VA_ERR_INVALID, "va_contour_moments: NULL argument");
    for (int f = 0; f < n; f++) {
        int k = npoints ? npoints[f] : max_points;
        k = k < 0 ? 0 : (k > max_points ? max_points : k);
        vao_contour_moments((const char *)points + (size_t)f * max_points * 8, k, is_float, out + (size_t)f * 10);
    }
    return VA_OK;
}
