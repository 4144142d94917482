// va_common.h -- shared helpers for the HIP sources of libvideoanalysis_hip.so (gfx950 only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/videoanalysis_hip.h"

namespace va {

// thread-local last-error message (va_last_error)
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *get_error();

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define VA_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t _e = (call);                                                             \
        if (_e != hipSuccess) {                                                             \
            va::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__,  \
                          __LINE__);                                                        \
            return _e == hipErrorOutOfMemory ? VA_ERR_NOMEM : VA_ERR_HIP;                   \
        }                                                                                   \
    } while (0)

#define VA_LAUNCH_CHECK(name)                                                               \
    do {                                                                                    \
        hipError_t _e = hipGetLastError();                                                  \
        if (_e != hipSuccess) {                                                             \
            va::set_error("launch of %s failed: %s", name, hipGetErrorString(_e));          \
            return VA_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

#define VA_REQUIRE(cond, ...)                                                               \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            va::set_error(__VA_ARGS__);                                                     \
            return VA_ERR_INVALID;                                                          \
        }                                                                                   \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront width
// largest frame the kernels take: per-frame byte offsets (4-byte labels) travel in the 32-bit
// fields of raw buffer descriptors / uint32 arithmetic, so h*w*4 must stay below 2^31
constexpr size_t kMaxFramePixels = (size_t)1 << 29;

// Optional per-stage timing with HIP events on the pipeline's own stream (bench.py's roofline
// numbers come from here).  Events are only recorded, never waited for, inside a run.
struct StageProfiler {
    static constexpr int kMaxMarks = 4096;
    bool enabled = false;
    int every = 1, runs = 0;     // events go into every `every`-th run (an event per stage costs the stream ~2 %)
    int n = 0, dropped = 0;
    hipEvent_t ev[kMaxMarks];
    const char *name[kMaxMarks];  // nullptr = start of a run
    bool created[kMaxMarks] = {};
    void mark(const char *nm, hipStream_t st)
    {
        if (!enabled)
            return;
        if (n >= kMaxMarks) {
            dropped++;
            return;
        }
        if (!created[n]) {
            if (hipEventCreate(&ev[n]) != hipSuccess)
                return;
            created[n] = true;
        }
        if (hipEventRecord(ev[n], st) != hipSuccess)
            return;
        name[n++] = nm;
    }
};

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
inline int words_per_row(int w) { return (w + 31) / 32; }

// ---- host-side kernels shared between translation units --------------------------------
// Gaussian taps (host): OpenCV's 8-bit fixed-point / float definitions. Return 0 or VA_ERR_*.
int gauss_ksize(double sigma, bool is_u8);
int gauss_taps_q8(double sigma, int *ksize, uint16_t *taps, int cap, int rule = VA_TAPS_CV4);
int gauss_taps_f32(double sigma, int *ksize, float *taps, int cap);

// ---- launchers (each enqueues on `stream`, returns VA_OK or an error) -------------------
constexpr int kMaxTaps = 255;  // by-value tap tables in the kernel arguments
struct TapsQ8 {
    int ksize;
    uint16_t t[kMaxTaps + 1];
};
struct TapsF32 {
    int ksize;
    float t[kMaxTaps + 1];
};

// generic (any radius / channel count) two-pass Gaussian through a u16 / f32 scratch in HBM
// scratch: gauss_generic_u8_scratch_bytes(n*h*w*c, taps) bytes (16-bit row sums while the tap sum allows)
size_t gauss_generic_u8_scratch_bytes(size_t count, const TapsQ8 &taps);
int launch_gauss_generic_u8(const uint8_t *src, uint8_t *dst, void *scratch, int n, int h,
                            int w, int c, const TapsQ8 &taps, hipStream_t st);
int launch_gauss_generic_f32(const float *src, float *dst, float *scratch, int n, int h, int w,
                             int c, const TapsF32 &taps, hipStream_t st);
// fast float32 path: LDS-staged row pass + register-window column pass (1 or 3 channels)
bool gauss_f32_fast_supported(int w, int c, const TapsF32 &taps);
int launch_gauss_f32_fast(const float *src, float *dst, float *scratch, int n, int h, int w, int c,
                          const TapsF32 &taps, hipStream_t st);
// float32 path in two kernels (va_gauss_f32_fused.hip): [EMA background + |difference| + row pass]
// with the background state in registers, then a marching column pass.  bg == nullptr: plain blur.
#if defined(__HIPCC__)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding
// GLOBAL access of the wave (s_waitcnt vmcnt(0)): with prefetch loads in flight for a later step, or
// streaming stores behind every step, that wait exposes a memory round trip per barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#endif
void gauss_f32_test_hook(int generic_columns);
bool gauss_f32_fused_supported(int h, int w, int c, const TapsF32 &taps);
int launch_gauss_f32_fused(const float *src, float *dst, float *scratch, const float *bg, float *bg_out,
                           int64_t n_seen, double rate, int n, int h, int w, int c, const TapsF32 &taps,
                           hipStream_t st, StageProfiler *prof = nullptr);
// fused single-channel u8 Gaussian (LDS-staged, dot4/dot2), radius <= 31; output either the
// blurred u8 frames (dst), and/or the thresholded bit mask (bits, blurred > thresh)
bool gauss_fused_supported(int w, int h, const TapsQ8 &taps);
int launch_gauss_fused_u8(const uint8_t *src, uint8_t *dst, uint32_t *bits, int thresh, int n,
                          int h, int w, const TapsQ8 &taps, hipStream_t st);
// the same contract on the matrix cores (Toeplitz products with v_mfma_i32_32x32x32_i8):
// taps <= 127, radius <= 16; preferred whenever it applies
bool gauss_mfma_supported(int w, int h, const TapsQ8 &taps);
// mask8_maxval > 0: dst receives the thresholded mask as bytes (maxval / 0) instead of the blur
int launch_gauss_mfma_u8(const uint8_t *src, uint8_t *dst, uint32_t *bits, int thresh, int n,
                         int h, int w, const TapsQ8 &taps, hipStream_t st, int mask8_maxval = 0);

// recip_scratch: bg_scratch_bytes(n) bytes of device memory for the per-frame reciprocals of the
// division-free running mean of batches above 256 frames (nullptr: the plain-division kernel is used);
// mean_in_u8_range: the caller vouches that the running-mean state lies in [0, 255] (saturation-free kernel)
size_t bg_scratch_bytes(int n);
int launch_bg(int mode, int dtype, const void *frames, void *diff, void *state, int64_t n_seen,
              double rate, int n, size_t px, hipStream_t st, double *recip_scratch = nullptr,
              bool mean_in_u8_range = false);
int launch_welford(const uint8_t *frames, double *mean, double *m2, int64_t n_seen, int n,
                   size_t px, hipStream_t st);

// running mean (m2 == nullptr) or Welford mean + M2 over frames of dtype VA_U8 / VA_I16 / VA_F32
int launch_temporal_stats(const void *frames, int dtype, double *mean, double *m2, int64_t n_seen, int n,
                          size_t px, hipStream_t st);
int launch_threshold_u8(const uint8_t *src, uint8_t *dst, size_t count, int thresh, int maxval,
                        hipStream_t st);
int launch_time_difference(const uint8_t *a, const uint8_t *b, int16_t *out, size_t count,
                           hipStream_t st);
int launch_mono_mean(const uint8_t *src, uint8_t *dst, size_t pixels, hipStream_t st);
// (va_synth.hip) four samples per thread, table-driven normalisation; false: shape not supported, nothing launched
bool launch_pointwise_u8_x4(const uint8_t *src, uint8_t *dst, size_t out_samples, int src_c, int mono, int normalize,
                            double fmin, double fmax, double alpha, double tmin, hipStream_t st);
// (n, h, w, c) interleaved u8 <-> (n, c, h, wp) planes padded to wp columns by reflection
int launch_channel_planes(const uint8_t *src, uint8_t *dst, int n, int h, int w, int wp, int c,
                          bool split, hipStream_t st);
// np.rot90(frame, k) on (N,H,W) frames of opaque elem_bytes-byte pixels
int launch_rot90(const void *src, void *dst, int n, int h, int w, int elem_bytes, int k,
                 hipStream_t st);
int launch_normalize_u8(const uint8_t *src, uint8_t *dst, size_t count, double fmin, double fmax,
                        double alpha, double tmin, hipStream_t st);

// u8 (src > thresh) or (src != 0 when thresh < 0 ... see .hip) -> bit mask, and back
int launch_pack_bits(const uint8_t *src, uint32_t *bits, int n, int h, int w, int thresh,
                     hipStream_t st);
int launch_unpack_bits(const uint32_t *bits, uint8_t *dst, int n, int h, int w, int maxval,
                       hipStream_t st);

// structuring element as per-row horizontal spans (RECT/CROSS/ELLIPSE are all row-convex)
struct RowSpans {
    int ksize;
    int anchor;
    int8_t lo[64];  // first / last+1 column (relative to the window's left edge); lo>=hi: empty
    int8_t hi[64];
};
int make_row_spans(int shape, int ksize, RowSpans *out);
// scratch (n*h*w bytes, nullable): lets rectangular elements run as a row pass + a column pass
int launch_morph_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int op,
                    const RowSpans &se, hipStream_t st, uint8_t *scratch = nullptr);
int launch_morph_bits(const uint32_t *src, uint32_t *dst, int n, int h, int w, int op,
                      const RowSpans &se, hipStream_t st);

// small stencils (va_stencil.hip)
int launch_detect_peaks(const uint8_t *src, uint8_t *dst, int n, int h, int w, int include_plateaus,
                        hipStream_t st);
int launch_detect_peaks_f32(const float *src, uint8_t *dst, int n, int h, int w, int include_plateaus,
                            hipStream_t st);
int launch_image_statistics_f32(const float *src, double *mean_out, double *var_out, int n, int h, int w,
                                const RowSpans &se, double prior, int exclude_center, hipStream_t st);
int launch_thinning_step(const uint8_t *img, uint8_t *eroded, uint8_t *skel, int n, int h, int w,
                         unsigned long long *nonzero, hipStream_t st);
size_t image_statistics_scratch_bytes(int n, int h, int w);
int launch_image_statistics(const uint8_t *src, double *mean_out, double *var_out, int n, int h,
                            int w, const RowSpans &se, double prior, int exclude_center,
                            void *scratch, hipStream_t st);

// the whole op sequence in one kernel (register-streaming for one or two small rectangles,
// LDS-resident otherwise)
bool morph_fused_supported(int w, const RowSpans *se, int count);
int launch_morph_fused(const uint32_t *src, uint32_t *dst, int n, int h,
                       int w, const int *ops, const RowSpans *se, int count, hipStream_t st);

// connected components on bit masks; labels doubles as the union-find forest
size_t ccl_workspace_bytes(int n, int h, int w);
// true: launch_ccl labels with one workgroup per frame (forest in LDS); false: chip-wide
// multi-pass path (large frames, small batches, or the test hook)
bool ccl_frame_kernel_used(int n, int h, int w);
void ccl_test_hook(int path, int lds_runs);   // see va_test_hook_labelling
size_t ccl_rows_workspace_bytes(int n, int h);   // launch_ccl's workspace (the caller owns the bit mask)
int launch_ccl(const uint32_t *bits, int32_t *labels, int32_t *counts, int n, int h, int w,
               int connectivity, void *workspace, size_t ws_bytes, int64_t *stats, int max_labels,
               hipStream_t st, StageProfiler *prof = nullptr,
               bool paint = true);
// the same in two halves: the labelling itself, and the write of the label image (+ per-label
// statistics), which reads only what the plan names -- the final bit mask, the run tables in the
// workspace (or the sparse forest words in the label image) -- and may run on another stream
struct CclPaintPlan {
    const uint32_t *bits;
    int32_t *labels;
    int n, h, w;          // n == 0: nothing to paint
    int64_t *stats;
    int max_labels;
    const int32_t *run_table, *row_off, *frame_mode;
    int table_stride, xcd_frames;
    int persistent_grid;   // > 0: paint with this many workgroups (multiple of 8), each walking several row blocks
};
int launch_ccl_front(const uint32_t *bits, int32_t *labels, int32_t *counts, int n, int h, int w,
                     int connectivity, void *workspace, size_t ws_bytes, int64_t *stats, int max_labels,
                     hipStream_t st, StageProfiler *prof, CclPaintPlan *plan);
int launch_ccl_paint(const CclPaintPlan &plan, hipStream_t st, StageProfiler *prof = nullptr);
// outer contour of the component with the largest contour area (8-connectivity), on a forest
// prepared by launch_ccl(..., paint = false): roots hold -(label) at their first pixel
int launch_largest_contour(const uint32_t *bits, const int32_t *forest, int n, int h, int w,
                           unsigned long long *best_keys, int32_t *points, int max_points,
                           int32_t *npoints, double *area, hipStream_t st);
// FilterNormalize for uint8 / float32 frames, any of the three target dtypes; seeded noise frames
int launch_normalize(const void *src, int src_dtype, void *dst, int dst_dtype, size_t count, double fmin,
                     double fmax, double alpha, double tmin, hipStream_t st);
int launch_prepare_u8(const uint8_t *src, uint8_t *dst, int n, int src_h, int src_w, int src_c, int left, int top,
                      int width, int height, int mono, int normalize, double fmin, double fmax, double alpha,
                      double tmin, hipStream_t st);
int launch_gaussian_noise(void *dst, int dtype, size_t count, double mean, double stdev, uint64_t seed,
                          uint64_t first_index, hipStream_t st);
// cv2.resize for uint8 / float32 frames (va_resize.hip); mode 0 nearest, 1 linear, 2 cubic, 3 area, 4 lanczos4
size_t resize_scratch_bytes(int sh, int sw, int dh, int dw);
int launch_resize_u8(const uint8_t *src, uint8_t *dst, int n, int sh, int sw, int c, int dh, int dw, int mode,
                     void *scratch, hipStream_t st);
int launch_resize_f32(const float *src, float *dst, int n, int sh, int sw, int c, int dh, int dw, int mode,
                      void *scratch, hipStream_t st);
// cv2.moments(contour): ten spatial moments (float64) per contour, points int32 or float32 (x, y)
int launch_contour_moments(const void *points, const int32_t *npoints, int n, int max_points,
                           int is_float, double *out, hipStream_t st);
int launch_stats_from_labels(const int32_t *labels, int n, int h, int w, int max_labels,
                             int64_t *stats, hipStream_t st);
int launch_largest_region(const int32_t *labels, const int32_t *counts, const int64_t *stats,
                          int n, int h, int w, int max_labels, int32_t *largest,
                          int64_t *largest_area, uint8_t *mask_out, hipStream_t st);

}  // namespace va
