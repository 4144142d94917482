/*
 * videoanalysis_hip.h -- C ABI of libvideoanalysis_hip.so (MI355X / gfx950)
 *
 * The drop-in boundary for the data-parallel hot path of david-zwicker/video-analysis:
 * the `video.filters` chain (background subtraction, Gaussian blur, threshold, morphology)
 * and the per-frame `video.analysis` image ops (labelling, areas, bounding boxes, moments).
 *
 * The reference is pure Python and has no FFI of its own; its boundary for this path is the
 * `VideoFilterBase._process_frame(frame) -> frame` protocol (video/io/base.py:182-189,
 * 369-380) plus the free functions of the video/analysis modules.  Each entry point below names the
 * reference call it replaces (paths relative to the reference checkout).  The ctypes stubs a
 * maintainer would add to the reference are shown in INTEGRATION.md; the build's own Python
 * host (video-analysis_amd/video/_hip.py) binds exactly these symbols.
 *
 * Conventions
 *   - plain C types only; every function returns 0 (VA_OK) or a negative errno-style code and
 *     never throws; va_last_error() gives the message of the calling thread's last failure.
 *   - "dev" pointers are HIP device pointers owned by the caller (e.g. torch tensors'
 *     data_ptr()); `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     kernels are enqueued asynchronously on `stream`; nothing synchronises unless stated.
 *   - frames are row-major contiguous (N, H, W[, C]); size=(W,H) as in video/io/base.py:119-125.
 *   - bit masks ("bits") are (N, H, ceil(W/32)) uint32, pixel x <-> bit (x & 31) of word x>>5,
 *     padding bits are 0.
 *   - threads: one device per process (va_init fixes it; every entry point selects it for the
 *     calling thread).  The stand-alone entry points keep no state between calls: their device
 *     scratch is a stream-ordered allocation on the call's own `stream`, so different threads may
 *     call them concurrently on different streams (the reference's VideoPreprocessor workers,
 *     video/io/parallel.py:398-400).  A va_pipeline_t handle owns its scratch and background
 *     state and must be used by one thread / one stream at a time.
 *   - sizes: frames of up to 2^29 - 1 pixels (the kernels address h*w*4 bytes through 32-bit
 *     buffer descriptors); larger frames are refused with VA_ERR_INVALID.
 */
#ifndef VIDEOANALYSIS_HIP_H
#define VIDEOANALYSIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VA_OK 0
#define VA_ERR_INVALID (-22) /* EINVAL  bad argument                     */
#define VA_ERR_NOMEM (-12)   /* ENOMEM  device/host allocation failed   */
#define VA_ERR_HIP (-5)      /* EIO     a HIP runtime call failed       */
#define VA_ERR_NODEV (-19)   /* ENODEV  no usable GPU                   */
#define VA_ERR_RANGE (-34)   /* ERANGE  capacity (max_labels, ...) exceeded */

/* dtypes */
#define VA_U8 0
#define VA_F32 1
#define VA_F64 2 /* targets of va_normalize / va_gaussian_noise only */
#define VA_I16 3 /* int16 frames (FilterTimeDifference's output): va_mean_any / va_welford_any only */
/* background modes (BUILD-DEFINED FilterBackground; arithmetic of video/analysis/video.py) */
#define VA_BG_NONE 0
#define VA_BG_MEAN 1   /* cumulative mean, float64 state: measure_mean, video/analysis/video.py:33 */
#define VA_BG_EMA 2    /* bg += rate*(frame-bg), float32 state (no reference counterpart)        */
#define VA_BG_STATIC 3 /* fixed float64 background image (e.g. a measure_mean() result)          */
/* morphology */
/* which OpenCV's 8-bit Gaussian taps (both written from upstream knowledge, unverifiable offline):
 * CV4: unsigned 8.8 fixed point with error diffusion, sum forced to 256 (OpenCV >= 4.x; default);
 * CV3: float32 getGaussianKernel, every tap cvRound(k * 256) on its own, sum not forced -- OpenCV 2.4 / 3.x,
 *      the era of the reference (cv2.findContours(...)[1], video/analysis/regions.py:180-182) */
#define VA_TAPS_CV4 0
#define VA_TAPS_CV3 1

#define VA_MORPH_ERODE 0
#define VA_MORPH_DILATE 1
#define VA_SHAPE_RECT 0    /* cv2.MORPH_RECT    */
#define VA_SHAPE_CROSS 1   /* cv2.MORPH_CROSS   (video/analysis/image.py:248) */
#define VA_SHAPE_ELLIPSE 2 /* cv2.MORPH_ELLIPSE */
#define VA_MAX_MORPH_OPS 4
#define VA_STATS_STRIDE 16 /* int64 per label, see va_moments_i64 */

/* ------------------------------------------------------------------ runtime / errors */
int va_init(int device);               /* select + warm up the GPU; VA_ERR_NODEV if none;
                                          a second, different device is VA_ERR_INVALID        */
int va_device_count(void);             /* number of visible GPUs (0 if none / no driver)     */
const char *va_version(void);
const char *va_last_error(void);       /* message for the calling thread's last failure      */

/* device memory helpers, so that a host without torch can drive the library */
/* The stand-alone entry points keep their device scratch between calls (a per-stream cache of blocks, at most
 * 6 GiB); va_trim synchronises the device and hands the cache (when it holds more than keep_bytes) and the default
 * memory pool's pages above keep_bytes back -- e.g. before another library in the process needs the memory. */
int va_trim(size_t keep_bytes);
int va_malloc(void **dev_ptr, size_t bytes);
int va_free(void *dev_ptr);
int va_host_alloc(void **host_ptr, size_t bytes); /* pinned host memory */
int va_host_free(void *host_ptr);
/* Host <-> device copies.  Pinned host memory (va_host_alloc, or registered by the caller): asynchronous on `stream`.
 * Pageable host memory (a NumPy array): the call waits for the stream's earlier work and returns when the copy is
 * complete -- large asynchronous copies to pageable memory were seen to leave part of the destination unwritten
 * after the stream had been synchronised. */
int va_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int va_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int va_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int va_memset(void *dst_dev, int value, size_t bytes, void *stream);
int va_stream_sync(void *stream);

/* streams and events, for hosts that overlap uploads, the chain and downloads (the role of the
 * reference's reader process / VideoPreprocessor threads, video/io/parallel.py:345-488) */
int va_stream_create(void **stream_out);
int va_stream_destroy(void *stream);
int va_event_create(void **event_out);
int va_event_destroy(void *event);
int va_event_record(void *event, void *stream);
int va_stream_wait_event(void *stream, void *event);
int va_event_sync(void *event);
int va_event_elapsed_ms(void *start_event, void *stop_event, float *ms_out);

/* ------------------------------------------------------------------ A1 Gaussian blur
 * replaces  cv2.GaussianBlur(frame.astype(np.uint8), (0, 0), sigma)
 *           FilterBlur._process_frame, video/filters.py:388-392
 * 8-bit: ksize = cvRound(6 sigma + 1)|1, unsigned 8.8 fixed-point taps, BORDER_REFLECT_101,
 * each of the `c` interleaved channels independently.  src != dst. */
int va_gaussian_u8(const uint8_t *src_dev, uint8_t *dst_dev, int n, int h, int w, int c,
                   double sigma, void *stream);
/* replaces  cv2.GaussianBlur(float_image, (0, 0), sigma), video/analysis/active_contour.py:108
 * ksize = cvRound(8 sigma + 1)|1, float32 taps and fmaf accumulation. */
int va_gaussian_f32(const float *src_dev, float *dst_dev, int n, int h, int w, int c,
                    double sigma, void *stream);
/* the same with the tap set named explicitly (VA_TAPS_CV4 = va_gaussian_u8's, VA_TAPS_CV3 = the
 * reference-era definition; same row/column arithmetic: integer row sums, (acc + 2^15) >> 16 saturated) */
int va_gaussian_u8_rule(const uint8_t *src_dev, uint8_t *dst_dev, int n, int h, int w, int c,
                        double sigma, int tap_rule, void *stream);
/* analytic taps (host side, no GPU needed): q8.8 taps sum to 256 */
int va_gauss_taps_q8(double sigma, int *ksize_out, uint16_t *taps_out, int capacity);
int va_gauss_taps_q8_rule(double sigma, int tap_rule, int *ksize_out, uint16_t *taps_out, int capacity);
int va_gauss_taps_f32(double sigma, int *ksize_out, float *taps_out, int capacity);

/* ------------------------------------------------------------------ A2 background model
 * replaces  mean = mean*n/(n + 1) + frame/(n + 1)       measure_mean, video/analysis/video.py:33
 * and implements the BUILD-DEFINED FilterBackground: for each of the n frames IN ORDER
 *     diff = sat_u8(trunc(|frame - bg|))   (f32 frames: |frame - bg|)     then update bg.
 * mode VA_BG_MEAN  : state = float64[px], n_seen = frames already folded into it
 * mode VA_BG_EMA   : state = float32[px], rate; the very first frame initialises bg = frame
 * mode VA_BG_STATIC: state = float64[px], read only
 * dtype VA_U8 (all modes) or VA_F32 (EMA only).  diff_out may be NULL (state update only). */
int va_bg_update(int mode, int dtype, const void *frames_dev, void *diff_out_dev,
                 void *state_dev, int64_t n_seen, double rate, int n, size_t px, void *stream);
/* measure_mean / measure_mean_std over frames of any dtype the reference meets (video/analysis/video.py:26-55):
 * dtype VA_U8, VA_I16 (FilterTimeDifference's int16) or VA_F32.  NumPy's promotions are kept:
 * `frame/(n + 1)` is rounded to float32 first for float32 frames, float64 otherwise. */
int va_mean_any(const void *frames_dev, int dtype, double *mean_dev, int64_t n_seen, int n, size_t px,
                void *stream);
int va_welford_any(const void *frames_dev, int dtype, double *mean_dev, double *m2_dev, int64_t n_seen,
                   int n, size_t px, void *stream);
/* replaces  measure_mean_std's Welford update, video/analysis/video.py:48-50 (float64 state) */
int va_welford_u8(const uint8_t *frames_dev, double *mean_dev, double *m2_dev, int64_t n_seen,
                  int n, size_t px, void *stream);

/* ------------------------------------------------------------------ A3 / A4 / A5 pointwise
 * replaces  this_frame.astype(np.int16) - prev_frame, FilterTimeDifference._compare_frames,
 *           video/filters.py:564-568 */
int va_time_difference_u8(const uint8_t *this_dev, const uint8_t *prev_dev, int16_t *out_dev,
                          size_t count, void *stream);
/* BUILD-DEFINED FilterThreshold; the reference idiom is `frame > t` boolean masks
 * (video/analysis/image.py:282,288,304):  out = src > thresh ? maxval : 0 */
int va_threshold_u8(const uint8_t *src_dev, uint8_t *dst_dev, size_t count, int thresh,
                    int maxval, void *stream);
/* replaces  np.mean(frame, axis=2).astype(frame.dtype), FilterMonochrome._process_frame,
 *           video/filters.py:365-366 (float64 mean of 3 channels, truncated) */
int va_mono_mean_u8(const uint8_t *src_dev, uint8_t *dst_dev, size_t pixels, void *stream);
/* replaces  FilterNormalize._process_frame, video/filters.py:126-132, for u8 -> u8:
 *           clip to [fmin,fmax]; (f - fmin)*alpha + tmin in float64; astype(uint8) */
int va_normalize_u8(const uint8_t *src_dev, uint8_t *dst_dev, size_t count, double fmin,
                    double fmax, double alpha, double tmin, void *stream);
/* the same for float32 or uint8 frames and uint8 / float32 / float64 targets (the reference takes
 * any dtype, video/filters.py:101-135): clip, affine map in float64, C cast to the target */
int va_normalize(const void *src_dev, int src_dtype, void *dst_dev, int dst_dtype, size_t count,
                 double fmin, double fmax, double alpha, double tmin, void *stream);
/* FilterCrop -> FilterMonochrome -> FilterNormalize in ONE pass over the source frames, on the device
 * (A5: the pointwise pre-stages of a chain, video/filters.py:238-248, 359-374, 126-132), so that a
 * chain which starts with them feeds the engine without a host round trip:
 *   crop = frame[top:top+height, left:left+width]; mono: -1 keep the channels, 0..2 that channel,
 *   3 np.mean(axis=2).astype(uint8); normalize != 0: clip, (f - fmin)*alpha + tmin, astype(uint8).
 * src (n, src_h, src_w, src_c) uint8 -> dst (n, height, width[, src_c when mono == -1]). */
#define VA_MONO_KEEP (-1)
#define VA_MONO_MEAN 3
int va_prepare_u8(const uint8_t *src_dev, uint8_t *dst_dev, int n, int src_h, int src_w, int src_c, int left,
                  int top, int width, int height, int mono, int normalize, double fmin, double fmax,
                  double alpha, double tmin, void *stream);
/* replaces  self.mean + self.std*np.random.randn(*self._frame_shape), VideoGaussianNoise.get_frame,
 *           video/io/computed.py:36-41, on the device (N4): sample i of the stream is a pure
 * function of (seed, i) -- Philox4x32-10 counter, Box-Muller in float64 -- so any frame can be
 * produced on its own: count samples starting at absolute sample index first_index
 * (= frame_index * samples_per_frame).  dtype VA_U8 (saturated to [0, 255], truncated), VA_F32
 * or VA_F64.  The reference's stream is unseeded NumPy state: agreement is statistical. */
int va_gaussian_noise(void *dst_dev, int dtype, size_t count, double mean, double stdev, uint64_t seed,
                      uint64_t first_index, void *stream);
/* replaces  np.rot90(frame, angle // 90), FilterRotate._process_frame, video/filters.py:339-344
 * (N4): n frames (h, w) of opaque elem_bytes-byte pixels (channels x dtype: 1, 2, 3, 4, 6, 8 or
 * 12 bytes) turned k quarter turns counter-clockwise; output frames are (w, h) for odd k. */
int va_rot90(const void *src_dev, void *dst_dev, int n, int h, int w, int elem_bytes, int k,
             void *stream);

/* replaces  cv2.resize(frame, self.size, interpolation=...), FilterResize._process_frame,
 *           video/filters.py:310-314 (N4).  uint8 frames (n, src_h, src_w[, c]) -> (n, dst_h, dst_w[, c]),
 * c <= 4 interleaved channels.  OpenCV's 8-bit definitions: nearest = floor(x * src/dst); linear and
 * cubic (A = -0.75) with 11-bit fixed-point weights and OpenCV's rounding steps (an exact 2x2
 * linear shrink is the area mean, as there); area = block means for integer shrink factors,
 * float cell-overlap weights for other shrinks, linear with area-style positions when growing;
 * lanczos4 (video/filters.py:293-294) = 8 x 8 taps, 11-bit fixed point, int32 accumulation.  src != dst.
 * va_resize_f32: float32 frames (FilterResize takes the video's dtype): the float instantiations of the
 * same algorithms -- float coefficients, products summed from the first tap to the last, no rounding. */
#define VA_INTER_NEAREST 0
#define VA_INTER_LINEAR 1
#define VA_INTER_CUBIC 2
#define VA_INTER_AREA 3
#define VA_INTER_LANCZOS4 4
int va_resize_u8(const uint8_t *src_dev, uint8_t *dst_dev, int n, int src_h, int src_w, int c, int dst_h,
                 int dst_w, int interpolation, void *stream);
int va_resize_f32(const float *src_dev, float *dst_dev, int n, int src_h, int src_w, int c, int dst_h,
                  int dst_w, int interpolation, void *stream);

/* ------------------------------------------------------------------ A6 morphology
 * replaces  cv2.erode / cv2.dilate(img, cv2.getStructuringElement(shape, (k, k))),
 *           video/analysis/image.py:248-251; anchor = centre, pixels outside the image never
 *           win (OpenCV's default border).  src != dst.  (n, h, w) u8. */
int va_morph_u8(const uint8_t *src_dev, uint8_t *dst_dev, int n, int h, int w, int op,
                int shape, int ksize, void *stream);

/* ------------------------------------------------------------------ A7 labelling
 * replaces  labels, num = ndimage.measurements.label(mask), video/analysis/regions.py:162
 * any non-zero mask byte is foreground; connectivity 4 (SciPy default) or 8; int32 labels
 * 1..L numbered in raster order of each component's first pixel; counts[f] = L of frame f.
 * labels_dev: (n,h,w) int32 (also used as the union-find forest while running).
 * workspace: va_label_workspace_bytes(n,h,w) bytes of device scratch. */
size_t va_label_workspace_bytes(int n, int h, int w);
int va_label_i32(const uint8_t *mask_dev, int32_t *labels_dev, int32_t *counts_dev, int n,
                 int h, int w, int connectivity, void *workspace_dev, size_t workspace_bytes,
                 void *stream);

/* ------------------------------------------------------------------ A7/A8/A9 per-label stats
 * replaces  [np.sum(labels == l) ...], video/analysis/regions.py:165-166;
 *           find_bounding_box, video/analysis/regions.py:113-149;
 *           cv2.moments(mask.astype(np.uint8)) spatial moments, video/analysis/image.py:353
 * stats_dev: (n, max_labels, 16) int64, for label l at [l-1]:
 *   0 area(m00) 1 m10 2 m01 3 m20 4 m11 5 m02 6 m30 7 m21 8 m12 9 m03
 *   10 xmin 11 ymin 12 xmax 13 ymax 14,15 reserved
 * labels above max_labels are ignored (check counts against max_labels on the host). */
int va_moments_i64(const int32_t *labels_dev, int n, int h, int w, int max_labels,
                   int64_t *stats_dev, void *stream);
/* replaces  label_max = np.argmax(areas) + 1; labels == label_max,
 *           get_largest_region, video/analysis/regions.py:169-174 (first maximum wins).
 * largest_dev[f] = label_max (0 when the frame is empty); mask_out_dev (nullable) = 0/1 u8. */
int va_largest_region(const int32_t *labels_dev, const int32_t *counts_dev,
                      const int64_t *stats_dev, int n, int h, int w, int max_labels,
                      int32_t *largest_dev, int64_t *largest_area_dev, uint8_t *mask_out_dev,
                      void *stream);

/* ------------------------------------------------------------------ A8 contour of the largest region
 * replaces  contours = cv2.findContours(mask.astype(np.uint8), cv2.RETR_EXTERNAL,
 *                                       cv2.CHAIN_APPROX_SIMPLE)[1]
 *           contour_id = np.argmax([cv2.contourArea(c) for c in contours])
 *           get_contour_from_largest_region, video/analysis/regions.py:178-197
 * 8-connected foreground (any non-zero byte), Suzuki-Abe outer border following from each
 * component's first raster pixel, points kept only where the direction changes, OpenCV's
 * most-recent-first contour order for the argmax tie rule.
 * points_dev: (n, max_points, 2) int32 (x, y); npoints_dev[f] = points of the winning contour
 * (if > max_points only the first max_points were stored); area_dev[f] = cv2.contourArea of it;
 * ncomponents_dev[f] = number of 8-connected components (0 -> "Could not find any contour"). */
size_t va_contour_workspace_bytes(int n, int h, int w);
int va_largest_contour(const uint8_t *mask_dev, int n, int h, int w, int32_t *points_dev,
                       int max_points, int32_t *npoints_dev, double *area_dev,
                       int32_t *ncomponents_dev, void *workspace_dev, size_t workspace_bytes,
                       void *stream);

/* ------------------------------------------------------------------ A9 contour moments
 * replaces  cv2.moments(contour), regionprops(contour=...), video/analysis/image.py:355, and
 *           cv2.moments(np.asarray(self.contour, np.float32)), Polygon.moments,
 *           video/analysis/shapes.py:527-533
 * Green's-theorem moments of n closed polygons, accumulated in float64 in OpenCV's point order
 * (bit-identical to its contourMoments; m00 >= 0 for either orientation; a degenerate contour
 * gives zeros).  points_dev: (n, max_points, 2) int32 (is_float == 0) or float32 (x, y) -- e.g.
 * the output of va_largest_contour; npoints_dev[f] = points of contour f (NULL: max_points each).
 * moments_out_dev: (n, 10) float64 = m00 m10 m01 m20 m11 m02 m30 m21 m12 m03; the central and
 * normalised moments follow on the host (completeMomentState). */
int va_contour_moments(const void *points_dev, const int32_t *npoints_dev, int n, int max_points,
                       int is_float, double *moments_out_dev, void *stream);

/* ------------------------------------------------------------------ N2 small stencils
 * replaces  detect_peaks(img, include_plateaus), video/analysis/image.py:267-306:
 *           ndimage.maximum_filter(img, footprint=8-neighbourhood) == img, minus the
 *           binary_erosion(img == 0, 8-neighbourhood, border_value=1) background (plateaus), or
 *           img > maximum over the 8 neighbours.  dst: 0/1 u8 mask. */
int va_detect_peaks_u8(const uint8_t *src_dev, uint8_t *dst_dev, int n, int h, int w,
                       int include_plateaus, void *stream);
/* the same on float32 maps (the reference calls it on distance / correlation maps): comparisons in float,
 * background = (img == 0) */
int va_detect_peaks_f32(const float *src_dev, uint8_t *dst_dev, int n, int h, int w, int include_plateaus,
                        void *stream);
/* replaces  the python fallback of mask_thinning, video/analysis/image.py:243-258 (3x3 cross):
 *           eroded = cv2.erode(img); temp = cv2.dilate(eroded); cv2.subtract(img, temp, temp);
 *           cv2.bitwise_or(skel, temp, skel); img = eroded   ... until img is empty.
 * img_dev is consumed (used as ping-pong scratch with scratch_dev); skel_dev receives the
 * skeleton.  Synchronises the stream once per 16 iterations (the loop's exit test reads one
 * survivor counter per iteration; steps past the emptying one change nothing). */
int va_mask_thinning_u8(uint8_t *img_dev, uint8_t *scratch_dev, uint8_t *skel_dev, int h, int w,
                        int *iterations_out, void *stream);
/* replaces  get_image_statistics, video/analysis/image.py:131-201: local mean and variance in a
 *           (2*ksize+1)^2 box or ellipse window of (img - prior), zero border.
 * kernel: 0 box, 1 ellipse.  mean_out_dev / var_out_dev: (n,h,w) float64; var_out may be NULL. */
int va_image_statistics_u8(const uint8_t *src_dev, double *mean_out_dev, double *var_out_dev, int n,
                           int h, int w, int kernel, int ksize, double prior, int exclude_center,
                           void *stream);

/* float32 images: truncated towards zero like the reference's `img.astype(np.int) - prior`
 * (video/analysis/image.py:175), then direct window sums in float64 */
int va_image_statistics_f32(const float *src_dev, double *mean_out_dev, double *var_out_dev, int n,
                            int h, int w, int kernel, int ksize, double prior, int exclude_center,
                            void *stream);

/* ------------------------------------------------------------------ fused pipeline
 * One handle per filter chain (not thread-safe; the reference's pull model is single-threaded,
 * video/io/base.py:207-223).  Runs, for a batch of n <= max_batch frames resident in HBM:
 *   bg-sub -> Gaussian -> threshold -> morphology ops -> labelling (+ stats)
 * i.e. FilterBackground -> FilterBlur -> FilterThreshold -> FilterMorphology ->
 * get_largest_region's label/areas, with bit-packed masks between the stages. */
typedef struct va_config {
    int32_t struct_size; /* = sizeof(va_config) */
    int32_t width, height, channels;
    int32_t dtype;     /* VA_U8 | VA_F32 */
    int32_t max_batch; /* frames per va_pipeline_run call, upper bound */
    int32_t bg_mode;   /* VA_BG_* */
    float bg_rate;     /* EMA rate */
    double sigma;      /* <= 0: no blur */
    int32_t thresh;    /* < 0: stop after the blur (no mask / labels) */
    int32_t maxval;    /* mask value written to mask_out (default 255) */
    int32_t morph_count;
    int32_t morph_op[VA_MAX_MORPH_OPS];
    int32_t morph_shape[VA_MAX_MORPH_OPS];
    int32_t morph_ksize[VA_MAX_MORPH_OPS];
    int32_t connectivity; /* 0: no labelling, 4, 8 */
    int32_t max_labels;   /* > 0: per-label stats capacity per frame */
    int32_t tap_rule;     /* VA_TAPS_CV4 (0, default) | VA_TAPS_CV3: the 8-bit Gaussian's tap set */
} va_config;

typedef struct va_pipeline va_pipeline_t;

int va_pipeline_create(const va_config *cfg, va_pipeline_t **out);
int va_pipeline_destroy(va_pipeline_t *p);
/* frames_dev: (n,H,W[,C]) of cfg.dtype (float32 pipelines: 16-byte aligned, as is filtered_out_dev;
 * VA_ERR_INVALID otherwise).  Any output may be NULL:
 *   filtered_out_dev : (n,H,W[,C]) cfg.dtype, the blurred background-subtracted frames
 *   mask_out_dev     : (n,H,W) u8 0/maxval after threshold + morphology
 *   labels_out_dev   : (n,H,W) int32
 *   counts_out_dev   : (n) int32 components per frame
 *   stats_out_dev    : (n,max_labels,16) int64 */
int va_pipeline_run(va_pipeline_t *p, const void *frames_dev, int n, void *filtered_out_dev,
                    uint8_t *mask_out_dev, int32_t *labels_out_dev, int32_t *counts_out_dev,
                    int64_t *stats_out_dev, void *stream);
/* Overlapped runs (off by default).  The chain's last kernel -- the write of the int32 label image and
 * the per-label statistics, which get_largest_region / regionprops consume
 * (video/analysis/regions.py:159-174) -- is bound by HBM stores, the stages before it by VALU and
 * latency.  With enable != 0 va_pipeline_run enqueues that write on a stream the pipeline owns, ordered
 * after the labelling, and returns; the next va_pipeline_run starts its background / blur / morphology /
 * labelling stages on the caller's stream at once, beside it (mask buffers and run tables exist twice).
 * Contract while enabled: labels_out / stats_out of a run are complete on a stream only after
 * va_pipeline_fence(p, that stream) (or a device synchronisation); counts_out / mask_out /
 * filtered_out stay ordered on the run's own stream as before.  A caller that hands consecutive runs
 * the same labels_out / stats_out buffer stays correct (the run's labelling stage then waits for the
 * previous write); alternate two buffers to overlap that stage as well.
 * va_pipeline_overlap synchronises the device; it allocates the second buffer set on first use.  The
 * deferred write runs as a persistent kernel of 4 workgroups per CU on a lowest-priority stream (bounded
 * footprint beside the caller's kernels); $VA_PAINT_WGS_PER_CU overrides the 4 for measurements (0: one
 * workgroup per row block), enable == 2 gives the stream default priority. */
int va_pipeline_overlap(va_pipeline_t *p, int enable);
/* make `stream` wait for every label-image write this pipeline has enqueued so far (asynchronous:
 * enqueues waits, never blocks the host) */
int va_pipeline_fence(va_pipeline_t *p, void *stream);
/* background-model state, so that a shard can start mid-video (SURVEY.md 5 "checkpoint") */
int va_bg_get_state(va_pipeline_t *p, void *state_host, size_t bytes, int64_t *n_seen);
int va_bg_set_state(va_pipeline_t *p, const void *state_host, size_t bytes, int64_t n_seen);
size_t va_bg_state_bytes(const va_pipeline_t *p);
/* name of the implementation used for the Gaussian stage ("fused-lds" / "generic") */
const char *va_pipeline_describe(const va_pipeline_t *p);

/* per-stage device time, measured with HIP events recorded on the run's own stream (what
 * bench.py's `roofline` object is computed from).  enable != 0 starts/reset recording; every
 * va_pipeline_run (enable > 1: every enable-th run, the first one included -- an event per stage costs
 * the stream about 2 % of the chain) then records one event per stage (never waits).  stage_times waits for the
 * last event and sums, per stage name, the elapsed ms and the number of launches.
 * names: capacity x 32 chars. */
int va_pipeline_profile(va_pipeline_t *p, int enable);
int va_pipeline_stage_times(va_pipeline_t *p, int capacity, char *names, double *total_ms,
                            int32_t *launches, int *nstages_out);

/* ------------------------------------------------------------------ test hooks
 * Same contracts as va_gaussian_u8 / va_morph_u8, but forcing one implementation so that the
 * parity tests can compare the generic two-pass Gaussian with the single-launch kernels, and the
 * bit-packed morphology used inside the pipeline with the u8 one (binary masks: != 0 -> 255). */
int va_gaussian_u8_generic(const uint8_t *src_dev, uint8_t *dst_dev, int n, int h, int w, int c,
                           double sigma, void *stream);
/* the LDS/VALU (dot4/dot2) single-launch kernel that va_gaussian_u8 falls back to when the tap
 * set does not fit the matrix-core kernel (taps > 127); c must be 1 */
int va_gaussian_u8_valu(const uint8_t *src_dev, uint8_t *dst_dev, int n, int h, int w, int c,
                        double sigma, void *stream);
int va_morph_bits_u8(const uint8_t *src_dev, uint8_t *dst_dev, int n, int h, int w, int op,
                     int shape, int ksize, void *stream);
/* pins the labelling code path of every later va_label_i32 / va_largest_contour / pipeline call
 * of this process: path 0 = library's choice (per-frame LDS kernel when its cost model beats the
 * chip-wide passes: frames up to 1080p in batches of about 16 frames or more, 4K frames from
 * about 64, small frames in any batch; chip-wide passes otherwise), 1 = chip-wide passes, 2 = per-frame kernel,
 * 3 = the library's choice, but the per-frame kernel hands its labels to the paint pass as sparse words in the
 * label image (round 1's convention) instead of compact run tables; 4 = the per-frame kernel (as 2),
 * staging the mask rows in LDS even where it could read its spans straight from global memory;
 * lds_runs > 0 caps the per-frame kernel's run table (frames above it take its large-frame mode) */
int va_test_hook_labelling(int path, int lds_runs);
/* force_valu != 0: pipelines created from now on run their 8-bit Gaussian in the LDS/VALU (dot4/dot2)
 * kernel instead of the matrix-core one (same bits; bench.py times the chain both ways) */
int va_test_hook_gaussian_u8(int force_valu);
/* bit 0: every later float32 Gaussian of this process runs its column pass in the runtime-radius
 * kernel, also for the radii (r = 4, 8, ... 36) that have an unrolled one; bit 1: the same for the
 * row pass (compile-time-radius kernels exist for the same radii) */
int va_test_hook_gaussian_f32(int generic);

/* ------------------------------------------------------------------ multi-GPU (RCCL)
 * Frames shard across ranks with no data-path collective; the only exchange is the final
 * gather of per-frame object counts.  librccl is resolved lazily (dlopen) on first use.
 * id_out/id: 128-byte ncclUniqueId produced on rank 0 and distributed by the host. */
int va_comm_unique_id(uint8_t id_out[128]);
int va_comm_init(void **comm_out, int world_size, int rank, const uint8_t id[128]);
int va_gather_counts(void *comm, const int32_t *send_dev, int32_t *recv_dev,
                     int count_per_rank, void *stream);
int va_comm_destroy(void *comm);

#ifdef __cplusplus
}
#endif
#endif /* VIDEOANALYSIS_HIP_H */
