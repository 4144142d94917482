// va_api.hip -- the extern "C" surface of libvideoanalysis_hip.so (include/videoanalysis_hip.h)
//
// Host-side plumbing only: argument checks, scratch management, kernel sequencing for the
// fused pipeline, lazy RCCL binding.  No exception crosses the ABI; every failure sets the
// thread-local message returned by va_last_error().
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>
#include <vector>
#include <new>

#include "va_common.h"

namespace va {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }

// Device scratch of the stand-alone entry points: one lease per call, nothing shared between concurrent calls --
// the reference's concurrent callers (VideoPreprocessor's worker threads, video/io/parallel.py:398-400) may use one
// stream each without seeing each other's intermediates -- and no call synchronises the device (unless the cache
// below overflows).
// Scratch of the stand-alone entry points.  A block is NOT handed back to the stream-ordered allocator when a call
// returns: hipFreeAsync right behind the kernels, followed by a large asynchronous copy to pageable host memory,
// made the next call of the same size return a partly unwritten result (a 28 MB span of zeros in a 531 MB blur;
// leaking the block instead made it disappear: DESIGN.md 13.10).  Blocks are kept per stream -- the next call on
// the same stream is ordered behind the kernels that still use the block -- and released by va_trim or when more
// than kScratchCacheCap bytes are cached (then behind a stream synchronisation).
struct ScratchBlock {
    void *ptr;
    size_t bytes;
    hipStream_t st;
};
static std::mutex g_scratch_mu;
static std::vector<ScratchBlock> g_scratch_free;
static size_t g_scratch_cached = 0;
constexpr size_t kScratchCacheCap = 6ull << 30;

static void scratch_release_all_locked()
{
    for (const ScratchBlock &b : g_scratch_free)
        (void)hipFree(b.ptr);
    g_scratch_free.clear();
    g_scratch_cached = 0;
}

struct ScratchLease {
    void *ptr = nullptr;
    size_t bytes = 0;
    hipStream_t st = nullptr;
    int acquire(size_t need, hipStream_t stream)
    {
        st = stream;
        need = need ? need : 256;
        {
            std::lock_guard<std::mutex> lock(g_scratch_mu);
            int best = -1;
            for (int i = 0; i < (int)g_scratch_free.size(); i++) {
                const ScratchBlock &b = g_scratch_free[i];
                if (b.st == st && b.bytes >= need && (best < 0 || b.bytes < g_scratch_free[best].bytes))
                    best = i;
            }
            if (best >= 0 && g_scratch_free[best].bytes <= 2 * need + (1u << 20)) {
                ptr = g_scratch_free[best].ptr;
                bytes = g_scratch_free[best].bytes;
                g_scratch_cached -= bytes;
                g_scratch_free.erase(g_scratch_free.begin() + best);
                return VA_OK;
            }
        }
        hipError_t e = hipMalloc(&ptr, need);
        if (e != hipSuccess) {                       // make room: drop what is cached, once
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
            {
                std::lock_guard<std::mutex> lock(g_scratch_mu);
                scratch_release_all_locked();
            }
            e = hipMalloc(&ptr, need);
        }
        if (e != hipSuccess) {
            ptr = nullptr;
            set_error("scratch: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
            return VA_ERR_NOMEM;
        }
        bytes = need;
        return VA_OK;
    }
    ~ScratchLease()
    {
        if (!ptr)
            return;
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (g_scratch_cached + bytes > kScratchCacheCap) {
            (void)hipStreamSynchronize(st);          // (nothing uses the blocks any more)
            (void)hipDeviceSynchronize();
            scratch_release_all_locked();
        }
        g_scratch_free.push_back(ScratchBlock{ptr, bytes, st});
        g_scratch_cached += bytes;
    }
};

// One device per process (one process per GPU): va_init records it, and every
// entry point that allocates or launches selects it for the calling thread first -- hipSetDevice
// is per thread, so a worker thread that never called va_init would otherwise run on device 0.
static int g_device = -1;
static thread_local int tl_device = -1;
int enter_device()
{
    if (g_device >= 0 && tl_device != g_device) {
        VA_HIP(hipSetDevice(g_device));
        tl_device = g_device;
    }
    return VA_OK;
}
#define VA_ENTER()                   \
    do {                             \
        int _rc = va::enter_device(); \
        if (_rc)                     \
            return _rc;              \
    } while (0)

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) & ~(a - 1); }

}  // namespace va

using namespace va;

struct va_pipeline {
    va_config cfg;
    size_t px;         // elements per frame (H*W*C)
    size_t frame_px;   // pixels per frame (H*W)
    int w32;
    void *bg_state;
    void *bg_state_alt;   // fused float32 EMA: the kernel writes the new state here, then the two swap
    bool bg_in_u8_range;  // running mean: every state value is known to lie in [0, 255]
    size_t bg_bytes;
    int64_t n_seen;
    double *bg_recip;  // per-frame reciprocals of the running mean's divisor
    void *diff;        // background-subtracted frames (cfg.dtype)
    void *blur;        // blurred frames when the caller does not ask for them (generic path)
    void *gscratch;    // generic Gaussian scratch (u16 / f32) or the planes of the re-laid-out u8 blur
    int planes_wp;     // > 0: u8 blur through reflected-padded single-channel planes of this width
    // Mask ping-pong and labelling workspace exist twice ("slots") once va_pipeline_overlap is on: the
    // paint pass of batch k reads slot k % 2 on the side stream while the stages of batch k + 1 fill the
    // other one.  slot 1 is allocated by va_pipeline_overlap.
    uint32_t *bits[2][2];
    void *ccl_ws[2];
    size_t ccl_ws_bytes;
    size_t bits_bytes;
    bool overlap;
    int slot;                       // slot of the next overlapped run
    hipStream_t side;               // paint passes of overlapped runs
    int paint_grid;                 // workgroups of the persistent paint pass of overlapped runs (0: one per row block)
    hipEvent_t ev_front;            // labelling of the current batch done (main stream)
    hipEvent_t ev_paint[2];         // paint pass that read slot s done (side stream)
    bool paint_pending[2];          // ev_paint[s] has been recorded and may still be running
    const char *paint_lo[2], *paint_hi[2];     // label image written by that paint pass
    const char *pstat_lo[2], *pstat_hi[2];     // statistics written by that paint pass
    int32_t *labels_scratch;
    int32_t *counts_scratch;
    TapsQ8 tq;
    TapsF32 tf;
    RowSpans se[VA_MAX_MORPH_OPS];
    bool fused;   // single-launch blur(+threshold+bits): MFMA or LDS/VALU kernel
    bool mfma;
    bool f32_fused;   // float32: [EMA + row pass] and marching column pass (va_gauss_f32_fused.hip)
    char desc[160];
    StageProfiler *prof;
};

// cv2.resize entry points (uint8 / float32 share everything but the kernels)
template <class T>
static int resize_any(const T *src, T *dst, int n, int src_h, int src_w, int c, int dst_h, int dst_w,
                      int interpolation, void *stream, const char *who)
{
    VA_ENTER();
    VA_REQUIRE(src && dst && src != dst, "%s: src/dst must be distinct non-NULL", who);
    VA_REQUIRE(n >= 0 && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0 && c >= 1 && c <= 4,
               "%s: bad shape (%d,%d,%d,%d) -> (%d,%d)", who, n, src_h, src_w, c, dst_h, dst_w);
    VA_REQUIRE((size_t)src_h * src_w < kMaxFramePixels && (size_t)dst_h * dst_w < kMaxFramePixels,
               "%s: frames above 2^29 pixels are not supported", who);
    VA_REQUIRE(interpolation >= VA_INTER_NEAREST && interpolation <= VA_INTER_LANCZOS4,
               "%s: interpolation %d not supported (0 nearest, 1 linear, 2 cubic, 3 area, 4 lanczos4)", who,
               interpolation);
    ScratchLease scratch;
    int rc = scratch.acquire(resize_scratch_bytes(src_h, src_w, dst_h, dst_w), as_stream(stream));
    if (rc)
        return rc;
    if constexpr (sizeof(T) == 1)
        return launch_resize_u8(src, dst, n, src_h, src_w, c, dst_h, dst_w, interpolation, scratch.ptr,
                                as_stream(stream));
    else
        return launch_resize_f32(src, dst, n, src_h, src_w, c, dst_h, dst_w, interpolation, scratch.ptr,
                                 as_stream(stream));
}

extern "C" {

// ------------------------------------------------------------------------------ runtime
const char *va_version(void) { return "videoanalysis_hip 0.1 (gfx950)"; }
const char *va_last_error(void) { return get_error(); }

int va_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int va_init(int device)
{
    int n = va_device_count();
    if (n <= 0) {
        set_error("va_init: no HIP device visible");
        return VA_ERR_NODEV;
    }
    VA_REQUIRE(device >= 0 && device < n, "va_init: device %d out of range [0,%d)", device, n);
    VA_REQUIRE(g_device < 0 || g_device == device,
               "va_init: this process already runs on device %d (one device per process)", g_device);
    VA_HIP(hipSetDevice(device));
    VA_HIP(hipFree(nullptr));  // force context creation
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess) {
        uint64_t keep = ~(uint64_t)0;   // keep freed scratch in the pool instead of unmapping it
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    g_device = device;
    tl_device = device;
    return VA_OK;
}

int va_trim(size_t keep_bytes)
{
    VA_ENTER();
    VA_REQUIRE(g_device >= 0, "va_trim: va_init has not run");
    hipMemPool_t pool;
    VA_HIP(hipDeviceGetDefaultMemPool(&pool, g_device));
    VA_HIP(hipDeviceSynchronize());
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);      // the cached scratch blocks of the stand-alone calls
        if (g_scratch_cached > keep_bytes)
            scratch_release_all_locked();
    }
    VA_HIP(hipMemPoolTrimTo(pool, keep_bytes));
    return VA_OK;
}

int va_malloc(void **dev_ptr, size_t bytes)
{
    VA_ENTER();
    VA_REQUIRE(dev_ptr, "va_malloc: NULL out pointer");
    *dev_ptr = nullptr;
    if (bytes == 0)
        return VA_OK;
    VA_HIP(hipMalloc(dev_ptr, bytes));
    return VA_OK;
}
int va_free(void *dev_ptr)
{
    VA_ENTER();
    if (dev_ptr)
        VA_HIP(hipFree(dev_ptr));
    return VA_OK;
}
int va_host_alloc(void **host_ptr, size_t bytes)
{
    VA_ENTER();
    VA_REQUIRE(host_ptr, "va_host_alloc: NULL out pointer");
    *host_ptr = nullptr;
    if (bytes == 0)
        return VA_OK;
    VA_HIP(hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    return VA_OK;
}
int va_host_free(void *host_ptr)
{
    VA_ENTER();
    if (host_ptr)
        VA_HIP(hipHostFree(host_ptr));
    return VA_OK;
}
// Host memory that is neither hipHostMalloc'ed nor registered is pageable.  Large asynchronous copies to pageable
// memory were seen to leave a span of the destination unwritten after hipStreamSynchronize (64 x 1080p float32 =
// 531 MB into a fresh NumPy array: 28 MB of zeros, tools/debug notes in DESIGN.md 13.10), so pageable transfers are
// blocking copies here -- ordered after the stream's earlier work, complete on return -- and only pinned memory is
// copied asynchronously.
static bool host_is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                  // (unregistered host memory: not an error for us)
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
int va_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    VA_ENTER();
    if (!bytes)
        return VA_OK;
    if (host_is_pinned(src)) {
        VA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    } else {
        VA_HIP(hipStreamSynchronize(as_stream(stream)));
        VA_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    }
    return VA_OK;
}
int va_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    VA_ENTER();
    if (!bytes)
        return VA_OK;
    if (host_is_pinned(dst)) {
        VA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    } else {
        VA_HIP(hipStreamSynchronize(as_stream(stream)));
        VA_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    }
    return VA_OK;
}
int va_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    VA_ENTER();
    if (bytes)
        VA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return VA_OK;
}
int va_memset(void *dst, int value, size_t bytes, void *stream)
{
    VA_ENTER();
    if (bytes)
        VA_HIP(hipMemsetAsync(dst, value, bytes, as_stream(stream)));
    return VA_OK;
}
int va_stream_sync(void *stream)
{
    VA_ENTER();
    VA_HIP(hipStreamSynchronize(as_stream(stream)));
    return VA_OK;
}

int va_stream_create(void **stream_out)
{
    VA_ENTER();
    VA_REQUIRE(stream_out, "va_stream_create: NULL argument");
    hipStream_t s;
    VA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = (void *)s;
    return VA_OK;
}
int va_stream_destroy(void *stream)
{
    VA_ENTER();
    if (stream)
        VA_HIP(hipStreamDestroy(as_stream(stream)));
    return VA_OK;
}
int va_event_create(void **event_out)
{
    VA_ENTER();
    VA_REQUIRE(event_out, "va_event_create: NULL argument");
    hipEvent_t e;
    VA_HIP(hipEventCreate(&e));
    *event_out = (void *)e;
    return VA_OK;
}
int va_event_destroy(void *event)
{
    VA_ENTER();
    if (event)
        VA_HIP(hipEventDestroy((hipEvent_t)event));
    return VA_OK;
}
int va_event_record(void *event, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(event, "va_event_record: NULL event");
    VA_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream)));
    return VA_OK;
}
int va_stream_wait_event(void *stream, void *event)
{
    VA_ENTER();
    VA_REQUIRE(event, "va_stream_wait_event: NULL event");
    VA_HIP(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)event, 0));
    return VA_OK;
}
int va_event_sync(void *event)
{
    VA_ENTER();
    VA_REQUIRE(event, "va_event_sync: NULL event");
    VA_HIP(hipEventSynchronize((hipEvent_t)event));
    return VA_OK;
}
int va_event_elapsed_ms(void *start_event, void *stop_event, float *ms_out)
{
    VA_ENTER();
    VA_REQUIRE(start_event && stop_event && ms_out, "va_event_elapsed_ms: NULL argument");
    VA_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)start_event, (hipEvent_t)stop_event));
    return VA_OK;
}

// ------------------------------------------------------------------------------ Gaussian
int va_gauss_taps_q8(double sigma, int *ksize_out, uint16_t *taps_out, int capacity)
{
    return va_gauss_taps_q8_rule(sigma, VA_TAPS_CV4, ksize_out, taps_out, capacity);
}
int va_gauss_taps_q8_rule(double sigma, int tap_rule, int *ksize_out, uint16_t *taps_out, int capacity)
{
    VA_REQUIRE(ksize_out && taps_out, "va_gauss_taps_q8: NULL argument");
    TapsQ8 t;
    int rc = gauss_taps_q8(sigma, &t.ksize, t.t, kMaxTaps, tap_rule);
    if (rc)
        return rc;
    if (t.ksize > capacity) {
        set_error("va_gauss_taps_q8: %d taps > capacity %d", t.ksize, capacity);
        return VA_ERR_RANGE;
    }
    memcpy(taps_out, t.t, sizeof(uint16_t) * t.ksize);
    *ksize_out = t.ksize;
    return VA_OK;
}
int va_gauss_taps_f32(double sigma, int *ksize_out, float *taps_out, int capacity)
{
    VA_REQUIRE(ksize_out && taps_out, "va_gauss_taps_f32: NULL argument");
    TapsF32 t;
    int rc = gauss_taps_f32(sigma, &t.ksize, t.t, kMaxTaps);
    if (rc)
        return rc;
    if (t.ksize > capacity) {
        set_error("va_gauss_taps_f32: %d taps > capacity %d", t.ksize, capacity);
        return VA_ERR_RANGE;
    }
    memcpy(taps_out, t.t, sizeof(float) * t.ksize);
    *ksize_out = t.ksize;
    return VA_OK;
}

// Frames the matrix-core Gaussian can take after a re-layout: colour frames (the channels are
// filtered independently, as OpenCV does) and widths that are not a multiple of 16 go through
// planes (frame, channel, h, wp).  A plane that is wider than the frame carries the reflected
// continuation of every row over at least the kernel's radius, so the blur of the plane is the
// blur of the frame on the first w columns.  Returns the plane width, or 0 when this path does
// not apply.
static int planes_width(int h, int w, int c, const TapsQ8 &t)
{
    const int wp = (w % 16 == 0) ? w : ((w + 16 + 15) / 16) * 16;
    if (c >= 1 && c <= 4 && wp - w < w && gauss_mfma_supported(wp, h, t))
        return wp;
    return 0;
}
static size_t planes_scratch_bytes(int n, int h, int wp, int c) { return 2 * (size_t)n * c * h * wp; }
static int blur_u8_planes(const uint8_t *src, uint8_t *dst, int n, int h, int w, int wp, int c,
                          const TapsQ8 &t, void *scratch, hipStream_t st)
{
    const size_t plane_bytes = (size_t)n * c * h * wp;                   // multiple of 16
    uint8_t *pin = (uint8_t *)scratch, *pout = pin + plane_bytes;        // both 16-byte aligned
    int rc = launch_channel_planes(src, pin, n, h, w, wp, c, true, st);
    if (rc)
        return rc;
    rc = launch_gauss_mfma_u8(pin, pout, nullptr, -1, n * c, h, wp, t, st);
    if (rc)
        return rc;
    return launch_channel_planes(pout, dst, n, h, w, wp, c, false, st);
}

int va_gaussian_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c, double sigma,
                   void *stream)
{
    return va_gaussian_u8_rule(src, dst, n, h, w, c, sigma, VA_TAPS_CV4, stream);
}

int va_gaussian_u8_rule(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c, double sigma,
                        int tap_rule, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_gaussian_u8: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && dst && src != dst, "va_gaussian_u8: src/dst must be distinct non-NULL");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0, "va_gaussian_u8: bad shape (%d,%d,%d,%d)", n, h,
               w, c);
    TapsQ8 t;
    int rc = gauss_taps_q8(sigma, &t.ksize, t.t, kMaxTaps, tap_rule);
    if (rc)
        return rc;
    if (c == 1 && reinterpret_cast<uintptr_t>(src) % 16 == 0 &&
        reinterpret_cast<uintptr_t>(dst) % 4 == 0) {
        if (gauss_mfma_supported(w, h, t))
            return launch_gauss_mfma_u8(src, dst, nullptr, -1, n, h, w, t, as_stream(stream));
        if (gauss_fused_supported(w, h, t))
            return launch_gauss_fused_u8(src, dst, nullptr, -1, n, h, w, t, as_stream(stream));
    }
    ScratchLease scratch;
    if (const int wp = planes_width(h, w, c, t)) {
        rc = scratch.acquire(planes_scratch_bytes(n, h, wp, c), as_stream(stream));
        if (rc)
            return rc;
        return blur_u8_planes(src, dst, n, h, w, wp, c, t, scratch.ptr, as_stream(stream));
    }
    rc = scratch.acquire(gauss_generic_u8_scratch_bytes((size_t)n * h * w * c, t), as_stream(stream));
    if (rc)
        return rc;
    return launch_gauss_generic_u8(src, dst, scratch.ptr, n, h, w, c, t, as_stream(stream));
}

// test hook: force the generic two-pass implementation
int va_gaussian_u8_generic(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c,
                           double sigma, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_gaussian_u8_generic: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && dst && src != dst, "va_gaussian_u8_generic: bad pointers");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0, "va_gaussian_u8_generic: bad shape");
    TapsQ8 t;
    int rc = gauss_taps_q8(sigma, &t.ksize, t.t, kMaxTaps);
    if (rc)
        return rc;
    ScratchLease scratch;
    rc = scratch.acquire(gauss_generic_u8_scratch_bytes((size_t)n * h * w * c, t), as_stream(stream));
    if (rc)
        return rc;
    return launch_gauss_generic_u8(src, dst, scratch.ptr, n, h, w, c, t, as_stream(stream));
}

// test hook: force the LDS/VALU (dot4/dot2) fused implementation
int va_gaussian_u8_valu(const uint8_t *src, uint8_t *dst, int n, int h, int w, int c, double sigma,
                        void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_gaussian_u8_valu: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && dst && src != dst, "va_gaussian_u8_valu: bad pointers");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0 && c == 1, "va_gaussian_u8_valu: bad shape");
    TapsQ8 t;
    int rc = gauss_taps_q8(sigma, &t.ksize, t.t, kMaxTaps);
    if (rc)
        return rc;
    return launch_gauss_fused_u8(src, dst, nullptr, -1, n, h, w, t, as_stream(stream));
}

// test hook: pin the labelling code path of every later call in this process
int va_test_hook_labelling(int path, int lds_runs)
{
    VA_REQUIRE(path >= 0 && path <= 4 && lds_runs >= 0, "va_test_hook_labelling: bad arguments");
    ccl_test_hook(path, lds_runs);
    return VA_OK;
}

// test / measurement hook: pipelines created while this is set run their 8-bit Gaussian on the VALU
// (dot4/dot2 LDS kernel) instead of the matrix cores -- the north star's "no MFMA" form of the chain
static int g_gauss_u8_valu = 0;
int va_test_hook_gaussian_u8(int force_valu)
{
    g_gauss_u8_valu = force_valu != 0;
    return VA_OK;
}

int va_test_hook_gaussian_f32(int generic_columns)
{
    gauss_f32_test_hook(generic_columns);
    return VA_OK;
}

int va_gaussian_f32(const float *src, float *dst, int n, int h, int w, int c, double sigma,
                    void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_gaussian_f32: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && dst && src != dst, "va_gaussian_f32: src/dst must be distinct non-NULL");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0, "va_gaussian_f32: bad shape (%d,%d,%d,%d)", n,
               h, w, c);
    TapsF32 t;
    int rc = gauss_taps_f32(sigma, &t.ksize, t.t, kMaxTaps);
    if (rc)
        return rc;
    ScratchLease scratch;
    rc = scratch.acquire((size_t)n * h * w * c * sizeof(float), as_stream(stream));
    if (rc)
        return rc;
    if (gauss_f32_fused_supported(h, w, c, t) && reinterpret_cast<uintptr_t>(src) % 16 == 0)
        return launch_gauss_f32_fused(src, dst, (float *)scratch.ptr, nullptr, nullptr, 0, 0.0, n, h, w, c, t,
                                      as_stream(stream));
    if (gauss_f32_fast_supported(w, c, t))
        return launch_gauss_f32_fast(src, dst, (float *)scratch.ptr, n, h, w, c, t, as_stream(stream));
    return launch_gauss_generic_f32(src, dst, (float *)scratch.ptr, n, h, w, c, t, as_stream(stream));
}

// ------------------------------------------------------------------------------ background
int va_bg_update(int mode, int dtype, const void *frames, void *diff_out, void *state,
                 int64_t n_seen, double rate, int n, size_t px, void *stream)
{
    VA_ENTER();
    double *recip = nullptr;
    ScratchLease scratch;
    if (mode == VA_BG_MEAN && dtype == VA_U8 && n > 0) {
        int rc = scratch.acquire(bg_scratch_bytes(n), as_stream(stream));
        if (rc)
            return rc;
        recip = (double *)scratch.ptr;
    }
    return launch_bg(mode, dtype, frames, diff_out, state, n_seen, rate, n, px, as_stream(stream),
                     recip);
}
int va_welford_u8(const uint8_t *frames, double *mean, double *m2, int64_t n_seen, int n,
                  size_t px, void *stream)
{
    VA_ENTER();
    return launch_welford(frames, mean, m2, n_seen, n, px, as_stream(stream));
}

int va_mean_any(const void *frames, int dtype, double *mean, int64_t n_seen, int n, size_t px, void *stream)
{
    VA_ENTER();
    return launch_temporal_stats(frames, dtype, mean, nullptr, n_seen, n, px, as_stream(stream));
}
int va_welford_any(const void *frames, int dtype, double *mean, double *m2, int64_t n_seen, int n, size_t px,
                   void *stream)
{
    VA_ENTER();
    VA_REQUIRE(m2, "va_welford_any: NULL argument");
    return launch_temporal_stats(frames, dtype, mean, m2, n_seen, n, px, as_stream(stream));
}

// ------------------------------------------------------------------------------ pointwise
int va_time_difference_u8(const uint8_t *a, const uint8_t *b, int16_t *out, size_t count,
                          void *stream)
{
    VA_ENTER();
    return launch_time_difference(a, b, out, count, as_stream(stream));
}
int va_threshold_u8(const uint8_t *src, uint8_t *dst, size_t count, int thresh, int maxval,
                    void *stream)
{
    VA_ENTER();
    return launch_threshold_u8(src, dst, count, thresh, maxval, as_stream(stream));
}
int va_mono_mean_u8(const uint8_t *src, uint8_t *dst, size_t pixels, void *stream)
{
    VA_ENTER();
    return launch_mono_mean(src, dst, pixels, as_stream(stream));
}
int va_normalize_u8(const uint8_t *src, uint8_t *dst, size_t count, double fmin, double fmax,
                    double alpha, double tmin, void *stream)
{
    VA_ENTER();
    return launch_normalize_u8(src, dst, count, fmin, fmax, alpha, tmin, as_stream(stream));
}

int va_normalize(const void *src, int src_dtype, void *dst, int dst_dtype, size_t count, double fmin,
                 double fmax, double alpha, double tmin, void *stream)
{
    VA_ENTER();
    return launch_normalize(src, src_dtype, dst, dst_dtype, count, fmin, fmax, alpha, tmin, as_stream(stream));
}

int va_prepare_u8(const uint8_t *src, uint8_t *dst, int n, int src_h, int src_w, int src_c, int left, int top,
                  int width, int height, int mono, int normalize, double fmin, double fmax, double alpha,
                  double tmin, void *stream)
{
    VA_ENTER();
    return launch_prepare_u8(src, dst, n, src_h, src_w, src_c, left, top, width, height, mono, normalize, fmin,
                             fmax, alpha, tmin, as_stream(stream));
}

int va_gaussian_noise(void *dst, int dtype, size_t count, double mean, double stdev, uint64_t seed,
                      uint64_t first_index, void *stream)
{
    VA_ENTER();
    return launch_gaussian_noise(dst, dtype, count, mean, stdev, seed, first_index, as_stream(stream));
}

int va_rot90(const void *src, void *dst, int n, int h, int w, int elem_bytes, int k, void *stream)
{
    VA_ENTER();
    return launch_rot90(src, dst, n, h, w, elem_bytes, k, as_stream(stream));
}

// ------------------------------------------------------------------------------ morphology
int va_morph_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int op, int shape,
                int ksize, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_morph_u8: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && dst && src != dst, "va_morph_u8: src/dst must be distinct non-NULL");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0, "va_morph_u8: bad shape (%d,%d,%d)", n, h, w);
    VA_REQUIRE(op == VA_MORPH_ERODE || op == VA_MORPH_DILATE, "va_morph_u8: bad op %d", op);
    RowSpans se;
    int rc = make_row_spans(shape, ksize, &se);
    if (rc)
        return rc;
    ScratchLease scratch;
    if (shape == VA_SHAPE_RECT && ksize >= 3 && n > 0) {
        rc = scratch.acquire((size_t)n * h * w, as_stream(stream));
        if (rc)
            return rc;
    }
    return launch_morph_u8(src, dst, n, h, w, op, se, as_stream(stream), (uint8_t *)scratch.ptr);
}

// test hook: morphology on the bit-packed representation used inside the pipeline
// (mask: any non-zero byte is foreground; dst gets 0 / 255)
int va_morph_bits_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int op, int shape,
                     int ksize, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_morph_bits_u8: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && dst, "va_morph_bits_u8: NULL argument");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0, "va_morph_bits_u8: bad shape");
    RowSpans se;
    int rc = make_row_spans(shape, ksize, &se);
    if (rc)
        return rc;
    size_t words = align_up((size_t)n * h * words_per_row(w) * sizeof(uint32_t));
    hipStream_t st = as_stream(stream);
    ScratchLease scratch;
    rc = scratch.acquire(2 * words, st);
    if (rc)
        return rc;
    uint32_t *b0 = (uint32_t *)scratch.ptr, *b1 = (uint32_t *)((char *)scratch.ptr + words);
    if ((rc = launch_pack_bits(src, b0, n, h, w, 0, st)))
        return rc;
    if ((rc = launch_morph_bits(b0, b1, n, h, w, op, se, st)))
        return rc;
    return launch_unpack_bits(b1, dst, n, h, w, 255, st);
}

// ------------------------------------------------------------------------------ labelling
size_t va_label_workspace_bytes(int n, int h, int w)
{
    if (n <= 0 || h <= 0 || w <= 0)
        return 256;
    return ccl_workspace_bytes(n, h, w);
}

int va_label_i32(const uint8_t *mask, int32_t *labels, int32_t *counts, int n, int h, int w,
                 int connectivity, void *workspace, size_t workspace_bytes, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_label_i32: frames above 2^29 pixels are not supported");
    VA_REQUIRE(mask && labels && counts && workspace, "va_label_i32: NULL argument");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0, "va_label_i32: bad shape (%d,%d,%d)", n, h, w);
    VA_REQUIRE(workspace_bytes >= va_label_workspace_bytes(n, h, w),
               "va_label_i32: workspace of %zu bytes < required %zu", workspace_bytes,
               va_label_workspace_bytes(n, h, w));
    if (n == 0)
        return VA_OK;
    hipStream_t st = as_stream(stream);
    size_t bits_bytes = align_up((size_t)n * h * words_per_row(w) * sizeof(uint32_t));
    uint32_t *bits = (uint32_t *)workspace;
    int rc = launch_pack_bits(mask, bits, n, h, w, 0, st);
    if (rc)
        return rc;
    return launch_ccl(bits, labels, counts, n, h, w, connectivity, (char *)workspace + bits_bytes,
                      workspace_bytes - bits_bytes, nullptr, 0, st);
}

int va_moments_i64(const int32_t *labels, int n, int h, int w, int max_labels, int64_t *stats,
                   void *stream)
{
    VA_ENTER();
    VA_REQUIRE(n >= 0 && h > 0 && w > 0, "va_moments_i64: bad shape (%d,%d,%d)", n, h, w);
    return launch_stats_from_labels(labels, n, h, w, max_labels, stats, as_stream(stream));
}

int va_largest_region(const int32_t *labels, const int32_t *counts, const int64_t *stats, int n,
                      int h, int w, int max_labels, int32_t *largest, int64_t *largest_area,
                      uint8_t *mask_out, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(n >= 0 && h > 0 && w > 0, "va_largest_region: bad shape (%d,%d,%d)", n, h, w);
    return launch_largest_region(labels, counts, stats, n, h, w, max_labels, largest, largest_area,
                                 mask_out, as_stream(stream));
}

// ------------------------------------------------------------------------------ stencils
int va_detect_peaks_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int include_plateaus,
                       void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_detect_peaks_u8: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && dst && src != dst, "va_detect_peaks_u8: bad pointers");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0, "va_detect_peaks_u8: bad shape");
    return launch_detect_peaks(src, dst, n, h, w, include_plateaus, as_stream(stream));
}

int va_detect_peaks_f32(const float *src, uint8_t *dst, int n, int h, int w, int include_plateaus, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_detect_peaks_f32: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && dst && (const void *)src != (const void *)dst, "va_detect_peaks_f32: bad pointers");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0, "va_detect_peaks_f32: bad shape");
    return launch_detect_peaks_f32(src, dst, n, h, w, include_plateaus, as_stream(stream));
}

int va_image_statistics_f32(const float *src, double *mean_out, double *var_out, int n, int h, int w, int kernel,
                            int ksize, double prior, int exclude_center, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_image_statistics_f32: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && mean_out, "va_image_statistics_f32: NULL argument");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0 && ksize >= 0, "va_image_statistics_f32: bad shape");
    VA_REQUIRE(kernel == 0 || kernel == 1, "va_image_statistics_f32: kernel must be 0 (box) or 1 (ellipse)");
    RowSpans se;
    int rc = make_row_spans(kernel == 0 ? VA_SHAPE_RECT : VA_SHAPE_ELLIPSE, 2 * ksize + 1, &se);
    if (rc)
        return rc;
    return launch_image_statistics_f32(src, mean_out, var_out, n, h, w, se, prior, exclude_center,
                                       as_stream(stream));
}

int va_mask_thinning_u8(uint8_t *img, uint8_t *scratch, uint8_t *skel, int h, int w,
                        int *iterations_out, void *stream)
{
    VA_REQUIRE(img && scratch && skel, "va_mask_thinning_u8: NULL argument");
    VA_REQUIRE(h > 0 && w > 0, "va_mask_thinning_u8: bad shape");
    VA_ENTER();
    hipStream_t st = as_stream(stream);
    // a 3x3-cross erosion empties any mask within min(h,w)/2 + 1 steps, except one that fills the
    // frame (the border never wins): the reference would loop forever there, we stop
    const int max_it = (h < w ? h : w) / 2 + 2;
    // One flag per iteration (set when a pixel survived the erosion); the exit condition is read back once per
    // kCheck iterations (steps enqueued past the emptying one see an empty image and change
    // nothing: eroded = temp = 0, skeleton |= 0), so the host waits ceil(iterations / kCheck)
    // times instead of once per step.
    const int kCheck = 16;
    const int total_it = max_it + 1;
    ScratchLease cnt;
    int rc = cnt.acquire(sizeof(unsigned long long) * (size_t)total_it, st);
    if (rc)
        return rc;
    unsigned long long *cnt_dev = (unsigned long long *)cnt.ptr;
    VA_HIP(hipMemsetAsync(cnt_dev, 0, sizeof(unsigned long long) * (size_t)total_it, st));
    VA_HIP(hipMemsetAsync(skel, 0, (size_t)h * w, st));
    uint8_t *cur = img, *nxt = scratch;
    unsigned long long host_cnt[kCheck];
    int done_it = -1;
    for (int it0 = 0; it0 < total_it && done_it < 0; it0 += kCheck) {
        const int k = (total_it - it0 < kCheck) ? total_it - it0 : kCheck;
        for (int j = 0; j < k; j++) {
            rc = launch_thinning_step(cur, nxt, skel, 1, h, w, cnt_dev + it0 + j, st);
            if (rc)
                return rc;
            uint8_t *t = cur;
            cur = nxt;
            nxt = t;
        }
        VA_HIP(hipStreamSynchronize(st));
        VA_HIP(hipMemcpy(host_cnt, cnt_dev + it0, sizeof(unsigned long long) * (size_t)k, hipMemcpyDeviceToHost));
        for (int j = 0; j < k; j++)
            if (host_cnt[j] == 0) {
                done_it = it0 + j;
                break;
            }
    }
    if (done_it < 0)
        done_it = max_it;
    if (iterations_out)
        *iterations_out = done_it + 1;
    return VA_OK;
}

int va_image_statistics_u8(const uint8_t *src, double *mean_out, double *var_out, int n, int h,
                           int w, int kernel, int ksize, double prior, int exclude_center,
                           void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_image_statistics_u8: frames above 2^29 pixels are not supported");
    VA_REQUIRE(src && mean_out, "va_image_statistics_u8: NULL argument");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0 && ksize >= 0, "va_image_statistics_u8: bad shape");
    VA_REQUIRE(kernel == 0 || kernel == 1, "va_image_statistics_u8: kernel must be 0 (box) or 1 (ellipse)");
    RowSpans se;
    int rc = make_row_spans(kernel == 0 ? VA_SHAPE_RECT : VA_SHAPE_ELLIPSE, 2 * ksize + 1, &se);
    if (rc)
        return rc;
    ScratchLease scratch;
    rc = scratch.acquire(image_statistics_scratch_bytes(n, h, w), as_stream(stream));
    if (rc)
        return rc;
    return launch_image_statistics(src, mean_out, var_out, n, h, w, se, prior, exclude_center,
                                   scratch.ptr, as_stream(stream));
}

// ------------------------------------------------------------------------------ contour
size_t va_contour_workspace_bytes(int n, int h, int w)
{
    if (n <= 0 || h <= 0 || w <= 0)
        return 256;
    return ccl_workspace_bytes(n, h, w) + align_up((size_t)n * h * w * sizeof(int32_t)) +
           align_up((size_t)n * sizeof(unsigned long long));
}

int va_largest_contour(const uint8_t *mask, int n, int h, int w, int32_t *points, int max_points,
                       int32_t *npoints, double *area, int32_t *ncomponents, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(h <= 0 || w <= 0 || (size_t)h * (size_t)w < kMaxFramePixels,
               "va_largest_contour: frames above 2^29 pixels are not supported");
    VA_REQUIRE(mask && points && npoints && ncomponents && workspace, "va_largest_contour: NULL argument");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0 && max_points > 0, "va_largest_contour: bad shape");
    VA_REQUIRE(workspace_bytes >= va_contour_workspace_bytes(n, h, w),
               "va_largest_contour: workspace of %zu bytes < required %zu", workspace_bytes,
               va_contour_workspace_bytes(n, h, w));
    if (n == 0)
        return VA_OK;
    hipStream_t st = as_stream(stream);
    char *ws = (char *)workspace;
    const size_t bits_bytes = align_up((size_t)n * h * words_per_row(w) * sizeof(uint32_t));
    const size_t rows_bytes = ccl_workspace_bytes(n, h, w) - bits_bytes;
    uint32_t *bits = (uint32_t *)ws;
    void *rows = ws + bits_bytes;
    int32_t *forest = (int32_t *)(ws + bits_bytes + rows_bytes);
    unsigned long long *keys =
        (unsigned long long *)((char *)forest + align_up((size_t)n * h * w * sizeof(int32_t)));
    int rc = launch_pack_bits(mask, bits, n, h, w, 0, st);
    if (rc)
        return rc;
    rc = launch_ccl(bits, forest, ncomponents, n, h, w, 8, rows, rows_bytes, nullptr, 0, st, nullptr,
                    /*paint=*/false);
    if (rc)
        return rc;
    return launch_largest_contour(bits, forest, n, h, w, keys, points, max_points, npoints, area, st);
}

int va_resize_u8(const uint8_t *src, uint8_t *dst, int n, int src_h, int src_w, int c, int dst_h, int dst_w,
                 int interpolation, void *stream)
{
    return resize_any<uint8_t>(src, dst, n, src_h, src_w, c, dst_h, dst_w, interpolation, stream, "va_resize_u8");
}

int va_resize_f32(const float *src, float *dst, int n, int src_h, int src_w, int c, int dst_h, int dst_w,
                  int interpolation, void *stream)
{
    return resize_any<float>(src, dst, n, src_h, src_w, c, dst_h, dst_w, interpolation, stream, "va_resize_f32");
}

int va_contour_moments(const void *points, const int32_t *npoints, int n, int max_points,
                       int is_float, double *moments_out, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(points && moments_out, "va_contour_moments: NULL argument");
    VA_REQUIRE(n >= 0 && max_points > 0, "va_contour_moments: bad shape (%d contours, %d points)", n,
               max_points);
    return launch_contour_moments(points, npoints, n, max_points, is_float, moments_out,
                                  as_stream(stream));
}

// ------------------------------------------------------------------------------ pipeline
static int pipeline_free(va_pipeline *p)
{
    void *ptrs[] = {p->bg_state, p->bg_state_alt, p->bg_recip, p->diff, p->blur, p->gscratch,
                    p->bits[0][0], p->bits[0][1], p->bits[1][0], p->bits[1][1],
                    p->ccl_ws[0], p->ccl_ws[1], p->labels_scratch, p->counts_scratch};
    for (void *q : ptrs)
        if (q)
            (void)hipFree(q);
    if (p->side)
        (void)hipStreamDestroy(p->side);
    if (p->ev_front)
        (void)hipEventDestroy(p->ev_front);
    for (hipEvent_t e : p->ev_paint)
        if (e)
            (void)hipEventDestroy(e);
    if (p->prof) {
        for (int i = 0; i < StageProfiler::kMaxMarks; i++)
            if (p->prof->created[i])
                (void)hipEventDestroy(p->prof->ev[i]);
        delete p->prof;
    }
    delete p;
    return VA_OK;
}

int va_pipeline_create(const va_config *cfg, va_pipeline_t **out)
{
    VA_ENTER();
    VA_REQUIRE(cfg && out, "va_pipeline_create: NULL argument");
    *out = nullptr;
    VA_REQUIRE(cfg->struct_size == (int32_t)sizeof(va_config),
               "va_pipeline_create: va_config.struct_size %d != %zu (ABI mismatch)",
               cfg->struct_size, sizeof(va_config));
    VA_REQUIRE(cfg->width > 0 && cfg->height > 0 && (cfg->channels == 1 || cfg->channels == 3),
               "va_pipeline_create: bad frame format %dx%dx%d", cfg->width, cfg->height,
               cfg->channels);
    VA_REQUIRE(cfg->dtype == VA_U8 || cfg->dtype == VA_F32, "va_pipeline_create: bad dtype %d",
               cfg->dtype);
    VA_REQUIRE(cfg->max_batch > 0, "va_pipeline_create: max_batch must be > 0");
    VA_REQUIRE(cfg->bg_mode >= VA_BG_NONE && cfg->bg_mode <= VA_BG_STATIC,
               "va_pipeline_create: bad bg_mode %d", cfg->bg_mode);
    VA_REQUIRE(cfg->morph_count >= 0 && cfg->morph_count <= VA_MAX_MORPH_OPS,
               "va_pipeline_create: morph_count %d out of range", cfg->morph_count);
    VA_REQUIRE(cfg->connectivity == 0 || cfg->connectivity == 4 || cfg->connectivity == 8,
               "va_pipeline_create: connectivity must be 0, 4 or 8");
    VA_REQUIRE(cfg->tap_rule == VA_TAPS_CV4 || cfg->tap_rule == VA_TAPS_CV3,
               "va_pipeline_create: tap_rule must be VA_TAPS_CV4 or VA_TAPS_CV3");
    const bool masks = cfg->thresh >= 0;
    if (cfg->dtype == VA_F32) {
        VA_REQUIRE(!masks, "va_pipeline_create: threshold/labelling need uint8 frames");
        VA_REQUIRE(cfg->bg_mode == VA_BG_NONE || cfg->bg_mode == VA_BG_EMA,
                   "va_pipeline_create: float32 frames support bg_mode NONE/EMA only");
    }
    if (masks)
        VA_REQUIRE(cfg->channels == 1, "va_pipeline_create: threshold/labelling need 1 channel");
    else
        VA_REQUIRE(cfg->morph_count == 0 && cfg->connectivity == 0,
                   "va_pipeline_create: morphology/labelling need a threshold (thresh >= 0)");
    VA_REQUIRE((size_t)cfg->width * cfg->height < kMaxFramePixels,
               "va_pipeline_create: frames of %dx%d exceed the supported 2^29 pixels (the kernels'"
               " 32-bit buffer descriptors address h*w*4 bytes)", cfg->width, cfg->height);

    va_pipeline *p = new (std::nothrow) va_pipeline();
    if (!p) {
        set_error("va_pipeline_create: out of host memory");
        return VA_ERR_NOMEM;
    }
    memset(p, 0, sizeof(*p));
    p->cfg = *cfg;
    if (p->cfg.maxval <= 0 || p->cfg.maxval > 255)
        p->cfg.maxval = 255;
    p->frame_px = (size_t)cfg->width * cfg->height;
    p->px = p->frame_px * cfg->channels;
    p->w32 = words_per_row(cfg->width);
    const size_t esz = cfg->dtype == VA_U8 ? 1 : 4;
    const size_t nb = (size_t)cfg->max_batch;
    int rc = VA_OK;
#define PIPE_TRY(expr)                    \
    do {                                  \
        rc = (expr);                      \
        if (rc) {                         \
            pipeline_free(p);             \
            return rc;                    \
        }                                 \
    } while (0)
#define PIPE_MALLOC(ptr, bytes)                                                          \
    do {                                                                                 \
        hipError_t _e = hipMalloc((void **)&(ptr), (bytes));                             \
        if (_e != hipSuccess) {                                                          \
            set_error("va_pipeline_create: hipMalloc(%zu) failed: %s", (size_t)(bytes),  \
                      hipGetErrorString(_e));                                            \
            pipeline_free(p);                                                            \
            return VA_ERR_NOMEM;                                                         \
        }                                                                                \
    } while (0)

    if (cfg->sigma > 0) {
        if (cfg->dtype == VA_U8) {
            PIPE_TRY(gauss_taps_q8(cfg->sigma, &p->tq.ksize, p->tq.t, kMaxTaps, cfg->tap_rule));
            p->mfma = cfg->channels == 1 && !g_gauss_u8_valu && gauss_mfma_supported(cfg->width, cfg->height, p->tq);
            p->fused = p->mfma ||
                       (cfg->channels == 1 && gauss_fused_supported(cfg->width, cfg->height, p->tq));
        } else {
            PIPE_TRY(gauss_taps_f32(cfg->sigma, &p->tf.ksize, p->tf.t, kMaxTaps));
            p->f32_fused = gauss_f32_fused_supported(cfg->height, cfg->width, cfg->channels, p->tf);
        }
    }
    for (int i = 0; i < cfg->morph_count; i++) {
        if (cfg->morph_op[i] != VA_MORPH_ERODE && cfg->morph_op[i] != VA_MORPH_DILATE) {
            set_error("va_pipeline_create: bad morph_op[%d]=%d", i, cfg->morph_op[i]);
            pipeline_free(p);
            return VA_ERR_INVALID;
        }
        PIPE_TRY(make_row_spans(cfg->morph_shape[i], cfg->morph_ksize[i], &p->se[i]));
    }
    if (cfg->bg_mode != VA_BG_NONE) {
        p->bg_bytes = p->px * (cfg->bg_mode == VA_BG_EMA ? sizeof(float) : sizeof(double));
        PIPE_MALLOC(p->bg_state, p->bg_bytes);
        if (p->f32_fused && cfg->bg_mode == VA_BG_EMA)
            PIPE_MALLOC(p->bg_state_alt, p->bg_bytes);
        p->bg_in_u8_range = true;                        // (zeros)
        hipError_t e = hipMemset(p->bg_state, 0, p->bg_bytes);
        if (e != hipSuccess) {
            set_error("va_pipeline_create: hipMemset failed: %s", hipGetErrorString(e));
            pipeline_free(p);
            return VA_ERR_HIP;
        }
        if (!p->f32_fused)      // (the fused float path never materialises the difference image)
            PIPE_MALLOC(p->diff, nb * p->px * esz);
        if (cfg->bg_mode == VA_BG_MEAN)
            PIPE_MALLOC(p->bg_recip, bg_scratch_bytes(cfg->max_batch));
    }
    if (cfg->sigma > 0 && !p->fused) {
        size_t gs = cfg->dtype == VA_U8 ? gauss_generic_u8_scratch_bytes(nb * p->px, p->tq) : nb * p->px * 4;
        if (cfg->dtype == VA_U8) {
            p->planes_wp = planes_width(cfg->height, cfg->width, cfg->channels, p->tq);
            if (p->planes_wp)
                gs = planes_scratch_bytes(cfg->max_batch, cfg->height, p->planes_wp, cfg->channels);
        }
        PIPE_MALLOC(p->gscratch, gs);
        PIPE_MALLOC(p->blur, nb * p->px * esz);
    }
    if (masks) {
        size_t bb = align_up(nb * cfg->height * p->w32 * sizeof(uint32_t));
        p->bits_bytes = bb;
        PIPE_MALLOC(p->bits[0][0], bb);
        PIPE_MALLOC(p->bits[0][1], bb);
        if (cfg->connectivity) {
            p->ccl_ws_bytes = ccl_rows_workspace_bytes(cfg->max_batch, cfg->height);
            PIPE_MALLOC(p->ccl_ws[0], p->ccl_ws_bytes);
            PIPE_MALLOC(p->counts_scratch, align_up(nb * sizeof(int32_t)));
            // forest / label scratch for runs that do not ask for the label image (counts only:
            // sparse forest words, never painted; stats only: painted here).  Allocated now so
            // that an out-of-memory shows at create time and no run ever calls hipMalloc.
            PIPE_MALLOC(p->labels_scratch, nb * p->frame_px * sizeof(int32_t));
        }
    }
    snprintf(p->desc, sizeof(p->desc), "bg=%d gauss=%s(ksize=%d) thresh=%d morph=%d ccl=%d",
             cfg->bg_mode,
             cfg->sigma > 0 ? (p->fused ? (p->mfma ? "mfma-i8" : "fused-lds")
                                        : (cfg->dtype == VA_F32 && p->f32_fused
                                               ? "f32-ema-row+col-march"
                                               : cfg->dtype == VA_F32 &&
                                                   gauss_f32_fast_supported(cfg->width, cfg->channels, p->tf)
                                               ? "f32-packed"
                                               : (p->planes_wp ? "mfma-i8-planes" : "generic")))
                            : "none",
             cfg->sigma > 0 ? (cfg->dtype == VA_U8 ? p->tq.ksize : p->tf.ksize) : 0, cfg->thresh,
             cfg->morph_count, cfg->connectivity);
#undef PIPE_TRY
#undef PIPE_MALLOC
    *out = p;
    return VA_OK;
}

int va_pipeline_destroy(va_pipeline_t *p)
{
    VA_ENTER();
    if (!p)
        return VA_OK;
    (void)hipDeviceSynchronize();
    return pipeline_free(p);
}

const char *va_pipeline_describe(const va_pipeline_t *p) { return p ? p->desc : ""; }

int va_pipeline_overlap(va_pipeline_t *p, int enable)
{
    VA_ENTER();
    VA_REQUIRE(p, "va_pipeline_overlap: NULL pipeline");
    VA_HIP(hipDeviceSynchronize());                  // no paint pass in flight across the switch
    p->paint_pending[0] = p->paint_pending[1] = false;
    p->slot = 0;
    if (!enable) {
        p->overlap = false;
        return VA_OK;
    }
    VA_REQUIRE(p->cfg.thresh >= 0 && p->cfg.connectivity,
               "va_pipeline_overlap: only a labelling pipeline has a paint pass to overlap");
#define OV_MALLOC(ptr, bytes)                                                               \
    do {                                                                                    \
        if (!(ptr)) {                                                                       \
            hipError_t _e = hipMalloc((void **)&(ptr), (bytes));                            \
            if (_e != hipSuccess) {                                                         \
                (ptr) = nullptr;                                                            \
                set_error("va_pipeline_overlap: hipMalloc(%zu) failed: %s", (size_t)(bytes), \
                          hipGetErrorString(_e));                                           \
                return VA_ERR_NOMEM;                                                        \
            }                                                                               \
        }                                                                                   \
    } while (0)
    OV_MALLOC(p->bits[1][0], p->bits_bytes);
    OV_MALLOC(p->bits[1][1], p->bits_bytes);
    OV_MALLOC(p->ccl_ws[1], p->ccl_ws_bytes);
#undef OV_MALLOC
    if (!p->side) {
        // lowest priority: the dispatcher places the workgroups of the caller's stream (VALU-/latency-
        // bound kernels with large register and LDS footprints) first, and the paint pass -- tens of
        // thousands of small store-only workgroups that would otherwise take every slot that frees
        // up -- fills what is left of the CUs and of the HBM bandwidth.  enable == 2: default priority
        // (kept for A/B measurements).
        int lo = 0, hi = 0;
        VA_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));        // lo = numerically greatest = least urgent
        VA_HIP(hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, enable == 2 ? 0 : lo));
    }
    if (!p->ev_front)
        VA_HIP(hipEventCreateWithFlags(&p->ev_front, hipEventDisableTiming));
    for (int s = 0; s < 2; s++)
        if (!p->ev_paint[s])
            VA_HIP(hipEventCreateWithFlags(&p->ev_paint[s], hipEventDisableTiming));
    {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g_device);
        // persistent paint pass, 4 workgroups per CU: bounded footprint beside the next batch's kernels
        // (measured best of 0 = one workgroup per row block, 1, 2, 4, 8: tools/overlap_probe.py);
        // $VA_PAINT_WGS_PER_CU overrides it for such measurements
        const char *e = getenv("VA_PAINT_WGS_PER_CU");
        const int per_cu = e ? atoi(e) : 4;
        p->paint_grid = per_cu > 0 ? ((cus * per_cu + 7) / 8) * 8 : 0;
    }
    p->overlap = true;
    return VA_OK;
}

int va_pipeline_fence(va_pipeline_t *p, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(p, "va_pipeline_fence: NULL pipeline");
    for (int s = 0; s < 2; s++)
        if (p->paint_pending[s])
            VA_HIP(hipStreamWaitEvent(as_stream(stream), p->ev_paint[s], 0));
    return VA_OK;
}

size_t va_bg_state_bytes(const va_pipeline_t *p) { return p ? p->bg_bytes : 0; }

int va_bg_get_state(va_pipeline_t *p, void *state_host, size_t bytes, int64_t *n_seen)
{
    VA_ENTER();
    VA_REQUIRE(p, "va_bg_get_state: NULL pipeline");
    if (n_seen)
        *n_seen = p->n_seen;
    if (state_host) {
        VA_REQUIRE(bytes == p->bg_bytes, "va_bg_get_state: %zu bytes given, state is %zu", bytes,
                   p->bg_bytes);
        if (bytes) {
            VA_HIP(hipDeviceSynchronize());
            VA_HIP(hipMemcpy(state_host, p->bg_state, bytes, hipMemcpyDeviceToHost));
        }
    }
    return VA_OK;
}

int va_bg_set_state(va_pipeline_t *p, const void *state_host, size_t bytes, int64_t n_seen)
{
    VA_ENTER();
    VA_REQUIRE(p, "va_bg_set_state: NULL pipeline");
    VA_REQUIRE(n_seen >= 0, "va_bg_set_state: n_seen must be >= 0");
    if (state_host) {
        VA_REQUIRE(bytes == p->bg_bytes, "va_bg_set_state: %zu bytes given, state is %zu", bytes,
                   p->bg_bytes);
        if (bytes) {
            VA_HIP(hipDeviceSynchronize());
            VA_HIP(hipMemcpy(p->bg_state, state_host, bytes, hipMemcpyHostToDevice));
        }
        p->bg_in_u8_range = false;
        if (p->cfg.bg_mode == VA_BG_MEAN) {              // (float64 state)
            const double *m = static_cast<const double *>(state_host);
            bool ok = true;
            for (size_t i = 0; i < bytes / sizeof(double) && ok; i++)
                ok = m[i] >= 0.0 && m[i] <= 255.0;       // (false for NaN)
            p->bg_in_u8_range = ok;
        }
    } else if (p->bg_bytes) {
        VA_HIP(hipDeviceSynchronize());
        VA_HIP(hipMemset(p->bg_state, 0, p->bg_bytes));
        p->bg_in_u8_range = true;
    }
    p->n_seen = n_seen;
    return VA_OK;
}

int va_pipeline_run(va_pipeline_t *p, const void *frames, int n, void *filtered_out,
                    uint8_t *mask_out, int32_t *labels_out, int32_t *counts_out,
                    int64_t *stats_out, void *stream)
{
    VA_ENTER();
    VA_REQUIRE(p && frames, "va_pipeline_run: NULL argument");
    const va_config &c = p->cfg;
    VA_REQUIRE(n >= 0 && n <= c.max_batch, "va_pipeline_run: n=%d exceeds max_batch=%d", n,
               c.max_batch);
    if (n == 0)
        return VA_OK;
    hipStream_t st = as_stream(stream);
    const bool masks = c.thresh >= 0;
    VA_REQUIRE(masks || !(mask_out || labels_out || counts_out || stats_out),
               "va_pipeline_run: mask/label outputs requested but the pipeline has no threshold");
    VA_REQUIRE(c.connectivity || !(labels_out || counts_out || stats_out),
               "va_pipeline_run: label outputs requested but connectivity == 0");
    VA_REQUIRE(!stats_out || c.max_labels > 0, "va_pipeline_run: stats_out needs max_labels > 0");
    const size_t esz = c.dtype == VA_U8 ? 1 : 4;
    int rc;
    const void *cur = frames;
    StageProfiler *prof = (p->prof && p->prof->enabled && p->prof->runs++ % p->prof->every == 0) ? p->prof : nullptr;
#define VA_MARK(nm)             \
    do {                        \
        if (prof)               \
            prof->mark(nm, st); \
    } while (0)
    VA_MARK(nullptr);

    // float32 frames: background update, difference and row pass in one kernel, then the columns
    // (16-byte aligned frame pointers: a contract of float32 pipelines, see the header)
    if (c.dtype == VA_F32 && p->f32_fused) {
        VA_REQUIRE(reinterpret_cast<uintptr_t>(frames) % 16 == 0 &&
                       (!filtered_out || reinterpret_cast<uintptr_t>(filtered_out) % 16 == 0),
                   "va_pipeline_run: float32 pipelines need 16-byte aligned frames_dev / filtered_out_dev "
                   "(hipMalloc'ed buffers and whole-frame offsets into them are)");
        void *dst = filtered_out ? filtered_out : p->blur;
        const bool ema = c.bg_mode == VA_BG_EMA;
        rc = launch_gauss_f32_fused((const float *)frames, (float *)dst, (float *)p->gscratch,
                                    ema ? (const float *)p->bg_state : nullptr,
                                    ema ? (float *)p->bg_state_alt : nullptr, p->n_seen, (double)c.bg_rate, n,
                                    c.height, c.width, c.channels, p->tf, st, prof);
        if (rc)
            return rc;
        if (ema && n > 0) {
            std::swap(p->bg_state, p->bg_state_alt);      // (stream order: later runs read what this one wrote)
            p->n_seen += n;
        }
        return VA_OK;
    }
    // 1. background subtraction (temporal, in frame order)
    if (c.bg_mode != VA_BG_NONE) {
        rc = launch_bg(c.bg_mode, c.dtype, cur, p->diff, p->bg_state, p->n_seen, (double)c.bg_rate,
                       n, p->px, st, p->bg_recip, p->bg_in_u8_range);
        if (rc)
            return rc;
        p->n_seen += n;
        cur = p->diff;
        VA_MARK("bg");
    }

    // Overlapped runs (va_pipeline_overlap): this batch's masks and run tables live in slot `slot`; the
    // paint pass that last read that slot (two runs ago) must be done before anything writes it.
    const bool want_ccl = masks && c.connectivity && (labels_out || counts_out || stats_out);
    const bool defer_paint = p->overlap && want_ccl && (labels_out != nullptr || stats_out != nullptr);
    const int slot = defer_paint ? p->slot : 0;
    if (p->overlap && p->paint_pending[slot])
        VA_HIP(hipStreamWaitEvent(st, p->ev_paint[slot], 0));
    uint32_t *const *bits = p->bits[slot];

    // 2. Gaussian blur (+ threshold + bit packing when fused)
    bool have_bits = false;
    if (c.sigma > 0) {
        if (c.dtype == VA_U8 && p->mfma && masks && mask_out && !filtered_out && c.morph_count == 0 && !want_ccl &&
            reinterpret_cast<uintptr_t>(mask_out) % 4 == 0) {
            // the chain ends at FilterThreshold's uint8 mask (BASELINE configs[1]): the Gaussian's
            // epilogue writes the 0 / maxval bytes itself -- no bit mask, no unpack pass
            rc = launch_gauss_mfma_u8((const uint8_t *)cur, mask_out, nullptr, c.thresh, n, c.height, c.width, p->tq,
                                      st, c.maxval);
            if (rc)
                return rc;
            VA_MARK("gauss_mfma_mask8");
            return VA_OK;
        }
        if (c.dtype == VA_U8 && p->fused) {
            if (p->mfma)
                rc = launch_gauss_mfma_u8((const uint8_t *)cur, (uint8_t *)filtered_out, masks ? bits[0] : nullptr,
                                          c.thresh, n, c.height, c.width, p->tq, st);
            else
                rc = launch_gauss_fused_u8((const uint8_t *)cur, (uint8_t *)filtered_out, masks ? bits[0] : nullptr,
                                           c.thresh, n, c.height, c.width, p->tq, st);
            if (rc)
                return rc;
            have_bits = masks;
            cur = filtered_out;  // may be NULL; not needed any more when have_bits
            VA_MARK(p->mfma ? "gauss_mfma" : "gauss_fused");
        } else {
            void *dst = filtered_out ? filtered_out : p->blur;
            if (c.dtype == VA_U8 && p->planes_wp)
                rc = blur_u8_planes((const uint8_t *)cur, (uint8_t *)dst, n, c.height, c.width,
                                    p->planes_wp, c.channels, p->tq, p->gscratch, st);
            else if (c.dtype == VA_U8)
                rc = launch_gauss_generic_u8((const uint8_t *)cur, (uint8_t *)dst,
                                             p->gscratch, n, c.height, c.width,
                                             c.channels, p->tq, st);
            else if (gauss_f32_fast_supported(c.width, c.channels, p->tf))
                rc = launch_gauss_f32_fast((const float *)cur, (float *)dst, (float *)p->gscratch, n,
                                           c.height, c.width, c.channels, p->tf, st);
            else
                rc = launch_gauss_generic_f32((const float *)cur, (float *)dst,
                                              (float *)p->gscratch, n, c.height, c.width,
                                              c.channels, p->tf, st);
            if (rc)
                return rc;
            cur = dst;
            VA_MARK(c.dtype == VA_U8 ? (p->planes_wp ? "gauss_planes" : "gauss_generic")
                                     : (gauss_f32_fast_supported(c.width, c.channels, p->tf) ? "gauss_f32"
                                                                                             : "gauss_generic"));
        }
    } else if (filtered_out) {
        VA_HIP(hipMemcpyAsync(filtered_out, cur, (size_t)n * p->px * esz, hipMemcpyDeviceToDevice,
                              st));
    }
    if (!masks)
        return VA_OK;

    // 3. threshold -> bit mask
    int b = 0;
    if (!have_bits) {
        rc = launch_pack_bits((const uint8_t *)cur, bits[0], n, c.height, c.width, c.thresh, st);
        if (rc)
            return rc;
        VA_MARK("threshold_pack");
    }
    // 4. morphology on bits (one fused kernel when the sequence allows it)
    int32_t *labels = nullptr;
    if (want_ccl) {
        labels = labels_out ? labels_out : p->labels_scratch;
    }
    if (c.morph_count > 0 && morph_fused_supported(c.width, p->se, c.morph_count)) {
        rc = launch_morph_fused(bits[b], bits[b ^ 1], n, c.height, c.width, c.morph_op, p->se,
                                c.morph_count, st);
        if (rc)
            return rc;
        b ^= 1;
        VA_MARK("morph_fused");
    } else {
        for (int i = 0; i < c.morph_count; i++) {
            rc = launch_morph_bits(bits[b], bits[b ^ 1], n, c.height, c.width, c.morph_op[i],
                                   p->se[i], st);
            if (rc)
                return rc;
            b ^= 1;
            VA_MARK(c.morph_op[i] == VA_MORPH_DILATE ? "morph_dilate" : "morph_erode");
        }
    }
    if (mask_out) {
        rc = launch_unpack_bits(bits[b], mask_out, n, c.height, c.width, c.maxval, st);
        if (rc)
            return rc;
        VA_MARK("mask_unpack");
    }
    // 5. labelling (+ statistics)
    if (want_ccl) {
        // counts alone come out of the labelling kernels; the label image (the chain's largest
        // write) is painted only for callers that read it or the per-label statistics
        const bool paint = labels_out != nullptr || stats_out != nullptr;
        int32_t *counts = counts_out ? counts_out : p->counts_scratch;
        // The labelling kernels may write the label image (sparse forest words: chip-wide passes,
        // frames beyond the LDS run table); a paint pass still in flight on the side stream that
        // writes the same image -- the caller reuses one label buffer for consecutive batches --
        // has to finish first.  Callers that alternate two label buffers never wait here.
        const char *lo = (const char *)labels, *hi = lo + (size_t)n * p->frame_px * sizeof(int32_t);
        const char *slo = (const char *)stats_out,
                   *shi = slo + (stats_out ? (size_t)n * c.max_labels * VA_STATS_STRIDE * sizeof(int64_t) : 0);
        for (int s = 0; s < 2 && p->overlap; s++)
            if (p->paint_pending[s] &&
                ((lo < p->paint_hi[s] && p->paint_lo[s] < hi) || (slo < p->pstat_hi[s] && p->pstat_lo[s] < shi)))
                VA_HIP(hipStreamWaitEvent(st, p->ev_paint[s], 0));
        if (!defer_paint) {
            rc = launch_ccl(bits[b], labels, counts, n, c.height, c.width, c.connectivity, p->ccl_ws[0],
                            p->ccl_ws_bytes, stats_out, c.max_labels, st, prof, paint);
            if (rc)
                return rc;
        } else {
            CclPaintPlan plan;
            rc = launch_ccl_front(bits[b], labels, counts, n, c.height, c.width, c.connectivity, p->ccl_ws[slot],
                                  p->ccl_ws_bytes, stats_out, c.max_labels, st, prof, &plan);
            if (rc)
                return rc;
            VA_HIP(hipEventRecord(p->ev_front, st));
            VA_HIP(hipStreamWaitEvent(p->side, p->ev_front, 0));
            if (prof)
                prof->mark(nullptr, p->side);            // (start of the side stream's part of this run)
            plan.persistent_grid = p->paint_grid;
            rc = launch_ccl_paint(plan, p->side, prof);
            if (rc)
                return rc;
            VA_HIP(hipEventRecord(p->ev_paint[slot], p->side));
            p->paint_pending[slot] = true;
            p->paint_lo[slot] = lo;
            p->paint_hi[slot] = hi;
            p->pstat_lo[slot] = slo;
            p->pstat_hi[slot] = shi;
            p->slot = slot ^ 1;
        }
    }
#undef VA_MARK
    return VA_OK;
}

int va_pipeline_profile(va_pipeline_t *p, int enable)
{
    VA_ENTER();
    VA_REQUIRE(p, "va_pipeline_profile: NULL pipeline");
    if (!p->prof) {
        p->prof = new (std::nothrow) StageProfiler();
        if (!p->prof) {
            set_error("va_pipeline_profile: out of host memory");
            return VA_ERR_NOMEM;
        }
    }
    p->prof->enabled = enable != 0;
    p->prof->every = enable > 1 ? enable : 1;
    p->prof->runs = 0;
    p->prof->n = 0;
    p->prof->dropped = 0;
    return VA_OK;
}

int va_pipeline_stage_times(va_pipeline_t *p, int capacity, char *names, double *total_ms,
                            int32_t *launches, int *nstages_out)
{
    VA_ENTER();
    VA_REQUIRE(p && names && total_ms && launches && nstages_out && capacity > 0,
               "va_pipeline_stage_times: bad argument");
    *nstages_out = 0;
    if (!p->prof || p->prof->n == 0)
        return VA_OK;
    StageProfiler &pr = *p->prof;
    for (int i = 0; i < pr.n; i++)                  // (marks of overlapped runs sit on two streams)
        VA_HIP(hipEventSynchronize(pr.ev[i]));
    int ns = 0;
    for (int i = 1; i < pr.n; i++) {
        if (!pr.name[i])
            continue;  // start-of-run marker
        float ms = 0.f;
        VA_HIP(hipEventElapsedTime(&ms, pr.ev[i - 1], pr.ev[i]));
        int k = 0;
        for (; k < ns; k++)
            if (strncmp(names + (size_t)k * 32, pr.name[i], 31) == 0)
                break;
        if (k == ns) {
            if (ns == capacity)
                continue;
            strncpy(names + (size_t)k * 32, pr.name[i], 31);
            names[(size_t)k * 32 + 31] = 0;
            total_ms[k] = 0;
            launches[k] = 0;
            ns++;
        }
        total_ms[k] += ms;
        launches[k] += 1;
    }
    *nstages_out = ns;
    return VA_OK;
}

// ------------------------------------------------------------------------------ RCCL (lazy)
struct NcclId {
    char internal[128];
};
namespace {
struct NcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, NcclId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
}  // namespace
static NcclApi g_nccl;

static int nccl_load()
{
    if (g_nccl.handle)
        return VA_OK;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)))
            break;
    if (!h) {
        set_error("RCCL not found: %s", dlerror());
        return VA_ERR_NODEV;
    }
    g_nccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    g_nccl.CommInitRank = (int (*)(void **, int, NcclId, int))dlsym(h, "ncclCommInitRank");
    g_nccl.AllGather =
        (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    g_nccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    g_nccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_nccl.GetUniqueId || !g_nccl.CommInitRank || !g_nccl.AllGather || !g_nccl.CommDestroy) {
        set_error("RCCL symbols missing in librccl.so");
        dlclose(h);
        return VA_ERR_NODEV;
    }
    g_nccl.handle = h;
    return VA_OK;
}
#define VA_NCCL(call)                                                                       \
    do {                                                                                    \
        int _r = (call);                                                                    \
        if (_r != 0) {                                                                      \
            set_error("%s failed: %s", #call,                                               \
                      g_nccl.GetErrorString ? g_nccl.GetErrorString(_r) : "rccl error");    \
            return VA_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

int va_comm_unique_id(uint8_t id_out[128])
{
    VA_ENTER();
    VA_REQUIRE(id_out, "va_comm_unique_id: NULL argument");
    int rc = nccl_load();
    if (rc)
        return rc;
    NcclId id;
    VA_NCCL(g_nccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, 128);
    return VA_OK;
}

int va_comm_init(void **comm_out, int world_size, int rank, const uint8_t id[128])
{
    VA_ENTER();
    VA_REQUIRE(comm_out && id, "va_comm_init: NULL argument");
    VA_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "va_comm_init: bad rank %d/%d",
               rank, world_size);
    int rc = nccl_load();
    if (rc)
        return rc;
    NcclId nid;
    memcpy(nid.internal, id, 128);
    VA_NCCL(g_nccl.CommInitRank(comm_out, world_size, nid, rank));
    return VA_OK;
}

int va_gather_counts(void *comm, const int32_t *send, int32_t *recv, int count_per_rank,
                     void *stream)
{
    VA_ENTER();
    VA_REQUIRE(comm && send && recv && count_per_rank >= 0, "va_gather_counts: bad argument");
    int rc = nccl_load();
    if (rc)
        return rc;
    VA_NCCL(g_nccl.AllGather(send, recv, (size_t)count_per_rank, /*ncclInt32*/ 2, comm,
                             as_stream(stream)));
    return VA_OK;
}

int va_comm_destroy(void *comm)
{
    VA_ENTER();
    if (!comm)
        return VA_OK;
    int rc = nccl_load();
    if (rc)
        return rc;
    VA_NCCL(g_nccl.CommDestroy(comm));
    return VA_OK;
}

}  // extern "C"
