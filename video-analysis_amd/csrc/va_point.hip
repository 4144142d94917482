// va_point.hip -- temporal background models and pointwise filters (HBM-streaming kernels)
//
// A2 measure_mean / measure_mean_std (video/analysis/video.py:26-55) + BUILD-DEFINED
//    FilterBackground; A3 FilterTimeDifference (video/filters.py:564-568); A4 FilterThreshold;
//    A5 FilterMonochrome / FilterNormalize (video/filters.py:101-135, 359-374).
//
// All of these are one-touch streams: every input byte is read once with 8-16 B per lane and
// every output byte written once.  The background recurrence is sequential in time and
// parallel over pixels, so one thread owns 8 consecutive pixels, keeps their float64 state in
// registers and walks the frames of the batch in order (state is read and written once per
// batch, not once per frame).  Built with -ffp-contract=off: the float64 expression
// mean*n/(n+1) + frame/(n+1) is evaluated with exactly NumPy's roundings (no fma).
#include <math.h>

#include "va_common.h"

namespace va {

namespace {

constexpr int kBlock = 256;
constexpr int kVec = 8;  // pixels per thread in the vector path (one 8-byte load per frame)

__device__ __forceinline__ uint8_t sat_u8_trunc(double d)
{
    // d >= 0: C truncation == trunc(); saturate on the integer side (one v_min instead of an
    // f64 compare + two selects).  Inputs beyond int range saturate in v_cvt_i32_f64 itself.
    const int v = (int)d;
    return (uint8_t)(v > 255 ? 255 : v);
}

// ---- cumulative mean, float64 state ------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(kBlock)
bg_mean_u8_kernel(const uint8_t *__restrict__ frames, uint8_t *__restrict__ diff,
                  double *__restrict__ mean, long long n_seen, int n, size_t px)
{
    size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * V;
    if (i0 >= px)
        return;
    double m[V];
#pragma unroll
    for (int k = 0; k < V; k++)
        m[k] = mean[i0 + k];
    for (int f = 0; f < n; f++) {
        uint8_t p[V];
        if (V == 8) {
            uint2 v = *reinterpret_cast<const uint2 *>(frames + (size_t)f * px + i0);
            memcpy(p, &v, 8);
        } else {
            p[0] = frames[(size_t)f * px + i0];
        }
        const double dn = (double)(n_seen + f), dn1 = (double)(n_seen + f + 1);
        uint8_t o[V];
#pragma unroll
        for (int k = 0; k < V; k++) {
            double fr = (double)p[k];
            o[k] = sat_u8_trunc(fabs(fr - m[k]));
            m[k] = m[k] * dn / dn1 + fr / dn1;
        }
        if (diff) {
            if (V == 8) {
                uint2 v;
                memcpy(&v, o, 8);
                *reinterpret_cast<uint2 *>(diff + (size_t)f * px + i0) = v;
            } else {
                diff[(size_t)f * px + i0] = o[0];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < V; k++)
        mean[i0 + k] = m[k];
}

// ---- cumulative mean, division-free inner loop ---------------------------------------------
// Per frame the divisor d = n+1 is the same for every pixel, so both quotients of
//     mean*n/(n+1) + frame/(n+1)
// are computed as  q0 = RN(x*y);  r = fma(-q0,d,x) (exact);  q = RN(q0 + r*y)  with y = RN(1/d)
// divided once per frame.  This is the final step of Markstein's division algorithm (IBM
// RS/6000, Itanium): with a correctly rounded reciprocal the result is the correctly rounded
// quotient, i.e. bit-identical to IEEE division (the exceptional divisors have an all-ones
// significand, impossible for an integer n+1 < 2^53).  Checked on the CPU exhaustively for
// frame/(n+1) (all 256 values x every n+1 <= 2^24, plus random n+1 < 2^45) and on 1.6e9 random
// (mean*n, n+1) pairs, and on the GPU against the oracle's plain divisions
// (tests/test_gpu_parity.py).  It replaces two ~30-instruction float64 division sequences per
// pixel by six FMA-class operations.
__device__ __forceinline__ double div_by_uniform(double x, double d, double y)
{
    const double q = x * y;
    const double r = fma(-q, d, x);
    return fma(r, y, q);
}

// y[f] = RN(1 / (n_seen + f + 1)) for the frames of one batch (one division per frame, not one
// per frame and thread)
__global__ void __launch_bounds__(kBlock)
bg_reciprocals_kernel(double *__restrict__ y, long long n_seen, int n)
{
    const int f = blockIdx.x * kBlock + threadIdx.x;
    if (f < n)
        y[f] = 1.0 / (double)(n_seen + f + 1);
}

// the reciprocals of a batch of up to 256 frames travel in the kernel arguments (divided on the host:
// IEEE division either way), which saves the launch of bg_reciprocals_kernel and its place in the queue
struct RecipArgs {
    double r[256];
    // lo[f] = 1/d - hi(r[f]) to 2^-53 relative, hi() = r[f] with its low 9 significand bits cleared: the
    // second quotient frame/(n + 1) of an 8-bit frame value then takes TWO operations, p = frame * hi (exact:
    // 8 + 44 bits) and fma(frame, lo, p) = RN(frame/d (1 + 2^-96)), which is RN(frame/d) for every d < 2^40
    // (a non-dyadic quotient stays 2^-54 / d away from every rounding midpoint; tools/verify_small_div.c checks
    // all d <= 2^24 and random d up to 2^40 exhaustively over the 256 frame values)
    double lo[256];
    __device__ double operator[](int f) const { return r[f]; }
};
__device__ __forceinline__ double clear_low9(double y)        // (uniform operand: scalar ALU)
{
    return __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, y) & ~0x1FFull);
}

// BOUNDED: the caller vouches that every mean lies in [0, 255] (true for a state that started at zero or
// inside that range: a mean of uint8 values stays there up to rounding), so |frame - mean| < 256 and the
// saturation of the difference -- one of the 14 operations per pixel -- can go
// SPLIT (RECIP = RecipArgs only): the second quotient in two operations, see RecipArgs::lo
template <int V, typename RECIP, bool BOUNDED, bool SPLIT = false>
__global__ void __launch_bounds__(kBlock)
bg_mean_u8_fast_kernel(const uint8_t *__restrict__ frames, uint8_t *__restrict__ diff,
                       double *__restrict__ mean, const RECIP recip, long long n_seen, int n, size_t px)
{
    size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * V;
    if (i0 >= px)
        return;
    double m[V];
#pragma unroll
    for (int k = 0; k < V; k++)
        m[k] = mean[i0 + k];
    typedef unsigned int v2u __attribute__((ext_vector_type(V / 4)));
    // frames and differences are one-touch streams: non-temporal loads/stores keep them out of L2.
    // Frames are walked in groups of kAhead: the next group's loads are issued before this
    // group's arithmetic (only 2 waves per SIMD exist at 16 px per thread), so a load has kAhead
    // frame steps to arrive and a CU keeps 8 waves x 8 KB in flight (measured per 256 x 1080p:
    // 4 frames 0.253 ms, 8 frames 0.241, 16 frames 0.250 -- registers cost the third wave).
    constexpr int kAhead = 8;
    const uint8_t *src = frames + i0;
    v2u cur[kAhead], nxt[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; j++)
        nxt[j] = __builtin_nontemporal_load(
            reinterpret_cast<const v2u *>(src + (size_t)min(j, n - 1) * px));
    for (int f0 = 0; f0 < n; f0 += kAhead) {
#pragma unroll
        for (int j = 0; j < kAhead; j++)
            cur[j] = nxt[j];
        if (f0 + kAhead < n) {
#pragma unroll
            for (int j = 0; j < kAhead; j++)
                nxt[j] = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(
                    src + (size_t)min(f0 + kAhead + j, n - 1) * px));
        }
#pragma unroll
        for (int j = 0; j < kAhead; j++) {
            const int f = f0 + j;
            if (f >= n)
                break;
            uint8_t p[V], o[V];
            memcpy(p, &cur[j], V);
            const double dn = (double)(n_seen + f), dn1 = (double)(n_seen + f + 1);
            const double y = recip[f];
            double yh = 0.0, yl = 0.0;
            if constexpr (SPLIT) {
                yh = clear_low9(y);
                yl = recip.lo[f];
            }
#pragma unroll
            for (int k = 0; k < V; k++) {
                const double fr = (double)p[k];
                o[k] = BOUNDED ? (uint8_t)(int)fabs(fr - m[k]) : sat_u8_trunc(fabs(fr - m[k]));
                double q2;
                if constexpr (SPLIT)
                    q2 = fma(fr, yl, fr * yh);
                else
                    q2 = div_by_uniform(fr, dn1, y);
                m[k] = div_by_uniform(m[k] * dn, dn1, y) + q2;
            }
            if (diff) {
                v2u v;
                memcpy(&v, o, V);
                __builtin_nontemporal_store(v, reinterpret_cast<v2u *>(diff + (size_t)f * px + i0));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < V; k++)
        mean[i0 + k] = m[k];
}

// (Tried and dropped, round 2: looking the second quotient frame/(n+1) up in a per-frame 256-entry
// float64 table in LDS instead of computing it -- three float64 operations fewer per pixel, but
// the data-dependent ds_read_b64 per pixel plus a barrier per four frames cost more than they
// saved: 0.293 ms against 0.269 ms per 256 x 1080p.)
template <int V>
__global__ void __launch_bounds__(kBlock)
bg_static_u8_kernel(const uint8_t *__restrict__ frames, uint8_t *__restrict__ diff,
                    const double *__restrict__ bg, int n, size_t px)
{
    size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * V;
    if (i0 >= px)
        return;
    double m[V];
#pragma unroll
    for (int k = 0; k < V; k++)
        m[k] = bg[i0 + k];
    for (int f = 0; f < n; f++) {
        uint8_t p[V], o[V];
        if (V == 8) {
            uint2 v = *reinterpret_cast<const uint2 *>(frames + (size_t)f * px + i0);
            memcpy(p, &v, 8);
        } else {
            p[0] = frames[(size_t)f * px + i0];
        }
#pragma unroll
        for (int k = 0; k < V; k++)
            o[k] = sat_u8_trunc(fabs((double)p[k] - m[k]));
        if (V == 8) {
            uint2 v;
            memcpy(&v, o, 8);
            *reinterpret_cast<uint2 *>(diff + (size_t)f * px + i0) = v;
        } else {
            diff[(size_t)f * px + i0] = o[0];
        }
    }
}

// ---- exponential moving average, float32 state -------------------------------------------
template <int V>
__global__ void __launch_bounds__(kBlock)
bg_ema_u8_kernel(const uint8_t *__restrict__ frames, uint8_t *__restrict__ diff,
                 float *__restrict__ bg, long long n_seen, float rate, int n, size_t px)
{
    size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * V;
    if (i0 >= px)
        return;
    float m[V];
#pragma unroll
    for (int k = 0; k < V; k++)
        m[k] = bg[i0 + k];
    for (int f = 0; f < n; f++) {
        uint8_t p[V], o[V];
        if (V == 8) {
            uint2 v = *reinterpret_cast<const uint2 *>(frames + (size_t)f * px + i0);
            memcpy(p, &v, 8);
        } else {
            p[0] = frames[(size_t)f * px + i0];
        }
        const bool first = (n_seen + f) == 0;
#pragma unroll
        for (int k = 0; k < V; k++) {
            float fr = (float)p[k];
            if (first)
                m[k] = fr;
            float d = fr - m[k];
            float a = truncf(fabsf(d));
            o[k] = (uint8_t)(int)(a > 255.0f ? 255.0f : a);
            float step = rate * d;
            m[k] = m[k] + step;
        }
        if (diff) {
            if (V == 8) {
                uint2 v;
                memcpy(&v, o, 8);
                *reinterpret_cast<uint2 *>(diff + (size_t)f * px + i0) = v;
            } else {
                diff[(size_t)f * px + i0] = o[0];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < V; k++)
        bg[i0 + k] = m[k];
}

// f32 frames: 4 pixels (16 B) per thread per frame
template <int V>
__global__ void __launch_bounds__(kBlock)
bg_ema_f32_kernel(const float *__restrict__ frames, float *__restrict__ diff,
                  float *__restrict__ bg, long long n_seen, float rate, int n, size_t px)
{
    size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * V;
    if (i0 >= px)
        return;
    float m[V];
#pragma unroll
    for (int k = 0; k < V; k++)
        m[k] = bg[i0 + k];
    for (int f = 0; f < n; f++) {
        float p[V], o[V];
        if (V == 4) {
            float4 v = *reinterpret_cast<const float4 *>(frames + (size_t)f * px + i0);
            p[0] = v.x, p[1] = v.y, p[2] = v.z, p[3] = v.w;
        } else {
            p[0] = frames[(size_t)f * px + i0];
        }
        const bool first = (n_seen + f) == 0;
#pragma unroll
        for (int k = 0; k < V; k++) {
            if (first)
                m[k] = p[k];
            float d = p[k] - m[k];
            o[k] = fabsf(d);
            float step = rate * d;
            m[k] = m[k] + step;
        }
        if (diff) {
            if (V == 4)
                *reinterpret_cast<float4 *>(diff + (size_t)f * px + i0) =
                    make_float4(o[0], o[1], o[2], o[3]);
            else
                diff[(size_t)f * px + i0] = o[0];
        }
    }
#pragma unroll
    for (int k = 0; k < V; k++)
        bg[i0 + k] = m[k];
}

// ---- Welford (measure_mean_std, video/analysis/video.py:48-50) ---------------------------
template <int V>
__global__ void __launch_bounds__(kBlock)
welford_u8_kernel(const uint8_t *__restrict__ frames, double *__restrict__ mean,
                  double *__restrict__ m2, long long n_seen, int n, size_t px)
{
    size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * V;
    if (i0 >= px)
        return;
    double m[V], q[V];
#pragma unroll
    for (int k = 0; k < V; k++) {
        m[k] = mean[i0 + k];
        q[k] = m2[i0 + k];
    }
    for (int f = 0; f < n; f++) {
        uint8_t p[V];
        if (V == 8) {
            uint2 v = *reinterpret_cast<const uint2 *>(frames + (size_t)f * px + i0);
            memcpy(p, &v, 8);
        } else {
            p[0] = frames[(size_t)f * px + i0];
        }
        const double dn1 = (double)(n_seen + f + 1);
#pragma unroll
        for (int k = 0; k < V; k++) {
            double fr = (double)p[k];
            double delta = fr - m[k];
            m[k] = m[k] + delta / dn1;
            q[k] = q[k] + delta * (fr - m[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < V; k++) {
        mean[i0 + k] = m[k];
        m2[i0 + k] = q[k];
    }
}

// ---- measure_mean / measure_mean_std on frames of any dtype the reference meets (uint8 from the
//      codecs, int16 from FilterTimeDifference, float32 from normalised videos), video/analysis/video.py:26-55.
// NumPy's promotions are part of the definition: `frame/(n + 1)` is float64 for integer frames but
// FLOAT32 for float32 frames (a float32 array divided by a Python int stays float32), and only then
// joins the float64 running mean; Welford's `frame - mean` promotes the frame exactly.
template <class T>
__device__ __forceinline__ double quotient_as_numpy(T v, double dn1)
{
    return (double)v / dn1;
}
template <>
__device__ __forceinline__ double quotient_as_numpy<float>(float v, double dn1)
{
    // float32 / float32(n + 1), correctly rounded (the double quotient rounds to the same float: 53 >= 2*24 + 2)
    return (double)(float)((double)v / (double)(float)dn1);
}

template <class T, bool WELFORD>
__global__ void __launch_bounds__(kBlock)
temporal_stats_kernel(const T *__restrict__ frames, double *__restrict__ mean, double *__restrict__ m2,
                      long long n_seen, int n, size_t px)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= px)
        return;
    double m = mean[i], q = WELFORD ? m2[i] : 0.0;
    for (int f = 0; f < n; f++) {
        const T v = frames[(size_t)f * px + i];
        const double dn = (double)(n_seen + f), dn1 = (double)(n_seen + f + 1);
        if (WELFORD) {
            const double fr = (double)v, delta = fr - m;
            m = m + delta / dn1;
            q = q + delta * (fr - m);
        } else {
            m = m * dn / dn1 + quotient_as_numpy<T>(v, dn1);
        }
    }
    mean[i] = m;
    if (WELFORD)
        m2[i] = q;
}

// ---- pointwise ------------------------------------------------------------------------------
// 16 bytes per lane where the buffers allow it, scalar tail otherwise
__global__ void __launch_bounds__(kBlock)
threshold_u8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t count,
                    int thresh, int maxval, int vec_ok)
{
    size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 16;
    if (i0 >= count)
        return;
    if (vec_ok && i0 + 16 <= count) {
        uint4 v = *reinterpret_cast<const uint4 *>(src + i0);
        uint8_t p[16];
        memcpy(p, &v, 16);
#pragma unroll
        for (int k = 0; k < 16; k++)
            p[k] = (int)p[k] > thresh ? (uint8_t)maxval : (uint8_t)0;
        memcpy(&v, p, 16);
        *reinterpret_cast<uint4 *>(dst + i0) = v;
    } else {
        for (size_t i = i0; i < count && i < i0 + 16; i++)
            dst[i] = (int)src[i] > thresh ? (uint8_t)maxval : (uint8_t)0;
    }
}

__global__ void __launch_bounds__(kBlock)
time_difference_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                       int16_t *__restrict__ out, size_t count, int vec_ok)
{
    size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 8;
    if (i0 >= count)
        return;
    if (vec_ok && i0 + 8 <= count) {
        uint2 va_ = *reinterpret_cast<const uint2 *>(a + i0);
        uint2 vb_ = *reinterpret_cast<const uint2 *>(b + i0);
        uint8_t pa[8], pb[8];
        int16_t o[8];
        memcpy(pa, &va_, 8);
        memcpy(pb, &vb_, 8);
#pragma unroll
        for (int k = 0; k < 8; k++)
            o[k] = (int16_t)((int)pa[k] - (int)pb[k]);
        uint4 vo;
        memcpy(&vo, o, 16);
        *reinterpret_cast<uint4 *>(out + i0) = vo;
    } else {
        for (size_t i = i0; i < count && i < i0 + 8; i++)
            out[i] = (int16_t)((int)a[i] - (int)b[i]);
    }
}

__global__ void __launch_bounds__(kBlock)
mono_mean_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t pixels)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= pixels)
        return;
    // np.mean(axis=2): float64 sum of the 3 channels, divided by 3.0, astype(uint8) truncates
    double s = (double)src[3 * i] + (double)src[3 * i + 1] + (double)src[3 * i + 2];
    dst[i] = (uint8_t)(int)(s / 3.0);
}

__global__ void __launch_bounds__(kBlock)
normalize_u8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t count,
                    double fmin, double fmax, double alpha, double tmin)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count)
        return;
    double f = (double)src[i];
    f = f < fmin ? fmin : (f > fmax ? fmax : f);   // np.clip(frame, fmin, fmax)
    double v = (f - fmin) * alpha + tmin;          // video/filters.py:129
    int iv = (int)v;                               // astype: C truncation
    dst[i] = (uint8_t)iv;                          // wraps like NumPy for out-of-range targets
}

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

size_t bg_scratch_bytes(int n) { return sizeof(double) * (size_t)(n > 0 ? n : 1); }

int launch_bg(int mode, int dtype, const void *frames, void *diff, void *state, int64_t n_seen,
              double rate, int n, size_t px, hipStream_t st, double *recip_scratch, bool mean_in_u8_range)
{
    VA_REQUIRE(frames && state, "va_bg_update: frames/state must not be NULL");
    VA_REQUIRE(n >= 0 && px > 0, "va_bg_update: bad sizes n=%d px=%zu", n, px);
    if (n == 0)
        return VA_OK;
    if (dtype == VA_U8) {
        const uint8_t *fr = (const uint8_t *)frames;
        uint8_t *df = (uint8_t *)diff;
        bool vec = (px % kVec == 0) && aligned(fr, 8) && (!df || aligned(df, 8));
        int grid = vec ? cdiv((long long)(px / kVec), kBlock) : cdiv((long long)px, kBlock);
        if (mode == VA_BG_MEAN) {
            // 16 px per thread when the streams allow 16-byte pieces (1 KB per wave instruction)
            // (8 px per thread, i.e. twice the waves with 8-byte accesses: 0.271-0.276 against 0.259-0.265 ms)
            const bool wide = px % 16 == 0 && aligned(fr, 16) && (!df || aligned(df, 16));
            if (vec && n <= 256) {
                RecipArgs ra;
                for (int f = 0; f < 256; f++) {
                    const double d = (double)(n_seen + (f < n ? f : 0) + 1), y = 1.0 / d;
                    unsigned long long bits;
                    memcpy(&bits, &y, 8);
                    bits &= ~0x1FFull;
                    double yh;
                    memcpy(&yh, &bits, 8);
                    ra.r[f] = y;
                    ra.lo[f] = fma(y, fma(-d, y, 1.0), y - yh);      // (host fma: correctly rounded)
                }
#ifdef BG_NO_SPLIT                                                    // (A/B builds: tools/debug/build_variant.sh)
                const bool split = false;
#else
                const bool split = n_seen + n < (1ll << 40);         // where the two-operation quotient is proven
#endif
                if (wide && mean_in_u8_range && split)
                    bg_mean_u8_fast_kernel<16, RecipArgs, true, true><<<cdiv((long long)(px / 16), kBlock), kBlock, 0, st>>>(
                        fr, df, (double *)state, ra, n_seen, n, px);
                else if (wide && mean_in_u8_range)
                    bg_mean_u8_fast_kernel<16, RecipArgs, true><<<cdiv((long long)(px / 16), kBlock), kBlock, 0, st>>>(
                        fr, df, (double *)state, ra, n_seen, n, px);
                else if (wide && split)
                    bg_mean_u8_fast_kernel<16, RecipArgs, false, true><<<cdiv((long long)(px / 16), kBlock), kBlock, 0, st>>>(
                        fr, df, (double *)state, ra, n_seen, n, px);
                else if (wide)
                    bg_mean_u8_fast_kernel<16, RecipArgs, false><<<cdiv((long long)(px / 16), kBlock), kBlock, 0, st>>>(
                        fr, df, (double *)state, ra, n_seen, n, px);
                else
                    bg_mean_u8_fast_kernel<8, RecipArgs, false><<<grid, kBlock, 0, st>>>(fr, df, (double *)state, ra,
                                                                                        n_seen, n, px);
            } else if (vec && recip_scratch) {
                bg_reciprocals_kernel<<<cdiv(n, kBlock), kBlock, 0, st>>>(recip_scratch, n_seen, n);
                VA_LAUNCH_CHECK("bg_reciprocals_kernel");
                if (wide)
                    bg_mean_u8_fast_kernel<16, const double *, false><<<cdiv((long long)(px / 16), kBlock), kBlock, 0, st>>>(
                        fr, df, (double *)state, recip_scratch, n_seen, n, px);
                else
                    bg_mean_u8_fast_kernel<8, const double *, false><<<grid, kBlock, 0, st>>>(
                        fr, df, (double *)state, recip_scratch, n_seen, n, px);
            } else if (vec)
                bg_mean_u8_kernel<8><<<grid, kBlock, 0, st>>>(fr, df, (double *)state, n_seen, n, px);
            else
                bg_mean_u8_kernel<1><<<grid, kBlock, 0, st>>>(fr, df, (double *)state, n_seen, n, px);
        } else if (mode == VA_BG_EMA) {
            if (vec)
                bg_ema_u8_kernel<8><<<grid, kBlock, 0, st>>>(fr, df, (float *)state, n_seen,
                                                            (float)rate, n, px);
            else
                bg_ema_u8_kernel<1><<<grid, kBlock, 0, st>>>(fr, df, (float *)state, n_seen,
                                                            (float)rate, n, px);
        } else if (mode == VA_BG_STATIC) {
            VA_REQUIRE(df, "va_bg_update: static mode needs diff_out");
            if (vec)
                bg_static_u8_kernel<8><<<grid, kBlock, 0, st>>>(fr, df, (const double *)state, n, px);
            else
                bg_static_u8_kernel<1><<<grid, kBlock, 0, st>>>(fr, df, (const double *)state, n, px);
        } else {
            VA_REQUIRE(false, "va_bg_update: unsupported mode %d for u8 frames", mode);
        }
    } else if (dtype == VA_F32) {
        VA_REQUIRE(mode == VA_BG_EMA, "va_bg_update: float32 frames support VA_BG_EMA only");
        const float *fr = (const float *)frames;
        float *df = (float *)diff;
        bool vec = (px % 4 == 0) && aligned(fr, 16) && (!df || aligned(df, 16));
        int grid = vec ? cdiv((long long)(px / 4), kBlock) : cdiv((long long)px, kBlock);
        if (vec)
            bg_ema_f32_kernel<4><<<grid, kBlock, 0, st>>>(fr, df, (float *)state, n_seen,
                                                         (float)rate, n, px);
        else
            bg_ema_f32_kernel<1><<<grid, kBlock, 0, st>>>(fr, df, (float *)state, n_seen,
                                                         (float)rate, n, px);
    } else {
        VA_REQUIRE(false, "va_bg_update: unknown dtype %d", dtype);
    }
    VA_LAUNCH_CHECK("bg kernel");
    return VA_OK;
}

int launch_welford(const uint8_t *frames, double *mean, double *m2, int64_t n_seen, int n,
                   size_t px, hipStream_t st)
{
    VA_REQUIRE(frames && mean && m2, "va_welford_u8: NULL argument");
    if (n <= 0)
        return VA_OK;
    bool vec = (px % kVec == 0) && aligned(frames, 8);
    int grid = vec ? cdiv((long long)(px / kVec), kBlock) : cdiv((long long)px, kBlock);
    if (vec)
        welford_u8_kernel<8><<<grid, kBlock, 0, st>>>(frames, mean, m2, n_seen, n, px);
    else
        welford_u8_kernel<1><<<grid, kBlock, 0, st>>>(frames, mean, m2, n_seen, n, px);
    VA_LAUNCH_CHECK("welford_u8_kernel");
    return VA_OK;
}

int launch_temporal_stats(const void *frames, int dtype, double *mean, double *m2, int64_t n_seen, int n,
                          size_t px, hipStream_t st)
{
    VA_REQUIRE(frames && mean, "temporal statistics: NULL argument");
    VA_REQUIRE(dtype == VA_U8 || dtype == VA_I16 || dtype == VA_F32,
               "temporal statistics: frames must be uint8, int16 or float32 (dtype code %d)", dtype);
    if (n <= 0 || px == 0)
        return VA_OK;
    const int grid = cdiv((long long)px, kBlock);
#define VA_TS(T)                                                                                            \
    do {                                                                                                    \
        if (m2)                                                                                             \
            temporal_stats_kernel<T, true><<<grid, kBlock, 0, st>>>((const T *)frames, mean, m2, n_seen, n, px);  \
        else                                                                                                \
            temporal_stats_kernel<T, false><<<grid, kBlock, 0, st>>>((const T *)frames, mean, m2, n_seen, n, px); \
    } while (0)
    if (dtype == VA_U8)
        VA_TS(uint8_t);
    else if (dtype == VA_I16)
        VA_TS(int16_t);
    else
        VA_TS(float);
#undef VA_TS
    VA_LAUNCH_CHECK("temporal_stats_kernel");
    return VA_OK;
}

int launch_threshold_u8(const uint8_t *src, uint8_t *dst, size_t count, int thresh, int maxval,
                        hipStream_t st)
{
    VA_REQUIRE(src && dst, "va_threshold_u8: NULL argument");
    if (count == 0)
        return VA_OK;
    int vec = aligned(src, 16) && aligned(dst, 16);
    threshold_u8_kernel<<<cdiv((long long)cdiv((long long)count, 16), kBlock), kBlock, 0, st>>>(
        src, dst, count, thresh, maxval, vec);
    VA_LAUNCH_CHECK("threshold_u8_kernel");
    return VA_OK;
}

int launch_time_difference(const uint8_t *a, const uint8_t *b, int16_t *out, size_t count,
                           hipStream_t st)
{
    VA_REQUIRE(a && b && out, "va_time_difference_u8: NULL argument");
    if (count == 0)
        return VA_OK;
    int vec = aligned(a, 8) && aligned(b, 8) && aligned(out, 16);
    time_difference_kernel<<<cdiv((long long)cdiv((long long)count, 8), kBlock), kBlock, 0, st>>>(
        a, b, out, count, vec);
    VA_LAUNCH_CHECK("time_difference_kernel");
    return VA_OK;
}

int launch_mono_mean(const uint8_t *src, uint8_t *dst, size_t pixels, hipStream_t st)
{
    VA_REQUIRE(src && dst, "va_mono_mean_u8: NULL argument");
    if (pixels == 0)
        return VA_OK;
    if (src != dst && launch_pointwise_u8_x4(src, dst, pixels, 3, 3, 0, 0.0, 0.0, 0.0, 0.0, st)) {
        VA_LAUNCH_CHECK("prepare_u8_x4_kernel");
        return VA_OK;
    }
    mono_mean_kernel<<<cdiv((long long)pixels, kBlock), kBlock, 0, st>>>(src, dst, pixels);
    VA_LAUNCH_CHECK("mono_mean_kernel");
    return VA_OK;
}

// ---- interleaved channels <-> planes (any frame through the single-channel Gaussian) ---------
// planes are laid out (frame, channel, h, wp): every plane is a single-channel frame of its own,
// wp >= w columns wide.  Columns w .. wp-1 hold the BORDER_REFLECT_101 continuation of the row, so
// a blur of the plane equals the blur of the original row wherever its window stays left of wp.
template <int C, bool SPLIT>
__global__ void __launch_bounds__(kBlock)
channel_planes_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int h, int w, int wp,
                      size_t total_quads)
{
    // thread = 4 consecutive plane columns of one row of one frame (wp % 4 == 0)
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total_quads)
        return;
    const int qpr = wp / 4;
    const int xq = (int)(t % qpr) * 4;
    const size_t rowi = t / qpr;                 // f * h + y
    const size_t f = rowi / h, y = rowi % h;
    const uint8_t *inter_c = (SPLIT ? src : nullptr);
    uint8_t *inter = SPLIT ? nullptr : dst;
    const size_t inter_row = rowi * (size_t)w * C;
    // four whole pixels inside the frame whose 4 C interleaved bytes are C aligned dwords: dword loads and stores on
    // both sides (the byte gathers / scatters below made a colour blur four times the cost of its three planes)
    const uint8_t *inter_any = SPLIT ? src : dst;
    if (C > 1 && xq + 3 < w && ((w * C) & 3) == 0 && (reinterpret_cast<uintptr_t>(inter_any) & 3) == 0) {
        uint32_t in[C], out[C];
        uint8_t b[4 * C], o[4 * C];
        if (SPLIT) {
#pragma unroll
            for (int c = 0; c < C; c++)
                in[c] = reinterpret_cast<const uint32_t *>(src + inter_row + (size_t)xq * C)[c];
            memcpy(b, in, 4 * C);
#pragma unroll
            for (int c = 0; c < C; c++)
                *reinterpret_cast<uint32_t *>(dst + ((f * C + c) * h + y) * (size_t)wp + xq) =
                    (uint32_t)b[c] | ((uint32_t)b[C + c] << 8) | ((uint32_t)b[2 * C + c] << 16) | ((uint32_t)b[3 * C + c] << 24);
        } else {
#pragma unroll
            for (int c = 0; c < C; c++)
                in[c] = *reinterpret_cast<const uint32_t *>(src + ((f * C + c) * h + y) * (size_t)wp + xq);
            memcpy(b, in, 4 * C);                      // b[4 c + i]: channel c of pixel i
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int c = 0; c < C; c++)
                    o[i * C + c] = b[4 * c + i];
            memcpy(out, o, 4 * C);
#pragma unroll
            for (int c = 0; c < C; c++)
                reinterpret_cast<uint32_t *>(dst + inter_row + (size_t)xq * C)[c] = out[c];
        }
        return;
    }
    if (SPLIT) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int x = xq + i;
                x = x < w ? x : 2 * (w - 1) - x;     // reflected continuation (wp - w < w)
                x = max(x, 0);
                v |= (uint32_t)inter_c[inter_row + (size_t)x * C + c] << (8 * i);
            }
            *reinterpret_cast<uint32_t *>(dst + ((f * C + c) * h + y) * (size_t)wp + xq) = v;
        }
    } else {
        uint8_t p[C][4];
#pragma unroll
        for (int c = 0; c < C; c++)
            *reinterpret_cast<uint32_t *>(p[c]) =
                *reinterpret_cast<const uint32_t *>(src + ((f * C + c) * h + y) * (size_t)wp + xq);
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (xq + i < w) {
#pragma unroll
                for (int c = 0; c < C; c++)
                    inter[inter_row + (size_t)(xq + i) * C + c] = p[c][i];
            }
    }
}

// split == true: (n, h, w, c) interleaved -> (n, c, h, wp) planes; false: the way back (crop to w)
int launch_channel_planes(const uint8_t *src, uint8_t *dst, int n, int h, int w, int wp, int c,
                          bool split, hipStream_t st)
{
    VA_REQUIRE(src && dst && wp % 4 == 0 && wp >= w && wp - w < w && c >= 1 && c <= 4,
               "channel planes: bad arguments");
    if (n == 0)
        return VA_OK;
    const size_t total = (size_t)n * h * (wp / 4);
    const unsigned grid = (unsigned)((total + kBlock - 1) / kBlock);
#define VA_CP(CC)                                                                                   \
    do {                                                                                            \
        if (split)                                                                                  \
            channel_planes_kernel<CC, true><<<grid, kBlock, 0, st>>>(src, dst, h, w, wp, total);    \
        else                                                                                        \
            channel_planes_kernel<CC, false><<<grid, kBlock, 0, st>>>(src, dst, h, w, wp, total);   \
    } while (0)
    if (c == 1)
        VA_CP(1);
    else if (c == 2)
        VA_CP(2);
    else if (c == 3)
        VA_CP(3);
    else
        VA_CP(4);
#undef VA_CP
    VA_LAUNCH_CHECK("channel_planes_kernel");
    return VA_OK;
}

// ---- np.rot90 (FilterRotate, video/filters.py:341) --------------------------------------------
// Pixels are opaque E-byte elements (channels x dtype).  A 32 x 32 tile goes through LDS so that
// both the reads and the writes of a wave are rows of consecutive elements, whatever the turn.
template <typename E>
__global__ void __launch_bounds__(kBlock)
rot90_kernel(const E *__restrict__ src, E *__restrict__ dst, int h, int w, int k)
{
    __shared__ E tile[32][33];
    const int ho = (k & 1) ? w : h, wo = (k & 1) ? h : w;     // output frame
    const int tiles_x = (wo + 31) >> 5, tiles_y = (ho + 31) >> 5;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const size_t f = t / tiles_y;
    const E *in = src + f * (size_t)h * w;
    E *out = dst + f * (size_t)h * w;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // 32 x 8 threads
    // output tile (i0.., j0..); out[i][j] = in[si][sj] with
    //   k=1: (j, w-1-i)   k=2: (h-1-i, w-1-j)   k=3: (h-1-j, i)   k=0: (i, j)
    const int i0 = ty * 32, j0 = tx * 32;
    // load: walk the SOURCE rows of the tile so that lx runs along a source row
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int a = ly + r, b = lx;       // tile-local coordinates in the source orientation
        int si, sj;
        if (k == 1) {            // source rows = output columns j, source cols descend with i
            si = j0 + a;
            sj = w - 1 - (i0 + b);
        } else if (k == 2) {
            si = h - 1 - (i0 + a);
            sj = w - 1 - (j0 + b);
        } else if (k == 3) {
            si = h - 1 - (j0 + a);
            sj = i0 + b;
        } else {
            si = i0 + a;
            sj = j0 + b;
        }
        if (si >= 0 && si < h && sj >= 0 && sj < w)
            tile[a][b] = in[(size_t)si * w + sj];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int i = i0 + ly + r, j = j0 + lx;
        if (i < ho && j < wo) {
            // which (a, b) of the load holds out[i][j]?
            const int di = ly + r, dj = lx;
            E v;
            if (k == 1)
                v = tile[dj][di];        // si = j, sj = w-1-i  ->  a = j-j0, b = i-i0
            else if (k == 3)
                v = tile[dj][di];        // si = h-1-j, sj = i  ->  a = j-j0, b = i-i0
            else
                v = tile[di][dj];
            out[(size_t)i * wo + j] = v;
        }
    }
}

struct Elem3 { uint8_t b[3]; };
struct Elem6 { uint16_t b[3]; };
struct Elem12 { uint32_t b[3]; };

int launch_rot90(const void *src, void *dst, int n, int h, int w, int elem_bytes, int k,
                 hipStream_t st)
{
    VA_REQUIRE(src && dst && src != dst, "va_rot90: src/dst must be distinct non-NULL");
    VA_REQUIRE(n >= 0 && h > 0 && w > 0, "va_rot90: bad shape (%d,%d,%d)", n, h, w);
    k &= 3;
    if (n == 0)
        return VA_OK;
    const int ho = (k & 1) ? w : h, wo = (k & 1) ? h : w;
    const size_t blocks = (size_t)n * ((ho + 31) / 32) * ((wo + 31) / 32);
    VA_REQUIRE(blocks < ((size_t)1 << 31), "va_rot90: too many tiles");
#define VA_ROT(T) rot90_kernel<T><<<(unsigned)blocks, kBlock, 0, st>>>((const T *)src, (T *)dst, h, w, k)
    switch (elem_bytes) {
    case 1: VA_ROT(uint8_t); break;
    case 2: VA_ROT(uint16_t); break;
    case 3: VA_ROT(Elem3); break;
    case 4: VA_ROT(uint32_t); break;
    case 6: VA_ROT(Elem6); break;
    case 8: VA_ROT(uint64_t); break;
    case 12: VA_ROT(Elem12); break;
    default:
        VA_REQUIRE(false, "va_rot90: unsupported pixel size of %d bytes", elem_bytes);
    }
#undef VA_ROT
    VA_LAUNCH_CHECK("rot90_kernel");
    return VA_OK;
}

int launch_normalize_u8(const uint8_t *src, uint8_t *dst, size_t count, double fmin, double fmax,
                        double alpha, double tmin, hipStream_t st)
{
    VA_REQUIRE(src && dst, "va_normalize_u8: NULL argument");
    if (count == 0)
        return VA_OK;
    // (in place is fine: a thread reads its four bytes before it writes them)
    if (launch_pointwise_u8_x4(src, dst, count, 1, -1, 1, fmin, fmax, alpha, tmin, st)) {
        VA_LAUNCH_CHECK("prepare_u8_x4_kernel");
        return VA_OK;
    }
    normalize_u8_kernel<<<cdiv((long long)count, kBlock), kBlock, 0, st>>>(src, dst, count, fmin,
                                                                          fmax, alpha, tmin);
    VA_LAUNCH_CHECK("normalize_u8_kernel");
    return VA_OK;
}

}  // namespace va
