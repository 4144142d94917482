// va_synth.hip -- remaining pointwise pieces of the video.filters / video.io surface (N4, A5)
//
//   va_normalize      FilterNormalize._process_frame for uint8 / float32 frames and uint8 /
//                     float32 / float64 targets            (video/filters.py:101-135)
//   va_gaussian_noise VideoGaussianNoise.get_frame on the device, seeded and seekable
//                                                          (video/io/computed.py:15-41)
#include "va_common.h"

namespace va {
namespace {

constexpr int kBlock = 256;

// np.clip(frame, fmin, fmax); (frame - fmin) * alpha + tmin in float64; astype(target)
template <typename SRC, typename DST>
__global__ void __launch_bounds__(kBlock)
normalize_kernel(const SRC *__restrict__ src, DST *__restrict__ dst, size_t count, double fmin, double fmax,
                 double alpha, double tmin)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count)
        return;
    double f = (double)src[i];
    f = f < fmin ? fmin : (f > fmax ? fmax : f);
    const double v = (f - fmin) * alpha + tmin;
    if (sizeof(DST) == 1)
        dst[i] = (DST)(int)v;          // astype(uint8): C truncation, wraps like NumPy
    else
        dst[i] = (DST)v;
}

// FilterCrop -> FilterMonochrome -> FilterNormalize in one pass over the source frames (A5: the
// pointwise pre-stages of the chain, fused with each other and run on the device inside the
// engine, so a chain that starts with them needs no host round trip):
//   crop      frame[top:top+height, left:left+width]            (video/filters.py:238-248)
//   mono      3: np.mean(frame, axis=2).astype(uint8) (float64 mean, truncated); 0..2: that channel;
//             -1: channels kept                                 (video/filters.py:359-374)
//   normalize clip, (f - fmin)*alpha + tmin in float64, astype(uint8)   (video/filters.py:126-132)
__global__ void __launch_bounds__(kBlock)
prepare_u8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int src_h, int src_w, int src_c,
                  int left, int top, int width, int height, int mono, int out_c, int normalize, double fmin,
                  double fmax, double alpha, double tmin, size_t total)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total)
        return;
    const int ch = (int)(i % out_c);
    const size_t p = i / out_c;
    const int x = (int)(p % width), y = (int)((p / width) % height);
    const size_t f = p / ((size_t)width * height);
    const uint8_t *px = src + ((f * src_h + (top + y)) * (size_t)src_w + (left + x)) * src_c;
    int v;
    if (mono == 3) {
        const double s = (double)px[0] + (double)px[1] + (double)px[2];
        v = (int)(s / 3.0);
    } else if (mono >= 0) {
        v = px[mono];
    } else {
        v = px[ch];
    }
    if (normalize) {
        double d = (double)v;
        d = d < fmin ? fmin : (d > fmax ? fmax : d);
        v = (int)((d - fmin) * alpha + tmin);
    }
    dst[i] = (uint8_t)v;
}

// The same, four consecutive output samples of one row per thread and one 4-byte store; grid = (dwords of an output
// row, rows, frames), so no 64-bit division per sample.  (One sample per thread is bound by the launch rate of waves:
// 8.3 M of them for 256 x 1080p, more than twice the time of the whole chain behind it.)
__global__ void __launch_bounds__(kBlock)
prepare_u8_x4_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int src_h, int src_w, int src_c,
                     int left, int top, int width, int height, int mono, int out_c, int normalize, double fmin,
                     double fmax, double alpha, double tmin)
{
    // normalisation maps one of 256 values: every workgroup tabulates it with the float64 arithmetic of the
    // one-sample kernel (thread t: value t), a sample then costs one LDS byte
    __shared__ uint8_t lut[kBlock];
    static_assert(kBlock == 256, "one table entry per thread");
    {
        double d = (double)threadIdx.x;
        d = d < fmin ? fmin : (d > fmax ? fmax : d);
        lut[threadIdx.x] = normalize ? (uint8_t)(int)((d - fmin) * alpha + tmin) : (uint8_t)threadIdx.x;
    }
    __syncthreads();
    const int q = blockIdx.x * kBlock + threadIdx.x;                 // dword of the output row
    const int row_samples = width * out_c;
    if (4 * q >= row_samples)
        return;
    const int y = blockIdx.y;
    const size_t f = blockIdx.z;
    const uint8_t *row = src + ((f * src_h + (top + y)) * (size_t)src_w + left) * src_c;
    const bool row_aligned = (reinterpret_cast<uintptr_t>(row) & 3) == 0;          // uniform
    uint32_t word = 0;
    if (row_aligned && (mono < 0 || (src_c == 3 && out_c == 1))) {
        // the thread's source bytes are 4 (channels kept) or 12 (three channels -> one) contiguous, aligned bytes:
        // dword loads instead of 4 ... 12 byte gathers (the texture addresser, not HBM, bounded the byte version)
        if (mono < 0) {
            const uint32_t in = *reinterpret_cast<const uint32_t *>(row + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; k++)
                word |= (uint32_t)lut[(in >> (8 * k)) & 0xFFu] << (8 * k);
        } else {
            const uint32_t *p3 = reinterpret_cast<const uint32_t *>(row + 12 * (size_t)q);
            const uint32_t d0 = p3[0], d1 = p3[1], d2 = p3[2];
            uint8_t b[12];
            memcpy(b, &d0, 4);
            memcpy(b + 4, &d1, 4);
            memcpy(b + 8, &d2, 4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned v = mono == 3 ? (((unsigned)b[3 * k] + b[3 * k + 1] + b[3 * k + 2]) * 43691u) >> 17
                                             : (mono == 0 ? b[3 * k] : (mono == 1 ? b[3 * k + 1] : b[3 * k + 2]));
                word |= (uint32_t)lut[v] << (8 * k);
            }
        }
        *reinterpret_cast<uint32_t *>(dst + (f * height + y) * (size_t)row_samples + 4 * q) = word;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int e = 4 * q + k;
        const int x = out_c == 1 ? e : e / out_c, ch = e - x * out_c;
        const uint8_t *px = row + (size_t)x * src_c;
        unsigned v;
        if (mono == 3) {
            // (int)(float64(a + b + c) / 3.0) == (a + b + c) / 3 for integers: the quotient's fraction is 0, 1/3 or
            // 2/3, never within an ulp of the next integer; 43691 = ceil(2^17 / 3) is exact up to 3 * 255
            const unsigned sum = (unsigned)px[0] + px[1] + px[2];
            v = (sum * 43691u) >> 17;
        } else if (mono >= 0) {
            v = px[mono];
        } else {
            v = px[ch];
        }
        word |= (uint32_t)lut[v] << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(dst + (f * height + y) * (size_t)row_samples + 4 * q) = word;
}

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so sample i of the stream is a pure
// function of (seed, i) -- any frame of the noise video can be produced on its own
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

// thread = one Philox block = two standard normals (Box-Muller on two 53-bit uniforms)
template <typename DST>
__global__ void __launch_bounds__(kBlock)
gaussian_noise_kernel(DST *__restrict__ dst, size_t count, double mean, double stdev, uint64_t seed,
                      uint64_t first_index)
{
    const size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;     // pair index inside this call
    // absolute sample indices first_index + 2j', aligned to pairs of the ABSOLUTE stream
    const uint64_t pair0 = first_index >> 1;
    const uint64_t g = pair0 + j;                                    // absolute pair index
    const uint64_t s0 = 2 * g, s1 = 2 * g + 1;                       // absolute sample indices
    if (s0 >= first_index + count)
        return;
    uint32_t r[4];
    philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const double two53 = 1.0 / 9007199254740992.0;
    const double u1 = ((double)((((uint64_t)r[0] << 32) | r[1]) >> 11) + 1.0) * two53;     // (0, 1]
    const double u2 = (double)((((uint64_t)r[2] << 32) | r[3]) >> 11) * two53;             // [0, 1)
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    const double z[2] = {rad * cs, rad * sn};
    const uint64_t s[2] = {s0, s1};
#pragma unroll
    for (int b = 0; b < 2; b++) {
        if (s[b] < first_index || s[b] >= first_index + count)
            continue;
        double v = z[b] * stdev + mean;
        if (sizeof(DST) == 1) {
            v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);             // integer videos saturate
            dst[s[b] - first_index] = (DST)(int)v;
        } else {
            dst[s[b] - first_index] = (DST)v;
        }
    }
}

}  // namespace

int launch_normalize(const void *src, int src_dtype, void *dst, int dst_dtype, size_t count, double fmin,
                     double fmax, double alpha, double tmin, hipStream_t st)
{
    VA_REQUIRE(src && dst, "va_normalize: NULL argument");
    VA_REQUIRE(src_dtype == VA_U8 || src_dtype == VA_F32, "va_normalize: source dtype must be VA_U8 or VA_F32");
    VA_REQUIRE(dst_dtype == VA_U8 || dst_dtype == VA_F32 || dst_dtype == VA_F64,
               "va_normalize: target dtype must be VA_U8, VA_F32 or VA_F64");
    if (count == 0)
        return VA_OK;
    const unsigned grid = (unsigned)cdiv((long long)count, kBlock);
#define VA_NORM(S, D) normalize_kernel<S, D><<<grid, kBlock, 0, st>>>((const S *)src, (D *)dst, count, fmin, fmax, alpha, tmin)
    if (src_dtype == VA_U8) {
        if (dst_dtype == VA_U8) VA_NORM(uint8_t, uint8_t);
        else if (dst_dtype == VA_F32) VA_NORM(uint8_t, float);
        else VA_NORM(uint8_t, double);
    } else {
        if (dst_dtype == VA_U8) VA_NORM(float, uint8_t);
        else if (dst_dtype == VA_F32) VA_NORM(float, float);
        else VA_NORM(float, double);
    }
#undef VA_NORM
    VA_LAUNCH_CHECK("normalize_kernel");
    return VA_OK;
}

int launch_prepare_u8(const uint8_t *src, uint8_t *dst, int n, int src_h, int src_w, int src_c, int left, int top,
                      int width, int height, int mono, int normalize, double fmin, double fmax, double alpha,
                      double tmin, hipStream_t st)
{
    VA_REQUIRE(src && dst && src != dst, "va_prepare_u8: src/dst must be distinct non-NULL");
    VA_REQUIRE(n >= 0 && src_h > 0 && src_w > 0 && src_c >= 1 && src_c <= 4, "va_prepare_u8: bad source shape");
    VA_REQUIRE(left >= 0 && top >= 0 && width > 0 && height > 0 && left + width <= src_w && top + height <= src_h,
               "va_prepare_u8: rectangle (%d,%d,%d,%d) outside the %dx%d frame", left, top, width, height, src_w, src_h);
    VA_REQUIRE(mono >= -1 && mono <= 3 && (mono < 0 || (mono == 3 ? src_c == 3 : mono < src_c)),
               "va_prepare_u8: mono mode %d does not fit %d channels", mono, src_c);
    const int out_c = mono < 0 ? src_c : 1;
    const size_t total = (size_t)n * height * width * out_c;
    if (total == 0)
        return VA_OK;
    if ((width * out_c) % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 && height <= 65535 && n <= 65535) {
        const dim3 grid((unsigned)cdiv(width * out_c / 4, kBlock), (unsigned)height, (unsigned)n);
        prepare_u8_x4_kernel<<<grid, kBlock, 0, st>>>(src, dst, src_h, src_w, src_c, left, top, width, height, mono,
                                                     out_c, normalize, fmin, fmax, alpha, tmin);
        VA_LAUNCH_CHECK("prepare_u8_x4_kernel");
        return VA_OK;
    }
    prepare_u8_kernel<<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(src, dst, src_h, src_w, src_c, left, top, width,
                                                                        height, mono, out_c, normalize, fmin, fmax,
                                                                        alpha, tmin, total);
    VA_LAUNCH_CHECK("prepare_u8_kernel");
    return VA_OK;
}

// a flat run of samples through the four-samples-per-thread kernel (FilterMonochrome / FilterNormalize on their own):
// false when the shape does not fit it (the caller then runs its one-sample kernel)
bool launch_pointwise_u8_x4(const uint8_t *src, uint8_t *dst, size_t out_samples, int src_c, int mono, int normalize,
                            double fmin, double fmax, double alpha, double tmin, hipStream_t st)
{
    const int out_c = mono < 0 ? src_c : 1;
    if (out_samples % 4 != 0 || out_samples >= (1ull << 31) || out_samples % out_c != 0 ||
        reinterpret_cast<uintptr_t>(dst) % 4 != 0)
        return false;
    const int width = (int)(out_samples / out_c);
    const dim3 grid((unsigned)cdiv((long long)(out_samples / 4), kBlock), 1u, 1u);
    prepare_u8_x4_kernel<<<grid, kBlock, 0, st>>>(src, dst, 1, width, src_c, 0, 0, width, 1, mono, out_c, normalize, fmin,
                                                 fmax, alpha, tmin);
    return true;
}

int launch_gaussian_noise(void *dst, int dtype, size_t count, double mean, double stdev, uint64_t seed,
                          uint64_t first_index, hipStream_t st)
{
    VA_REQUIRE(dst, "va_gaussian_noise: NULL argument");
    VA_REQUIRE(dtype == VA_U8 || dtype == VA_F32 || dtype == VA_F64, "va_gaussian_noise: bad dtype %d", dtype);
    if (count == 0)
        return VA_OK;
    const uint64_t pairs = ((first_index + count + 1) >> 1) - (first_index >> 1);
    const unsigned grid = (unsigned)cdiv((long long)pairs, kBlock);
    if (dtype == VA_U8)
        gaussian_noise_kernel<uint8_t><<<grid, kBlock, 0, st>>>((uint8_t *)dst, count, mean, stdev, seed, first_index);
    else if (dtype == VA_F32)
        gaussian_noise_kernel<float><<<grid, kBlock, 0, st>>>((float *)dst, count, mean, stdev, seed, first_index);
    else
        gaussian_noise_kernel<double><<<grid, kBlock, 0, st>>>((double *)dst, count, mean, stdev, seed, first_index);
    VA_LAUNCH_CHECK("gaussian_noise_kernel");
    return VA_OK;
}

}  // namespace va
