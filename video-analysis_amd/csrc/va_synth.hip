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
