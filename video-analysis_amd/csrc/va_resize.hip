// va_resize.hip -- N4: cv2.resize for uint8 frames
//
// replaces  cv2.resize(frame, self.size, interpolation=self.interpolation),
//           FilterResize._process_frame, video/filters.py:310-314 (mode choice :274-293)
//
// OpenCV's 8-bit resize is a table-driven algorithm: per output column / row an index and a few
// fixed-point (11-bit) or float weights, then integer arithmetic with fixed rounding steps.  The
// tables (a few KB) are built on the host exactly as OpenCV builds them -- same float / double
// expressions, same saturating conversions -- and the per-pixel arithmetic runs on the GPU, one
// thread per output sample.  The op is a gather with 1-16 taps per sample: bound by the read of
// the source frames (L2-friendly: neighbouring outputs share taps) and the write of the result.
//   mode 0 NEAREST  1 LINEAR (an exact 2x2 shrink is AREA, as in OpenCV)  2 CUBIC (A = -0.75)
//   mode 3 AREA: integer shrink factors = block means, other shrinks = cell-overlap weights in
//          float (summation order of OpenCV's ResizeArea_Invoker), growing = LINEAR with
//          area-style sample positions
//   mode 4 LANCZOS4: 8 x 8 taps, 11-bit fixed point like CUBIC, int32 accumulation
// float32 frames (launch_resize_f32): the same tables with float coefficients, products summed
// from the first tap to the last in float (no contraction), no rounding step at the end.
// Parity with real OpenCV is unpinned offline (no cv2); the oracle restates the same published
// algorithm independently in C (the test-side CPU restatement) and the two are compared bit
// for bit by the GPU parity tests.
#include <math.h>

#include <vector>

#include "va_common.h"

namespace va {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

template <class T>
__global__ void __launch_bounds__(kBlock)
resize_nn_kernel(const T *__restrict__ src, T *__restrict__ dst, const int *__restrict__ xofs,
                 const int *__restrict__ yofs, int sh, int sw, int c, int dh, int dw, size_t total)
{
    // grid = (samples of an output row, output rows, frames): no 64-bit divisions per sample
    const int e = blockIdx.x * kBlock + threadIdx.x;                 // dx * c + ch
    if (e >= dw * c)
        return;
    (void)total;
    const int dx = c == 1 ? e : e / c, ch = e - dx * c, dy = blockIdx.y;
    const size_t f = blockIdx.z;
    const size_t i = ((f * dh + dy) * (size_t)dw) * c + e;
    dst[i] = src[(f * sh + yofs[dy]) * (size_t)sw * c + (size_t)xofs[dx] * c + ch];
}

// one output sample of the fixed-point interpolating modes (linear, cubic, Lanczos-4; `area` when growing)
template <int KS>
__device__ __forceinline__ uint8_t taps_sample(const uint8_t *__restrict__ frame, const int *__restrict__ xofs,
                                               const short *__restrict__ ialpha, const short *__restrict__ ibeta,
                                               int sy0, int xmax, int sh, int sw, int c, int dx, int ch, int dy)
{
    const int sx = xofs[dx];
    int rows[KS];
#pragma unroll
    for (int k = 0; k < KS; k++) {
        int sy = sy0 - KS / 2 + 1 + k;
        sy = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        const uint8_t *S = frame + (size_t)sy * sw * c + ch;
        int v = 0;
        if (KS == 2) {
            if (dx < xmax)
                v = S[(size_t)sx * c] * ialpha[dx * 2] + S[(size_t)(sx + 1) * c] * ialpha[dx * 2 + 1];
            else
                v = S[(size_t)sx * c] * 2048;
        } else {
#pragma unroll
            for (int j = 0; j < KS; j++) {
                int sxj = sx - KS / 2 + 1 + j;
                sxj = sxj < 0 ? 0 : (sxj > sw - 1 ? sw - 1 : sxj);     // replicated border
                v += S[(size_t)sxj * c] * ialpha[dx * KS + j];
            }
        }
        rows[k] = v;
    }
    int out;
    if (KS == 2) {
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        out = (((b0 * (rows[0] >> 4)) >> 16) + ((b1 * (rows[1] >> 4)) >> 16) + 2) >> 2;
    } else if (KS == 4) {
        long long s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            s += (long long)ibeta[dy * 4 + k] * rows[k];
        out = (int)((s + (1 << 21)) >> 22);
    } else {                                    // LANCZOS4: OpenCV accumulates in int (and wraps like it)
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < KS; k++)
            s += (uint32_t)((int)ibeta[dy * KS + k] * rows[k]);
        out = (int)(s + (1u << 21)) >> 22;
    }
    return sat_u8(out);
}

// ROWS output rows per thread (a loop, not unrolled): a wave of the one-row kernel lives for 64 samples, and a frame
// that grows to 2880 x 1620 then needs 78 k waves -- the launch rate of waves, not loads or arithmetic, set its time
// (linear, cubic and area all took 1.17 ms for 64 frames)
template <int KS, int ROWS>
__global__ void __launch_bounds__(kBlock)
resize_taps_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int *__restrict__ xofs,
                   const short *__restrict__ ialpha, const int *__restrict__ yofs,
                   const short *__restrict__ ibeta, int xmax, int sh, int sw, int c, int dh, int dw,
                   size_t total)
{
    // grid = (samples of an output row, output rows / ROWS, frames): no 64-bit divisions per sample
    const int e = blockIdx.x * kBlock + threadIdx.x;                 // dx * c + ch
    if (e >= dw * c)
        return;
    (void)total;
    const int dx = c == 1 ? e : e / c, ch = e - dx * c;
    const size_t f = blockIdx.z;
    const uint8_t *frame = src + f * (size_t)sh * sw * c;
#pragma unroll 1
    for (int r = 0; r < ROWS; r++) {
        const int dy = blockIdx.y * ROWS + r;
        if (dy >= dh)
            break;
        const size_t i = ((f * dh + dy) * (size_t)dw) * c + e;
        dst[i] = taps_sample<KS>(frame, xofs, ialpha, ibeta, yofs[dy], xmax, sh, sw, c, dx, ch, dy);
    }
}

// nearest neighbour, single-channel frames whose output rows are whole dwords: four samples per thread and one
// 4-byte store (0.64 -> 0.34 ms for 64 x 1080p -> 2880 x 1620: the one-sample kernel spends its time on 1.2 M tiny
// workgroups and byte stores).  The interpolating modes gain nothing from it (measured: linear 1.17 -> 1.29 ms,
// cubic 1.18 -> 2.36): they are bound by their byte gathers, seven to twenty per sample.
__global__ void __launch_bounds__(kBlock)
resize_nn_x4_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int *__restrict__ xofs,
                    const int *__restrict__ yofs, int sh, int sw, int dh, int dw)
{
    const int q = dw >> 2;
    const unsigned idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (unsigned)(q * dh))
        return;
    const int dy = idx / q, dx0 = 4 * (idx - dy * q);
    const size_t f = blockIdx.z;
    const uint8_t *row = src + (f * sh + yofs[dy]) * (size_t)sw;
    const int4 xo = *reinterpret_cast<const int4 *>(xofs + dx0);
    const uint32_t word = row[xo.x] | ((uint32_t)row[xo.y] << 8) | ((uint32_t)row[xo.z] << 16) | ((uint32_t)row[xo.w] << 24);
    *reinterpret_cast<uint32_t *>(dst + (f * dh + dy) * (size_t)dw + dx0) = word;
}

// float32 frames: float coefficients, left-to-right sums of products (-ffp-contract=off: no FMA)
template <int KS>
__global__ void __launch_bounds__(kBlock)
resize_taps_f32_kernel(const float *__restrict__ src, float *__restrict__ dst, const int *__restrict__ xofs,
                       const float *__restrict__ alpha, const int *__restrict__ yofs,
                       const float *__restrict__ beta, int xmax, int sh, int sw, int c, int dh, int dw,
                       size_t total)
{
    // grid = (samples of an output row, output rows, frames): no 64-bit divisions per sample
    const int e = blockIdx.x * kBlock + threadIdx.x;                 // dx * c + ch
    if (e >= dw * c)
        return;
    (void)total;
    const int dx = c == 1 ? e : e / c, ch = e - dx * c, dy = blockIdx.y;
    const size_t f = blockIdx.z;
    const size_t i = ((f * dh + dy) * (size_t)dw) * c + e;
    const float *frame = src + f * (size_t)sh * sw * c;
    const int sx = xofs[dx];
    float out = 0.f;
#pragma unroll
    for (int k = 0; k < KS; k++) {
        int sy = yofs[dy] - KS / 2 + 1 + k;
        sy = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        const float *S = frame + (size_t)sy * sw * c + ch;
        float v = 0.f;
        if (KS == 2) {
            if (dx < xmax) {
                const float p0 = S[(size_t)sx * c] * alpha[dx * 2], p1 = S[(size_t)(sx + 1) * c] * alpha[dx * 2 + 1];
                v = p0 + p1;
            } else {
                v = S[(size_t)sx * c] * 1.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < KS; j++) {
                int sxj = sx - KS / 2 + 1 + j;
                sxj = sxj < 0 ? 0 : (sxj > sw - 1 ? sw - 1 : sxj);
                const float pr = S[(size_t)sxj * c] * alpha[dx * KS + j];
                v += pr;
            }
        }
        const float pr = v * beta[dy * KS + k];
        out = k == 0 ? pr : out + pr;
    }
    dst[i] = out;
}

__global__ void __launch_bounds__(kBlock)
resize_area_fast_f32_kernel(const float *__restrict__ src, float *__restrict__ dst, int isx, int isy,
                            int wfull, int sh, int sw, int c, int dh, int dw, size_t total)
{
    // grid = (samples of an output row, output rows, frames): no 64-bit divisions per sample
    const int e = blockIdx.x * kBlock + threadIdx.x;                 // dx * c + ch
    if (e >= dw * c)
        return;
    (void)total;
    const int dx = c == 1 ? e : e / c, ch = e - dx * c, dy = blockIdx.y;
    const size_t f = blockIdx.z;
    const size_t i = ((f * dh + dy) * (size_t)dw) * c + e;
    const float *frame = src + f * (size_t)sh * sw * c + ch;
    const int sx0 = dx * isx, sy0 = dy * isy;
    float sum = 0.f;
    int count = 0;
    for (int yy = 0; yy < isy && sy0 + yy < sh; yy++)
        for (int xx = 0; xx < isx && sx0 + xx < sw; xx++) {
            sum += frame[((size_t)(sy0 + yy) * sw + sx0 + xx) * c];
            count++;
        }
    dst[i] = (sy0 + isy <= sh && dx < wfull) ? sum * (1.f / (float)(isx * isy)) : sum / (float)count;
}

__global__ void __launch_bounds__(kBlock)
resize_area_fast_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int isx, int isy,
                        int wfull, int sh, int sw, int c, int dh, int dw, size_t total)
{
    // grid = (samples of an output row, output rows, frames): no 64-bit divisions per sample
    const int e = blockIdx.x * kBlock + threadIdx.x;                 // dx * c + ch
    if (e >= dw * c)
        return;
    (void)total;
    const int dx = c == 1 ? e : e / c, ch = e - dx * c, dy = blockIdx.y;
    const size_t f = blockIdx.z;
    const size_t i = ((f * dh + dy) * (size_t)dw) * c + e;
    const uint8_t *frame = src + f * (size_t)sh * sw * c + ch;
    const int sx0 = dx * isx, sy0 = dy * isy;
    int sum = 0, count = 0;
    for (int yy = 0; yy < isy && sy0 + yy < sh; yy++)
        for (int xx = 0; xx < isx && sx0 + xx < sw; xx++) {
            sum += frame[((size_t)(sy0 + yy) * sw + sx0 + xx) * c];
            count++;
        }
    int out;
    if (sy0 + isy <= sh && dx < wfull) {
        if (isx == 2 && isy == 2)
            out = (sum + 2) >> 2;
        else
            out = __float2int_rn((float)sum * (1.f / (float)(isx * isy)));
    } else {
        out = __float2int_rn((float)sum / (float)count);
    }
    dst[i] = sat_u8(out);
}

struct DecAlpha {
    int si, di;
    float alpha;
};

__device__ __forceinline__ void store_area(uint8_t *p, float v) { *p = sat_u8(__float2int_rn(v)); }
__device__ __forceinline__ void store_area(float *p, float v) { *p = v; }

template <class T>
__global__ void __launch_bounds__(kBlock)
resize_area_kernel(const T *__restrict__ src, T *__restrict__ dst, const DecAlpha *__restrict__ xtab,
                   const int *__restrict__ xstart, const DecAlpha *__restrict__ ytab,
                   const int *__restrict__ ystart, int sh, int sw, int c, int dh, int dw, size_t total)
{
    // grid = (samples of an output row, output rows, frames): no 64-bit divisions per sample
    const int e = blockIdx.x * kBlock + threadIdx.x;                 // dx * c + ch
    if (e >= dw * c)
        return;
    (void)total;
    const int dx = c == 1 ? e : e / c, ch = e - dx * c, dy = blockIdx.y;
    const size_t f = blockIdx.z;
    const size_t i = ((f * dh + dy) * (size_t)dw) * c + e;
    const T *frame = src + f * (size_t)sh * sw * c + ch;
    float sum = 0.f;
    bool firstrow = true;
    for (int j = ystart[dy]; j < ystart[dy + 1]; j++) {
        const T *S = frame + (size_t)ytab[j].si * sw * c;
        float buf = 0.f;
        for (int k = xstart[dx]; k < xstart[dx + 1]; k++) {
            const float prod = (float)S[(size_t)xtab[k].si * c] * xtab[k].alpha;
            buf += prod;
        }
        const float t = ytab[j].alpha * buf;
        sum = firstrow ? t : sum + t;
        firstrow = false;
    }
    store_area(dst + i, sum);
}

short sat_short(float v)
{
    const long r = lrintf(v);
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

void cubic_coeffs(float x, float *cf)
{
    const float A = -0.75f;
    cf[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    cf[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    cf[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    cf[3] = 1.f - cf[0] - cf[1] - cf[2];
}

void lanczos4_coeffs(float x, float *coeffs)          // OpenCV's interpolateLanczos4
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-07f) {
        for (int i = 0; i < 8; i++)
            coeffs[i] = 0;
        coeffs[3] = 1;
        return;
    }
    float sum = 0;
    const double pi = 3.1415926535897932384626433832795;
    const double y0 = -(double)(x + 3) * pi * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        const double y = -(double)(x + 3 - i) * pi * 0.25;
        coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += coeffs[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++)
        coeffs[i] *= sum;
}

void area_tab(int ssize, int dsize, double scale, std::vector<DecAlpha> &tab, std::vector<int> &start)
{
    start.assign(dsize + 1, 0);
    for (int dx = 0; dx < dsize; dx++) {
        start[dx] = (int)tab.size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = fmin(scale, ssize - fsx1);
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3)
            tab.push_back({sx1 - 1, dx, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; sx++)
            tab.push_back({sx, dx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3)
            tab.push_back({sx2, dx, (float)(fmin(fmin(fsx2 - sx2, 1.), cell) / cell)});
    }
    start[dsize] = (int)tab.size();
}

// index / weight tables of the LINEAR and CUBIC paths along one axis
void taps_tab(int ssize, int dsize, int ksize, bool area_mode, bool clamp_linear, std::vector<int> &ofs,
              std::vector<short> &coef, int *xmax_out, std::vector<float> *fcoef = nullptr)
{
    const double inv = (double)dsize / ssize, scale = 1. / inv;
    const int ksize2 = ksize / 2;
    ofs.resize(dsize);
    coef.resize((size_t)dsize * ksize);
    if (fcoef)
        fcoef->resize((size_t)dsize * ksize);
    int xmax = dsize;
    float cbuf[8];
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
        if (clamp_linear) {        // the x axis: positions are clamped in the table (rows are clipped on use)
            if (sx < ksize2 - 1 && sx < 0 && ksize == 2)
                fx = 0, sx = 0;
            if (sx + ksize2 >= ssize) {
                xmax = d < xmax ? d : xmax;
                if (sx >= ssize - 1 && ksize == 2)
                    fx = 0, sx = ssize - 1;
            }
        }
        ofs[d] = sx;
        if (ksize == 2)
            cbuf[0] = 1.f - fx, cbuf[1] = fx;
        else if (ksize == 4)
            cubic_coeffs(fx, cbuf);
        else
            lanczos4_coeffs(fx, cbuf);
        for (int k = 0; k < ksize; k++) {
            coef[(size_t)d * ksize + k] = sat_short(cbuf[k] * 2048);
            if (fcoef)
                (*fcoef)[(size_t)d * ksize + k] = cbuf[k];
        }
    }
    if (xmax_out)
        *xmax_out = xmax;
}

}  // namespace

// scratch_bytes: size of the device buffer launch_resize_u8 needs for its tables
size_t resize_scratch_bytes(int sh, int sw, int dh, int dw)
{
    const size_t ints = (size_t)dw + dh + 4 + (size_t)dw + dh + 4;
    const size_t shorts = 16 * ((size_t)dw + dh);        // (up to 8 taps; float tables for float32 frames)
    const size_t tabs = 2 * ((size_t)sw + sh) + 2 * ((size_t)dw + dh) + 8;
    return ints * 4 + shorts * 2 + tabs * sizeof(DecAlpha) + 1024;
}

template <class T>
static int launch_resize(const T *src, T *dst, int n, int sh, int sw, int c, int dh, int dw, int mode,
                         void *scratch, hipStream_t st)
{
    constexpr bool F32 = sizeof(T) == 4;
    VA_REQUIRE(src && dst && scratch, "resize: NULL argument");
    VA_REQUIRE(n >= 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && c > 0 && c <= 4, "resize: bad shape");
    VA_REQUIRE(mode >= 0 && mode <= 4, "resize: interpolation must be 0 nearest, 1 linear, 2 cubic, 3 area or 4 lanczos4");
    VA_REQUIRE(dh <= 65535, "resize: target frames of more than 65535 rows are not supported");
    const size_t total = (size_t)n * dh * dw * c;
    if (total == 0)
        return VA_OK;
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
    // one launch covers up to 65535 frames (gridDim.z); longer batches go out in pieces (same tables)
    if (n > 65535) {
        for (int a = 0; a < n; a += 65535) {
            const int k = n - a < 65535 ? n - a : 65535;
            int rc = launch_resize<T>(src + (size_t)a * sh * sw * c, dst + (size_t)a * dh * dw * c, k, sh, sw, c, dh, dw, mode,
                                      scratch, st);
            if (rc)
                return rc;
        }
        return VA_OK;
    }
    const dim3 grid((unsigned)cdiv((long long)dw * c, kBlock), (unsigned)dh, (unsigned)n);
    // four samples per thread: single-channel uint8 frames, output rows of whole, aligned dwords
    const bool x4 = !F32 && c == 1 && dw % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0 &&
                    (long long)(dw / 4) * dh < (1ll << 31);
    const dim3 grid4((unsigned)cdiv((long long)(dw / 4) * dh, kBlock), 1u, (unsigned)n);
    const bool many_rows = (long long)n * dh * cdiv((long long)dw * c, kBlock) >= 65536;   // enough workgroups either way
    char *base = (char *)scratch;
    size_t used = 0;
    auto push = [&](const void *host, size_t bytes, void **dev) -> int {
        used = (used + 15) & ~(size_t)15;
        *dev = base + used;
        // (pageable source that dies with this call: a blocking copy -- complete on return; the destination is this
        //  call's own scratch, nothing on the stream uses it yet)
        VA_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
        used += bytes;
        return VA_OK;
    };
    int rc;
    if (mode == 0) {
        std::vector<int> xo(dw), yo(dh);
        const double ifx = 1. / inv_sx, ify = 1. / inv_sy;
        for (int x = 0; x < dw; x++) {
            const int sx = (int)floor(x * ifx);
            xo[x] = sx < sw - 1 ? sx : sw - 1;
        }
        for (int y = 0; y < dh; y++) {
            const int sy = (int)floor(y * ify);
            yo[y] = sy < sh - 1 ? sy : sh - 1;
        }
        void *dx, *dy;
        if ((rc = push(xo.data(), xo.size() * 4, &dx)) || (rc = push(yo.data(), yo.size() * 4, &dy)))
            return rc;
        if constexpr (!F32) {
            if (x4) {
                resize_nn_x4_kernel<<<grid4, kBlock, 0, st>>>(src, dst, (const int *)dx, (const int *)dy, sh, sw, dh, dw);
                VA_LAUNCH_CHECK("resize_nn_x4_kernel");
                return VA_OK;
            }
        }
        resize_nn_kernel<T><<<grid, kBlock, 0, st>>>(src, dst, (const int *)dx, (const int *)dy, sh, sw, c, dh, dw, total);
        VA_LAUNCH_CHECK("resize_nn_kernel");
        return VA_OK;
    }
    const int iscale_x = (int)nearbyint(scale_x), iscale_y = (int)nearbyint(scale_y);
    const bool area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 &&
                           fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (mode == 1 && area_fast && iscale_x == 2 && iscale_y == 2)
        mode = 3;
    if (mode == 3 && scale_x >= 1 && scale_y >= 1) {
        if (area_fast) {
            int wfull = (int)(sw / scale_x);
            wfull = wfull < dw ? wfull : dw;
            if constexpr (F32)
                resize_area_fast_f32_kernel<<<grid, kBlock, 0, st>>>(src, dst, iscale_x, iscale_y, wfull, sh, sw, c, dh,
                                                                    dw, total);
            else
                resize_area_fast_kernel<<<grid, kBlock, 0, st>>>(src, dst, iscale_x, iscale_y, wfull, sh, sw, c, dh,
                                                                dw, total);
            VA_LAUNCH_CHECK("resize_area_fast_kernel");
            return VA_OK;
        }
        std::vector<DecAlpha> xt, yt;
        std::vector<int> xs, ys;
        area_tab(sw, dw, scale_x, xt, xs);
        area_tab(sh, dh, scale_y, yt, ys);
        void *dxt, *dxs, *dyt, *dys;
        if ((rc = push(xt.data(), xt.size() * sizeof(DecAlpha), &dxt)) || (rc = push(xs.data(), xs.size() * 4, &dxs)) ||
            (rc = push(yt.data(), yt.size() * sizeof(DecAlpha), &dyt)) || (rc = push(ys.data(), ys.size() * 4, &dys)))
            return rc;
        resize_area_kernel<T><<<grid, kBlock, 0, st>>>(src, dst, (const DecAlpha *)dxt, (const int *)dxs,
                                                      (const DecAlpha *)dyt, (const int *)dys, sh, sw, c, dh, dw, total);
        VA_LAUNCH_CHECK("resize_area_kernel");
        return VA_OK;
    }
    const bool area_mode = mode == 3;
    const int ksize = mode == 2 ? 4 : (mode == 4 ? 8 : 2);
    std::vector<int> xo, yo;
    std::vector<short> ia, ib;
    std::vector<float> fa, fb;
    int xmax = dw;
    taps_tab(sw, dw, ksize, area_mode, true, xo, ia, &xmax, &fa);
    taps_tab(sh, dh, ksize, area_mode, false, yo, ib, nullptr, &fb);
    void *dxo, *dia, *dyo, *dib;
    if ((rc = push(xo.data(), xo.size() * 4, &dxo)) || (rc = push(yo.data(), yo.size() * 4, &dyo)))
        return rc;
    if constexpr (F32) {
        if ((rc = push(fa.data(), fa.size() * 4, &dia)) || (rc = push(fb.data(), fb.size() * 4, &dib)))
            return rc;
#define VA_F32_TAPS(KS)                                                                                              \
    resize_taps_f32_kernel<KS><<<grid, kBlock, 0, st>>>(src, dst, (const int *)dxo, (const float *)dia, (const int *)dyo, \
                                                       (const float *)dib, xmax, sh, sw, c, dh, dw, total)
        if (ksize == 2)
            VA_F32_TAPS(2);
        else if (ksize == 4)
            VA_F32_TAPS(4);
        else
            VA_F32_TAPS(8);
#undef VA_F32_TAPS
    } else {
        if ((rc = push(ia.data(), ia.size() * 2, &dia)) || (rc = push(ib.data(), ib.size() * 2, &dib)))
            return rc;
#define VA_U8_TAPS(KS)                                                                                               \
    do {                                                                                                             \
        if (many_rows)                                                                                               \
            resize_taps_kernel<KS, 4><<<dim3(grid.x, (unsigned)cdiv(dh, 4), grid.z), kBlock, 0, st>>>(                \
                src, dst, (const int *)dxo, (const short *)dia, (const int *)dyo, (const short *)dib, xmax, sh, sw, c, \
                dh, dw, total);                                                                                      \
        else                                                                                                         \
            resize_taps_kernel<KS, 1><<<grid, kBlock, 0, st>>>(src, dst, (const int *)dxo, (const short *)dia,        \
                                                              (const int *)dyo, (const short *)dib, xmax, sh, sw, c,  \
                                                              dh, dw, total);                                        \
    } while (0)
        if (ksize == 2)
            VA_U8_TAPS(2);
        else if (ksize == 4)
            VA_U8_TAPS(4);
        else
            VA_U8_TAPS(8);
#undef VA_U8_TAPS
    }
    VA_LAUNCH_CHECK("resize_taps_kernel");
    return VA_OK;
}

int launch_resize_u8(const uint8_t *src, uint8_t *dst, int n, int sh, int sw, int c, int dh, int dw, int mode,
                     void *scratch, hipStream_t st)
{
    return launch_resize<uint8_t>(src, dst, n, sh, sw, c, dh, dw, mode, scratch, st);
}

int launch_resize_f32(const float *src, float *dst, int n, int sh, int sw, int c, int dh, int dw, int mode,
                      void *scratch, hipStream_t st)
{
    return launch_resize<float>(src, dst, n, sh, sw, c, dh, dw, mode, scratch, st);
}

}  // namespace va
