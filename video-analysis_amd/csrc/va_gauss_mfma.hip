// va_gauss_mfma.hip -- single-channel 8-bit Gaussian blur (+ threshold + bit packing) on the
// CDNA4 matrix cores
//
// replaces cv2.GaussianBlur(frame.astype(np.uint8), (0, 0), sigma)
//          FilterBlur._process_frame, video/filters.py:388-392
// and, in the pipeline, the BUILD-DEFINED FilterThreshold that follows it.
//
// Why MFMA here: with sigma = 5 the blur is 31 + 31 multiply-adds per pixel, 6x more VALU issue
// time than the pixel's HBM time (va_gauss_fused.hip is that VALU formulation, kept for tap sets
// this kernel cannot take).  A 1-D convolution over a 32-pixel tile is a product with a banded
// Toeplitz matrix of the taps, 32 outputs x 64 inputs, which is exactly two
// v_mfma_i32_32x32x32_i8: the matrix pipe does 1024 MAC/clk/SIMD against 57 for v_dot4.
// All integer, so the result is bit-identical to the oracle's fixed-point definition.
//
// One WAVE owns a 32-column strip of one frame and marches down it 32 rows per step; waves do
// not talk to each other (no barriers after start-up):
//
//   global --16 B/lane, rows reflected (BORDER_REFLECT_101), prefetched one step ahead-->
//   registers: x ^ 0x80 (u8 -> i8, x' = x - 128) --> per-wave LDS tile, 32 rows x 64 columns
//   row pass : A = pixels (lane = image row, 16 consecutive bytes of its LDS row),
//              B = Toeplitz(taps) --> X[col on the lane][16 rows in registers] = sum w x'
//              (|X| <= 2^15: exactly an int16)
//   repack   : 16 x v_perm_b32: low bytes (^0x80 -> i8) and high bytes (already i8) of X become
//              two operand fragments -- the accumulator layout of one MFMA is the A-operand
//              layout of the next when the sum runs over the rows held in registers
//   col pass : D^T = X^T . Toeplitz(-taps), once for the high and once for the low bytes, over
//              the previous and the current row tile (64 input rows -> 32 output rows);
//              every additive constant of the u8->i8 shifts enters through the C operand
//   output   : t = (hi << 8) + lo = acc_min - 1 - acc, so "blur > thresh" is t's sign bit:
//              v_alignbit collects the 16 columns a lane holds, one shuffle joins the two half
//              rows, lanes 0..31 store one mask word each; u8 output: (acc + 2^15) >> 16.
//
// Exactness: with w_k the q8.8 taps (sum S <= 256, each <= 127) and x' = x - 128,
//   row:  sum w x = X + 128 S,  X = 256 hi + lo' + 128  (hi = X >> 8 signed, lo' = (X & 255) - 128)
//   col:  acc = sum w (X + 128 S) = 256 sum w hi + sum w lo' + (128 + 128 S) S
// i.e. the same integer the oracle accumulates; nothing is rounded before (acc + 2^15) >> 16.
#include "va_common.h"

namespace va {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef u4 u4_unaligned __attribute__((aligned(1)));

constexpr int kWaves = 4;          // waves (strips) per workgroup
constexpr int kLdsStride = 80;     // bytes per staged row: 64 + 16, stride/16 odd -> ds_read_b128
                                   // of 32 rows is bank-conflict free
constexpr int kTapTable = 128;     // zero-padded tap table, tap i at [48 + i]

struct MfmaTaps {
    int ksize;
    int sum;                       // sum of the taps (256 for every OpenCV tap set)
    int8_t w[36];
};

__device__ __forceinline__ uint32_t bswap32(uint32_t v)
{
    return __builtin_amdgcn_perm(0u, v, 0x00010203u);
}

template <bool HAS_DST, bool HAS_BITS>
__global__ __launch_bounds__(kWaves * 64, 4) void gauss_mfma_kernel(
    const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t *__restrict__ bits,
    int thresh, int h, int w, int w32, int nstrips, int blocks_per_frame, int nframes, MfmaTaps tp)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kWaves][32 * kLdsStride];
    __shared__ int8_t s_taps[kTapTable];

    // ---- start-up: zero-padded tap table in LDS (the only workgroup-wide step) -------------
    if (threadIdx.x < kTapTable) {
        const int e = (int)threadIdx.x - 48;
        int v = 0;
#pragma unroll
        for (int k = 0; k < 33; k++)
            v = (e == k && k < tp.ksize) ? (int)tp.w[k] : v;
        s_taps[threadIdx.x] = (int8_t)v;
    }
    __syncthreads();

    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs, so give every XCD
    // whole frames -- neighbouring strips then share their halo columns in that XCD's L2.
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int frame = (q / blocks_per_frame) * 8 + xcd;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int strip = (q % blocks_per_frame) * kWaves + wave;
    if (frame >= nframes || strip >= nstrips)
        return;

    const int R = tp.ksize >> 1;
    const int nn = lane & 31, hh = lane >> 5;
    const int x0 = strip * 32;

    // ---- Toeplitz operand fragments (constant for the whole kernel) -------------------------
    // position p = 0..63 inside the 64-wide window that starts 16 before the tile; output o:
    // weight = tap[p - 16 - o + R].  row pass: p = 32 q + 16 hh + j; column pass: byte j of the
    // repacked accumulator is tile row (j & 3) + 8 (j >> 2) + 4 hh.
    v4i brow[2], bcol[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            uint32_t wr = 0, wc = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int pr = 32 * ks + 16 * hh + 4 * g + b;
                const int pc = 32 * ks + b + 8 * g + 4 * hh;
                const int tr = s_taps[pr - 16 - nn + R + 48];
                const int tc = -(int)s_taps[pc - 16 - nn + R + 48];
                wr |= (uint32_t)(uint8_t)tr << (8 * b);
                wc |= (uint32_t)(uint8_t)tc << (8 * b);
            }
            brow[ks][g] = (int)wr;
            bcol[ks][g] = (int)wc;
        }

    // acc = 256 Yhi + Ylo + k0; the column MFMAs run on the negated taps and start the low
    // chain at c_lo, so that t = (nYhi << 8) + nYlo = acc_min - 1 - acc.
    const int k0 = (128 + 128 * tp.sum) * tp.sum;
    const int acc_min = thresh < 0 ? 0 : ((thresh + 1) << 16) - 32768;
    const int c_lo = acc_min - 1 - k0;
    const int k_dst = acc_min - 1 + 32768;   // (acc + 2^15) = k_dst - t

    const uint8_t *fsrc = src + (size_t)frame * h * w;
    uint8_t *stage = s_stage[wave];
    const int chunk = lane & 3, srow = lane >> 2;
    const int xc = x0 - 16 + 16 * chunk;                 // first column of this lane's chunk
    const bool border_strip = (x0 == 0) || (x0 + 48 > w);  // wave-uniform
    const bool mirrored = xc < 0 || xc >= w;             // whole chunk outside (w % 16 == 0)
    const int xs = xc < 0 ? -xc - 15 : (xc >= w ? 2 * w - 17 - xc : xc);
    const int ntiles = (h + 31) >> 5;                    // output tiles; row tiles 0..ntiles

    auto fetch = [&](int t, u4 &ga, u4 &gb) {
#pragma unroll
        for (int part = 0; part < 2; part++) {
            int y = 32 * t - 16 + srow + 16 * part;
            y = y < 0 ? -y : y;
            y = y >= h ? 2 * (h - 1) - y : y;
            y = min(max(y, 0), h - 1);
            const uint8_t *p = fsrc + (size_t)y * w + xs;
            u4 g;
            if (!border_strip) {
                g = *reinterpret_cast<const u4 *>(p);
            } else {
                g = *reinterpret_cast<const u4_unaligned *>(p);
                const u4 r = {bswap32(g.w), bswap32(g.z), bswap32(g.y), bswap32(g.x)};
                g = mirrored ? r : g;
            }
            (part ? gb : ga) = g;
        }
    };

    u4 ga, gb;
    fetch(0, ga, gb);
    v4i prev_hi = {}, prev_lo = {};

    for (int t = 0; t <= ntiles; t++) {
        // ---- stage row tile t (rows 32 t - 16 ...), prefetch the next one --------------------
        *reinterpret_cast<u4 *>(stage + srow * kLdsStride + 16 * chunk) = ga ^ 0x80808080u;
        *reinterpret_cast<u4 *>(stage + (srow + 16) * kLdsStride + 16 * chunk) = gb ^ 0x80808080u;
        if (t < ntiles)
            fetch(t + 1, ga, gb);
        __builtin_amdgcn_wave_barrier();
        const v4i a0 = *reinterpret_cast<const v4i *>(stage + nn * kLdsStride + 16 * hh);
        const v4i a1 = *reinterpret_cast<const v4i *>(stage + nn * kLdsStride + 32 + 16 * hh);
        __builtin_amdgcn_wave_barrier();

        // ---- row pass -------------------------------------------------------------------------
        v16i x = {};
        x = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, brow[0], x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, brow[1], x, 0, 0, 0);

        // ---- accumulator -> two i8 operand fragments -----------------------------------------
        v4i cur_hi, cur_lo;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)x[4 * g + 1], (uint32_t)x[4 * g], 0x05010400u);
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)x[4 * g + 3], (uint32_t)x[4 * g + 2], 0x05010400u);
            cur_lo[g] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
            cur_hi[g] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302u);
        }

        if (t > 0) {
            // ---- column pass for output rows 32 (t-1) ... -------------------------------------
            v16i yh = {}, yl;
#pragma unroll
            for (int i = 0; i < 16; i++)
                yl[i] = c_lo;
            yh = __builtin_amdgcn_mfma_i32_32x32x32_i8(prev_hi, bcol[0], yh, 0, 0, 0);
            yl = __builtin_amdgcn_mfma_i32_32x32x32_i8(prev_lo, bcol[0], yl, 0, 0, 0);
            yh = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur_hi, bcol[1], yh, 0, 0, 0);
            yl = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur_lo, bcol[1], yl, 0, 0, 0);

            // lane = output row nn (+ half hh), register i = column (i&3) + 8 (i>>2) + 4 hh
            const int y = 32 * (t - 1) + nn;
            uint32_t p = 0;
            int tv[16];
#pragma unroll
            for (int i = 15; i >= 0; i--) {
                tv[i] = (yh[i] << 8) + yl[i];
                if (HAS_BITS)
                    p = __builtin_amdgcn_alignbit(p, (uint32_t)tv[i], 31);
            }
            if (HAS_DST) {
                if (y < h) {
                    uint8_t *drow = dst + ((size_t)frame * h + y) * w + x0 + 4 * hh;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const uint32_t b0 = (uint32_t)(k_dst - tv[4 * g]) >> 16;
                        const uint32_t b1 = (uint32_t)(k_dst - tv[4 * g + 1]) >> 16;
                        const uint32_t b2 = (uint32_t)(k_dst - tv[4 * g + 2]) >> 16;
                        const uint32_t b3 = (uint32_t)(k_dst - tv[4 * g + 3]) >> 16;
                        if (x0 + 8 * g + 4 * hh < w)
                            *reinterpret_cast<uint32_t *>(drow + 8 * g) =
                                b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                    }
                }
            }
            if (HAS_BITS) {
                uint32_t wd = (p & 0xFu) | ((p & 0xF0u) << 4) | ((p & 0xF00u) << 8) | ((p & 0xF000u) << 12);
                wd <<= 4 * hh;
                wd |= (uint32_t)__shfl_xor((int)wd, 32, 64);
                const int valid = min(32, w - x0);
                if (valid < 32)
                    wd &= (1u << valid) - 1u;
                if (hh == 0 && y < h)
                    bits[((size_t)frame * h + y) * w32 + strip] = wd;
            }
        }
        prev_hi = cur_hi;
        prev_lo = cur_lo;
    }
}

}  // namespace

bool gauss_mfma_supported(int w, int h, const TapsQ8 &taps)
{
    if (w < 64 || h < 32 || (w % 16) != 0)   // single reflection; 16-byte chunks in or out
        return false;
    if (taps.ksize < 3 || taps.ksize / 2 > 16)
        return false;
    int sum = 0;
    for (int i = 0; i < taps.ksize; i++) {
        if (taps.t[i] > 127)                    // i8 operands
            return false;
        sum += taps.t[i];
    }
    return sum <= 256;                          // row sums must fit an int16
}

int launch_gauss_mfma_u8(const uint8_t *src, uint8_t *dst, uint32_t *bits, int thresh, int n,
                         int h, int w, const TapsQ8 &taps, hipStream_t st)
{
    VA_REQUIRE(src && (dst || bits), "mfma gaussian: no output requested");
    VA_REQUIRE(gauss_mfma_supported(w, h, taps), "mfma gaussian: unsupported kernel size %d",
               taps.ksize);
    VA_REQUIRE(reinterpret_cast<uintptr_t>(src) % 16 == 0 &&
                   (!dst || reinterpret_cast<uintptr_t>(dst) % 4 == 0),
               "mfma gaussian: the frame buffers must be 16-byte aligned");
    if (n == 0)
        return VA_OK;
    MfmaTaps tp = {};
    tp.ksize = taps.ksize;
    for (int i = 0; i < taps.ksize; i++) {
        tp.w[i] = (int8_t)taps.t[i];
        tp.sum += taps.t[i];
    }
    const int nstrips = cdiv(w, 32), w32 = words_per_row(w);
    const int bpf = cdiv(nstrips, kWaves);
    dim3 grid((unsigned)(8 * cdiv(n, 8) * bpf));
    if (dst && bits)
        gauss_mfma_kernel<true, true><<<grid, kWaves * 64, 0, st>>>(src, dst, bits, thresh, h, w, w32, nstrips, bpf, n, tp);
    else if (dst)
        gauss_mfma_kernel<true, false><<<grid, kWaves * 64, 0, st>>>(src, dst, bits, thresh, h, w, w32, nstrips, bpf, n, tp);
    else
        gauss_mfma_kernel<false, true><<<grid, kWaves * 64, 0, st>>>(src, dst, bits, thresh, h, w, w32, nstrips, bpf, n, tp);
    VA_LAUNCH_CHECK("gauss_mfma_kernel");
    return VA_OK;
}

}  // namespace va
