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
// A workgroup of four waves owns a 128-column strip of one frame and marches down it 32 rows per
// step; each wave computes one 32-column tile.  The input tile (32 rows x 160 columns with the
// halo) is loaded by the whole workgroup, whole cache lines per row (3 lines per 128 useful
// bytes; per-wave 64-byte windows moved 6):
//
//   global --16 B/lane, rows reflected (BORDER_REFLECT_101), prefetched one step ahead-->
//   registers: x ^ 0x80 (u8 -> i8, x' = x - 128) --> LDS tile (double buffered, one barrier/step)
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
//              v_alignbit collects the 16 columns a lane holds, v_permlane32_swap joins the two
//              half rows; u8 output: (acc + 2^15) >> 16.  Mask words and u8 tiles gather in LDS
//              and leave as 16-byte buffer stores (4-byte stores scattered over 32 rows per
//              instruction cost 30 % of the kernel although they add no HBM bytes).
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

constexpr int kWaves = 4;          // waves (32-column tiles) per workgroup
constexpr int kTileCols = kWaves * 32;
constexpr int kLdsStride = kTileCols + 32 + 16;   // staged row: strip + halo + pad; stride/16 odd
                                                  // -> ds_read_b128 of 32 rows is conflict free
static_assert((kLdsStride / 16) % 2 == 1, "LDS row stride must be an odd multiple of 16 bytes");
constexpr int kTapTable = 128;     // zero-padded tap table, tap i at [48 + i]
#ifndef GMF_DST_AUX
#define GMF_DST_AUX 0              // cache policy bits of the u8 output stores (2 = non-temporal)
#endif
constexpr int kFlushTiles = 8;     // mask words are written out every 8 output tiles
constexpr int kFlushRows = 32 * kFlushTiles;
static_assert(kFlushRows == kWaves * 64, "one thread per collected row at a flush");

struct MfmaTaps {
    int ksize;
    int sum;                       // sum of the taps (256 for every OpenCV tap set)
    int8_t w[36];
};

__device__ __forceinline__ uint32_t bswap32(uint32_t v)
{
    return __builtin_amdgcn_perm(0u, v, 0x00010203u);
}

// MASK8 (with HAS_DST): dst receives the thresholded mask as bytes, maxval where blur > thresh and 0
// elsewhere (the chain that ends at FilterThreshold's uint8 mask: no bit mask, no unpack pass)
template <bool HAS_DST, bool HAS_BITS, bool MASK8 = false>
__global__ __launch_bounds__(kWaves * 64, 4) void gauss_mfma_kernel(
    const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t *__restrict__ bits,
    int thresh, int h, int w, int w32, int nstrips, int blocks_per_frame, int nframes, MfmaTaps tp,
    int maxval = 255)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[2][32 * kLdsStride];
    __shared__ int8_t s_taps[kTapTable];
    // outputs are collected in LDS and leave as 16-byte pieces: four-byte stores scattered over 32
    // rows per instruction cost this kernel 30 % (measured), although they add no HBM bytes
    __shared__ __attribute__((aligned(16))) uint32_t s_bits[2][kFlushRows][kWaves];        // HAS_BITS
    // (row stride 33 words: the epilogue's word stores -- 32 lanes, one per tile row, same column --
    //  fall on 32 different banks; with a 128-byte stride they all hit ONE bank, 32 LDS cycles per
    //  store, which made the u8 output cost as much as the whole rest of the kernel)
    constexpr int kDstStride = kTileCols / 4 + 1;
    __shared__ uint32_t s_dst[HAS_DST ? 2 : 1][HAS_DST ? 32 : 1][kDstStride];

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
    if (frame >= nframes)                                // whole workgroup
        return;
    const bool active = strip < nstrips;                 // waves right of the frame only load

    const int xb = (q % blocks_per_frame) * kTileCols;   // first column of the workgroup's strip
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
    const int ntiles = (h + 31) >> 5;                    // output tiles; row tiles 0..ntiles

    // ---- cooperative loads: 32 rows x 10 chunks of 16 B.  load A: thread -> (row tid/8, chunk
    //      tid%8): eight lanes per row read 128 contiguous bytes; load B: the last two chunks of
    //      every row, issued by every wave alike (identical bytes land on identical LDS
    //      addresses) so that all waves have the same number of loads in flight.
    const int tid = threadIdx.x;
    const int row_a = tid >> 3, ch_a = tid & 7;
    const int row_b = lane >> 1, ch_b = 8 + (lane & 1);
    auto chunk_src = [&](int ch, bool &mir) {            // source column of a chunk (w % 16 == 0)
        const int xc = xb - 16 + 16 * ch;
        mir = xc < 0 || xc >= w;
        const int xs = xc < 0 ? -xc - 15 : (xc >= w ? 2 * w - 17 - xc : xc);
        return min(max(xs, 0), w - 16);                  // far outside: never used, keep in range
    };
    bool mir_a, mir_b;
    const int xs_a = chunk_src(ch_a, mir_a), xs_b = chunk_src(ch_b, mir_b);
    const bool border_block = (xb == 0) || (xb + kTileCols + 16 > w);   // workgroup-uniform

    auto row_of = [&](int t, int r) {                     // BORDER_REFLECT_101, then clamp
        int y = 32 * t - 16 + r;
        y = y < 0 ? -y : y;
        y = y >= h ? 2 * (h - 1) - y : y;
        return min(max(y, 0), h - 1);
    };
    auto fetch = [&](int t, u4 &ga, u4 &gb) {
        const uint8_t *pa, *pb;
        if (t >= 1 && 32 * t + 16 <= h) {                // tile inside the frame: add-only
            const uint8_t *base = fsrc + (size_t)(32 * t - 16) * w;
            pa = base + (size_t)row_a * w + xs_a;
            pb = base + (size_t)row_b * w + xs_b;
        } else {
            pa = fsrc + (size_t)row_of(t, row_a) * w + xs_a;
            pb = fsrc + (size_t)row_of(t, row_b) * w + xs_b;
        }
        // straight-line loads (aligned except for mirrored border chunks)
        ga = *reinterpret_cast<const u4_unaligned *>(pa);
        if (wave == 0)
            gb = *reinterpret_cast<const u4_unaligned *>(pb);
    };
    auto stage_write = [&](uint8_t *buf, u4 ga, u4 gb) {
        if (border_block) {                               // uniform; VALU only
            const u4 ra = {bswap32(ga.w), bswap32(ga.z), bswap32(ga.y), bswap32(ga.x)};
            const u4 rb = {bswap32(gb.w), bswap32(gb.z), bswap32(gb.y), bswap32(gb.x)};
            ga = mir_a ? ra : ga;
            gb = mir_b ? rb : gb;
        }
        *reinterpret_cast<u4 *>(buf + row_a * kLdsStride + 16 * ch_a) = ga ^ 0x80808080u;
        if (wave == 0)
            *reinterpret_cast<u4 *>(buf + row_b * kLdsStride + 16 * ch_b) = gb ^ 0x80808080u;
    };

    // sixteen registers that stay c_lo: the C operand of the low-byte chain
    v16i c_init;
#pragma unroll
    for (int i = 0; i < 16; i++)
        c_init[i] = c_lo;
    asm volatile("" : "+v"(c_init));

    // Outputs go through buffer stores (a lane that must not store gets an out-of-range offset,
    // which the hardware drops: no branch, so the waits can count the stores in flight).
    const uint32_t kOob = 0xFFFFFFFFu;
    __amdgpu_buffer_rsrc_t bits_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        HAS_BITS ? (void *)(bits + (size_t)frame * h * w32) : nullptr, 0, h * w32 * 4, 0x00027000);
    __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        HAS_DST ? (void *)(dst + (size_t)frame * h * w) : nullptr, 0, h * w, 0x00027000);
    const int xbw = xb >> 5;                              // first mask word of the workgroup's strip
    const int valid = max(0, min(32, w - x0));
    const uint32_t colmask = valid < 32 ? (1u << valid) - 1u : ~0u;
    const int frag_off = nn * kLdsStride + 32 * wave + 16 * hh;           // this lane's A bytes

    u4 ga, gb = {};
    fetch(0, ga, gb);
    stage_write(s_stage[0], ga, gb);
    fetch(1, ga, gb);                                     // ntiles >= 1
    __syncthreads();
    v4i prev_hi = {}, prev_lo = {};
    v16i yh = {}, yl = {};

    // Software pipeline, one barrier per step.  Step t:
    //   row MFMAs of tile t  ||  epilogue of the column sums of output tile t-2 (issued last step)
    //   stage tile t+1, prefetch tile t+2
    //   repack tile t, issue the column MFMAs of output tile t-1 (consumed next step)
    // so that neither MFMA result is waited for.  Steps 0, 1 and ntiles+1 run the same code on
    // rows outside the frame: their stores are masked and their loads clamped.
    for (int t = 0; t <= ntiles + 1; t++) {
        const uint8_t *cur_buf = s_stage[t & 1];
        uint8_t *next_buf = s_stage[(t + 1) & 1];
        // ---- row pass of tile t (rows 32 t - 16 ...) ---------------------------------------------
        const v4i a0 = *reinterpret_cast<const v4i *>(cur_buf + frag_off);
        const v4i a1 = *reinterpret_cast<const v4i *>(cur_buf + frag_off + 32);
        v16i x = {};
        // (u8 outputs: the tile the previous step's epilogue left in s_dst is read now and stored
        //  further down, so that neither the LDS read nor the store sits at the end of a step)
        v4i vout = {};
        if (HAS_DST)
#pragma unroll
            for (int j = 0; j < 4; j++)                   // (four word reads: conflict free at this stride)
                vout[j] = (int)s_dst[(t + 1) & 1][tid >> 3][(tid & 7) * 4 + j];
        x = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, brow[0], x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, brow[1], x, 0, 0, 0);

        // ---- epilogue of output tile t-2: lane = output row nn (+ half hh), register i = column
        //      (i&3) + 8 (i>>2) + 4 hh ---------------------------------------------------------------
        {
            uint32_t p = 0;
            int tv[16];
#pragma unroll
            for (int i = 15; i >= 0; i--) {
                tv[i] = (yh[i] << 8) + yl[i];
                if (HAS_BITS)
                    p = __builtin_amdgcn_alignbit(p, (uint32_t)tv[i], 31);
            }
            const int u = t - 2;                               // output tile of this epilogue
#ifndef GMF_NO_DST_LDS                                    // (ablation build: no byte conversion, no LDS gather)
            if (HAS_DST) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    if (MASK8) {
                        // |t| < 2^24, so t's top byte is 0xFF where t < 0 (blur > thresh) and 0 elsewhere:
                        // three byte permutes collect four of them, one AND turns 0xFF into maxval
                        const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)tv[4 * g + 1], (uint32_t)tv[4 * g], 0x0c0c0703u);
                        const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)tv[4 * g + 3], (uint32_t)tv[4 * g + 2], 0x07030c0cu);
                        s_dst[u & 1][nn][8 * wave + 2 * g + hh] = (t01 | t23) & (0x01010101u * (uint32_t)maxval);
                    } else {
                        // u8 blur: (acc + 2^15) >> 16, and acc + 2^15 = k_dst - t < 2^24: the result is byte 2
                        const uint32_t d0 = (uint32_t)(k_dst - tv[4 * g]), d1 = (uint32_t)(k_dst - tv[4 * g + 1]);
                        const uint32_t d2 = (uint32_t)(k_dst - tv[4 * g + 2]), d3 = (uint32_t)(k_dst - tv[4 * g + 3]);
                        s_dst[u & 1][nn][8 * wave + 2 * g + hh] =
                            __builtin_amdgcn_perm(d1, d0, 0x0c0c0602u) | __builtin_amdgcn_perm(d3, d2, 0x06020c0cu);
                    }
                }
            }
#endif
            if (HAS_BITS) {
                uint32_t wd = (p & 0xFu) | ((p & 0xF0u) << 4) | ((p & 0xF00u) << 8) | ((p & 0xF000u) << 12);
                wd <<= 4 * hh;
                // lanes 0..31 receive the word of lane + 32 (v_permlane32_swap, no LDS trip)
                const auto sw = __builtin_amdgcn_permlane32_swap(wd, wd, false, false);
                if (hh == 0 && u >= 0)
                    s_bits[(u / kFlushTiles) & 1][(u % kFlushTiles) * 32 + nn][wave] =
                        active ? (wd | sw[1]) & colmask : 0u;
            }
        }

        // ---- stage tile t+1 into the other buffer (nobody reads it before the barrier at the end
        //      of this step; its previous readers passed the last barrier), prefetch tile t+2 ------
        stage_write(next_buf, ga, gb);
        fetch(min(t + 2, ntiles), ga, gb);               // (a second register set, two steps ahead: 0.175 -> 0.188 ms)
        if (HAS_DST) {                                    // output tile t-3 (after this step's loads: vmcnt retires in order)
            const int us = t - 3, r = tid >> 3, c16 = (tid & 7) * 16, y = 32 * us + r;
#ifdef GMF_NO_DST_STORE                                   // ablation build: every u8 store dropped by the hardware
            const bool ok = false;
#else
            const bool ok = us >= 0 && y < h && xb + c16 < w;            // w % 16 == 0
#endif
            __builtin_amdgcn_raw_buffer_store_b128(vout, dst_rsrc, ok ? (uint32_t)(y * w + xb + c16) : kOob, 0, GMF_DST_AUX);
        }

        // ---- accumulator -> two i8 operand fragments -----------------------------------------
        v4i cur_hi, cur_lo;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)x[4 * g + 1], (uint32_t)x[4 * g], 0x05010400u);
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)x[4 * g + 3], (uint32_t)x[4 * g + 2], 0x05010400u);
            cur_lo[g] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
            cur_hi[g] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302u);
        }

        // ---- column pass for output rows 32 (t-1) ..., consumed by the next step ----------------
        yh = __builtin_amdgcn_mfma_i32_32x32x32_i8(prev_hi, bcol[0], v16i{}, 0, 0, 0);
        yl = __builtin_amdgcn_mfma_i32_32x32x32_i8(prev_lo, bcol[0], c_init, 0, 0, 0);
        yh = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur_hi, bcol[1], yh, 0, 0, 0);
        yl = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur_lo, bcol[1], yl, 0, 0, 0);
        prev_hi = cur_hi;
        prev_lo = cur_lo;
        lds_barrier();   // (not __syncthreads(): the fetch of tile t + 2 and the stores stay in flight)

        // ---- what the epilogues collected leaves as 16-byte pieces (other buffer than the next
        //      epilogue writes) ----------------------------------------------------------------------
        const int u = t - 2;
        if (HAS_BITS && u >= 0 && (u % kFlushTiles == kFlushTiles - 1 || u == ntiles - 1)) {
            const int g = u / kFlushTiles, y = g * kFlushRows + tid;
            const bool ok = y < h && tid < (u % kFlushTiles + 1) * 32;
            const v4i v = *reinterpret_cast<const v4i *>(&s_bits[g & 1][tid][0]);
            const uint32_t off = (uint32_t)(y * w32 + xbw) * 4u;
            if (xbw + kWaves <= w32) {                                   // whole strip inside the row
                __builtin_amdgcn_raw_buffer_store_b128(v, bits_rsrc, ok ? off : kOob, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < kWaves; k++)
                    __builtin_amdgcn_raw_buffer_store_b32(v[k], bits_rsrc,
                                                          ok && xbw + k < w32 ? off + 4 * k : kOob, 0, 0);
            }
        }
    }
    if (HAS_DST) {                                        // the last output tile is still in s_dst
        const int us = ntiles - 1, r = tid >> 3, c16 = (tid & 7) * 16, y = 32 * us + r;
        const bool ok = y < h && xb + c16 < w;
        v4i v;
#pragma unroll
        for (int j = 0; j < 4; j++)
            v[j] = (int)s_dst[us & 1][r][(tid & 7) * 4 + j];
        __builtin_amdgcn_raw_buffer_store_b128(v, dst_rsrc, ok ? (uint32_t)(y * w + xb + c16) : kOob, 0, 0);
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
                         int h, int w, const TapsQ8 &taps, hipStream_t st, int mask8_maxval)
{
    VA_REQUIRE(src && (dst || bits), "mfma gaussian: no output requested");
    VA_REQUIRE(mask8_maxval <= 0 || (dst && !bits && thresh >= 0 && mask8_maxval <= 255),
               "mfma gaussian: the byte-mask form writes dst only and needs a threshold");
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
    const int bpf = cdiv(w, kTileCols);
    dim3 grid((unsigned)(8 * cdiv(n, 8) * bpf));
    if (mask8_maxval > 0)
        gauss_mfma_kernel<true, false, true><<<grid, kWaves * 64, 0, st>>>(src, dst, bits, thresh, h, w, w32, nstrips, bpf, n, tp,
                                                                            mask8_maxval);
    else if (dst && bits)
        gauss_mfma_kernel<true, true><<<grid, kWaves * 64, 0, st>>>(src, dst, bits, thresh, h, w, w32, nstrips, bpf, n, tp);
    else if (dst)
        gauss_mfma_kernel<true, false><<<grid, kWaves * 64, 0, st>>>(src, dst, bits, thresh, h, w, w32, nstrips, bpf, n, tp);
    else
        gauss_mfma_kernel<false, true><<<grid, kWaves * 64, 0, st>>>(src, dst, bits, thresh, h, w, w32, nstrips, bpf, n, tp);
    VA_LAUNCH_CHECK("gauss_mfma_kernel");
    return VA_OK;
}

}  // namespace va
