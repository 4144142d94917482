// va_morph.hip -- erosion / dilation (A6) on u8 images and on bit-packed masks
//
// replaces cv2.erode / cv2.dilate(img, cv2.getStructuringElement(shape, (k, k)))
//          mask_thinning, video/analysis/image.py:248-251
// OpenCV semantics: anchor = (k/2, k/2); pixels outside the image never win (default
// morphologyDefaultBorderValue()).  RECT / CROSS / ELLIPSE are row-convex, so an element is a
// list of per-row column spans (RowSpans).
//
// The pipeline keeps masks bit-packed (32 px per dword): a 1080p mask is 259 KB instead of
// 2 MB, horizontal neighbourhoods are shifts/ORs of three adjacent words and a 5x5 operation
// touches 15 words per 32 output pixels.
#include <math.h>

#include "va_common.h"

namespace va {

int make_row_spans(int shape, int ksize, RowSpans *out)
{
    VA_REQUIRE(ksize >= 1, "morphology: ksize must be >= 1 (got %d)", ksize);
    if (ksize > 63) {
        set_error("morphology: ksize %d > 63 is not supported", ksize);
        return VA_ERR_RANGE;
    }
    VA_REQUIRE(shape >= VA_SHAPE_RECT && shape <= VA_SHAPE_ELLIPSE, "morphology: bad shape %d",
               shape);
    out->ksize = ksize;
    out->anchor = ksize / 2;
    const int r = ksize / 2, c = ksize / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < ksize; i++) {
        int j1 = 0, j2 = 0;
        if (shape == VA_SHAPE_RECT || (shape == VA_SHAPE_CROSS && i == r)) {
            j2 = ksize;
        } else if (shape == VA_SHAPE_CROSS) {
            j1 = c;
            j2 = j1 + 1;
        } else {  // cv::getStructuringElement(MORPH_ELLIPSE)
            int dy = i - r;
            if (abs(dy) <= r) {
                int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
                j1 = c - dx > 0 ? c - dx : 0;
                j2 = c + dx + 1 < ksize ? c + dx + 1 : ksize;
            }
        }
        out->lo[i] = (int8_t)j1;
        out->hi[i] = (int8_t)j2;
    }
    return VA_OK;
}

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------ u8 <-> bit masks
__global__ void __launch_bounds__(kBlock)
pack_bits_kernel(const uint8_t *__restrict__ src, uint32_t *__restrict__ bits, int w, int w32,
                 size_t total_words, int thresh, int vec_ok)
{
    size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total_words)
        return;
    int wi = (int)(t % w32);
    size_t rowi = t / w32;
    const uint8_t *p = src + rowi * (size_t)w + (size_t)wi * 32;
    int nvalid = w - wi * 32;
    nvalid = nvalid > 32 ? 32 : nvalid;
    uint32_t m = 0;
    if (vec_ok && nvalid == 32) {
        uint4 a = reinterpret_cast<const uint4 *>(p)[0];
        uint4 b = reinterpret_cast<const uint4 *>(p)[1];
        uint8_t px[32];
        memcpy(px, &a, 16);
        memcpy(px + 16, &b, 16);
#pragma unroll
        for (int k = 0; k < 32; k++)
            m |= (uint32_t)((int)px[k] > thresh) << k;
    } else {
        for (int k = 0; k < nvalid; k++)
            m |= (uint32_t)((int)p[k] > thresh) << k;
    }
    bits[t] = m;
}

// rows of whole words, 16-byte aligned output: one thread = 16 pixels (half a word), so that the lanes of a
// wave write 1 KB of consecutive bytes per store instruction (one word per thread left every 128-byte line
// half written by each of its two stores: 0.136 ms per 256 x 1080p; this form: see DESIGN.md)
__global__ void __launch_bounds__(kBlock)
unpack_bits_wide_kernel(const uint32_t *__restrict__ bits, uint8_t *__restrict__ dst, size_t total_halves,
                        int maxval)
{
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total_halves)
        return;
    const uint32_t m = (bits[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    v4u v;
#pragma unroll
    for (int q = 0; q < 4; q++)      // four bits -> four bytes of 0 / 1 (bit k lands in byte k), times maxval
        v[q] = ((((m >> (4 * q)) & 0xFu) * 0x00204081u) & 0x01010101u) * (uint32_t)maxval;
    __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(dst + 16 * t));
}

__global__ void __launch_bounds__(kBlock)
unpack_bits_kernel(const uint32_t *__restrict__ bits, uint8_t *__restrict__ dst, int w, int w32,
                   size_t total_words, int maxval, int vec_ok)
{
    size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total_words)
        return;
    int wi = (int)(t % w32);
    size_t rowi = t / w32;
    uint8_t *p = dst + rowi * (size_t)w + (size_t)wi * 32;
    int nvalid = w - wi * 32;
    nvalid = nvalid > 32 ? 32 : nvalid;
    uint32_t m = bits[t];
    if (vec_ok && nvalid == 32) {
        uint8_t px[32];
#pragma unroll
        for (int k = 0; k < 32; k++)
            px[k] = (m >> k) & 1u ? (uint8_t)maxval : (uint8_t)0;
        uint4 a, b;
        memcpy(&a, px, 16);
        memcpy(&b, px + 16, 16);
        reinterpret_cast<uint4 *>(p)[0] = a;
        reinterpret_cast<uint4 *>(p)[1] = b;
    } else {
        for (int k = 0; k < nvalid; k++)
            p[k] = (m >> k) & 1u ? (uint8_t)maxval : (uint8_t)0;
    }
}

// ------------------------------------------------------------------ morphology on bit masks
// one thread = one output word (32 pixels).  ERODE = AND over the element, DILATE = OR.
template <bool DILATE>
__global__ void __launch_bounds__(kBlock)
morph_bits_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int h, int w,
                  int w32, size_t total_words, RowSpans se)
{
    size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total_words)
        return;
    const int wi = (int)(t % w32);
    const size_t rowi = t / w32;
    const int y = (int)(rowi % h);
    const uint32_t *frame = src + (rowi - y) * (size_t)w32;
    const uint32_t border = DILATE ? 0u : 0xFFFFFFFFu;
    const int tail = w & 31;  // valid bits in the last word (0 = all 32)
    const uint32_t last_valid = tail ? (0xFFFFFFFFu >> (32 - tail)) : 0xFFFFFFFFu;
    uint32_t acc = border;
    for (int i = 0; i < se.ksize; i++) {
        const int yy = y + i - se.anchor;
        if (yy < 0 || yy >= h)
            continue;
        const int lo = se.lo[i] - se.anchor, hi = se.hi[i] - 1 - se.anchor;  // dx range
        if (lo > hi)
            continue;
        const uint32_t *row = frame + (size_t)yy * w32;
        uint32_t c = row[wi];
        uint32_t l = wi > 0 ? row[wi - 1] : border;
        uint32_t r = wi + 1 < w32 ? row[wi + 1] : border;
        if (!DILATE) {  // pixels right of the image edge must not win the minimum
            if (wi == w32 - 1)
                c |= ~last_valid;
            if (wi + 1 == w32 - 1)
                r |= ~last_valid;
        }
        uint32_t racc = border;
        for (int dx = lo; dx <= hi; dx++) {
            uint32_t s;  // s[x] = src[x + dx]
            if (dx == 0)
                s = c;
            else if (dx > 0)
                s = (c >> dx) | (r << (32 - dx));
            else
                s = (c << -dx) | (l >> (32 + dx));
            racc = DILATE ? (racc | s) : (racc & s);
        }
        acc = DILATE ? (acc | racc) : (acc & racc);
    }
    if (wi == w32 - 1)
        acc &= last_valid;
    dst[t] = acc;
}

// ------------------------------------------------------------------ fused op sequence on bits
// The pipeline's whole morphology chain (e.g. 5x5 dilate then 5x5 erode) in one kernel: a block
// owns a band of kBandRows rows of one frame, stages the band plus the chain's total vertical
// reach in LDS and applies the ops LDS -> LDS; only the final band is written.
struct MorphSeq {
    int count;
    int reach_total;            // sum over ops of max(anchor, ksize-1-anchor)
    int op[VA_MAX_MORPH_OPS];   // VA_MORPH_*
    RowSpans se[VA_MAX_MORPH_OPS];
};
constexpr int kBandRows = 32;

__device__ __forceinline__ uint32_t morph_word_lds(const uint32_t *buf, int rbase, int y, int wi,
                                                  int h, int w32, uint32_t last_valid,
                                                  const RowSpans &se, bool dilate)
{
    const uint32_t border = dilate ? 0u : 0xFFFFFFFFu;
    uint32_t acc = border;
    for (int i = 0; i < se.ksize; i++) {
        const int yy = y + i - se.anchor;
        if (yy < 0 || yy >= h)
            continue;
        const int lo = se.lo[i] - se.anchor, hi = se.hi[i] - 1 - se.anchor;
        if (lo > hi)
            continue;
        const uint32_t *row = buf + (size_t)(yy - rbase) * w32;
        uint32_t c = row[wi];
        uint32_t l = wi > 0 ? row[wi - 1] : border;
        uint32_t r = wi + 1 < w32 ? row[wi + 1] : border;
        if (!dilate) {
            if (wi == w32 - 1)
                c |= ~last_valid;
            if (wi + 1 == w32 - 1)
                r |= ~last_valid;
        }
        uint32_t racc = border;
        for (int dx = lo; dx <= hi; dx++) {
            uint32_t sft;
            if (dx == 0)
                sft = c;
            else if (dx > 0)
                sft = (c >> dx) | (r << (32 - dx));
            else
                sft = (c << -dx) | (l >> (32 + dx));
            racc = dilate ? (racc | sft) : (racc & sft);
        }
        acc = dilate ? (acc | racc) : (acc & racc);
    }
    if (wi == w32 - 1)
        acc &= last_valid;
    return acc;
}

// result[x] = src[x + dx] for |dx| <= 31, from the word and its two neighbours (funnel shift)
__device__ __forceinline__ uint32_t shift_words(uint32_t l, uint32_t c, uint32_t r, int dx)
{
    if (dx == 0)
        return c;
    return dx > 0 ? __builtin_amdgcn_alignbit(r, c, dx) : __builtin_amdgcn_alignbit(c, l, 32 + dx);
}

// thread layout: 64 word columns x 4 rows; loops stride over rows / column chunks (no div/mod)
__global__ void __launch_bounds__(kBlock)
morph_fused_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int h, int w, int w32, int nbands, MorphSeq seq)
{
    extern __shared__ uint32_t s_rows[];   // 3 x (kBandRows + 2*reach_total) x w32
    const int band = blockIdx.x % nbands, f = blockIdx.x / nbands;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int y0 = band * kBandRows;
    const int nrows = kBandRows + 2 * seq.reach_total;
    const int rbase = y0 - seq.reach_total;            // frame row of LDS row 0
    uint32_t *bufA = s_rows, *bufB = s_rows + (size_t)nrows * w32, *bufH = bufB + (size_t)nrows * w32;
    const uint32_t *fsrc = src + (size_t)f * h * w32;
    const int tail = w & 31;
    const uint32_t last_valid = tail ? (0xFFFFFFFFu >> (32 - tail)) : 0xFFFFFFFFu;

    // stage the band: 8 independent loads in flight per thread before the first LDS store
    // (a plain load->store loop exposes one full memory latency per row)
    for (int wi = tx; wi < w32; wi += 64) {
        for (int r0 = ty; r0 < nrows; r0 += 32) {
            uint32_t tmp[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = r0 + 4 * j, y = rbase + r;
                tmp[j] = (r < nrows && y >= 0 && y < h) ? fsrc[(size_t)y * w32 + wi] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = r0 + 4 * j;
                if (r < nrows)
                    bufA[r * w32 + wi] = tmp[j];
            }
        }
    }
    __syncthreads();

    int reach = 0;
    for (int k = 0; k < seq.count; k++) {
        const RowSpans &se = seq.se[k];
        const int a = se.anchor, b = se.ksize - 1 - se.anchor;
        reach += a > b ? a : b;
        const int lo = max(0, rbase + reach), hi = min(h, rbase + nrows - reach);  // rows to produce
        const bool dil = seq.op[k] == VA_MORPH_DILATE;
        const uint32_t border = dil ? 0u : 0xFFFFFFFFu;
        bool rect = true;                       // same non-empty span in every element row
        for (int i = 1; i < se.ksize; i++)
            rect = rect && se.lo[i] == se.lo[0] && se.hi[i] == se.hi[0];
        rect = rect && se.lo[0] < se.hi[0];
        if (rect) {
            // separable: horizontal pass over every input row the band needs, then vertical
            const int dlo = se.lo[0] - a, dhi = se.hi[0] - 1 - a;
            const int ylo = max(0, lo - a), yhi = min(h, hi + b);
            for (int y = ylo + ty; y < yhi; y += 4) {
                const uint32_t *row = bufA + (y - rbase) * w32;
                for (int wi = tx; wi < w32; wi += 64) {
                    uint32_t c = row[wi];
                    uint32_t l = wi > 0 ? row[wi - 1] : border;
                    uint32_t r = wi + 1 < w32 ? row[wi + 1] : border;
                    if (!dil) {     // pixels right of the frame must not win the minimum
                        if (wi == w32 - 1)
                            c |= ~last_valid;
                        if (wi + 1 == w32 - 1)
                            r |= ~last_valid;
                    }
                    uint32_t racc = border;
                    for (int dx = dlo; dx <= dhi; dx++) {
                        const uint32_t sft = shift_words(l, c, r, dx);
                        racc = dil ? (racc | sft) : (racc & sft);
                    }
                    bufH[(y - rbase) * w32 + wi] = racc;
                }
            }
            __syncthreads();
            for (int y = lo + ty; y < hi; y += 4) {
                const int i0 = max(0, a - y), i1 = min(se.ksize, h + a - y);   // rows inside the frame
                for (int wi = tx; wi < w32; wi += 64) {
                    uint32_t acc = border;
                    for (int i = i0; i < i1; i++) {
                        const uint32_t v = bufH[(y + i - a - rbase) * w32 + wi];
                        acc = dil ? (acc | v) : (acc & v);
                    }
                    if (wi == w32 - 1)
                        acc &= last_valid;
                    bufB[(y - rbase) * w32 + wi] = acc;
                }
            }
        } else {
            for (int y = lo + ty; y < hi; y += 4)
                for (int wi = tx; wi < w32; wi += 64)
                    bufB[(y - rbase) * w32 + wi] =
                        morph_word_lds(bufA, rbase, y, wi, h, w32, last_valid, se, dil);
        }
        __syncthreads();
        uint32_t *t = bufA;
        bufA = bufB;
        bufB = t;
    }

    // write the band
    const int yend = min(h, y0 + kBandRows);
    uint32_t *fdst = dst + (size_t)f * h * w32;
    for (int y = y0 + ty; y < yend; y += 4) {
        const uint32_t *row = bufA + (y - rbase) * w32;
        for (int wi = tx; wi < w32; wi += 64)
            fdst[(size_t)y * w32 + wi] = row[wi];
    }
}

// ------------------------------------------------------------------ register-streaming rect ops
// Fast path for the common chains (one or two odd-sized RECT ops, k <= 7; e.g. the 5x5 closing
// of the benchmark): no LDS, no barriers.  A wave marches down a band of rows with lane <-> word
// column; a k x k rectangle is a horizontal OR of funnel-shifted words followed by a vertical OR
// over a k-row register window.  Erosion is dilation of the complement (pixels outside the frame
// are 0 in both pictures, which is exactly OpenCV's "outside never wins" border), so every op is
// an OR-dilation between runtime complement masks.  The second op takes its left/right
// neighbour words from the adjacent lanes, which costs halo lanes on either side: a wave stores 60
// word columns (62 for a single op).
constexpr int kStreamBand = 32;

template <int K>
__device__ __forceinline__ uint32_t hdilate(uint32_t l, uint32_t c, uint32_t r)
{
    uint32_t acc = c;
#pragma unroll
    for (int d = 1; d <= K / 2; d++) {
        acc |= __builtin_amdgcn_alignbit(r, c, d);        // src[x + d]
        acc |= __builtin_amdgcn_alignbit(c, l, 32 - d);   // src[x - d]
    }
    return acc;
}

// One load per row and lane -- the left/right words come from the neighbouring lanes by DPP wave
// shifts -- and eight rows of loads in flight (a row-by-row march pays one memory round trip per
// row: 0.081 against 0.036 ms for the benchmark's closing).
constexpr int kStreamGroup = 8;

__device__ __forceinline__ uint32_t lane_left(uint32_t v)    // value of lane - 1 (0 into lane 0)
{
    return __builtin_amdgcn_update_dpp(0u, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t lane_right(uint32_t v)   // value of lane + 1 (0 into lane 63)
{
    return __builtin_amdgcn_update_dpp(0u, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}

template <int K0, int K1>
__global__ void __launch_bounds__(kBlock)
morph_stream_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int n, int h, int w,
                     int w32, int nbands, int nchunks, uint32_t flip_in, uint32_t flip_mid,
                     uint32_t flip_out)
{
    constexpr int A0 = K0 / 2, A1 = K1 / 2;
    // lanes [LO, LO+COLS) store; the others only supply neighbour words (two per side for two ops)
    constexpr int LO = K1 ? 2 : 1;
    constexpr int COLS = K1 ? 60 : 62;
    const int lane = threadIdx.x & 63;
    size_t item = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);   // (frame, band, chunk)
    const int chunk = (int)(item % nchunks);
    item /= nchunks;
    const int band = (int)(item % nbands);
    const size_t f = item / nbands;
    if (f >= (size_t)n)
        return;   // surplus waves of the last block (wave-uniform)
    const int wi = chunk * COLS + lane - LO;             // this lane's word column
    const bool col_in = wi >= 0 && wi < w32;
    const bool col_store = col_in && lane >= LO && lane < LO + COLS;
    const int tail = w & 31;
    const uint32_t valid = !col_in ? 0u : (wi == w32 - 1 && tail ? (0xFFFFFFFFu >> (32 - tail)) : 0xFFFFFFFFu);
    const uint32_t *fsrc = src + f * (size_t)h * w32 + min(max(wi, 0), w32 - 1);
    uint32_t *fdst = dst + f * (size_t)h * w32;

    const int y0 = band * kStreamBand;
    if (y0 >= h)
        return;
    const int y_first = y0 - A0 - A1;                       // first input row marched
    const int y_last = min(h, y0 + kStreamBand) - 1 + A0 + A1;

    // rows y .. y+7: unconditional loads from clamped rows, masked afterwards
    auto load_group = [&](int y, uint32_t (&g)[kStreamGroup]) {
#pragma unroll
        for (int k = 0; k < kStreamGroup; k++)
            g[k] = fsrc[(size_t)min(max(y + k, 0), h - 1) * w32];
#pragma unroll
        for (int k = 0; k < kStreamGroup; k++)
            g[k] = (y + k >= 0 && y + k < h) ? ((g[k] ^ flip_in) & valid) : 0u;
    };

    uint32_t win0[K0], win1[K1 ? K1 : 1];
#pragma unroll
    for (int k = 0; k < K0; k++)
        win0[k] = 0u;
#pragma unroll
    for (int k = 0; k < (K1 ? K1 : 1); k++)
        win1[k] = 0u;

    uint32_t cur[kStreamGroup], nxt[kStreamGroup];
    load_group(y_first, nxt);
    for (int yg = y_first; yg <= y_last; yg += kStreamGroup) {
#pragma unroll
        for (int k = 0; k < kStreamGroup; k++)
            cur[k] = nxt[k];
        if (yg + kStreamGroup <= y_last)
            load_group(yg + kStreamGroup, nxt);    // the next eight rows fly behind this group's ALU work
#pragma unroll
        for (int k = 0; k < kStreamGroup; k++) {
            const int y = yg + k;
            if (y > y_last)
                break;
            const uint32_t c = cur[k], l = lane_left(c), r = lane_right(c);
            // ---- op 0: horizontal pass of input row y, vertical window -> row y - A0
#pragma unroll
            for (int j = 0; j + 1 < K0; j++)
                win0[j] = win0[j + 1];
            win0[K0 - 1] = hdilate<K0>(l, c, r);
            uint32_t v0 = 0u;
#pragma unroll
            for (int j = 0; j < K0; j++)
                v0 |= win0[j];
            const int ya = y - A0;                    // row of v0
            uint32_t out;
            int yo;
            if constexpr (K1 == 0) {
                out = (v0 ^ flip_out) & valid;
                yo = ya;
            } else {
                // input of op 1: rows outside the frame contribute nothing
                const uint32_t t = (ya >= 0 && ya < h) ? ((v0 ^ flip_mid) & valid) : 0u;
                const uint32_t tl = lane_left(t), tr = lane_right(t);
#pragma unroll
                for (int j = 0; j + 1 < K1; j++)
                    win1[j] = win1[j + 1];
                win1[K1 ? K1 - 1 : 0] = hdilate<(K1 ? K1 : 1)>(tl, t, tr);
                uint32_t v1 = 0u;
#pragma unroll
                for (int j = 0; j < K1; j++)
                    v1 |= win1[j];
                out = (v1 ^ flip_out) & valid;
                yo = ya - A1;
            }
            if (col_store && yo >= y0 && yo < y0 + kStreamBand && yo < h)
                fdst[(size_t)yo * w32 + wi] = out;
        }
    }
}

// ------------------------------------------------------------------ morphology on u8 images
template <bool DILATE>
__global__ void __launch_bounds__(kBlock)
morph_u8_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int h, int w,
                size_t total, RowSpans se)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    const int x = (int)(e % w);
    const size_t rowi = e / w;
    const int y = (int)(rowi % h);
    const uint8_t *frame = src + (rowi - y) * (size_t)w;
    int best = DILATE ? 0 : 255;
    for (int i = 0; i < se.ksize; i++) {
        const int yy = y + i - se.anchor;
        if (yy < 0 || yy >= h)
            continue;
        int x0 = x + se.lo[i] - se.anchor, x1 = x + se.hi[i] - 1 - se.anchor;
        x0 = x0 < 0 ? 0 : x0;
        x1 = x1 >= w ? w - 1 : x1;
        const uint8_t *row = frame + (size_t)yy * w;
        for (int xx = x0; xx <= x1; xx++) {
            int v = row[xx];
            best = DILATE ? (v > best ? v : best) : (v < best ? v : best);
        }
    }
    dst[e] = (uint8_t)best;
}

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

int launch_pack_bits(const uint8_t *src, uint32_t *bits, int n, int h, int w, int thresh,
                     hipStream_t st)
{
    const int w32 = words_per_row(w);
    size_t total = (size_t)n * h * w32;
    if (total == 0)
        return VA_OK;
    int vec = (w % 32 == 0) && aligned(src, 16);
    pack_bits_kernel<<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(src, bits, w, w32, total,
                                                                       thresh, vec);
    VA_LAUNCH_CHECK("pack_bits_kernel");
    return VA_OK;
}

int launch_unpack_bits(const uint32_t *bits, uint8_t *dst, int n, int h, int w, int maxval,
                       hipStream_t st)
{
    const int w32 = words_per_row(w);
    size_t total = (size_t)n * h * w32;
    if (total == 0)
        return VA_OK;
    int vec = (w % 32 == 0) && aligned(dst, 16);
    if (vec)       // (rows of whole words: the output is one contiguous run of 32 bytes per word)
        unpack_bits_wide_kernel<<<cdiv((long long)(2 * total), kBlock), kBlock, 0, st>>>(bits, dst, 2 * total,
                                                                                         maxval);
    else
        unpack_bits_kernel<<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(bits, dst, w, w32, total,
                                                                             maxval, vec);
    VA_LAUNCH_CHECK("unpack_bits_kernel");
    return VA_OK;
}

int launch_morph_bits(const uint32_t *src, uint32_t *dst, int n, int h, int w, int op,
                      const RowSpans &se, hipStream_t st)
{
    const int w32 = words_per_row(w);
    size_t total = (size_t)n * h * w32;
    if (total == 0)
        return VA_OK;
    int grid = cdiv((long long)total, kBlock);
    if (op == VA_MORPH_DILATE)
        morph_bits_kernel<true><<<grid, kBlock, 0, st>>>(src, dst, h, w, w32, total, se);
    else
        morph_bits_kernel<false><<<grid, kBlock, 0, st>>>(src, dst, h, w, w32, total, se);
    VA_LAUNCH_CHECK("morph_bits_kernel");
    return VA_OK;
}

bool morph_fused_supported(int w, const RowSpans *se, int count)
{
    if (count < 1 || count > VA_MAX_MORPH_OPS)
        return false;
    int reach = 0;
    for (int k = 0; k < count; k++) {
        int a = se[k].anchor, b = se[k].ksize - 1 - se[k].anchor;
        reach += a > b ? a : b;
    }
    size_t lds = 3 * (size_t)(kBandRows + 2 * reach) * words_per_row(w) * sizeof(uint32_t);
    return reach <= 16 && lds <= 64 * 1024;
}

static bool is_rect_odd(const RowSpans &se, int kmax)
{
    if (!(se.ksize & 1) || se.ksize > kmax || se.ksize < 3)
        return false;
    for (int i = 0; i < se.ksize; i++)
        if (se.lo[i] != 0 || se.hi[i] != se.ksize)
            return false;
    return true;
}

template <int K0, int K1>
static int launch_stream(const uint32_t *src, uint32_t *dst, int n, int h, int w, const int *ops,
                         hipStream_t st)
{
    const int w32 = words_per_row(w);
    const int cols = K1 ? 60 : 62;
    const int nchunks = cdiv(w32, cols), nbands = cdiv(h, kStreamBand);
    const size_t items = (size_t)n * nbands * nchunks;
    // erode = complement . dilate . complement; adjacent complements cancel
    const uint32_t e0 = ops[0] == VA_MORPH_ERODE ? 0xFFFFFFFFu : 0u;
    const uint32_t e1 = K1 ? (ops[1] == VA_MORPH_ERODE ? 0xFFFFFFFFu : 0u) : 0u;
    const uint32_t flip_in = e0, flip_mid = e0 ^ e1, flip_out = K1 ? e1 : e0;
    // the grid is rounded up to whole blocks: surplus waves see y0 >= h via the band index
    const int waves_per_block = kBlock / 64;
    const size_t blocks = (items + waves_per_block - 1) / waves_per_block;
    morph_stream_kernel<K0, K1><<<(unsigned)blocks, kBlock, 0, st>>>(
        src, dst, n, h, w, w32, nbands, nchunks, flip_in, flip_mid, flip_out);
    VA_LAUNCH_CHECK("morph_stream_kernel");
    return VA_OK;
}

int launch_morph_fused(const uint32_t *src, uint32_t *dst, int n, int h, int w, const int *ops,
                       const RowSpans *se, int count, hipStream_t st)
{
    VA_REQUIRE(morph_fused_supported(w, se, count), "fused morphology: unsupported sequence");
    if (n == 0 || h == 0 || w == 0)
        return VA_OK;
    if (count <= 2 && is_rect_odd(se[0], 7) && (count == 1 || is_rect_odd(se[1], 7))) {
        const int k0 = se[0].ksize, k1 = count == 2 ? se[1].ksize : 0;
#define VA_STREAM_CASE(A, B) \
    if (k0 == A && k1 == B)  \
        return launch_stream<A, B>(src, dst, n, h, w, ops, st);
        VA_STREAM_CASE(3, 0) VA_STREAM_CASE(5, 0) VA_STREAM_CASE(7, 0)
        VA_STREAM_CASE(3, 3) VA_STREAM_CASE(3, 5) VA_STREAM_CASE(3, 7)
        VA_STREAM_CASE(5, 3) VA_STREAM_CASE(5, 5) VA_STREAM_CASE(5, 7)
        VA_STREAM_CASE(7, 3) VA_STREAM_CASE(7, 5) VA_STREAM_CASE(7, 7)
#undef VA_STREAM_CASE
    }
    MorphSeq seq;
    memset(&seq, 0, sizeof(seq));
    seq.count = count;
    for (int k = 0; k < count; k++) {
        seq.op[k] = ops[k];
        seq.se[k] = se[k];
        int a = se[k].anchor, b = se[k].ksize - 1 - se[k].anchor;
        seq.reach_total += a > b ? a : b;
    }
    const int w32 = words_per_row(w);
    const int nbands = cdiv(h, kBandRows);
    const size_t lds = 3 * (size_t)(kBandRows + 2 * seq.reach_total) * w32 * sizeof(uint32_t);
    morph_fused_kernel<<<nbands * n, kBlock, lds, st>>>(src, dst, h, w, w32, nbands,
                                                       seq);
    VA_LAUNCH_CHECK("morph_fused_kernel");
    return VA_OK;
}

// rectangles are separable: a k-wide row pass into `scratch`, then a k-tall column pass
// (2k loads per pixel instead of k^2; pixels outside the frame never win in either pass)
template <bool DILATE, bool VERT>
__global__ void __launch_bounds__(kBlock)
morph_u8_line_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int h, int w, size_t total,
                     int ksize, int anchor)
{
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= total)
        return;
    const int x = (int)(e % w);
    const size_t rowi = e / w;
    const int y = (int)(rowi % h);
    const uint8_t *frame = src + (rowi - y) * (size_t)w;
    int best = DILATE ? 0 : 255;
    const int c = VERT ? y : x, len = VERT ? h : w;
    int a = c - anchor, b = c - anchor + ksize - 1;
    a = a < 0 ? 0 : a;
    b = b >= len ? len - 1 : b;
    for (int t = a; t <= b; t++) {
        const int v = VERT ? frame[(size_t)t * w + x] : frame[(size_t)y * w + t];
        best = DILATE ? (v > best ? v : best) : (v < best ? v : best);
    }
    dst[e] = (uint8_t)best;
}

namespace {

// Grey-level erosion / dilation, four samples per thread (rows of whole, aligned dwords).  The running extreme of four
// neighbouring samples is kept as two pairs of 16-bit lanes (even bytes, odd bytes: v_pk_max_u16 / v_pk_min_u16 on
// both), and an element of the structuring element contributes the frame dword that starts `s` bytes from the
// thread's own -- one v_alignbyte of two aligned dwords -- so a thread issues (span + 3) / 4 + 1 loads per row of the
// element instead of four byte loads per element.  Pixels outside the frame never win (OpenCV's default border).
// grid = (dwords of a row, rows, frames): no division per sample.  The one-sample kernels above remain for other
// widths; they are bound by the launch rate of waves.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <bool DILATE>
struct Extreme4 {
    us2 even, odd;
    __device__ __forceinline__ Extreme4()
    {
        const unsigned short v = DILATE ? 0 : 255;
        even = us2{v, v};
        odd = us2{v, v};
    }
    __device__ __forceinline__ void take(uint32_t d)
    {
        const us2 e = __builtin_bit_cast(us2, d & 0x00FF00FFu), o = __builtin_bit_cast(us2, (d >> 8) & 0x00FF00FFu);
        even = DILATE ? __builtin_elementwise_max(even, e) : __builtin_elementwise_min(even, e);
        odd = DILATE ? __builtin_elementwise_max(odd, o) : __builtin_elementwise_min(odd, o);
    }
    __device__ __forceinline__ uint32_t word() const
    {
        return __builtin_bit_cast(uint32_t, even) | (__builtin_bit_cast(uint32_t, odd) << 8);
    }
};

// every row i of the element: samples x + lo[i] - anchor ... x + hi[i] - 1 - anchor of row y + i - anchor
template <bool DILATE>
__global__ void __launch_bounds__(kBlock)
morph_u8_rows_x4_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int h, int w, RowSpans se)
{
    const int q = blockIdx.x * kBlock + threadIdx.x, wq = w >> 2;
    if (q >= wq)
        return;
    const int y = blockIdx.y;
    const size_t f = blockIdx.z;
    const uint32_t *frame = reinterpret_cast<const uint32_t *>(src + f * (size_t)h * w);
    const uint32_t neutral = DILATE ? 0u : 0xFFFFFFFFu;
    Extreme4<DILATE> best;
    for (int i = 0; i < se.ksize; i++) {
        const int yy = y + i - se.anchor;
        const int lo = se.lo[i], hi = se.hi[i];
        if (yy < 0 || yy >= h || hi <= lo)
            continue;                                    // (uniform)
        const uint32_t *row = frame + (size_t)yy * wq;
        // byte offsets lo - anchor ... hi - 1 - anchor from the thread's dword; aligned dword d holds bytes 4d ... 4d+3
        int off = lo - se.anchor;                        // uniform
        int d = q + (off >> 2);                          // floor division
        uint32_t cur = (d >= 0 && d < wq) ? row[d] : neutral;
        uint32_t nxt = (d + 1 >= 0 && d + 1 < wq) ? row[d + 1] : neutral;
        for (int s = lo; s < hi; s++, off++) {
            const int sh = off & 3;
            if (sh == 0 && s != lo) {
                d++;
                cur = nxt;
                nxt = (d + 1 >= 0 && d + 1 < wq) ? row[d + 1] : neutral;
            }
            best.take(__builtin_amdgcn_alignbyte(nxt, cur, sh));
        }
    }
    reinterpret_cast<uint32_t *>(dst + (f * h + y) * (size_t)w)[q] = best.word();
}

// a column of `ksize` rows (the second pass of a rectangle)
template <bool DILATE>
__global__ void __launch_bounds__(kBlock)
morph_u8_vert_x4_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int h, int w, int ksize, int anchor)
{
    const int q = blockIdx.x * kBlock + threadIdx.x, wq = w >> 2;
    if (q >= wq)
        return;
    const int y = blockIdx.y;
    const size_t f = blockIdx.z;
    const uint32_t *frame = reinterpret_cast<const uint32_t *>(src + f * (size_t)h * w);
    int a = y - anchor, b = y - anchor + ksize - 1;
    a = a < 0 ? 0 : a;
    b = b >= h ? h - 1 : b;
    Extreme4<DILATE> best;
    for (int t = a; t <= b; t++)
        best.take(frame[(size_t)t * wq + q]);
    reinterpret_cast<uint32_t *>(dst + (f * h + y) * (size_t)w)[q] = best.word();
}

}  // namespace

int launch_morph_u8(const uint8_t *src, uint8_t *dst, int n, int h, int w, int op,
                    const RowSpans &se, hipStream_t st, uint8_t *scratch)
{
    if (w % 4 == 0 && h <= 65535 && n <= 65535 && n > 0 && src != dst && reinterpret_cast<uintptr_t>(src) % 4 == 0 &&
        reinterpret_cast<uintptr_t>(dst) % 4 == 0 && (!scratch || reinterpret_cast<uintptr_t>(scratch) % 4 == 0)) {
        const dim3 grid((unsigned)cdiv(w / 4, kBlock), (unsigned)h, (unsigned)n);
        bool full = scratch != nullptr && se.ksize >= 3;     // a rectangle: one row of the element, then a column
        for (int i = 0; i < se.ksize && full; i++)
            full = se.lo[i] == 0 && se.hi[i] == se.ksize;
        if (full) {
            RowSpans line = {};
            line.ksize = 1;
            line.anchor = 0;
            line.lo[0] = (int8_t)(-se.anchor);
            line.hi[0] = (int8_t)(se.ksize - se.anchor);
            if (op == VA_MORPH_DILATE) {
                morph_u8_rows_x4_kernel<true><<<grid, kBlock, 0, st>>>(src, scratch, h, w, line);
                morph_u8_vert_x4_kernel<true><<<grid, kBlock, 0, st>>>(scratch, dst, h, w, se.ksize, se.anchor);
            } else {
                morph_u8_rows_x4_kernel<false><<<grid, kBlock, 0, st>>>(src, scratch, h, w, line);
                morph_u8_vert_x4_kernel<false><<<grid, kBlock, 0, st>>>(scratch, dst, h, w, se.ksize, se.anchor);
            }
        } else if (op == VA_MORPH_DILATE) {
            morph_u8_rows_x4_kernel<true><<<grid, kBlock, 0, st>>>(src, dst, h, w, se);
        } else {
            morph_u8_rows_x4_kernel<false><<<grid, kBlock, 0, st>>>(src, dst, h, w, se);
        }
        VA_LAUNCH_CHECK("morph_u8_x4 kernels");
        return VA_OK;
    }
    bool rect = scratch != nullptr && se.ksize >= 3;
    for (int i = 0; i < se.ksize && rect; i++)
        rect = se.lo[i] == 0 && se.hi[i] == se.ksize;
    if (rect) {
        size_t total = (size_t)n * h * w;
        if (total == 0)
            return VA_OK;
        const int grid = cdiv((long long)total, kBlock);
        if (op == VA_MORPH_DILATE) {
            morph_u8_line_kernel<true, false><<<grid, kBlock, 0, st>>>(src, scratch, h, w, total, se.ksize, se.anchor);
            morph_u8_line_kernel<true, true><<<grid, kBlock, 0, st>>>(scratch, dst, h, w, total, se.ksize, se.anchor);
        } else {
            morph_u8_line_kernel<false, false><<<grid, kBlock, 0, st>>>(src, scratch, h, w, total, se.ksize, se.anchor);
            morph_u8_line_kernel<false, true><<<grid, kBlock, 0, st>>>(scratch, dst, h, w, total, se.ksize, se.anchor);
        }
        VA_LAUNCH_CHECK("morph_u8_line_kernel");
        return VA_OK;
    }

    size_t total = (size_t)n * h * w;
    if (total == 0)
        return VA_OK;
    int grid = cdiv((long long)total, kBlock);
    if (op == VA_MORPH_DILATE)
        morph_u8_kernel<true><<<grid, kBlock, 0, st>>>(src, dst, h, w, total, se);
    else
        morph_u8_kernel<false><<<grid, kBlock, 0, st>>>(src, dst, h, w, total, se);
    VA_LAUNCH_CHECK("morph_u8_kernel");
    return VA_OK;
}

}  // namespace va
