// va_gauss_f32.hip -- float32 Gaussian blur, fast two-pass path (1 or 3 interleaved channels)
//
// replaces cv2.GaussianBlur(float image, (0, 0), sigma), video/analysis/active_contour.py:108
// (BASELINE.json configs[4]: 1080p x 3 channels, sigma = 9 -> 73 taps).
//
// Bit-exact with the oracle's definition: row pass = fmaf chain over the taps in order, column
// pass = centre tap then symmetric pairs (T[y+k] + T[y-k]) folded with fmaf.  Unlike the u8
// kernel there is no packed trick: ~146 FMA-class operations per sample, so both passes are
// organised to be VALU-bound, not load-bound:
//   row pass : one block per image row; the row (+ reflected halo) is staged once in LDS; a
//              thread owns P consecutive samples of the interleaved row, keeps a rolling window
//              of P + C inputs in registers and per tap reads only C new floats (register
//              indices are compile-time: the tap loop is unrolled over one window period)
//   col pass : a thread owns one sample column for 8 consecutive rows and streams outwards from
//              the centre: two 8-row register rings, two new (coalesced, prefetched) rows per tap
// The intermediate image goes through HBM once (write + read = 8 B per sample); a fully fused
// kernel would need a (2r+8)-row float ring in LDS (61 KB at r = 36 for a 64-pixel strip).
#include "va_common.h"

namespace va {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

template <int C, int P>
__global__ void __launch_bounds__(kBlock, 4)
gauss_row_f32_kernel(const float *__restrict__ src, float *__restrict__ tmp, int w, TapsF32 taps,
                     int vec4)
{
    constexpr int WIN = P + C;          // rolling window: P samples in use + C arriving
    constexpr int PERIOD = WIN / C;     // taps after which the window slots repeat
    static_assert(WIN % C == 0, "P must be a multiple of C");
    static_assert(P % 2 == 1, "an odd per-thread stride keeps the LDS reads of a wave conflict-free");
    extern __shared__ float s_row[];    // w*C + 2*r*C samples: the row with its reflected halo
    const int n = taps.ksize, r = n >> 1;
    const int rw = w * C, halo = r * C;
    const float *row = src + (size_t)blockIdx.x * rw;
    // interior: straight (vectorised when aligned) copy; only the 2*halo edge samples reflect
    if (vec4) {
        const float4 *row4 = reinterpret_cast<const float4 *>(row);
        for (int i = threadIdx.x; i < rw / 4; i += kBlock) {
            const float4 v = row4[i];
            float *d = s_row + halo + 4 * i;
            d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < rw; i += kBlock)
            s_row[halo + i] = row[i];
    }
    for (int i = threadIdx.x; i < 2 * halo; i += kBlock) {
        const int q = i < halo ? i - halo : rw + (i - halo);   // sample index relative to the row start
        const int px = q >= 0 ? q / C : -((-q + C - 1) / C);
        const int ch = q - px * C;
        s_row[q + halo] = row[reflect101(px, w) * C + ch];
    }
    __syncthreads();

    for (int i0 = threadIdx.x * P; i0 < rw; i0 += kBlock * P) {
        float win[WIN], acc[P];
#pragma unroll
        for (int j = 0; j < P; j++) {
            acc[j] = 0.0f;
            win[j] = s_row[i0 + j];
        }
        // full window periods: branch-free, the C samples of the next tap are fetched from LDS
        // before this tap's P fmas and installed after them (software pipeline, depth one tap)
        int k0 = 0;
        for (; k0 + PERIOD <= n; k0 += PERIOD) {
            float wk[PERIOD];               // this period's taps, fetched together (scalar loads)
#pragma unroll
            for (int t = 0; t < PERIOD; t++)
                wk[t] = taps.t[k0 + t];
#pragma unroll
            for (int t = 0; t < PERIOD; t++) {
                float nx[C];
#pragma unroll
                for (int c = 0; c < C; c++)     // (one tap past the end reads the LDS padding)
                    nx[c] = s_row[i0 + (k0 + t) * C + P + c];
#pragma unroll
                for (int j = 0; j < P; j++)
                    acc[j] = fmaf(win[(t * C + j) % WIN], wk[t], acc[j]);
#pragma unroll
                for (int c = 0; c < C; c++)
                    win[(t * C + P + c) % WIN] = nx[c];
                // keep the pipeline depth at one tap: without this the scheduler hoists the LDS
                // reads of many taps and triples the register footprint
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // remaining taps (fewer than one period)
#pragma unroll
        for (int t = 0; t < PERIOD; t++) {
            const int k = k0 + t;
            if (k < n) {                                // uniform
                const float wk = taps.t[k];
                float nx[C];
#pragma unroll
                for (int c = 0; c < C; c++)
                    nx[c] = s_row[i0 + k * C + P + c];
#pragma unroll
                for (int j = 0; j < P; j++)
                    acc[j] = fmaf(win[(t * C + j) % WIN], wk, acc[j]);
#pragma unroll
                for (int c = 0; c < C; c++)
                    win[(t * C + P + c) % WIN] = nx[c];
            }
        }
        float *out = tmp + (size_t)blockIdx.x * rw + i0;
#pragma unroll
        for (int j = 0; j < P; j++)
            if (i0 + j < rw)
                out[j] = acc[j];
    }
}

// Column pass.  For output rows y0..y0+7 of one sample column, tap k needs the two 8-row
// groups T[y0+k .. y0+7+k] and T[y0-k .. y0+7-k]; from one tap to the next each group slides by
// one row.  So the thread keeps just two 8-entry rings (slot = row mod 8, compile-time after
// unrolling 8 taps), fetches two new rows per tap -- prefetched kPF taps ahead -- and the whole
// pass needs ~45 VGPRs instead of a (2r+8)-row register window.
template <int R>
__global__ void __launch_bounds__(kBlock, 6)
gauss_col_f32_kernel(const float *__restrict__ tmp, float *__restrict__ dst, int h, int rw,
                     int nx, int nbands, TapsF32 taps)
{
    static_assert(R == 8, "ring arithmetic below assumes 8 rows per thread");
    constexpr int kPF = 8;                 // taps of prefetch distance (covers an L2 round trip)
    // XCD-aware tile order (blocks are dealt round-robin over the 8 XCDs): XCD x handles the
    // column strips x, x+8, ... and walks the row bands of a frame in order, so the rows that
    // consecutive bands share are re-read from that XCD's own L2 instead of the fabric
    const int nxg = (nx + 7) >> 3;
    const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
    const int bx = xcd + 8 * (m % nxg);
    const int band = (m / nxg) % nbands, fz = m / (nxg * nbands);
    const int i = bx * kBlock + threadIdx.x;
    if (bx >= nx || i >= rw)
        return;
    const int y0 = band * R;
    const int r = taps.ksize >> 1;
    const float *frame = tmp + (size_t)fz * h * rw + i;
    float *out = dst + (size_t)fz * h * rw + i;
    auto rowv = [&](int y) { return frame[(size_t)reflect101(y, h) * rw]; };

    float up[R], dn[R], acc[R];            // up[s]: row y with (y - y0) % 8 == s, same for dn
#pragma unroll
    for (int j = 0; j < R; j++) {
        up[j] = dn[j] = rowv(y0 + j);
        acc[j] = fmaf(up[j], taps.t[r], 0.0f);
    }
    float pu[kPF], pd[kPF];                // rows arriving for taps k .. k+kPF-1
#pragma unroll
    for (int q = 0; q < kPF; q++) {
        pu[q] = rowv(y0 + R - 1 + 1 + q);
        pd[q] = rowv(y0 - 1 - q);
    }
    for (int k0 = 1; k0 <= r; k0 += 8) {   // k0 = 1 (mod 8): every ring / queue slot below is static
        float wk8[8];                          // this period's taps, one batch of scalar loads
#pragma unroll
        for (int t = 0; t < 8; t++)
            wk8[t] = taps.t[min(r + k0 + t, kMaxTaps)];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int k = k0 + t;
            if (k <= r) {                               // uniform
                const float wk = wk8[t];
                up[t] = pu[t % kPF];                    // row y0+7+k replaces row y0+k-1
                dn[7 - t] = pd[t % kPF];                // row y0-k   replaces row y0+8-k
                if (k + kPF <= r) {                     // uniform: fetch the rows of tap k+kPF
                    pu[t % kPF] = rowv(y0 + R - 1 + k + kPF);
                    pd[t % kPF] = rowv(y0 - k - kPF);
                }
#pragma unroll
                for (int j = 0; j < R; j++)
                    acc[j] = fmaf(up[(j + 1 + t) % 8] + dn[(j + 7 - t) % 8], wk, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < R; j++)
        if (y0 + j < h)
            out[(size_t)(y0 + j) * rw] = acc[j];
}


// ================================================================== packed (v_pk_*_f32) kernels
// Same arithmetic, organised around three measured facts: packed fp32 halves the
// VALU instructions when both halves of a register pair are independent outputs; four-byte stores
// scattered over a row are far more expensive than their bytes; a pass that re-reads its input
// nine times through L2 is bound by L2, not by HBM.
//   row pass : a workgroup filters TWO image rows at once; LDS holds (row y, row y+1) sample
//              pairs, so a thread's window and accumulators are float2 and every fma is a
//              v_pk_fma_f32 whose halves are the two rows (always register-aligned).  Results go
//              back through LDS and leave as 16-byte row pieces.
//   col pass : a workgroup stages a (32 + 2r)-row x 128-sample tile in LDS once (3.25 loads per
//              output instead of 9.25 through L2); a thread owns TWO adjacent sample columns
//              (float2) for 8 rows, with the same two 8-row rings as above fed from LDS.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

constexpr int kRowThreads = 384;

template <int C, int P>
__global__ void __launch_bounds__(kRowThreads)
gauss_row2_f32_kernel(const float *__restrict__ src, float *__restrict__ tmp, size_t total_rows, int w,
                      TapsF32 taps, int vec4)
{
    constexpr int WIN = P + C;
    constexpr int PERIOD = WIN / C;
    static_assert(WIN % C == 0, "P must be a multiple of C");
    extern __shared__ f2 s2[];          // sample q of both rows at s2[q + halo]
    const int n = taps.ksize, r = n >> 1;
    const int rw = w * C, halo = r * C;
    const size_t ya = (size_t)blockIdx.x * 2, yb = min(ya + 1, total_rows - 1);
    const float *rowa = src + ya * rw, *rowb = src + yb * rw;
    if (vec4) {
        for (int i = threadIdx.x; i < rw / 4; i += kRowThreads) {
            const f4 a = *reinterpret_cast<const f4 *>(rowa + 4 * i);
            const f4 b = *reinterpret_cast<const f4 *>(rowb + 4 * i);
            f2 *d = s2 + halo + 4 * i;
            d[0] = f2{a.x, b.x}, d[1] = f2{a.y, b.y}, d[2] = f2{a.z, b.z}, d[3] = f2{a.w, b.w};
        }
    } else {
        for (int i = threadIdx.x; i < rw; i += kRowThreads)
            s2[halo + i] = f2{rowa[i], rowb[i]};
    }
    for (int i = threadIdx.x; i < 2 * halo; i += kRowThreads) {
        const int q = i < halo ? i - halo : rw + (i - halo);   // sample index relative to the row start
        const int px = q >= 0 ? q / C : -((-q + C - 1) / C);
        const int ch = q - px * C;
        const int sidx = reflect101(px, w) * C + ch;
        s2[q + halo] = f2{rowa[sidx], rowb[sidx]};
    }
    for (int i = threadIdx.x; i < 32; i += kRowThreads)          // padding read one tap past the end
        s2[rw + 2 * halo + i] = f2{0.0f, 0.0f};
    __syncthreads();

    const int nseg = (rw + P - 1) / P;
    const int rounds = (nseg + kRowThreads - 1) / kRowThreads;   // uniform: barriers inside
    for (int rd = 0; rd < rounds; rd++) {
        const int i0 = (rd * kRowThreads + (int)threadIdx.x) * P;
        const bool act = i0 < rw;
        f2 win[WIN], acc[P];
        if (act) {
#pragma unroll
            for (int j = 0; j < P; j++) {
                acc[j] = f2{0.0f, 0.0f};
                win[j] = s2[i0 + j];
            }
            int k0 = 0;
            for (; k0 + PERIOD <= n; k0 += PERIOD) {
                float wk[PERIOD];
#pragma unroll
                for (int t = 0; t < PERIOD; t++)
                    wk[t] = taps.t[k0 + t];
#pragma unroll
                for (int t = 0; t < PERIOD; t++) {
                    f2 nx[C];
#pragma unroll
                    for (int c = 0; c < C; c++)
                        nx[c] = s2[i0 + (k0 + t) * C + P + c];
                    const f2 w2 = f2{wk[t], wk[t]};
#pragma unroll
                    for (int j = 0; j < P; j++)
                        acc[j] = pk_fma(win[(t * C + j) % WIN], w2, acc[j]);
#pragma unroll
                    for (int c = 0; c < C; c++)
                        win[(t * C + P + c) % WIN] = nx[c];
                }
            }
#pragma unroll
            for (int t = 0; t < PERIOD; t++) {
                const int k = k0 + t;
                if (k < n) {                                // uniform
                    const f2 w2 = f2{taps.t[k], taps.t[k]};
                    f2 nx[C];
#pragma unroll
                    for (int c = 0; c < C; c++)
                        nx[c] = s2[i0 + k * C + P + c];
#pragma unroll
                    for (int j = 0; j < P; j++)
                        acc[j] = pk_fma(win[(t * C + j) % WIN], w2, acc[j]);
#pragma unroll
                    for (int c = 0; c < C; c++)
                        win[(t * C + P + c) % WIN] = nx[c];
                }
            }
        }
        // results through LDS (the staged input of this round's segments is dead only after
        // every thread has finished its window reads)
        __syncthreads();
        if (act) {
#pragma unroll
            for (int j = 0; j < P; j++)
                if (i0 + j < rw)
                    s2[halo + i0 + j] = acc[j];
        }
        __syncthreads();
        const int lo = rd * kRowThreads * P, hi = min(rw, lo + kRowThreads * P);
        float *outa = tmp + ya * rw, *outb = tmp + yb * rw;
        if (vec4 && lo % 4 == 0) {
            for (int i = lo / 4 + threadIdx.x; 4 * i < hi; i += kRowThreads) {
                const f2 *sp = s2 + halo + 4 * i;
                if (4 * i + 3 < hi) {
                    *reinterpret_cast<f4 *>(outa + 4 * i) = f4{sp[0].x, sp[1].x, sp[2].x, sp[3].x};
                    if (yb != ya)
                        *reinterpret_cast<f4 *>(outb + 4 * i) = f4{sp[0].y, sp[1].y, sp[2].y, sp[3].y};
                } else {
                    for (int q = 4 * i; q < hi; q++) {
                        outa[q] = s2[halo + q].x;
                        if (yb != ya)
                            outb[q] = s2[halo + q].y;
                    }
                }
            }
        } else {
            for (int i = lo + threadIdx.x; i < hi; i += kRowThreads) {
                outa[i] = s2[halo + i].x;
                if (yb != ya)
                    outb[i] = s2[halo + i].y;
            }
        }
        // (a further round would need the staged input again: rows wider than one round are not
        // routed here, see gauss_f32_packed_supported)
    }
}

constexpr int kColTileRows = 64, kColTileCols = 64;      // outputs per workgroup: 8 row groups x 32 column pairs
constexpr int kColStride = kColTileCols + 2;             // floats per staged row (even: float2 reads stay aligned)

__global__ void __launch_bounds__(kBlock, 4)
gauss_col2_f32_kernel(const float *__restrict__ tmp, float *__restrict__ dst, int h, int rw, int ncolt,
                      int nrowt, TapsF32 taps, int vec4)
{
    extern __shared__ float tile[];     // (64 + 2r) rows x 64 samples; row j <-> image row y0 - r + j
    const int r = taps.ksize >> 1;
    int b = blockIdx.x;
    const int ct = b % ncolt;
    b /= ncolt;
    const int rt = b % nrowt;
    const size_t fz = b / nrowt;
    const int x0 = ct * kColTileCols, y0 = rt * kColTileRows;
    const float *frame = tmp + fz * (size_t)h * rw;
    const int nrows = kColTileRows + 2 * r;
    // stage: 16 float4 per row; loads in batches of four so that several are in flight
    const int items = nrows * (kColTileCols / 4);
    for (int base = 0; base < items; base += 4 * kBlock) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = min(base + u * kBlock + (int)threadIdx.x, items - 1);
            const int row = idx >> 4, c4 = (idx & 15) * 4;
            const float *g = frame + (size_t)reflect101(y0 - r + row, h) * rw + x0 + c4;
            if (vec4 && x0 + c4 + 3 < rw) {
                v[u] = *reinterpret_cast<const f4 *>(g);
            } else {
                v[u].x = x0 + c4 + 0 < rw ? g[0] : 0.0f;
                v[u].y = x0 + c4 + 1 < rw ? g[1] : 0.0f;
                v[u].z = x0 + c4 + 2 < rw ? g[2] : 0.0f;
                v[u].w = x0 + c4 + 3 < rw ? g[3] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = base + u * kBlock + (int)threadIdx.x;
            if (idx < items) {
                float *d = tile + (idx >> 4) * kColStride + (idx & 15) * 4;
                *reinterpret_cast<f2 *>(d) = f2{v[u].x, v[u].y};
                *reinterpret_cast<f2 *>(d + 2) = f2{v[u].z, v[u].w};
            }
        }
    }
    __syncthreads();

    const int cp = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = x0 + 2 * cp;
    if (col >= rw)
        return;
    constexpr int RS = kColStride / 2;                                // row stride in f2
    const f2 *tcol = reinterpret_cast<const f2 *>(tile) + cp;
    const int tr0 = 8 * rg + r;                                      // tile row of output j = 0
    f2 up[8], dn[8], acc[8];
    {
        const f2 wc = f2{taps.t[r], taps.t[r]};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            up[j] = dn[j] = tcol[(tr0 + j) * RS];
            acc[j] = pk_fma(up[j], wc, f2{0.0f, 0.0f});
        }
    }
    for (int k0 = 1; k0 <= r; k0 += 8) {   // k0 = 1 (mod 8): every ring slot below is static
        float wk8[8];
        f2 nu[8], nd[8];                   // this period's sixteen rows, read from LDS in one batch
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int k = min(k0 + t, r);
            wk8[t] = taps.t[r + k];
            nu[t] = tcol[(tr0 + 7 + k) * RS];   // row y+7+k replaces row y+k-1
            nd[t] = tcol[(tr0 - k) * RS];       // row y-k   replaces row y+8-k
        }
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (k0 + t <= r) {                          // uniform
                up[t] = nu[t];
                dn[7 - t] = nd[t];
                const f2 w2 = f2{wk8[t], wk8[t]};
#pragma unroll
                for (int j = 0; j < 8; j++)
                    acc[j] = pk_fma(up[(j + 1 + t) % 8] + dn[(j + 7 - t) % 8], w2, acc[j]);
            }
        }
    }
    float *out = dst + fz * (size_t)h * rw + col;
    const int yb = y0 + 8 * rg;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (yb + j < h) {
            float *o = out + (size_t)(yb + j) * rw;
            if (col + 1 < rw && (rw % 2 == 0))
                *reinterpret_cast<f2 *>(o) = acc[j];
            else {
                o[0] = acc[j].x;
                if (col + 1 < rw)
                    o[1] = acc[j].y;
            }
        }
    }
}

// rows that fit one round of the packed row kernel and a column tile that fits 64 KB of LDS
bool gauss_f32_packed_supported(int w, int c, const TapsF32 &taps)
{
    const int r = taps.ksize / 2, rw = w * c, P = 15;
    if ((rw + P - 1) / P > kRowThreads)
        return false;
    if (((size_t)rw + 2 * (size_t)r * c + 32) * sizeof(f2) > 64 * 1024)
        return false;
    return (size_t)(kColTileRows + 2 * r) * kColStride * sizeof(float) <= 60 * 1024;
}

}  // namespace

bool gauss_f32_fast_supported(int w, int c, const TapsF32 &taps)
{
    const int r = taps.ksize / 2;
    if (c != 1 && c != 3)
        return false;
    if (r > 127 || w < 1)
        return false;
    return ((size_t)w * c + 2 * (size_t)r * c + 32) * sizeof(float) <= 64 * 1024;   // staged row fits LDS
}

int launch_gauss_f32_fast(const float *src, float *dst, float *scratch, int n, int h, int w, int c,
                          const TapsF32 &taps, hipStream_t st)
{
    VA_REQUIRE(gauss_f32_fast_supported(w, c, taps), "float gaussian: unsupported shape/kernel");
    if (n == 0 || h == 0)
        return VA_OK;
    const int r = taps.ksize / 2;
    const int rw = w * c;
    // + 32 samples: the last (partial) segment of a row reads a little past the staged row
    const size_t lds = ((size_t)rw + 2 * (size_t)r * c + 32) * sizeof(float);
    const unsigned rows = (unsigned)((size_t)n * h);
    const int vec4 = (rw % 4 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0);
    if (gauss_f32_packed_supported(w, c, taps)) {
        const size_t total_rows = (size_t)n * h;
        const size_t lds2 = ((size_t)rw + 2 * (size_t)r * c + 32) * sizeof(f2);
        const unsigned pairs = (unsigned)((total_rows + 1) / 2);
        const int vec4s = vec4 && (reinterpret_cast<uintptr_t>(scratch) % 16 == 0);
        if (c == 1)
            gauss_row2_f32_kernel<1, 15><<<pairs, kRowThreads, lds2, st>>>(src, scratch, total_rows, w, taps, vec4s);
        else
            gauss_row2_f32_kernel<3, 15><<<pairs, kRowThreads, lds2, st>>>(src, scratch, total_rows, w, taps, vec4s);
        VA_LAUNCH_CHECK("gauss_row2_f32_kernel");
        const int ncolt = cdiv(rw, kColTileCols), nrowt = cdiv(h, kColTileRows);
        const size_t ldsc = (size_t)(kColTileRows + 2 * r) * kColStride * sizeof(float);
        const int vec4c = (rw % 4 == 0) && (reinterpret_cast<uintptr_t>(scratch) % 16 == 0) &&
                          (reinterpret_cast<uintptr_t>(dst) % 8 == 0);
        gauss_col2_f32_kernel<<<(unsigned)((size_t)ncolt * nrowt * n), kBlock, ldsc, st>>>(
            scratch, dst, h, rw, ncolt, nrowt, taps, vec4c);
        VA_LAUNCH_CHECK("gauss_col2_f32_kernel");
        return VA_OK;
    }
    if (c == 1)
        gauss_row_f32_kernel<1, 17><<<rows, kBlock, lds, st>>>(src, scratch, w, taps, vec4);
    else
        gauss_row_f32_kernel<3, 27><<<rows, kBlock, lds, st>>>(src, scratch, w, taps, vec4);
    VA_LAUNCH_CHECK("gauss_row_f32_kernel");
    const int nx = cdiv(rw, kBlock), nbands = cdiv(h, 8);
    const unsigned grid = (unsigned)(8 * ((nx + 7) / 8) * nbands * n);
    gauss_col_f32_kernel<8><<<grid, kBlock, 0, st>>>(scratch, dst, h, rw, nx, nbands, taps);
    VA_LAUNCH_CHECK("gauss_col_f32_kernel");
    return VA_OK;
}

}  // namespace va
