// va_ccl.hip -- connected-component labelling on bit-packed masks (A7), per-label statistics
//               (A7 areas, A8 bounding boxes, A9 spatial moments), largest region.
//
// replaces labels, num = ndimage.measurements.label(mask)      video/analysis/regions.py:162
//          areas = [np.sum(labels == l) ...]; argmax            video/analysis/regions.py:165-174
//          find_bounding_box                                   video/analysis/regions.py:113-149
//          cv2.moments(mask.astype(np.uint8)) spatial moments  video/analysis/image.py:353
//
// Design (MI355X-first, not a pixel-per-thread port of a CPU two-pass labeller):
//   * the mask is bit-packed, so a 1080p frame is 60 dwords per row; ONE WAVE OWNS ONE ROW
//     (lane <-> dword) and finds runs of foreground with bit arithmetic (m & ~(m << 1)).
//   * the union-find forest has one node per RUN, identified by the linear pixel index of the
//     run's first pixel and stored sparsely in the int32 label image itself -- no extra
//     full-size scratch, and forest traffic scales with the number of runs, not pixels.
//   * unions are lock-free (atomicMin towards the smaller index), so a component's root is
//     its first pixel in raster order == SciPy's numbering rule: label = 1 + number of roots
//     with a smaller index.  Ranks come from a per-row root count + one scan per frame.
//   * the label image is written exactly once, coalesced (16 B per lane), by the paint pass;
//     per-label statistics are accumulated from run segments in the same pass (closed-form
//     sums of x, x^2, x^3 per segment), never from pixels.
// HBM traffic per frame ~= bit mask (re-read by each small pass, L2-resident) + 4 B/px labels.
#include "va_common.h"

namespace va {

namespace {

constexpr int kBlock = 256;            // 4 waves
constexpr int kRowsPerBlock = kBlock / kWave;
constexpr bool kNtStores = true;   // label image is write-once streaming output

// Forest loads/stores that the compiler may not cache in registers but the hardware may serve
// from L1/L2 (workgroup scope): a stale value is always an OLDER parent of the node, i.e. still
// an ancestor with a smaller index (parents only decrease), so walks terminate and unions stay
// correct -- the one place that needs the truth is the linking atomicMin itself, whose return
// value tells whether the node really was a root (see unite).  Serving the walks from cache
// instead of agent-scope (memory-side) loads is worth ~2x on the sparse passes.
__device__ __forceinline__ int ld_forest(const int32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void st_forest(int32_t *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// root of `a` with path halving (plain stores: any ancestor is a valid parent)
__device__ int find_root(int32_t *L, int a)
{
    for (;;) {
        int p = ld_forest(L + a);
        if (p == a)
            return a;
        int gp = ld_forest(L + p);
        if (gp == p)
            return p;
        st_forest(L + a, gp);
        a = gp;
    }
}

// read-only walk for the flatten pass: there every run writes its own final root, and a
// concurrent halving store from a passing walker could overwrite it with a mere ancestor
__device__ int find_root_ro(const int32_t *L, int a)
{
    for (;;) {
        int p = ld_forest(L + a);
        if (p == a)
            return a;
        a = p;
    }
}

// lock-free union: the larger root is linked under the smaller one
__device__ void unite(int32_t *L, int a, int b)
{
    for (;;) {
        a = find_root(L, a);
        b = find_root(L, b);
        if (a == b)
            return;
        if (a > b) {
            int t = a;
            a = b;
            b = t;
        }
        int old = atomicMin(L + b, a);  // a < b
        if (old == b)
            return;  // b was a root and now hangs under a
        b = old;     // b had been linked meanwhile: carry on with its (old) parent
    }
}

// first pixel of the run that contains pixel x (bit x of `row` must be set)
__device__ __forceinline__ int run_start(const uint32_t *row, int x)
{
    int w = x >> 5;
    uint32_t z = ~row[w] & (0xFFFFFFFFu >> (31 - (x & 31)));  // zero bits at or below x
    while (z == 0) {
        if (w == 0)
            return 0;
        --w;
        z = ~row[w];
    }
    return (w << 5) + 32 - __clz(z);
}

struct RowCtx {
    bool valid;
    int lane, f, y;
    size_t row;  // f*h + y
};

__device__ __forceinline__ RowCtx row_ctx(int h, size_t total_rows, unsigned block = blockIdx.x)
{
    RowCtx c;
    c.lane = threadIdx.x & (kWave - 1);
    c.row = (size_t)block * kRowsPerBlock + (threadIdx.x >> 6);
    c.valid = c.row < total_rows;
    size_t r = c.valid ? c.row : 0;
    c.f = (int)(r / h);
    c.y = (int)(r % h);
    return c;
}

// The sparse passes (init / link / flatten / rank) map a wave onto EIGHT consecutive rows:
// lane = (row = lane & 7, word group = lane >> 3), each lane walks a contiguous span of the
// row's words.  Compared with one row per wave this cuts the number of waves 8x and, more
// importantly, puts the unions of vertically adjacent runs (the same word column on
// consecutive rows -- blob edges) on different lanes, so their dependent load/atomic chains run
// in parallel instead of one after the other.
constexpr int kRowsPerWave = 8;
constexpr int kSparseRowsPerBlock = kRowsPerWave * (kBlock / kWave);   // 32

struct SpanCtx {
    bool valid;      // this lane has a row and a non-empty word span
    int lane, f, y;
    size_t row;      // f*h + y
    int w0, w1;      // word span [w0, w1)
};

__device__ __forceinline__ SpanCtx span_ctx(int h, int w32, size_t total_rows)
{
    SpanCtx c;
    c.lane = threadIdx.x & (kWave - 1);
    c.row = (size_t)blockIdx.x * kSparseRowsPerBlock + (size_t)(threadIdx.x >> 6) * kRowsPerWave +
            (c.lane & (kRowsPerWave - 1));
    const int g = c.lane >> 3, G = (w32 + 7) >> 3;
    c.w0 = g * G;
    c.w1 = min(w32, c.w0 + G);
    const bool has_row = c.row < total_rows;
    c.valid = has_row && c.w0 < c.w1;
    const size_t r = has_row ? c.row : 0;
    c.f = (int)(r / h);
    c.y = (int)(r % h);
    return c;
}

// The chip-wide labelling passes run either over all frames of a batch or over the frames the
// per-frame kernel handed back (more runs than its LDS table holds): grid = (row blocks of one
// frame, frame slots), every block loops over the listed frames with stride gridDim.y.
struct FrameList {
    const int32_t *list;    // frame indices; nullptr: frames 0 .. n-1
    const int32_t *count;   // number of listed frames (device memory), unused without a list
    int n;
};
__device__ __forceinline__ int frames_listed(const FrameList &fl) { return fl.list ? *fl.count : fl.n; }
__device__ __forceinline__ int frame_at(const FrameList &fl, int i) { return fl.list ? fl.list[i] : i; }

__device__ __forceinline__ SpanCtx span_ctx_in_frame(int f, int h, int w32)
{
    SpanCtx c;
    c.lane = threadIdx.x & (kWave - 1);
    c.f = f;
    c.y = blockIdx.x * kSparseRowsPerBlock + (int)(threadIdx.x >> 6) * kRowsPerWave +
          (c.lane & (kRowsPerWave - 1));
    const int g = c.lane >> 3, G = (w32 + 7) >> 3;
    c.w0 = g * G;
    c.w1 = min(w32, c.w0 + G);
    const bool has_row = c.y < h;
    c.valid = has_row && c.w0 < c.w1;
    if (!has_row)
        c.y = 0;
    c.row = (size_t)f * h + c.y;
    return c;
}

// ---- K1: every run's first pixel becomes a singleton tree ---------------------------------
// (the *_span functions are the per-lane bodies: the chip-wide kernels below map lanes to rows
// through blockIdx, the per-frame kernel's large-frame mode loops over its frame's rows)
// Words c0-1 .. c0+kGChunk of a mask row in global memory, every load issued before the first
// use (clamped addresses, masked values): m[0] = word c0-1, m[1 + k] = word c0 + k, 0 outside
// the row.  A word-by-word loop pays one L2 round trip per word.
constexpr int kGChunk = 8;
__device__ __forceinline__ void load_words(const uint32_t *row, int c0, int w32,
                                           uint32_t (&m)[kGChunk + 2])
{
#pragma unroll
    for (int k = 0; k < kGChunk + 2; k++)
        m[k] = row[min(max(c0 - 1 + k, 0), w32 - 1)];
#pragma unroll
    for (int k = 0; k < kGChunk + 2; k++) {
        const int wi = c0 - 1 + k;
        m[k] = (wi >= 0 && wi < w32) ? m[k] : 0u;
    }
}

__device__ __forceinline__ void init_span(const uint32_t *row, int32_t *L, int y, int w, int w32,
                                          int w0, int w1)
{
    for (int c0 = w0; c0 < w1; c0 += kGChunk) {
        uint32_t mw[kGChunk + 2];
        load_words(row, c0, w32, mw);
#pragma unroll
        for (int k = 0; k < kGChunk; k++) {
            if (c0 + k >= w1)
                break;
            uint32_t s = mw[k + 1] & ~((mw[k + 1] << 1) | (mw[k] >> 31));
            while (s) {
                int b = __ffs(s) - 1;
                s &= s - 1;
                int idx = y * w + ((c0 + k) << 5) + b;
                L[idx] = idx;
            }
        }
    }
}

__global__ void __launch_bounds__(kBlock)
ccl_init_kernel(const uint32_t *__restrict__ bits, int32_t *__restrict__ labels, int h, int w,
                int w32, FrameList fl)
{
    const int nf = frames_listed(fl);
    for (int i = blockIdx.y; i < nf; i += gridDim.y) {
        const SpanCtx c = span_ctx_in_frame(frame_at(fl, i), h, w32);
        if (c.valid)
            init_span(bits + c.row * w32, labels + (size_t)c.f * h * w, c.y, w, w32, c.w0, c.w1);
    }
}

// ---- K2: link runs of row y with runs of row y-1 --------------------------------------------
template <bool CONN8>
__device__ __forceinline__ void link_span(const uint32_t *row, int32_t *L, int y, int w, int w32,
                                          int w0, int w1)
{
    const uint32_t *up = row - w32;
    const int base = y * w, ubase = (y - 1) * w;
    // (word-by-word: here the chunked form of the other passes was slower -- the walks and the
    // backward searches of run_start dominate, and the unrolled body costs registers)
    uint32_t mp = w0 > 0 ? row[w0 - 1] : 0u, upv = w0 > 0 ? up[w0 - 1] : 0u;
    for (int wi = w0; wi < w1; wi++) {
        const uint32_t m = row[wi], u = up[wi];
        // vertical contacts: one union per maximal run of (m & u)
        uint32_t v = m & u;
        uint32_t vs = v & ~((v << 1) | ((mp & upv) >> 31));
        while (vs) {
            int b = __ffs(vs) - 1;
            vs &= vs - 1;
            int x = (wi << 5) + b;
            unite(L, base + run_start(row, x), ubase + run_start(up, x));
        }
        if (CONN8) {
            const uint32_t mn = wi + 1 < w32 ? row[wi + 1] : 0u, un = wi + 1 < w32 ? up[wi + 1] : 0u;
            // (y,x) ~ (y-1,x+1), needed only if neither (y-1,x) nor (y,x+1) is set
            uint32_t uR = (u >> 1) | (un << 31), mR = (m >> 1) | (mn << 31);
            uint32_t dr = m & uR & ~u & ~mR;
            while (dr) {
                int b = __ffs(dr) - 1;
                dr &= dr - 1;
                int x = (wi << 5) + b;
                // (y-1,x) is background, so (y-1,x+1) starts a run
                unite(L, base + run_start(row, x), ubase + x + 1);
            }
            // (y,x) ~ (y-1,x-1), needed only if neither (y-1,x) nor (y,x-1) is set
            uint32_t uL = (u << 1) | (upv >> 31), mL = (m << 1) | (mp >> 31);
            uint32_t dl = m & uL & ~u & ~mL;
            while (dl) {
                int b = __ffs(dl) - 1;
                dl &= dl - 1;
                int x = (wi << 5) + b;
                // (y,x-1) is background, so (y,x) starts a run
                unite(L, base + x, ubase + run_start(up, x - 1));
            }
        }
        mp = m;
        upv = u;
    }
}

template <bool CONN8>
__global__ void __launch_bounds__(kBlock)
ccl_link_kernel(const uint32_t *__restrict__ bits, int32_t *__restrict__ labels, int h, int w,
                int w32, FrameList fl)
{
    const int nf = frames_listed(fl);
    for (int i = blockIdx.y; i < nf; i += gridDim.y) {
        const SpanCtx c = span_ctx_in_frame(frame_at(fl, i), h, w32);
        if (c.valid && c.y > 0)
            link_span<CONN8>(bits + c.row * w32, labels + (size_t)c.f * h * w, c.y, w, w32, c.w0, c.w1);
    }
}

// ---- K3: flatten every run to its root, count roots per row ---------------------------------
__device__ __forceinline__ int flatten_span(const uint32_t *row, int32_t *L, int y, int w, int w32,
                                            int w0, int w1)
{
    int cnt = 0;
    for (int c0 = w0; c0 < w1; c0 += kGChunk) {
        uint32_t mw[kGChunk + 2];
        load_words(row, c0, w32, mw);
#pragma unroll
        for (int k = 0; k < kGChunk; k++) {
            if (c0 + k >= w1)
                break;
            uint32_t s = mw[k + 1] & ~((mw[k + 1] << 1) | (mw[k] >> 31));
            while (s) {
                int b = __ffs(s) - 1;
                s &= s - 1;
                int idx = y * w + ((c0 + k) << 5) + b;
                int r = find_root_ro(L, idx);
                if (r == idx)
                    cnt++;
                else
                    st_forest(L + idx, r);
            }
        }
    }
    return cnt;
}

// sum over the word groups of each row (lanes with equal lane & (RPW - 1)); RPW rows per wave
template <int RPW = 8>
__device__ __forceinline__ int row_sum(int v)
{
#pragma unroll
    for (int o = RPW; o < kWave; o <<= 1)
        v += __shfl_xor(v, o, kWave);
    return v;
}

// exclusive prefix over the word groups of each row (lanes r, r + RPW, r + 2 RPW, ...)
template <int RPW = 8>
__device__ __forceinline__ int row_prefix(int v, int lane)
{
    int incl = v;
#pragma unroll
    for (int o = RPW; o < kWave; o <<= 1) {
        int t = __shfl_up(incl, o, kWave);
        if (lane >= o)
            incl += t;
    }
    return incl - v;
}

// inclusive prefix over contiguous groups of G lanes (G = 8 or 16; g = lane % G): DPP row shifts, no
// LDS crossbar round trips (the strided row_prefix above pays three dependent ones)
template <int G>
__device__ __forceinline__ int seg_inclusive(int v, int g)
{
    int x = v, t;
    t = __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    x += g >= 1 ? t : 0;
    t = __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    x += g >= 2 ? t : 0;
    t = __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    x += g >= 4 ? t : 0;
    if (G > 8) {
        t = __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
        x += g >= 8 ? t : 0;
    }
    return x;
}

// exclusive prefix over the 64 lanes of the wave (DPP: Hillis-Steele inside the rows of 16, then
// the two row broadcasts); *total = sum over the wave
__device__ __forceinline__ int wave_prefix(int v, int *total)
{
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    *total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

__global__ void __launch_bounds__(kBlock)
ccl_flatten_kernel(const uint32_t *__restrict__ bits, int32_t *__restrict__ labels,
                   int32_t *__restrict__ row_cnt, int h, int w, int w32, FrameList fl)
{
    const int nf = frames_listed(fl);
    for (int i = blockIdx.y; i < nf; i += gridDim.y) {
        const SpanCtx c = span_ctx_in_frame(frame_at(fl, i), h, w32);
        int cnt = 0;
        if (c.valid)
            cnt = flatten_span(bits + c.row * w32, labels + (size_t)c.f * h * w, c.y, w, w32, c.w0, c.w1);
        cnt = row_sum(cnt);
        const int y = blockIdx.x * kSparseRowsPerBlock + (int)(threadIdx.x >> 6) * kRowsPerWave + c.lane;
        if (c.lane < kRowsPerWave && y < h)
            row_cnt[(size_t)c.f * h + y] = cnt;
    }
}

// ---- K4: per frame exclusive scan of the row counts ------------------------------------------
__global__ void __launch_bounds__(kBlock)
ccl_rowscan_kernel(const int32_t *__restrict__ row_cnt, int32_t *__restrict__ row_off,
                   int32_t *__restrict__ counts, int h, FrameList fl)
{
    __shared__ int part[kBlock];
    if ((int)blockIdx.x >= frames_listed(fl))
        return;
    const int f = frame_at(fl, blockIdx.x), t = threadIdx.x;
    const int32_t *cnt = row_cnt + (size_t)f * h;
    int32_t *off = row_off + (size_t)f * h;
    const int chunk = (h + kBlock - 1) / kBlock;
    const int y0 = t * chunk, y1 = min(h, y0 + chunk);
    int s = 0;
    for (int y = y0; y < y1; y++)
        s += cnt[y];
    part[t] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 256 partials
    for (int o = 1; o < kBlock; o <<= 1) {
        int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;  // exclusive prefix of this thread's chunk
    for (int y = y0; y < y1; y++) {
        off[y] = run;
        run += cnt[y];
    }
    if (t == kBlock - 1 && counts)
        counts[f] = part[kBlock - 1];
}

// ---- K5: roots get their final label, stored negated ----------------------------------------
// all lanes of the wave must call (shuffles); `valid` lanes own a row and a non-empty span
template <int RPW = 8>
__device__ __forceinline__ void rank_span(const uint32_t *row, int32_t *L, int y, int w, int w32,
                                          int w0, int w1, int lane, bool valid, int row_offset)
{
    // pass 1: roots in this lane's span (every root test of a chunk is loaded before any is used)
    int nroots = 0;
    if (valid) {
        for (int c0 = w0; c0 < w1; c0 += kGChunk) {
            uint32_t mw[kGChunk + 2];
            load_words(row, c0, w32, mw);
#pragma unroll
            for (int k = 0; k < kGChunk; k++) {
                if (c0 + k >= w1)
                    break;
                uint32_t s = mw[k + 1] & ~((mw[k + 1] << 1) | (mw[k] >> 31));
                while (s) {
                    int b = __ffs(s) - 1;
                    s &= s - 1;
                    int idx = y * w + ((c0 + k) << 5) + b;
                    nroots += L[idx] == idx;
                }
            }
        }
    }
    const int before = row_prefix<RPW>(nroots, lane);
    if (!valid)
        return;
    int k = row_offset + before;
    // pass 2: number them (roots still hold L[idx] == idx: only this lane rewrites its span)
    for (int c0 = w0; c0 < w1; c0 += kGChunk) {
        uint32_t mw[kGChunk + 2];
        load_words(row, c0, w32, mw);
#pragma unroll
        for (int q = 0; q < kGChunk; q++) {
            if (c0 + q >= w1)
                break;
            uint32_t s = mw[q + 1] & ~((mw[q + 1] << 1) | (mw[q] >> 31));
            while (s) {
                int b = __ffs(s) - 1;
                s &= s - 1;
                int idx = y * w + ((c0 + q) << 5) + b;
                if (L[idx] == idx)
                    L[idx] = -(++k);
            }
        }
    }
}

__global__ void __launch_bounds__(kBlock)
ccl_rank_kernel(const uint32_t *__restrict__ bits, int32_t *__restrict__ labels,
                const int32_t *__restrict__ row_off, int h, int w, int w32, FrameList fl)
{
    const int nf = frames_listed(fl);
    for (int i = blockIdx.y; i < nf; i += gridDim.y) {
        const SpanCtx c = span_ctx_in_frame(frame_at(fl, i), h, w32);
        rank_span(bits + c.row * w32, labels + (size_t)c.f * h * w, c.y, w, w32, c.w0, c.w1, c.lane,
                  c.valid, c.valid ? row_off[c.row] : 0);
    }
}

// ---- per-frame labelling: one workgroup owns one frame, forest in LDS ------------------------
// The five chip-wide passes above cost ~0.28 ms per 256 x 1080p batch although they move
// almost no data: every hop of a union-find walk is a dependent access to a sparse node in the
// 2 GB label image.  A frame's forest is tiny (a 1080p frame has ~3 k runs), so one workgroup
// per frame keeps it in LDS:
//   * runs get COMPACT ids in raster order: id = (runs in earlier rows, from an exclusive scan
//     of per-row run counts) + (run starts at or left of the pixel, by popcounts) - 1.  No
//     backward search for a run's first pixel, no node in global memory;
//   * every wave sweeps 8 rows at a time: the rows (contiguous in memory) are copied into the
//     wave's LDS stage with coalesced 16-byte loads, lanes (row, word group) then work on LDS;
//   * contacts between rows are first QUEUED per wave (slots from a wave-wide DPP scan of the
//     lanes' contact counts, plain LDS stores), then united with all lanes busy (in-place
//     unions serialise: a wave runs the longest lane's chain at every word);
//     unions are ds_min atomics towards the smaller id, so the root is again the component's
//     first run in raster order and label = 1 + number of roots with a smaller id;
//   * the only global traffic is the bit mask (read three times, L2-resident after the first)
//     and one sparse write per run: -(label) at the run's first pixel (bit 30 set for runs that
//     are not their component's first), which is what ccl_paint_kernel / the contour tracer read.
// Frames with more runs than the LDS table holds (noise, checkerboards) are labelled by the
// same workgroup with the forest in the label image (the *_span bodies above): same result, but
// one CU per frame -- a bounded slow path for pathological masks (handing such frames back to
// the chip-wide passes was measured: five near-empty launches cost every batch 4 %).
constexpr int kFrameThreads = 1024;
constexpr int kFrameWaves = kFrameThreads / kWave;                // 16
constexpr int kFrameLdsWords = 37 * 1024;                         // 148 KB, partitioned per launch
constexpr int kQueue = 128;                                       // queued pairs per wave
constexpr int kMinLdsRuns = 4096;                                 // else: chip-wide path
constexpr int kMaxFrameWords = 320 * 1024;                        // mask words per frame (1080p: 64.8 k, 4K: 259 k)
constexpr int kWantLdsRuns = 16384;                               // below: try 4 rows per wave (smaller stages)
constexpr int kNonRootBit = 1 << 30;

constexpr int kChunk = 8;                                         // words per lane and chunk
constexpr int kStagePad = 4;                                      // zero words left of a staged row

struct FrameLayout {   // word offsets into the kernel's LDS array
    int row_stride;    // staged row: kStagePad zeros, the row, zeros up to 8 lanes x chunks + 4; ODD (see frame_layout)
    int stage_words;   // per wave: 9 staged rows of the mask
    int queue_off, rowbase_off, parent_off, lds_runs;
};
// rpw rows per wave (8 or 4) <-> 64 / rpw lanes per row, each with span_chunks() chunks of 8 words
inline int span_chunks(int w32, int rpw)
{
    const int groups = kWave / rpw;
    return ((w32 + groups - 1) / groups + kChunk - 1) / kChunk;
}
inline FrameLayout frame_layout(int h, int w32, int rpw)
{
    FrameLayout l;
    // odd stride: a lane reads the same word k of ITS span (8-word stride within a row) in every step, so
    // the rows a half-wave covers must fall on different banks mod 8 (a stride that is a multiple of 4,
    // as 16-byte stage writes would need, puts 32 lanes on 4 banks: SQ_LDS_BANK_CONFLICT was 55 % of
    // the kernel's LDS cycles)
    l.row_stride = (kStagePad + (kWave / rpw) * span_chunks(w32, rpw) * kChunk + 4) | 1;
    l.stage_words = (rpw + 1) * l.row_stride;
    l.queue_off = kFrameWaves * l.stage_words;
    l.rowbase_off = l.queue_off + kFrameWaves * 2 * kQueue;
    l.parent_off = l.rowbase_off + ((h + 1 + 3) & ~3);
    l.lds_runs = kFrameLdsWords - l.parent_off;
    return l;
}

__device__ __forceinline__ int lds_ld(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int lds_find_root(int *P, int a)
{
    for (;;) {
        const int p = lds_ld(P + a);
        if (p == a)
            return a;
        const int gp = lds_ld(P + p);
        if (gp == p)
            return p;
        lds_st(P + a, gp);   // path halving: any ancestor is a valid parent
        a = gp;
    }
}
__device__ __forceinline__ int lds_find_root_ro(const int *P, int a)
{
    for (;;) {
        const int p = lds_ld(P + a);
        if (p == a)
            return a;
        a = p;
    }
}
__device__ __forceinline__ void lds_unite(int *P, int a, int b)
{
    for (;;) {
        a = lds_find_root(P, a);
        b = lds_find_root(P, b);
        if (a == b)
            return;
        if (a > b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(P + b, a);   // a < b
        if (old == b)
            return;
        b = old;
    }
}

// LDS traffic between the lanes of ONE wave: DS instructions of a wave execute in program
// order, so only the compiler has to be kept from reordering them
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct FrameSpan {
    bool valid;   // this lane has a row of the frame
    int lane, r, y, w0, w1;   // word span [w0, w1): NCH whole chunks (zero-padded in the stage)
};
template <int NCH, int RPW>
__device__ __forceinline__ FrameSpan frame_span(int y0, int h, int w32)
{
    FrameSpan c;
    c.lane = threadIdx.x & (kWave - 1);
    c.r = c.lane & (RPW - 1);
    c.y = y0 + c.r;
    c.w0 = (c.lane / RPW) * NCH * kChunk;
    c.w1 = min(w32, c.w0 + NCH * kChunk);    // large-frame mode reads global memory: real bounds
    c.valid = c.y < h;
    return c;
}

// NCH chunks of a staged row starting at word w0, each with the word on either side:
// m[j][0] = word c0-1, m[j][1 + k] = word c0 + k; the stage is zero outside the row
template <int NCH>
__device__ __forceinline__ void load_span(const int *row, int w0, uint32_t (&m)[NCH][kChunk + 2])
{
#pragma unroll
    for (int j = 0; j < NCH; j++)
#pragma unroll
        for (int k = 0; k < kChunk + 2; k++)
            m[j][k] = (uint32_t)row[w0 + j * kChunk - 1 + k];
}

// run starts in the span
template <int NCH>
__device__ __forceinline__ int count_starts(const uint32_t (&m)[NCH][kChunk + 2])
{
    int n = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++)
#pragma unroll
        for (int k = 0; k < kChunk; k++)
            n += __popc(m[j][k + 1] & ~((m[j][k + 1] << 1) | (m[j][k] >> 31)));
    return n;
}

// bits 0..b of a word
__device__ __forceinline__ uint32_t upto(int b) { return (2u << b) - 1u; }

// exclusive scan of one value per thread over the workgroup; *total (LDS) receives the sum
__device__ __forceinline__ int block_exclusive_scan(int v, int *s_part, int *total)
{
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(incl, o, kWave);
        if (lane >= o)
            incl += t;
    }
    if (lane == kWave - 1)
        s_part[wv] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < kFrameWaves; i++) {
        const int p = s_part[i];
        before += i < wv ? p : 0;
        all += p;
    }
    if (threadIdx.x == 0)
        *total = all;
    __syncthreads();
    return before + incl - v;
}

// in place: per-row counts -> exclusive prefix; returns the total through *total
__device__ __forceinline__ void scan_rows(int *rowbase, int h, int *s_part, int *total)
{
    const int per = (h + kFrameThreads - 1) / kFrameThreads;
    const int y0 = min(h, (int)threadIdx.x * per), y1 = min(h, y0 + per);
    int sum = 0;
    for (int y = y0; y < y1; y++)
        sum += rowbase[y];
    int run = block_exclusive_scan(sum, s_part, total);
    for (int y = y0; y < y1; y++) {
        const int n = rowbase[y];
        rowbase[y] = run;
        run += n;
    }
    __syncthreads();
}

// `table` != nullptr: instead of sweeping the rows a third time to drop one sparse word per run into
// the label image (which the paint pass then has to chase: L[first pixel] -> L[root]), the frame's
// labels leave as a compact table -- label of run id i at table[f * table_stride + i], first run id
// of row y at rowbase_g[f * h + y], mode[f] = 1 -- written linearly straight from the LDS forest.
// The paint pass numbers the runs of its row the same way (row base + run starts to the left) and
// needs ONE cached load per run.  Frames in large-frame mode keep the sparse convention (mode 0);
// so does the contour tracer, which reads roots at first pixels (table == nullptr).
#ifdef VA_CCL_STAMPS
__device__ long long g_ccl_stamps[16];
__device__ int g_ccl_xcc[2][16];      // XCC_ID of the first 16 workgroups of ccl_frame [0] and ccl_paint [1]
#define CCL_ACC(v, t0) do { long long t1_ = __builtin_amdgcn_s_memtime(); v += t1_ - t0; t0 = t1_; } while (0)
#define CCL_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ccl_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CCL_STAMP(i) do { } while (0)
#define CCL_ACC(v, t0) do { } while (0)
#endif
template <bool CONN8, int NCH, int RPW>
__global__ void __launch_bounds__(kFrameThreads)
ccl_frame_kernel(const uint32_t *__restrict__ bits, int32_t *__restrict__ labels,
                 int32_t *__restrict__ counts, int h, int w, int w32, FrameLayout lay, int direct,
                 int32_t *__restrict__ table, int table_stride, int32_t *__restrict__ rowbase_g,
                 int32_t *__restrict__ mode)
{
    __shared__ __attribute__((aligned(16))) int s_mem[kFrameLdsWords];
    __shared__ int s_part[kFrameWaves];
    __shared__ int s_total;

    const int f = blockIdx.x, tid = threadIdx.x, wv = tid >> 6, lane = tid & (kWave - 1);
#ifdef VA_CCL_STAMPS
    if (tid == 0 && f < 16)
        g_ccl_xcc[0][f] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;     // HW_REG_XCC_ID
#endif
    const uint32_t *fbits = bits + (size_t)f * h * w32;
    int32_t *L = labels + (size_t)f * h * w;
    int *stage = s_mem + wv * lay.stage_words + kStagePad;   // slot j <-> image row y0 - 1 + j
    const int rs = lay.row_stride;
    const bool spans_direct = NCH == 1 && direct;              // sweeps 1 and 3 read global memory (below)
    if (!spans_direct)
        for (int i = lane; i < lay.stage_words; i += kWave)   // the pads stay zero for good
            stage[i - kStagePad] = 0;
    int *queue = s_mem + lay.queue_off + wv * 2 * kQueue;
    int *rowbase = s_mem + lay.rowbase_off;
    int *parent = s_mem + lay.parent_off;
    constexpr int kRowsPerIter = kFrameWaves * RPW;         // rows per sweep step of the workgroup
    const int sweeps = (h + kRowsPerIter - 1) / kRowsPerIter;

    // stage image rows y0 - 1 ... y0 + 7 of this wave's sweep step (slot 0 is the row above y0).
    // All loads are issued before any is used -- one memory round trip per sweep step --
    // from clamped, always valid addresses: slots of rows outside the frame hold a copy of an
    // edge row, and no lane that reads them is `valid`.
    typedef int v4i __attribute__((ext_vector_type(4)));
    struct RowRegs {
        v4i v[RPW + 1];   // words lane, lane + 64, lane + 128, lane + 192 of each row (w32 <= 256)
    };
    const int nld = (w32 + kWave - 1) / kWave;       // word loads per row (wave-uniform)
    auto issue_rows = [&](int y0, RowRegs &rr) {
        const int c0 = min(lane, w32 - 1), c1 = min(lane + kWave, w32 - 1);
        const int c2 = min(lane + 2 * kWave, w32 - 1), c3 = min(lane + 3 * kWave, w32 - 1);
#pragma unroll
        for (int j = 0; j <= RPW; j++) {
            const uint32_t *src = fbits + (size_t)min(max(y0 - 1 + j, 0), h - 1) * w32;
            rr.v[j].x = (int)src[c0];
            if (nld > 1)
                rr.v[j].y = (int)src[c1];
            if (nld > 2) {
                rr.v[j].z = (int)src[c2];
                rr.v[j].w = (int)src[c3];
            }
        }
    };
    auto commit_rows = [&](const RowRegs &rr) {
        wave_sync();                                    // earlier readers of the stage are done
#pragma unroll
        for (int j = 0; j <= RPW; j++) {
            if (lane < w32)
                stage[j * rs + lane] = rr.v[j].x;
            if (nld > 1 && lane + kWave < w32)
                stage[j * rs + lane + kWave] = rr.v[j].y;
            if (nld > 2) {
                if (lane + 2 * kWave < w32)
                    stage[j * rs + lane + 2 * kWave] = rr.v[j].z;
                if (lane + 3 * kWave < w32)
                    stage[j * rs + lane + 3 * kWave] = rr.v[j].w;
            }
        }
        wave_sync();
    };
    // ---- 1. runs per row, exclusive scan -> first run id of every row -----------------------
    // (the next step's rows are in flight while this step's are counted: one memory round trip
    // per sweep step would otherwise be exposed 2 x 8.4 times per frame)
    CCL_STAMP(0);
    // Direct form (NCH == 1, rows 16-byte aligned): the lanes of a row are CONTIGUOUS (r = lane / G,
    // g = lane % G), each loads its 8-word span with two 16-byte loads -- no LDS stage, no stage
    // synchronisation; the word left/right of the span comes from the neighbour lane (DPP) and the
    // per-row prefix is a DPP scan over G lanes.
    constexpr int G = kWave / RPW;
    const int dr = lane / G, dg = lane % G;
    struct SpanRegs {
        v4i a, b;
    };
    auto load_direct = [&](int y, SpanRegs &sp) {    // zero outside the frame and past the row's end
        sp.a = v4i{0, 0, 0, 0};
        sp.b = v4i{0, 0, 0, 0};
        if (y >= 0 && y < h) {
            const uint32_t *src = fbits + (size_t)y * w32 + 8 * dg;
            if (8 * dg + 4 <= w32)
                sp.a = *reinterpret_cast<const v4i *>(src);
            if (8 * dg + 8 <= w32)
                sp.b = *reinterpret_cast<const v4i *>(src + 4);
        }
    };
    auto expand = [&](const SpanRegs &sp, uint32_t (&m)[NCH][kChunk + 2]) {
        m[0][1] = sp.a.x, m[0][2] = sp.a.y, m[0][3] = sp.a.z, m[0][4] = sp.a.w;
        m[0][5] = sp.b.x, m[0][6] = sp.b.y, m[0][7] = sp.b.z, m[0][8] = sp.b.w;
        const uint32_t left = __builtin_amdgcn_update_dpp(0u, m[0][8], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
        const uint32_t right = __builtin_amdgcn_update_dpp(0u, m[0][1], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
        m[0][0] = dg > 0 ? left : 0u;
        m[0][9] = dg < G - 1 ? right : 0u;
    };
    // the row stages are unused in this form: where they are large enough they keep every span's
    // prefix (run starts of its row to the left) from the first sweep for the second
    int *pref = s_mem;
    const bool keep_pref = h * G <= lay.queue_off;
    if (spans_direct) {
        SpanRegs pre;
        load_direct(wv * RPW + dr, pre);
        for (int it = 0; it < sweeps; it++) {      // (deeper prefetch changes nothing: the sweep is issue-bound)
            const int y = it * kRowsPerIter + wv * RPW + dr;
            const SpanRegs cur = pre;
            load_direct(y + kRowsPerIter, pre);
            uint32_t mw[NCH][kChunk + 2];
            expand(cur, mw);
            const int own = count_starts<NCH>(mw);
            const int n = seg_inclusive<G>(own, dg);
            if (dg == G - 1 && y < h)
                rowbase[y] = n;
            if (keep_pref && y < h)
                pref[y * G + dg] = n - own;
        }
    } else {
    RowRegs rr1;
    issue_rows(wv * RPW, rr1);
    for (int it = 0; it < sweeps; it++) {
        const int y0 = it * kRowsPerIter + wv * RPW;
        commit_rows(rr1);
        issue_rows(y0 + kRowsPerIter, rr1);        // clamped rows past the frame: harmless
        const FrameSpan c = frame_span<NCH, RPW>(y0, h, w32);
        uint32_t mw[NCH][kChunk + 2];
        load_span<NCH>(stage + (c.r + 1) * rs, c.w0, mw);
        int n = c.valid ? count_starts<NCH>(mw) : 0;
        n = row_sum<RPW>(n);
        if (lane < RPW && c.y < h)
            rowbase[c.y] = n;
    }
    }
    __syncthreads();
    CCL_STAMP(1);
    scan_rows(rowbase, h, s_part, &s_total);
    const int nruns = s_total;
    CCL_STAMP(2);

    if (table && tid == 0)
        mode[f] = nruns <= lay.lds_runs ? 1 : 0;
    if (nruns > lay.lds_runs) {
        // ---- large-frame mode: forest in the label image, same passes as the chip-wide path ---
        for (int it = 0; it < sweeps; it++) {
            const FrameSpan c = frame_span<NCH, RPW>(it * kRowsPerIter + wv * RPW, h, w32);
            if (c.valid && c.w0 < c.w1)
                init_span(fbits + (size_t)c.y * w32, L, c.y, w, w32, c.w0, c.w1);
        }
        __syncthreads();
        for (int it = 0; it < sweeps; it++) {
            const FrameSpan c = frame_span<NCH, RPW>(it * kRowsPerIter + wv * RPW, h, w32);
            if (c.valid && c.w0 < c.w1 && c.y > 0)
                link_span<CONN8>(fbits + (size_t)c.y * w32, L, c.y, w, w32, c.w0, c.w1);
        }
        __syncthreads();
        for (int it = 0; it < sweeps; it++) {
            const FrameSpan c = frame_span<NCH, RPW>(it * kRowsPerIter + wv * RPW, h, w32);
            int n = c.valid && c.w0 < c.w1 ? flatten_span(fbits + (size_t)c.y * w32, L, c.y, w, w32, c.w0, c.w1) : 0;
            n = row_sum<RPW>(n);
            if (lane < RPW && c.y < h)
                rowbase[c.y] = n;
        }
        __syncthreads();
        scan_rows(rowbase, h, s_part, &s_total);
        if (tid == 0 && counts)
            counts[f] = s_total;
        for (int it = 0; it < sweeps; it++) {
            const FrameSpan c = frame_span<NCH, RPW>(it * kRowsPerIter + wv * RPW, h, w32);
            const int yy = c.y < h ? c.y : 0;
            const bool ok = c.valid && c.w0 < c.w1;
            rank_span<RPW>(fbits + (size_t)yy * w32, L, c.y, w, w32, c.w0, c.w1, lane, ok,
                      ok ? rowbase[c.y] : 0);
        }
        return;
    }

    // ---- 2. singleton trees ----------------------------------------------------------------------
    for (int i = tid; i < nruns; i += kFrameThreads)
        parent[i] = i;
    __syncthreads();

    // The wave's queue of pairs: every lane counts what it will queue, a wave scan hands out the
    // slots, then the lanes write their pairs with plain LDS stores (an atomic per pair would
    // put a dependent LDS round trip into every loop iteration).  Pairs past the queue's end are
    // acted on at once by the caller.
    auto put = [&](int slot, int a, int b) -> bool {
        if (slot >= kQueue)
            return false;
        queue[2 * slot] = a;
        queue[2 * slot + 1] = b;
        return true;
    };

    // ---- 3. link runs of row y with runs of row y-1 -----------------------------------------------
#ifdef VA_CCL_STAMPS
    long long a_commit = 0, a_count = 0, a_pairs = 0, a_drain = 0, a_t0 = __builtin_amdgcn_s_memtime();
#endif
    // contacts of one sweep step: mw / uw = this lane's span of row y and of row y - 1, cc / cu = runs of
    // those rows that start left of the span
    auto link_rows = [&](bool act, int y, const uint32_t (&mw)[NCH][kChunk + 2],
                         const uint32_t (&uw)[NCH][kChunk + 2], int cc, int cu) {
        // contacts of this lane's span (the same bit tricks as below, counted)
        int ncontacts = 0;
        if (act) {
#pragma unroll
            for (int j = 0; j < NCH; j++)
#pragma unroll
                for (int k = 0; k < kChunk; k++) {
                    const uint32_t mp = mw[j][k], m = mw[j][k + 1], upv = uw[j][k], u = uw[j][k + 1];
                    const uint32_t v = m & u;
                    ncontacts += __popc(v & ~((v << 1) | ((mp & upv) >> 31)));
                    if (CONN8) {
                        const uint32_t mn = mw[j][k + 2], un = uw[j][k + 2];
                        const uint32_t uR = (u >> 1) | (un << 31), mR = (m >> 1) | (mn << 31);
                        const uint32_t uL = (u << 1) | (upv >> 31), mL = (m << 1) | (mp >> 31);
                        ncontacts += __popc(m & uR & ~u & ~mR) + __popc(m & uL & ~u & ~mL);
                    }
                }
        }
        int nq_total;
        int slot = wave_prefix(ncontacts, &nq_total);
        CCL_ACC(a_count, a_t0);
        if (nq_total == 0)                 // wave-uniform: nothing touches the rows above in this step
            return;
        if (act) {
            cc += rowbase[y] - 1;        // id of the run that holds pixel x = cc + starts in the
            cu += rowbase[y - 1] - 1;    // span at or left of x (same for the row above)
#pragma unroll
            for (int j = 0; j < NCH; j++) {
#pragma unroll
                for (int k = 0; k < kChunk; k++) {
                    const uint32_t mp = mw[j][k], m = mw[j][k + 1], upv = uw[j][k], u = uw[j][k + 1];
                    const uint32_t sc = m & ~((m << 1) | (mp >> 31)), su = u & ~((u << 1) | (upv >> 31));
                    // vertical contacts: one union per maximal run of (m & u)
                    const uint32_t v = m & u;
                    uint32_t vs = v & ~((v << 1) | ((mp & upv) >> 31));
                    while (vs) {
                        const int b = __ffs(vs) - 1;
                        vs &= vs - 1;
                        const int ia = cc + __popc(sc & upto(b)), ib = cu + __popc(su & upto(b));
                        if (!put(slot++, ia, ib))
                            lds_unite(parent, ia, ib);
                    }
                    if (CONN8) {
                        const uint32_t mn = mw[j][k + 2], un = uw[j][k + 2];
                        // (y,x) ~ (y-1,x+1), needed only if neither (y-1,x) nor (y,x+1) is set;
                        // (y-1,x) is background, so (y-1,x+1) starts the run after the starts <= x
                        const uint32_t uR = (u >> 1) | (un << 31), mR = (m >> 1) | (mn << 31);
                        uint32_t dr = m & uR & ~u & ~mR;
                        while (dr) {
                            const int b = __ffs(dr) - 1;
                            dr &= dr - 1;
                            const int ia = cc + __popc(sc & upto(b)), ib = cu + __popc(su & upto(b)) + 1;
                            if (!put(slot++, ia, ib))
                                lds_unite(parent, ia, ib);
                        }
                        // (y,x) ~ (y-1,x-1), needed only if neither (y-1,x) nor (y,x-1) is set: no
                        // run of the row above starts at x, so x-1 is in the run of the last start <= x
                        const uint32_t uL = (u << 1) | (upv >> 31), mL = (m << 1) | (mp >> 31);
                        uint32_t dl = m & uL & ~u & ~mL;
                        while (dl) {
                            const int b = __ffs(dl) - 1;
                            dl &= dl - 1;
                            const int ia = cc + __popc(sc & upto(b)), ib = cu + __popc(su & upto(b));
                            if (!put(slot++, ia, ib))
                                lds_unite(parent, ia, ib);
                        }
                    }
                    cc += __popc(sc);
                    cu += __popc(su);
                }
            }
        }
        // drain: every lane takes queued contacts, so the wave runs one union chain deep
        wave_sync();
        CCL_ACC(a_pairs, a_t0);
        const int nq = min(nq_total, kQueue);
        for (int i = lane; i < nq; i += kWave)
            lds_unite(parent, queue[2 * i], queue[2 * i + 1]);
    };
    CCL_STAMP(3);
    if (spans_direct) {
        SpanRegs pre, pre_up;
        load_direct(wv * RPW + dr, pre);
        load_direct(wv * RPW + dr - 1, pre_up);
        for (int it = 0; it < sweeps; it++) {
            const int y = it * kRowsPerIter + wv * RPW + dr;
            const SpanRegs cur = pre, up = pre_up;
            load_direct(y + kRowsPerIter, pre);
            load_direct(y + kRowsPerIter - 1, pre_up);
            const bool act = y < h && y > 0;
            uint32_t mw[NCH][kChunk + 2], uw[NCH][kChunk + 2];
            expand(cur, mw);
            expand(up, uw);
            int cc = 0, cu = 0;
            if (keep_pref) {
                if (act) {
                    cc = pref[y * G + dg];
                    cu = pref[(y - 1) * G + dg];
                }
            } else {
                const int nc = act ? count_starts<NCH>(mw) : 0;
                const int nu = act ? count_starts<NCH>(uw) : 0;
                cc = seg_inclusive<G>(nc, dg) - nc;
                cu = seg_inclusive<G>(nu, dg) - nu;
            }
            link_rows(act, y, mw, uw, cc, cu);
        }
    } else {
    RowRegs rr3;
    issue_rows(wv * RPW, rr3);
    for (int it = 0; it < sweeps; it++) {
        const int y0 = it * kRowsPerIter + wv * RPW;
        commit_rows(rr3);
        CCL_ACC(a_commit, a_t0);
        issue_rows(y0 + kRowsPerIter, rr3);
        const FrameSpan c = frame_span<NCH, RPW>(y0, h, w32);
        const bool act = c.valid && c.y > 0;
        uint32_t mw[NCH][kChunk + 2], uw[NCH][kChunk + 2];
        load_span<NCH>(stage + (c.r + 1) * rs, c.w0, mw);
        load_span<NCH>(stage + c.r * rs, c.w0, uw);
        const int nc = act ? count_starts<NCH>(mw) : 0;
        const int nu = act ? count_starts<NCH>(uw) : 0;
        // runs of the row that start left of this lane's span
        int cc = row_prefix<RPW>(nc, lane), cu = row_prefix<RPW>(nu, lane);
        link_rows(act, c.y, mw, uw, cc, cu);
        CCL_ACC(a_drain, a_t0);
    }
    }
#ifdef VA_CCL_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
        g_ccl_stamps[8] = a_commit; g_ccl_stamps[9] = a_count; g_ccl_stamps[10] = a_pairs; g_ccl_stamps[11] = a_drain;
    }
#endif
    __syncthreads();
    CCL_STAMP(4);

    // ---- 4. flatten, rank the roots in id (= raster) order -----------------------------------------
    const int ids_per_thread = (nruns + kFrameThreads - 1) / kFrameThreads;
    const int i0 = min(nruns, tid * ids_per_thread), i1 = min(nruns, i0 + ids_per_thread);
    int nroots = 0;
    for (int i = i0; i < i1; i++) {
        const int r = lds_find_root_ro(parent, i);
        if (r == i)
            nroots++;
        else
            lds_st(parent + i, r);
    }
    __syncthreads();                       // every parent is final (a root or a root's id)
    int k = block_exclusive_scan(nroots, s_part, &s_total);
    if (tid == 0 && counts)
        counts[f] = s_total;
    for (int i = i0; i < i1; i++)
        if (parent[i] == i)
            parent[i] = -(++k);            // roots: -(label); only this thread touches entry i
    __syncthreads();
    CCL_STAMP(5);

    if (table) {
        // ---- 5a. the labels leave as a table (coalesced), with the first run id of every row --------
        int32_t *tf = table + (size_t)f * table_stride;
        for (int i = tid; i < nruns; i += kFrameThreads) {
            int v = parent[i];
            if (v >= 0)
                v = parent[v];                 // a root's entry: -(label)
            tf[i] = -v;
        }
        int32_t *rb = rowbase_g + (size_t)f * h;
        for (int y = tid; y < h; y += kFrameThreads)
            rb[y] = rowbase[y];
        __syncthreads();
        CCL_STAMP(6);
        return;
    }
    // ---- 5. one sparse write per run: what the paint pass and the contour tracer read ---------------
    // The next sweep step's row loads are issued BEFORE this step's stores, and the stores are
    // branch-free buffer stores (out-of-range offset = dropped): vmcnt retires in order, so loads
    // issued after a store would wait for the store's (slow) acknowledgement every step.
    __amdgpu_buffer_rsrc_t lab_rsrc = __builtin_amdgcn_make_buffer_rsrc(L, 0, h * w * 4, 0x00027000);
    auto label_of = [&](int id) {
        int v = parent[id];
        if (v >= 0)
            v = -((-parent[v]) | kNonRootBit);
        return v;
    };
    if (spans_direct)                                          // the stages held span prefixes until now
        for (int i = lane; i < lay.stage_words; i += kWave)
            stage[i - kStagePad] = 0;
    RowRegs rr;
    issue_rows(wv * RPW, rr);
    for (int it = 0; it < sweeps; it++) {
        const int y0 = it * kRowsPerIter + wv * RPW;
        commit_rows(rr);
        issue_rows(y0 + kRowsPerIter, rr);         // clamped rows past the frame: harmless
        const FrameSpan c = frame_span<NCH, RPW>(y0, h, w32);
        uint32_t mw[NCH][kChunk + 2];
        load_span<NCH>(stage + (c.r + 1) * rs, c.w0, mw);
        const int nc = c.valid ? count_starts<NCH>(mw) : 0;
        int id = row_prefix<RPW>(nc, lane);
        int nq_total;
        int slot = wave_prefix(nc, &nq_total);
        if (c.valid) {
            id += rowbase[c.y];
            const int base = c.y * w + (c.w0 << 5);
#pragma unroll
            for (int j = 0; j < NCH; j++) {
#pragma unroll
                for (int k = 0; k < kChunk; k++) {
                    uint32_t sc = mw[j][k + 1] & ~((mw[j][k + 1] << 1) | (mw[j][k] >> 31));
                    while (sc) {
                        const int b = __ffs(sc) - 1;
                        sc &= sc - 1;
                        const int pix = base + ((j * kChunk + k) << 5) + b;
                        if (!put(slot++, pix, id))
                            L[pix] = label_of(id);
                        id++;
                    }
                }
            }
        }
        wave_sync();
        const int nq = min(nq_total, kQueue);
#pragma unroll
        for (int r = 0; r < kQueue / kWave; r++) {
            const int i = lane + r * kWave;
            const bool ok = i < nq;
            int pix = 0, v = 0;
            if (ok) {
                pix = queue[2 * i];
                v = label_of(queue[2 * i + 1]);
            }
            __builtin_amdgcn_raw_buffer_store_b32(v, lab_rsrc, ok ? (uint32_t)pix * 4u : 0xFFFFFFFFu, 0, 0);
        }
    }
}

// ---- K6: paint the label image (one coalesced write), optional per-label statistics ---------
__device__ __forceinline__ void stats_add(int64_t *st, int y, int xs, int len)
{
    const long long xe = xs + len - 1, a = xs - 1;
    const long long S0 = len;
    const long long S1 = (long long)len * (xs + xe) / 2;
    const long long S2 = xe * (xe + 1) * (2 * xe + 1) / 6 - a * (a + 1) * (2 * a + 1) / 6;
    const long long te = xe * (xe + 1) / 2, ta = a * (a + 1) / 2;
    const long long S3 = te * te - ta * ta;
    const long long Y = y;
    unsigned long long *u = reinterpret_cast<unsigned long long *>(st);
    atomicAdd(u + 0, (unsigned long long)S0);
    atomicAdd(u + 1, (unsigned long long)S1);
    atomicAdd(u + 2, (unsigned long long)(Y * S0));
    atomicAdd(u + 3, (unsigned long long)S2);
    atomicAdd(u + 4, (unsigned long long)(Y * S1));
    atomicAdd(u + 5, (unsigned long long)(Y * Y * S0));
    atomicAdd(u + 6, (unsigned long long)S3);
    atomicAdd(u + 7, (unsigned long long)(Y * S2));
    atomicAdd(u + 8, (unsigned long long)(Y * Y * S1));
    atomicAdd(u + 9, (unsigned long long)(Y * Y * Y * S0));
    long long *s = reinterpret_cast<long long *>(st);
    atomicMin(s + 10, (long long)xs);
    atomicMin(s + 11, Y);
    atomicMax(s + 12, xe);
    atomicMax(s + 13, Y);
}

// WORDS: mask words of a row the block keeps in LDS (64: rows up to 2048 pixels, 128: up to 4096 -- 4K)
template <bool STATS, int WORDS = kWave>
__global__ void __launch_bounds__(kBlock)
ccl_paint_kernel(const uint32_t *__restrict__ bits, int32_t *__restrict__ labels, int h, int w,
                 int w32, size_t total_rows, int64_t *__restrict__ stats, int max_labels,
                 int vec_ok, const int32_t *__restrict__ table, int table_stride,
                 const int32_t *__restrict__ rowbase_g, const int32_t *__restrict__ mode, int xcd_frames,
                 int vblocks)
{
    // vblocks = row blocks to paint.  gridDim.x == vblocks: one per workgroup; a smaller grid (a
    // multiple of 8, so that a workgroup's blocks stay on its XCD's frames) walks them with stride
    // gridDim.x -- the persistent form used when the pass runs beside other kernels
    // (va_pipeline_overlap): its footprint on a CU is then bounded by the grid, not by what is free.
    __shared__ uint32_t s_m[kRowsPerBlock][WORDS];
    __shared__ uint32_t s_heads[kRowsPerBlock][WORDS];
    __shared__ int32_t s_lab[kRowsPerBlock][WORDS][16];

    for (int vb = blockIdx.x; vb < vblocks; vb += gridDim.x) {
    RowCtx c = row_ctx(h, total_rows, (unsigned)vb);
#ifdef VA_CCL_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 16)
        g_ccl_xcc[1][blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
#endif
    if (xcd_frames) {
        // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs and the per-frame
        // labelling kernel ran frame f on XCD f % 8 (one workgroup per frame), so the run table, the
        // row bases and the mask rows of a frame are still in THAT XCD's L2: give every XCD the rows
        // of its own frames (frames come in groups of 8; xcd_frames = blocks per frame)
        const int xcd = vb & 7, q = vb >> 3;
        const int fr = (q / xcd_frames) * 8 + xcd;
        const int rb = (q % xcd_frames) * kRowsPerBlock + (threadIdx.x >> 6);
        c.f = fr;
        c.y = rb;
        c.valid = rb < h && (size_t)fr * h < total_rows;
        c.row = (size_t)(c.valid ? fr : 0) * h + (c.valid ? rb : 0);
        if (!c.valid) {
            c.f = 0;
            c.y = 0;
        }
    }
    const int wv = threadIdx.x >> 6;
    const uint32_t *row = bits + (c.valid ? c.row : 0) * w32;
    int32_t *L = labels + (size_t)c.f * h * w;
    int32_t *out = L + (size_t)c.y * w;
    int64_t *fstats = STATS ? stats + (size_t)c.f * max_labels * VA_STATS_STRIDE : nullptr;
    int carry = 0;  // label of the run that covers the last pixel of the previous chunk
    // run-table mode (see ccl_frame_kernel): labels come from table[run id], run ids from the row's
    // first id plus the run starts to the left (wave scan over the row's words)
    // (both loads are issued together: the row base must not wait for the mode word)
    const int md = (table != nullptr && c.valid) ? mode[c.f] : 0;
    int id_next = (table != nullptr && c.valid) ? rowbase_g[c.row] : 0;   // id of the next run that starts in this row
    const bool use_table = md != 0;                                      // wave-uniform
    const int32_t *tf = use_table ? table + (size_t)c.f * table_stride : nullptr;

    // The block's rows are contiguous in the label image.  Where the block keeps whole rows in LDS (w32 <= WORDS)
    // and is itself whole, phase A runs for every chunk first and phase B then writes the span's 1 KiB pieces
    // INTERLEAVED over the block's waves (wave v: pieces v, v + 4, ...; a piece may straddle two rows) instead of
    // one row per wave: at any moment the block's stores fall into neighbouring KiB
    // (tools/microbench/fill_rate.hip: the fewer consecutive KiB a wave streams, the closer to the fill rate).
    // (h % 4 == 0: a block never straddles two frames and its four rows are valid or invalid together)
#ifndef VA_PAINT_NO_INTERLEAVE
    const bool whole = w32 <= WORDS && vec_ok && (w & 3) == 0 && (h & (kRowsPerBlock - 1)) == 0 && c.valid;
#else
    const bool whole = false;
#endif
    for (int w0 = 0; w0 < w32; w0 += kWave) {
        const int so = whole ? w0 : 0;               // where this chunk's words sit in the LDS rows
        // ---- phase A: lane <-> word; label of every run (segment) inside the word
        const int wi = w0 + c.lane;
        uint32_t m = 0, heads = 0;
        int last_label = 0;
        uint32_t prev_bit = 0, starts_w = 0;
        if (c.valid && wi < w32) {
            m = row[wi];
            prev_bit = wi > 0 ? row[wi - 1] >> 31 : 0u;
            starts_w = m & ~((m << 1) | prev_bit);
        }
        int id_word = 0;                                   // id of the first run that starts in this word
        if (use_table) {
            // run starts in the words to the left: almost every word holds at most one start, so two
            // ballots do (a full wave scan only for rows with a word that starts several runs)
            const int cnt = __popc(starts_w);
            const unsigned long long b1 = __ballot(cnt >= 1), b2 = __ballot(cnt >= 2);
            if (b2 == 0) {                                  // wave-uniform
                id_word = id_next + __popcll(b1 & ((1ull << c.lane) - 1ull));
                id_next += __popcll(b1);
            } else {
                int incl = cnt;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const int t = __shfl_up(incl, o, kWave);
                    if (c.lane >= o)
                        incl += t;
                }
                id_word = id_next + incl - cnt;
                id_next += __shfl(incl, kWave - 1, kWave);
            }
        }
        if (c.valid && wi < w32) {
            const uint32_t prev = prev_bit;
            const uint32_t starts = starts_w;
            heads = starts | (m & 1u);
            uint32_t hb = heads;
            int k = 0;
            while (hb) {
                const int b = __ffs(hb) - 1;
                hb &= hb - 1;
                int lab;
                if (use_table) {
                    // the j-th start of the word is run id_word + j; a run entering from the left is
                    // the last one started before this word
                    const int j = __popc(starts & ((1u << b) - 1u));
                    lab = tf[(b == 0 && prev) ? id_word - 1 : id_word + j];
                } else if (b == 0 && prev) {
                    // the run enters from the left: look for its first pixel inside this chunk
                    int start = -1;
                    for (int ww = wi - 1; ww >= w0; --ww) {
                        uint32_t z = ~row[ww];
                        if (z) {
                            start = (ww << 5) + 32 - __clz(z);
                            break;
                        }
                    }
                    if (start < 0 && (w0 == 0 || (row[w0 - 1] >> 31) == 0u))
                        start = w0 << 5;  // the run begins exactly at the chunk's first pixel
                    if (start < 0) {
                        lab = carry;
                    } else {
                        int v = L[c.y * w + start];
                        if (v >= 0)
                            v = L[v];
                        lab = (v < 0 ? -v : v) & (kNonRootBit - 1);
                    }
                } else {
                    int v = L[c.y * w + (wi << 5) + b];
                    if (v >= 0)
                        v = L[v];  // root entry: -label, or +label if its row is already painted
                    lab = (v < 0 ? -v : v) & (kNonRootBit - 1);   // per-frame kernel: bit 30 = not a root
                }
                s_lab[wv][so + c.lane][k++] = lab;
                last_label = lab;
                if (STATS && lab >= 1 && lab <= max_labels) {
                    const uint32_t seg = m >> b;
                    const int len = (~seg) ? __ffs(~seg) - 1 : 32 - b;
                    stats_add(fstats + (size_t)(lab - 1) * VA_STATS_STRIDE, c.y, (wi << 5) + b,
                              len);
                }
            }
        }
        s_m[wv][so + c.lane] = m;
        s_heads[wv][so + c.lane] = heads;
        // label of the run covering the chunk's last pixel (lane 63's word, msb set)
        const int lane63_label = __shfl(last_label, kWave - 1, kWave);
        const uint32_t lane63_m = __shfl(m, kWave - 1, kWave);
        if (whole) {                                 // (phase B follows the last chunk, below)
            carry = (lane63_m >> 31) ? lane63_label : 0;
            continue;
        }
        __syncthreads();

        // ---- phase B: lanes write 4 consecutive pixels each, 1 KiB per wave-instruction
        if (c.valid) {
            const int xbase = w0 << 5;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int p = i * 256 + c.lane * 4;
                const int x = xbase + p;
                if (x >= w)
                    continue;
                const int wl = p >> 5;
                const uint32_t mw = s_m[wv][wl], hd = s_heads[wv][wl];
                int v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int bit = (p & 31) + j;
                    v[j] = 0;
                    if ((mw >> bit) & 1u) {
                        const int k = __popc(hd & (0xFFFFFFFFu >> (31 - bit))) - 1;
                        v[j] = s_lab[wv][wl][k];
                    }
                }
                if (vec_ok && x + 3 < w) {
                    typedef int v4i __attribute__((ext_vector_type(4)));
                    v4i val = {v[0], v[1], v[2], v[3]};
                    if (kNtStores)
                        __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(out + x));
                    else
                        *reinterpret_cast<v4i *>(out + x) = val;
                } else {
                    for (int j = 0; j < 4 && x + j < w; j++)
                        out[x + j] = v[j];
                }
            }
        }
        carry = (lane63_m >> 31) ? lane63_label : 0;
        if (w0 + kWave < w32)          // (a barrier after the last chunk would only hold the wave until its
            __syncthreads();           // stores are acknowledged)
    }
    if (whole) {
        __syncthreads();
        const int y0 = c.y - wv;                                 // first row of the block
        int32_t *span = L + (size_t)y0 * w;
        const int span_px = kRowsPerBlock * w, pieces = (span_px + 255) >> 8;
        for (int g = wv; g < pieces; g += kRowsPerBlock) {
            const int off = g * 256 + c.lane * 4;
            if (off >= span_px)
                continue;
            const int r = (off >= w) + (off >= 2 * w) + (off >= 3 * w);
            static_assert(kRowsPerBlock == 4, "row of a span offset: three comparisons");
            const int x = off - r * w;
            const int wl = x >> 5;
            const uint32_t mw = s_m[r][wl], hd = s_heads[r][wl];
            int v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int bit = (x & 31) + j;
                v[j] = 0;
                if ((mw >> bit) & 1u) {
                    const int k = __popc(hd & (0xFFFFFFFFu >> (31 - bit))) - 1;
                    v[j] = s_lab[r][wl][k];
                }
            }
            typedef int v4i __attribute__((ext_vector_type(4)));
            const v4i val = {v[0], v[1], v[2], v[3]};
            if (kNtStores)
                __builtin_nontemporal_store(val, reinterpret_cast<v4i *>(span + off));
            else
                *reinterpret_cast<v4i *>(span + off) = val;
        }
    }
    // (persistent form: the interleaved phase B reads the other waves' LDS slices, so the next row block's
    //  phase A must not start before every wave is through)
    if (vb + (int)gridDim.x < vblocks)
        __syncthreads();
    }   // vb
}

__global__ void __launch_bounds__(kBlock)
stats_init_kernel(int64_t *__restrict__ stats, size_t entries, int h, int w)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= entries * VA_STATS_STRIDE)
        return;
    int k = (int)(i % VA_STATS_STRIDE);
    stats[i] = k == 10 ? w : k == 11 ? h : (k == 12 || k == 13) ? -1 : 0;
}

// per-label statistics from an arbitrary label image: one thread = 32 consecutive pixels,
// equal-label segments are summed in closed form and flushed with atomics
__global__ void __launch_bounds__(kBlock)
stats_from_labels_kernel(const int32_t *__restrict__ labels, int h, int w, int w32,
                         size_t total_words, int64_t *__restrict__ stats, int max_labels)
{
    size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= total_words)
        return;
    const int wi = (int)(t % w32);
    const size_t rowi = t / w32;
    const int y = (int)(rowi % h);
    const int f = (int)(rowi / h);
    const int32_t *row = labels + rowi * (size_t)w;
    int64_t *fstats = stats + (size_t)f * max_labels * VA_STATS_STRIDE;
    const int x0 = wi << 5, x1 = min(w, x0 + 32);
    int cur = 0, xs = 0;
    if (x1 - x0 == 32 && (w & 3) == 0 && (reinterpret_cast<uintptr_t>(labels) & 15) == 0) {
        // the thread's 128 bytes as eight 16-byte loads, all in flight at once (32 dependent 4-byte loads at a
        // 128-byte stride between lanes made this kernel 1.9 ms per 64 x 1080p)
        int4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = reinterpret_cast<const int4 *>(row + x0)[k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int ls[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int l = ls[j], x = x0 + 4 * k + j;
                if (l != cur) {
                    if (cur >= 1 && cur <= max_labels)
                        stats_add(fstats + (size_t)(cur - 1) * VA_STATS_STRIDE, y, xs, x - xs);
                    cur = l;
                    xs = x;
                }
            }
        }
        if (cur >= 1 && cur <= max_labels)
            stats_add(fstats + (size_t)(cur - 1) * VA_STATS_STRIDE, y, xs, x1 - xs);
        return;
    }
    for (int x = x0; x <= x1; x++) {
        int l = x < x1 ? row[x] : 0;
        if (l != cur) {
            if (cur >= 1 && cur <= max_labels)
                stats_add(fstats + (size_t)(cur - 1) * VA_STATS_STRIDE, y, xs, x - xs);
            cur = l;
            xs = x;
        }
    }
}

// first maximum of the areas (np.argmax rule), one block per frame
__global__ void __launch_bounds__(kBlock)
largest_label_kernel(const int32_t *__restrict__ counts, const int64_t *__restrict__ stats,
                     int max_labels, int32_t *__restrict__ largest,
                     int64_t *__restrict__ largest_area)
{
    __shared__ long long s_area[kBlock];
    __shared__ int s_lab[kBlock];
    const int f = blockIdx.x, t = threadIdx.x;
    int cnt = counts[f];
    cnt = cnt > max_labels ? max_labels : cnt;
    const int64_t *st = stats + (size_t)f * max_labels * VA_STATS_STRIDE;
    long long best = -1;
    int bl = 0;
    for (int l = t; l < cnt; l += kBlock) {  // ascending l per thread: strict > keeps the first
        long long a = st[(size_t)l * VA_STATS_STRIDE];
        if (a > best) {
            best = a;
            bl = l + 1;
        }
    }
    s_area[t] = best;
    s_lab[t] = bl;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if (t < o) {
            long long a = s_area[t + o];
            int l = s_lab[t + o];
            if (a > s_area[t] || (a == s_area[t] && l != 0 && (s_lab[t] == 0 || l < s_lab[t]))) {
                s_area[t] = a;
                s_lab[t] = l;
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        largest[f] = s_lab[0];
        if (largest_area)
            largest_area[f] = s_lab[0] ? s_area[0] : 0;
    }
}

__global__ void __launch_bounds__(kBlock)
select_label_kernel(const int32_t *__restrict__ labels, const int32_t *__restrict__ largest,
                    uint8_t *__restrict__ mask, size_t frame_px, size_t total)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= total)
        return;
    int lm = largest[i / frame_px];
    mask[i] = (lm != 0 && labels[i] == lm) ? 1 : 0;
}

// ---- A8: outer border following (cv2.findContours RETR_EXTERNAL / CHAIN_APPROX_SIMPLE) -----
// OpenCV's icvFetchContour walk, started at a component's first raster pixel (= its forest
// root).  Direction codes: 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE (y grows downwards).
struct BitImage {
    const uint32_t *bits;
    int h, w, w32;
    __device__ __forceinline__ int at(int x, int y) const
    {
        if (x < 0 || x >= w || y < 0 || y >= h)
            return 0;
        return (bits[(size_t)y * w32 + (x >> 5)] >> (x & 31)) & 1u;
    }
};
__device__ __forceinline__ int code_dx(int s) { return (0x1 | 0x2 | 0x80) >> s & 1 ? 1 : ((0x8 | 0x10 | 0x20) >> s & 1 ? -1 : 0); }
__device__ __forceinline__ int code_dy(int s) { return (0x2 | 0x4 | 0x8) >> s & 1 ? -1 : ((0x20 | 0x40 | 0x80) >> s & 1 ? 1 : 0); }

// the eight neighbours of (x, y) as a mask, bit s = the pixel in direction s: eight independent loads (clamped
// coordinates, the result masked), one memory round trip per step of the walk instead of one per probed neighbour
__device__ __forceinline__ uint32_t neighbours8(const BitImage &im, int x, int y)
{
    uint32_t nb = 0;
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int xx = x + code_dx(s), yy = y + code_dy(s);
        const bool in = (unsigned)xx < (unsigned)im.w && (unsigned)yy < (unsigned)im.h;
        const int xc = min(max(xx, 0), im.w - 1), yc = min(max(yy, 0), im.h - 1);
        const uint32_t bit = (im.bits[(size_t)yc * im.w32 + (xc >> 5)] >> (xc & 31)) & 1u;
        nb |= (in ? bit : 0u) << s;
    }
    return nb;
}

template <class Emit>
__device__ void trace_outer_border(const BitImage &im, int x0, int y0, Emit &emit)
{
    int s_end = 4, s = 4, x1, y1;
    const uint32_t nb0 = neighbours8(im, x0, y0);
    do {
        s = (s - 1) & 7;
        x1 = x0 + code_dx(s);
        y1 = y0 + code_dy(s);
    } while (!((nb0 >> s) & 1u) && s != s_end);
    if (s == s_end) {   // single pixel
        emit(x0, y0);
        return;
    }
    int x3 = x0, y3 = y0, px = x0, py = y0, prev_s = s ^ 4;
    const long long max_steps = 8ll * im.h * im.w + 16;   // a border visits a pixel at most 8 times
    for (long long step = 0; step < max_steps; step++) {
        int x4 = x3, y4 = y3;
        const uint32_t nb = neighbours8(im, x3, y3);
        while (s < 15) {
            ++s;
            x4 = x3 + code_dx(s & 7);
            y4 = y3 + code_dy(s & 7);
            if ((nb >> (s & 7)) & 1u)
                break;
        }
        s &= 7;
        if (s != prev_s) {   // CHAIN_APPROX_SIMPLE: keep a point only where the direction changes
            emit(px, py);
            prev_s = s;
        }
        px += code_dx(s);
        py += code_dy(s);
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1)
            break;
        x3 = x4;
        y3 = y4;
        s = (s + 4) & 7;
    }
}

struct AreaEmit {   // shoelace sum over the emitted points (cv2.contourArea before *0.5 and abs)
    long long cross = 0;
    int n = 0, fx = 0, fy = 0, lx = 0, ly = 0;
    __device__ __forceinline__ void operator()(int x, int y)
    {
        if (n == 0) {
            fx = x;
            fy = y;
        } else {
            cross += (long long)lx * y - (long long)ly * x;
        }
        lx = x;
        ly = y;
        n++;
    }
    __device__ __forceinline__ long long twice_area() const
    {
        long long c = cross + ((long long)lx * fy - (long long)ly * fx);   // close the polygon
        return c < 0 ? -c : c;
    }
};

// every component: trace, keep max (contour area, then first-pixel index) per frame.  OpenCV
// returns the contours most-recent-first, so np.argmax's "first maximum" is the component
// whose first pixel comes LAST in raster order: the larger index wins ties.
__global__ void __launch_bounds__(kBlock)
contour_areas_kernel(const uint32_t *__restrict__ bits, const int32_t *__restrict__ forest,
                     unsigned long long *__restrict__ best, int h, int w, int w32,
                     size_t total_rows)
{
    const SpanCtx c = span_ctx(h, w32, total_rows);
    if (!c.valid)
        return;
    const uint32_t *row = bits + c.row * w32;
    const int32_t *L = forest + (size_t)c.f * h * w;
    BitImage im{bits + (size_t)c.f * h * w32, h, w, w32};
    uint32_t prev = c.w0 > 0 ? row[c.w0 - 1] >> 31 : 0u;
    for (int wi = c.w0; wi < c.w1; wi++) {
        const uint32_t m = row[wi];
        uint32_t s = m & ~((m << 1) | prev);
        prev = m >> 31;
        while (s) {
            const int b = __ffs(s) - 1;
            s &= s - 1;
            const int x = (wi << 5) + b, idx = c.y * w + x;
            const int v = L[idx];
            if (v >= 0 || (-v & kNonRootBit))
                continue;   // not a component's first pixel
            AreaEmit e;
            trace_outer_border(im, x, c.y, e);
            const unsigned long long key =
                ((unsigned long long)e.twice_area() << 32) | (unsigned long long)(unsigned)(idx + 1);
            atomicMax(best + c.f, key);
        }
    }
}

struct PointEmit {
    int32_t *pts;
    int cap;
    AreaEmit a;
    __device__ __forceinline__ void operator()(int x, int y)
    {
        if (a.n < cap) {
            pts[2 * a.n] = x;
            pts[2 * a.n + 1] = y;
        }
        a(x, y);
    }
};

__global__ void contour_points_kernel(const uint32_t *__restrict__ bits,
                                      const unsigned long long *__restrict__ best, int h, int w,
                                      int w32, int32_t *__restrict__ points, int max_points,
                                      int32_t *__restrict__ npoints, double *__restrict__ area)
{
    const int f = blockIdx.x;
    if (threadIdx.x != 0)
        return;
    const unsigned long long key = best[f];
    if (key == 0) {
        npoints[f] = 0;
        if (area)
            area[f] = 0.0;
        return;
    }
    const int idx = (int)(key & 0xFFFFFFFFull) - 1;
    BitImage im{bits + (size_t)f * h * w32, h, w, w32};
    PointEmit e{points + (size_t)f * max_points * 2, max_points, AreaEmit()};
    trace_outer_border(im, idx % w, idx / w, e);
    npoints[f] = e.a.n;
    if (area)
        area[f] = 0.5 * (double)e.a.twice_area();
}

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

// the per-frame LDS kernel takes frames whose row table and row stages leave room for at least
// kMinLdsRuns runs, for batches large enough that one CU per frame beats the chip-wide passes
// test hook (va_test_hook_labelling): 0 = the library chooses, 1 = chip-wide passes, 2 = per-frame
// kernel whatever the batch size; g_ccl_lds_runs > 0 caps the per-frame kernel's run table
static int g_ccl_path = 0;
static int g_ccl_lds_runs = 0;
static int g_ccl_sparse = 0;   // hook path 3: the library's choice of kernel, labels handed to the paint
                               // pass as sparse words in the label image (the round-1 convention)
static int g_ccl_no_direct = 0; // hook path 4: the per-frame kernel (as path 2), staging its rows in LDS even
                               // where it could read its spans straight from global memory
void ccl_test_hook(int path, int lds_runs)
{
    g_ccl_sparse = path == 3;
    g_ccl_no_direct = path == 4;
    g_ccl_path = path == 4 ? 2 : path == 3 ? 0 : path;
    g_ccl_lds_runs = lds_runs;
}

// rows per wave the per-frame kernel would use (8; 4 for wide frames, whose nine-row stages
// would leave too little LDS for the run table), 0 when the frame does not fit at all
static int frame_rows_per_wave(int h, int w32)
{
    if ((long long)h * w32 > kMaxFrameWords || w32 > 128)    // (the row staging covers 128 words)
        return 0;
    for (int rpw = 8; rpw >= 4; rpw >>= 1)
        if (span_chunks(w32, rpw) <= 2 &&
            frame_layout(h, w32, rpw).lds_runs >= (rpw == 8 ? kWantLdsRuns : kMinLdsRuns))
            return rpw;
    return span_chunks(w32, 8) <= 2 && frame_layout(h, w32, 8).lds_runs >= kMinLdsRuns ? 8 : 0;
}

bool ccl_frame_kernel_used(int n, int h, int w)
{
    if (g_ccl_path == 1)
        return false;
    const int rpw = frame_rows_per_wave(h, words_per_row(w));
    if (rpw == 0)
        return false;
    // Cost model from measurements of the whole chain (tools/labelling_paths.py; ms, frame size wf
    // in 1080p frames): the chip-wide passes scale with the batch and pay four more launches; one
    // workgroup per frame grows only mildly with the batch (contention), and its 4-rows-per-wave
    // form for wide frames is slower per word.
    if (g_ccl_path != 2) {
        const double wf = (double)h * words_per_row(w) / 64800.0, fill = (double)min(n, 256) / 256.0;
        const double t_frame = wf * (rpw == 8 ? 0.04 + 0.066 * fill : 0.075) + 0.01;
        const double t_chip = 0.277 * wf * n / 256.0 + 0.03;
        if (t_frame >= t_chip)
            return false;
    }
    return true;
}

// [row_cnt][row_off][run-label tables of the per-frame kernel][mode]
static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
size_t ccl_rows_workspace_bytes(int n, int h)
{
    return 2 * up256((size_t)n * h * sizeof(int32_t)) + up256((size_t)n * kFrameLdsWords * sizeof(int32_t)) +
           up256((size_t)n * sizeof(int32_t));
}

size_t ccl_workspace_bytes(int n, int h, int w)
{
    // [bit mask for the u8 entry point][rows workspace]
    size_t bits = (size_t)n * h * words_per_row(w) * sizeof(uint32_t);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(bits) + ccl_rows_workspace_bytes(n, h);
}

// workspace here = row_cnt + row_off only (the caller owns the bit mask)
// The labelling proper: everything up to (not including) the write of the label image.  `plan`
// (nullable: nobody will paint) receives what launch_ccl_paint needs, so that the caller may run the
// store-bound paint pass on another stream, beside the next batch's VALU-bound stages.
int launch_ccl_front(const uint32_t *bits, int32_t *labels, int32_t *counts, int n, int h, int w,
                     int connectivity, void *workspace, size_t ws_bytes, int64_t *stats, int max_labels,
                     hipStream_t st, StageProfiler *prof, CclPaintPlan *plan)
{
    const bool paint = plan != nullptr;
    if (plan)
        plan->n = 0;                             // (nothing to paint until the front has been enqueued)
#define VA_MARK(nm)      \
    do {                 \
        if (prof)        \
            prof->mark(nm, st); \
    } while (0)
    VA_REQUIRE(connectivity == 4 || connectivity == 8, "label: connectivity must be 4 or 8 (got %d)",
               connectivity);
    VA_REQUIRE(bits && labels && workspace, "label: NULL argument");
    VA_REQUIRE((size_t)h * (size_t)w < kMaxFramePixels,
               "label: frames of %dx%d exceed the supported 2^29 pixels (32-bit buffer descriptors "
               "address h*w*4 bytes of labels)", w, h);
    if (n == 0 || h == 0 || w == 0)
        return VA_OK;
    const int w32 = words_per_row(w);
    const size_t total_rows = (size_t)n * h;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    VA_REQUIRE(ws_bytes >= ccl_rows_workspace_bytes(n, h), "label: workspace too small");
    int32_t *row_cnt = (int32_t *)workspace;
    int32_t *row_off = (int32_t *)((char *)workspace + up(total_rows * sizeof(int32_t)));
    int32_t *run_table = (int32_t *)((char *)workspace + 2 * up(total_rows * sizeof(int32_t)));
    int32_t *frame_mode = (int32_t *)((char *)run_table + up((size_t)n * kFrameLdsWords * sizeof(int32_t)));
    bool table_mode = false;                 // set when the per-frame kernel hands its labels over as tables
    int table_stride = 0;

    const dim3 sgrid_all((unsigned)cdiv(h, kSparseRowsPerBlock), (unsigned)min(n, 4096));
    FrameList all{nullptr, nullptr, n};
    auto chip_wide = [&](const dim3 &sgrid, const FrameList &fl, int nscan, bool init) -> int {
        if (init) {
            ccl_init_kernel<<<sgrid, kBlock, 0, st>>>(bits, labels, h, w, w32, fl);
            VA_LAUNCH_CHECK("ccl_init_kernel");
            VA_MARK("ccl_init");
        }
        if (connectivity == 8)
            ccl_link_kernel<true><<<sgrid, kBlock, 0, st>>>(bits, labels, h, w, w32, fl);
        else
            ccl_link_kernel<false><<<sgrid, kBlock, 0, st>>>(bits, labels, h, w, w32, fl);
        VA_LAUNCH_CHECK("ccl_link_kernel");
        VA_MARK("ccl_link");
        ccl_flatten_kernel<<<sgrid, kBlock, 0, st>>>(bits, labels, row_cnt, h, w, w32, fl);
        VA_LAUNCH_CHECK("ccl_flatten_kernel");
        VA_MARK("ccl_flatten");
        ccl_rowscan_kernel<<<nscan, kBlock, 0, st>>>(row_cnt, row_off, counts, h, fl);
        VA_LAUNCH_CHECK("ccl_rowscan_kernel");
        VA_MARK("ccl_rowscan");
        ccl_rank_kernel<<<sgrid, kBlock, 0, st>>>(bits, labels, row_off, h, w, w32, fl);
        VA_LAUNCH_CHECK("ccl_rank_kernel");
        VA_MARK("ccl_rank");
        return VA_OK;
    };
    if (ccl_frame_kernel_used(n, h, w)) {
        // one workgroup per frame, forest in LDS
        const int rpw = frame_rows_per_wave(h, w32);
        FrameLayout lay = frame_layout(h, w32, rpw);
        if (g_ccl_lds_runs > 0)                            // test hook: force the large-frame mode
            lay.lds_runs = min(lay.lds_runs, g_ccl_lds_runs);
        const int nch = span_chunks(w32, rpw);
        // spans straight from 16-byte global loads (no LDS stage) where rows are 16-byte aligned
        const int direct = nch == 1 && (w32 % 4 == 0) && aligned(bits, 16) && !g_ccl_no_direct;
        table_mode = paint && !g_ccl_sparse; // (the contour tracer reads roots in the label image instead)
        table_stride = lay.lds_runs;
#define VA_FRAME_LAUNCH(C8, NCH, RPW)                                                                       \
    ccl_frame_kernel<C8, NCH, RPW><<<n, kFrameThreads, 0, st>>>(bits, labels, counts, h, w, w32, lay, direct,  \
                                                                 table_mode ? run_table : nullptr, lay.lds_runs, \
                                                                 row_off, frame_mode)
#define VA_FRAME_CASE(NCH, RPW)                 \
    if (nch == NCH && rpw == RPW) {             \
        if (connectivity == 8)                  \
            VA_FRAME_LAUNCH(true, NCH, RPW);    \
        else                                    \
            VA_FRAME_LAUNCH(false, NCH, RPW);   \
    }
        VA_FRAME_CASE(1, 8) VA_FRAME_CASE(2, 8) VA_FRAME_CASE(1, 4) VA_FRAME_CASE(2, 4)
#undef VA_FRAME_CASE
#undef VA_FRAME_LAUNCH
        VA_LAUNCH_CHECK("ccl_frame_kernel");
        VA_MARK("ccl_frame");
    } else {
        int rc = chip_wide(sgrid_all, all, n, true);
        if (rc)
            return rc;
    }
    if (!paint)
        return VA_OK;
    // frames follow the XCD of their labelling workgroup when the per-frame kernel ran (see the kernel)
    plan->bits = bits;
    plan->labels = labels;
    plan->n = n;
    plan->h = h;
    plan->w = w;
    plan->stats = stats;
    plan->max_labels = max_labels;
    plan->run_table = table_mode ? run_table : nullptr;
    plan->table_stride = table_stride;
    plan->row_off = row_off;
    plan->frame_mode = frame_mode;
    plan->xcd_frames = table_mode ? cdiv(h, kRowsPerBlock) : 0;
    plan->persistent_grid = 0;
#undef VA_MARK
    return VA_OK;
}

int launch_ccl_paint(const CclPaintPlan &pl, hipStream_t st, StageProfiler *prof)
{
#define VA_MARK(nm)      \
    do {                 \
        if (prof)        \
            prof->mark(nm, st); \
    } while (0)
    if (pl.n <= 0)
        return VA_OK;
    const int n = pl.n, h = pl.h, w = pl.w, w32 = words_per_row(pl.w);
    const size_t total_rows = (size_t)n * h;
    const int vec = (w % 4 == 0) && aligned(pl.labels, 16);
    const int grid = cdiv((long long)total_rows, kRowsPerBlock);          // paint: wave = row
    const int vblocks = pl.xcd_frames ? 8 * pl.xcd_frames * cdiv(n, 8) : grid;
    const int pgrid = pl.persistent_grid > 0 ? min(vblocks, pl.persistent_grid) : vblocks;
    // rows of 65 ... 128 mask words (4K) keep whole rows in LDS as well: 36 KB per block, 4 blocks per CU -- the
    // pass runs as fast at that occupancy (measured with a padded 1080p build)
    const bool wide = w32 > kWave && w32 <= 2 * kWave && vec && (w & 3) == 0 && (h & (kRowsPerBlock - 1)) == 0;
#define VA_PAINT_LAUNCH(ST, WORDS, STATSP, ML)                                                                      \
    ccl_paint_kernel<ST, WORDS><<<pgrid, kBlock, 0, st>>>(pl.bits, pl.labels, h, w, w32, total_rows, STATSP, ML, vec, \
                                                        pl.run_table, pl.table_stride, pl.row_off, pl.frame_mode,    \
                                                        pl.xcd_frames, vblocks)
    if (pl.stats && pl.max_labels > 0) {
        size_t entries = (size_t)n * pl.max_labels;
        stats_init_kernel<<<cdiv((long long)entries * VA_STATS_STRIDE, kBlock), kBlock, 0, st>>>(
            pl.stats, entries, h, w);
        VA_LAUNCH_CHECK("stats_init_kernel");
        VA_MARK("stats_init");
        if (wide)
            VA_PAINT_LAUNCH(true, 2 * kWave, pl.stats, pl.max_labels);
        else
            VA_PAINT_LAUNCH(true, kWave, pl.stats, pl.max_labels);
    } else {
        if (wide)
            VA_PAINT_LAUNCH(false, 2 * kWave, nullptr, 0);
        else
            VA_PAINT_LAUNCH(false, kWave, nullptr, 0);
    }
#undef VA_PAINT_LAUNCH
    VA_LAUNCH_CHECK("ccl_paint_kernel");
    VA_MARK("ccl_paint");
#undef VA_MARK
    return VA_OK;
}

int launch_ccl(const uint32_t *bits, int32_t *labels, int32_t *counts, int n, int h, int w,
               int connectivity, void *workspace, size_t ws_bytes, int64_t *stats, int max_labels,
               hipStream_t st, StageProfiler *prof, bool paint)
{
    CclPaintPlan plan;
    int rc = launch_ccl_front(bits, labels, counts, n, h, w, connectivity, workspace, ws_bytes, stats, max_labels,
                              st, prof, paint ? &plan : nullptr);
    if (rc || !paint)
        return rc;
    return launch_ccl_paint(plan, st, prof);
}

int launch_stats_from_labels(const int32_t *labels, int n, int h, int w, int max_labels,
                             int64_t *stats, hipStream_t st)
{
    VA_REQUIRE(labels && stats && max_labels > 0, "moments: NULL argument / max_labels <= 0");
    if (n == 0 || h == 0 || w == 0)
        return VA_OK;
    size_t entries = (size_t)n * max_labels;
    stats_init_kernel<<<cdiv((long long)entries * VA_STATS_STRIDE, kBlock), kBlock, 0, st>>>(
        stats, entries, h, w);
    VA_LAUNCH_CHECK("stats_init_kernel");
    const int w32 = words_per_row(w);
    size_t total_words = (size_t)n * h * w32;
    stats_from_labels_kernel<<<cdiv((long long)total_words, kBlock), kBlock, 0, st>>>(
        labels, h, w, w32, total_words, stats, max_labels);
    VA_LAUNCH_CHECK("stats_from_labels_kernel");
    return VA_OK;
}

int launch_largest_region(const int32_t *labels, const int32_t *counts, const int64_t *stats,
                          int n, int h, int w, int max_labels, int32_t *largest,
                          int64_t *largest_area, uint8_t *mask_out, hipStream_t st)
{
    VA_REQUIRE(counts && stats && largest && max_labels > 0, "largest_region: NULL argument");
    if (n == 0)
        return VA_OK;
    largest_label_kernel<<<n, kBlock, 0, st>>>(counts, stats, max_labels, largest, largest_area);
    VA_LAUNCH_CHECK("largest_label_kernel");
    if (mask_out) {
        VA_REQUIRE(labels, "largest_region: labels required for mask_out");
        size_t px = (size_t)h * w, total = px * n;
        if (total) {
            select_label_kernel<<<cdiv((long long)total, kBlock), kBlock, 0, st>>>(labels, largest,
                                                                                  mask_out, px, total);
            VA_LAUNCH_CHECK("select_label_kernel");
        }
    }
    return VA_OK;
}

}  // namespace va

namespace va {
int launch_largest_contour(const uint32_t *bits, const int32_t *forest, int n, int h, int w,
                           unsigned long long *best_keys, int32_t *points, int max_points,
                           int32_t *npoints, double *area, hipStream_t st)
{
    VA_REQUIRE(bits && forest && best_keys && points && npoints && max_points > 0,
               "largest_contour: bad argument");
    if (n == 0)
        return VA_OK;
    const int w32 = words_per_row(w);
    const size_t total_rows = (size_t)n * h;
    VA_HIP(hipMemsetAsync(best_keys, 0, sizeof(unsigned long long) * n, st));
    const int sgrid = cdiv((long long)total_rows, 32);
    contour_areas_kernel<<<sgrid, 256, 0, st>>>(bits, forest, best_keys, h, w, w32, total_rows);
    VA_LAUNCH_CHECK("contour_areas_kernel");
    contour_points_kernel<<<n, 64, 0, st>>>(bits, best_keys, h, w, w32, points, max_points, npoints,
                                          area);
    VA_LAUNCH_CHECK("contour_points_kernel");
    return VA_OK;
}
}  // namespace va

#ifdef VA_CCL_STAMPS
extern "C" int va_debug_ccl_stamps(long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(va::g_ccl_stamps), sizeof(long long) * 16);
}
extern "C" int va_debug_ccl_xcc(int *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(va::g_ccl_xcc), sizeof(int) * 32);
}
#endif
