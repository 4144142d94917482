// va_gauss_fused.hip -- fused single-channel 8-bit Gaussian blur (+ threshold + bit packing)
//
// replaces cv2.GaussianBlur(frame.astype(np.uint8), (0, 0), sigma)
//          FilterBlur._process_frame, video/filters.py:388-392
// and, in the pipeline, the BUILD-DEFINED FilterThreshold that follows it.
//
// One workgroup owns a vertical strip of TW columns of one frame and MARCHES DOWN it, sixteen
// rows per step, so every input byte is read from HBM once (plus a 16-column halo) and nothing
// but the final output is written:
//
//   global --(16 B/lane, prefetched one step ahead)--> s_in  : 16 staged input rows (u8)
//   row pass : v_dot4_u32_u8 over 36-byte windows of s_in with byte-shifted copies of the q8.8
//              taps (all taps <= 255, so four taps x four pixels are one instruction)
//              -> u16 results, two vertically adjacent rows packed per dword -> s_t ring
//   col pass : every thread owns one column, keeps RP+8 packed row pairs in registers and
//              produces 16 output rows with v_dot2_u32_u16 (two rows x two taps per instruction)
//   output   : u8 bytes, and/or (blur > thresh) as a bit mask via wave ballots (8 B per wave-row)
//
// Arithmetic is exactly the oracle's: row sum u16 (<= 255*256), column sum u32,
// (acc + 2^15) >> 16.  Borders: BORDER_REFLECT_101 in both directions ("virtual" rows above and
// below the frame are real rows fetched through the reflection).
// One barrier per step; the s_t ring has 4 pair-rows of slack so that the next step's row pass
// never overwrites what a slow wave still reads in this step's column pass.
#include "va_common.h"

namespace va {

namespace {

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot4(a, b, c, false);
}
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b),
                                  c, false);
}

// BORDER_REFLECT_101 for indices that overshoot the frame by less than its size (frames are
// at least 32 x 32 here and the halo is 16, see gauss_fused_supported), branch-free.  Indices
// further out belong to columns/rows whose results are never stored: they are only clamped
// so that the load stays inside the frame.
__device__ __forceinline__ int reflect101(int p, int len)
{
    p = p < 0 ? -p : p;
    p = p >= len ? 2 * (len - 1) - p : p;
    return min(max(p, 0), len - 1);
}

// staged columns left and right of the strip (>= padded radius): 16 for radii up to 16, 32 for the wide kernels
// (sigma up to 10.5: radius 17 ... 32)
constexpr int kMaxHalo = 32;
constexpr int kRowsPerStep = 16;   // 16 vs 8 rows per step: -1.5 % time (fixed costs amortised)

template <int HALO>
struct FusedWeights {
    uint32_t wrow[4][HALO / 2 + 1];  // byte weights for output o = 0..3 over the thread's (4 + 2 HALO)-byte window
    uint32_t we[HALO + 1];           // (tap[2j], tap[2j+1]) pairs for even output rows
    uint32_t wo[HALO + 1];           // (tap[2j-1], tap[2j]) pairs for odd output rows
};

template <int TW, int RP, int ROWS, bool HAS_DST, bool HAS_BITS>
__global__ void __launch_bounds__(TW)
gauss_fused_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                   uint32_t *__restrict__ bits, int thresh, int h, int w, int w32, int nstrips,
                   FusedWeights<(RP <= 16 ? 16 : 32)> wt)
{
    constexpr int kHalo = RP <= 16 ? 16 : 32;
    constexpr int WD = kHalo / 2 + 1;        // dwords of a row-pass window
    constexpr int IW = TW + 2 * kHalo;       // staged bytes per input row
    constexpr int IWD = IW / 4;
    constexpr int PPS = ROWS / 2;            // packed pair-rows produced per step
    constexpr int NP = PPS + RP;             // pair-rows the column pass reads
    constexpr int LAG = (PPS - 1 + RP) / PPS;   // steps between producing and consuming rows
    constexpr int RING = PPS * (LAG + 2);    // the next step's writes never touch this step's reads
    constexpr int VPR = IW / 16;             // 16-byte vectors per staged row
    constexpr int NV = VPR * ROWS;           // vectors staged per step
    constexpr int VPT = (NV + TW - 1) / TW;  // ... per thread
    constexpr int UPR = TW / 4;              // row-pass units (4 pixels x 2 rows) per pair-row
    constexpr int UPT = PPS * UPR / TW;      // units per thread and step
    static_assert(PPS * UPR % TW == 0, "row-pass units must divide evenly");
    static_assert(RP <= kHalo && (TW % 64) == 0 && NP <= PPS * (LAG + 1), "unsupported geometry");
    static_assert(ROWS == 8 || ROWS == 16, "ballot packing below handles 8 or 16 rows");

    __shared__ __attribute__((aligned(16))) uint32_t s_in[2][ROWS][IWD];
    __shared__ __attribute__((aligned(16))) uint32_t s_t[RING][TW];

    const int tid = threadIdx.x;
    const int strip = blockIdx.x % nstrips, f = blockIdx.x / nstrips;
    const int X0 = strip * TW;
    const uint8_t *img = src + (size_t)f * h * w;
    // A staged vector = 16 columns [gx0, gx0+16) of one row.  Frame rows are 16-byte aligned and
    // w % 16 == 0, so a vector lies entirely inside the frame or entirely outside.  Outside
    // vectors exist only next to the left/right frame border; they are the mirror image
    // (BORDER_REFLECT_101) of in-frame columns and are assembled in registers from two aligned
    // loads:   left : t[i] = col(K - i), K = -gx0 = 16 (or 32)      right: t[i] = col(w - 2 - J - i), J = gx0 - w = 0 (or 16)
    auto ld16 = [&](const uint8_t *p) { return *reinterpret_cast<const uint4 *>(p); };
    auto fetch = [&](int s, int v) -> uint4 {
        const int vrow = v / VPR, vcol = v % VPR;
        const int gx0 = X0 - kHalo + vcol * 16;
        const int yy = reflect101(ROWS * s + vrow - RP, h);
        const uint8_t *rowp = img + (size_t)yy * w;
        if (gx0 >= 0 && gx0 + 16 <= w)
            return ld16(rowp + gx0);
        if (gx0 < 0 && gx0 >= -kHalo) {
            const uint4 A = ld16(rowp - gx0 - 16), B = ld16(rowp - gx0);
            const uint32_t r0 = __builtin_bswap32(A.w), r1 = __builtin_bswap32(A.z),
                           r2 = __builtin_bswap32(A.y), r3 = __builtin_bswap32(A.x);
            return make_uint4((r0 << 8) | (B.x & 0xFFu), __builtin_amdgcn_alignbyte(r1, r0, 3),
                              __builtin_amdgcn_alignbyte(r2, r1, 3),
                              __builtin_amdgcn_alignbyte(r3, r2, 3));
        }
        if (gx0 >= w && gx0 < w + kHalo) {
            const uint4 C = ld16(rowp + 2 * w - 32 - gx0), D = ld16(rowp + 2 * w - 16 - gx0);
            const uint32_t r0 = __builtin_bswap32(D.w), r1 = __builtin_bswap32(D.z),
                           r2 = __builtin_bswap32(D.y), r3 = __builtin_bswap32(D.x);
            return make_uint4(__builtin_amdgcn_alignbyte(r1, r0, 1),
                              __builtin_amdgcn_alignbyte(r2, r1, 1),
                              __builtin_amdgcn_alignbyte(r3, r2, 1),
                              (r3 >> 8) | (C.w & 0xFF000000u));
        }
        return make_uint4(0, 0, 0, 0);   // columns whose results are never stored
    };
    auto stage = [&](int buf, int v, const uint4 &val) {
        *reinterpret_cast<uint4 *>(&s_in[buf][v / VPR][(v % VPR) * 4]) = val;
    };

#pragma unroll
    for (int i = 0; i < VPT; i++)
        if (tid + i * TW < NV)
            stage(0, tid + i * TW, fetch(0, tid + i * TW));
    __syncthreads();

    const int nsteps = (h + ROWS - 1) / ROWS + LAG;
    const int x = X0 + tid;                     // this thread's column in the column pass
    const int wave = tid >> 6, lane = tid & 63;

    // column-pass tap pairs live in VGPRs: together with the 36 row-pass dwords they do not fit
    // the SGPR file, and SGPR spills would cost one v_readlane per v_dot2
    uint32_t we[RP + 1], wo[RP + 1];
#pragma unroll
    for (int j = 0; j <= RP; j++) {
        we[j] = wt.we[j];
        wo[j] = wt.wo[j];
        asm volatile("" : "+v"(we[j]), "+v"(wo[j]));
    }

    for (int s = 0; s < nsteps; s++) {
        const int buf = s & 1;
        const bool more = s + 1 < nsteps;
        uint4 pre[VPT];
#pragma unroll
        for (int i = 0; i < VPT; i++) {
            pre[i] = make_uint4(0, 0, 0, 0);
            if (more && tid + i * TW < NV)
                pre[i] = fetch(s + 1, tid + i * TW);
        }

        // ---- row pass: unit u = (pair-row pr, 4 columns xq): virtual rows ROWS*s + 2pr, +1
#pragma unroll
        for (int ui = 0; ui < UPT; ui++) {
            const int u = tid + ui * TW;
            const int pr = u / UPR, xq = u % UPR;
            const uint32_t *r0 = &s_in[buf][2 * pr][xq];
            const uint32_t *r1 = &s_in[buf][2 * pr + 1][xq];
            uint32_t d0[WD], d1[WD];
#pragma unroll
            for (int d = 0; d < WD; d++) {
                d0[d] = r0[d];
                d1[d] = r1[d];
            }
            // eight independent accumulators, taps outermost: back-to-back v_dot4 on one
            // accumulator would need a wait state each
            uint32_t a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
#pragma unroll
            for (int d = 0; d < WD; d++) {
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    a0[o] = udot4(d0[d], wt.wrow[o][d], a0[o]);
                    a1[o] = udot4(d1[d], wt.wrow[o][d], a1[o]);
                }
                // scheduling fence on all but the two accumulators written last: naming those
                // too would make the compiler pad the dot -> use wait states with s_nops
                asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]),
                                  "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]));
            }
            uint32_t o4[4];
#pragma unroll
            for (int o = 0; o < 4; o++)
                o4[o] = a0[o] | (a1[o] << 16);   // both <= 255*256
            const int slot = (PPS * s + pr) % RING;
            *reinterpret_cast<uint4 *>(&s_t[slot][4 * xq]) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        }

        // ---- stage the next step's input rows into the other buffer
#pragma unroll
        for (int i = 0; i < VPT; i++)
            if (more && tid + i * TW < NV)
                stage(buf ^ 1, tid + i * TW, pre[i]);
        __syncthreads();

        // ---- column pass: output rows ROWS*g .. ROWS*g + ROWS-1 of column x
        const int g = s - LAG;
        if (g >= 0 && ROWS * g < h) {
            uint32_t pd[NP];
            int slot = (PPS * g) % RING;
#pragma unroll
            for (int j = 0; j < NP; j++) {
                pd[j] = s_t[slot][tid];
                slot = slot + 1 == RING ? 0 : slot + 1;
            }
            uint32_t acc[ROWS];
#pragma unroll
            for (int i = 0; i < ROWS; i++)
                acc[i] = 0;
#pragma unroll
            for (int j = 0; j <= RP; j++) {      // taps outermost: ROWS independent dot2 chains
#pragma unroll
                for (int i = 0; i < ROWS; i++)
                    acc[i] = udot2(pd[i / 2 + j], (i & 1) ? wo[j] : we[j], acc[i]);
                // pin the interleaved order (the scheduler would otherwise re-serialise the chains)
                if (ROWS == 8)
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]),
                                      "+v"(acc[4]), "+v"(acc[5]));
                else
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]),
                                      "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                                      "+v"(acc[8 % ROWS]), "+v"(acc[9 % ROWS]), "+v"(acc[10 % ROWS]),
                                      "+v"(acc[11 % ROWS]), "+v"(acc[12 % ROWS]), "+v"(acc[13 % ROWS]));
            }
            const int y0 = ROWS * g;
            const bool xin = x < w;
            const uint32_t acc_min = thresh < 0 ? 0u : (((uint32_t)thresh + 1u) << 16) - 32768u;
            const unsigned long long xmask = __ballot(xin);   // columns inside the frame
            uint32_t myword = 0;                 // lane l < 2*ROWS stores row l>>1, half l&1
#pragma unroll
            for (int i = 0; i < ROWS; i++) {
                if (HAS_DST && xin && y0 + i < h)
                    dst[((size_t)f * h + y0 + i) * w + x] = (uint8_t)((acc[i] + 32768u) >> 16);
                if (HAS_BITS) {
                    // ((acc + 2^15) >> 16) > thresh  <=>  acc >= ((thresh + 1) << 16) - 2^15.
                    // The compare result IS the 64-column bit row (an SGPR pair); v_writelane
                    // drops its halves into lanes 2i and 2i+1 of the word the wave will store.
                    const unsigned long long b =
                        __builtin_amdgcn_uicmp(acc[i], acc_min, 35 /* ICMP_UGE */) & xmask;
                    const uint32_t blo = (uint32_t)b, bhi = (uint32_t)(b >> 32);
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"(blo), "n"(2 * i));
                    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"(bhi), "n"(2 * i + 1));
                }
            }
            if (HAS_BITS) {
                const int wi = (X0 >> 5) + wave * 2 + (lane & 1);
                const int yy = y0 + (lane >> 1);
                if (lane < 2 * ROWS && yy < h && wi < w32)
                    bits[((size_t)f * h + yy) * w32 + wi] = myword;
            }
        }
    }
}

template <int HALO>
bool build_weights(const TapsQ8 &taps, int RP, FusedWeights<HALO> *wt)
{
    const int r = taps.ksize / 2;
    if (r > RP || RP > HALO)
        return false;
    uint32_t P[2 * kMaxHalo + 2] = {};   // zero-padded taps, centred at RP
    for (int i = 0; i < taps.ksize; i++) {
        if (taps.t[i] > 255)
            return false;
        P[i + RP - r] = taps.t[i];
    }
    memset(wt, 0, sizeof(*wt));
    for (int o = 0; o < 4; o++)
        for (int j = 0; j < 4 + 2 * HALO; j++) {
            const int i = j - (HALO - RP + o);
            if (i >= 0 && i <= 2 * RP)
                wt->wrow[o][j / 4] |= P[i] << (8 * (j % 4));
        }
    for (int j = 0; j <= RP; j++) {
        const uint32_t lo_e = P[2 * j], hi_e = (2 * j + 1 <= 2 * RP) ? P[2 * j + 1] : 0;
        const uint32_t lo_o = j > 0 ? P[2 * j - 1] : 0, hi_o = P[2 * j];
        wt->we[j] = lo_e | (hi_e << 16);
        wt->wo[j] = lo_o | (hi_o << 16);
    }
    return true;
}

int padded_radius(int r) { return r <= 4 ? 4 : (r <= 8 ? 8 : (r <= 16 ? 16 : (r <= 24 ? 24 : 32))); }

// strip width that wastes the fewest columns (ties: the wider strip, fewer halo re-reads)
int pick_tw(int w)
{
    const int cand[3] = {256, 192, 128};
    int best = 256, best_cols = 1 << 30;
    for (int tw : cand) {
        int cols = cdiv(w, tw) * tw;
        if (cols < best_cols) {
            best_cols = cols;
            best = tw;
        }
    }
    return best;
}

template <int TW, int HALO>
int launch_tw(const uint8_t *src, uint8_t *dst, uint32_t *bits, int thresh, int n, int h, int w,
              int RP, const FusedWeights<HALO> &wt, hipStream_t st)
{
    const int nstrips = cdiv(w, TW);
    const int w32 = words_per_row(w);
    dim3 grid((unsigned)(nstrips * n));
#define VA_GF_LAUNCH(RPV, D, B)                                                                  \
    gauss_fused_kernel<TW, RPV, kRowsPerStep, D, B><<<grid, TW, 0, st>>>(src, dst, bits, thresh, h, w, w32, nstrips, wt)
#define VA_GF_RP(RPV)                                      \
    do {                                                   \
        if (dst && bits)                                   \
            VA_GF_LAUNCH(RPV, true, true);                 \
        else if (dst)                                      \
            VA_GF_LAUNCH(RPV, true, false);                \
        else                                               \
            VA_GF_LAUNCH(RPV, false, true);                \
    } while (0)
    if constexpr (HALO == 16) {
        if (RP == 4)
            VA_GF_RP(4);
        else if (RP == 8)
            VA_GF_RP(8);
        else
            VA_GF_RP(16);
    } else {
        if (RP == 24)
            VA_GF_RP(24);
        else
            VA_GF_RP(32);
    }
#undef VA_GF_RP
#undef VA_GF_LAUNCH
    VA_LAUNCH_CHECK("gauss_fused_kernel");
    return VA_OK;
}

template <int HALO>
int launch_halo(const uint8_t *src, uint8_t *dst, uint32_t *bits, int thresh, int n, int h, int w, int RP,
                const TapsQ8 &taps, hipStream_t st)
{
    FusedWeights<HALO> wt;
    VA_REQUIRE(build_weights(taps, RP, &wt), "fused gaussian: cannot build the weight tables");
    switch (pick_tw(w)) {
    case 128:
        return launch_tw<128>(src, dst, bits, thresh, n, h, w, RP, wt, st);
    case 192:
        return launch_tw<192>(src, dst, bits, thresh, n, h, w, RP, wt, st);
    default:
        return launch_tw<256>(src, dst, bits, thresh, n, h, w, RP, wt, st);
    }
}

}  // namespace

bool gauss_fused_supported(int w, int h, const TapsQ8 &taps)
{
    if (w < 32 || h < 32 || (w % 16) != 0)   // single reflection; 16-byte aligned rows
        return false;
    const int r = taps.ksize / 2;
    if (taps.ksize < 3 || r > kMaxHalo)
        return false;
    if (r > 16 && (w < 64 || h < 64))        // the wide kernels: two mirrored vectors per row end, 32 mirrored rows
        return false;
    int sum = 0;
    for (int i = 0; i < taps.ksize; i++) {
        if (taps.t[i] > 255)
            return false;
        sum += taps.t[i];
    }
    return sum <= 256;          // 16-bit row sums, and (acc + 2^15) >> 16 <= 255 without saturation
}

int launch_gauss_fused_u8(const uint8_t *src, uint8_t *dst, uint32_t *bits, int thresh, int n,
                          int h, int w, const TapsQ8 &taps, hipStream_t st)
{
    VA_REQUIRE(src && (dst || bits), "fused gaussian: no output requested");
    VA_REQUIRE(gauss_fused_supported(w, h, taps), "fused gaussian: unsupported kernel size %d",
               taps.ksize);
    if (n == 0)
        return VA_OK;
    const int RP = padded_radius(taps.ksize / 2);
    VA_REQUIRE(reinterpret_cast<uintptr_t>(src) % 16 == 0,
               "fused gaussian: the frame buffer must be 16-byte aligned");
    if (RP <= 16)
        return launch_halo<16>(src, dst, bits, thresh, n, h, w, RP, taps, st);
    return launch_halo<32>(src, dst, bits, thresh, n, h, w, RP, taps, st);
}

}  // namespace va
