// va_gauss_f32_fused.hip -- float32 path of BASELINE.json configs[4] in two kernels:
//
//   1. ema_row_f32_kernel   adaptive background (EMA) + |frame - bg| + Gaussian ROW pass
//      (row_is_f32_kernel: the same with the radius at compile time, r = 4, 8 ... 36)
//   2. col_march_f32_kernel Gaussian COLUMN pass, marching down the frame
//      (col_sym_f32_kernel: the same with the radius at compile time, r = 4, 8 ... 36)
//
// replaces  FilterBackground(mode='ema') (BUILD-DEFINED, arithmetic of oracle vao_bg_ema_f32)
//           followed by cv2.GaussianBlur(float image, (0, 0), sigma),
//           video/analysis/active_contour.py:108
//
// Why two kernels and not three (EMA, rows, columns) or one: the EMA is a recurrence over time,
// the blur a stencil over space.  sigma = 9 means 146 FMA-class operations per sample against
// 8 compulsory bytes, so each pass is worth about as much VALU time (1.7 ms per 256 x 1080p x 3
// at the packed-fp32 rate) as its HBM time (2.1 ms); three kernels move 38 GB for 12.7 GB of
// compulsory traffic.  Fusing everything would need every workgroup to own a 2-d tile plus a
// 36-pixel halo on all sides for all frames (3x redundant work at 256 CUs).  So:
//
//   * kernel 1 keeps the background state in REGISTERS: a workgroup owns one contiguous chunk of
//     the flattened frame (about total/1024 samples, i.e. every CU gets the same work; the chunk
//     may straddle row ends) for ALL frames of the batch, reads each frame's chunk once, updates
//     its state, stages |frame - bg| (with the row halos; reflected at row ends) in LDS as float2
//     pairs (first half of the chunk, second half), runs the row pass with v_pk_fma_f32 (both
//     halves use the same tap, so every packed operand is register-aligned), and writes the
//     row-filtered samples back as 16-byte pieces.  The difference image never exists in HBM.
//   * kernel 2 gives a workgroup a 64-sample column strip of one frame and marches down 64 rows
//     at a time, keeping the last 2r rows in LDS, so the intermediate image is read exactly once
//     (the tile-per-workgroup version read it 2.1 times).
//
// Arithmetic order is the oracle's (vao_bg_ema_f32 / vao_gaussian_f32): no contraction in the
// EMA (built with -ffp-contract=off), in-order fmaf chain over the taps in the row pass, centre
// tap then symmetric pairs in the column pass -- results are bit-identical.
#include "va_common.h"

namespace va {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1)
        return 0;
    while (p < 0 || p >= len)
        p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

constexpr int kT = 256;        // threads per workgroup (4 waves, one per SIMD; 4 workgroups per CU)
constexpr int kP = 15;         // outputs per thread and half (multiple of the channel count)
constexpr int kNV = 7;         // float4 loads per thread and frame
constexpr int kMaxSeg = 4;     // row pieces a chunk may consist of
constexpr int kHalfCap = kP * kT;   // window positions per half

struct RowPlan {
    int L;        // samples per chunk (multiple of 4)
    int nchunks;
};

// a chunk of L flat samples touches at most this many rows
inline int max_segments(long long L, int rw) { return (int)((L + rw - 2) / rw) + 1; }

// Row pieces of one sub-chunk [cs, ce) of the flattened frame and where they sit in the padded
// array X = [halo | piece 0 | halo][halo | piece 1 | halo]...  (all uniform: SGPRs)
struct ChunkTab {
    int cs, ce;            // flat sample range owned
    int FS, FE;            // flat range loaded (with the halos of the first / last piece), float4 units
    int XL, Hoff;          // length of X, start of its second half
    int rs[kMaxSeg];       // flat index of the row start of piece k
    int qa[kMaxSeg], qb[kMaxSeg];   // in-row sample range of piece k (qa == qb: unused)
    int xoff[kMaxSeg];     // start of piece k's padded block in X
    int delta[kMaxSeg];    // a loaded flat sample f of piece k sits at X index f + delta[k] ...
    int blo[kMaxSeg], bhi[kMaxSeg];   // ... provided that index lies in [blo[k], bhi[k])
    int pofs[kMaxSeg];     // the output for flat sample f is O[f + pofs[k]]
};

__device__ __forceinline__ ChunkTab make_tab(int cs_in, int L, int total, int h, int rw, int halo)
{
    ChunkTab T;
    T.cs = min(cs_in, total);
    T.ce = min(T.cs + L, total);
    const int yfirst = T.cs / rw;
    T.XL = 0;
#pragma unroll
    for (int k = 0; k < kMaxSeg; k++) {
        T.rs[k] = min(yfirst + k, h) * rw;           // (rows past the frame: empty pieces)
        const int a = max(T.cs, T.rs[k]), b = min(T.ce, min(T.rs[k] + rw, total));
        T.qa[k] = a < b ? a - T.rs[k] : 0;
        T.qb[k] = a < b ? b - T.rs[k] : 0;
        T.xoff[k] = T.XL;
        if (a < b)
            T.XL += (T.qb[k] - T.qa[k]) + 2 * halo;
    }
    T.Hoff = (T.XL + 1) >> 1;
    T.FS = T.cs;
    T.FE = T.ce;
#pragma unroll
    for (int k = 0; k < kMaxSeg; k++) {
        if (T.qa[k] < T.qb[k]) {
            T.FS = min(T.FS, T.rs[k] + max(0, ((T.qa[k] - halo) >> 2) << 2));
            T.FE = max(T.FE, T.rs[k] + min(rw, ((T.qb[k] + halo + 3) >> 2) << 2));
        }
        T.delta[k] = T.xoff[k] + halo - T.rs[k] - T.qa[k];
        T.pofs[k] = T.xoff[k] - T.rs[k] - T.qa[k];
        T.blo[k] = T.xoff[k];
        T.bhi[k] = T.qa[k] < T.qb[k] ? T.xoff[k] + (T.qb[k] - T.qa[k]) + 2 * halo : T.xoff[k];
    }
    return T;
}

// Workgroup = 8 waves with two roles (wave-uniform): waves 0-3 LOAD (they own the background
// state, read the frames, stage |frame - bg| into LDS and write the results out), waves 4-7
// COMPUTE (the row pass).  A workgroup owns two sub-chunks; while the compute waves filter one
// of them from its LDS buffer, the loader waves stage the other one into the second buffer and
// already have the loads after that in flight, so HBM traffic and VALU work overlap inside every
// workgroup instead of depending on the phase of its neighbours (measured before the split: all
// workgroups of a CU ran their memory phases together and then their VALU phases together --
// 2.9 ms + 2.4 ms).  Two workgroups per CU: 2 compute + 2 loader waves per SIMD.
template <int C, bool EMA>
__global__ void __launch_bounds__(2 * kT, 4)
ema_row_f32_kernel(const float *__restrict__ frames, float *__restrict__ tmp, const float *__restrict__ bg,
                   float *__restrict__ bg_out, long long n_seen, float rate, int nframes, int h, int w, int L,
                   TapsF32 taps)
{
    extern __shared__ f2 lds[];    // [buffer 0][buffer 1][staging map]
    constexpr int WIN = kP + C;
    constexpr int PERIOD = WIN / C;
    static_assert(WIN % C == 0, "kP must be a multiple of C");
    const int rw = w * C;
    const int nt = taps.ksize, r = nt >> 1, halo = r * C;
    const int total = h * rw;                       // < 2^31: frames of at most 2^29 pixels, C <= 3
    const size_t fstride = (size_t)total;
    const int reach = (nt - 1) * C + kP;            // X entries a window start may read ahead
    const int s2n = kHalfCap + reach + 1;           // entries of one buffer (allocated: s2n + 7)
    f2 *const buf0 = lds, *const buf1 = lds + (s2n + 7);
    unsigned short *const sinfo = reinterpret_cast<unsigned short *>(lds + 2 * (s2n + 7));   // [2][kNV][kT]
    const bool loader = threadIdx.x < kT;           // wave-uniform
    const int tid = threadIdx.x & (kT - 1);

    const ChunkTab T0 = make_tab((2 * blockIdx.x) * L, L, total, h, rw, halo);
    const ChunkTab T1 = make_tab((2 * blockIdx.x + 1) * L, L, total, h, rw, halo);
    const int f0 = EMA ? 0 : blockIdx.y, f1 = EMA ? nframes : blockIdx.y + 1;
    const int Q = 2 * (f1 - f0);                    // phases: (frame, sub-chunk) in order

    // an X element lives in .x of entry u (u < s2n) and/or in .y of entry u - Hoff (u >= Hoff)
    auto x_store = [&](float *sf, int Hoff, int u, float v) {
        if (u < s2n)
            sf[2 * u] = v;
        if (u >= Hoff)
            sf[2 * (u - Hoff) + 1] = v;
    };
    auto x_load = [&](const float *sf, int Hoff, int u) -> float {
        return u < s2n ? sf[2 * u] : sf[2 * (u - Hoff) + 1];
    };

    // ======================================================================= loader waves
    if (loader) {
        // staging map, computed once and parked in LDS (16 bits per load slot): for slot m of this
        // thread (the float4 at flat index FS + 4*(tid + kT*m); its four samples lie in one row
        // because rw % 4 == 0) the X index of its first sample and a class: 0 nothing to stage,
        // 1 all four samples go to the first half only, 2 to the second half only, 3 anything else
        auto make_map = [&](const ChunkTab &T, int sub) {
#pragma unroll
            for (int m = 0; m < kNV; m++) {
                const int fm = T.FS + 4 * tid + 4 * kT * m;
                int info = 0;
                if (fm < T.FE) {
                    int dk = T.delta[0], lo = T.blo[0], hi = T.bhi[0];
#pragma unroll
                    for (int k = 1; k < kMaxSeg; k++)
                        if (fm >= T.rs[k] && T.qa[k] < T.qb[k]) {
                            dk = T.delta[k];
                            lo = T.blo[k];
                            hi = T.bhi[k];
                        }
                    const int u0 = fm + dk;
                    int nvalid = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        nvalid += (u0 + e >= lo && u0 + e < hi) ? 1 : 0;
                    const int cls = nvalid == 0 ? 0
                                    : (nvalid == 4 && u0 + 3 < T.Hoff) ? 1
                                    : (nvalid == 4 && u0 >= s2n)      ? 2
                                                                      : 3;
                    info = ((u0 + 8) << 2) | cls;
                }
                sinfo[(sub * kNV + m) * kT + tid] = (unsigned short)info;
            }
        };
        make_map(T0, 0);
        make_map(T1, 1);
        float st0[kNV][4], st1[kNV][4];             // background state of both sub-chunks (+ halos)
        auto load_state = [&](const ChunkTab &T, float (&st)[kNV][4]) {
#pragma unroll
            for (int m = 0; m < kNV; m++) {
                const int fm = T.FS + 4 * tid + 4 * kT * m;
                f4 v = f4{0.f, 0.f, 0.f, 0.f};
                if (EMA && fm < T.FE)
                    v = *reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(bg) + (unsigned)(4 * fm));
                st[m][0] = v.x, st[m][1] = v.y, st[m][2] = v.z, st[m][3] = v.w;
            }
        };
        load_state(T0, st0);
        load_state(T1, st1);
        const f2 rate2 = f2{rate, rate};
        f4 ldA[kNV];
        auto issue_loads = [&](const ChunkTab &T, int f, f4 (&ld)[kNV]) {
            const char *fr = reinterpret_cast<const char *>(frames + (size_t)f * fstride);   // uniform base + 32-bit lane byte offset
            int fb = T.FS + 4 * tid;
            asm volatile("" : "+v"(fb));            // (keeps seven load addresses out of loop-invariant registers)
#pragma unroll
            for (int m = 0; m < kNV; m++)
                if (fb + 4 * kT * m < T.FE)
                    ld[m] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(fr + (unsigned)(4 * (fb + 4 * kT * m))));
        };
        // background update + |difference| of the loaded sub-chunk into buffer `buf`
        auto stage = [&](const ChunkTab &T, int sub, f2 *buf, float (&st)[kNV][4], bool first, const f4 (&ld)[kNV]) {
            float *sf = reinterpret_cast<float *>(buf);
            int sbase = sub * kNV * kT + tid;
            asm volatile("" : "+v"(sbase));
#pragma unroll
            for (int m = 0; m < kNV; m++) {
                const int info = sinfo[sbase + m * kT];
                const int cls = info & 3, u0 = (info >> 2) - 8;
                if (cls != 0) {
                    f2 va = f2{ld[m].x, ld[m].y}, vb = f2{ld[m].z, ld[m].w};
                    if (EMA) {
                        f2 sa = f2{st[m][0], st[m][1]}, sb = f2{st[m][2], st[m][3]};
                        if (first) {
                            sa = va;
                            sb = vb;
                        }
                        const f2 da = va - sa, db = vb - sb;            // oracle order: d = x - bg
                        const f2 ta = rate2 * da, tb = rate2 * db;      // step = rate * d   (no contraction)
                        sa = sa + ta;                                   // bg = bg + step
                        sb = sb + tb;
                        va = __builtin_elementwise_abs(da);
                        vb = __builtin_elementwise_abs(db);
                        st[m][0] = sa.x, st[m][1] = sa.y, st[m][2] = sb.x, st[m][3] = sb.y;
                    }
                    const float v[4] = {va.x, va.y, vb.x, vb.y};
                    if (cls == 1) {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            sf[2 * (u0 + e)] = v[e];
                    } else if (cls == 2) {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            sf[2 * (u0 + e - T.Hoff) + 1] = v[e];
                    } else {
                        // block edges / the band both halves read: bounds of this sample's piece
                        const int fm = T.FS + 4 * tid + 4 * kT * m;
                        int lo = T.blo[0], hi = T.bhi[0];
#pragma unroll
                        for (int k = 1; k < kMaxSeg; k++)
                            if (fm >= T.rs[k] && T.qa[k] < T.qb[k]) {
                                lo = T.blo[k];
                                hi = T.bhi[k];
                            }
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (u0 + e >= lo && u0 + e < hi)
                                x_store(sf, T.Hoff, u0 + e, v[e]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // halos beyond a row end: BORDER_REFLECT_101, copied inside LDS
        auto fixup = [&](const ChunkTab &T, f2 *buf) {
            float *sf = reinterpret_cast<float *>(buf);
#pragma unroll
            for (int k = 0; k < kMaxSeg; k++) {
                if (T.qa[k] >= T.qb[k] || (T.qa[k] >= halo && T.qb[k] + halo <= rw))   // uniform: nothing past a row end
                    continue;
                for (int j = tid; j < halo; j += kT) {
                    const int ql = T.qa[k] - halo + j;                  // left halo slot j
                    if (ql < 0) {                                       // before the row start
                        const int px = -((-ql + C - 1) / C), ch = ql - px * C;
                        const int qs = reflect101(px, w) * C + ch;
                        x_store(sf, T.Hoff, T.xoff[k] + j, x_load(sf, T.Hoff, T.xoff[k] + halo + (qs - T.qa[k])));
                    }
                    const int qr = T.qb[k] + j;                         // right halo slot j
                    if (qr >= rw) {                                     // past the row end
                        const int px = qr / C, ch = qr - px * C;
                        const int qs = reflect101(px, w) * C + ch;
                        x_store(sf, T.Hoff, T.xoff[k] + halo + (T.qb[k] - T.qa[k]) + j,
                                x_load(sf, T.Hoff, T.xoff[k] + halo + (qs - T.qa[k])));
                    }
                }
            }
        };
        // results of phase q leave as 16-byte pieces (the loaders' share: slots 2m)
        auto copy_out = [&](const ChunkTab &T, const f2 *buf, int f, int part) {
            const float *O = reinterpret_cast<const float *>(buf);
            char *out = reinterpret_cast<char *>(tmp + (size_t)f * fstride);
            int ocs = T.cs + 4 * tid;
            asm volatile("" : "+v"(ocs));
#pragma unroll
            for (int m = 0; m < 4; m++) {
                // the 1024 samples of output slot (2m + part) lie in at most two pieces
                const int sa = T.cs + 4 * kT * (2 * m + part), sb = min(sa + 4 * kT, T.ce) - 1;
                if (sa >= T.ce)                      // uniform
                    break;
                int pa = T.pofs[0], pb = T.pofs[0], rsb = 0, ka = 0, kb = 0;
#pragma unroll
                for (int k = 1; k < kMaxSeg; k++)
                    if (T.qa[k] < T.qb[k]) {
                        if (sa >= T.rs[k]) {
                            pa = T.pofs[k];
                            ka = k;
                        }
                        if (sb >= T.rs[k]) {
                            pb = T.pofs[k];
                            rsb = T.rs[k];
                            kb = k;
                        }
                    }
                const int fo = ocs + 4 * kT * (2 * m + part);
                if (fo < T.ce) {
                    int p = fo + (fo >= rsb ? pb : pa);
                    if (kb - ka > 1) {               // uniform: rows shorter than a slot, three pieces
                        p = fo + T.pofs[0];
#pragma unroll
                        for (int k = 1; k < kMaxSeg; k++)
                            if (T.qa[k] < T.qb[k] && fo >= T.rs[k])
                                p = fo + T.pofs[k];
                    }
                    const f4 v = f4{O[p], O[p + 1], O[p + 2], O[p + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(out + (unsigned)(4 * fo)));
                }
            }
        };

        // The loader waves' few instructions must not queue behind the compute waves' dense
        // v_pk_fma stream on the shared SIMDs (measured: 8-12k cycles for ~300 instructions).
        __builtin_amdgcn_s_setprio(3);
        // prologue: phase 0's data staged, phase 1's loads in flight
        issue_loads(T0, f0, ldA);
        stage(T0, 0, buf0, st0, EMA && (n_seen + f0) == 0, ldA);
        issue_loads(T1, f0, ldA);
        __syncthreads();
        fixup(T0, buf0);
        __syncthreads();
        for (int q = 0; q < Q; q += 2) {
            const int f = f0 + (q >> 1);
            // ---- phase q: the compute waves filter buffer 0 (sub-chunk 0 of frame f) ----------
            // (one register set for the loads: a second one, so that the next loads could be issued
            // before this staging, spills the background state -- measured slower)
            stage(T1, 1, buf1, st1, EMA && (n_seen + f) == 0, ldA);     // sub-chunk 1 of frame f
            if (f + 1 < f1)
                issue_loads(T0, f + 1, ldA);
            __syncthreads();                                            // A: row pass done, staging done
            fixup(T1, buf1);
            __syncthreads();                                            // B: O written
            copy_out(T0, buf0, f, 0);
            __syncthreads();                                            // C: buffer 0 free
            // ---- phase q + 1: buffer 1 (sub-chunk 1 of frame f) ----------------------------------
            if (f + 1 < f1) {
                stage(T0, 0, buf0, st0, false, ldA);                    // sub-chunk 0 of frame f + 1
                issue_loads(T1, f + 1, ldA);
            }
            __syncthreads();
            if (f + 1 < f1)
                fixup(T0, buf0);
            __syncthreads();
            copy_out(T1, buf1, f, 0);
            __syncthreads();
        }
        if (EMA) {
            auto save_state = [&](const ChunkTab &T, float (&st)[kNV][4]) {
#pragma unroll
                for (int m = 0; m < kNV; m++) {
                    const int fm = T.FS + 4 * tid + 4 * kT * m;
                    if (fm >= T.cs && fm < T.ce)                      // only the sub-chunk's own samples
                        *reinterpret_cast<f4 *>(reinterpret_cast<char *>(bg_out) + (unsigned)(4 * fm)) =
                            f4{st[m][0], st[m][1], st[m][2], st[m][3]};
                }
            };
            save_state(T0, st0);
            save_state(T1, st1);
        }
        return;
    }

    // ====================================================================== compute waves
    const int i0 = kP * tid;
    // row pass: in-order fmaf chain over the taps, two halves per lane; results into O[] (the
    // same buffer: window start p <-> output X index p + halo) after barrier A
    auto row_pass = [&](const ChunkTab &T, const f2 *s2, f2 (&acc)[kP]) {
        if (i0 >= T.Hoff)
            return;
        f2 win[WIN];
#pragma unroll
        for (int j = 0; j < kP; j++) {
            acc[j] = f2{0.0f, 0.0f};
            win[j] = s2[i0 + j];
        }
        int k0 = 0;
        for (; k0 + PERIOD <= nt; k0 += PERIOD) {
            float wk[PERIOD];
#pragma unroll
            for (int t = 0; t < PERIOD; t++)
                wk[t] = taps.t[k0 + t];
#pragma unroll
            for (int t = 0; t < PERIOD; t++) {
                f2 nx[C];
#pragma unroll
                for (int c = 0; c < C; c++)
                    nx[c] = s2[i0 + (k0 + t) * C + kP + c];
                const f2 w2 = f2{wk[t], wk[t]};
#pragma unroll
                for (int j = 0; j < kP; j++)
                    acc[j] = pk_fma(win[(t * C + j) % WIN], w2, acc[j]);
#pragma unroll
                for (int c = 0; c < C; c++)
                    win[(t * C + kP + c) % WIN] = nx[c];
            }
        }
#pragma unroll
        for (int t = 0; t < PERIOD; t++) {
            const int k = k0 + t;
            if (k < nt) {                                // uniform
                const f2 w2 = f2{taps.t[k], taps.t[k]};
                f2 nx[C];
#pragma unroll
                for (int c = 0; c < C; c++)
                    nx[c] = s2[i0 + k * C + kP + c];
#pragma unroll
                for (int j = 0; j < kP; j++)
                    acc[j] = pk_fma(win[(t * C + j) % WIN], w2, acc[j]);
#pragma unroll
                for (int c = 0; c < C; c++)
                    win[(t * C + kP + c) % WIN] = nx[c];
            }
        }
    };
    auto write_O = [&](const ChunkTab &T, f2 *buf, const f2 (&acc)[kP]) {
        if (i0 >= T.Hoff)
            return;
        float *O = reinterpret_cast<float *>(buf);
#pragma unroll
        for (int j = 0; j < kP; j++) {
            O[i0 + j] = acc[j].x;
            O[T.Hoff + i0 + j] = acc[j].y;
        }
    };
    auto copy_out_c = [&](const ChunkTab &T, const f2 *buf, int f) {   // the compute waves' share: slots 2m + 1
        const float *O = reinterpret_cast<const float *>(buf);
        char *out = reinterpret_cast<char *>(tmp + (size_t)f * fstride);
        int ocs = T.cs + 4 * tid;
        asm volatile("" : "+v"(ocs));
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int sa = T.cs + 4 * kT * (2 * m + 1), sb = min(sa + 4 * kT, T.ce) - 1;
            if (sa >= T.ce)                          // uniform
                break;
            int pa = T.pofs[0], pb = T.pofs[0], rsb = 0, ka = 0, kb = 0;
#pragma unroll
            for (int k = 1; k < kMaxSeg; k++)
                if (T.qa[k] < T.qb[k]) {
                    if (sa >= T.rs[k]) {
                        pa = T.pofs[k];
                        ka = k;
                    }
                    if (sb >= T.rs[k]) {
                        pb = T.pofs[k];
                        rsb = T.rs[k];
                        kb = k;
                    }
                }
            const int fo = ocs + 4 * kT * (2 * m + 1);
            if (fo < T.ce) {
                int p = fo + (fo >= rsb ? pb : pa);
                if (kb - ka > 1) {                   // uniform: rows shorter than a slot, three pieces
                    p = fo + T.pofs[0];
#pragma unroll
                    for (int k = 1; k < kMaxSeg; k++)
                        if (T.qa[k] < T.qb[k] && fo >= T.rs[k])
                            p = fo + T.pofs[k];
                }
                const f4 v = f4{O[p], O[p + 1], O[p + 2], O[p + 3]};
                __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(out + (unsigned)(4 * fo)));
            }
        }
    };
    __syncthreads();        // (prologue of the loader waves)
    __syncthreads();
    for (int q = 0; q < Q; q += 2) {
        const int f = f0 + (q >> 1);
        f2 acc[kP];
        row_pass(T0, buf0, acc);
        __syncthreads();                                                // A
        write_O(T0, buf0, acc);
        __syncthreads();                                                // B
        copy_out_c(T0, buf0, f);
        __syncthreads();                                                // C
        row_pass(T1, buf1, acc);
        __syncthreads();
        write_O(T1, buf1, acc);
        __syncthreads();
        copy_out_c(T1, buf1, f);
        __syncthreads();
    }
}

// ------------------------------------------------------------------ row pass, radius at compile time
// Second form of kernel 1 for the radii of integer sigmas (r = 4 sigma = 4 ... 36).  Same ownership (a workgroup keeps the
// background state of its two sub-chunks in registers for all frames; loader waves / compute waves; two
// LDS buffers) with three changes that the fixed radius allows:
//   * the row pass is INPUT-stationary: a compute thread owns 31 consecutive outputs per half and walks
//     over its 31 + 2rC inputs once; each input feeds the ~31/C outputs it belongs to, with the tap a
//     compile-time index into SGPRs.  In-order per output (taps ascending), so the bits are the oracle's.
//     8 LDS reads per output instead of 16, no register window to rotate;
//   * therefore the staged samples can sit in LDS as PLAIN floats (the two halves of a lane's pair are
//     two dwords 31 * 256 floats apart: one ds_read2st64_b32), a loaded float4 is staged with ONE aligned
//     16-byte LDS write wherever it lies (no classes, no per-sample bounds: blocks are padded to float4
//     boundaries), and results leave as aligned 16-byte LDS reads;
//   * one workgroup of 8 waves per CU (256 VGPRs): both sub-chunks have their own load registers, so a
//     sub-chunk's next frame is requested as soon as its current one is staged -- two phases ahead.
namespace is {

// -DROWIS_STAMPS=1 | 2 | 3 (debug builds only, tools/debug/rowis_stamps.py): cycles every loader wave (1) and / or
// every compute wave (2) of one workgroup spends in each barrier and in each part of a phase, summed over the
// launch.  Scalar registers only (s_memtime, 32-bit sums): the kernel has no vector register to spare.
#ifdef ROWIS_STAMPS
__device__ unsigned int g_rowis_stamps[16][8];   // [wave][barrier A, B, C, part 1, part 2, part 3, -, total]
__device__ __forceinline__ unsigned rowis_now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return (unsigned)t;
}
#define ROWIS_T0_() const unsigned t0_ = rowis_now()
#define ROWIS_ACC_(idx) stamp_acc##idx += rowis_now() - t0_
#define ROWIS_BARRIER_(idx) do { ROWIS_T0_(); lds_barrier(); ROWIS_ACC_(idx); } while (0)
#endif
#if defined(ROWIS_STAMPS) && (ROWIS_STAMPS & 1)
#define LSTAMP_T0() ROWIS_T0_()
#define LSTAMP_ACC(idx) ROWIS_ACC_(idx)
#define LSTAMP_BARRIER(idx) ROWIS_BARRIER_(idx)
#else
#define LSTAMP_T0() do { } while (0)
#define LSTAMP_ACC(idx) do { } while (0)
#define LSTAMP_BARRIER(idx) lds_barrier()
#endif
#if defined(ROWIS_STAMPS) && (ROWIS_STAMPS & 2)
#define CSTAMP_T0() ROWIS_T0_()
#define CSTAMP_ACC(idx) ROWIS_ACC_(idx)
#define CSTAMP_BARRIER(idx) ROWIS_BARRIER_(idx)
#else
#define CSTAMP_T0() do { } while (0)
#define CSTAMP_ACC(idx) do { } while (0)
#define CSTAMP_BARRIER(idx) lds_barrier()
#endif

#ifndef ROWIS_CPOS
#define ROWIS_CPOS 4         // barrier C sits at ROWIS_CPOS / 8 of the compute waves' row pass
#endif
constexpr int kTW = 512;                 // loader threads: 8 waves (2 per SIMD)
constexpr int kTC = 512;                 // compute threads: 8 waves (2 per SIMD)
// outputs per compute thread and half: a template parameter P2 (odd: conflict-free LDS strides).  15 fits every
// shape the plan accepts; 13 is chosen when a sub-chunk still fits (1080p x 3 at one workgroup per CU: 13 086 of
// 13 312 window positions in use instead of 13 086 of 15 360, i.e. 13 % fewer FMAs issued on padding).
constexpr int kP2Max = 15;
constexpr int hoff_of(int p2) { return p2 * kTC; }       // floats between the halves of a lane's pair (multiple of 64)
constexpr int xw_of(int p2) { return 2 * p2 * kTC; }     // window starts per buffer
constexpr int kNS = 7;                   // float4 loads per loader thread and sub-chunk
constexpr int kSeg = 4;                  // row pieces per sub-chunk

struct Tab {
    int cs, ce;          // flat sample range owned
    int FS, FE;          // flat range loaded (multiples of 4)
    int XL;              // floats of X in use
    int rs[kSeg];        // flat index of the row start of piece k
    int qa[kSeg], qb[kSeg];   // in-row sample range of piece k (qa == qb: unused)
    int D[kSeg];         // in-row sample q of piece k sits at X[q + D[k]]; D % 4 == 0
};

__host__ __device__ inline int floor4(int v) { return (v >> 2) << 2; }
__host__ __device__ inline int ceil4(int v) { return ((v + 3) >> 2) << 2; }

__host__ __device__ inline Tab make_tab(int cs_in, int L, int total, int h, int rw, int halo)
{
    Tab T;
    T.cs = min(cs_in, total);
    T.ce = min(T.cs + L, total);
    const int yfirst = T.cs / rw;
    int end = 0;
    T.FS = T.cs;
    T.FE = T.ce;
#pragma unroll
    for (int k = 0; k < kSeg; k++) {
        T.rs[k] = min(yfirst + k, h) * rw;
        const int a = max(T.cs, T.rs[k]), b = min(T.ce, min(T.rs[k] + rw, total));
        T.qa[k] = a < b ? a - T.rs[k] : 0;
        T.qb[k] = a < b ? b - T.rs[k] : 0;
        T.D[k] = 0;
        if (a < b) {
            const int lo = floor4(T.qa[k] - halo), hi = ceil4(T.qb[k] + halo);
            T.D[k] = end + 4 - lo;                 // the block occupies X[end + 4, end + 4 + hi - lo)
            end += 4 + (hi - lo);
            T.FS = min(T.FS, T.rs[k] + max(0, lo));
            T.FE = max(T.FE, T.rs[k] + min(rw, hi));
        }
    }
    T.XL = end + 4;
    return T;
}

// a sub-chunk of L flat samples fits: at most kSeg row pieces, its padded blocks inside the two halves,
// its loads inside kNS float4 per loader thread
inline bool fits(long long L, int rw, int halo, int p2)
{
    const long long ns = (L + rw - 2) / rw + 1;
    return ns <= kSeg && L + ns * (2 * halo + 16) + 8 <= (long long)xw_of(p2) &&
           (L + 2 * halo + 8 + 3) / 4 <= (long long)kNS * kTW;
}

template <int C, int RAD, int kP2, bool EMA>
__global__ void __launch_bounds__(kTW + kTC)   // (HALO = RAD C <= 108 < kTW)
row_is_f32_kernel(const float *__restrict__ frames, float *__restrict__ tmp, const float *__restrict__ bg,
                  float *__restrict__ bg_out, long long n_seen, float rate, int nframes, int h, int w, int L,
                  TapsF32 taps)
{
    constexpr int HALO = RAD * C, NT = 2 * RAD + 1;
    constexpr int kHoff = hoff_of(kP2), kXW = xw_of(kP2);
    constexpr int NIN = kP2 + 2 * HALO;              // inputs a compute thread walks over
    constexpr int XCAP = kXW + 2 * HALO + 8;   // floats per buffer
    __shared__ __attribute__((aligned(16))) float Xs[2][XCAP];
    // where a loader thread's loaded float4s are staged: kNS 16-bit X offsets per thread and sub-chunk,
    // packed so that ONE 16-byte LDS read fetches them all (seven 2-byte reads were seven LDS round trips
    // per staging, each behind the compute waves' own LDS traffic)
    static_assert(kNS <= 8, "the staging map of a thread is one 16-byte LDS read");
    __shared__ __attribute__((aligned(16))) unsigned short smap[2][kTW][8];
    __shared__ __attribute__((aligned(16))) unsigned short cmap[2][kTW][8];   // copy-out: O index / 4 of a thread's slots

    constexpr bool ema = EMA;                        // (a run-time flag cost eight v_cndmask per staged float4)
    const int rw = w * C, total = h * rw;
    const size_t fstride = (size_t)total;
    const int f0 = ema ? 0 : blockIdx.y, f1 = ema ? nframes : blockIdx.y + 1;
    // Every CU runs the same phases in lock step, so the whole chip would load, then store, in bursts (the
    // copy-out stores block their waves for as long as the memory system takes to accept them: 0.7 of this
    // kernel's 4.4 ms).  A one-time start skew between workgroups -- nothing re-synchronises them afterwards --
    // spreads the bursts: 4.37 -> 4.16 ms per 256 x 1080p x 3 (0, 2, 4 and 8 groups of about half a phase
    // measured).  Only where a workgroup runs long enough to repay the wait (the whole batch, EMA mode).
    if (f1 - f0 >= 16)
        for (int k = 0; k < (int)(blockIdx.x & 7); k++)
            __builtin_amdgcn_s_sleep(127);
    const bool loader = threadIdx.x < kTW;           // wave-uniform
#ifdef ROWIS_STAMPS
    unsigned stamp_acc0 = 0, stamp_acc1 = 0, stamp_acc2 = 0, stamp_acc3 = 0, stamp_acc4 = 0, stamp_acc5 = 0;
    const unsigned stamp_start = rowis_now();
#endif
    const int tid = loader ? threadIdx.x : threadIdx.x - kTW;
    const Tab T0 = make_tab((2 * blockIdx.x) * L, L, total, h, rw, HALO);
    const Tab T1 = make_tab((2 * blockIdx.x + 1) * L, L, total, h, rw, HALO);

    // results of a phase leave as 16-byte pieces: output float4 fo of piece k sits at
    // O[fo - rs[k] + D[k] - HALO] (O = the buffer itself, indexed by window start).  Loader waves only:
    // a store blocks its wave for as long as the memory system takes to accept it, and the compute waves
    // have the next row pass to run meanwhile.  Where slot m of a thread (the float4 at flat index
    // cs + 4 (tid + kTW m)) sits in O is the same for every frame: looked up once (cmap, one 16-byte LDS read
    // per copy-out); slots past the chunk end are dropped by the buffer store's own range check, so the loop has
    // no per-lane branch and no address arithmetic (it had 18 vector instructions per slot).
    auto copy_out = [&](int sub, const Tab &T, const float *O, int f) {
        typedef unsigned int u4v __attribute__((ext_vector_type(4)));
        int t = tid;
        asm volatile("" : "+v"(t));
        const u4v cm = *reinterpret_cast<const u4v *>(&cmap[sub][t][0]);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tmp + (size_t)f * fstride, 0, T.ce * 4, 0x00027000);
        int ob = 4 * (T.cs + 4 * tid);
        asm volatile("" : "+v"(ob));
        const f4 *O4 = reinterpret_cast<const f4 *>(O);
        auto fetch = [&](int m) { return O4[(cm[m >> 1] >> (16 * (m & 1))) & 0xFFFFu]; };
        // Always kNS stores (those past the chunk end are dropped by the range check): a store count that depends
        // on a branch makes the compiler wait for ALL stores of the copy-out -- their acknowledgements, a memory
        // round trip -- before the next staging may use the frame loads issued ahead of them (vmcnt counts both).
        // The LDS read of slot m + 1 is in flight while slot m's store waits to be accepted.
        f4 nxt = fetch(0);
#pragma unroll
        for (int m = 0; m < kNS; m++) {
            const f4 v = nxt;
            if (m + 1 < kNS)
                nxt = fetch(m + 1);
#ifdef ROWIS_NO_STORE
            if (v.x == 123.456f)
#endif
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), rsrc, ob, 16 * kTW * m, 2 /* nt */);
            // Wait states behind every 16-byte buffer store whose soffset is a register.  The compiler keeps vector
            // writes away from the data registers of a wide store only when soffset is NOT a register (LLVM's rule
            // for this hazard: GCNHazardRecognizer::createsVALUHazard); on gfx950 a store with a register soffset
            // needs the distance too: a vector instruction that overwrote one of the four data registers right behind
            // the store corrupted that dword of the stored float4 -- intermittently, when the store had to wait for
            // the memory pipeline (found with an experimental layout whose address arithmetic reused the registers at
            // once; the kernel as shipped had the same instruction pair in its plain-blur variants).
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 3" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ======================================================================= loader waves
    if (loader) {
        // where slot m of this thread (the float4 at flat index FS + 4 (tid + kT m)) is staged:
        // X index / 4 + 1, or 0 for a slot without data
        auto make_map = [&](const Tab &T, int sub) {
#pragma unroll
            for (int m = 0; m < kNS; m++) {
                const int fm = T.FS + 4 * tid + 4 * kTW * m;
                int info = 0;
                if (fm < T.FE) {
#pragma unroll
                    for (int k = 0; k < kSeg; k++) {
                        const int q = fm - T.rs[k];
                        if (T.qa[k] < T.qb[k] && q >= max(0, floor4(T.qa[k] - HALO)) &&
                            q < min(rw, ceil4(T.qb[k] + HALO)))
                            info = ((q + T.D[k]) >> 2) + 1;
                    }
                }
                smap[sub][tid][m] = (unsigned short)info;
                // copy-out slot m: output float4 fo of piece k sits at O[fo - rs[k] + D[k] - HALO]
                const int fo = T.cs + 4 * tid + 4 * kTW * m;
                int p = HALO;
                if (fo < T.ce) {
                    p = fo - T.rs[0] + T.D[0];
#pragma unroll
                    for (int k = 1; k < kSeg; k++)
                        if (T.qa[k] < T.qb[k] && fo >= T.rs[k])
                            p = fo - T.rs[k] + T.D[k];
                }
                cmap[sub][tid][m] = (unsigned short)((p - HALO) >> 2);
            }
            if (kNS < 8) {
                smap[sub][tid][7] = 0;
                cmap[sub][tid][7] = 0;
            }
        };
        make_map(T0, 0);
        make_map(T1, 1);
        f4 st0[kNS], st1[kNS];                       // background state of both sub-chunks (+ halos)
        auto load_state = [&](const Tab &T, f4 (&st)[kNS]) {
#pragma unroll
            for (int m = 0; m < kNS; m++) {
                const int fm = T.FS + 4 * tid + 4 * kTW * m;
                st[m] = f4{0.f, 0.f, 0.f, 0.f};
                if (ema && fm < T.FE)
                    st[m] = *reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(bg) + (unsigned)(4 * fm));
            }
        };
        load_state(T0, st0);
        load_state(T1, st1);
        // One register set for the loads: a sub-chunk's next frame is requested when the OTHER sub-chunk
        // has just been staged, a whole phase (the compute waves' row pass) before it is needed.  The
        // loads are branch-free buffer loads (slots past FE read zero, frames past the last one re-read
        // it): behind exec-masked loads in conditional code the compiler waits vmcnt(0) before every
        // load that reuses a register.
        f4 ld[kNS];
        auto issue_loads = [&](const Tab &T, int f) {
            const int fc = min(f, f1 - 1);
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(frames) + (size_t)fc * fstride, 0, T.FE * 4, 0x00027000);
            int fb = 4 * (T.FS + 4 * tid);
            asm volatile("" : "+v"(fb));
#ifdef ROWIS_NO_LOAD
            if (rate == 123.0f)
#endif
#pragma unroll
            for (int m = 0; m < kNS; m++)
                ld[m] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, fb, 16 * kTW * m, 2 /* nt */));
        };
        // background update + |difference| of the loaded sub-chunk into X
        auto stage = [&](int sub, float *X, f4 (&st)[kNS], bool first) {
            int t = tid;
            asm volatile("" : "+v"(t));
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            const u4v map = *reinterpret_cast<const u4v *>(&smap[sub][t][0]);
            if (first) {                                 // (uniform; the very first frame only) bg = the frame itself
#pragma unroll
                for (int m = 0; m < kNS; m++)
                    st[m] = ld[m];
            }
#pragma unroll
            for (int m = 0; m < kNS; m++) {
                const int info = (int)((map[m >> 1] >> (16 * (m & 1))) & 0xFFFFu);
                if (info != 0) {
                    f4 v = ld[m];
                    if (ema) {
                        const f4 s = st[m];
                        const f4 d = v - s;                              // oracle order: d = x - bg
                        const f4 step = f4{rate, rate, rate, rate} * d;  // step = rate * d (no contraction)
                        st[m] = s + step;                                // bg = bg + step
                        v = __builtin_elementwise_abs(d);
                    }
                    *reinterpret_cast<f4 *>(X + 4 * (info - 1)) = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // halos beyond a row end: BORDER_REFLECT_101, copied inside LDS
        auto fixup = [&](const Tab &T, float *X) {
#pragma unroll
            for (int k = 0; k < kSeg; k++) {
                if (T.qa[k] >= T.qb[k] || (T.qa[k] >= HALO && T.qb[k] + HALO <= rw))   // uniform
                    continue;
                int j = tid;                          // (opaque: nothing per-thread is kept across phases --
                asm volatile("" : "+v"(j));           // it would be spilled, and a spill reload waits vmcnt(0))
                if (j < HALO) {                       // HALO <= kTW: one sample per thread and side
                    const int ql = T.qa[k] - HALO + j;                  // left halo sample j
                    if (ql < 0) {                                       // w > RAD: one reflection
                        const int px = -((-ql + C - 1) / C), ch = ql - px * C;
                        X[ql + T.D[k]] = X[-px * C + ch + T.D[k]];
                    }
                    const int qr = T.qb[k] + j;                         // right halo sample j
                    if (qr >= rw) {
                        const int px = qr / C, ch = qr - px * C;
                        X[qr + T.D[k]] = X[(2 * (w - 1) - px) * C + ch + T.D[k]];
                    }
                }
            }
        };

        __builtin_amdgcn_s_setprio(3);
        issue_loads(T0, f0);
        stage(0, Xs[0], st0, ema && (n_seen + f0) == 0);
        issue_loads(T1, f0);
        lds_barrier();
        fixup(T0, Xs[0]);
        lds_barrier();
        {
            // kNS stores that the range check drops (a buffer of no bytes): the loop is then entered the way its
            // back edge arrives -- a sub-chunk's loads, then kNS stores -- and the staging at its top waits for
            // the loads only (vmcnt(2 kNS - 1 - m)); entered with the loads alone outstanding, the compiler had to
            // assume vmcnt(kNS - 1 - m), which on every later pass waited for the previous copy-out's stores as well
#ifndef ROWIS_NO_DUMMY_STORES
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(tmp, 0, 0, 0x00027000);
#pragma unroll
            for (int m = 0; m < kNS; m++)
                __builtin_amdgcn_raw_buffer_store_b128(u4v{0u, 0u, 0u, 0u}, none, 0, 16 * m, 0);
#endif
        }
        lds_barrier();                                                  // (the C of the compute waves' first row pass)
        for (int f = f0; f < f1; f++) {
            // ---- the compute waves filter buffer 0 (sub-chunk 0 of frame f)
            { LSTAMP_T0(); stage(1, Xs[1], st1, ema && (n_seen + f) == 0);
            issue_loads(T0, f + 1); LSTAMP_ACC(3); }
            LSTAMP_BARRIER(0);                                           // A: row pass done, staging done
            { LSTAMP_T0(); fixup(T1, Xs[1]); LSTAMP_ACC(4); }
            LSTAMP_BARRIER(1);                                           // B: O written, buffer 1 complete
            { LSTAMP_T0(); copy_out(0, T0, Xs[0], f); LSTAMP_ACC(5); }    // (the compute waves are in buffer 1)
            // ---- buffer 1 (sub-chunk 1 of frame f)
            LSTAMP_BARRIER(2);                                           // C: every loader wave has copied buffer 0 out
            { LSTAMP_T0();
            if (f + 1 < f1)                                             //    (the compute waves pass C half-way through
                stage(0, Xs[0], st0, false);                            //    their row pass; the state must not see a frame twice)
            issue_loads(T1, f + 1); LSTAMP_ACC(3); }
            LSTAMP_BARRIER(0);
            { LSTAMP_T0(); fixup(T0, Xs[0]); LSTAMP_ACC(4); }
            LSTAMP_BARRIER(1);
            { LSTAMP_T0(); copy_out(1, T1, Xs[1], f); LSTAMP_ACC(5); }
            LSTAMP_BARRIER(2);                                           // C: buffer 1 copied out before it is staged again
        }
#if defined(ROWIS_STAMPS) && (ROWIS_STAMPS & 1)
        if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0) {
            unsigned *o = g_rowis_stamps[threadIdx.x >> 6];
            o[0] = stamp_acc0, o[1] = stamp_acc1, o[2] = stamp_acc2, o[3] = stamp_acc3, o[4] = stamp_acc4;
            o[5] = stamp_acc5, o[7] = rowis_now() - stamp_start;
        }
#endif
        if (ema) {
            // into a SECOND buffer: a neighbouring workgroup that starts late reads the halo of its chunk
            // -- samples this workgroup owns -- from `bg`, which must still hold the state before the batch
            auto save_state = [&](const Tab &T, const f4 (&st)[kNS]) {
#pragma unroll
                for (int m = 0; m < kNS; m++) {
                    const int fm = T.FS + 4 * tid + 4 * kTW * m;
                    if (fm >= T.cs && fm < T.ce)
                        *reinterpret_cast<f4 *>(reinterpret_cast<char *>(bg_out) + (unsigned)(4 * fm)) = st[m];
                }
            };
            save_state(T0, st0);
            save_state(T1, st1);
        }
        return;
    }

    // ====================================================================== compute waves
    // window start p <-> output X index p + HALO; thread tid owns p = kP2 tid + j and kHoff + kP2 tid + j: the
    // two halves of a packed operand (v_pk_fma_f32: plain v_fma_f32 has HALF the flop rate), read from LDS
    // with one ds_read2st64_b32.  The taps are scalar operands; the set is symmetric (checked on the host),
    // so RAD + 1 scalar registers do (2 RAD + 1 spill, and every use of a spilled one is a v_readlane).
    // a compute thread's index is rebuilt from the lane id where it is needed: every register is in use, and
    // kept across the row pass it was spilled -- two scratch reloads (a memory round trip each, the second one
    // between barriers A and B with the loader waves waiting) per phase
    const int cwave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x - kTW) >> 6);
    auto compute_tid = [&]() {
        int lane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        return (cwave << 6) | lane;
    };
    auto row_pass = [&](const float *X, f2 (&acc)[kP2]) {
        int off = kP2 * compute_tid();
        asm volatile("" : "+v"(off));
        const float *xb = X + off;
#pragma unroll
        for (int j = 0; j < kP2; j++)
            acc[j] = f2{0.0f, 0.0f};
        constexpr int kAhead = 16;               // inputs in flight (x kP2 / C FMAs each): well above the LDS latency
        f2 xs[kAhead];
#pragma unroll
        for (int i = 0; i < kAhead; i++)
            xs[i] = f2{xb[i], xb[kHoff + i]};
        __builtin_amdgcn_sched_barrier(0);
#ifdef ROWIS_NO_FMA
        constexpr int kInputsRun = 12;
#else
        constexpr int kInputsRun = NIN;
#endif
#pragma clang loop unroll(full)
        for (int i = 0; i < kInputsRun; i++) {
            // barrier C of the loader waves (the buffer they copied out may be staged again): passed here,
            // half-way, so that neither side waits for long
            if (i == kInputsRun * ROWIS_CPOS / 8)
                CSTAMP_BARRIER(2);
            const f2 x = xs[i % kAhead];
            if (i + kAhead < NIN)
                xs[i % kAhead] = f2{xb[i + kAhead], xb[kHoff + i + kAhead]};
            // input i belongs to the outputs j = i (mod C), with tap (i - j) / C
#pragma clang loop unroll(full)
            for (int jj = 0; jj < (kP2 + C - 1) / C; jj++) {
                const int j = i % C + jj * C, kk = (i - j) / C;
                if (j < kP2 && j <= i && kk < NT) {
                    const float t = taps.t[kk <= RAD ? kk : NT - 1 - kk];
                    acc[j] = pk_fma(x, f2{t, t}, acc[j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto write_O = [&](float *O, const f2 (&acc)[kP2]) {
        int off = kP2 * compute_tid();
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int j = 0; j < kP2; j++) {
            O[off + j] = acc[j].x;
            O[kHoff + off + j] = acc[j].y;
        }
    };
    lds_barrier();          // (prologue of the loader waves)
    lds_barrier();
    const int Q = 2 * (f1 - f0);
    for (int q = 0; q < Q; q++) {                    // (one copy of the unrolled row pass for both buffers)
        float *X = Xs[q & 1];
        f2 acc[kP2];
        { CSTAMP_T0(); row_pass(X, acc); CSTAMP_ACC(3); }   // (barrier C of the previous phase's copy-out inside)
        CSTAMP_BARRIER(0);                                              // A
        { CSTAMP_T0(); write_O(X, acc); CSTAMP_ACC(4); }
        CSTAMP_BARRIER(1);                                              // B
    }
    lds_barrier();                                                      // (the C after the last copy-out)
#if defined(ROWIS_STAMPS) && (ROWIS_STAMPS & 2)
    if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0) {
        unsigned *o = g_rowis_stamps[threadIdx.x >> 6];
        o[0] = stamp_acc0, o[1] = stamp_acc1, o[2] = stamp_acc2, o[3] = stamp_acc3, o[4] = stamp_acc4;
        o[7] = rowis_now() - stamp_start;
    }
#endif
}

}  // namespace is

// ------------------------------------------------------------------------------ column pass
constexpr int kColRows = 64, kColCols = 64;              // outputs per step: 8 row groups x 32 column pairs
constexpr int kColStride = kColCols + 2;                 // floats per staged row

__global__ void __launch_bounds__(256, 3)
col_march_f32_kernel(const float *__restrict__ tmp, float *__restrict__ dst, int h, int rw, int ncolt,
                     TapsF32 taps, int nframes)
{
    extern __shared__ float tile[];     // (64 + 2r) rows x 64 samples; row j <-> image row y0 - r + j
    const int r = taps.ksize >> 1;
    const int tid = threadIdx.x;
    // (whole frames per XCD, as in col_sym_f32_kernel)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int ct = q % ncolt;
    const size_t fz = (size_t)(q / ncolt) * 8 + xcd;
    if (fz >= (size_t)nframes)
        return;
    const int x0 = ct * kColCols;
    const float *frame = tmp + fz * (size_t)h * rw;
    float *oframe = dst + fz * (size_t)h * rw;
    const int nrows = kColRows + 2 * r;

    // BORDER_REFLECT_101 of a row index: one reflection covers every row a step can ask for
    // unless the frame is shorter than the tile (then: the general periodic form)
    const int period = h > 1 ? 2 * (h - 1) : 1;
    auto reflect_row = [&](int yin) -> int {
        int y = yin < 0 ? -yin : yin;
        y = y >= h ? period - y : y;
        if (__builtin_expect(y < 0 || y >= h, 0)) {
            int m = yin % period;
            m = m < 0 ? m + period : m;
            y = m < h ? m : period - m;
        }
        return y;
    };
    // 4 samples of image row reflect(y); rw % 4 == 0, and a strip's last columns past the row
    // end read (unused) samples from the row's last float4 instead
    auto load4 = [&](int y, int c4) -> f4 {
        const float *g = frame + (size_t)reflect_row(y) * rw + min(x0 + c4, rw - 4);
        return __builtin_nontemporal_load(reinterpret_cast<const f4 *>(g));
    };
    auto put4 = [&](int row, int c4, f4 v) {
        float *d = tile + row * kColStride + c4;
        *reinterpret_cast<f2 *>(d) = f2{v.x, v.y};
        *reinterpret_cast<f2 *>(d + 2) = f2{v.z, v.w};
    };

    // first tile: rows -r .. 63 + r
    {
        const int items = nrows * (kColCols / 4);
        for (int base = 0; base < items; base += 4 * 256) {
            f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = min(base + u * 256 + tid, items - 1);
                v[u] = load4(-r + (idx >> 4), (idx & 15) * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = base + u * 256 + tid;
                if (idx < items)
                    put4(idx >> 4, (idx & 15) * 4, v[u]);
            }
        }
    }
    __syncthreads();

    const int cp = tid & 31, rg = tid >> 5;
    const int col = x0 + 2 * cp;
    constexpr int RS = kColStride / 2;                                // row stride in f2
    const int tr0_init = 8 * rg + r;                                 // tile row of output j = 0

    for (int y0 = 0; y0 < h; y0 += kColRows) {
        // the tile addresses below do not depend on y0; hidden from the optimiser, which would
        // otherwise keep every (row group, tap) address of the step in a register across steps
        int tr0 = tr0_init;
        asm volatile("" : "+v"(tr0));
        int toff = cp;
        asm volatile("" : "+v"(toff));
        const f2 *tcol = reinterpret_cast<const f2 *>(tile) + toff;
        // the 64 rows the next step adds: image rows y0 + 64 + r .. y0 + 127 + r
        const bool more = y0 + kColRows < h;
        f4 pf[4];
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = u * 256 + tid;                        // 64 rows x 16 float4
                pf[u] = load4(y0 + kColRows + r + (idx >> 4), (idx & 15) * 4);
            }
        }
        // ---- centre tap, then symmetric pairs streaming outwards (two 8-row rings) -----------
        if (col < rw) {
            f2 up[8], dn[8], acc[8];
            {
                const f2 wc = f2{taps.t[r], taps.t[r]};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    up[j] = dn[j] = tcol[(tr0 + j) * RS];
                    acc[j] = pk_fma(up[j], wc, f2{0.0f, 0.0f});
                }
            }
            // Full groups of 8 taps, branch-free (k0 = 1 (mod 8): every ring slot below is static).
            // Tap t of a group replaces ring slot t of `up` (row y+7+k for row y+k-1) and slot 7-t
            // of `dn` (row y-k for row y+8-k); after the eighth tap the whole ring has been
            // replaced.  So the eight arriving rows are kept in a second array, each use picks
            // old or new by slot (static after unrolling), and the two arrays swap roles for the
            // next group: no register moves at all.
            auto group = [&](f2 (&ru)[8], f2 (&nu_)[8], f2 (&rd)[8], f2 (&nd_)[8], int kg) {
                float wk8[8];
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    wk8[t] = taps.t[r + kg + t];
                    nu_[t] = tcol[(tr0 + 7 + kg + t) * RS];
                    nd_[7 - t] = tcol[(tr0 - kg - t) * RS];
                }
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const f2 w2 = f2{wk8[t], wk8[t]};
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int su = (j + 1 + t) % 8, sd = (j + 7 - t) % 8;
                        const f2 U = su <= t ? nu_[su] : ru[su];
                        const f2 D = (7 - sd) <= t ? nd_[sd] : rd[sd];
                        acc[j] = pk_fma(U + D, w2, acc[j]);
                    }
                }
            };
            f2 nu[8], nd[8];
            int k0 = 1;
            for (; k0 + 15 <= r; k0 += 16) {
                group(up, nu, dn, nd, k0);
                group(nu, up, nd, dn, k0 + 8);
            }
            if (k0 + 7 <= r) {                                  // uniform: an odd number of full groups
                group(up, nu, dn, nd, k0);
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    up[t] = nu[t];
                    dn[t] = nd[t];
                }
                k0 += 8;
            }
            // the last r % 8 taps
#pragma unroll
            for (int t = 0; t < 7; t++) {
                if (k0 + t <= r) {                              // uniform
                    up[t] = tcol[(tr0 + 7 + k0 + t) * RS];
                    dn[7 - t] = tcol[(tr0 - k0 - t) * RS];
                    const float wk = taps.t[r + k0 + t];
                    const f2 w2 = f2{wk, wk};
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        acc[j] = pk_fma(up[(j + 1 + t) % 8] + dn[(j + 7 - t) % 8], w2, acc[j]);
                }
            }
            const int yb = y0 + 8 * rg;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (yb + j < h) {
                    float *o = oframe + (size_t)(yb + j) * rw + col;
                    if (col + 1 < rw && (rw % 2 == 0))
                        __builtin_nontemporal_store(acc[j], reinterpret_cast<f2 *>(o));
                    else {
                        o[0] = acc[j].x;
                        if (col + 1 < rw)
                            o[1] = acc[j].y;
                    }
                }
            }
        }
        if (!more)
            break;
        __syncthreads();                                   // every read of this step's tile is done
        // ---- slide: tile rows [64, 64 + 2r) become rows [0, 2r) ---------------------------------
        if (2 * r * (kColCols / 2) <= 9 * 256) {           // uniform; r <= 36: every source row is read
            constexpr int kMv = 9;                         // into registers before the first one is
            f2 mv[kMv];                                    // overwritten: two barriers per step
            const int items = 2 * r * (kColCols / 2);
#pragma unroll
            for (int u = 0; u < kMv; u++) {
                const int idx = u * 256 + tid;
                if (idx < items)
                    mv[u] = reinterpret_cast<const f2 *>(tile)[(kColRows + (idx >> 5)) * RS + (idx & 31)];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kMv; u++) {
                const int idx = u * 256 + tid;
                if (idx < items)
                    reinterpret_cast<f2 *>(tile)[(idx >> 5) * RS + (idx & 31)] = mv[u];
            }
        } else {                                           // larger radii: 64 rows at a time
            for (int c0 = 0; c0 < 2 * r; c0 += kColRows) {
                const int rows = min(kColRows, 2 * r - c0);
                f2 mv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int idx = u * 256 + tid;         // 64 rows x 32 f2
                    if ((idx >> 5) < rows)
                        mv[u] = reinterpret_cast<const f2 *>(tile)[(c0 + kColRows + (idx >> 5)) * RS + (idx & 31)];
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int idx = u * 256 + tid;
                    if ((idx >> 5) < rows)
                        reinterpret_cast<f2 *>(tile)[(c0 + (idx >> 5)) * RS + (idx & 31)] = mv[u];
                }
                if (c0 + kColRows < 2 * r)
                    __syncthreads();
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = u * 256 + tid;
            put4(2 * r + (idx >> 4), (idx & 15) * 4, pf[u]);
        }
        __syncthreads();
    }
}


// ---- column pass, radius known at compile time ----------------------------------------------------
// Same march as col_march_f32_kernel, for the radii integer sigmas give (r = 4 sigma): with the tap
// loop fully unrolled every ring slot is static, so a thread keeps R = 15 or 16 output rows (not 8)
// in three plain register arrays -- rows above, rows below, accumulators -- without role swaps or
// moves: R + 2r LDS reads feed R (2r + 1) packed operations (1 : 13 for sigma 9; the 8-row form
// reads 1 : 7 and kept the LDS pipe of a CU 77 % busy), and a step covers 8 R rows, so the 2r-row
// slide is paid half as often.  R = 15 makes 1080 rows nine whole steps.
template <int RAD, int R>
__global__ void __launch_bounds__(256, 3)
col_sym_f32_kernel(const float *__restrict__ tmp, float *__restrict__ dst, int h, int rw, int ncolt,
                   TapsF32 taps, int nframes)
{
    constexpr int ROWS = 8 * R;                          // output rows per step
    constexpr int NROWS = ROWS + 2 * RAD;                // tile rows; row j <-> image row y0 - RAD + j
    constexpr int RS = kColCols / 2;                     // row stride in f2 (no pad: f2 reads of a
                                                         // half-wave cover one whole row)
    extern __shared__ float tile[];
    const int tid = threadIdx.x;
    // workgroups are dealt round-robin over the 8 XCDs: give every XCD whole frames, so that the 256-byte pieces
    // neighbouring column strips read and write in the same rows meet in ONE L2 (2.87 -> 2.76 ms per 256 x 1080p x 3)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int ct = q % ncolt;
    const size_t fz = (size_t)(q / ncolt) * 8 + xcd;
    if (fz >= (size_t)nframes)
        return;
    const int x0 = ct * kColCols;
    const float *frame = tmp + fz * (size_t)h * rw;
    float *oframe = dst + fz * (size_t)h * rw;

    const int period = h > 1 ? 2 * (h - 1) : 1;
    auto reflect_row = [&](int yin) -> int {             // BORDER_REFLECT_101 (see col_march_f32_kernel)
        int y = yin < 0 ? -yin : yin;
        y = y >= h ? period - y : y;
        if (__builtin_expect(y < 0 || y >= h, 0)) {
            int m = yin % period;
            m = m < 0 ? m + period : m;
            y = m < h ? m : period - m;
        }
        return y;
    };
    auto load4 = [&](int y, int c4) -> f4 {
        const float *g = frame + (size_t)reflect_row(y) * rw + min(x0 + c4, rw - 4);
        return __builtin_nontemporal_load(reinterpret_cast<const f4 *>(g));
    };
    auto put4 = [&](int row, int c4, f4 v) {
        float *d = tile + row * kColCols + c4;
        *reinterpret_cast<f2 *>(d) = f2{v.x, v.y};
        *reinterpret_cast<f2 *>(d + 2) = f2{v.z, v.w};
    };

    {   // first tile: rows -RAD .. ROWS + RAD - 1
        constexpr int items = NROWS * (kColCols / 4);
#pragma unroll 1
        for (int base = 0; base < items; base += 4 * 256) {
            f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = min(base + u * 256 + tid, items - 1);
                v[u] = load4(-RAD + (idx >> 4), (idx & 15) * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = base + u * 256 + tid;
                if (idx < items)
                    put4(idx >> 4, (idx & 15) * 4, v[u]);
            }
        }
    }
    __syncthreads();

    const int cp = tid & 31, rg = tid >> 5;
    const int col = x0 + 2 * cp;
    constexpr int NPF = (ROWS * (kColCols / 4) + 255) / 256;      // float4 per thread and step
    constexpr int NMV = (2 * RAD * RS + 255) / 256;               // f2 per thread in the slide

    for (int y0 = 0; y0 < h; y0 += ROWS) {
        // (the tile addresses do not depend on y0: hidden from the optimiser, which would otherwise
        // keep all of them in registers across steps)
        int toff = R * rg * RS + cp;
        asm volatile("" : "+v"(toff));
        const f2 *tlo = reinterpret_cast<const f2 *>(tile) + toff;  // tile row of image row c0 - RAD (c0: output j = 0)
        int rw_o = rw;                                               // (as above: no per-row output offsets kept in
        asm volatile("" : "+s"(rw_o));                               // registers across steps)
        const bool more = y0 + ROWS < h;
        // the next step's rows, in flight during the arithmetic (ts: the thread id, opaque per step -- else
        // every staging index and address below is computed once and kept in registers for good)
        int ts = tid;
        asm volatile("" : "+v"(ts));
        f4 pf[NPF];
        constexpr int kEarly = NPF / 2;                    // loads issued before the arithmetic (registers
        const bool plain = y0 + 2 * ROWS + RAD <= h;       // for all of them would cost the third workgroup
        __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(   // of a CU), the rest after it
            const_cast<float *>(frame) + (size_t)(y0 + ROWS + RAD) * rw_o, 0, ROWS * rw_o * 4, 0x00027000);
        auto prefetch = [&](int u0, int u1) {
            if (plain) {                                   // uniform: every row of the next step exists
                const int c4 = min(x0 + (ts & 15) * 4, rw - 4);
                const int voff = ((ts >> 4) * rw_o + c4) * 4;
#pragma unroll
                for (int u = u0; u < u1; u++) {            // (the last float4 of R = 15 steps: dropped below)
                    const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(irsrc, voff, u * 16 * rw_o * 4, 2 /* nt */);
                    pf[u] = __builtin_bit_cast(f4, raw);
                }
            } else {
#pragma unroll
                for (int u = u0; u < u1; u++) {
                    const int idx = min(u * 256 + ts, ROWS * (kColCols / 4) - 1);
                    pf[u] = load4(y0 + ROWS + RAD + (idx >> 4), (idx & 15) * 4);
                }
            }
        };
        if (more)
            prefetch(0, kEarly);
        if (col < rw) {
            // centre tap, then symmetric pairs outwards (the oracle's order).  Ring slots: the row
            // c0 + m (m may be negative) lives in slot m mod R of `up` resp. `dn`.
            f2 up[R], dn[R], acc[R];
            {
                const f2 wc = f2{taps.t[RAD], taps.t[RAD]};
#pragma unroll
                for (int j = 0; j < R; j++) {
                    up[j] = dn[j] = tlo[(RAD + j) * RS];
                    acc[j] = pk_fma(up[j], wc, f2{0.0f, 0.0f});
                }
            }
            // the two rows a tap brings in are read kAheadTaps taps early; the scheduling fence after
            // every tap keeps the compiler from hoisting ALL reads of the unrolled chain to its top
            // (which spilled 58 registers)
            constexpr int kAheadTaps = 3;
            f2 su[kAheadTaps], sd[kAheadTaps];
#pragma unroll
            for (int k = 1; k <= kAheadTaps && k <= RAD; k++) {
                su[k % kAheadTaps] = tlo[(RAD + R - 1 + k) * RS];
                sd[k % kAheadTaps] = tlo[(RAD - k) * RS];
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef COLSYM_NO_FMA
            constexpr int kTapsRun = 2;
#else
            constexpr int kTapsRun = RAD;
#endif
#pragma unroll
            for (int k = 1; k <= kTapsRun; k++) {
                up[(R - 1 + k) % R] = su[k % kAheadTaps];                      // row c0 + R - 1 + k
                dn[((-k) % R + R) % R] = sd[k % kAheadTaps];                   // row c0 - k
                if (k + kAheadTaps <= RAD) {
                    su[k % kAheadTaps] = tlo[(RAD + R - 1 + k + kAheadTaps) * RS];
                    sd[k % kAheadTaps] = tlo[(RAD - k - kAheadTaps) * RS];
                }
                const f2 w2 = f2{taps.t[RAD + k], taps.t[RAD + k]};
                // four pair sums, then their four FMAs: a packed result consumed by the very next
                // instruction costs a wait state (the compiler paired every add with its FMA and put an
                // s_nop between them)
#pragma unroll
                for (int j0 = 0; j0 < R; j0 += 4) {
                    f2 sum[4];
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (j0 + u < R)
                            sum[u] = up[(j0 + u + k) % R] + dn[((j0 + u - k) % R + R) % R];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (j0 + u < R)
                            acc[j0 + u] = pk_fma(sum[u], w2, acc[j0 + u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // (pinned: the optimiser otherwise sinks each accumulator's whole chain into its guarded
            // store below -- one output at a time, all R + 2 RAD rows live in registers)
#pragma unroll
            for (int j = 0; j < R; j++)
                asm volatile("" : "+v"(acc[j]));
            // branch-free buffer stores: the descriptor covers this step's rows that exist, rows past the
            // frame fall outside it and are dropped (rw is a multiple of 4: a column pair is whole)
            const int rows_here = min(ROWS, h - y0);
            __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                oframe + (size_t)y0 * rw_o, 0, rows_here * rw_o * 4, 0x00027000);
            const int voff = (R * rg * rw_o + col) * 4;
#pragma unroll
            for (int j = 0; j < R; j++)
#ifdef COLSYM_NO_STORE
                if (acc[j].x == 123.456f)
#endif
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, acc[j]), orsrc, voff, j * rw_o * 4,
                                                      2 /* nt */);
        }
        if (!more)
            break;
        asm volatile("" : "+v"(ts));
        prefetch(kEarly, NPF);
        __syncthreads();                                   // every read of this step's tile is done
        // ---- slide: tile rows [ROWS, ROWS + 2 RAD) become rows [0, 2 RAD); every source row is in a
        // register before anything is written (the new rows below land on source rows as well)
        {
            f2 mv[NMV];
            constexpr int items = 2 * RAD * RS;
#pragma unroll
            for (int u = 0; u < NMV; u++) {
                const int idx = u * 256 + ts;
                if (idx < items)
                    mv[u] = reinterpret_cast<const f2 *>(tile)[ROWS * RS + idx];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NMV; u++) {
                const int idx = u * 256 + ts;
                if (idx < items)
                    reinterpret_cast<f2 *>(tile)[idx] = mv[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NPF; u++) {
            const int idx = u * 256 + ts;
            if (idx < ROWS * (kColCols / 4))
                put4(2 * RAD + (idx >> 4), (idx & 15) * 4, pf[u]);
        }
        __syncthreads();
    }
}

template <int R>
static bool launch_col_sym(int r, const float *tmp, float *dst, int n, int h, int rw, const TapsF32 &taps,
                           hipStream_t st)
{
    const int ncolt = cdiv(rw, kColCols);
    const unsigned grid = (unsigned)((size_t)ncolt * 8 * cdiv(n, 8));      // (frames in groups of 8: one per XCD)
#define VA_COL_SYM(RAD)                                                                            \
    case RAD:                                                                                      \
        col_sym_f32_kernel<RAD, R><<<grid, 256, (size_t)(8 * R + 2 * RAD) * kColCols * sizeof(float), st>>>( \
            tmp, dst, h, rw, ncolt, taps, n);                                                      \
        return true;
    switch (r) {
        VA_COL_SYM(4) VA_COL_SYM(8) VA_COL_SYM(12) VA_COL_SYM(16) VA_COL_SYM(20) VA_COL_SYM(24)
        VA_COL_SYM(28) VA_COL_SYM(32) VA_COL_SYM(36)
    default:
        return false;
    }
#undef VA_COL_SYM
}

// chunk length for the row kernel: every CU the same work when the frame allows it
bool plan_rows(int h, int w, int c, const TapsF32 &taps, int slots, RowPlan *plan)
{
    const int r = taps.ksize / 2, halo = r * c, rw = w * c;
    if ((c != 1 && c != 3) || w <= r || rw % 4 != 0 || kP % c != 0)
        return false;
    const long long total = (long long)h * rw;
    long long L = ((total + slots - 1) / slots + 3) / 4 * 4;
    auto fits = [&](long long len) {
        const int ns = max_segments(len, rw);
        return ns <= kMaxSeg && len + (long long)ns * 2 * halo <= 2LL * kHalfCap &&
               (len + 2 * halo + 8) / 4 <= (long long)kNV * kT;
    };
    if (L < 1024)
        L = total < 1024 ? (total + 3) / 4 * 4 : 1024;
    while (L > 4 && !fits(L))
        L -= 4;
    if (!fits(L) || L < 4 * c)
        return false;
    plan->L = (int)L;
    plan->nchunks = (int)((total + L - 1) / L);
    return true;
}

static int g_f32_col_generic = 0;   // test hook: the runtime-radius column kernel for every radius
static int g_f32_row_generic = 0;   // test hook: the runtime-radius row kernel for every radius

// the compile-time-radius row kernel: sub-chunk length (0: shape not supported)
static int plan_rows_is(int h, int w, int c, int r, int cus, const TapsF32 &taps, int *p2_out)
{
    *p2_out = is::kP2Max;
    const int halo = r * c, rw = w * c;
    if ((c != 1 && c != 3) || w <= r || rw % 4 != 0)
        return 0;
    if (r < 4 || r > 36 || r % 4 != 0)                 // integer sigmas 1 ... 9 (r = 4 sigma)
        return 0;
    for (int k = 0; k < r; k++)                        // the kernel keeps one half of the (symmetric) tap set
        if (taps.t[k] != taps.t[2 * r - k])
            return 0;
    const long long total = (long long)h * rw;
    long long L = ((total + 2 * cus - 1) / (2 * cus) + 3) / 4 * 4;
    if (L < 1024)
        L = total < 1024 ? (total + 3) / 4 * 4 : 1024;
    if (is::fits(L, rw, halo, 13)) {                   // the equal split fits the narrower layout
        *p2_out = 13;
        return (int)L;
    }
    while (L > 4 && !is::fits(L, rw, halo, is::kP2Max))
        L -= 4;
    if (!is::fits(L, rw, halo, is::kP2Max) || L < 4 * c)
        return 0;
    return (int)L;
}

template <int C, int P2>
static bool launch_row_is(int r, unsigned nwg, unsigned ny, const float *src, float *tmp, const float *bg,
                          float *bg_out, long long n_seen, float rate, int n, int h, int w, int L,
                          const TapsF32 &taps, hipStream_t st)
{
    const dim3 grid(nwg, ny);
#define VA_ROW_IS(RAD)                                                                                        \
    case RAD:                                                                                                 \
        if (bg)                                                                                               \
            is::row_is_f32_kernel<C, RAD, P2, true><<<grid, is::kTW + is::kTC, 0, st>>>(src, tmp, bg, bg_out, n_seen, rate, n, h, w, L, taps); \
        else                                                                                                  \
            is::row_is_f32_kernel<C, RAD, P2, false><<<grid, is::kTW + is::kTC, 0, st>>>(src, tmp, bg, bg_out, n_seen, rate, n, h, w, L, taps); \
        return true;
    switch (r) {
        VA_ROW_IS(4) VA_ROW_IS(8) VA_ROW_IS(12) VA_ROW_IS(16) VA_ROW_IS(20) VA_ROW_IS(24) VA_ROW_IS(28) VA_ROW_IS(32)
        VA_ROW_IS(36)
    default:
        return false;
    }
#undef VA_ROW_IS
}

}  // namespace

void gauss_f32_test_hook(int generic)
{
    g_f32_col_generic = generic & 1;
    g_f32_row_generic = (generic >> 1) & 1;
}

bool gauss_f32_fused_supported(int h, int w, int c, const TapsF32 &taps)
{
    RowPlan p;
    const int r = taps.ksize / 2;
    if (r > 60)          // the marching column tile must fit 64 KB of LDS four times per CU
        return false;
    if ((long long)h * w * c >= (1ll << 30))    // byte offsets inside a frame travel as 32-bit values (4 * h*w*c)
        return false;
    return plan_rows(h, w, c, taps, 1024, &p);
}

// bg == nullptr: plain blur of the frames; otherwise EMA background subtraction first: the state before the
// batch is read from bg, the state after it is written to bg_out (a different buffer: see save_state)
int launch_gauss_f32_fused(const float *src, float *dst, float *scratch, const float *bg, float *bg_out,
                           int64_t n_seen, double rate, int n, int h, int w, int c, const TapsF32 &taps,
                           hipStream_t st, StageProfiler *prof)
{
    VA_REQUIRE(!bg || (bg_out && bg_out != bg), "fused float gaussian: the state needs a second buffer");
    RowPlan plan;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    VA_REQUIRE(plan_rows(h, w, c, taps, 4 * cus, &plan), "fused float gaussian: unsupported shape");
    VA_REQUIRE(reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(scratch) % 16 == 0 &&
                   (!bg || (reinterpret_cast<uintptr_t>(bg) % 16 == 0 && reinterpret_cast<uintptr_t>(bg_out) % 16 == 0)),
               "fused float gaussian: buffers must be 16-byte aligned");
    if (n == 0 || h == 0)
        return VA_OK;
    const int r = taps.ksize / 2, rw = w * c;
    const size_t lds1 = 2 * (size_t)(kHalfCap + (taps.ksize - 1) * c + kP + 8) * sizeof(f2) +
                        2 * (size_t)kNV * kT * sizeof(unsigned short);
    const int nwg = (plan.nchunks + 1) / 2;          // two sub-chunks per workgroup
    // radii of common integer sigmas: the input-stationary kernel (one workgroup per CU)
    int p2 = is::kP2Max;
    const int Lis = g_f32_row_generic ? 0 : plan_rows_is(h, w, c, r, cus, taps, &p2);
    bool rows_done = false;
    if (Lis > 0) {
        const long long total = (long long)h * rw;
        const unsigned nwg2 = (unsigned)(((total + Lis - 1) / Lis + 1) / 2), ny = bg ? 1u : (unsigned)n;
#define VA_ROW_IS_ARGS r, nwg2, ny, src, scratch, bg, bg_out, n_seen, (float)rate, n, h, w, Lis, taps, st
        if (p2 == 13)
            rows_done = c == 1 ? launch_row_is<1, 13>(VA_ROW_IS_ARGS) : launch_row_is<3, 13>(VA_ROW_IS_ARGS);
        else
            rows_done = c == 1 ? launch_row_is<1, 15>(VA_ROW_IS_ARGS) : launch_row_is<3, 15>(VA_ROW_IS_ARGS);
#undef VA_ROW_IS_ARGS
    }
    if (rows_done) {
    } else if (bg) {
        if (c == 1)
            ema_row_f32_kernel<1, true><<<nwg, 2 * kT, lds1, st>>>(src, scratch, bg, bg_out, n_seen, (float)rate, n, h, w, plan.L, taps);
        else
            ema_row_f32_kernel<3, true><<<nwg, 2 * kT, lds1, st>>>(src, scratch, bg, bg_out, n_seen, (float)rate, n, h, w, plan.L, taps);
    } else {
        const dim3 grid((unsigned)nwg, (unsigned)n);
        if (c == 1)
            ema_row_f32_kernel<1, false><<<grid, 2 * kT, lds1, st>>>(src, scratch, nullptr, nullptr, 0, 0.f, n, h, w, plan.L, taps);
        else
            ema_row_f32_kernel<3, false><<<grid, 2 * kT, lds1, st>>>(src, scratch, nullptr, nullptr, 0, 0.f, n, h, w, plan.L, taps);
    }
    VA_LAUNCH_CHECK("ema_row_f32_kernel");
    if (prof)
        prof->mark(bg ? "ema_row_f32" : "row_f32", st);
    // radii of integer sigmas: the unrolled 15/16-rows-per-thread kernel (whichever wastes fewer rows
    // in the last step); any other radius: the 8-row kernel with its runtime tap loop
    const int waste16 = cdiv(h, 128) * 128 - h, waste15 = cdiv(h, 120) * 120 - h;
    const bool unrolled = !g_f32_col_generic &&
                          (waste15 < waste16 ? launch_col_sym<15>(r, scratch, dst, n, h, rw, taps, st)
                                             : launch_col_sym<16>(r, scratch, dst, n, h, rw, taps, st));
    if (!unrolled) {
        const int ncolt = cdiv(rw, kColCols);
        const size_t lds2 = (size_t)(kColRows + 2 * r) * kColStride * sizeof(float);
        col_march_f32_kernel<<<(unsigned)((size_t)ncolt * 8 * cdiv(n, 8)), 256, lds2, st>>>(scratch, dst, h, rw, ncolt, taps, n);
    }
    VA_LAUNCH_CHECK("col_f32 kernel");
    if (prof)
        prof->mark("col_f32", st);
    return VA_OK;
}

#ifdef ROWIS_STAMPS
extern "C" int va_debug_rowis_stamps(unsigned int *out32)
{
    return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(is::g_rowis_stamps), 128 * sizeof(unsigned int));
}
#endif

}  // namespace va
