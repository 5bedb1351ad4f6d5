// Fused dilated depthwise convolution kernels (forward and backward), gfx950.
//
// A block owns CG consecutive channels of one batch element and one time tile [t0, t0 + TT).  For the
// common 4 s / 8 kHz workload (T = 3999 frames) a tile is the whole row, so no halo is ever re-read.
// Per channel the block stages the transformed input row (tile + dilation halo) in LDS, synchronises,
// and every thread then produces 4 consecutive output frames from LDS taps.  Input-side PReLU + layer
// norm are applied while staging (zero padding is applied AFTER the norm, as F.pad does in
// src/models/tdcn.py:132), statistics of the output-side PReLU are reduced on the way out, so each
// activation crosses HBM exactly once in each direction.
#include <algorithm>

#include "ctn_common.h"

// Activations stream through these kernels once: non-temporal loads and stores (4 % on the forward, 2 % on the backward
// kernel in the training step).  -DCTN_DBG_DW_TEMPORAL restores plain accesses for comparison.
#if defined(CTN_DBG_DW_TEMPORAL)
#define CTN_DW_LD ctn_ld4
#define CTN_DW_ST ctn_st4
#elif defined(CTN_DBG_DW_TEMPORAL_ST)
#define CTN_DW_LD ctn_ld4_nt
#define CTN_DW_ST ctn_st4
#elif defined(CTN_DBG_DW_TEMPORAL_LD)
#define CTN_DW_LD ctn_ld4
#define CTN_DW_ST ctn_st4_nt
#else
#define CTN_DW_LD ctn_ld4_nt
#define CTN_DW_ST ctn_st4_nt
#endif

namespace {

constexpr int NTH = 256;
constexpr int MAXP = 8;

__device__ __forceinline__ int floor4(int v) { return v & ~3; }

// 4 consecutive floats starting at row[off] (off >= 0, row 16-byte aligned): one ds_read_b128 when off % 4 == 0
// (every layer with dilation % 4 == 0), scalar reads otherwise.  off % 4 is the same in every lane.
__device__ __forceinline__ f32x4 lds_read4(const float* row, int off) {
    f32x4 r;
    if ((off & 3) == 0) {
        r = ctn_ld4(row + off);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = row[off + e];
    }
    return r;
}

// Barrier that orders LDS accesses only.  __syncthreads() also waits for every outstanding global load, which ends a
// prefetch issued in front of it.
#define DW_LDS_BARRIER()                                    \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                       \
        asm volatile("" ::: "memory");                      \
    } while (0)

// The same for the whole-row kernels, which know off % 4 from their launch arguments alone (`rem`, a scalar): when it is
// not zero, the two aligned vectors around the four floats are read and the pick is a scalar branch.  Reads up to
// row[(off & ~3) + 7]: those kernels' rows carry 8 floats of slack behind the last tap.
__device__ __forceinline__ f32x4 lds_read4s(const float* row, int off, int rem) {
    const float* p = row + (off - rem);
    const f32x4 lo = ctn_ld4(p);
    if (rem == 0) return lo;
    const f32x4 hi = ctn_ld4(p + 4);
    if (rem == 1) return f32x4{lo[1], lo[2], lo[3], hi[0]};
    if (rem == 2) return f32x4{lo[2], lo[3], hi[0], hi[1]};
    return f32x4{lo[3], hi[0], hi[1], hi[2]};
}

// Per-frame statistics of 4 consecutive frames; gLN (st == 0) was hoisted to scalars by the caller.
__device__ __forceinline__ void stat4(const float* p, int64_t sb, int64_t st, int b, int t, int T, float hoisted,
                                      float (&o)[4]) {
    if (st == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = hoisted;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = p[(int64_t)b * sb + (int64_t)min(t + e, T - 1) * st];
    }
}

constexpr int NIT = 5;   // float4s staged per thread and round: 5 * 256 * 4 = 5120 frames covers T' = 3999 + halo

// ------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(NTH) void dw_fwd_kernel(const ctn_dw_fwd_args a, const int CG, const int TT,
                                                     const int lw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[2 * (NTH / 64)];
    const int b = blockIdx.z, c0 = blockIdx.y * CG, t0 = blockIdx.x * TT;
    const int T = a.T, P = a.P, d = a.dilation, padL = a.pad_left;
    const int tA = floor4(t0 - padL);
    const int tid = threadIdx.x;
    const float ain = a.alpha_in ? a.alpha_in[0] : 1.f;
    const float aout = a.alpha_out ? a.alpha_out[0] : 1.f;
    const bool do_stats = a.alpha_out != nullptr || a.esums != nullptr || a.ecol != nullptr;
    const bool norm = (a.xflags & CTN_XF_NORM) != 0;
    const bool prelu_in = (a.xflags & CTN_XF_PRELU) != 0;
    const int cend = min(c0 + CG, a.C);
    const int t_end = min(t0 + TT, T);
    const int lw4 = lw / 4;
    float mean0 = 0.f, rstd0 = 1.f;
    if (norm && a.sst == 0) {
        mean0 = a.mean[(int64_t)b * a.ssb];
        rstd0 = a.rstd[(int64_t)b * a.ssb];
    }

    float s1 = 0.f, s2 = 0.f;                 // gLN partial sums
    float cs1[4] = {0, 0, 0, 0}, cs2[4] = {0, 0, 0, 0};   // cLN per-frame sums (one float4 per thread)

    for (int c = c0; c < cend; ++c) {
        // ---- stage the transformed, zero-padded input row: all loads of a round are issued before any is used
        const float* xrow = a.x + (int64_t)b * a.xbs + (int64_t)c * a.ldx;
        const float g = norm ? a.gamma[c] : 1.f;
        const float be = norm ? a.beta[c] : 0.f;
        for (int r0 = 0; r0 < lw4; r0 += NIT * NTH) {
            f32x4 v[NIT];
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int i4 = r0 + tid + j * NTH;
                const int t = tA + 4 * i4;
                v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i4 < lw4 && t >= 0 && t < T) v[j] = CTN_DW_LD(xrow + t);
            }
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int i4 = r0 + tid + j * NTH;
                const int t = tA + 4 * i4;
                if (i4 >= lw4) continue;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (t >= 0 && t < T) {
                    float mu[4], rs[4];
                    if (norm) {
                        stat4(a.mean, a.ssb, a.sst, b, t, T, mean0, mu);
                        stat4(a.rstd, a.ssb, a.sst, b, t, T, rstd0, rs);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float u = v[j][e];
                        if (prelu_in) u = ctn_prelu(u, ain);
                        if (norm) u = (u - mu[e]) * rs[e] * g + be;
                        o[e] = (t + e < T) ? u : 0.f;
                    }
                }
                ctn_st4(&lds[4 * i4], o);
            }
        }
        __syncthreads();
        // ---- taps
        float w[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) w[k] = (k < P) ? a.w[(int64_t)c * P + k] : 0.f;
        const float bias = a.bias ? a.bias[c] : 0.f;
        float* yrow = a.y + (int64_t)b * a.ybs + (int64_t)c * a.ldy;
        for (int t = t0 + 4 * tid; t < t_end; t += 4 * NTH) {
            const int base = t - padL - tA;
            f32x4 o = {bias, bias, bias, bias};
#pragma unroll
            for (int k = 0; k < MAXP; ++k) {
                if (k < P) {
                    const f32x4 x4 = lds_read4(lds, base + k * d);
                    o += w[k] * x4;
                }
            }
            CTN_DW_ST(yrow + t, o);
            if (do_stats) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = (t + e < T) ? ctn_prelu(o[e], aout) : 0.f;
                    if (a.cln) {
                        cs1[e] += u;
                        cs2[e] += u * u;
                    } else {
                        s1 += u;
                        s2 += u * u;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!do_stats) return;
    if (a.cln) {
        const int t = t0 + 4 * tid;   // host guarantees TT <= 4 * NTH in cLN mode
        if (t < t_end) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t + e < T) {
                    atomicAdd(a.ecol + ((int64_t)b * 2 + 0) * a.ecol_ld + t + e, cs1[e]);
                    atomicAdd(a.ecol + ((int64_t)b * 2 + 1) * a.ecol_ld + t + e, cs2[e]);
                }
        }
    } else {
        float v[2] = {s1, s2};
        const float tot = ctn_block_sum<2>(v, red);
        if (tid < 2) atomicAdd(a.esums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + tid, (double)tot);
        __shared__ unsigned fin_flag;
        ctn_finish_sums(a.fin, a.esums + (int64_t)b * CTN_STAT_SLOTS * 2, b, gridDim.x * gridDim.y, &fin_flag);
    }
}

// ------------------------------------------------------------------------------------------ forward, whole rows
// The separator's hot case as straight-line code: per-sample statistics (gLN) or no norm at all, one block per
// (batch, channel) row of up to 4 * FIT * NTH frames.  Every global load of the row is issued at the top of the
// kernel from clamped addresses (validity by selects, no branch around a load), so a block has its whole 16 KB row in
// flight at once and eight resident blocks per CU keep the memory pipe full; the kernel above stages a row in rounds
// behind guarded loads, which the compiler turns into branches with full vmcnt waits.  The arithmetic is the same,
// expression for expression, so both kernels produce identical outputs.
constexpr int FIT = 4;                 // float4 per thread: rows of up to 4 * 256 * 4 = 4096 frames
template <int P>
__global__ __launch_bounds__(NTH) void dw_fwd_row_kernel(const ctn_dw_fwd_args a, const int lw, const int HL4) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[2 * (NTH / 64)];
    const int b = blockIdx.y, c = blockIdx.x;
    const int T = a.T, d = a.dilation, padL = a.pad_left;
    const int tid = threadIdx.x;
    const float* xrow = a.x + (int64_t)b * a.xbs + (int64_t)c * a.ldx;
    f32x4 v[FIT];
#pragma unroll
    for (int j = 0; j < FIT; ++j) {
        const int tq = min(4 * (tid + j * NTH), (int)a.ldx - 4);
        v[j] = CTN_DW_LD(xrow + tq);
    }
    const bool norm = (a.xflags & CTN_XF_NORM) != 0;
    const bool prelu_in = (a.xflags & CTN_XF_PRELU) != 0;
    const bool do_stats = a.esums != nullptr;
    const float ain = (prelu_in && a.alpha_in) ? a.alpha_in[0] : 1.f;
    const float aout = a.alpha_out ? a.alpha_out[0] : 1.f;
    const float mean0 = norm ? a.mean[(int64_t)b * a.ssb] : 0.f;
    const float rstd0 = norm ? a.rstd[(int64_t)b * a.ssb] : 1.f;
    const float g = norm ? a.gamma[c] : 1.f;
    const float be = norm ? a.beta[c] : 0.f;
    float w[P];
#pragma unroll
    for (int k = 0; k < P; ++k) w[k] = a.w[(int64_t)c * P + k];
    const float bias = a.bias ? a.bias[c] : 0.f;
    // zero padding on both sides of the row slots [HL4, HL4 + 4 * FIT * NTH)
    for (int i = tid; i < HL4 / 4; i += NTH) ctn_st4(&lds[4 * i], f32x4{0.f, 0.f, 0.f, 0.f});
    for (int i = HL4 / 4 + FIT * NTH + tid; i < lw / 4; i += NTH) ctn_st4(&lds[4 * i], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int j = 0; j < FIT; ++j) {
        const int t = 4 * (tid + j * NTH);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = v[j][e];
            if (prelu_in) u = ctn_prelu(u, ain);
            if (norm) u = __builtin_fmaf((u - mean0) * rstd0, g, be);   // dw_bwd_gln_rc_kernel recomputes exactly this
            o[e] = (t + e < T) ? u : 0.f;
        }
        ctn_st4(&lds[HL4 + t], o);
    }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    float* yrow = a.y + (int64_t)b * a.ybs + (int64_t)c * a.ldy;
    int rem[P];                          // tap offsets modulo 4: HL4 and t are multiples of 4
#pragma unroll
    for (int k = 0; k < P; ++k) rem[k] = (k * d - padL) & 3;
#pragma unroll
    for (int j = 0; j < FIT; ++j) {
        const int t = 4 * (tid + j * NTH);
        if (t < T) {
            const int base = HL4 + t - padL;
            f32x4 o = {bias, bias, bias, bias};
#pragma unroll
            for (int k = 0; k < P; ++k) o += w[k] * lds_read4s(lds, base + k * d, rem[k]);
            CTN_DW_ST(yrow + t, o);
            if (do_stats) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = (t + e < T) ? ctn_prelu(o[e], aout) : 0.f;
                    s1 += u;
                    s2 += u * u;
                }
            }
        }
    }
    if (!do_stats) return;
    float sv[2] = {s1, s2};
    const float tot = ctn_block_sum<2>(sv, red);
    if (tid < 2) atomicAdd(a.esums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + tid, (double)tot);
    __shared__ unsigned fin_flag;
    ctn_finish_sums(a.fin, a.esums + (int64_t)b * CTN_STAT_SLOTS * 2, b, gridDim.x, &fin_flag);
}

// Per-frame sums of a block's time tile -> global rows col[0][t], col[1][t].  A thread holds four consecutive frames;
// adding them from registers makes every atomic instruction of a wave touch 64 addresses 16 bytes apart (sixteen
// 64-byte lines), and the memory side executes atomics per line.  Through LDS, consecutive lanes add consecutive
// frames: four lines per instruction.  All threads of the block must call; LDS (>= 8 * NTH floats) is overwritten.
__device__ __forceinline__ void col_atomics(float* lds, const float (&cs1)[4], const float (&cs2)[4], float* col,
                                            const int64_t col_ld, const int t0, const int t_end) {
    const int tid = threadIdx.x;
    __syncthreads();
    ctn_st4(&lds[4 * tid], f32x4{cs1[0], cs1[1], cs1[2], cs1[3]});
    ctn_st4(&lds[4 * NTH + 4 * tid], f32x4{cs2[0], cs2[1], cs2[2], cs2[3]});
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + q * NTH;
        if (t0 + i < t_end) {
            atomicAdd(col + t0 + i, lds[i]);
            atomicAdd(col + col_ld + t0 + i, lds[4 * NTH + i]);
        }
    }
}

// ------------------------------------------------------------------------------------------ forward, cLN
// Cumulative norm: the statistics of the output are per-frame sums over ALL channels, so a block walks CG channels of
// one time tile (TT <= 4 * NTH frames: one float4 of output per thread and channel) and keeps the per-frame sums in
// registers, one atomic pair per frame and block at the end.  Unlike the kernel above, which stages one row per
// barrier pair, CS rows are staged at once: CS times the loads in flight and a CS-th of the barriers -- per-frame
// statistics make a row tile short (1024 frames + halo), and one such row per phase leaves the memory pipe idle.
template <int CS>
__global__ __launch_bounds__(NTH) void dw_fwd_cln_kernel(const ctn_dw_fwd_args a, const int CG, const int TT,
                                                         const int lw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];     // CS rows of lw floats
    const int b = blockIdx.z, c0 = blockIdx.y * CG, t0 = blockIdx.x * TT;
    const int T = a.T, P = a.P, d = a.dilation, padL = a.pad_left;
    const int tA = floor4(t0 - padL);
    const int tid = threadIdx.x;
    const float ain = a.alpha_in ? a.alpha_in[0] : 1.f;
    const float aout = a.alpha_out ? a.alpha_out[0] : 1.f;
    const bool do_stats = a.ecol != nullptr;
    const bool norm = (a.xflags & CTN_XF_NORM) != 0;
    const bool prelu_in = (a.xflags & CTN_XF_PRELU) != 0;
    const int cend = min(c0 + CG, a.C);
    const int t_end = min(t0 + TT, T);
    const int lw4 = lw / 4;
    const int items = CS * lw4;                       // float4s to stage per phase
    constexpr int NITC = 6;                           // CS * (4 * NTH + halo + 8) / 4 <= NITC * NTH for halos <= 500
    const int tl4 = ((T + 3) & ~3) - 4;               // last whole float4 of a row
    const float* mrow = norm ? a.mean + (int64_t)b * a.ssb : nullptr;
    const float* rrow = norm ? a.rstd + (int64_t)b * a.ssb : nullptr;

    // The block's per-channel constants in LDS: a staged float4 belongs to a channel that differs from thread to thread,
    // and gamma / beta fetched from global memory where they are used are a dependent round trip per staged float4
    // -- six per phase, which is what a phase then takes.
    constexpr int MAXCG = 16;                         // the host launches this kernel with CG <= MAXCG
    __shared__ float pgam[MAXCG], pbet[MAXCG], pbias[MAXCG], pw[MAXCG * MAXP];
    if (tid < MAXCG) {
        const int c = min(c0 + tid, a.C - 1);
        pgam[tid] = norm ? a.gamma[c] : 1.f;
        pbet[tid] = norm ? a.beta[c] : 0.f;
        pbias[tid] = a.bias ? a.bias[c] : 0.f;
    }
    if (tid < MAXCG * MAXP) {
        const int c = c0 + tid / MAXP, k = tid % MAXP;
        pw[tid] = (c < a.C && k < P) ? a.w[(int64_t)c * P + k] : 0.f;
    }
    __syncthreads();
    float cs1[4] = {0, 0, 0, 0}, cs2[4] = {0, 0, 0, 0};
    const int t = t0 + 4 * tid;                       // this thread's output frames
    // which staged float4s are this thread's does not depend on the channel group, and neither do the per-frame
    // statistics (they belong to the frames): fetched once, as whole float4s of the aligned statistics rows
    int rr[NITC], ii[NITC];
    f32x4 mu4[NITC], rs4[NITC];
#pragma unroll
    for (int j = 0; j < NITC; ++j) {
        const int idx = tid + j * NTH;
        const int r = idx / lw4, i4 = idx - r * lw4;
        rr[j] = r, ii[j] = i4;
        const int tt = min(max(tA + 4 * i4, 0), tl4);
        mu4[j] = norm ? ctn_ld4(mrow + tt) : f32x4{0.f, 0.f, 0.f, 0.f};
        rs4[j] = norm ? ctn_ld4(rrow + tt) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    // the rows of a phase are requested as soon as the phase before has staged its own: they arrive under its taps
    f32x4 v[NITC];
    auto load_phase = [&](const int cg) {
#pragma unroll
        for (int j = 0; j < NITC; ++j) {
            const int c = min(cg + rr[j], cend - 1);
            const int tt = min(max(tA + 4 * ii[j], 0), tl4);
            v[j] = CTN_DW_LD(a.x + (int64_t)b * a.xbs + (int64_t)c * a.ldx + tt);
        }
    };
    if (c0 < cend) load_phase(c0);
    for (int cg = c0; cg < cend; cg += CS) {
        // ---- stage CS transformed, zero-padded rows
#pragma unroll
        for (int j = 0; j < NITC; ++j) {
            const int idx = tid + j * NTH;
            if (idx >= items) continue;
            const int r = rr[j], i4 = ii[j];
            const int c = cg + r;
            const int tt = tA + 4 * i4;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (c < cend && tt >= 0 && tt < T) {
                const float g = pgam[c - c0], be = pbet[c - c0];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float u = v[j][e];
                    if (prelu_in) u = ctn_prelu(u, ain);
                    if (norm) u = (u - mu4[j][e]) * rs4[j][e] * g + be;
                    o[e] = (tt + e < T) ? u : 0.f;
                }
            }
            ctn_st4(&lds[r * lw + 4 * i4], o);
        }
        if (cg + CS < cend) load_phase(cg + CS);
        DW_LDS_BARRIER();
        // ---- taps: one float4 of output per thread and channel
        if (t < t_end) {
            const int base = t - padL - tA;
#pragma unroll
            for (int r = 0; r < CS; ++r) {
                const int c = cg + r;
                if (c >= cend) break;
                const float bias = pbias[c - c0];
                f32x4 o = {bias, bias, bias, bias};
#pragma unroll
                for (int k = 0; k < MAXP; ++k) {
                    if (k < P) {
                        const float wk = pw[(c - c0) * MAXP + k];
                        o += wk * lds_read4(lds + r * lw, base + k * d);
                    }
                }
                CTN_DW_ST(a.y + (int64_t)b * a.ybs + (int64_t)c * a.ldy + t, o);
                if (do_stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = (t + e < T) ? ctn_prelu(o[e], aout) : 0.f;
                        cs1[e] += u;
                        cs2[e] += u * u;
                    }
                }
            }
        }
        DW_LDS_BARRIER();
    }
    if (do_stats) col_atomics(lds, cs1, cs2, a.ecol + (int64_t)b * 2 * a.ecol_ld, a.ecol_ld, t0, t_end);
}

// ------------------------------------------------------------------------------------------ backward
// LDS rows (same indexing, origin tA, halo H on both sides):
//   D [i] = dy2[tA + i]   (gradient w.r.t. the conv output, after the output-side norm/PReLU backward)
//   Xh[i] = xhat1[tA + i] (normalised input BEFORE gamma/beta, zero outside [0, T))
// The conv input is n1 = gamma1 * xhat1 + beta1 inside [0, T) and 0 outside, hence
//   dw[k] = gamma1 * sum D[t] * Xh[t + k d - padL] + beta1 * sum D[t] * [0 <= t + k d - padL < T].
template <int P>
__global__ __launch_bounds__(NTH) void dw_bwd_kernel(const ctn_dw_bwd_args a, const int CG, const int TT,
                                                     const int lw, const int H) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NV = 5 + 2 * P;
    constexpr int NITB = 3;
    __shared__ float red[(NV > 3 ? NV : 3) * (NTH / 64)];
    float* D = lds;
    float* Xh = lds + lw;
    const int b = blockIdx.z, c0 = blockIdx.y * CG, t0 = blockIdx.x * TT;
    const int T = a.T, d = a.dilation, padL = a.pad_left;
    const int tA = floor4(t0 - H);
    const int tid = threadIdx.x;
    const float a1 = a.alpha1 ? a.alpha1[0] : 1.f;
    const float a2 = a.alpha2 ? a.alpha2[0] : 1.f;
    const bool obwd = (a.oflags & CTN_XF_NORMBWD) != 0;
    const bool inorm = (a.xflags & CTN_XF_NORM) != 0;
    const bool iprelu = (a.xflags & CTN_XF_PRELU) != 0;
    const int cend = min(c0 + CG, a.C);
    const int t_end = min(t0 + TT, T);
    const int lw4 = lw / 4;
    float m2h = 0.f, r2h = 1.f, c1h = 0.f, c2h = 0.f, m1h = 0.f, r1h = 1.f;
    if (obwd && a.sst2 == 0) {
        const int64_t si = (int64_t)b * a.ssb2;
        m2h = a.mean2[si]; r2h = a.rstd2[si]; c1h = a.c1_2[si]; c2h = a.c2_2[si];
    }
    if (inorm && a.sst1 == 0) {
        m1h = a.mean1[(int64_t)b * a.ssb1];
        r1h = a.rstd1[(int64_t)b * a.ssb1];
    }

    float dalpha2 = 0.f, s1 = 0.f, s2 = 0.f;
    float cs1[4] = {0, 0, 0, 0}, cs2[4] = {0, 0, 0, 0};

    for (int c = c0; c < cend; ++c) {
        const float* dnrow = a.dn2 + (int64_t)b * a.dn2bs + (int64_t)c * a.lddn2;
        const float* y2row = obwd ? a.y2 + (int64_t)b * a.y2bs + (int64_t)c * a.ldy2 : nullptr;
        const float* xrow = a.x + (int64_t)b * a.xbs + (int64_t)c * a.ldx;
        const float g2 = obwd ? a.gamma2[c] : 1.f;
        float dgamma2 = 0.f, dbeta2 = 0.f, dbias = 0.f;
        // ---- stage D and Xh (all loads of a round first)
        for (int r0 = 0; r0 < lw4; r0 += NITB * NTH) {
            f32x4 vdn[NITB], vx[NITB], vy[NITB];
#pragma unroll
            for (int j = 0; j < NITB; ++j) {
                const int i4 = r0 + tid + j * NTH;
                const int t = tA + 4 * i4;
                vdn[j] = vx[j] = vy[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i4 < lw4 && t >= 0 && t < T) {
                    vdn[j] = CTN_DW_LD(dnrow + t);
                    vx[j] = CTN_DW_LD(xrow + t);
                    if (obwd) vy[j] = CTN_DW_LD(y2row + t);
                }
            }
#pragma unroll
            for (int j = 0; j < NITB; ++j) {
                const int i4 = r0 + tid + j * NTH;
                const int t = tA + 4 * i4;
                if (i4 >= lw4) continue;
                f32x4 dv = {0.f, 0.f, 0.f, 0.f}, xv = {0.f, 0.f, 0.f, 0.f};
                if (t >= 0 && t < T) {
                    float mu2[4], rs2[4], k1[4], k2[4], mu1[4], rs1[4];
                    if (obwd) {
                        stat4(a.mean2, a.ssb2, a.sst2, b, t, T, m2h, mu2);
                        stat4(a.rstd2, a.ssb2, a.sst2, b, t, T, r2h, rs2);
                        stat4(a.c1_2, a.ssb2, a.sst2, b, t, T, c1h, k1);
                        stat4(a.c2_2, a.ssb2, a.sst2, b, t, T, c2h, k2);
                    }
                    if (inorm) {
                        stat4(a.mean1, a.ssb1, a.sst1, b, t, T, m1h, mu1);
                        stat4(a.rstd1, a.ssb1, a.sst1, b, t, T, r1h, rs1);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool valid = t + e < T;
                        const bool interior = valid && t + e >= t0 && t + e < t_end;
                        const float dn = vdn[j][e];
                        float dy = dn;
                        if (obwd) {
                            const float y2 = vy[j][e];
                            const float u = ctn_prelu(y2, a2);
                            const float xc = u - mu2[e];
                            const float du = g2 * dn * rs2[e] + k1[e] + xc * k2[e];
                            dy = y2 > 0.f ? du : a2 * du;
                            if (interior) {
                                dgamma2 += dn * xc * rs2[e];
                                dbeta2 += dn;
                                if (!(y2 > 0.f)) dalpha2 += du * y2;
                            }
                        }
                        if (interior) dbias += dy;
                        dv[e] = valid ? dy : 0.f;
                        float u1 = vx[j][e];
                        if (iprelu) u1 = ctn_prelu(u1, a1);
                        if (inorm) u1 = (u1 - mu1[e]) * rs1[e];
                        xv[e] = valid ? u1 : 0.f;
                    }
                }
                ctn_st4(&D[4 * i4], dv);
                ctn_st4(&Xh[4 * i4], xv);
            }
        }
        __syncthreads();
        // ---- input gradient, weight gradient, norm-1 reductions
        float w[P], dwa[P], dwb[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            w[k] = a.w[(int64_t)c * P + k];
            dwa[k] = 0.f;
            dwb[k] = 0.f;
        }
        const float g1 = inorm ? a.gamma1[c] : 1.f;
        float dgamma1 = 0.f, dbeta1 = 0.f;
        float* dxrow = a.dx + (int64_t)b * a.dxbs + (int64_t)c * a.lddx;
        for (int t = t0 + 4 * tid; t < t_end; t += 4 * NTH) {
            const int i0 = t - tA;
            const f32x4 dy4 = ctn_ld4(&D[i0]);
            const f32x4 xh4 = ctn_ld4(&Xh[i0]);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const f32x4 dk = lds_read4(D, i0 - k * d + padL);
                const f32x4 xk = lds_read4(Xh, i0 + k * d - padL);
                o += w[k] * dk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ti = t + e + k * d - padL;
                    dwa[k] += dy4[e] * xk[e];
                    dwb[k] += (ti >= 0 && ti < T) ? dy4[e] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (t + e < T) {
                    const float g = g1 * o[e];
                    dgamma1 += o[e] * xh4[e];
                    dbeta1 += o[e];
                    if (a.cln) {
                        cs1[e] += g;
                        cs2[e] += g * xh4[e];
                    } else {
                        s1 += g;
                        s2 += g * xh4[e];
                    }
                }
            }
            CTN_DW_ST(dxrow + t, o);
        }
        // ---- per-channel reductions
        float v[NV];
        v[0] = dgamma2; v[1] = dbeta2; v[2] = dbias; v[3] = dgamma1; v[4] = dbeta1;
        const float be1 = inorm ? a.beta1[c] : 0.f;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            v[5 + k] = g1 * dwa[k] + be1 * dwb[k];
            v[5 + P + k] = 0.f;
        }
        const float tot = ctn_block_sum<NV>(v, red);   // also the barrier before LDS is overwritten
        if (tid == 0 && obwd && a.dgamma2) atomicAdd(a.dgamma2 + c, tot);
        if (tid == 1 && obwd && a.dbeta2) atomicAdd(a.dbeta2 + c, tot);
        if (tid == 2 && a.dbias) atomicAdd(a.dbias + c, tot);
        if (tid == 3 && inorm && a.dgamma1) atomicAdd(a.dgamma1 + c, tot);
        if (tid == 4 && inorm && a.dbeta1) atomicAdd(a.dbeta1 + c, tot);
        if (tid >= 5 && tid < 5 + P) atomicAdd(a.dw + (int64_t)c * P + (tid - 5), tot);
    }
    // ---- per-sample reductions
    {
        float v[3] = {dalpha2, s1, s2};
        const float tot = ctn_block_sum<3>(v, red);
        if (tid == 0 && obwd && a.dalpha2) atomicAdd(a.dalpha2 + ctn_slot(), (double)tot);
        if (!a.cln && a.isums && (tid == 1 || tid == 2)) atomicAdd(a.isums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + (tid - 1), (double)tot);
        __shared__ unsigned fin_flag;
        if (!a.cln && a.isums)
            ctn_finish_sums(a.fin, a.isums + (int64_t)b * CTN_STAT_SLOTS * 2, b, gridDim.x * gridDim.y, &fin_flag);
    }
    if (a.cln && a.icol) {
        const int t = t0 + 4 * tid;
        if (t < t_end) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t + e < T) {
                    atomicAdd(a.icol + ((int64_t)b * 2 + 0) * a.icol_ld + t + e, cs1[e]);
                    atomicAdd(a.icol + ((int64_t)b * 2 + 1) * a.icol_ld + t + e, cs2[e]);
                }
        }
    }
}

// Sums NP (a power of two <= 32) per-lane values over a wave with NP - 1 + log2(64 / NP) exchanges instead of 6 * NP: at
// every step a lane keeps one half of its values and hands the other half to its partner, so the values are spread
// over the lanes while they are summed.  Returns, in every lane l, the wave total of value l % NP.
// (The steps are template instances, not a loop over `step >>= 1`: the compiler does not unroll that loop, and the
// array is then indexed at run time -- a sixteen-way chain of compares and selects in front of every exchange.)
template <int NP, int STEP>
struct wave_scatter_step {
    static __device__ __forceinline__ void run(float (&v)[NP], const int lane) {
        const bool upper = (lane & STEP) != 0;
#pragma unroll
        for (int i = 0; i < STEP; ++i) {
            const float send = upper ? v[i] : v[i + STEP];
            const float keep = upper ? v[i + STEP] : v[i];
            v[i] = keep + __shfl_xor(send, STEP);
        }
        wave_scatter_step<NP, STEP / 2>::run(v, lane);
    }
};
template <int NP>
struct wave_scatter_step<NP, 0> {
    static __device__ __forceinline__ void run(float (&)[NP], const int) {}
};
template <int NP>
__device__ __forceinline__ float wave_sum_scatter(float (&v)[NP]) {
    static_assert(NP >= 2 && NP <= 32 && (NP & (NP - 1)) == 0, "a power of two");
    const int lane = threadIdx.x & 63;
    wave_scatter_step<NP, NP / 2>::run(v, lane);
    float t = v[0];
#pragma unroll
    for (int m = NP; m < 64; m <<= 1) t += __shfl_xor(t, m);
    return t;
}

// ------------------------------------------------------------------------------------------ backward, cLN tiles
// Cumulative norm on both sides: per-frame statistics, so a block walks CG channels of one time tile (TT <= 4 * NTH
// frames) and keeps the per-frame sums of the norm-1 backward in registers.  What the general kernel above does per
// channel and element -- guarded loads, six per-frame statistics fetched as scalars -- happens here once per block:
// the statistics of a thread's staging columns do not depend on the channel, so they are loaded as float4s before the
// channel loop; two channels are staged per phase, all their loads issued first from clamped addresses, validity
// applied by selects.  LDS per channel: D | Xh as above.
template <int P>
__global__ __launch_bounds__(NTH) void dw_bwd_cln_kernel(const ctn_dw_bwd_args a, const int CG, const int TT,
                                                         const int lw, const int H) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [CS][D | Xh][lw]
    constexpr int CS = 2, NITB = 2;                    // host guarantees lw / 4 <= NITB * NTH
    constexpr int NV = 5 + P;
    constexpr int NP = CS * NV <= 16 ? 16 : 32;        // per-phase values padded to a power of two
    constexpr int MAXPH = 8;                           // phases per block: the host launches with CG <= CS * MAXPH
    __shared__ float red[NTH / 64];
    __shared__ float chan[MAXPH * NP];                 // per-channel sums of the block, added to by every wave
    const int b = blockIdx.z, c0 = blockIdx.y * CG, t0 = blockIdx.x * TT;
    const int T = a.T, d = a.dilation, padL = a.pad_left;
    const int tA = floor4(t0 - H);
    const int tid = threadIdx.x;
    const float a1 = a.alpha1[0], a2 = a.alpha2[0];
    const int cend = min(c0 + CG, a.C);
    const int t_end = min(t0 + TT, T);
    const int lw4 = lw / 4;
    const int tl4 = ((T + 3) & ~3) - 4;

    // per-frame statistics of this thread's staging columns
    f32x4 mu2[NITB], rs2[NITB], k1[NITB], k2[NITB], mu1[NITB], rs1[NITB];
    int ts[NITB];
#pragma unroll
    for (int j = 0; j < NITB; ++j) {
        const int i4 = tid + j * NTH;
        const int t = tA + 4 * i4;
        ts[j] = min(max(t, 0), tl4);
        const int64_t s2 = (int64_t)b * a.ssb2 + ts[j], s1 = (int64_t)b * a.ssb1 + ts[j];
        mu2[j] = ctn_ld4(a.mean2 + s2); rs2[j] = ctn_ld4(a.rstd2 + s2);
        k1[j] = ctn_ld4(a.c1_2 + s2);   k2[j] = ctn_ld4(a.c2_2 + s2);
        mu1[j] = ctn_ld4(a.mean1 + s1); rs1[j] = ctn_ld4(a.rstd1 + s1);
    }
    float dalpha2 = 0.f;
    float cs1[4] = {0, 0, 0, 0}, cs2[4] = {0, 0, 0, 0};
    const int t = t0 + 4 * tid;                        // this thread's output frames (one float4 per channel)
    if (tid < MAXPH * NP) chan[tid] = 0.f;             // visible after the first barrier of the loop
    // the block's per-channel constants, fetched once: read from global memory inside the phases they are a dependent
    // round trip in front of every phase's arithmetic
    constexpr int MAXCG = CS * MAXPH;
    __shared__ float pg2[MAXCG], pg1[MAXCG], pb1[MAXCG], pw[MAXCG * P];
    if (tid < MAXCG) {
        const int c = min(c0 + tid, a.C - 1);
        pg2[tid] = a.gamma2[c];
        pg1[tid] = a.gamma1[c];
        pb1[tid] = a.beta1[c];
    }
    if (tid < MAXCG * P) {
        const int c = min(c0 + tid / P, a.C - 1);
        pw[tid] = a.w[(int64_t)c * P + tid % P];
    }
    __syncthreads();

    // (Requesting a phase's rows one phase ahead was measured and is not done: the registers it holds across the tap
    // loops cost the third resident block, 391 against 363 us.)
    f32x4 vdn[CS][NITB], vx[CS][NITB], vy[CS][NITB];
    auto load_phase = [&](const int cg) {
#pragma unroll
        for (int r = 0; r < CS; ++r) {
            const int c = min(cg + r, cend - 1);
#pragma unroll
            for (int j = 0; j < NITB; ++j) {
                vdn[r][j] = CTN_DW_LD(a.dn2 + (int64_t)b * a.dn2bs + (int64_t)c * a.lddn2 + ts[j]);
                vy[r][j] = CTN_DW_LD(a.y2 + (int64_t)b * a.y2bs + (int64_t)c * a.ldy2 + ts[j]);
                vx[r][j] = CTN_DW_LD(a.x + (int64_t)b * a.xbs + (int64_t)c * a.ldx + ts[j]);
            }
        }
    };
    for (int cg = c0; cg < cend; cg += CS) {
        load_phase(cg);
        float v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i] = 0.f;
#pragma unroll
        for (int r = 0; r < CS; ++r) {
            const bool ch_ok = cg + r < cend;
            const float g2 = pg2[min(cg + r, cend - 1) - c0];
            float* D = lds + (size_t)r * 2 * lw;
            float* Xh = D + lw;
            float dgamma2 = 0.f, dbeta2 = 0.f, dbias = 0.f;
#pragma unroll
            for (int j = 0; j < NITB; ++j) {
                const int i4 = tid + j * NTH;
                if (i4 >= lw4) continue;
                // The frame index passes through an opaque move once per phase: the validity masks below depend on
                // nothing else, the compiler would otherwise keep all sixteen of them (64-bit lane masks) alive across
                // the phases, and with the scalar registers gone they come back through v_readlane -- a quarter of the
                // loop's vector instructions.  Recomputing them is four compares each.
                int tj = tA + 4 * i4;
                asm volatile("" : "+v"(tj));
                f32x4 dv, xv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool valid = ch_ok && tj + e >= 0 && tj + e < T;
                    const bool interior = valid && tj + e >= t0 && tj + e < t_end;
                    const float dn = vdn[r][j][e], y2 = vy[r][j][e];
                    const float sl = y2 > 0.f ? 1.f : a2;
                    const float xc = sl * y2 - mu2[j][e];
                    const float du = g2 * dn * rs2[j][e] + k1[j][e] + xc * k2[j][e];
                    const float dy = sl * du;
                    dgamma2 += interior ? dn * xc * rs2[j][e] : 0.f;
                    dbeta2 += interior ? dn : 0.f;
                    dalpha2 += (interior && !(y2 > 0.f)) ? du * y2 : 0.f;
                    dbias += interior ? dy : 0.f;
                    dv[e] = valid ? dy : 0.f;
                    const float u1 = (ctn_prelu(vx[r][j][e], a1) - mu1[j][e]) * rs1[j][e];
                    xv[e] = valid ? u1 : 0.f;
                }
                ctn_st4(&D[4 * i4], dv);
                ctn_st4(&Xh[4 * i4], xv);
            }
            v[r * NV + 0] = dgamma2; v[r * NV + 1] = dbeta2; v[r * NV + 2] = dbias;
        }
        __syncthreads();
        // ---- input gradient, weight gradient, norm-1 reductions
        if (t < t_end) {
            const int i0 = t - tA;
#pragma unroll
            for (int r = 0; r < CS; ++r) {
                const int c = cg + r;
                if (c >= cend) break;
                const float* D = lds + (size_t)r * 2 * lw;
                const float* Xh = D + lw;
                const float g1 = pg1[c - c0], be1 = pb1[c - c0];
                const f32x4 dy4 = ctn_ld4(&D[i0]);
                const f32x4 xh4 = ctn_ld4(&Xh[i0]);
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    const f32x4 dk = lds_read4(D, i0 - k * d + padL);
                    const f32x4 xk = lds_read4(Xh, i0 + k * d - padL);
                    o += pw[(c - c0) * P + k] * dk;
                    float dwa = 0.f, dwb = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ti = t + e + k * d - padL;
                        dwa += dy4[e] * xk[e];
                        dwb += (ti >= 0 && ti < T) ? dy4[e] : 0.f;
                    }
                    v[r * NV + 5 + k] = g1 * dwa + be1 * dwb;
                }
                float dgamma1 = 0.f, dbeta1 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (t + e < T) {
                        const float g = g1 * o[e];
                        dgamma1 += o[e] * xh4[e];
                        dbeta1 += o[e];
                        cs1[e] += g;
                        cs2[e] += g * xh4[e];
                    }
                }
                v[r * NV + 3] = dgamma1; v[r * NV + 4] = dbeta1;
                CTN_DW_ST(a.dx + (int64_t)b * a.dxbs + (int64_t)c * a.lddx + t, o);
            }
        }
        // ---- per-channel reductions of both channels at once: summed over the wave, then into the block's LDS table
        const float tot = wave_sum_scatter<NP>(v);
        if ((tid & 63) < NP) atomicAdd(&chan[((cg - c0) / CS) * NP + (tid & 63)], tot);
        __syncthreads();                               // the LDS rows are free for the next phase
    }
    if (tid < MAXPH * NP) {                            // one global atomic per channel and quantity
        const int ph = tid / NP, i = tid - ph * NP;
        const int r = i / NV, q = i - r * NV, c = c0 + ph * CS + r;
        if (i < CS * NV && c < cend) {
            const float tot = chan[tid];
            if (q == 0 && a.dgamma2) atomicAdd(a.dgamma2 + c, tot);
            if (q == 1 && a.dbeta2) atomicAdd(a.dbeta2 + c, tot);
            if (q == 2 && a.dbias) atomicAdd(a.dbias + c, tot);
            if (q == 3 && a.dgamma1) atomicAdd(a.dgamma1 + c, tot);
            if (q == 4 && a.dbeta1) atomicAdd(a.dbeta1 + c, tot);
            if (q >= 5) atomicAdd(a.dw + (int64_t)c * P + (q - 5), tot);
        }
    }
    {
        float v1[1] = {dalpha2};
        const float tot = ctn_block_sum<1>(v1, red);
        if (tid == 0 && a.dalpha2) atomicAdd(a.dalpha2 + ctn_slot(), (double)tot);
    }
    if (a.icol) col_atomics(lds, cs1, cs2, a.icol + (int64_t)b * 2 * a.icol_ld, a.icol_ld, t0, t_end);
}

// ------------------------------------------------------------------------------------------ backward, gLN rows
// The separator's hot case as straight-line code: per-batch statistics (gLN), output-side norm/PReLU backward,
// input-side PReLU + norm, one block per (batch, channel) row that fits 4 float4 per thread.  Every global load
// of the row is issued before anything is consumed (unconditional, clamped addresses, validity by selects):
// the general kernel above guards its loads and statistics fetches, which the compiler turns into branches
// with full vmcnt waits.
constexpr int GIT = 4;                 // float4 per thread: rows of up to 4 * 256 * 4 = 4096 frames
template <int P>
__global__ __launch_bounds__(NTH) void dw_bwd_gln_kernel(const ctn_dw_bwd_args a, const int lw, const int H4) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NV = 8 + P;
    __shared__ float red[NV * (NTH / 64)];
    float* D = lds;
    float* Xh = lds + lw;
    const int b = blockIdx.y, c = blockIdx.x;
    const int T = a.T, d = a.dilation, padL = a.pad_left;
    const int tid = threadIdx.x;
    const int ldmin = (int)min(min(a.lddn2, a.ldy2), a.ldx);      // every row holds at least this many floats
    const float* dnrow = a.dn2 + (int64_t)b * a.dn2bs + (int64_t)c * a.lddn2;
    const float* y2row = a.y2 + (int64_t)b * a.y2bs + (int64_t)c * a.ldy2;
    const float* xrow = a.x + (int64_t)b * a.xbs + (int64_t)c * a.ldx;
    f32x4 vdn[GIT], vy[GIT], vx[GIT];
#pragma unroll
    for (int j = 0; j < GIT; ++j) {
        const int tq = min(4 * (tid + j * NTH), ldmin - 4);
        vdn[j] = CTN_DW_LD(dnrow + tq);
        vy[j] = CTN_DW_LD(y2row + tq);
        vx[j] = CTN_DW_LD(xrow + tq);
    }
    const float a1 = a.alpha1[0], a2 = a.alpha2[0];
    const float m2 = a.mean2[(int64_t)b * a.ssb2], r2 = a.rstd2[(int64_t)b * a.ssb2];
    const float k1 = a.c1_2[(int64_t)b * a.ssb2], k2 = a.c2_2[(int64_t)b * a.ssb2];
    const float m1 = a.mean1[(int64_t)b * a.ssb1], r1 = a.rstd1[(int64_t)b * a.ssb1];
    const float g2 = a.gamma2[c], g1 = a.gamma1[c], be1 = a.beta1[c];
    float w[P];
#pragma unroll
    for (int k = 0; k < P; ++k) w[k] = a.w[(int64_t)c * P + k];
    // zero halos: [0, H4) and [H4 + 4 * GIT * NTH ... ) are covered by clearing everything outside the row slots
    for (int i = tid; i < H4 / 4; i += NTH) {
        ctn_st4(&D[4 * i], f32x4{0.f, 0.f, 0.f, 0.f});
        ctn_st4(&Xh[4 * i], f32x4{0.f, 0.f, 0.f, 0.f});
    }
    for (int i = H4 / 4 + GIT * NTH + tid; i < lw / 4; i += NTH) {
        ctn_st4(&D[4 * i], f32x4{0.f, 0.f, 0.f, 0.f});
        ctn_st4(&Xh[4 * i], f32x4{0.f, 0.f, 0.f, 0.f});
    }
    float dgamma2 = 0.f, dbeta2 = 0.f, dbias = 0.f, dalpha2 = 0.f;
#pragma unroll
    for (int j = 0; j < GIT; ++j) {
        const int t = 4 * (tid + j * NTH);
        f32x4 dv, xv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool valid = t + e < T;
            const float dn = vdn[j][e], y2 = vy[j][e];
            const float u = ctn_prelu(y2, a2);
            const float xc = u - m2;
            const float du = g2 * dn * r2 + k1 + xc * k2;
            const float dy = y2 > 0.f ? du : a2 * du;
            dgamma2 += valid ? dn * xc * r2 : 0.f;
            dbeta2 += valid ? dn : 0.f;
            dalpha2 += (valid && !(y2 > 0.f)) ? du * y2 : 0.f;
            dbias += valid ? dy : 0.f;
            dv[e] = valid ? dy : 0.f;
            const float xh = (ctn_prelu(vx[j][e], a1) - m1) * r1;
            xv[e] = valid ? xh : 0.f;
        }
        ctn_st4(&D[H4 + t], dv);
        ctn_st4(&Xh[H4 + t], xv);
    }
    __syncthreads();
    float dwa[P], dwb[P];
#pragma unroll
    for (int k = 0; k < P; ++k) dwa[k] = dwb[k] = 0.f;
    float dgamma1 = 0.f, dbeta1 = 0.f, s1 = 0.f, s2 = 0.f;
    float* dxrow = a.dx + (int64_t)b * a.dxbs + (int64_t)c * a.lddx;
#pragma unroll
    for (int j = 0; j < GIT; ++j) {
        const int t = 4 * (tid + j * NTH);
        if (t < T) {
            const int i0 = H4 + t;
            const f32x4 dy4 = ctn_ld4(&D[i0]);
            const f32x4 xh4 = ctn_ld4(&Xh[i0]);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const f32x4 dk = lds_read4(D, i0 - k * d + padL);
                const f32x4 xk = lds_read4(Xh, i0 + k * d - padL);
                o += w[k] * dk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ti = t + e + k * d - padL;
                    dwa[k] += dy4[e] * xk[e];
                    dwb[k] += (ti >= 0 && ti < T) ? dy4[e] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (t + e < T) {
                    const float g = g1 * o[e];
                    dgamma1 += o[e] * xh4[e];
                    dbeta1 += o[e];
                    s1 += g;
                    s2 += g * xh4[e];
                }
            }
            CTN_DW_ST(dxrow + t, o);
        }
    }
    float v[NV];
    v[0] = dgamma2; v[1] = dbeta2; v[2] = dbias; v[3] = dgamma1; v[4] = dbeta1;
    v[5] = dalpha2; v[6] = s1; v[7] = s2;
#pragma unroll
    for (int k = 0; k < P; ++k) v[8 + k] = g1 * dwa[k] + be1 * dwb[k];
    const float tot = ctn_block_sum<NV>(v, red);
    if (tid == 0 && a.dgamma2) atomicAdd(a.dgamma2 + c, tot);
    if (tid == 1 && a.dbeta2) atomicAdd(a.dbeta2 + c, tot);
    if (tid == 2 && a.dbias) atomicAdd(a.dbias + c, tot);
    if (tid == 3 && a.dgamma1) atomicAdd(a.dgamma1 + c, tot);
    if (tid == 4 && a.dbeta1) atomicAdd(a.dbeta1 + c, tot);
    if (tid == 5 && a.dalpha2) atomicAdd(a.dalpha2 + ctn_slot(), (double)tot);
    if ((tid == 6 || tid == 7) && a.isums)
        atomicAdd(a.isums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + (tid - 6), (double)tot);
    if (tid >= 8 && tid < 8 + P) atomicAdd(a.dw + (int64_t)c * P + (tid - 8), tot);
    __shared__ unsigned fin_flag;
    if (a.isums) ctn_finish_sums(a.fin, a.isums + (int64_t)b * CTN_STAT_SLOTS * 2, b, gridDim.x, &fin_flag);
}

// ------------------------------------------------------------------------------------------ backward, gLN rows, y2 recomputed
// The same row-per-block backward, but the convolution output y2 is recomputed from the input row instead of being
// read: three tensors cross HBM (x and dn2 in, dn1 out) instead of four.  The LDS row first holds the convolution's
// actual input N -- gamma1 * xhat1 + beta1 inside the row, zero outside -- so that
//   y2[t]  = bias + sum_k w[k] N[t + k d - padL]            (what the forward kernel computed, same expression order)
//   dw[k]  = sum_t dy2[t] N[t + k d - padL]                 (no separate beta1 term: the padding zeros are in N)
// and the taps fetched for y2 are used for dw in the same breath; then the SAME row is overwritten with dy2 for the
// taps of the input gradient.  xhat1 and dy2 of a thread's own frames stay in registers (the thread that stages frame
// t is the one that finishes it).  One LDS row of 17 KB instead of two: eight blocks per CU instead of four, i.e.
// twice the rows in flight per CU, which is what a kernel of load / compute / store phases needs to keep HBM busy.
// Phases: stage N | y2, dy2, dw | stage dy2 | dn1 and the norm-1 sums; three barriers.
// AL: dilation and left padding are multiples of 4, so every tap is one aligned 16-byte LDS read and the loops carry no
// alignment branches (they would fence the scheduling of the reads of different taps).
// DD, PL: dilation and left padding as compile-time constants (0 = take them from the arguments).  With them the
// unaligned taps of the small dilations -- three aligned reads per phase and a pick of four registers each -- cost no
// instructions at all: the picks become register names.  Instantiated for the non-causal dilations 1 and 2.
template <int P, bool AL, int DD = 0, int PL = 0>
__global__ __launch_bounds__(NTH) void dw_bwd_gln_rc_kernel(const ctn_dw_bwd_args a, const int lw, const int H4) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NV = 10 + P;
    __shared__ float red[NV * (NTH / 64)];
    float* row = lds;
    const int b = blockIdx.y, c = blockIdx.x;
    const int T = a.T, d = DD ? DD : a.dilation, padL = DD ? PL : a.pad_left;
    const int tid = threadIdx.x;
    const int ldmin = (int)min(a.lddn2, a.ldx);
    const float* dnrow = a.dn2 + (int64_t)b * a.dn2bs + (int64_t)c * a.lddn2;
    const float* xrow = a.x + (int64_t)b * a.xbs + (int64_t)c * a.ldx;
    f32x4 vdn[GIT], xh[GIT];
#pragma unroll
    for (int j = 0; j < GIT; ++j) {
        const int tq = min(4 * (tid + j * NTH), ldmin - 4);
        xh[j] = CTN_DW_LD(xrow + tq);
        vdn[j] = CTN_DW_LD(dnrow + tq);
    }
    const float a1 = a.alpha1[0], a2 = a.alpha2[0];
    const float m2 = a.mean2[(int64_t)b * a.ssb2], r2 = a.rstd2[(int64_t)b * a.ssb2];
    const float k1 = a.c1_2[(int64_t)b * a.ssb2], k2 = a.c2_2[(int64_t)b * a.ssb2];
    const float m1 = a.mean1[(int64_t)b * a.ssb1], r1 = a.rstd1[(int64_t)b * a.ssb1];
    const float g2 = a.gamma2[c], g1 = a.gamma1[c], be1 = a.beta1[c];
    const float bias = a.bias ? a.bias[c] : 0.f;
    float w[P];
#pragma unroll
    for (int k = 0; k < P; ++k) w[k] = a.w[(int64_t)c * P + k];
    int remf[P], remb[P];                // tap offsets modulo 4 (H4 and t are multiples of 4): forward taps, backward taps
#pragma unroll
    for (int k = 0; k < P; ++k) {
        remf[k] = (k * d - padL) & 3;
        remb[k] = (padL - k * d) & 3;
    }
    // the halos on both sides of the row slots stay zero through both uses of the row
    for (int i = tid; i < H4 / 4; i += NTH) ctn_st4(&row[4 * i], f32x4{0.f, 0.f, 0.f, 0.f});
    for (int i = H4 / 4 + GIT * NTH + tid; i < lw / 4; i += NTH) ctn_st4(&row[4 * i], f32x4{0.f, 0.f, 0.f, 0.f});
    // ---- phase 1: the convolution's input row
#pragma unroll
    for (int j = 0; j < GIT; ++j) {
        const int t = 4 * (tid + j * NTH);
        f32x4 nv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool valid = t + e < T;
            const float h = (ctn_prelu(xh[j][e], a1) - m1) * r1;
            xh[j][e] = valid ? h : 0.f;
            nv[e] = valid ? __builtin_fmaf(h, g1, be1) : 0.f;
        }
        ctn_st4(&row[H4 + t], nv);
    }
    __syncthreads();
    // ---- phase 2: y2, the gradient of the convolution output (kept in vdn), the tap gradients
    float dgamma2 = 0.f, dbeta2 = 0.f, dbias = 0.f;
    // slope gradient sum du * min(y2, 0) with du = A dn + k1 + xc k2 as three sums: the parts of du cancel, and so do
    // the terms across the sample, so they are combined in fp64 per block instead of being rounded per element
    float da_dn = 0.f, da_1 = 0.f, da_xc = 0.f;
    float dwa[P];
#pragma unroll
    for (int k = 0; k < P; ++k) dwa[k] = 0.f;
#pragma unroll
    for (int j = 0; j < GIT; ++j) {
        const int t = 4 * (tid + j * NTH);
        f32x4 dv = {0.f, 0.f, 0.f, 0.f};
        if (t < T) {
            const int i0 = H4 + t;
            f32x4 nk[P];
            f32x4 y4 = {bias, bias, bias, bias};
#pragma unroll
            for (int k = 0; k < P; ++k) {
                nk[k] = AL ? ctn_ld4(row + i0 + k * d - padL) : lds_read4s(row, i0 + k * d - padL, remf[k]);
                y4 += w[k] * nk[k];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool valid = t + e < T;
                const float dn = vdn[j][e], y2 = y4[e];
                const float u = ctn_prelu(y2, a2);
                const float xc = u - m2;
                const float du = g2 * dn * r2 + k1 + xc * k2;
                const float dy = y2 > 0.f ? du : a2 * du;
                dgamma2 += valid ? dn * xc * r2 : 0.f;
                dbeta2 += valid ? dn : 0.f;
                const float yn = (valid && !(y2 > 0.f)) ? y2 : 0.f;
                da_dn = __builtin_fmaf(dn, yn, da_dn);
                da_1 += yn;
                da_xc = __builtin_fmaf(xc, yn, da_xc);
                dbias += valid ? dy : 0.f;
                dv[e] = valid ? dy : 0.f;
            }
#pragma unroll
            for (int k = 0; k < P; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) dwa[k] += dv[e] * nk[k][e];
        }
        vdn[j] = dv;
    }
    __syncthreads();              // every tap of N has been read: the row is free
#pragma unroll
    for (int j = 0; j < GIT; ++j) ctn_st4(&row[H4 + 4 * (tid + j * NTH)], vdn[j]);
    __syncthreads();
    // ---- phase 3: the input gradient and the norm-1 backward sums
    // the norm-1 backward sums of this row are g1 * (sum dn1) and g1 * (sum dn1 * xhat1): the per-channel gradients
    // times gamma1, so they need no accumulators of their own
    float dgamma1 = 0.f, dbeta1 = 0.f;
    float* dxrow = a.dx + (int64_t)b * a.dxbs + (int64_t)c * a.lddx;
#pragma unroll
    for (int j = 0; j < GIT; ++j) {
        const int t = 4 * (tid + j * NTH);
        if (t < T) {
            const int i0 = H4 + t;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < P; ++k)
                o += w[k] * (AL ? ctn_ld4(row + i0 - k * d + padL) : lds_read4s(row, i0 - k * d + padL, remb[k]));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (t + e < T) {
                    dgamma1 += o[e] * xh[j][e];
                    dbeta1 += o[e];
                }
            }
            CTN_DW_ST(dxrow + t, o);
        }
    }
    float v[NV];
    v[0] = dgamma2; v[1] = dbeta2; v[2] = dbias; v[3] = dgamma1; v[4] = dbeta1;
    v[5] = da_dn; v[6] = g1 * dbeta1; v[7] = g1 * dgamma1; v[8] = da_1; v[9] = da_xc;
#pragma unroll
    for (int k = 0; k < P; ++k) v[10 + k] = dwa[k];
    const float tot = ctn_block_sum<NV>(v, red);
    if (tid == 0 && a.dgamma2) atomicAdd(a.dgamma2 + c, tot);
    if (tid == 1 && a.dbeta2) atomicAdd(a.dbeta2 + c, tot);
    if (tid == 2 && a.dbias) atomicAdd(a.dbias + c, tot);
    if (tid == 3 && a.dgamma1) atomicAdd(a.dgamma1 + c, tot);
    if (tid == 4 && a.dbeta1) atomicAdd(a.dbeta1 + c, tot);
    // threads 5, 8, 9 hold the three slope sums: fold them on thread 5 through LDS
    if (tid == 8 || tid == 9) red[tid - 8] = tot;
    __syncthreads();
    if (tid == 5 && a.dalpha2)
        atomicAdd(a.dalpha2 + ctn_slot(), (double)g2 * (double)r2 * (double)tot + (double)k1 * (double)red[0] +
                                                   (double)k2 * (double)red[1]);
    if ((tid == 6 || tid == 7) && a.isums)
        atomicAdd(a.isums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + (tid - 6), (double)tot);
    if (tid >= 10 && tid < 10 + P) atomicAdd(a.dw + (int64_t)c * P + (tid - 10), tot);
    __shared__ unsigned fin_flag;
    if (a.isums) ctn_finish_sums(a.fin, a.isums + (int64_t)b * CTN_STAT_SLOTS * 2, b, gridDim.x, &fin_flag);
}

bool act_ok(const float* p, int64_t ld, int64_t bs, int T) {
    return p && ctn_aligned16(p) && ld % 4 == 0 && bs % 4 == 0 && ld >= T;
}

// Tile choice: whole rows when they fit (no halo re-reads); cLN needs <= 1 float4 per thread per channel.
void pick_tile(int T, int halo2, bool cln, int lds_rows, int* CG, int* TT, int* lw) {
    const int max_floats = (cln ? 48 : 40) * 1024 / 4 / lds_rows;
    int tt = (T + 3) & ~3;
    if (cln) tt = tt < 4 * NTH ? tt : 4 * NTH;
    while (tt + halo2 + 3 > max_floats && tt > 256) tt = ((tt / 2) + 3) & ~3;
    *TT = tt;
    *lw = ((tt + halo2 + 3 + 3) & ~3) + 4;   // + 4: slack for the unaligned tap reads
    *CG = cln ? 8 : 1;
}

}  // namespace

extern "C" {

int ctn_dw_fwd(const ctn_dw_fwd_args* args, void* stream) {
    CTN_REQUIRE(args, "ctn_dw_fwd: null args");
    const ctn_dw_fwd_args& a = *args;
    CTN_REQUIRE(a.batch > 0 && a.C > 0 && a.T > 0 && a.P >= 1 && a.P <= MAXP && a.dilation >= 1,
                "ctn_dw_fwd: bad shape (P <= %d)", MAXP);
    CTN_REQUIRE(a.pad_left >= 0 && a.pad_left <= (a.P - 1) * a.dilation, "ctn_dw_fwd: pad_left out of range");
    CTN_REQUIRE(act_ok(a.x, a.ldx, a.xbs, a.T) && act_ok(a.y, a.ldy, a.ybs, a.T) && a.w, "ctn_dw_fwd: x / y layout");
    if (a.xflags & CTN_XF_NORM)
        CTN_REQUIRE(a.mean && a.rstd && a.gamma && a.beta, "ctn_dw_fwd: NORM needs mean, rstd, gamma, beta");
    const bool stats = a.alpha_out || a.esums || a.ecol;
    if (stats) CTN_REQUIRE(a.cln ? (a.ecol && a.ecol_ld >= a.T) : (a.esums != nullptr), "ctn_dw_fwd: statistics buffer");
    CTN_REQUIRE(ctn_finish_ok(a.fin) && (!a.fin.count || (a.esums && !a.cln)), "ctn_dw_fwd: fin needs esums (gLN), out0, out1, n");
    int CG, TT, lw;
    pick_tile(a.T, (a.P - 1) * a.dilation, a.cln != 0, 1, &CG, &TT, &lw);
    CTN_REQUIRE((size_t)lw * 4 <= 64 * 1024, "ctn_dw_fwd: dilation halo too large for LDS");
    constexpr int CS = 4;                               // rows staged at once by the cLN kernel
    // statistics rows are read as float4s: unit stride, 16-byte aligned rows whose pitch covers T rounded up to 4
    const bool stat_rows_ok = !(a.xflags & CTN_XF_NORM) ||
                              (a.sst == 1 && ctn_aligned16(a.mean) && ctn_aligned16(a.rstd) && a.ssb % 4 == 0 &&
                               a.ssb >= ((a.T + 3) & ~3));
    if (a.cln && stats && (a.P - 1) * a.dilation <= 500 && stat_rows_ok) {
        CG = 16;
        dim3 grid(ctn_cdiv(a.T, TT), ctn_cdiv(a.C, CG), a.batch);
        const size_t shm = std::max((size_t)CS * lw, (size_t)8 * NTH) * 4;   // rows, or the per-frame sums on the way out
        hipLaunchKernelGGL(dw_fwd_cln_kernel<CS>, grid, dim3(NTH), shm, (hipStream_t)stream, a, CG, TT, lw);
        CTN_CHECK_LAUNCH("ctn_dw_fwd");
        return CTN_OK;
    }
    {   // whole-row kernel: per-sample statistics (gLN) or none on the input, gLN sums or none on the output
        const bool in_ok = !(a.xflags & CTN_XF_NORM) || a.sst == 0;
        const bool out_ok = !stats || (!a.cln && a.esums);
        const int HL4 = (a.pad_left + 3) & ~3;
        const int HR4 = ((a.P - 1) * a.dilation - a.pad_left + 3) & ~3;
        const int lwr = HL4 + 4 * FIT * NTH + HR4 + 8;
        if (in_ok && out_ok && a.P <= 4 && a.T >= 4 && a.T <= 4 * FIT * NTH && (size_t)lwr * 4 <= 60 * 1024 &&
            a.C <= 65535 * 32 && a.batch <= 65535) {
            dim3 gridr(a.C, a.batch);
            const size_t shm = (size_t)lwr * 4;
            hipStream_t s = (hipStream_t)stream;
            switch (a.P) {
                case 1: hipLaunchKernelGGL(dw_fwd_row_kernel<1>, gridr, dim3(NTH), shm, s, a, lwr, HL4); break;
                case 2: hipLaunchKernelGGL(dw_fwd_row_kernel<2>, gridr, dim3(NTH), shm, s, a, lwr, HL4); break;
                case 3: hipLaunchKernelGGL(dw_fwd_row_kernel<3>, gridr, dim3(NTH), shm, s, a, lwr, HL4); break;
                default: hipLaunchKernelGGL(dw_fwd_row_kernel<4>, gridr, dim3(NTH), shm, s, a, lwr, HL4); break;
            }
            CTN_CHECK_LAUNCH("ctn_dw_fwd");
            return CTN_OK;
        }
    }
    dim3 grid(ctn_cdiv(a.T, TT), ctn_cdiv(a.C, CG), a.batch);
    hipLaunchKernelGGL(dw_fwd_kernel, grid, dim3(NTH), (size_t)lw * 4, (hipStream_t)stream, a, CG, TT, lw);
    CTN_CHECK_LAUNCH("ctn_dw_fwd");
    return CTN_OK;
}

int ctn_dw_bwd(const ctn_dw_bwd_args* args, void* stream) {
    CTN_REQUIRE(args, "ctn_dw_bwd: null args");
    const ctn_dw_bwd_args& a = *args;
    CTN_REQUIRE(a.batch > 0 && a.C > 0 && a.T > 0 && a.P >= 1 && a.P <= MAXP && a.dilation >= 1,
                "ctn_dw_bwd: bad shape (P <= %d)", MAXP);
    CTN_REQUIRE(a.pad_left >= 0 && a.pad_left <= (a.P - 1) * a.dilation, "ctn_dw_bwd: pad_left out of range");
    CTN_REQUIRE(act_ok(a.dn2, a.lddn2, a.dn2bs, a.T) && act_ok(a.x, a.ldx, a.xbs, a.T) &&
                    act_ok(a.dx, a.lddx, a.dxbs, a.T) && a.w && a.dw,
                "ctn_dw_bwd: dn2 / x / dx layout");
    if (a.oflags & CTN_XF_NORMBWD)
        CTN_REQUIRE((a.recompute_y2 || act_ok(a.y2, a.ldy2, a.y2bs, a.T)) && a.mean2 && a.rstd2 && a.c1_2 && a.c2_2 && a.gamma2,
                    "ctn_dw_bwd: output-side NORMBWD needs y2, stats, gamma2");
    if (a.xflags & CTN_XF_NORM)
        CTN_REQUIRE(a.mean1 && a.rstd1 && a.gamma1 && a.beta1, "ctn_dw_bwd: input-side NORM needs stats, gamma1, beta1");
    if (a.cln) CTN_REQUIRE(!a.icol || a.icol_ld >= a.T, "ctn_dw_bwd: icol_ld < T");
    CTN_REQUIRE(ctn_finish_ok(a.fin) && (!a.fin.count || (a.isums && !a.cln)), "ctn_dw_bwd: fin needs isums (gLN), out0, out1, n, rstd");
    const int padR = (a.P - 1) * a.dilation - a.pad_left;
    const int H = a.pad_left > padR ? a.pad_left : padR;
    hipStream_t s = (hipStream_t)stream;
    {   // whole-row gLN kernel: the separator's configuration
        const bool full = (a.oflags & CTN_XF_NORMBWD) && (a.xflags & CTN_XF_NORM) && (a.xflags & CTN_XF_PRELU) &&
                          a.alpha1 && a.alpha2 && !a.cln && a.sst1 == 0 && a.sst2 == 0;
        const int H4 = (H + 3) & ~3;
        const int lwg = H4 + 4 * GIT * NTH + H4 + 8;
        if (full && a.P <= 4 && a.T <= 4 * GIT * NTH && (size_t)lwg * 8 <= 60 * 1024 && a.T >= 4) {
            dim3 grid(a.C, a.batch);
            const size_t shm = (size_t)lwg * 8;
            if (a.recompute_y2) {
                const size_t shm1 = (size_t)lwg * 4;          // one row
                const bool al = ((a.dilation | a.pad_left) & 3) == 0;
#define CTN_RC_CASE(PP)                                                                                             \
    case PP:                                                                                                        \
        if (al) hipLaunchKernelGGL((dw_bwd_gln_rc_kernel<PP, true>), grid, dim3(NTH), shm1, s, a, lwg, H4);          \
        else hipLaunchKernelGGL((dw_bwd_gln_rc_kernel<PP, false>), grid, dim3(NTH), shm1, s, a, lwg, H4);            \
        break;
                if (a.P == 3 && a.dilation == 1 && a.pad_left == 1)
                    hipLaunchKernelGGL((dw_bwd_gln_rc_kernel<3, false, 1, 1>), grid, dim3(NTH), shm1, s, a, lwg, H4);
                else if (a.P == 3 && a.dilation == 2 && a.pad_left == 2)
                    hipLaunchKernelGGL((dw_bwd_gln_rc_kernel<3, false, 2, 2>), grid, dim3(NTH), shm1, s, a, lwg, H4);
                else
                    switch (a.P) { CTN_RC_CASE(1) CTN_RC_CASE(2) CTN_RC_CASE(3) default: CTN_RC_CASE(4) }
#undef CTN_RC_CASE
                CTN_CHECK_LAUNCH("ctn_dw_bwd");
                return CTN_OK;
            }
            switch (a.P) {
                case 1: hipLaunchKernelGGL(dw_bwd_gln_kernel<1>, grid, dim3(NTH), shm, s, a, lwg, H4); break;
                case 2: hipLaunchKernelGGL(dw_bwd_gln_kernel<2>, grid, dim3(NTH), shm, s, a, lwg, H4); break;
                case 3: hipLaunchKernelGGL(dw_bwd_gln_kernel<3>, grid, dim3(NTH), shm, s, a, lwg, H4); break;
                default: hipLaunchKernelGGL(dw_bwd_gln_kernel<4>, grid, dim3(NTH), shm, s, a, lwg, H4); break;
            }
            CTN_CHECK_LAUNCH("ctn_dw_bwd");
            return CTN_OK;
        }
    }
    if (a.oflags & CTN_XF_NORMBWD)
        CTN_REQUIRE(act_ok(a.y2, a.ldy2, a.y2bs, a.T), "ctn_dw_bwd: this launch cannot recompute y2, pass it");
    int CG, TT, lw;
    pick_tile(a.T, 2 * H, a.cln != 0, 2, &CG, &TT, &lw);
    CTN_REQUIRE((size_t)lw * 8 <= 64 * 1024, "ctn_dw_bwd: dilation halo too large for LDS");
    {   // cumulative norm on both sides (the causal separator's configuration), statistics rows 16-byte aligned
        auto row_ok = [&](const float* p, int64_t sb) { return ctn_aligned16(p) && sb % 4 == 0 && sb >= ((a.T + 3) & ~3); };
        const bool full = (a.oflags & CTN_XF_NORMBWD) && (a.xflags & CTN_XF_NORM) && (a.xflags & CTN_XF_PRELU) &&
                          a.alpha1 && a.alpha2 && a.cln && a.sst1 == 1 && a.sst2 == 1 && a.P <= 4 && a.T >= 4 &&
                          lw / 4 <= 2 * NTH && row_ok(a.mean1, a.ssb1) && row_ok(a.rstd1, a.ssb1) &&
                          row_ok(a.mean2, a.ssb2) && row_ok(a.rstd2, a.ssb2) && row_ok(a.c1_2, a.ssb2) &&
                          row_ok(a.c2_2, a.ssb2);
        if (full) {
            CG = 16;
            dim3 gridc(ctn_cdiv(a.T, TT), ctn_cdiv(a.C, CG), a.batch);
            const size_t shm = std::max((size_t)2 * 2 * lw, (size_t)8 * NTH) * 4;
            switch (a.P) {
                case 1: hipLaunchKernelGGL(dw_bwd_cln_kernel<1>, gridc, dim3(NTH), shm, s, a, CG, TT, lw, H); break;
                case 2: hipLaunchKernelGGL(dw_bwd_cln_kernel<2>, gridc, dim3(NTH), shm, s, a, CG, TT, lw, H); break;
                case 3: hipLaunchKernelGGL(dw_bwd_cln_kernel<3>, gridc, dim3(NTH), shm, s, a, CG, TT, lw, H); break;
                default: hipLaunchKernelGGL(dw_bwd_cln_kernel<4>, gridc, dim3(NTH), shm, s, a, CG, TT, lw, H); break;
            }
            CTN_CHECK_LAUNCH("ctn_dw_bwd");
            return CTN_OK;
        }
    }
    dim3 grid(ctn_cdiv(a.T, TT), ctn_cdiv(a.C, CG), a.batch);
    const size_t shmem = (size_t)lw * 8;
#define CTN_DW_BWD_CASE(PP) \
    case PP: hipLaunchKernelGGL(dw_bwd_kernel<PP>, grid, dim3(NTH), shmem, s, a, CG, TT, lw, H); break;
    switch (a.P) {
        CTN_DW_BWD_CASE(1) CTN_DW_BWD_CASE(2) CTN_DW_BWD_CASE(3) CTN_DW_BWD_CASE(4)
        CTN_DW_BWD_CASE(5) CTN_DW_BWD_CASE(6) CTN_DW_BWD_CASE(7) CTN_DW_BWD_CASE(8)
    }
#undef CTN_DW_BWD_CASE
    CTN_CHECK_LAUNCH("ctn_dw_bwd");
    return CTN_OK;
}

}  // extern "C"
