// Batched pointwise-convolution GEMM on the f32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   Y[b] (M x T) = A (k-major, K x M) . xform(X[b]) (K x T)        per batch element b
//
// Block tile BM x 128 (time) x 32 (k), 4 waves.  Both operand tiles sit k-major in LDS
// (As[k][m], Xs[k][t]) so every MFMA operand fetch is a conflict-free ds_read_b32 of 32
// consecutive floats per half-wave.  The next k-tile is fetched into registers while the MFMAs
// of the current one run; the operand transform (PReLU / layer norm / norm backward) is applied to
// those registers just before they are written to LDS, so normalised activations never exist in HBM.
//
// MFMA 32x32x2 f32 lane maps (l = lane, h = l >> 5):
//   A operand: A[i = l & 31][k = h]     B operand: B[k = h][j = l & 31]
//   D: column j = l & 31, row i = (r & 3) + 8 * (r >> 2) + 4 * h for accumulator register r in [0, 16).
#include <stdlib.h>

#include <algorithm>

#include "ctn_common.h"

namespace {

constexpr int BN = 128;
constexpr int BK = 32;
constexpr int NTH = 256;

__device__ __forceinline__ f32x4 load_row4(const float* row, int col, int ncols, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec && col + 3 < ncols) {
        v = ctn_ld4(row + col);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < ncols) v[e] = row[col + e];
    }
    return v;
}

template <int BM, int WAVES_M, bool BWD>
__global__ __launch_bounds__(NTH) void gemm_nn_kernel(const ctn_gemm_nn_args a, const int tiles_m,
                                                      const int tiles_n, const int vecA) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NA = (BK * BM / 4) / NTH;   // float4 A loads per thread
    constexpr int NX = (BK * BN / 4) / NTH;   // = 4
    static_assert(NA >= 1 && TM >= 1 && TN >= 1, "tile");

    __shared__ __attribute__((aligned(16))) float As[BK * BM];
    __shared__ __attribute__((aligned(16))) float Xs[BK * BN];
    __shared__ float red[2 * (NTH / 64)];

    // XCD-aware block order: consecutive logical tiles (which share the X tile) stay on one XCD.
    const int total = gridDim.x, hb = blockIdx.x;
    const int q = total >> 3, rr = total & 7, xcd = hb & 7;
    const int lb = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (hb >> 3);
    const int tile_m = lb % tiles_m;
    const int rest = lb / tiles_m;
    const int tile_n = rest % tiles_n;
    const int b = rest / tiles_n;
    const int m_blk = tile_m * BM, t_blk = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WTM, wn0 = (wave % WAVES_N) * WTN;

    const int M = a.M, T = a.T, K = a.K, K1 = a.K1;
    const int xflags = a.xflags;
    const float alpha = a.alpha ? a.alpha[0] : 1.f;

    // this thread's X columns are the same for every k-tile
    const int xc = (tid & 31) * 4;           // column inside the tile
    const int xr0 = tid >> 5;                // first row inside the tile (+8 per j)
    const int xt = t_blk + xc;               // global frame of element 0
    Stat4 st;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        st.mean[e] = 0.f; st.rstd[e] = 1.f; st.c1[e] = 0.f; st.c2[e] = 0.f;
    }
    if (xflags & (CTN_XF_NORM | CTN_XF_NORMBWD)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tt = min(xt + e, T - 1);
            const int64_t i = (int64_t)b * a.ssb + (int64_t)tt * a.sst;
            st.mean[e] = a.mean[i];
            st.rstd[e] = a.rstd[i];
            if (BWD) {
                st.c1[e] = a.c1[i];
                st.c2[e] = a.c2[i];
            }
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA], rx[NX], rx2[BWD ? NX : 1];
    float rg[NX], rb[NX];

    auto fetch = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int idx = tid + j * NTH;
            const int k = k0 + idx / (BM / 4);
            const int m = m_blk + (idx % (BM / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < K) {
                const float* row = (k < K1) ? a.A + (int64_t)k * a.lda : a.A2 + (int64_t)(k - K1) * a.lda2;
                v = load_row4(row, m, M, vecA != 0);
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int k = k0 + xr0 + 8 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            f32x4 v2 = {0.f, 0.f, 0.f, 0.f};
            float g = 1.f, be = 0.f;
            if (k < K) {
                if (k < K1) {
                    if (xt < a.ldx) v = ctn_ld4(a.X + (int64_t)b * a.xbs + (int64_t)k * a.ldx + xt);
                } else {
                    if (xt < a.ldxb) v = ctn_ld4(a.Xb + (int64_t)b * a.xbbs + (int64_t)(k - K1) * a.ldxb + xt);
                }
                if (BWD) {
                    if (xt < a.ldx2) v2 = ctn_ld4(a.X2 + (int64_t)b * a.x2bs + (int64_t)k * a.ldx2 + xt);
                }
                if (xflags & (CTN_XF_NORM | CTN_XF_NORMBWD)) {
                    g = a.gamma[k];
                    if (!BWD) be = a.beta[k];
                }
            }
            rx[j] = v;
            if (BWD) rx2[j] = v2;
            rg[j] = g;
            rb[j] = be;
        }
    };

    auto stage = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int idx = tid + j * NTH;
            ctn_st4(&As[(idx / (BM / 4)) * BM + (idx % (BM / 4)) * 4], ra[j]);
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int kr = xr0 + 8 * j;
            f32x4 v = rx[j];
            if (k0 + kr < K) {
                if (BWD) {
                    f32x4 y = rx2[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = ctn_prelu(y[e], alpha);
                        const float du = rg[j] * v[e] * st.rstd[e] + st.c1[e] + (u - st.mean[e]) * st.c2[e];
                        v[e] = y[e] > 0.f ? du : alpha * du;
                    }
                } else {
                    if (xflags & CTN_XF_PRELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ctn_prelu(v[e], alpha);
                    }
                    if (xflags & CTN_XF_NORM) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] - st.mean[e]) * st.rstd[e] * rg[j] + rb[j];
                    }
                }
            }
            ctn_st4(&Xs[kr * BN + xc], v);
        }
    };

    const int nk = (K + BK - 1) / BK;
    fetch(0);
    for (int kt = 0; kt < nk; ++kt) {
        stage(kt);
        __syncthreads();
        if (kt + 1 < nk) fetch(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + half;
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = As[k * BM + wm0 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = Xs[k * BN + wn0 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    // Accumulator register r of tile (i, j) is output row m(i, r), frame t(j).  Every mode first gathers what it
    // needs from memory for 16 rows at once (16 independent loads in flight), then computes and stores, so no
    // load ever waits behind a store.
    const int emode = a.emode;
    const float ea = a.ealpha ? a.ealpha[0] : 1.f;
    float s1 = 0.f, s2 = 0.f;
    float cs1[TN], cs2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) cs1[j] = cs2[j] = 0.f;
    const int m_w = m_blk + wm0 + 4 * half;
    const int t_w = t_blk + wn0 + l31;

    auto rowm = [&](int i, int r) { return m_w + i * 32 + (r & 3) + 8 * (r >> 2); };

    if (emode == CTN_EPI_HEADS) {
        const int Mo = a.Mo;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float bs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rowm(i, r);
                float v = 0.f;
                if (m < M) {
                    if (m < Mo) {
                        if (a.bias) v = a.bias[m];
                    } else if (a.bias2) {
                        v = a.bias2[m - Mo];
                    }
                }
                bs[r] = v;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t_w + j * 32;
                const bool tv = t < T;
                float in[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rowm(i, r);
                    float v = 0.f;
                    if (m < M && tv) {
                        if (m < Mo) {
                            if (a.R) v = a.R[(int64_t)b * a.rbs + (int64_t)m * a.ldr + t];
                        } else if (a.accumulate) {
                            v = a.Y2[(int64_t)b * a.y2bs + (int64_t)(m - Mo) * a.ldy2 + t];
                        }
                    }
                    in[r] = v;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rowm(i, r);
                    if (m < M && tv) {
                        const float v = acc[i][j][r] + bs[r] + in[r];
                        if (m < Mo)
                            a.Y[(int64_t)b * a.ybs + (int64_t)m * a.ldy + t] = v;
                        else
                            a.Y2[(int64_t)b * a.y2bs + (int64_t)(m - Mo) * a.ldy2 + t] = v;
                    }
                }
            }
        }
    } else if (emode == CTN_EPI_PRELUBWD || emode == CTN_EPI_NORMBWD) {
        const bool nb = emode == CTN_EPI_NORMBWD;
        float em[TN], er[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int t = t_w + j * 32;
            em[j] = 0.f;
            er[j] = 1.f;
            if (nb && t < T) {
                const int64_t si = (int64_t)b * a.essb + (int64_t)t * a.esst;
                em[j] = a.emean[si];
                er[j] = a.erstd[si];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float gm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rowm(i, r);
                gm[r] = (nb && m < M) ? a.egamma[m] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t_w + j * 32;
                const bool tv = t < T;
                float e[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rowm(i, r);
                    e[r] = (m < M && tv) ? a.E[(int64_t)b * a.ebs + (int64_t)m * a.lde + t] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rowm(i, r);
                    if (m < M && tv) {
                        float v = acc[i][j][r];
                        if (nb) {
                            const float xh = (ctn_prelu(e[r], ea) - em[j]) * er[j];
                            const float g = gm[r] * v;
                            cs1[j] += g;
                            cs2[j] += g * xh;
                        } else if (!(e[r] > 0.f)) {
                            s1 += v * e[r];
                            v *= ea;
                        }
                        a.Y[(int64_t)b * a.ybs + (int64_t)m * a.ldy + t] = v;
                    }
                }
            }
        }
    } else {   // STORE, STATS, SIGMOID, ADD
        const bool has_r = (emode == CTN_EPI_ADD) && a.R != nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float bs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rowm(i, r);
                bs[r] = (a.bias && m < M) ? a.bias[m] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t_w + j * 32;
                const bool tv = t < T;
                float in[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rowm(i, r);
                    in[r] = (has_r && m < M && tv) ? a.R[(int64_t)b * a.rbs + (int64_t)m * a.ldr + t] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rowm(i, r);
                    if (m < M && tv) {
                        float v = acc[i][j][r] + bs[r] + in[r];
                        if (emode == CTN_EPI_SIGMOID) v = 1.f / (1.f + expf(-v));
                        a.Y[(int64_t)b * a.ybs + (int64_t)m * a.ldy + t] = v;
                        if (emode == CTN_EPI_STATS) {
                            const float u = ctn_prelu(v, ea);
                            cs1[j] += u;
                            cs2[j] += u * u;
                        }
                    }
                }
            }
        }
    }
    if (emode == CTN_EPI_STATS || emode == CTN_EPI_NORMBWD) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (a.ecol) {   // per-frame sums over this wave's rows; fold the two half-waves first
                const int t = t_w + j * 32;
                const float c1 = cs1[j] + __shfl_xor(cs1[j], 32);
                const float c2 = cs2[j] + __shfl_xor(cs2[j], 32);
                if (half == 0 && t < T) {
                    atomicAdd(a.ecol + (int64_t)b * a.ecol_bs + t, c1);
                    atomicAdd(a.ecol + (int64_t)b * a.ecol_bs + a.ecol_ld + t, c2);
                }
            } else {
                s1 += cs1[j];
                s2 += cs2[j];
            }
        }
    }
    const bool sums = (emode == CTN_EPI_PRELUBWD) ||
                      ((emode == CTN_EPI_STATS || emode == CTN_EPI_NORMBWD) && a.esums != nullptr);
    if (sums) {
        float v[2] = {s1, s2};
        const float tot = ctn_block_sum<2>(v, red);
        if (emode == CTN_EPI_PRELUBWD) {
            if (tid == 0 && a.esums) atomicAdd(a.esums + ctn_slot(), (double)tot);
        } else if (tid < 2) {
            atomicAdd(a.esums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + tid, (double)tot);
        }
    }
    __shared__ unsigned fin_flag;
    if ((emode == CTN_EPI_STATS || emode == CTN_EPI_NORMBWD) && a.esums != nullptr)
        ctn_finish_sums(a.fin, a.esums + (int64_t)b * CTN_STAT_SLOTS * 2, b, tiles_m * tiles_n, &fin_flag);
}

// ================================================================================================
// Fast path for 16-byte-aligned operands and M % 4 == 0 (every paper-config shape).
//
// 128 x FBN_ output tile (FBN_ = 128 or 64), BK = 16, 4 waves of 64 x FBN_/2.  Operand tiles go
// global -> LDS directly (global_load_lds_dwordx4, no staging registers) into two buffers, so the loads of
// k-tile kt+1 are in flight while the MFMAs of k-tile kt run, with one barrier per k-tile.  The operand
// transform (template flags XF, so the inner loop has no branches) is applied to the B fragment right
// after its ds_read (per-frame statistics live in registers, per-channel gamma/beta in LDS), so it rides
// in the VALU slots between MFMAs.  The accumulators leave through LDS: each thread then handles float4s of
// whole rows, which makes every epilogue load/store a coalesced 16-byte access and keeps register use low
// enough for several resident blocks per CU.  These GEMMs sit close to the roofline ridge (51 FLOP/B for
// M=512,K=128), so the stores of one block have to overlap the MFMAs of another: the narrow tile gives
// several generations of blocks per launch, which takes co-resident blocks out of lockstep.
// ================================================================================================
constexpr int FBM = 128, FBK = 16;
#ifndef CTN_NN_W3
#define CTN_NN_W3 4      // waves per SIMD asked of the compiler for the three-piece variant without operand transform: it takes
                       // 131 registers unasked, three more than four waves allow; 20 bytes of spill buy the fourth wave
                       // (-6 % per launch).  The variants with a transform lose with the same request (+6 %).
#endif

#ifndef CTN_NN_WIDE8_MIN_BLOCKS
#define CTN_NN_WIDE8_MIN_BLOCKS 256   // a launch with fewer tiles than this (the CUs of the chip) takes the next narrower tile
#endif
#ifndef CTN_NN_PLAIN_BELOW_BLOCKS
#define CTN_NN_PLAIN_BELOW_BLOCKS 768   // fewer 128 x 128 tiles than this: the in-kernel split, when the plain operand is there
#endif
#ifndef CTN_NN_W5
#define CTN_NN_W5 3      // waves per SIMD of the wave-split prepared variant: its LDS footprint allows three blocks per CU
#endif

#ifdef CTN_DBG_STAMPS
// Diagnostic build only: wave 0 of every block records shader-clock stamps of its phases.
__device__ unsigned long long g_stamps[8192 * 8];
#define CTN_STAMP(i)                                                                       \
    do {                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 8192)                                         \
            g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();                 \
    } while (0)
extern "C" int ctn_debug_read_stamps(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8192 * 8);
}
#else
#define CTN_STAMP(i)
#endif

// 4 blocks (waves per SIMD) per CU for both widths: 128 unified registers per lane, so that a launch of
// 1024 x k blocks runs in whole generations.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Eight floats -> their two leading bf16 pieces, both rounded to nearest: hi = bf16(x), lo = bf16(x - hi) (the
// remainder is exact in fp32).  hi + lo carries 17 mantissa bits of x.  Per pair of floats: two packed conversions,
// two unpacking operations and two subtractions.
__device__ __forceinline__ void ctn_split_bf16(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
    union { bf16x8 v; bf16x2 p[4]; unsigned u[4]; } h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h.p[i] = __builtin_convertvector(f32x2{x[2 * i], x[2 * i + 1]}, bf16x2);
        const float h0 = __uint_as_float(h.u[i] << 16), h1 = __uint_as_float(h.u[i] & 0xffff0000u);
        l.p[i] = __builtin_convertvector(f32x2{x[2 * i] - h0, x[2 * i + 1] - h1}, bf16x2);
    }
    hi = h.v, lo = l.v;
}

// Three pieces by truncation, hi + mid + lo = x exactly: each piece takes the next eight significant bits of the 24.
__device__ __forceinline__ void ctn_split_bf16_3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    union { bf16x8 v; unsigned u[4]; } h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned x0 = __float_as_uint(x[2 * i]), x1 = __float_as_uint(x[2 * i + 1]);
        h.u[i] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        const float r0 = x[2 * i] - __uint_as_float(x0 & 0xffff0000u), r1 = x[2 * i + 1] - __uint_as_float(x1 & 0xffff0000u);
        const unsigned ru0 = __float_as_uint(r0), ru1 = __float_as_uint(r1);
        m.u[i] = __builtin_amdgcn_perm(ru1, ru0, 0x07060302u);
        const float q0 = r0 - __uint_as_float(ru0 & 0xffff0000u), q1 = r1 - __uint_as_float(ru1 & 0xffff0000u);
        l.u[i] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
    }
    hi = h.v, mid = m.v, lo = l.v;
}

// LDS stores the compiler does not see as such.  It orders every LDS store it knows of behind ALL outstanding
// global-to-LDS DMA (s_waitcnt vmcnt(0)), whatever their addresses, which would end the prefetch the DMA exists for.
// The caller orders these stores itself: s_waitcnt lgkmcnt(0) before the barrier that publishes them.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ctn_lds_store16_untracked(void* p, u32x4 v) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)p;
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ctn_lds_store8_untracked(void* p, u32x2 v) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)p;
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// SPLIT = 2: the products run on the bf16 MFMA (16 k per instruction, 8 passes) from operands split into two bf16
// pieces, three instructions per tile and k-tile -- hi*hi + hi*lo + lo*hi, fp32 accumulation -- instead of eight f32
// MFMAs of 16 passes each.  SPLIT = 3: three pieces and the six leading products, which keeps all 24 mantissa bits.
// SPLIT = 4: the arithmetic of SPLIT = 3 with each operand element split ONCE per block instead of once per wave that
// reads it.  A arrives prepared (ctn_weight_prep: three bf16 planes in fragment order, fetched by DMA as they are); the
// fp32 X tile lands in a ring of three buffers and is transformed, split and written to LDS as fragments by all threads
// -- one (column, k-group) each -- one k-tile ahead of the MFMAs that consume it.  The MFMA phase then is six 16-byte
// LDS reads per fragment set and the matrix instructions; products and their order are those of SPLIT = 3, so the
// results are bit-identical.
// SPLIT = 5: the same arithmetic again, A prepared as for SPLIT = 4, but the fp32 X tile stays where the DMA put it and
// each wave transforms and splits ITS fragment in registers.  The waves of a block sit side by side along time (one
// 128-row x 32-column strip each at the wide tile), so no two waves split the same X element, and the weights -- the
// operand every wave needs in full -- arrive split: 8 values are split per wave and k-tile where SPLIT = 3 splits 32,
// with no LDS store and no extra barrier in the loop.
// SPLIT = 6: SPLIT = 5 with EIGHT waves side by side (512 threads, FBN_ = 256 columns): one set of A fragments per k-tile
// serves twice the matrix instructions, and two blocks of 56 KB put 16 waves on a CU where three of the four-wave blocks
// put 12; a launch of the training step's shapes is then a whole number of generations of the chip's 512 block slots.
template <int FBN_, int XF, bool BWD, int SPLIT = 0>
__global__ __launch_bounds__(SPLIT == 6 ? 2 * NTH : NTH) __attribute__((amdgpu_waves_per_eu(SPLIT == 0 ? 4 : (SPLIT == 2 ? 3 : (SPLIT == 3 && XF == 0 && !BWD ? CTN_NN_W3 : (SPLIT == 5 ? CTN_NN_W5 : (SPLIT == 6 ? 4 : 2)))), 4))) void gemm_nn_fast_kernel(const ctn_gemm_nn_args a, const int tiles_m,
                                                           const int tiles_n) {
    constexpr int NTHB = SPLIT == 6 ? 2 * NTH : NTH;      // threads of a block
    constexpr int NWV = NTHB / 64;
    static_assert(SPLIT != 6 || FBN_ == 256, "eight-wave variant: 256 columns");
    static_assert(SPLIT == 6 || FBN_ <= 128, "four-wave variants: 64 or 128 columns");
    // LDS: [2 buffers][A tile | X tile | (X2 tile)] | gamma[Kp] | beta[Kp];  epilogue staging aliases the tiles
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TILE_A = FBK * FBM, TILE_X = FBK * FBN_;
    constexpr int BUF = TILE_A + TILE_X + (BWD ? TILE_X : 0);
    constexpr bool PREPW = SPLIT == 5 || SPLIT == 6;
    constexpr bool W14 = PREPW && FBN_ >= 128;            // the waves side by side, each over all 128 rows
    constexpr int WN = PREPW ? 32 : FBN_ / 2, NJ = WN / 32;   // wave tile (32 NI) x WN = NI x NJ MFMA tiles
    constexpr int NI = W14 ? 4 : 2;
    constexpr int XLPR = FBN_ / 4;                        // lanes per X-tile row in a 1 KiB piece
    constexpr int XRP = 64 / XLPR;                        // X-tile rows per piece
    constexpr int XPW = (FBK / XRP) / NWV;                // X pieces per wave
    __shared__ float red[2 * NWV];
    __shared__ float rowc[2][FBM];

    CTN_STAMP(0);
#ifdef CTN_DBG_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_stamps[blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    const int total = gridDim.x, hb = blockIdx.x;
    const int q8 = total >> 3, rr = total & 7, xcd = hb & 7;
    const int lb = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (hb >> 3);
    const int tile_m = lb % tiles_m;
    const int rest = lb / tiles_m;
    const int tile_n = rest % tiles_n;
    const int b = rest / tiles_n;
    const int m_blk = tile_m * FBM, t_blk = tile_n * FBN_;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = W14 ? 0 : (wave >> 1) * 64, wn0 = W14 ? wave * 32 : (wave & 1) * WN;
    const int M = a.M, T = a.T, K = a.K, K1 = a.K1;
    const float alpha = a.alpha ? a.alpha[0] : 1.f;
    const int nk = (K + FBK - 1) / FBK;
    const int Kp = nk * FBK;
    constexpr bool PREP = SPLIT == 4;
    constexpr int APSZ = 3 * 2 * FBM * 4;                 // PREP: floats of one buffer of A fragments (3 planes x 16 k)
    constexpr int BPSZ = 3 * 2 * FBN_ * 4;                //       ... of X fragments
    constexpr int XLSZ = TILE_X * (BWD ? 2 : 1);          //       ... of one landing buffer (X and, NORMBWD, X2)
    constexpr int BUFW = APSZ + XLSZ;                     // PREPW: one buffer = A fragments | X tile | (X2 tile)
    float* Gs = lds + (PREP ? 2 * APSZ + 2 * BPSZ + 3 * XLSZ : (PREPW ? 2 * BUFW : 2 * BUF));
    float* Bes = Gs + Kp;
    constexpr bool has_norm = BWD || (XF & CTN_XF_NORM) != 0;
    if (has_norm) {
        for (int k = tid; k < Kp; k += NTHB) {
            Gs[k] = k < K ? a.gamma[k] : 0.f;
            if (!BWD) Bes[k] = k < K ? a.beta[k] : 0.f;      // (the norm-backward variant has no beta row)
        }
    }
    if (tid < FBM) {              // epilogue row constants of this tile: bias and (NORMBWD epilogue) gamma
        const int m = m_blk + tid;
        float bv_ = 0.f, gv_ = 0.f;
        if (m < M) {
            if (a.emode == CTN_EPI_HEADS && m >= a.Mo) {
                if (a.bias2) bv_ = a.bias2[m - a.Mo];
            } else if (a.bias) {
                bv_ = a.bias[m];
            }
            if (a.emode == CTN_EPI_NORMBWD) gv_ = a.egamma[m];
        }
        rowc[0][tid] = bv_;
        rowc[1][tid] = gv_;
    }
    // per-frame statistics of this lane's columns
    float mean[NJ], rstd[NJ], c1[NJ], c2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        mean[j] = 0.f, rstd[j] = 1.f, c1[j] = 0.f, c2[j] = 0.f;
        if (has_norm) {
            const int t = min(t_blk + wn0 + j * 32 + l31, T - 1);
            const int64_t si = (int64_t)b * a.ssb + (int64_t)t * a.sst;
            mean[j] = a.mean[si];
            rstd[j] = a.rstd[si];
            if (BWD) {
                c1[j] = a.c1[si];
                c2[j] = a.c2[si];
            }
        }
    }
    // constants of the operand transforms, arranged so that the k-loop spends the fewest vector instructions:
    //   norm:          (u - mean) * rstd * g + be          = fma(fma(u, rstd, nmr), g, be),   nmr = -mean * rstd
    //   norm backward: g * dn * rstd + c1 + (u - mean) * c2 = fma(g * dn, rstd, fma(u, c2, c1m)), c1m = c1 - mean * c2
    float nmr[NJ], c1m[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        nmr[j] = -mean[j] * rstd[j];
        c1m[j] = c1[j] - mean[j] * c2[j];
    }
    const float plim = ctn_prelu_lim(alpha);

    // Tiles are fetched with buffer_load_dwordx4 ... lds: the descriptor and the k-dependent part of the address
    // (soffset) are scalar, the per-lane part (voffset) is loop-invariant, so a k-tile's loads cost no vector
    // instructions -- vector issue slots beside the MFMAs are what limits this loop.  A lane whose float4 lies
    // outside the matrix gets a voffset beyond the descriptor's range and reads zeros.
    constexpr int RS_BYTES = 0x40000000;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ldA1 = a.lda, ldA2 = a.lda2, ldX1 = a.ldx, ldXb = a.ldxb, ldX2 = a.ldx2;
    const auto rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, RS_BYTES, 0x00020000);
    const auto rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A2 ? a.A2 : a.A), 0, RS_BYTES, 0x00020000);
    const auto rX1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (int64_t)b * a.xbs), 0, RS_BYTES, 0x00020000);
    const auto rXb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Xb ? a.Xb + (int64_t)b * a.xbbs : a.X), 0, RS_BYTES,
                                                       0x00020000);
    const auto rX2 = __builtin_amdgcn_make_buffer_rsrc((void*)(BWD ? a.X2 + (int64_t)b * a.x2bs : a.X), 0, RS_BYTES,
                                                       0x00020000);
    // an A piece covers tile rows 2p, 2p+1 (32 lanes x 16 B per row), an X piece XRP rows of FBN_ floats
    const int arow = lane >> 5, acol = (lane & 31) * 4;
    const int xrow = lane / XLPR, xcol = (lane % XLPR) * 4;
    const bool m_ok = m_blk + acol < M;
    const int tcol = t_blk + xcol;
    const int voA1 = m_ok ? (arow * ldA1 + m_blk + acol) * 4 : RS_BYTES;
    const int voA2 = m_ok ? (arow * ldA2 + m_blk + acol) * 4 : RS_BYTES;
    const int voX1 = tcol < ldX1 ? (xrow * ldX1 + tcol) * 4 : RS_BYTES;
    const int voXb = tcol < ldXb ? (xrow * ldXb + tcol) * 4 : RS_BYTES;
    const int voX2 = tcol < ldX2 ? (xrow * ldX2 + tcol) * 4 : RS_BYTES;
#define CTN_BLDS16(rs, lp, vo, so) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
    auto issue = [&](int kt, int buf) {
        float* base = lds + buf * BUF;
        const int k0 = kt * FBK;
        if (k0 < K1) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {             // A tile: 8 pieces, 2 per wave
                const int p = wv * 2 + pp;
                CTN_BLDS16(rA1, base + p * 256, voA1, (k0 + 2 * p) * ldA1 * 4);
            }
#pragma unroll
            for (int pp = 0; pp < XPW; ++pp) {           // X tile
                const int p = wv * XPW + pp;
                CTN_BLDS16(rX1, base + TILE_A + p * 256, voX1, (k0 + XRP * p) * ldX1 * 4);
            }
        } else {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int p = wv * 2 + pp;
                CTN_BLDS16(rA2, base + p * 256, voA2, (k0 - K1 + 2 * p) * ldA2 * 4);
            }
#pragma unroll
            for (int pp = 0; pp < XPW; ++pp) {
                const int p = wv * XPW + pp;
                CTN_BLDS16(rXb, base + TILE_A + p * 256, voXb, (k0 - K1 + XRP * p) * ldXb * 4);
            }
        }
        if (BWD) {
#pragma unroll
            for (int pp = 0; pp < XPW; ++pp) {
                const int p = wv * XPW + pp;
                CTN_BLDS16(rX2, base + TILE_A + TILE_X + p * 256, voX2, (k0 + XRP * p) * ldX2 * 4);
            }
        }
    };
#undef CTN_BLDS16

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (PREPW) {
        const int KG = 2 * nk, Mp = tiles_m * FBM;        // k-groups and padded row count of the prepared matrix
        const auto rAp = __builtin_amdgcn_make_buffer_rsrc((void*)a.A_prep, 0, 3 * KG * Mp * 16, 0x00020000);
        const int voAp = lane * 16;
#define CTN_BLDS16(rs, lp, vo, so) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
        auto issue_w = [&](int kt, int buf) {
            float* base = lds + buf * BUFW;
            const int k0 = kt * FBK;
#ifndef CTN_DBG_W_NO_ALOADS           // (timing builds: results are wrong)
            if (NWV == 4 || wv < 4) {                    // (uniform) eight waves: the first four fetch the A fragments
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) {         // A fragments: 12 pieces of 1 KiB (plane, k-group, row half)
                    const int q = wv * 3 + pp;
                    const int pl = q >> 2, h = (q >> 1) & 1, mh = q & 1;
                    CTN_BLDS16(rAp, base + q * 256, voAp, ((pl * KG + 2 * kt + h) * Mp + m_blk + mh * 64) * 16);
                }
            }
#endif
#ifndef CTN_DBG_W_NO_XLOADS
#pragma unroll
            for (int pp = 0; pp < XPW; ++pp) {           // X tile as it is in memory
                const int p = wv * XPW + pp;
                if (k0 < K1)
                    CTN_BLDS16(rX1, base + APSZ + p * 256, voX1, (k0 + XRP * p) * ldX1 * 4);
                else
                    CTN_BLDS16(rXb, base + APSZ + p * 256, voXb, (k0 - K1 + XRP * p) * ldXb * 4);
                if (BWD) CTN_BLDS16(rX2, base + APSZ + TILE_X + p * 256, voX2, (k0 + XRP * p) * ldX2 * 4);
            }
#endif
        };
#undef CTN_BLDS16
        static_assert(NJ == 1, "wave-split prepared variant: one X fragment per wave");
        issue_w(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();          // tile kt is complete; buffer (kt+1)&1 is free
#ifndef CTN_DBG_NO_LOADS
            if (kt + 1 < nk) issue_w(kt + 1, (kt + 1) & 1);
#endif
            const bf16x8* A8 = reinterpret_cast<const bf16x8*>(lds + (kt & 1) * BUFW);
            const float* Xs = lds + (kt & 1) * BUFW + APSZ;
            const float* X2s = Xs + TILE_X;
            const int k0 = kt * FBK;
            bf16x8 bh, bm, bl;
            {
                float bf[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = 8 * half + q;
                    float v = Xs[k * FBN_ + wn0 + l31];
                    if (BWD) {
                        const float y = X2s[k * FBN_ + wn0 + l31];
                        const float sl = y > 0.f ? 1.f : alpha;
                        v = sl * __builtin_fmaf(Gs[k0 + k] * v, rstd[0], __builtin_fmaf(sl * y, c2[0], c1m[0]));
                    }
                    if (XF & CTN_XF_PRELU) v = ctn_prelu_med(v, alpha, plim);
                    if (XF & CTN_XF_NORM) v = __builtin_fmaf(__builtin_fmaf(v, rstd[0], nmr[0]), Gs[k0 + k], Bes[k0 + k]);
                    bf[q] = v;
                }
#ifdef CTN_DBG_W_NOSPLIT              // (timing builds: results are wrong)
                union { bf16x8 v; float f[4]; } raw;
                raw.f[0] = bf[0] + bf[4], raw.f[1] = bf[1] + bf[5], raw.f[2] = bf[2] + bf[6], raw.f[3] = bf[3] + bf[7];
                bh = bm = bl = raw.v;
#else
                ctn_split_bf16_3(bf, bh, bm, bl);
#endif
            }
#pragma unroll
            for (int ip = 0; ip < NI; ip += 2) {          // two row tiles at a time: their MFMAs alternate
                bf16x8 ah[2], am[2], al[2];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int ix = half * FBM + wm0 + (ip + ii) * 32 + l31;
#ifdef CTN_DBG_W_NOAREAD             // (timing builds: results are wrong)
                    ah[ii] = am[ii] = al[ii] = bh;
                    if (ix == 1 << 30) ah[ii] = A8[ix];
#else
                    ah[ii] = A8[ix], am[ii] = A8[2 * FBM + ix], al[ii] = A8[4 * FBM + ix];
#endif
                }
#define CTN_NN_MFMA(a_, b_)                                                                                             \
    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                    \
        acc[ip + ii][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[ii], b_, acc[ip + ii][0], 0, 0, 0)
                CTN_NN_MFMA(al, bh);
                CTN_NN_MFMA(ah, bl);
                CTN_NN_MFMA(am, bm);
                CTN_NN_MFMA(am, bh);
                CTN_NN_MFMA(ah, bm);
                CTN_NN_MFMA(ah, bh);
#undef CTN_NN_MFMA
            }
        }
    } else if constexpr (PREP) {
        float* const APb = lds;
        float* const BPb = lds + 2 * APSZ;
        float* const XLb = BPb + 2 * BPSZ;
        const int KG = 2 * nk, Mp = tiles_m * FBM;        // k-groups and padded row count of the prepared matrix
        const auto rAp = __builtin_amdgcn_make_buffer_rsrc((void*)a.A_prep, 0, 3 * KG * Mp * 16, 0x00020000);
        const int voAp = lane * 16;
#define CTN_BLDS16(rs, lp, vo, so) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
        // A fragments of k-tile kt: 12 pieces of 1 KiB (plane, k-group, half of the 128 rows), three per wave
        auto issue_ap = [&](int kt, float* ap) {
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {
                const int q = wv * 3 + pp;
                const int pl = q >> 2, h = (q >> 1) & 1, mh = q & 1;
                CTN_BLDS16(rAp, ap + q * 256, voAp, ((pl * KG + 2 * kt + h) * Mp + m_blk + mh * 64) * 16);
            }
        };
        auto issue_x = [&](int kt, float* xl) {
            const int k0 = kt * FBK;
#pragma unroll
            for (int pp = 0; pp < XPW; ++pp) {
                const int p = wv * XPW + pp;
                if (k0 < K1)
                    CTN_BLDS16(rX1, xl + p * 256, voX1, (k0 + XRP * p) * ldX1 * 4);
                else
                    CTN_BLDS16(rXb, xl + p * 256, voXb, (k0 - K1 + XRP * p) * ldXb * 4);
                if (BWD) CTN_BLDS16(rX2, xl + TILE_X + p * 256, voX2, (k0 + XRP * p) * ldX2 * 4);
            }
        };
#undef CTN_BLDS16
        constexpr int NXOPS = XPW * (BWD ? 2 : 1);        // DMA instructions per wave and landing buffer
        auto wait_but_newest_x = [&](bool newest_in_flight) {   // in-order completion: all but the last NXOPS loads
            // lgkmcnt(0): the fragment stores of convert() are invisible to the compiler's own counting
            if (newest_in_flight) {
                if constexpr (NXOPS == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
                else if constexpr (NXOPS == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
        };
        static_assert(NXOPS == 1 || NXOPS == 2 || NXOPS == 4, "landing-buffer DMA count");
        // conversion: thread -> column ccol of the tile and KPT consecutive k rows starting at ckb
        constexpr int KPT = FBN_ / 16;
        const int ccol = tid % FBN_, ckb = (tid / FBN_) * KPT;
        float cmean = 0.f, crstd = 1.f, cc1 = 0.f, cc2 = 0.f;
        if (has_norm) {
            const int t = min(t_blk + ccol, T - 1);
            const int64_t si = (int64_t)b * a.ssb + (int64_t)t * a.sst;
            cmean = a.mean[si];
            crstd = a.rstd[si];
            if (BWD) {
                cc1 = a.c1[si];
                cc2 = a.c2[si];
            }
        }
        const float cnmr = -cmean * crstd, cc1m = cc1 - cmean * cc2;
        auto convert = [&](const float* xl, float* bp, int k0) {
            float f[KPT];
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const int k = ckb + q;
                float v = xl[k * FBN_ + ccol];
                if (BWD) {
                    const float y = xl[TILE_X + k * FBN_ + ccol];
                    const float sl = y > 0.f ? 1.f : alpha;
                    v = sl * __builtin_fmaf(Gs[k0 + k] * v, crstd, __builtin_fmaf(sl * y, cc2, cc1m));
                }
                if (XF & CTN_XF_PRELU) v = ctn_prelu_med(v, alpha, plim);
                if (XF & CTN_XF_NORM) v = __builtin_fmaf(__builtin_fmaf(v, crstd, cnmr), Gs[k0 + k], Bes[k0 + k]);
                f[q] = v;
            }
            unsigned ph[KPT / 2], pm[KPT / 2], pl[KPT / 2];
#pragma unroll
            for (int i = 0; i < KPT / 2; ++i) {           // the truncating three-piece split of ctn_split_bf16_3
                const unsigned x0 = __float_as_uint(f[2 * i]), x1 = __float_as_uint(f[2 * i + 1]);
                ph[i] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
                const float r0 = f[2 * i] - __uint_as_float(x0 & 0xffff0000u), r1 = f[2 * i + 1] - __uint_as_float(x1 & 0xffff0000u);
                const unsigned ru0 = __float_as_uint(r0), ru1 = __float_as_uint(r1);
                pm[i] = __builtin_amdgcn_perm(ru1, ru0, 0x07060302u);
                const float q0 = r0 - __uint_as_float(ru0 & 0xffff0000u), q1 = r1 - __uint_as_float(ru1 & 0xffff0000u);
                pl[i] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
            }
            // fragment (plane, k-group g, column) is 16 bytes = rows 8g .. 8g+7; a thread with KPT == 4 fills half of one
            unsigned* bpu = reinterpret_cast<unsigned*>(bp);
            const int g = ckb >> 3, sub = (ckb & 7) >> 1;
            if constexpr (KPT == 8) {
                ctn_lds_store16_untracked(bpu + ((0 * 2 + g) * FBN_ + ccol) * 4, u32x4{ph[0], ph[1], ph[2], ph[3]});
                ctn_lds_store16_untracked(bpu + ((1 * 2 + g) * FBN_ + ccol) * 4, u32x4{pm[0], pm[1], pm[2], pm[3]});
                ctn_lds_store16_untracked(bpu + ((2 * 2 + g) * FBN_ + ccol) * 4, u32x4{pl[0], pl[1], pl[2], pl[3]});
            } else {
                ctn_lds_store8_untracked(bpu + ((0 * 2 + g) * FBN_ + ccol) * 4 + sub, u32x2{ph[0], ph[1]});
                ctn_lds_store8_untracked(bpu + ((1 * 2 + g) * FBN_ + ccol) * 4 + sub, u32x2{pm[0], pm[1]});
                ctn_lds_store8_untracked(bpu + ((2 * 2 + g) * FBN_ + ccol) * 4 + sub, u32x2{pl[0], pl[1]});
            }
        };
        static_assert(KPT == 8 || KPT == 4, "conversion slice");
        auto mma = [&](const float* ap, const float* bp) {
            const bf16x8* A8 = reinterpret_cast<const bf16x8*>(ap);
            const bf16x8* B8 = reinterpret_cast<const bf16x8*>(bp);
            bf16x8 bh[NJ], bm[NJ], bl[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ix = half * FBN_ + wn0 + j * 32 + l31;
                bh[j] = B8[ix], bm[j] = B8[2 * FBN_ + ix], bl[j] = B8[4 * FBN_ + ix];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ix = half * FBM + wm0 + i * 32 + l31;
                const bf16x8 ah = A8[ix], am = A8[2 * FBM + ix], al = A8[4 * FBM + ix];
#define CTN_NN_MFMA(a_, b_)                                                                                             \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_[j], acc[i][j], 0, 0, 0)
                CTN_NN_MFMA(al, bh);
                CTN_NN_MFMA(ah, bl);
                CTN_NN_MFMA(am, bm);
                CTN_NN_MFMA(am, bh);
                CTN_NN_MFMA(ah, bm);
                CTN_NN_MFMA(ah, bh);
#undef CTN_NN_MFMA
            }
        };
        // Order of a wave's DMA instructions: X(0), X(1) | A(0), X(2) | A(1), X(3) | ...  Before the barrier of iteration
        // kt everything but X(kt+2) must have landed: A(kt) for the MFMAs, X(kt+1) for the conversion.
        issue_x(0, XLb);
        if (nk > 1) issue_x(1, XLb + XLSZ);
        wait_but_newest_x(nk > 1);
        __syncthreads();          // X(0) complete; also publishes the gamma / beta rows and the epilogue's row constants
        issue_ap(0, APb);
        if (nk > 2) issue_x(2, XLb + 2 * XLSZ);
        convert(XLb, BPb, 0);
        for (int kt = 0; kt < nk; ++kt) {
            // A(kt), X(kt+1) and the fragments of X(kt) are complete, everything of kt-1 is free.  The bare barrier
            // instruction: __syncthreads() carries a fence that waits for ALL global-to-LDS DMA, X(kt+2) included; the
            // waits above it are the ordering this hand-over needs, and the empty asm pins the LDS reads below it.
            wait_but_newest_x(kt + 2 < nk);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#ifndef CTN_DBG_PREP_NO_LOADS
            if (kt + 1 < nk) issue_ap(kt + 1, APb + ((kt + 1) & 1) * APSZ);
            if (kt + 3 < nk) issue_x(kt + 3, XLb + (kt % 3) * XLSZ);
#endif
#ifndef CTN_DBG_PREP_NO_CONVERT       // (timing builds: results are wrong)
            if (kt + 1 < nk) convert(XLb + ((kt + 1) % 3) * XLSZ, BPb + ((kt + 1) & 1) * BPSZ, (kt + 1) * FBK);
#endif
#ifndef CTN_DBG_PREP_NO_MMA
            mma(APb + (kt & 1) * APSZ, BPb + (kt & 1) * BPSZ);
#endif
        }
    } else {
    issue(0, 0);
    CTN_STAMP(1);
    for (int kt = 0; kt < nk; ++kt) {
        // Every wave's pieces of tile kt must have landed before ANY wave reads the tile: drain this wave's
        // LDS-DMA loads explicitly, the compiler does not always put the vmcnt wait ahead of the barrier.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();          // tile kt is complete; buffer (kt+1)&1 is free
#ifdef CTN_DBG_STAMPS
        if (kt == 0) CTN_STAMP(2);
#endif
#ifndef CTN_DBG_NO_LOADS
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
#endif
        const float* As = lds + (kt & 1) * BUF;
        const float* Xs = As + TILE_A;
        const float* X2s = Xs + TILE_X;
        const int k0 = kt * FBK;
        if constexpr (SPLIT != 0) {
            static_assert(FBK == 16, "split variant: one bf16 MFMA per 16-row k-tile");
            // lane half h owns rows 8h .. 8h+7 of the k-tile; one fragment is live in fp32 at a time
            bf16x8 bh[NJ], bm[NJ], bl[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float bf[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = 8 * half + q;
                    float v = Xs[k * FBN_ + wn0 + j * 32 + l31];
                    if (BWD) {
                        const float y = X2s[k * FBN_ + wn0 + j * 32 + l31];
                        const float sl = y > 0.f ? 1.f : alpha;
                        v = sl * __builtin_fmaf(Gs[k0 + k] * v, rstd[j], __builtin_fmaf(sl * y, c2[j], c1m[j]));
                    }
                    if (XF & CTN_XF_PRELU) v = ctn_prelu_med(v, alpha, plim);
                    if (XF & CTN_XF_NORM) v = __builtin_fmaf(__builtin_fmaf(v, rstd[j], nmr[j]), Gs[k0 + k], Bes[k0 + k]);
                    bf[q] = v;
                }
                if (SPLIT == 3) ctn_split_bf16_3(bf, bh[j], bm[j], bl[j]);
                else ctn_split_bf16(bf, bh[j], bl[j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float af[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) af[q] = As[(8 * half + q) * FBM + wm0 + i * 32 + l31];
                bf16x8 ah, am, al;
                if (SPLIT == 3) ctn_split_bf16_3(af, ah, am, al);
                else ctn_split_bf16(af, ah, al);
                // smallest products first; the tiles alternate so that consecutive instructions are independent
#define CTN_NN_MFMA(a_, b_)                                                                                             \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_[j], acc[i][j], 0, 0, 0)
                CTN_NN_MFMA(al, bh);
                CTN_NN_MFMA(ah, bl);
                if (SPLIT == 3) {
                    CTN_NN_MFMA(am, bm);
                    CTN_NN_MFMA(am, bh);
                    CTN_NN_MFMA(ah, bm);
                }
                CTN_NN_MFMA(ah, bh);
#undef CTN_NN_MFMA
            }
        } else
#pragma unroll
        for (int kk = 0; kk < FBK / 2; ++kk) {
            const int k = 2 * kk + half;
            float av[2], bv[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = As[k * FBM + wm0 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = Xs[k * FBN_ + wn0 + j * 32 + l31];
            if (BWD) {
                const float g = Gs[k0 + k];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float y = X2s[k * FBN_ + wn0 + j * 32 + l31];
                    const float sl = y > 0.f ? 1.f : alpha;           // PReLU slope at y
                    const float du = __builtin_fmaf(g * bv[j], rstd[j], __builtin_fmaf(sl * y, c2[j], c1m[j]));
                    bv[j] = sl * du;
                }
            } else {
                if (XF & CTN_XF_PRELU) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bv[j] = ctn_prelu_med(bv[j], alpha, plim);
                }
                if (XF & CTN_XF_NORM) {
                    const float g = Gs[k0 + k], be = Bes[k0 + k];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bv[j] = __builtin_fmaf(__builtin_fmaf(bv[j], rstd[j], nmr[j]), g, be);
                }
            }
#ifndef CTN_DBG_NO_MFMA
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
#else
            acc[0][0][kk] += av[0] + av[1] + bv[0] + bv[NJ - 1];
#endif
        }
    }
    }   // !PREP
    CTN_STAMP(3);
    __syncthreads();              // every wave is done with the tiles: reuse them as epilogue staging
    CTN_STAMP(4);
#ifdef CTN_DBG_NO_EPILOGUE
    {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[i][j][r];
        if (keep == 123.456f) a.Y[0] = keep;
        return;
    }
#endif

    // ------------------------------------------------------------------ epilogue through LDS
    // The barriers between staging and storing order LDS traffic only.  __syncthreads() also waits for every global
    // store of the wave to be acknowledged (its fence covers global memory), which would put one full write latency
    // between the two passes of every block; the stores of pass 0 stay in flight across these barriers instead.
#ifdef CTN_DBG_EPI_FULL_BARRIER
#define CTN_LDS_BARRIER() __syncthreads()
#else
#define CTN_LDS_BARRIER()                                   \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                       \
        asm volatile("" ::: "memory");                      \
    } while (0)
#endif
    float* stg = lds;             // 64 x FBN_ floats <= 2 * BUF
    constexpr int TPR = FBN_ / 4;                 // threads per staged row
    constexpr int RS = NTHB / TPR;                // rows per sweep of the block
    constexpr int PROWS = SPLIT == 6 ? 32 : 64;   // tile rows staged per pass
    constexpr int NPASS = FBM / PROWS;
    const int emode = a.emode;
    const float ea = a.ealpha ? a.ealpha[0] : 1.f;
    const int c4 = (tid % TPR) * 4, r0 = tid / TPR;        // this thread: columns c4..c4+3, rows r0 + RS q
    const int t = t_blk + c4;
    const bool col_ok = t < a.ldy;                          // the float4 lies inside the row
    float em[4] = {0, 0, 0, 0}, er[4] = {1, 1, 1, 1};
    if (emode == CTN_EPI_NORMBWD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t si = (int64_t)b * a.essb + (int64_t)min(t + e, T - 1) * a.esst;
            em[e] = a.emean[si];
            er[e] = a.erstd[si];
        }
    }
    float s1 = 0.f, s2 = 0.f;
    float cs1[4] = {0, 0, 0, 0}, cs2[4] = {0, 0, 0, 0};
    const bool want_stats = (emode == CTN_EPI_STATS || emode == CTN_EPI_NORMBWD);
    // Row constants come from LDS (rowc, filled in the prologue) and the optional per-element input (residual,
    // accumulate target or pre-activation) is fetched 4 rows ahead with unconditional loads: a load inside the
    // row loop would put a full vmcnt wait -- which also waits for the stores before it -- in front of every store.
    constexpr int NQ = PROWS / RS;
    const int split = (emode == CTN_EPI_HEADS) ? a.Mo : M;          // rows >= split go to Y2
    const float* lo_base = a.Y; int64_t lo_ld = a.ldy, lo_bs = a.ybs; bool lo_use = false;
    const float* hi_base = a.Y; int64_t hi_ld = a.ldy, hi_bs = a.ybs; bool hi_use = false;
    if ((emode == CTN_EPI_HEADS || emode == CTN_EPI_ADD) && a.R) {
        lo_base = a.R; lo_ld = a.ldr; lo_bs = a.rbs; lo_use = true;
    } else if (emode == CTN_EPI_PRELUBWD || emode == CTN_EPI_NORMBWD) {
        lo_base = a.E; lo_ld = a.lde; lo_bs = a.ebs; lo_use = true;
    }
    if (emode == CTN_EPI_HEADS && a.accumulate && split < M) {
        hi_base = a.Y2; hi_ld = a.ldy2; hi_bs = a.y2bs; hi_use = true;
    }
    const bool has_aux = lo_use || hi_use;
    const int tl = min(t, (int)a.ldy - 4);                          // clamped column for the unconditional loads
    // Rows >= split without an input of their own (a HEADS launch that writes instead of accumulating) must not be
    // priced in rows of Y: Y holds only `split` rows there, and with more skip than residual channels the unconditional
    // prefetch would read past the end of the last batch entry's Y.  They fetch row 0 of Y instead (value unused).
    auto aux_ptr = [&](int m) {
        const int mc = min(m, M - 1);
        if (mc < split) return lo_base + (int64_t)b * lo_bs + (int64_t)mc * lo_ld + tl;
        if (!hi_use) return (const float*)a.Y + (int64_t)b * a.ybs + tl;
        return hi_base + (int64_t)b * hi_bs + (int64_t)(mc - split) * hi_ld + tl;
    };
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        // every wave stages its MFMA row tile `pass` (staging rows 0..31: tile rows 32 pass .. of the upper wave
        // pair, 32..63: of the lower pair), so half of each wave's accumulators are dead from here on
        if constexpr (SPLIT == 6) {   // eight waves side by side: row tile `pass` of every wave's strip
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stg[((r & 3) + 8 * (r >> 2) + 4 * half) * FBN_ + wn0 + l31] = acc[pass][0][r];
        } else if constexpr (W14) {      // side-by-side waves: row tiles 2 pass and 2 pass + 1 of every wave's strip
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stg[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * FBN_ + wn0 + l31] = acc[pass * 2 + ii][0][r];
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stg[((wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * FBN_ + wn0 + j * 32 + l31] = acc[pass][j][r];
        }
        CTN_LDS_BARRIER();
        auto tile_row = [&](int srow) { return W14 ? pass * PROWS + srow : (srow >> 5) * 64 + pass * 32 + (srow & 31); };
        f32x4 aux[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) aux[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_aux) {
#pragma unroll
            for (int q = 0; q < (NQ < 4 ? NQ : 4); ++q) aux[q] = ctn_ld4(aux_ptr(m_blk + tile_row(r0 + RS * q)));
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = r0 + RS * q;
            const int trow = tile_row(row);
            const int m = m_blk + trow;
            const bool ok = col_ok && m < M;
            const bool lower = m < split;
            f32x4 v = ctn_ld4(&stg[row * FBN_ + c4]);
            f32x4 x = aux[q & 3];
            if (has_aux && q + 4 < NQ) aux[q & 3] = ctn_ld4(aux_ptr(m_blk + tile_row(r0 + RS * (q + 4))));
            if (!(lower ? lo_use : hi_use)) x = f32x4{0.f, 0.f, 0.f, 0.f};
            const float bsv = rowc[0][trow];
            float* op = lower ? a.Y + (int64_t)b * a.ybs + (int64_t)m * a.ldy + t
                              : a.Y2 + (int64_t)b * a.y2bs + (int64_t)(m - split) * a.ldy2 + t;
            if (emode == CTN_EPI_PRELUBWD || emode == CTN_EPI_NORMBWD) {
                if (emode == CTN_EPI_NORMBWD) {
                    const float gmv = rowc[1][trow];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (ok && t + e < T) {
                            const float xh = (ctn_prelu(x[e], ea) - em[e]) * er[e];
                            const float g = gmv * v[e];
                            cs1[e] += g;
                            cs2[e] += g * xh;
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!(x[e] > 0.f)) {
                            if (ok && t + e < T) s1 += v[e] * x[e];
                            v[e] *= ea;
                        }
                    }
                }
            } else {
                v += x;
                v += bsv;
                if (emode == CTN_EPI_SIGMOID) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + expf(-v[e]));
                }
                if (emode == CTN_EPI_STATS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (ok && t + e < T) {
                            const float u = ctn_prelu(v[e], ea);
                            cs1[e] += u;
                            cs2[e] += u * u;
                        }
                    }
                }
            }
            if (ok) ctn_st4(op, v);
        }
        CTN_LDS_BARRIER();
    }
    if (want_stats) {
        if (a.ecol) {            // per-frame sums: fold the RS row groups through LDS, one atomic per frame
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                stg[(r0 * 2 + 0) * FBN_ + c4 + e] = cs1[e];
                stg[(r0 * 2 + 1) * FBN_ + c4 + e] = cs2[e];
            }
            __syncthreads();
            const int col = tid % FBN_, which = tid / FBN_;
            if (which < 2) {
                float tot = 0.f;
#pragma unroll
                for (int g = 0; g < RS; ++g) tot += stg[(g * 2 + which) * FBN_ + col];
                const int tc = t_blk + col;
                if (tc < T) atomicAdd(a.ecol + (int64_t)b * a.ecol_bs + (int64_t)which * a.ecol_ld + tc, tot);
            }
        } else {
            s1 = cs1[0] + cs1[1] + cs1[2] + cs1[3];
            s2 = cs2[0] + cs2[1] + cs2[2] + cs2[3];
        }
    }
    CTN_STAMP(5);
    const bool sums = (emode == CTN_EPI_PRELUBWD) || (want_stats && a.esums != nullptr);
    if (sums) {
        float v[2] = {s1, s2};
        const float tot = ctn_block_sum<2>(v, red);
        if (emode == CTN_EPI_PRELUBWD) {
            if (tid == 0 && a.esums) atomicAdd(a.esums + ctn_slot(), (double)tot);
        } else if (tid < 2) {
            atomicAdd(a.esums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + tid, (double)tot);
        }
    }
    __shared__ unsigned fin_flag;
    if (want_stats && a.esums != nullptr)
        ctn_finish_sums(a.fin, a.esums + (int64_t)b * CTN_STAT_SLOTS * 2, b, tiles_m * tiles_n, &fin_flag);
    CTN_STAMP(6);
}

bool fast_ok(const ctn_gemm_nn_args& a) {
#ifdef CTN_DBG_NO_FAST
    return false;
#endif
    auto al = [](const void* p) { return p == nullptr || ctn_aligned16(p); };
    if (a.M <= 64 || a.M % 4 != 0) return false;
    if (a.A_prep) {
        if (!ctn_aligned16(a.A_prep) || ctn_weight_prep_bytes(a.K, a.M) >= 0x40000000) return false;
    } else {
        if (!ctn_aligned16(a.A) || a.lda % 4 != 0) return false;
        if (a.K1 < a.K && (!ctn_aligned16(a.A2) || a.lda2 % 4 != 0)) return false;
    }
    if (!ctn_aligned16(a.Y) || a.ldy % 4 != 0 || a.ybs % 4 != 0) return false;
    if (a.R && (!al(a.R) || a.ldr % 4 != 0 || a.rbs % 4 != 0 || a.ldr < a.ldy)) return false;
    if (a.Y2 && (!al(a.Y2) || a.ldy2 % 4 != 0 || a.y2bs % 4 != 0 || a.ldy2 < a.ldy)) return false;
    if (a.E && (!al(a.E) || a.lde % 4 != 0 || a.ebs % 4 != 0 || a.lde < a.ldy)) return false;
    if ((int64_t)a.K > 4096 || a.K % FBK != 0 || a.K1 % FBK != 0) return false;
    // byte offsets inside one operand (one batch item) must stay below the buffer descriptors' range
    const int64_t lim = 0x40000000 / 4 - 4096;
    if ((int64_t)a.K * a.ldx >= lim) return false;
    if (!a.A_prep && (int64_t)a.K * a.lda >= lim) return false;
    if (a.K1 < a.K && ((!a.A_prep && (int64_t)a.K * a.lda2 >= lim) || (int64_t)a.K * a.ldxb >= lim)) return false;
    if ((a.xflags & CTN_XF_NORMBWD) && (int64_t)a.K * a.ldx2 >= lim) return false;
    return true;
}

// Dynamic LDS of one block of the fast kernel: two buffers of operand tiles (three tiles with the norm-backward operand),
// the gamma / beta rows of the operand transform, and at least the epilogue's staging area.
size_t fast_lds_bytes(const ctn_gemm_nn_args& a, int width) {
    const int Kp = ctn_cdiv(a.K, FBK) * FBK;
    const bool bwd = (a.xflags & CTN_XF_NORMBWD) != 0;
    const size_t buf = (size_t)(FBK * FBM + FBK * width + (bwd ? FBK * width : 0));
    size_t floats = 2 * buf + (bwd ? 1 : 2) * (size_t)Kp;               // gamma row, and beta for the forward norm
    if (floats < (size_t)64 * width) floats = (size_t)64 * width;      // epilogue staging
    return floats * sizeof(float);
}

// Tile width: the caller's tile_width (64 or 128) if given.  Otherwise the width whose
// blocks fill whole generations of the chip's block slots best (all blocks of a launch take the same time, so a
// partly filled last generation is pure tail); the wide tile wins ties, it re-reads less.  Slots per CU: 4 by
// registers, fewer when a width's LDS footprint does not fit four times into the CU's 160 KB (the wide tile with
// the norm-backward operand: 3).
int pick_fbn(const ctn_gemm_nn_args& a) {
    if (a.tile_width == 64 || a.tile_width == 128) return a.tile_width;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    constexpr size_t LDS_PER_CU = 160 * 1024, LDS_STATIC = 1280;   // static arrays of the kernel, rounded up
    auto fill = [&](int width) {
        const int64_t blocks = (int64_t)ctn_cdiv(a.M, FBM) * a.batch * ctn_cdiv(a.T, width);
        const int per_cu = std::max((int)std::min<size_t>(4, LDS_PER_CU / (fast_lds_bytes(a, width) + LDS_STATIC)), 1);
        const int64_t slots = (int64_t)cus * per_cu;
        // fewer resident blocks also hide less latency: measured, the wide norm-backward tile at 2 blocks per CU
        // is 12 % slower than the narrow one at 4 with both filling their generations
        return (double)blocks / (double)((blocks + slots - 1) / slots * slots) * (0.5 + 0.125 * per_cu);
    };
    return fill(128) >= fill(64) - 0.02 ? 128 : 64;
}

// pieces per operand of the call's arithmetic mode: 0 (f32 MFMA), 2 or 3 (bf16 MFMA on split operands)
int gemm_split(int arith) {
    return arith == CTN_ARITH_F32 ? 0 : (arith == CTN_ARITH_BF16X2 ? 2 : 3);
}

template <int FBN_>
int launch_fast_n(const ctn_gemm_nn_args& a, hipStream_t s) {
    const int tiles_m = ctn_cdiv(a.M, FBM), tiles_n = ctn_cdiv(a.T, FBN_);
    const int64_t total = (int64_t)tiles_m * tiles_n * a.batch;
    if (total > 0x7fffffff) {
        ctn_set_error("ctn_gemm_nn: grid too large");
        return CTN_ERR_ARG;
    }
    const bool bwd = (a.xflags & CTN_XF_NORMBWD) != 0;
    const size_t shmem = fast_lds_bytes(a, FBN_);
    const dim3 grid((unsigned)total), block(NTH);
    const int split = gemm_split(a.arith);
#define CTN_NN_SPLIT_CASES(SP)                                                                                          \
    do {                                                                                                                \
        if (bwd) {                                                                                                      \
            hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, 0, true, SP>), grid, block, shmem, s, a, tiles_m, tiles_n);   \
            return CTN_OK;                                                                                              \
        }                                                                                                               \
        switch (a.xflags & (CTN_XF_PRELU | CTN_XF_NORM)) {                                                              \
            case 0:                                                                                                     \
                hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, 0, false, SP>), grid, block, shmem, s, a, tiles_m,        \
                                   tiles_n);                                                                            \
                break;                                                                                                  \
            case CTN_XF_PRELU:                                                                                          \
                hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, CTN_XF_PRELU, false, SP>), grid, block, shmem, s, a,      \
                                   tiles_m, tiles_n);                                                                   \
                break;                                                                                                  \
            case CTN_XF_NORM:                                                                                           \
                hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, CTN_XF_NORM, false, SP>), grid, block, shmem, s, a,       \
                                   tiles_m, tiles_n);                                                                   \
                break;                                                                                                  \
            default:                                                                                                    \
                hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, CTN_XF_PRELU | CTN_XF_NORM, false, SP>), grid, block,     \
                                   shmem, s, a, tiles_m, tiles_n);                                                      \
        }                                                                                                               \
        return CTN_OK;                                                                                                  \
    } while (0)
    if (split == 2) CTN_NN_SPLIT_CASES(2);
    if (split == 3) CTN_NN_SPLIT_CASES(3);
#undef CTN_NN_SPLIT_CASES
    if (bwd) {
        hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, 0, true>), grid, block, shmem, s, a, tiles_m, tiles_n);
        return CTN_OK;
    }
    switch (a.xflags & (CTN_XF_PRELU | CTN_XF_NORM)) {
        case 0:
            hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, 0, false>), grid, block, shmem, s, a, tiles_m, tiles_n);
            break;
        case CTN_XF_PRELU:
            hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, CTN_XF_PRELU, false>), grid, block, shmem, s, a, tiles_m,
                               tiles_n);
            break;
        case CTN_XF_NORM:
            hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, CTN_XF_NORM, false>), grid, block, shmem, s, a, tiles_m,
                               tiles_n);
            break;
        default:
            hipLaunchKernelGGL((gemm_nn_fast_kernel<FBN_, CTN_XF_PRELU | CTN_XF_NORM, false>), grid, block, shmem, s, a,
                               tiles_m, tiles_n);
    }
    return CTN_OK;
}

// Prepared-A variant.  LDS: two buffers of A fragments, two of X fragments, three landing buffers, gamma / beta rows.
size_t prep_lds_bytes(const ctn_gemm_nn_args& a, int width) {
    const int Kp = ctn_cdiv(a.K, FBK) * FBK;
    const bool bwd = (a.xflags & CTN_XF_NORMBWD) != 0;
    size_t floats = 2 * (size_t)(3 * 2 * FBM * 4) + 2 * (size_t)(3 * 2 * width * 4) +
                    3 * (size_t)(FBK * width) * (bwd ? 2 : 1) + (bwd ? 1 : 2) * (size_t)Kp;
    if (floats < (size_t)64 * width) floats = (size_t)64 * width;
    return floats * sizeof(float);
}

// Wave-split prepared variant.  LDS: two buffers of (A fragments | X tile | X2 tile with the norm-backward operand), the
// gamma / beta rows, and at least the epilogue's staging area.
size_t prepw_lds_bytes(const ctn_gemm_nn_args& a, int width) {
    const int Kp = ctn_cdiv(a.K, FBK) * FBK;
    const bool bwd = (a.xflags & CTN_XF_NORMBWD) != 0;
    size_t floats = 2 * ((size_t)(3 * 2 * FBM * 4) + (size_t)(FBK * width) * (bwd ? 2 : 1)) + (bwd ? 1 : 2) * (size_t)Kp;
    const size_t staging = (size_t)(width == 256 ? 32 : 64) * width;
    if (floats < staging) floats = staging;
    return floats * sizeof(float);
}

template <int FBN_, int XF, bool BWD, int SP = 4>
int launch_prep_k(const ctn_gemm_nn_args& a, hipStream_t s) {
    const int tiles_m = ctn_cdiv(a.M, FBM), tiles_n = ctn_cdiv(a.T, FBN_);
    const int64_t total = (int64_t)tiles_m * tiles_n * a.batch;
    if (total > 0x7fffffff) {
        ctn_set_error("ctn_gemm_nn: grid too large");
        return CTN_ERR_ARG;
    }
    const size_t shmem = SP >= 5 ? prepw_lds_bytes(a, FBN_) : prep_lds_bytes(a, FBN_);
    auto kern = gemm_nn_fast_kernel<FBN_, XF, BWD, SP>;
    if (shmem > 48 * 1024) {      // more dynamic LDS than a kernel gets by default: ask once per device
        static bool asked[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!asked[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    120 * 1024) != hipSuccess) {
                ctn_set_error("ctn_gemm_nn: cannot raise the dynamic LDS limit of the prepared-weight kernel");
                return CTN_ERR_LAUNCH;
            }
            asked[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(SP == 6 ? 2 * NTH : NTH), shmem, s, a, tiles_m, tiles_n);
    return CTN_OK;
}

template <int FBN_, int SP>
int launch_prep_n(const ctn_gemm_nn_args& a, hipStream_t s) {
    if (a.xflags & CTN_XF_NORMBWD) return launch_prep_k<FBN_, 0, true, SP>(a, s);
    switch (a.xflags & (CTN_XF_PRELU | CTN_XF_NORM)) {
        case 0: return launch_prep_k<FBN_, 0, false, SP>(a, s);
        case CTN_XF_PRELU: return launch_prep_k<FBN_, CTN_XF_PRELU, false, SP>(a, s);
        case CTN_XF_NORM: return launch_prep_k<FBN_, CTN_XF_NORM, false, SP>(a, s);
        default: return launch_prep_k<FBN_, CTN_XF_PRELU | CTN_XF_NORM, false, SP>(a, s);
    }
}

// Tile width of the prepared-A variants: the caller's, else 128 columns (measured on the wave-split kernel with the
// norm-backward operand: 132 us at 128 columns and two blocks per CU, 149 us at 64 columns and three).
int launch_prep(const ctn_gemm_nn_args& a, hipStream_t s) {
    // block-split kernel: the second landing tile of the norm-backward operand leaves room for one block per CU at 128
    // wave-split kernel: 128 columns unless that leaves CUs without a block (small batches), then 64
    const bool narrow = a.prep_kernel == CTN_PREP_BLOCK_SPLIT
                            ? (a.xflags & CTN_XF_NORMBWD) != 0
                            : (int64_t)ctn_cdiv(a.M, FBM) * ctn_cdiv(a.T, 128) * a.batch < CTN_NN_WIDE8_MIN_BLOCKS;
    const int width = (a.tile_width == 64 || a.tile_width == 128) ? a.tile_width : (narrow ? 64 : 128);
    if (a.prep_kernel == CTN_PREP_BLOCK_SPLIT)
        return width == 128 ? launch_prep_n<128, 4>(a, s) : launch_prep_n<64, 4>(a, s);
    // eight waves over 256 columns unless the caller names a width or a kernel; not with the norm-backward operand,
    // whose second X tile would leave room for one such block per CU
    // ... and not for launches that would leave CUs without a block of it (small batches): the four-wave kernel's tiles
    // are half as wide
    const int64_t blocks8 = (int64_t)ctn_cdiv(a.M, FBM) * ctn_cdiv(a.T, 256) * a.batch;
    const bool wide8 = a.prep_kernel == CTN_PREP_WAVE_SPLIT8 ||
                       (a.prep_kernel == CTN_PREP_DEFAULT && a.tile_width == 0 && !(a.xflags & CTN_XF_NORMBWD) &&
                        blocks8 >= CTN_NN_WIDE8_MIN_BLOCKS);
    if (wide8) return launch_prep_n<256, 6>(a, s);
    return width == 128 ? launch_prep_n<128, 5>(a, s) : launch_prep_n<64, 5>(a, s);
}

int launch_fast(const ctn_gemm_nn_args& a, hipStream_t s) {
    return pick_fbn(a) == 128 ? launch_fast_n<128>(a, s) : launch_fast_n<64>(a, s);
}

template <int BM, int WAVES_M>
int launch(const ctn_gemm_nn_args& a, int vecA, hipStream_t s) {
    const int tiles_m = ctn_cdiv(a.M, BM), tiles_n = ctn_cdiv(a.T, BN);
    const int64_t total = (int64_t)tiles_m * tiles_n * a.batch;
    if (total > 0x7fffffff) {
        ctn_set_error("ctn_gemm_nn: grid too large");
        return CTN_ERR_ARG;
    }
    if (a.xflags & CTN_XF_NORMBWD)
        hipLaunchKernelGGL((gemm_nn_kernel<BM, WAVES_M, true>), dim3((unsigned)total), dim3(NTH), 0, s, a, tiles_m,
                           tiles_n, vecA);
    else
        hipLaunchKernelGGL((gemm_nn_kernel<BM, WAVES_M, false>), dim3((unsigned)total), dim3(NTH), 0, s, a,
                           tiles_m, tiles_n, vecA);
    return CTN_OK;
}

}  // namespace

extern "C" int ctn_gemm_nn(const ctn_gemm_nn_args* args, void* stream) {
    CTN_REQUIRE(args, "ctn_gemm_nn: null args");
    const ctn_gemm_nn_args& a = *args;
    CTN_REQUIRE(a.M > 0 && a.T > 0 && a.K > 0 && a.batch > 0, "ctn_gemm_nn: empty problem");
    CTN_REQUIRE(a.arith >= CTN_ARITH_DEFAULT && a.arith <= CTN_ARITH_BF16X3, "ctn_gemm_nn: arith is one of CTN_ARITH_*");
    CTN_REQUIRE(a.tile_width == 0 || a.tile_width == 64 || a.tile_width == 128, "ctn_gemm_nn: tile_width is 0, 64 or 128");
    CTN_REQUIRE(a.prep_kernel >= CTN_PREP_DEFAULT && a.prep_kernel <= CTN_PREP_WAVE_SPLIT8,
                "ctn_gemm_nn: prep_kernel is one of CTN_PREP_*");
    CTN_REQUIRE((a.A || a.A_prep) && a.X && a.Y, "ctn_gemm_nn: null A / X / Y");
    CTN_REQUIRE(a.K1 >= 0 && a.K1 <= a.K, "ctn_gemm_nn: K1 out of range");
    CTN_REQUIRE(a.K1 == a.K || ((a.A2 || a.A_prep) && a.Xb), "ctn_gemm_nn: split K needs A2 and Xb");
    if (a.A_prep)
        CTN_REQUIRE((a.arith == CTN_ARITH_DEFAULT || a.arith == CTN_ARITH_BF16X3) && fast_ok(a),
                    "ctn_gemm_nn: a prepared A needs arith CTN_ARITH_BF16X3, M > 64, M %% 4 == 0, K %% 16 == 0, K1 %% 16 == 0, "
                    "K <= 4096 and 16-byte aligned operands with pitches that are multiples of 4");
    CTN_REQUIRE(ctn_aligned16(a.X) && a.ldx % 4 == 0 && a.xbs % 4 == 0 && a.ldx >= a.T,
                "ctn_gemm_nn: X must be 16-byte aligned, ldx %% 4 == 0, ldx >= T");
    if (a.K1 < a.K)
        CTN_REQUIRE(ctn_aligned16(a.Xb) && a.ldxb % 4 == 0 && a.xbbs % 4 == 0 && a.ldxb >= a.T,
                    "ctn_gemm_nn: Xb layout");
    CTN_REQUIRE(a.ldy >= a.T, "ctn_gemm_nn: ldy < T");
    if (a.xflags & CTN_XF_NORMBWD) {
        CTN_REQUIRE(a.X2 && ctn_aligned16(a.X2) && a.ldx2 % 4 == 0 && a.x2bs % 4 == 0 && a.ldx2 >= a.T,
                    "ctn_gemm_nn: NORMBWD needs X2");
        CTN_REQUIRE(a.mean && a.rstd && a.c1 && a.c2 && a.gamma, "ctn_gemm_nn: NORMBWD needs mean/rstd/c1/c2/gamma");
        CTN_REQUIRE(a.K1 == a.K, "ctn_gemm_nn: NORMBWD with split K is not supported");
    } else if (a.xflags & CTN_XF_NORM) {
        CTN_REQUIRE(a.mean && a.rstd && a.gamma && a.beta, "ctn_gemm_nn: NORM needs mean/rstd/gamma/beta");
        CTN_REQUIRE(a.K1 == a.K, "ctn_gemm_nn: NORM with split K is not supported");
    }
    switch (a.emode) {
        case CTN_EPI_STORE: case CTN_EPI_SIGMOID: case CTN_EPI_ADD:
            break;
        case CTN_EPI_STATS:
            CTN_REQUIRE((a.esums != nullptr) != (a.ecol != nullptr), "ctn_gemm_nn: STATS needs esums xor ecol");
            break;
        case CTN_EPI_HEADS:
            CTN_REQUIRE(a.Mo >= 0 && a.Mo <= a.M && (a.Mo == a.M || a.Y2) && a.ldy2 >= (a.Mo < a.M ? a.T : 0),
                        "ctn_gemm_nn: HEADS needs Y2");
            break;
        case CTN_EPI_PRELUBWD:
            CTN_REQUIRE(a.E && a.lde >= a.T, "ctn_gemm_nn: PRELUBWD needs E");
            break;
        case CTN_EPI_NORMBWD:
            CTN_REQUIRE(a.E && a.lde >= a.T && a.emean && a.erstd && a.egamma &&
                            ((a.esums != nullptr) != (a.ecol != nullptr)),
                        "ctn_gemm_nn: NORMBWD epilogue needs E, emean, erstd, egamma and esums xor ecol");
            break;
        default:
            CTN_REQUIRE(false, "ctn_gemm_nn: unknown emode %d", a.emode);
    }
    CTN_REQUIRE(ctn_finish_ok(a.fin) &&
                    (!a.fin.count || ((a.emode == CTN_EPI_STATS || a.emode == CTN_EPI_NORMBWD) && a.esums && !a.ecol)),
                "ctn_gemm_nn: fin needs a STATS / NORMBWD epilogue with esums, and out0, out1, n (COEFS: rstd)");
    const int vecA = ctn_aligned16(a.A) && a.lda % 4 == 0 &&
                     (a.K1 == a.K || (ctn_aligned16(a.A2) && a.lda2 % 4 == 0));
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // A launch too small to give every CU a block of the prepared-operand kernels (small batches) runs better on the
    // in-kernel split, whose narrower tiles and four blocks per CU fill the chip: taken when the caller also passed the
    // plain operand (the results are bit-identical either way).
    const bool small_launch = (int64_t)ctn_cdiv(a.M, FBM) * ctn_cdiv(a.T, 128) * a.batch < CTN_NN_PLAIN_BELOW_BLOCKS;
    const int64_t desc_lim = 0x40000000 / 4 - 4096;       // (floats inside one buffer descriptor, as in fast_ok)
    const bool plain_ok = a.A && ctn_aligned16(a.A) && a.lda % 4 == 0 && (int64_t)a.K * a.lda < desc_lim &&
                          (a.K1 == a.K || (a.A2 && ctn_aligned16(a.A2) && a.lda2 % 4 == 0 && (int64_t)a.K * a.lda2 < desc_lim));
    if (a.A_prep && !(small_launch && plain_ok && a.prep_kernel == CTN_PREP_DEFAULT && a.tile_width == 0))
        rc = launch_prep(a, s);
    else if (fast_ok(a))
        rc = launch_fast(a, s);
    else if (a.M > 64)
        rc = launch<128, 2>(a, vecA, s);
    else if (a.M > 32)
        rc = launch<64, 2>(a, vecA, s);
    else
        rc = launch<32, 1>(a, vecA, s);
    if (rc != CTN_OK) return rc;
    CTN_CHECK_LAUNCH("ctn_gemm_nn");
    return CTN_OK;
}
