// Weight-gradient GEMM of the pointwise convolutions on the f32 MFMA, gfx950.
//
//   C[m][n] += sum_b sum_t xformA(A)[b][m][t] * xformB(B)[b][n][t]
//
// Both operands are time-contiguous, so the contraction runs along the rows of both tiles.  A block
// owns a BM x BN output tile and one (batch, time-chunk) slice of the contraction and adds its partial
// tile to C with fp32 atomics (256 contiguous bytes per wave instruction).  LDS tiles are row-major with
// rows of BK + 4 floats: each lane fetches 4 consecutive frames of its row with one conflict-free
// ds_read_b128 and feeds them to 4 MFMAs (the two half-waves take alternate 4-frame groups, so every
// frame is summed exactly once).  The A-operand loader can apply the norm/PReLU backward transform and
// reduce the bias and PReLU-slope gradients on the way.
#include <type_traits>

#include "ctn_common.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr int NTH = 256;

template <int BM, int BN, int WAVES_M, bool BWD>
__global__ __launch_bounds__(NTH) void gemm_nt_kernel(const ctn_gemm_nt_args a, const int tiles_m,
                                                      const int tiles_n, const int nchunks) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NA = (BM * BK / 4) / NTH;   // float4 loads per thread per k-tile
    constexpr int NB = (BN * BK / 4) / NTH;
    static_assert(NA >= 1 && NB >= 1 && TM >= 1 && TN >= 1, "tile");

    __shared__ __attribute__((aligned(16))) float As[BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDK];
    __shared__ float red[(NTH / 64)];

    int lb = blockIdx.x;
    const int tile_n = lb % tiles_n; lb /= tiles_n;
    const int tile_m = lb % tiles_m; lb /= tiles_m;
    const int chunk = lb % nchunks;
    const int b = lb / nchunks;
    const int m_blk = tile_m * BM, n_blk = tile_n * BN;
    const int T = a.T;
    const int t_begin = chunk * a.kchunk, t_end = min(t_begin + a.kchunk, T);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave / WAVES_N) * WTM, wn0 = (wave % WAVES_N) * WTN;

    // loader mapping: 8 consecutive threads cover one row's 32 frames
    const int lc = (tid & 7) * 4;     // frame offset inside the k-tile
    const int lr0 = tid >> 3;         // row (+32 per j)

    const float a_alpha = a.a_alpha ? a.a_alpha[0] : 1.f;
    const float b_alpha = a.b_alpha ? a.b_alpha[0] : 1.f;
    const bool do_rowsum = (tile_n == 0) && (a.a_rowsum != nullptr || a.a_rowsum2 != nullptr);
    const bool do_dalpha = BWD && (tile_n == 0) && (a.a_dalpha != nullptr);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA], ra2[BWD ? NA : 1], rb[NB];
    float rowsum[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) rowsum[j] = 0.f;
    float dalpha = 0.f;

    auto fetch = [&](int t0) {
        const int t = t0 + lc;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int m = m_blk + lr0 + 32 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f}, v2 = {0.f, 0.f, 0.f, 0.f};
            if (m < a.M && t < t_end) {
                if (m < a.M1)
                    v = ctn_ld4(a.A + (int64_t)b * a.abs_ + (int64_t)m * a.lda + t);
                else
                    v = ctn_ld4(a.Ab + (int64_t)b * a.abbs + (int64_t)(m - a.M1) * a.ldab + t);
                if (BWD) v2 = ctn_ld4(a.A2 + (int64_t)b * a.a2bs + (int64_t)m * a.lda2 + t);
            }
            ra[j] = v;
            if (BWD) ra2[j] = v2;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n_blk + lr0 + 32 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < a.N && t < t_end) v = ctn_ld4(a.B + (int64_t)b * a.bbs + (int64_t)n * a.ldb + t);
            rb[j] = v;
        }
    };

    auto stage = [&](int t0) {
        const int t = t0 + lc;
        float mean[4], rstd[4], c1[4], c2[4];
        if (BWD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t i = (int64_t)b * a.a_ssb + (int64_t)min(t + e, T - 1) * a.a_sst;
                mean[e] = a.a_mean[i]; rstd[e] = a.a_rstd[i]; c1[e] = a.a_c1[i]; c2[e] = a.a_c2[i];
            }
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int m = m_blk + lr0 + 32 * j;
            f32x4 v = ra[j];
            if (BWD) {
                const float g = (m < a.M) ? a.a_gamma[m] : 0.f;
                const f32x4 y = ra2[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = ctn_prelu(y[e], a_alpha);
                    const float du = g * v[e] * rstd[e] + c1[e] + (u - mean[e]) * c2[e];
                    const bool valid = (m < a.M) && (t + e < t_end);
                    v[e] = valid ? (y[e] > 0.f ? du : a_alpha * du) : 0.f;
                    if (do_dalpha && valid && !(y[e] > 0.f)) dalpha += du * y[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!(t + e < t_end)) v[e] = 0.f;
            }
            rowsum[j] += v[0] + v[1] + v[2] + v[3];
            ctn_st4(&As[(lr0 + 32 * j) * LDK + lc], v);
        }
        float bmean[4], brstd[4];
        if (a.bflags & CTN_XF_NORM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t i = (int64_t)b * a.b_ssb + (int64_t)min(t + e, T - 1) * a.b_sst;
                bmean[e] = a.b_mean[i]; brstd[e] = a.b_rstd[i];
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n_blk + lr0 + 32 * j;
            f32x4 v = rb[j];
            if (n < a.N) {
                if (a.bflags & CTN_XF_PRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ctn_prelu(v[e], b_alpha);
                }
                if (a.bflags & CTN_XF_NORM) {
                    const float g = a.b_gamma[n], be = a.b_beta[n];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (v[e] - bmean[e]) * brstd[e] * g + be;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(n < a.N && t + e < t_end)) v[e] = 0.f;
            ctn_st4(&Bs[(lr0 + 32 * j) * LDK + lc], v);
        }
    };

    if (t_begin < t_end) {
        fetch(t_begin);
        for (int t0 = t_begin; t0 < t_end; t0 += BK) {
            stage(t0);
            __syncthreads();
            if (t0 + BK < t_end) fetch(t0 + BK);
#pragma unroll
            for (int qd = 0; qd < BK / 8; ++qd) {
                const int kc = 4 * (2 * qd + half);
                f32x4 av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = ctn_ld4(&As[(wm0 + i * 32 + l31) * LDK + kc]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = ctn_ld4(&Bs[(wn0 + j * 32 + l31) * LDK + kc]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    // partial tile -> C
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n_blk + wn0 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_blk + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M && n < a.N) {
                    float* p = (m < a.M1) ? a.C + (int64_t)m * a.ldc + n : a.C2 + (int64_t)(m - a.M1) * a.ldc2 + n;
                    atomicAdd(p, acc[i][j][r]);
                }
            }
        }

    if (do_rowsum) {   // 8 neighbouring lanes hold one row's partial sums
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            float s = rowsum[j];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            const int m = m_blk + lr0 + 32 * j;
            if ((tid & 7) == 0 && m < a.M) {
                if (m < a.M1) {
                    if (a.a_rowsum) atomicAdd(a.a_rowsum + m, s);
                } else if (a.a_rowsum2) {
                    atomicAdd(a.a_rowsum2 + (m - a.M1), s);
                }
            }
        }
    }
    if (BWD && a.a_dalpha != nullptr) {
        float v[1] = {do_dalpha ? dalpha : 0.f};
        const float tot = ctn_block_sum<1>(v, red);
        if (tid == 0 && do_dalpha) atomicAdd(a.a_dalpha + ctn_slot(), (double)tot);
    }
}

template <int BM, int BN, int WAVES_M>
void launch(const ctn_gemm_nt_args& a, hipStream_t s) {
    const int tiles_m = ctn_cdiv(a.M, BM), tiles_n = ctn_cdiv(a.N, BN);
    const int nchunks = ctn_cdiv(a.T, a.kchunk);
    const unsigned total = (unsigned)tiles_m * tiles_n * nchunks * a.batch;
    if (a.aflags & CTN_XF_NORMBWD)
        hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WAVES_M, true>), dim3(total), dim3(NTH), 0, s, a, tiles_m,
                           tiles_n, nchunks);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WAVES_M, false>), dim3(total), dim3(NTH), 0, s, a, tiles_m,
                           tiles_n, nchunks);
}

// ================================================================================================
// Fast path: TM x TN output tile (4 waves of TM/2 x TN/2), 16 or 32 frames per k-tile, operands fetched straight into LDS with
// buffer_load_dwordx4 ... lds (scalar descriptor and k offset, loop-invariant per-lane offset: no vector
// instructions per k-tile) into two buffers.  A DMA piece is 1 KiB (8 rows x 128 B) written lane-linearly, so the
// usual row padding is impossible; instead lane (row r, slot s) of a piece FETCHES the row's 16-byte frame
// group s ^ f(r) (f(r) = (r >> 1) & 7 for 32-frame rows), and the fragment reads undo that XOR: each lane then gets 4 consecutive frames
// of its row with one conflict-free ds_read_b128 and feeds them to 4 MFMAs (half-wave h owns frame groups
// 2c + h, so every frame is summed once).  Operand transforms, the tail mask and the bias / PReLU-slope
// reductions run on the fragments, in the shadow of the MFMAs.  Needs per-batch (not per-frame)
// statistics, i.e. gLN; cumulative-norm launches and ragged shapes use the kernel above.
// ================================================================================================

// CLN: the statistics are rows over frames (cumulative norm) instead of one value per batch element; every use
// fetches the four frames of its group with one 16-byte load from a row that stays in L2.
#ifndef CTN_NT_WKS
#define CTN_NT_WKS 3     // waves per SIMD asked for the frame-divided variant (its LDS footprint allows three blocks per CU)
#endif
typedef __bf16 ntbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ntbf16x2 __attribute__((ext_vector_type(2)));
typedef float ntf32x2 __attribute__((ext_vector_type(2)));

// Eight floats -> two bf16 pieces each, hi = bf16(x), lo = bf16(x - hi), both rounded to nearest.
__device__ __forceinline__ void nt_split_bf16(const float (&x)[8], ntbf16x8& hi, ntbf16x8& lo) {
    union { ntbf16x8 v; ntbf16x2 p[4]; unsigned u[4]; } h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h.p[i] = __builtin_convertvector(ntf32x2{x[2 * i], x[2 * i + 1]}, ntbf16x2);
        const float h0 = __uint_as_float(h.u[i] << 16), h1 = __uint_as_float(h.u[i] & 0xffff0000u);
        l.p[i] = __builtin_convertvector(ntf32x2{x[2 * i] - h0, x[2 * i + 1] - h1}, ntbf16x2);
    }
    hi = h.v, lo = l.v;
}

// Three pieces by truncation, hi + mid + lo = x exactly: each piece takes the next eight significant bits of the 24.
__device__ __forceinline__ void nt_split_bf16_3(const float (&x)[8], ntbf16x8& hi, ntbf16x8& mid, ntbf16x8& lo) {
    union { ntbf16x8 v; unsigned u[4]; } h, m, l;
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

// SPLIT = 2: products on the bf16 MFMA from operands split into two bf16 pieces (hi*hi + hi*lo + lo*hi, fp32
// accumulation), one instruction triple per 16 frames instead of eight f32 MFMAs; SPLIT = 3: three pieces and the six
// leading products (all 24 mantissa bits).  See CTN_ARITH_* in include/ctn.h.
// KS (split modes, 32-frame k-tiles): the two waves of a tile row share its COLUMNS and divide the k-tile's FRAMES
// between them (wave pair member 0 the first 16 frames, member 1 the last 16) instead of dividing the columns.  Each A
// fragment is then split by one wave instead of two -- 32 values per wave and 24 matrix instructions where the
// column-divided form splits 48 -- and both waves add their partial tiles to C, which is an atomic target anyway.
template <int TM, int TN, int FK, bool BWD, int BXF, bool CLN, int SPLIT = 0, bool KS = false>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(KS ? CTN_NT_WKS : 1, 8))) void gemm_nt_fast_kernel(
    const ctn_gemm_nt_args a, const int tiles_m, const int tiles_n, const int nchunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(!KS || (SPLIT != 0 && FK == 32 && !BWD), "frame-divided wave pairs: split modes, 32-frame k-tiles");
    constexpr int FTA = TM * FK, FTB = TN * FK;           // floats per operand tile
    constexpr int BUF = FTA + FTB + (BWD ? FTA : 0);     // A | B | (A2)
    constexpr int WTM = TM / 2, WTN = KS ? TN : TN / 2;   // wave tile WTM x WTN
    constexpr int TTM = WTM / 32, TTN = WTN / 32;         // MFMA tiles per wave
    constexpr int SPR = FK / 4;                           // 16-byte slots (frame groups) per tile row
    constexpr int RPP = 64 / SPR;                         // rows per 1 KiB DMA piece
    constexpr int SH = FK == 16 ? 2 : 1;                  // slot XOR of row r: (r >> SH) & (SPR - 1)
    constexpr int PPA = TM / RPP / 4, PPB = TN / RPP / 4; // DMA pieces per wave and operand tile
    static_assert(FK == 16 || FK == 32, "k-tile");
    __shared__ float red[(NTH / 64)];

    // Workgroups are dealt round-robin over the 8 XCDs: renumber so that one XCD gets a contiguous range of
    // logical blocks.  All output tiles of one (batch, chunk) slice -- which re-read the same operand rows --
    // then share an L2 instead of fetching those rows from HBM once per XCD.
    const int total = gridDim.x, hb = blockIdx.x;
    const int q8 = total >> 3, rr = total & 7, xcd = hb & 7;
    int lb = (xcd < rr ? xcd * (q8 + 1) : rr * (q8 + 1) + (xcd - rr) * q8) + (hb >> 3);
    const int tile_n = lb % tiles_n; lb /= tiles_n;
    const int tile_m = lb % tiles_m; lb /= tiles_m;
    const int chunk = lb % nchunks;
    const int b = lb / nchunks;
    const int m_blk = tile_m * TM, n_blk = tile_n * TN;
    const int M = a.M, N = a.N, M1 = a.M1;
    const int t_begin = chunk * a.kchunk, t_end = min(t_begin + a.kchunk, a.T);
    const int nk = (t_end - t_begin + FK - 1) / FK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm0 = (wave >> 1) * WTM, wn0 = KS ? 0 : (wave & 1) * WTN;

    constexpr int RS_BYTES = 0x40000000;
    const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A ? a.A + (int64_t)b * a.abs_ : a.Ab), 0, RS_BYTES,
                                                      0x00020000);
    const auto rAb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Ab ? a.Ab + (int64_t)b * a.abbs : a.A), 0, RS_BYTES,
                                                       0x00020000);
    const auto rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(BWD ? a.A2 + (int64_t)b * a.a2bs : a.B), 0, RS_BYTES,
                                                       0x00020000);
    const auto rB = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + (int64_t)b * a.bbs), 0, RS_BYTES, 0x00020000);
    const int lda = a.lda, ldab = a.ldab, lda2 = a.lda2, ldb = a.ldb;
    // DMA lane: row lane / SPR of the piece, LDS slot lane % SPR, fetching frame group slot ^ ((row >> SH) & (SPR-1))
    // (row = row in the tile).  A lane whose frame group starts at or beyond the row pitch gets an out-of-range
    // offset (reads zeros) on the last k-tile of a row, so no load ever leaves the buffer.
    const int prow = lane / SPR, pslot = lane % SPR;
    int pgA[PPA], pgB[PPB];
#pragma unroll
    for (int pp = 0; pp < PPA; ++pp) pgA[pp] = (pslot ^ (((RPP * (wv * PPA + pp) + prow) >> SH) & (SPR - 1))) * 4;
#pragma unroll
    for (int pp = 0; pp < PPB; ++pp) pgB[pp] = (pslot ^ (((RPP * (wv * PPB + pp) + prow) >> SH) & (SPR - 1))) * 4;
    const int ldlim = min(min(M1 > 0 ? lda : 0x7fffffff, M1 < M ? ldab : 0x7fffffff), min(BWD ? lda2 : 0x7fffffff, ldb));
#define CTN_BLDS16(rs, lp, vo, so) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lp), 16, vo, so, 0, 0)
    auto issue = [&](int t0, int buf) {
#ifdef CTN_DBG_NT_NO_LOADS
        return;
#endif
        float* base = lds + buf * BUF;
        const bool edge = t0 + FK > ldlim;               // uniform; true for at most the last k-tile of a row
        const int oob = edge ? RS_BYTES : 0;             // added to the offsets of lanes past the row pitch
#pragma unroll
        for (int pp = 0; pp < PPA; ++pp) {
            const int p = wv * PPA + pp;
            const int m = m_blk + RPP * p;
            const int g = pgA[pp];
            if (m < M) {
                if (m < M1)
                    CTN_BLDS16(rA, base + p * 256, (prow * lda + g) * 4 + (t0 + g >= lda ? oob : 0), (m * lda + t0) * 4);
                else
                    CTN_BLDS16(rAb, base + p * 256, (prow * ldab + g) * 4 + (t0 + g >= ldab ? oob : 0),
                               ((m - M1) * ldab + t0) * 4);
                if (BWD)
                    CTN_BLDS16(rA2, base + FTA + FTB + p * 256, (prow * lda2 + g) * 4 + (t0 + g >= lda2 ? oob : 0),
                               (m * lda2 + t0) * 4);
            }
        }
#pragma unroll
        for (int pp = 0; pp < PPB; ++pp) {
            const int p = wv * PPB + pp;
            const int n = n_blk + RPP * p;
            const int g = pgB[pp];
            if (n < N) CTN_BLDS16(rB, base + FTA + p * 256, (prow * ldb + g) * 4 + (t0 + g >= ldb ? oob : 0), (n * ldb + t0) * 4);
        }
    };
#undef CTN_BLDS16

    // per-lane row constants
    const float a_alpha = a.a_alpha ? a.a_alpha[0] : 1.f;
    const float b_alpha = a.b_alpha ? a.b_alpha[0] : 1.f;
    const float bplim = ctn_prelu_lim(b_alpha);
    float bg[TTN], bb[TTN];
#pragma unroll
    for (int j = 0; j < TTN; ++j) {
        bg[j] = 1.f, bb[j] = 0.f;
        const int n = n_blk + wn0 + j * 32 + l31;
        if ((BXF & CTN_XF_NORM) && n < N) {
            bg[j] = a.b_gamma[n];
            bb[j] = a.b_beta[n];
        }
    }
    constexpr bool BSTAT = CLN && (BXF & CTN_XF_NORM) != 0;
    float amean = 0.f, arstd = 1.f, ac1 = 0.f, ac2 = 0.f, bmean = 0.f, brstd = 1.f;
    if (BWD && !CLN) {
        const int64_t si = (int64_t)b * a.a_ssb;
        amean = a.a_mean[si]; arstd = a.a_rstd[si]; ac1 = a.a_c1[si]; ac2 = a.a_c2[si];
    }
    if ((BXF & CTN_XF_NORM) && !CLN) {
        const int64_t si = (int64_t)b * a.b_ssb;
        bmean = a.b_mean[si]; brstd = a.b_rstd[si];
    }
    const int tstat = ((a.T + 3) & ~3) - 4;               // last float4 of a statistics row
    const float* pam = BWD ? a.a_mean + (int64_t)b * a.a_ssb : nullptr;
    const float* par = BWD ? a.a_rstd + (int64_t)b * a.a_ssb : nullptr;
    const float* pa1 = BWD ? a.a_c1 + (int64_t)b * a.a_ssb : nullptr;
    const float* pa2 = BWD ? a.a_c2 + (int64_t)b * a.a_ssb : nullptr;
    const float* pbm = (BXF & CTN_XF_NORM) ? a.b_mean + (int64_t)b * a.b_ssb : nullptr;
    const float* pbr = (BXF & CTN_XF_NORM) ? a.b_rstd + (int64_t)b * a.b_ssb : nullptr;
    if ((BXF & CTN_XF_NORM) && !BSTAT) {
        // per-batch statistics: (u - mean) * rstd * gamma + beta as one fma per element, u * bg + bb
#pragma unroll
        for (int j = 0; j < TTN; ++j) {
            bg[j] *= brstd;
            bb[j] -= bmean * bg[j];
        }
    }
    // cLN: B statistics of this half-wave's frame groups (frame-divided pairs: of the wave's own two groups).  They are
    // requested one k-tile ahead, behind the barrier, so that they arrive under the matrix phase: fetched where they
    // are used, every k-tile waits out a round trip to L2 in front of its barrier.
    constexpr int NBS = KS ? 2 : FK / 8;
    f32x4 bmv[NBS], brv[NBS], bmn[NBS], brn[NBS];
    auto load_bstat = [&](int t0) {
        if (BSTAT) {
#pragma unroll
            for (int i = 0; i < NBS; ++i) {
                const int c = KS ? 2 * (wave & 1) + i : i;
                const int tf = min(t0 + 4 * (2 * c + half), tstat);
                bmn[i] = ctn_ld4(pbm + tf);
                brn[i] = ctn_ld4(pbr + tf);
            }
        }
    };
    auto take_bstat = [&]() {
        if (BSTAT) {
#pragma unroll
            for (int i = 0; i < NBS; ++i) bmv[i] = bmn[i], brv[i] = brn[i];
        }
    };
    // the two waves of a tile row read the same A rows: one of them reduces, and only in the first tile column
    // (frame-divided pairs read different frames of those rows: both reduce)
    const bool reducer = (tile_n == 0) && (KS || (wave & 1) == 0);
    const bool do_rowsum = reducer && (a.a_rowsum != nullptr || a.a_rowsum2 != nullptr || a.a_rowsum_b != nullptr);
    float rowsum[TTM];
#pragma unroll
    for (int i = 0; i < TTM; ++i) rowsum[i] = 0.f;
    double dalpha = 0.0;     // hundreds of signed terms per thread: fp32 here costs 1e-4 of the slope gradient

    f32x16 acc[TTM][TTN];
#pragma unroll
    for (int i = 0; i < TTM; ++i)
#pragma unroll
        for (int j = 0; j < TTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fsw = (l31 >> SH) & (SPR - 1);
    const int arow_off = (wm0 + l31) * FK, brow_off = (wn0 + l31) * FK;

    auto compute = [&](const float* As, int t0, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const float* Bs = As + FTA;
        if constexpr (SPLIT != 0) {
#pragma unroll
            for (int c2i = 0; c2i < (KS ? 1 : FK / 16); ++c2i) {   // 16 frames per bf16 MFMA: this half-wave's two frame groups
                const int c2 = KS ? (wave & 1) : c2i;
                float a8[TTM][8], b8[TTN][8];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = 2 * c2 + u;
                    const int grp = 2 * c + half;
                    const int so = ((grp ^ fsw) * 4);
                    const int tf = t0 + 4 * grp;
#pragma unroll
                    for (int i = 0; i < TTM; ++i) {
                        const f32x4 v = ctn_ld4(&As[arow_off + i * 32 * FK + so]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) a8[i][4 * u + e] = (!BWD && TAIL && !(tf + e < t_end)) ? 0.f : v[e];
                        if (!BWD && do_rowsum)
                            rowsum[i] += (a8[i][4 * u] + a8[i][4 * u + 1]) + (a8[i][4 * u + 2] + a8[i][4 * u + 3]);
                    }
#pragma unroll
                    for (int j = 0; j < TTN; ++j) {
                        f32x4 v = ctn_ld4(&Bs[brow_off + j * 32 * FK + so]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float w = v[e];
                            if (BXF & CTN_XF_PRELU) w = ctn_prelu_med(w, b_alpha, bplim);
                            if (BXF & CTN_XF_NORM)
                                w = BSTAT ? (w - bmv[KS ? u : c][e]) * brv[KS ? u : c][e] * bg[j] + bb[j]
                                          : __builtin_fmaf(w, bg[j], bb[j]);
                            if (TAIL && !(tf + e < t_end)) w = 0.f;
                            b8[j][4 * u + e] = w;
                        }
                    }
                }
                ntbf16x8 bh[TTN], bm[TTN], bl[TTN];
#pragma unroll
                for (int j = 0; j < TTN; ++j) {
#ifdef CTN_DBG_NT_NO_SPLIT      // timing only: the raw bits stand in for the pieces
                    union { float f[8]; ntbf16x8 v[2]; } u;
                    for (int q = 0; q < 8; ++q) u.f[q] = b8[j][q];
                    bh[j] = u.v[0], bl[j] = u.v[1], bm[j] = u.v[0];
#else
                    if (SPLIT == 3) nt_split_bf16_3(b8[j], bh[j], bm[j], bl[j]);
                    else nt_split_bf16(b8[j], bh[j], bl[j]);
#endif
                }
#pragma unroll
                for (int i = 0; i < TTM; ++i) {
                    ntbf16x8 ah, am, al;
#ifdef CTN_DBG_NT_NO_SPLIT
                    union { float f[8]; ntbf16x8 v[2]; } u;
                    for (int q = 0; q < 8; ++q) u.f[q] = a8[i][q];
                    ah = u.v[0], al = u.v[1], am = u.v[0];
#else
                    if (SPLIT == 3) nt_split_bf16_3(a8[i], ah, am, al);
                    else nt_split_bf16(a8[i], ah, al);
#endif
#ifdef CTN_DBG_NT_NO_MFMA       // timing only: keep the pieces alive without the matrix instructions
#pragma unroll
                    for (int j = 0; j < TTN; ++j)
                        asm volatile("" ::"v"(ah), "v"(al), "v"(am), "v"(bh[j]), "v"(bl[j]), "v"(bm[j]));
                    continue;
#endif
                    // smallest products first; the tiles alternate so that consecutive instructions are independent
#define CTN_NT_MFMA(a_, b_)                                                                                             \
    _Pragma("unroll") for (int j = 0; j < TTN; ++j)                                                                     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_[j], acc[i][j], 0, 0, 0)
                    CTN_NT_MFMA(al, bh);
                    CTN_NT_MFMA(ah, bl);
                    if (SPLIT == 3) {
                        CTN_NT_MFMA(am, bm);
                        CTN_NT_MFMA(am, bh);
                        CTN_NT_MFMA(ah, bm);
                    }
                    CTN_NT_MFMA(ah, bh);
#undef CTN_NT_MFMA
                }
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < FK / 8; ++c) {
            const int grp = 2 * c + half;                 // logical frame group of this half-wave
            const int so = ((grp ^ fsw) * 4);             // its LDS slot in this lane's rows
            f32x4 av[TTM], bv[TTN];
#pragma unroll
            for (int i = 0; i < TTM; ++i) av[i] = ctn_ld4(&As[arow_off + i * 32 * FK + so]);
#pragma unroll
            for (int j = 0; j < TTN; ++j) bv[j] = ctn_ld4(&Bs[brow_off + j * 32 * FK + so]);
            const int tf = t0 + 4 * grp;                  // first frame of the group
            if (!BWD && TAIL) {          // (BWD: the in-place transform below already zeroed the tail)
#pragma unroll
                for (int i = 0; i < TTM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!(tf + e < t_end)) av[i][e] = 0.f;
            }
            if (!BWD && do_rowsum) {
#pragma unroll
                for (int i = 0; i < TTM; ++i) rowsum[i] += (av[i][0] + av[i][1]) + (av[i][2] + av[i][3]);
            }
#pragma unroll
            for (int j = 0; j < TTN; ++j) {
                if (BXF & CTN_XF_PRELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[j][e] = ctn_prelu_med(bv[j][e], b_alpha, bplim);
                }
                if (BXF & CTN_XF_NORM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        bv[j][e] = BSTAT ? (bv[j][e] - bmv[c][e]) * brv[c][e] * bg[j] + bb[j]
                                         : __builtin_fmaf(bv[j][e], bg[j], bb[j]);
                }
                if (TAIL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!(tf + e < t_end)) bv[j][e] = 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TTM; ++i)
#pragma unroll
                    for (int j = 0; j < TTN; ++j)
#ifdef CTN_DBG_NT_NO_MFMA
                        acc[i][j][e] += av[i][e] * bv[j][e];
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
#endif
        }
    };

    // NORMBWD: the A tile is transformed ONCE per block, in place, by all threads (thread -> one float4 of one
    // row per 64 rows), before the MFMA phase: at fragment level every element would be transformed by each
    // of the waves that read it, and the ~12 vector instructions per element then outweigh the MFMAs.
    constexpr int XRP = NTH / SPR;                      // tile rows covered by one sweep of the block
    constexpr int XPT = TM / XRP;                       // float4 per thread
    const int xrow = tid / SPR, xslot = tid % SPR;
    const int xgrp = xslot ^ ((xrow >> SH) & (SPR - 1));   // logical frame group held by this LDS slot
    float xg[XPT], xsum[XPT];
    bool xv[XPT];
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
        const int m = m_blk + xrow + XRP * u;
        xv[u] = m < M;
        xg[u] = (BWD && xv[u]) ? a.a_gamma[m] * (CLN ? 1.f : arstd) : 0.f;
        xsum[u] = 0.f;
    }
    const bool xreduce = BWD && tile_n == 0;
    const float ac1m = ac1 - amean * ac2;               // gLN: c1 + (u - mean) * c2 = fma(u, c2, ac1m)
    // cLN: the four per-frame statistics of this thread's frame group, requested one k-tile before the transform that
    // uses them (same reason as the B statistics above)
    f32x4 am4, ar4, a14, a24, amn, arn, a1n, a2n;
    auto load_astat = [&](int t0) {
        if (BWD && CLN) {
            const int ts = min(t0 + 4 * xgrp, tstat);
            amn = ctn_ld4(pam + ts); arn = ctn_ld4(par + ts); a1n = ctn_ld4(pa1 + ts); a2n = ctn_ld4(pa2 + ts);
        }
    };
    auto take_astat = [&]() {
        if (BWD && CLN) am4 = amn, ar4 = arn, a14 = a1n, a24 = a2n;
    };
    auto transform = [&](float* As, int t0) {
        float* A2s = As + FTA + FTB;
        const int tf = t0 + 4 * xgrp;
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int o = (xrow + XRP * u) * FK + xslot * 4;
            f32x4 v = ctn_ld4(&As[o]);
            const f32x4 y = ctn_ld4(&A2s[o]);
            float da4 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {      // 8 vector ops per element: this loop is 20 % of the launch
                const float sl = y[e] > 0.f ? 1.f : a_alpha;          // PReLU slope at y
                const float du = CLN ? xg[u] * ar4[e] * v[e] + a14[e] + (sl * y[e] - am4[e]) * a24[e]
                                     : __builtin_fmaf(xg[u], v[e], __builtin_fmaf(sl * y[e], ac2, ac1m));   // xg = gamma * rstd
                const bool valid = xv[u] && (tf + e < t_end);
                v[e] = valid ? sl * du : 0.f;
                if (xreduce && valid && !(y[e] > 0.f)) da4 += du * y[e];
            }
            if (xreduce) dalpha += (double)da4;    // 4 terms in fp32, the hundreds of signed terms above them in fp64
            if (xreduce) xsum[u] += (v[0] + v[1]) + (v[2] + v[3]);
            ctn_st4(&As[o], v);
        }
    };

    // The ragged last k-tile (at most one per block) runs after the loop as code of its own: with both forms of
    // compute() inside the loop the compiler moves the accumulators out of and back into the AGPRs on every trip.
    const int nfull = (t_end - t_begin) / FK;             // k-tiles that lie wholly inside the chunk
    if (!BWD) {
        if (nk > 0) issue(t_begin, 0);
        load_bstat(t_begin);
        for (int kt = 0; kt < nfull; ++kt) {
            const int t0 = t_begin + kt * FK;
            // every wave's pieces of tile kt must have landed before any wave reads the tile
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            take_bstat();
            __syncthreads();
            if (kt + 1 < nk) issue(t0 + FK, (kt + 1) & 1);
            load_bstat(t0 + FK);
            compute(lds + (kt & 1) * BUF, t0, std::false_type{});
        }
        if (nfull < nk) {
            const int t0 = t_begin + nfull * FK;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            take_bstat();
            __syncthreads();
            compute(lds + (nfull & 1) * BUF, t0, std::true_type{});
        }
    } else {
        // NORMBWD: ring of three buffers.  In iteration kt the waves run the MFMAs of tile kt (transformed in the
        // previous iteration) and, in the same instruction stream, transform tile kt+1 (landed one iteration
        // ago), while the DMA of tile kt+2 is in flight: one barrier per k-tile and the transform's vector work
        // sits in the MFMAs' shadow instead of in a phase of its own.
        if (nk > 0) {
            issue(t_begin, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (nk > 1) issue(t_begin + FK, 1);
            load_astat(t_begin);
            take_astat();
            transform(lds, t_begin);
            load_astat(t_begin + FK);
        }
        load_bstat(t_begin);
        for (int kt = 0; kt < nfull; ++kt) {
            const int t0 = t_begin + kt * FK;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt+1 have landed
            take_bstat();
            take_astat();                                        // statistics of tile kt+1's frames
            __syncthreads();                                     // tile kt transformed, tile kt+1 complete, tile kt-1 free
            load_bstat(t0 + FK);
            load_astat(t0 + 2 * FK);
            if (kt + 2 < nk) issue(t0 + 2 * FK, (kt + 2) % 3);
            if (kt + 1 < nk) transform(lds + ((kt + 1) % 3) * BUF, t0 + FK);
            compute(lds + (kt % 3) * BUF, t0, std::false_type{});
        }
        if (nfull < nk) {
            const int t0 = t_begin + nfull * FK;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            take_bstat();
            __syncthreads();                                     // the ragged tile has been transformed
            compute(lds + (nfull % 3) * BUF, t0, std::true_type{});
        }
    }

    // partial tile -> C
#ifdef CTN_DBG_NT_NO_ATOMICS
    if (acc[0][0][0] == 123.456f)
#endif
#pragma unroll
    for (int i = 0; i < TTM; ++i)
#pragma unroll
        for (int j = 0; j < TTN; ++j) {
            const int n = n_blk + wn0 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_blk + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < M && n < N) {
                    float* p = (m < M1) ? a.C + (int64_t)m * a.ldc + n : a.C2 + (int64_t)(m - M1) * a.ldc2 + n;
                    atomicAdd(p, acc[i][j][r]);
                }
            }
        }

    if (BWD && xreduce && (a.a_rowsum != nullptr || a.a_rowsum2 != nullptr)) {   // SPR neighbouring threads share a row
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            float sv = xsum[u];
            sv += __shfl_xor(sv, 1);
            sv += __shfl_xor(sv, 2);
            if (SPR == 8) sv += __shfl_xor(sv, 4);
            const int m = m_blk + xrow + XRP * u;
            if (xslot == 0 && m < M) {
                if (m < M1) {
                    if (a.a_rowsum) atomicAdd(a.a_rowsum + m, sv);
                } else if (a.a_rowsum2) {
                    atomicAdd(a.a_rowsum2 + (m - M1), sv);
                }
            }
        }
    }
    if (!BWD && do_rowsum) {   // the two half-waves hold the two halves of a row's frames
#pragma unroll
        for (int i = 0; i < TTM; ++i) {
            float sv = rowsum[i];
            sv += __shfl_xor(sv, 32);
            const int m = m_blk + wm0 + i * 32 + l31;
            if (half == 0 && m < M) {
                if (m < M1) {
                    if (a.a_rowsum) atomicAdd(a.a_rowsum + m, sv);
                } else if (a.a_rowsum2) {
                    atomicAdd(a.a_rowsum2 + (m - M1), sv);
                }
                if (a.a_rowsum_b) atomicAdd(a.a_rowsum_b + (int64_t)b * M + m, sv);
            }
        }
    }
    // by-product for ctn_heads_bwd_coefs: the block's partial tile of this batch element's C, dotted with Wd
    if (!BWD && a.wdot_sums != nullptr) {
        double wd = 0.0;
#pragma unroll
        for (int i = 0; i < TTM; ++i)
#pragma unroll
            for (int j = 0; j < TTN; ++j) {
                const int n = n_blk + wn0 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m_blk + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < M && n < N) {
                        const float wv = (m < M1) ? a.wdot[(int64_t)m * a.ldwd + n] : a.wdot2[(int64_t)(m - M1) * a.ldwd2 + n];
                        wd += (double)acc[i][j][r] * (double)wv;
                    }
                }
            }
        const double tot = ctn_wave_sum(wd);
        if (lane == 0) atomicAdd(a.wdot_sums + (int64_t)b * CTN_STAT_SLOTS + ctn_slot(), tot);
    }
    if (BWD && a.a_dalpha != nullptr && tile_n == 0) {
        __shared__ double dred[NTH / 64];
        const double tot = ctn_block_sum_f64(dalpha, dred);
        if (tid == 0) atomicAdd(a.a_dalpha + ctn_slot(), tot);
    }
}

bool fast_ok(const ctn_gemm_nt_args& a) {
    if (a.M < 16 || a.N < 16) return false;
    if (a.M % 16 != 0 || a.N % 16 != 0 || a.M1 % 16 != 0) return false;    // DMA pieces are 8 or 16 whole rows
    const int64_t lim = 0x40000000 / 4 - 4096;
    if ((int64_t)a.M * a.lda >= lim || (int64_t)a.M * a.ldab >= lim || (int64_t)a.N * a.ldb >= lim) return false;
    // statistics: one value per batch element (stride 0) or a row over frames (stride 1, pitch % 4 == 0, aligned);
    // both operands must use the same kind
    auto rows_ok = [&](const float* p, int64_t sb) { return ctn_aligned16(p) && sb % 4 == 0 && sb >= ((a.T + 3) & ~3); };
    const bool a_st = (a.aflags & CTN_XF_NORMBWD) != 0, b_st = (a.bflags & CTN_XF_NORM) != 0;
    if (a_st) {
        if ((int64_t)a.M * a.lda2 >= lim) return false;
        if (a.a_sst == 1) {
            if (!rows_ok(a.a_mean, a.a_ssb) || !rows_ok(a.a_rstd, a.a_ssb) || !rows_ok(a.a_c1, a.a_ssb) ||
                !rows_ok(a.a_c2, a.a_ssb))
                return false;
        } else if (a.a_sst != 0) {
            return false;
        }
    }
    if (b_st) {
        if (a.b_sst == 1) {
            if (!rows_ok(a.b_mean, a.b_ssb) || !rows_ok(a.b_rstd, a.b_ssb)) return false;
        } else if (a.b_sst != 0) {
            return false;
        }
    }
    if (a_st && b_st && a.a_sst != a.b_sst) return false;
    return true;
}

// k-tile: 32 frames (half the barriers per MFMA) unless the NORMBWD variant's third operand tile would then cost
// a resident block per CU.
// k-tile: 32 frames (half the barriers per MFMA) for the plain variants, 16 with the NORMBWD variant's ring of
// three buffers.  Tile: 64 x 64, or 64 x 128 for NORMBWD when N allows: the transformed operand is then
// transformed once, not once per tile column, and each barrier covers twice the MFMAs.
bool wide_n(const ctn_gemm_nt_args& a) { return (a.aflags & CTN_XF_NORMBWD) && a.N % 128 == 0; }

// Plain / B-transform variants: 128 x 64 when M allows.  A transformed B value then feeds two MFMAs instead of one and
// the operand bytes drawn from L2 per MFMA drop by a quarter.
bool tall_m(const ctn_gemm_nt_args& a) { return !(a.aflags & CTN_XF_NORMBWD) && a.M % 128 == 0; }

bool stat_rows(const ctn_gemm_nt_args& a) {
    return ((a.aflags & CTN_XF_NORMBWD) && a.a_sst == 1) || ((a.bflags & CTN_XF_NORM) && a.b_sst == 1);
}

#ifndef CTN_NT_BWD_FK
#define CTN_NT_BWD_FK 16
#endif
template <int TM, int TN, bool BWD, int BXF, int FK = (BWD ? CTN_NT_BWD_FK : 32), bool KS = false>
void launch_fast_x(const ctn_gemm_nt_args& a, hipStream_t s) {
    const int tiles_m = ctn_cdiv(a.M, TM), tiles_n = ctn_cdiv(a.N, TN);
    const int nchunks = ctn_cdiv(a.T, a.kchunk);
    const unsigned total = (unsigned)tiles_m * tiles_n * nchunks * a.batch;
    const size_t tile_floats = (size_t)(TM + TN + (BWD ? TM : 0)) * FK;
    const size_t shmem = (BWD ? 3 : 2) * tile_floats * sizeof(float);
    const int split = a.arith == CTN_ARITH_F32 ? 0 : (a.arith == CTN_ARITH_BF16X2 ? 2 : 3);
    const bool rows = stat_rows(a);
#define CTN_NT_LAUNCH(CLN_, SP)                                                                                        \
    hipLaunchKernelGGL((gemm_nt_fast_kernel<TM, TN, FK, BWD, BXF, CLN_, SP, (KS && SP != 0)>), dim3(total), dim3(NTH), \
                       shmem, s, a, tiles_m, tiles_n, nchunks)
    if (rows) {
        if (split == 3) CTN_NT_LAUNCH(true, 3);
        else if (split == 2) CTN_NT_LAUNCH(true, 2);
        else CTN_NT_LAUNCH(true, 0);
    } else {
        if (split == 3) CTN_NT_LAUNCH(false, 3);
        else if (split == 2) CTN_NT_LAUNCH(false, 2);
        else CTN_NT_LAUNCH(false, 0);
    }
#undef CTN_NT_LAUNCH
}

void launch_fast(const ctn_gemm_nt_args& a, hipStream_t s) {
    const int bxf = a.bflags & (CTN_XF_PRELU | CTN_XF_NORM);
    if (a.aflags & CTN_XF_NORMBWD) {
        if (wide_n(a)) {
            if (bxf == 0) launch_fast_x<64, 128, true, 0>(a, s);
            else if (bxf == CTN_XF_PRELU) launch_fast_x<64, 128, true, CTN_XF_PRELU>(a, s);
            else if (bxf == CTN_XF_NORM) launch_fast_x<64, 128, true, CTN_XF_NORM>(a, s);
            else launch_fast_x<64, 128, true, CTN_XF_PRELU | CTN_XF_NORM>(a, s);
        } else {
            if (bxf == 0) launch_fast_x<64, 64, true, 0>(a, s);
            else if (bxf == CTN_XF_PRELU) launch_fast_x<64, 64, true, CTN_XF_PRELU>(a, s);
            else if (bxf == CTN_XF_NORM) launch_fast_x<64, 64, true, CTN_XF_NORM>(a, s);
            else launch_fast_x<64, 64, true, CTN_XF_PRELU | CTN_XF_NORM>(a, s);
        }
    } else if (tall_m(a)) {
#ifdef CTN_DBG_NT_NO_KS
        constexpr bool ks = false;
#else
        constexpr bool ks = true;
#endif
        if (bxf == 0) launch_fast_x<128, 64, false, 0, 32, ks>(a, s);
        else if (bxf == CTN_XF_PRELU) launch_fast_x<128, 64, false, CTN_XF_PRELU, 32, ks>(a, s);
        else if (bxf == CTN_XF_NORM) launch_fast_x<128, 64, false, CTN_XF_NORM, 32, ks>(a, s);
        else launch_fast_x<128, 64, false, CTN_XF_PRELU | CTN_XF_NORM, 32, ks>(a, s);
    } else {
        if (bxf == 0) launch_fast_x<64, 64, false, 0>(a, s);
        else if (bxf == CTN_XF_PRELU) launch_fast_x<64, 64, false, CTN_XF_PRELU>(a, s);
        else if (bxf == CTN_XF_NORM) launch_fast_x<64, 64, false, CTN_XF_NORM>(a, s);
        else launch_fast_x<64, 64, false, CTN_XF_PRELU | CTN_XF_NORM>(a, s);
    }
}

bool act_ok(const float* p, int64_t ld, int64_t bs, int T) {
    return p && ctn_aligned16(p) && ld % 4 == 0 && bs % 4 == 0 && ld >= T;
}

}  // namespace

// whole-output kernels for the residual layer's two weight-gradient shapes (ctn_gemm_nt_whole.hip)
int ctn_gemm_nt_whole_try(const ctn_gemm_nt_args& a, hipStream_t s, int64_t* ws_query);

extern "C" int ctn_gemm_nt(const ctn_gemm_nt_args* args, void* stream) {
    CTN_REQUIRE(args, "ctn_gemm_nt: null args");
    const ctn_gemm_nt_args& a = *args;
    CTN_REQUIRE(a.M > 0 && a.N > 0 && a.T > 0 && a.batch > 0, "ctn_gemm_nt: empty problem");
    CTN_REQUIRE(a.kchunk >= BK && a.kchunk % BK == 0, "ctn_gemm_nt: kchunk must be a positive multiple of %d", BK);
    CTN_REQUIRE(a.M1 >= 0 && a.M1 <= a.M, "ctn_gemm_nt: M1 out of range");
    CTN_REQUIRE(a.M1 == 0 || act_ok(a.A, a.lda, a.abs_, a.T), "ctn_gemm_nt: A layout");
    CTN_REQUIRE(a.M1 == a.M || act_ok(a.Ab, a.ldab, a.abbs, a.T), "ctn_gemm_nt: Ab layout");
    CTN_REQUIRE(act_ok(a.B, a.ldb, a.bbs, a.T), "ctn_gemm_nt: B layout");
    CTN_REQUIRE((a.M1 == 0 || a.C) && (a.M1 == a.M || a.C2), "ctn_gemm_nt: null C");
    if (a.aflags & CTN_XF_NORMBWD) {
        CTN_REQUIRE(act_ok(a.A2, a.lda2, a.a2bs, a.T) && a.a_mean && a.a_rstd && a.a_c1 && a.a_c2 && a.a_gamma,
                    "ctn_gemm_nt: NORMBWD needs A2, stats and gamma");
        CTN_REQUIRE(a.M1 == a.M, "ctn_gemm_nt: NORMBWD with split rows is not supported");
    }
    if (a.bflags & CTN_XF_NORM)
        CTN_REQUIRE(a.b_mean && a.b_rstd && a.b_gamma && a.b_beta, "ctn_gemm_nt: NORM needs stats, gamma, beta");
    if (a.a_rowsum_b || a.wdot_sums) {
        CTN_REQUIRE(fast_ok(a) && !(a.aflags & CTN_XF_NORMBWD), "ctn_gemm_nt: the per-batch by-products need the LDS-DMA "
                    "kernel (M, N, M1 multiples of 16) without the NORMBWD operand");
        CTN_REQUIRE(!a.wdot_sums || ((a.M1 == 0 || a.wdot) && (a.M1 == a.M || a.wdot2)), "ctn_gemm_nt: wdot_sums needs wdot / wdot2");
    }
    hipStream_t s = (hipStream_t)stream;
    if (ctn_gemm_nt_whole_try(a, s, nullptr)) {
        CTN_CHECK_LAUNCH("ctn_gemm_nt");
        return CTN_OK;
    }
    if (fast_ok(a))
        launch_fast(a, s);
    else if (a.M > 64 && a.N > 64)
        launch<128, 128, 2>(a, s);
    else if (a.M > 64)
        launch<128, 64, 2>(a, s);
    else if (a.N > 64)
        launch<64, 128, 2>(a, s);
    else
        launch<64, 64, 2>(a, s);
    CTN_CHECK_LAUNCH("ctn_gemm_nt");
    return CTN_OK;
}
