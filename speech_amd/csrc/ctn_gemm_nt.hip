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

bool act_ok(const float* p, int64_t ld, int64_t bs, int T) {
    return p && ctn_aligned16(p) && ld % 4 == 0 && bs % 4 == 0 && ld >= T;
}

}  // namespace

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
    hipStream_t s = (hipStream_t)stream;
    if (a.M > 64 && a.N > 64)
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
