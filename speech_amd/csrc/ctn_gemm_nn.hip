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
    const int emode = a.emode;
    const float ea = a.ealpha ? a.ealpha[0] : 1.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int t = t_blk + wn0 + j * 32 + l31;
        const bool tv = t < T;
        float em = 0.f, er = 1.f;
        if (emode == CTN_EPI_NORMBWD && tv) {
            const int64_t i = (int64_t)b * a.essb + (int64_t)t * a.esst;
            em = a.emean[i];
            er = a.erstd[i];
        }
        float cs1 = 0.f, cs2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_blk + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < M && tv) {
                    float v = acc[i][j][r];
                    if (emode == CTN_EPI_HEADS) {
                        if (m < a.Mo) {
                            v += a.bias ? a.bias[m] : 0.f;
                            if (a.R) v += a.R[(int64_t)b * a.rbs + (int64_t)m * a.ldr + t];
                            a.Y[(int64_t)b * a.ybs + (int64_t)m * a.ldy + t] = v;
                        } else {
                            const int m2 = m - a.Mo;
                            v += a.bias2 ? a.bias2[m2] : 0.f;
                            float* p = a.Y2 + (int64_t)b * a.y2bs + (int64_t)m2 * a.ldy2 + t;
                            if (a.accumulate) v += *p;
                            *p = v;
                        }
                    } else {
                        float* yp = a.Y + (int64_t)b * a.ybs + (int64_t)m * a.ldy + t;
                        if (emode == CTN_EPI_PRELUBWD) {
                            const float e = a.E[(int64_t)b * a.ebs + (int64_t)m * a.lde + t];
                            if (!(e > 0.f)) {
                                s1 += v * e;
                                v *= ea;
                            }
                            *yp = v;
                        } else if (emode == CTN_EPI_NORMBWD) {
                            *yp = v;
                            const float e = a.E[(int64_t)b * a.ebs + (int64_t)m * a.lde + t];
                            const float xh = (ctn_prelu(e, ea) - em) * er;
                            const float g = a.egamma[m] * v;
                            cs1 += g;
                            cs2 += g * xh;
                        } else {
                            v += a.bias ? a.bias[m] : 0.f;
                            if (emode == CTN_EPI_SIGMOID) {
                                v = 1.f / (1.f + expf(-v));
                            } else if (emode == CTN_EPI_ADD) {
                                if (a.R) v += a.R[(int64_t)b * a.rbs + (int64_t)m * a.ldr + t];
                            }
                            *yp = v;
                            if (emode == CTN_EPI_STATS) {
                                const float u = ctn_prelu(v, ea);
                                cs1 += u;
                                cs2 += u * u;
                            }
                        }
                    }
                }
            }
        }
        if (emode == CTN_EPI_STATS || emode == CTN_EPI_NORMBWD) {
            if (a.ecol) {   // per-frame sums over this wave's rows; fold the two half-waves first
                cs1 += __shfl_xor(cs1, 32);
                cs2 += __shfl_xor(cs2, 32);
                if (half == 0 && tv) {
                    atomicAdd(a.ecol + (int64_t)b * a.ecol_bs + t, cs1);
                    atomicAdd(a.ecol + (int64_t)b * a.ecol_bs + a.ecol_ld + t, cs2);
                }
            } else {
                s1 += cs1;
                s2 += cs2;
            }
        }
    }
    const bool sums = (emode == CTN_EPI_PRELUBWD) ||
                      ((emode == CTN_EPI_STATS || emode == CTN_EPI_NORMBWD) && a.esums != nullptr);
    if (sums) {
        float v[2] = {s1, s2};
        const float tot = ctn_block_sum<2>(v, red);
        if (emode == CTN_EPI_PRELUBWD) {
            if (tid == 0 && a.esums) atomicAdd(a.esums, (double)tot);
        } else if (tid < 2) {
            atomicAdd(a.esums + (int64_t)b * 2 + tid, (double)tot);
        }
    }
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
    CTN_REQUIRE(a.A && a.X && a.Y, "ctn_gemm_nn: null A / X / Y");
    CTN_REQUIRE(a.K1 >= 0 && a.K1 <= a.K, "ctn_gemm_nn: K1 out of range");
    CTN_REQUIRE(a.K1 == a.K || (a.A2 && a.Xb), "ctn_gemm_nn: split K needs A2 and Xb");
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
    const int vecA = ctn_aligned16(a.A) && a.lda % 4 == 0 &&
                     (a.K1 == a.K || (ctn_aligned16(a.A2) && a.lda2 % 4 == 0));
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (a.M > 64)
        rc = launch<128, 2>(a, vecA, s);
    else if (a.M > 32)
        rc = launch<64, 2>(a, vecA, s);
    else
        rc = launch<32, 1>(a, vecA, s);
    if (rc != CTN_OK) return rc;
    CTN_CHECK_LAUNCH("ctn_gemm_nn");
    return CTN_OK;
}
