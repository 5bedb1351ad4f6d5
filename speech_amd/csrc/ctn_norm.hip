// Layer-norm statistics, application and backward pieces for gLN / cLN (gfx950).
//
// Both norms are n = (u - mean) * rstd * gamma[c] + beta[c] with u = prelu(x) (alpha = 1: identity).
//   gLN: mean/rstd per sample  (nn.GroupNorm(1, C, eps), src/modules/norm.py:18): rstd = 1/sqrt(var+eps)
//   cLN: mean/rstd per frame   (src/modules/norm.py:78-90): cumulative over time, rstd = 1/(sqrt(var)+eps)
//
// Backward.  With g = gamma*dn, xhat = (u-mean)*rstd:
//   gLN:  du = rstd*(g - mean(g) - xhat*mean(g*xhat))
//            = g*rstd + c1 + (u-mean)*c2,   c1 = -rstd*S1/n,  c2 = -rstd^2*S2/n,
//         S1 = sum g, S2 = sum g*xhat over the sample, n = C*T.
//   cLN:  N_t = C*(t+1), s_t = rstd_t, q_t = sqrt(var_t) = 1/s_t - eps,
//         G1_t = sum_c g, G2_t = sum_c g*xhat (per frame),
//         dvar_t  = -G2_t * s_t / (2 q_t)                (d s/d var = -s^2/(2 q), and sum_c g*(u-mean) = G2/s)
//         dmean_t = -s_t*G1_t - 2*mean_t*dvar_t
//         du[c,tau] = g*s_tau + sum_{t>=tau} dmean_t/N_t + 2*u*sum_{t>=tau} dvar_t/N_t
//                   = g*rstd + c1 + (u-mean)*c2,  c2 = 2*Bv, c1 = A + 2*mean*Bv,
//         A_tau = sum_{t>=tau} dmean_t/N_t, Bv_tau = sum_{t>=tau} dvar_t/N_t.
#include "ctn_common.h"

namespace {

constexpr int NT = 256;      // threads per block
constexpr int CG = 16;       // channels per block
constexpr int TT = NT * 4;   // frames per block (one float4 per thread)

struct StatPtr {
    const float* mean; const float* rstd; const float* c1; const float* c2;
    int64_t sb, st;
};

__device__ __forceinline__ void load_stat4(const StatPtr& s, int b, int t, int T, bool bwd, Stat4& o) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int tt = (t + e < T) ? t + e : (T - 1);
        const int64_t i = (int64_t)b * s.sb + (int64_t)tt * s.st;
        o.mean[e] = s.mean ? s.mean[i] : 0.f;
        o.rstd[e] = s.rstd ? s.rstd[i] : 1.f;
        o.c1[e] = (bwd && s.c1) ? s.c1[i] : 0.f;
        o.c2[e] = (bwd && s.c2) ? s.c2[i] : 0.f;
    }
}

// ---------------------------------------------------------------- statistics
__global__ __launch_bounds__(NT) void norm_stats_kernel(const float* __restrict__ x, int C, int T, int64_t ld,
                                                        const float* __restrict__ alpha, double* sums,
                                                        float* col, int64_t col_ld) {
    __shared__ float red[2 * (NT / 64)];
    const int b = blockIdx.z, c0 = blockIdx.y * CG, t = blockIdx.x * TT + threadIdx.x * 4;
    const float a = alpha ? alpha[0] : 1.f;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (t < T) {
        const int cend = min(c0 + CG, C);
        for (int c = c0; c < cend; ++c) {
            f32x4 v = ctn_ld4(x + ((int64_t)b * C + c) * ld + t);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = (t + e < T) ? ctn_prelu(v[e], a) : 0.f;
                s1[e] += u;
                s2[e] += u * u;
            }
        }
    }
    if (col) {
        if (t < T) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t + e < T) {
                    atomicAdd(col + ((int64_t)b * 2 + 0) * col_ld + t + e, s1[e]);
                    atomicAdd(col + ((int64_t)b * 2 + 1) * col_ld + t + e, s2[e]);
                }
        }
    } else {
        float v[2] = {s1[0] + s1[1] + s1[2] + s1[3], s2[0] + s2[1] + s2[2] + s2[3]};
        float tot = ctn_block_sum<2>(v, red);
        if (threadIdx.x < 2) atomicAdd(sums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + threadIdx.x, (double)tot);
    }
}

__global__ void gln_finalize_kernel(const double* __restrict__ sums, int batch, double count, float eps,
                                    float* mean, float* rstd) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double S1 = 0.0, S2 = 0.0;
    for (int s = 0; s < CTN_STAT_SLOTS; ++s) {
        S1 += sums[((int64_t)b * CTN_STAT_SLOTS + s) * 2];
        S2 += sums[((int64_t)b * CTN_STAT_SLOTS + s) * 2 + 1];
    }
    const double m = S1 / count;
    double var = S2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[b] = (float)m;
    rstd[b] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ void gln_bwd_coefs_kernel(const double* __restrict__ sums, const float* __restrict__ rstd, int batch,
                                     double count, float* c1, float* c2) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double r = rstd[b];
    double S1 = 0.0, S2 = 0.0;
    for (int s = 0; s < CTN_STAT_SLOTS; ++s) {
        S1 += sums[((int64_t)b * CTN_STAT_SLOTS + s) * 2];
        S2 += sums[((int64_t)b * CTN_STAT_SLOTS + s) * 2 + 1];
    }
    c1[b] = (float)(-r * S1 / count);
    c2[b] = (float)(-r * r * S2 / count);
}

// ctn_heads_bwd_coefs.  One block per weight row m: v_m = sum_c gamma_c W[m][c] (consecutive threads read consecutive
// floats), then every batch element's S1 accumulator takes R_b[m] v_m (S1_b = sum_m R_b[m] v_m), and the row of the weight
// gradient takes beta_c sum_b R_b[m].  The block that finishes last -- a ticket taken after the block's additions have
// been performed; accumulators and ticket travel through device-scope atomics only -- turns S1 and the slots of S2 into
// c1, c2 of every batch element.
constexpr int HC_NT = 256;
__global__ __launch_bounds__(HC_NT) void heads_bwd_coefs_kernel(
    const float* __restrict__ W, int64_t ld, const float* __restrict__ W2, int64_t ld2, int M, int M1, int H,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rowsum_b,
    const double* __restrict__ wdot_sums, const float* __restrict__ rstd, int batch, double count, float* c1, float* c2,
    float* dW, int64_t lddw, float* dW2, int64_t lddw2, double* s1acc, unsigned* ticket) {
    __shared__ double dred[HC_NT / 64];
    __shared__ double vsh, dbsh;
    __shared__ unsigned last;
    const int tid = threadIdx.x, m = blockIdx.x;
    const float* wrow = (m < M1) ? W + (int64_t)m * ld : W2 + (int64_t)(m - M1) * ld2;
    double part = 0.0;
    for (int c = tid; c < H; c += HC_NT) part += (double)gamma[c] * (double)wrow[c];
    const double v = ctn_block_sum_f64(part, dred);                 // thread 0 holds it
    if (tid == 0) vsh = v;
    __syncthreads();
    double dbp = 0.0;
    for (int b = tid; b < batch; b += HC_NT) {
        const double r = rowsum_b[(int64_t)b * M + m];
        __hip_atomic_fetch_add(s1acc + b, r * vsh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dbp += r;
    }
    const double db = ctn_block_sum_f64(dbp, dred);
    if (tid == 0) dbsh = db;
    __syncthreads();
    float* grow = (m < M1) ? dW + (int64_t)m * lddw : dW2 + (int64_t)(m - M1) * lddw2;
    const float dbf = (float)dbsh;
    for (int c = tid; c < H; c += HC_NT) atomicAdd(grow + c, beta[c] * dbf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this block's additions have been performed
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)M - 1 ? 1u : 0u;
    __syncthreads();
    if (last == 0u) return;
    for (int b = tid; b < batch; b += HC_NT) {
        const double S1 = __hip_atomic_load(s1acc + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double S2 = 0.0;
        for (int sl = 0; sl < CTN_STAT_SLOTS; ++sl) S2 += wdot_sums[(int64_t)b * CTN_STAT_SLOTS + sl];
        const double r = rstd[b];
        c1[b] = (float)(-r * S1 / count);
        c2[b] = (float)(-r * r * S2 / count);
    }
}

// Block-wide exclusive scan of one double per thread (1024 threads); returns the exclusive prefix and
// writes the block total to *total.
constexpr int SCAN_NT = 1024;
__device__ __forceinline__ double block_excl_scan(double v, double* smem /*[17]*/, double* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double n = __shfl_up(inc, o);
        if (lane >= o) inc += n;
    }
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int w = 0; w < SCAN_NT / 64; ++w) {
            double tmp = smem[w];
            smem[w] = run;
            run += tmp;
        }
        smem[SCAN_NT / 64] = run;
    }
    __syncthreads();
    const double excl = smem[wave] + inc - v;
    *total = smem[SCAN_NT / 64];
    __syncthreads();
    return excl;
}

__global__ __launch_bounds__(SCAN_NT) void cln_finalize_kernel(const float* __restrict__ col, int C, int T,
                                                               int64_t col_ld, float eps, float* mean,
                                                               float* rstd, int64_t stat_ld) {
    __shared__ double sm[SCAN_NT / 64 + 1];
    const int b = blockIdx.x;
    const float* c1p = col + ((int64_t)b * 2 + 0) * col_ld;
    const float* c2p = col + ((int64_t)b * 2 + 1) * col_ld;
    const int per = (T + SCAN_NT - 1) / SCAN_NT;
    const int t0 = threadIdx.x * per, t1 = min(t0 + per, T);
    double a1 = 0.0, a2 = 0.0;
    for (int t = t0; t < t1; ++t) {
        a1 += (double)c1p[t];
        a2 += (double)c2p[t];
    }
    double tot;
    double p1 = block_excl_scan(a1, sm, &tot);
    double p2 = block_excl_scan(a2, sm, &tot);
    for (int t = t0; t < t1; ++t) {
        p1 += (double)c1p[t];
        p2 += (double)c2p[t];
        const double cnt = (double)C * (double)(t + 1);
        const double m = p1 / cnt;
        double var = p2 / cnt - m * m;
        if (var < 0.0) var = 0.0;
        mean[(int64_t)b * stat_ld + t] = (float)m;
        rstd[(int64_t)b * stat_ld + t] = (float)(1.0 / (sqrt(var) + (double)eps));
    }
}

__global__ __launch_bounds__(SCAN_NT) void cln_bwd_coefs_kernel(const float* __restrict__ col, int64_t col_ld,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, int64_t stat_ld,
                                                                int C, int T, float eps, float* c1, float* c2) {
    __shared__ double sm[SCAN_NT / 64 + 1];
    const int b = blockIdx.x;
    const float* g1p = col + ((int64_t)b * 2 + 0) * col_ld;
    const float* g2p = col + ((int64_t)b * 2 + 1) * col_ld;
    const float* mp = mean + (int64_t)b * stat_ld;
    const float* rp = rstd + (int64_t)b * stat_ld;
    const int per = (T + SCAN_NT - 1) / SCAN_NT;
    const int t0 = threadIdx.x * per, t1 = min(t0 + per, T);
    auto terms = [&](int t, double& qm, double& qv) {
        const double s = rp[t], mu = mp[t];
        const double q = 1.0 / s - (double)eps;
        const double dvar = q > 0.0 ? -(double)g2p[t] * s / (2.0 * q) : 0.0;
        const double dmean = -s * (double)g1p[t] - 2.0 * mu * dvar;
        const double cnt = (double)C * (double)(t + 1);
        qm = dmean / cnt;
        qv = dvar / cnt;
    };
    double a1 = 0.0, a2 = 0.0;
    for (int t = t0; t < t1; ++t) {
        double qm, qv;
        terms(t, qm, qv);
        a1 += qm;
        a2 += qv;
    }
    double tot1, tot2;
    double p1 = block_excl_scan(a1, sm, &tot1);
    double p2 = block_excl_scan(a2, sm, &tot2);
    for (int t = t0; t < t1; ++t) {
        double qm, qv;
        terms(t, qm, qv);
        const double A = tot1 - p1;    // suffix sum including t
        const double Bv = tot2 - p2;
        c1[(int64_t)b * stat_ld + t] = (float)(A + 2.0 * (double)mp[t] * Bv);
        c2[(int64_t)b * stat_ld + t] = (float)(2.0 * Bv);
        p1 += qm;
        p2 += qv;
    }
}

// ---------------------------------------------------------------- apply (forward)
__global__ __launch_bounds__(NT) void norm_apply_kernel(const float* __restrict__ x, int C, int T, int64_t ld,
                                                        const float* __restrict__ alpha, StatPtr sp,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* y, int64_t ldy) {
    const int b = blockIdx.z, c0 = blockIdx.y * CG, t = blockIdx.x * TT + threadIdx.x * 4;
    if (t >= T) return;
    const float a = alpha ? alpha[0] : 1.f;
    Stat4 st;
    load_stat4(sp, b, t, T, false, st);
    const int cend = min(c0 + CG, C);
    for (int c = c0; c < cend; ++c) {
        f32x4 v = ctn_ld4(x + ((int64_t)b * C + c) * ld + t);
        const float g = gamma[c], be = beta[c];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (ctn_prelu(v[e], a) - st.mean[e]) * st.rstd[e] * g + be;
        ctn_st4(y + ((int64_t)b * C + c) * ldy + t, o);
    }
}

// ---------------------------------------------------------------- backward reductions
__global__ __launch_bounds__(NT) void norm_bwd_reduce_kernel(const float* __restrict__ dn, int64_t lddn,
                                                             const float* __restrict__ x, int64_t ld, int C,
                                                             int T, const float* __restrict__ alpha, StatPtr sp,
                                                             const float* __restrict__ gamma, double* sums,
                                                             float* col, int64_t col_ld, float* dgamma,
                                                             float* dbeta) {
    __shared__ float red[(2 * CG + 2) * (NT / 64)];
    const int b = blockIdx.z, c0 = blockIdx.y * CG, t = blockIdx.x * TT + threadIdx.x * 4;
    const float a = alpha ? alpha[0] : 1.f;
    Stat4 st;
    load_stat4(sp, b, min(t, T - 1), T, false, st);
    float v[2 * CG + 2];
#pragma unroll
    for (int i = 0; i < 2 * CG + 2; ++i) v[i] = 0.f;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ci = 0; ci < CG; ++ci) {
        const int c = c0 + ci;
        if (c < C && t < T) {
            f32x4 d = ctn_ld4(dn + ((int64_t)b * C + c) * lddn + t);
            f32x4 xv = ctn_ld4(x + ((int64_t)b * C + c) * ld + t);
            const float gm = gamma[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (t + e < T) {
                    const float xh = (ctn_prelu(xv[e], a) - st.mean[e]) * st.rstd[e];
                    const float g = gm * d[e];
                    s1[e] += g;
                    s2[e] += g * xh;
                    v[2 * ci] += d[e] * xh;
                    v[2 * ci + 1] += d[e];
                }
            }
        }
    }
    if (col) {
        if (t < T) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t + e < T) {
                    atomicAdd(col + ((int64_t)b * 2 + 0) * col_ld + t + e, s1[e]);
                    atomicAdd(col + ((int64_t)b * 2 + 1) * col_ld + t + e, s2[e]);
                }
        }
    } else {
        v[2 * CG] = s1[0] + s1[1] + s1[2] + s1[3];
        v[2 * CG + 1] = s2[0] + s2[1] + s2[2] + s2[3];
    }
    float tot = ctn_block_sum<2 * CG + 2>(v, red);
    const int i = threadIdx.x;
    if (i < 2 * CG) {
        const int c = c0 + (i >> 1);
        if (c < C) atomicAdd(((i & 1) ? dbeta : dgamma) + c, tot);
    } else if (i < 2 * CG + 2 && !col) {
        atomicAdd(sums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + (i - 2 * CG), (double)tot);
    }
}

__global__ __launch_bounds__(NT) void norm_bwd_apply_kernel(const float* __restrict__ dn, int64_t lddn,
                                                            const float* __restrict__ x, int64_t ld, int C,
                                                            int T, const float* __restrict__ alpha, StatPtr sp,
                                                            const float* __restrict__ gamma, float* dx,
                                                            int64_t lddx, double* dalpha) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.z, c0 = blockIdx.y * CG, t = blockIdx.x * TT + threadIdx.x * 4;
    const float a = alpha ? alpha[0] : 1.f;
    float da = 0.f;
    if (t < T) {
        Stat4 st;
        load_stat4(sp, b, t, T, true, st);
        const int cend = min(c0 + CG, C);
        for (int c = c0; c < cend; ++c) {
            f32x4 d = ctn_ld4(dn + ((int64_t)b * C + c) * lddn + t);
            f32x4 xv = ctn_ld4(x + ((int64_t)b * C + c) * ld + t);
            const float gm = gamma[c];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = ctn_prelu(xv[e], a);
                const float du = gm * d[e] * st.rstd[e] + st.c1[e] + (u - st.mean[e]) * st.c2[e];
                o[e] = xv[e] > 0.f ? du : a * du;
                if (t + e < T && !(xv[e] > 0.f)) da += du * xv[e];
            }
            ctn_st4(dx + ((int64_t)b * C + c) * lddx + t, o);
        }
    }
    if (dalpha) {
        float v[1] = {da};
        float tot = ctn_block_sum<1>(v, red);
        if (threadIdx.x == 0) atomicAdd(dalpha + ctn_slot(), (double)tot);
    }
}

bool ok_act(const void* p, int64_t ld) { return p && ctn_aligned16(p) && (ld % 4) == 0; }

}  // namespace

extern "C" {

int ctn_norm_stats(const float* x, int batch, int C, int T, int64_t ld, const float* alpha, double* sums,
                   float* col, int64_t col_ld, void* stream) {
    CTN_REQUIRE(ok_act(x, ld) && ld >= T, "ctn_norm_stats: x must be 16-byte aligned with ld %% 4 == 0, ld >= T");
    CTN_REQUIRE(batch > 0 && C > 0 && T > 0, "ctn_norm_stats: empty shape");
    CTN_REQUIRE((sums != nullptr) != (col != nullptr), "ctn_norm_stats: exactly one of sums / col");
    CTN_REQUIRE(!col || col_ld >= T, "ctn_norm_stats: col_ld < T");
    dim3 grid(ctn_cdiv(T, TT), ctn_cdiv(C, CG), batch);
    hipLaunchKernelGGL(norm_stats_kernel, grid, dim3(NT), 0, (hipStream_t)stream, x, C, T, ld, alpha, sums, col,
                       col_ld);
    CTN_CHECK_LAUNCH("ctn_norm_stats");
    return CTN_OK;
}

int ctn_gln_finalize(const double* sums, int batch, double count, float eps, float* mean, float* rstd,
                     void* stream) {
    CTN_REQUIRE(sums && mean && rstd && batch > 0 && count > 0, "ctn_gln_finalize: bad arguments");
    hipLaunchKernelGGL(gln_finalize_kernel, dim3(ctn_cdiv(batch, 64)), dim3(64), 0, (hipStream_t)stream, sums,
                       batch, count, eps, mean, rstd);
    CTN_CHECK_LAUNCH("ctn_gln_finalize");
    return CTN_OK;
}

int ctn_cln_finalize(const float* col, int batch, int C, int T, int64_t col_ld, float eps, float* mean,
                     float* rstd, int64_t stat_ld, void* stream) {
    CTN_REQUIRE(col && mean && rstd && batch > 0 && C > 0 && T > 0 && col_ld >= T && stat_ld >= T,
                "ctn_cln_finalize: bad arguments");
    hipLaunchKernelGGL(cln_finalize_kernel, dim3(batch), dim3(SCAN_NT), 0, (hipStream_t)stream, col, C, T, col_ld,
                       eps, mean, rstd, stat_ld);
    CTN_CHECK_LAUNCH("ctn_cln_finalize");
    return CTN_OK;
}

int ctn_norm_apply(const float* x, int batch, int C, int T, int64_t ld, const float* alpha, const float* mean,
                   const float* rstd, int64_t ssb, int64_t sst, const float* gamma, const float* beta, float* y,
                   int64_t ldy, void* stream) {
    CTN_REQUIRE(ok_act(x, ld) && ok_act(y, ldy) && ld >= T && ldy >= T, "ctn_norm_apply: bad x / y layout");
    CTN_REQUIRE(mean && rstd && gamma && beta && batch > 0 && C > 0 && T > 0, "ctn_norm_apply: bad arguments");
    StatPtr sp{mean, rstd, nullptr, nullptr, ssb, sst};
    dim3 grid(ctn_cdiv(T, TT), ctn_cdiv(C, CG), batch);
    hipLaunchKernelGGL(norm_apply_kernel, grid, dim3(NT), 0, (hipStream_t)stream, x, C, T, ld, alpha, sp, gamma,
                       beta, y, ldy);
    CTN_CHECK_LAUNCH("ctn_norm_apply");
    return CTN_OK;
}

int ctn_norm_bwd_reduce(const float* dn, int64_t lddn, const float* x, int64_t ld, int batch, int C, int T,
                        const float* alpha, const float* mean, const float* rstd, int64_t ssb, int64_t sst,
                        const float* gamma, double* sums, float* col, int64_t col_ld, float* dgamma,
                        float* dbeta, void* stream) {
    CTN_REQUIRE(ok_act(dn, lddn) && ok_act(x, ld) && ld >= T && lddn >= T, "ctn_norm_bwd_reduce: bad layout");
    CTN_REQUIRE(mean && rstd && gamma && dgamma && dbeta && batch > 0 && C > 0 && T > 0,
                "ctn_norm_bwd_reduce: bad arguments");
    CTN_REQUIRE((sums != nullptr) != (col != nullptr), "ctn_norm_bwd_reduce: exactly one of sums / col");
    StatPtr sp{mean, rstd, nullptr, nullptr, ssb, sst};
    dim3 grid(ctn_cdiv(T, TT), ctn_cdiv(C, CG), batch);
    hipLaunchKernelGGL(norm_bwd_reduce_kernel, grid, dim3(NT), 0, (hipStream_t)stream, dn, lddn, x, ld, C, T,
                       alpha, sp, gamma, sums, col, col_ld, dgamma, dbeta);
    CTN_CHECK_LAUNCH("ctn_norm_bwd_reduce");
    return CTN_OK;
}

int ctn_gln_bwd_coefs(const double* sums, const float* rstd, int batch, double count, float* c1, float* c2,
                      void* stream) {
    CTN_REQUIRE(sums && rstd && c1 && c2 && batch > 0 && count > 0, "ctn_gln_bwd_coefs: bad arguments");
    hipLaunchKernelGGL(gln_bwd_coefs_kernel, dim3(ctn_cdiv(batch, 64)), dim3(64), 0, (hipStream_t)stream, sums,
                       rstd, batch, count, c1, c2);
    CTN_CHECK_LAUNCH("ctn_gln_bwd_coefs");
    return CTN_OK;
}

int ctn_heads_bwd_coefs(const float* W, int64_t ld, const float* W2, int64_t ld2, int M, int M1, int H,
                        const float* gamma, const float* beta, const float* rowsum_b, const double* wdot_sums,
                        const float* rstd, int batch, double count, float* c1, float* c2, float* dW, int64_t lddw,
                        float* dW2, int64_t lddw2, double* s1acc, unsigned* ticket, void* stream) {
    CTN_REQUIRE(M > 0 && H > 0 && batch > 0 && M1 >= 0 && M1 <= M && count > 0, "ctn_heads_bwd_coefs: bad shape");
    CTN_REQUIRE(s1acc && ticket, "ctn_heads_bwd_coefs: needs the zeroed work area (s1acc[batch] doubles, one ticket word)");
    CTN_REQUIRE((M1 == 0 || (W && dW && ld >= H && lddw >= H)) && (M1 == M || (W2 && dW2 && ld2 >= H && lddw2 >= H)),
                "ctn_heads_bwd_coefs: W / dW of both row ranges");
    CTN_REQUIRE(gamma && beta && rowsum_b && wdot_sums && rstd && c1 && c2, "ctn_heads_bwd_coefs: null pointer");
    hipLaunchKernelGGL(heads_bwd_coefs_kernel, dim3(M), dim3(HC_NT), 0, (hipStream_t)stream, W, ld, W2, ld2, M, M1, H,
                       gamma, beta, rowsum_b, wdot_sums, rstd, batch, count, c1, c2, dW, lddw, dW2, lddw2, s1acc, ticket);
    CTN_CHECK_LAUNCH("ctn_heads_bwd_coefs");
    return CTN_OK;
}

int ctn_cln_bwd_coefs(const float* col, int64_t col_ld, const float* mean, const float* rstd, int64_t stat_ld,
                      int batch, int C, int T, float eps, float* c1, float* c2, void* stream) {
    CTN_REQUIRE(col && mean && rstd && c1 && c2 && batch > 0 && C > 0 && T > 0 && col_ld >= T && stat_ld >= T,
                "ctn_cln_bwd_coefs: bad arguments");
    hipLaunchKernelGGL(cln_bwd_coefs_kernel, dim3(batch), dim3(SCAN_NT), 0, (hipStream_t)stream, col, col_ld, mean,
                       rstd, stat_ld, C, T, eps, c1, c2);
    CTN_CHECK_LAUNCH("ctn_cln_bwd_coefs");
    return CTN_OK;
}

int ctn_norm_bwd_apply(const float* dn, int64_t lddn, const float* x, int64_t ld, int batch, int C, int T,
                       const float* alpha, const float* mean, const float* rstd, const float* c1, const float* c2,
                       int64_t ssb, int64_t sst, const float* gamma, float* dx, int64_t lddx, double* dalpha,
                       void* stream) {
    CTN_REQUIRE(ok_act(dn, lddn) && ok_act(x, ld) && ok_act(dx, lddx) && ld >= T && lddn >= T && lddx >= T,
                "ctn_norm_bwd_apply: bad layout");
    CTN_REQUIRE(mean && rstd && c1 && c2 && gamma && batch > 0 && C > 0 && T > 0,
                "ctn_norm_bwd_apply: bad arguments");
    StatPtr sp{mean, rstd, c1, c2, ssb, sst};
    dim3 grid(ctn_cdiv(T, TT), ctn_cdiv(C, CG), batch);
    hipLaunchKernelGGL(norm_bwd_apply_kernel, grid, dim3(NT), 0, (hipStream_t)stream, dn, lddn, x, ld, C, T, alpha,
                       sp, gamma, dx, lddx, dalpha);
    CTN_CHECK_LAUNCH("ctn_norm_bwd_apply");
    return CTN_OK;
}

}  // extern "C"
