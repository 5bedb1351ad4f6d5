// Utterance-level PIT SI-SDR (forward search + backward of the winning permutation) and the flat
// clip + Adam optimizer step, gfx950.
//
// SI-SDR(e, t) = 10 log10( (|a t|^2 + eps) / (|a t - e|^2 + eps) ),  a = <e,t> / (|t|^2 + eps)
// (src/criterion/sdr.py:135-137).  The S x S matrix of pairwise values is built in two passes over the
// signals: pass 1 the dot products <e_i, t_j> and |t_j|^2, pass 2 the residual energies |a_ij t_j - e_i|^2
// formed explicitly (the algebraic shortcut cancels catastrophically at high SDR).  Per-thread partial
// sums are fp32 over a few dozen samples, everything above that is accumulated in fp64, so the S! candidate
// losses are accurate to fp32 rounding and the arg-min reproduces the reference's choice, with exact ties
// (bit-identical candidates) resolved to the first permutation like torch.min.
#include <mutex>

#include "ctn_common.h"

// Work area of ctn_pit_sisdr_fwd_noalloc: the dot products and residual energies the two passes accumulate, for callers
// that cannot bring a workspace of their own.  Every GPU the library runs on holds its own copy of this array until the
// process ends; host code reaches it only through the per-device address resolved in pit_scratch_ptr() below (the
// array's name is a host-side stand-in, not a device address).
static __device__ __align__(256) double scratch[CTN_PIT_NOALLOC_MAX_WORK];

namespace {

constexpr int NTH = 256;
constexpr int CHUNK = 4096;   // samples per block

// work layout per batch element: dot[S*S] | tt[S] | res[S*S]
template <int S>
__global__ __launch_bounds__(NTH) void pit_pass1_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                        int T, int64_t ss, int64_t bs, double* work) {
    __shared__ float red[(S * S + S) * (NTH / 64)];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * CHUNK, t1 = min(t0 + CHUNK, T);
    float v[S * S + S];
#pragma unroll
    for (int i = 0; i < S * S + S; ++i) v[i] = 0.f;
    for (int t = t0 + threadIdx.x; t < t1; t += NTH) {
        float e[S], g[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            e[i] = est[(int64_t)b * bs + (int64_t)i * ss + t];
            g[i] = tgt[(int64_t)b * bs + (int64_t)i * ss + t];
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
#pragma unroll
            for (int j = 0; j < S; ++j) v[i * S + j] += e[i] * g[j];
            v[S * S + i] += g[i] * g[i];
        }
    }
    const float tot = ctn_block_sum<S * S + S>(v, red);
    if (threadIdx.x < S * S + S) atomicAdd(work + (int64_t)b * (2 * S * S + S) + threadIdx.x, (double)tot);
}

template <int S>
__global__ __launch_bounds__(NTH) void pit_pass2_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                        int T, int64_t ss, int64_t bs, float eps, double* work,
                                                        int plain_sdr = 0) {
    __shared__ float red[(S * S) * (NTH / 64)];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * CHUNK, t1 = min(t0 + CHUNK, T);
    double* wb = work + (int64_t)b * (2 * S * S + S);
    float alpha[S * S];
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
        for (int j = 0; j < S; ++j)
            alpha[i * S + j] = plain_sdr ? 1.f : (float)wb[i * S + j] / ((float)wb[S * S + j] + eps);
    float v[S * S];
#pragma unroll
    for (int i = 0; i < S * S; ++i) v[i] = 0.f;
    for (int t = t0 + threadIdx.x; t < t1; t += NTH) {
        float e[S], g[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            e[i] = est[(int64_t)b * bs + (int64_t)i * ss + t];
            g[i] = tgt[(int64_t)b * bs + (int64_t)i * ss + t];
        }
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int j = 0; j < S; ++j) {
                const float r = alpha[i * S + j] * g[j] - e[i];
                v[i * S + j] += r * r;
            }
    }
    const float tot = ctn_block_sum<S * S>(v, red);
    if (threadIdx.x < S * S) atomicAdd(wb + S * S + S + threadIdx.x, (double)tot);
}

template <int S>
__global__ void pit_finalize_kernel(const double* __restrict__ work, int batch, float eps, int fixed_perm,
                                    float* pair, float* loss_b, int64_t* perm, float* coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double* wb = work + (int64_t)b * (2 * S * S + S);
    float pr[S * S];
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            const float dot = (float)wb[i * S + j], tt = (float)wb[S * S + j], res = (float)wb[S * S + S + i * S + j];
            const float alpha = dot / (tt + eps);
            const float num = alpha * alpha * tt + eps;
            const float den = res + eps;
            const float v = 10.f * log10f(num / den);
            pr[i * S + j] = v;
            pair[(int64_t)b * S * S + i * S + j] = v;
        }
    // permutations in itertools (lexicographic) order; keep the first minimum
    int p[S], best[S];
    for (int i = 0; i < S; ++i) p[i] = i;
    float best_loss = 0.f;
    bool first = true;
    while (true) {
        float acc = 0.f;
        for (int i = 0; i < S; ++i) acc += -pr[i * S + p[i]];
        const float l = acc / (float)S;
        if (first || l < best_loss) {
            best_loss = l;
            for (int i = 0; i < S; ++i) best[i] = p[i];
            first = false;
        }
        if (fixed_perm) break;
        // next lexicographic permutation
        int k = S - 2;
        while (k >= 0 && p[k] > p[k + 1]) --k;
        if (k < 0) break;
        int l2 = S - 1;
        while (p[l2] < p[k]) --l2;
        int tmp = p[k]; p[k] = p[l2]; p[l2] = tmp;
        for (int lo = k + 1, hi = S - 1; lo < hi; ++lo, --hi) {
            tmp = p[lo]; p[lo] = p[hi]; p[hi] = tmp;
        }
    }
    loss_b[b] = best_loss;
    // gradient coefficients of the winning assignment (fp64):
    //   d sisdr / d e = K [ 2 a tt / ((tt+eps) num) - 2 rt / ((tt+eps) den) + 2 a / den ] t  -  K (2/den) e
    //   with K = 10/ln 10, rt = <a t - e, t> = -dot * eps / (tt + eps)
    const double K = 10.0 / log(10.0);
    for (int i = 0; i < S; ++i) {
        const int j = best[i];
        perm[(int64_t)b * S + i] = j;
        const double dot = wb[i * S + j], tt = wb[S * S + j], res = wb[S * S + S + i * S + j];
        const double a = dot / (tt + (double)eps);
        const double num = a * a * tt + (double)eps, den = res + (double)eps;
        const double rt = -dot * (double)eps / (tt + (double)eps);
        const double ct = K * (2.0 * a * tt / ((tt + eps) * num) - 2.0 * rt / ((tt + eps) * den) + 2.0 * a / den);
        const double ce = -K * 2.0 / den;
        coef[((int64_t)b * S + i) * 2 + 0] = (float)(-ct / (double)S);
        coef[((int64_t)b * S + i) * 2 + 1] = (float)(-ce / (double)S);
    }
}

__global__ __launch_bounds__(NTH) void pit_bwd_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                      int S, int T, int64_t ss, int64_t bs,
                                                      const int64_t* __restrict__ perm, const float* __restrict__ coef,
                                                      const float* __restrict__ gscale, int gs_bstride,
                                                      int gs_istride, float* dest, int64_t dss, int64_t dbs) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int t = blockIdx.x * NTH + threadIdx.x;
    if (t >= T) return;
    const int j = (int)perm[(int64_t)b * S + i];
    const float g = gscale[(int64_t)b * gs_bstride + (int64_t)i * gs_istride];
    const float a = coef[((int64_t)b * S + i) * 2 + 0] * g, c = coef[((int64_t)b * S + i) * 2 + 1] * g;
    dest[(int64_t)b * dbs + (int64_t)i * dss + t] =
        a * tgt[(int64_t)b * bs + (int64_t)j * ss + t] + c * est[(int64_t)b * bs + (int64_t)i * ss + t];
}

// ------------------------------------------------------------------------------------------ pairwise matrix
// Every pair's value and derivative, for criteria that need more than the winning assignment (Sinkhorn PIT) or the
// plain SDR: pair[b][i][j] = SI-SDR(e_i, t_j) (kind 0) or SDR(e_i, t_j) = 10 log10((|t|^2 + eps) / (|t - e|^2 + eps))
// (kind 1; src/criterion/sdr.py:19-20), and d pair / d e_i = ct * t_j + ce * e_i with (ct, ce) = paircoef[b][i][j].
template <int S>
__global__ void pair_finalize_kernel(const double* __restrict__ work, int batch, float eps, int kind, float* pair,
                                     float* paircoef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double* wb = work + (int64_t)b * (2 * S * S + S);
    const double K = 10.0 / log(10.0);
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            const int64_t o = (int64_t)b * S * S + i * S + j;
            const float dotf = (float)wb[i * S + j], ttf = (float)wb[S * S + j], resf = (float)wb[S * S + S + i * S + j];
            const double dot = wb[i * S + j], tt = wb[S * S + j], den = wb[S * S + S + i * S + j] + (double)eps;
            if (kind == 1) {
                pair[o] = 10.f * log10f((ttf + eps) / (resf + eps));
                paircoef[2 * o + 0] = (float)(2.0 * K / den);
                paircoef[2 * o + 1] = (float)(-2.0 * K / den);
            } else {
                const float alpha = dotf / (ttf + eps);
                pair[o] = 10.f * log10f((alpha * alpha * ttf + eps) / (resf + eps));
                const double a = dot / (tt + (double)eps);
                const double num = a * a * tt + (double)eps;
                const double rt = -dot * (double)eps / (tt + (double)eps);
                paircoef[2 * o + 0] = (float)(K * (2.0 * a * tt / ((tt + eps) * num) - 2.0 * rt / ((tt + eps) * den) + 2.0 * a / den));
                paircoef[2 * o + 1] = (float)(-K * 2.0 / den);
            }
        }
}

// dest_i = sum_j A[b][i][j] t_j + C[b][i] e_i
__global__ __launch_bounds__(NTH) void pair_bwd_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                       int S, int T, int64_t ss, int64_t bs,
                                                       const float* __restrict__ A, const float* __restrict__ C,
                                                       float* dest, int64_t dss, int64_t dbs) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int t = blockIdx.x * NTH + threadIdx.x;
    if (t >= T) return;
    float v = C[(int64_t)b * S + i] * est[(int64_t)b * bs + (int64_t)i * ss + t];
    for (int j = 0; j < S; ++j)
        v = __builtin_fmaf(A[((int64_t)b * S + i) * S + j], tgt[(int64_t)b * bs + (int64_t)j * ss + t], v);
    dest[(int64_t)b * dbs + (int64_t)i * dss + t] = v;
}

// ------------------------------------------------------------------------------------------ optimizer
__global__ __launch_bounds__(NTH) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* acc) {
    __shared__ float red[NTH / 64];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NTH + threadIdx.x; i < n; i += (int64_t)gridDim.x * NTH) s += g[i] * g[i];
    float v[1] = {s};
    const float tot = ctn_block_sum<1>(v, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)tot);
}

__global__ __launch_bounds__(NTH) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const double* sumsq, float max_norm, float grad_scale, float lr,
                                                   float beta1, float beta2, float eps, float bc1, float bc2_sqrt) {
    float clip = grad_scale;
    if (max_norm > 0.f && sumsq) {
        const float norm = (float)sqrt(sumsq[0]) * fabsf(grad_scale);
        const float c = max_norm / (norm + 1e-6f);
        clip *= fminf(c, 1.f);
    }
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * NTH + threadIdx.x; i < n; i += (int64_t)gridDim.x * NTH) {
        const float gi = g[i] * clip;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= step_size * (mi / denom);
    }
}

// Device address of `scratch` on the current device, resolved once per device (with that device current) and kept.
constexpr int MAX_DEVICES = 64;
double* g_scratch_ptr[MAX_DEVICES] = {};
std::mutex g_scratch_mutex;

int pit_scratch_ptr(double** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        ctn_set_error("ctn_pit_sisdr_fwd_noalloc: hipGetDevice failed or device index %d out of range (%s)", dev,
                      hipGetErrorString(e));
        return CTN_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        double* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            ctn_set_error("ctn_pit_sisdr_fwd_noalloc: hipGetSymbolAddress failed on device %d: %s", dev,
                          hipGetErrorString(e));
            return CTN_ERR_LAUNCH;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return CTN_OK;
}

template <int S>
void pit_launch(const float* est, const float* tgt, int batch, int T, int64_t ss, int64_t bs, float eps, int fixed_perm,
                double* work, float* pair, float* loss_b, int64_t* perm, float* coef, hipStream_t s) {
    dim3 grid(ctn_cdiv(T, CHUNK), batch);
    hipLaunchKernelGGL(pit_pass1_kernel<S>, grid, dim3(NTH), 0, s, est, tgt, T, ss, bs, work);
    hipLaunchKernelGGL(pit_pass2_kernel<S>, grid, dim3(NTH), 0, s, est, tgt, T, ss, bs, eps, work, 0);
    hipLaunchKernelGGL(pit_finalize_kernel<S>, dim3(ctn_cdiv(batch, 64)), dim3(64), 0, s, work, batch, eps, fixed_perm,
                       pair, loss_b, perm, coef);
}

}  // namespace

extern "C" {

int ctn_pit_sisdr_fwd_noalloc(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs,
                              float eps, int fixed_perm, float* pair, float* loss_b, int64_t* perm, float* coef,
                              void* stream) {
    CTN_REQUIRE(est && tgt && pair && loss_b && perm && coef, "ctn_pit_sisdr_fwd_noalloc: null pointer");
    CTN_REQUIRE(batch > 0 && T > 0 && S >= 1 && S <= 4, "ctn_pit_sisdr_fwd_noalloc: supports 1..4 sources");
    const int64_t need = (int64_t)batch * (2 * S * S + S);
    CTN_REQUIRE(need <= CTN_PIT_NOALLOC_MAX_WORK,
                "ctn_pit_sisdr_fwd_noalloc: batch * (2*S*S + S) = %lld exceeds the built-in work area of %d doubles "
                "(use ctn_pit_sisdr_fwd with a workspace of your own)", (long long)need, CTN_PIT_NOALLOC_MAX_WORK);
    double* work = nullptr;
    const int rc = pit_scratch_ptr(&work);
    if (rc != CTN_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    // the array keeps the previous call's sums: the part this call accumulates into starts at zero
    hipError_t e = hipMemsetAsync(work, 0, (size_t)need * sizeof(double), s);
    if (e != hipSuccess) {
        ctn_set_error("ctn_pit_sisdr_fwd_noalloc: hipMemsetAsync failed: %s", hipGetErrorString(e));
        return CTN_ERR_LAUNCH;
    }
    switch (S) {
        case 1: pit_launch<1>(est, tgt, batch, T, ss, bs, eps, fixed_perm, work, pair, loss_b, perm, coef, s); break;
        case 2: pit_launch<2>(est, tgt, batch, T, ss, bs, eps, fixed_perm, work, pair, loss_b, perm, coef, s); break;
        case 3: pit_launch<3>(est, tgt, batch, T, ss, bs, eps, fixed_perm, work, pair, loss_b, perm, coef, s); break;
        default: pit_launch<4>(est, tgt, batch, T, ss, bs, eps, fixed_perm, work, pair, loss_b, perm, coef, s); break;
    }
    CTN_CHECK_LAUNCH("ctn_pit_sisdr_fwd_noalloc");
    return CTN_OK;
}

int ctn_pit_sisdr_fwd(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs, float eps,
                      int fixed_perm, double* work, float* pair, float* loss_b, int64_t* perm, float* coef,
                      void* stream) {
    CTN_REQUIRE(est && tgt && work && pair && loss_b && perm && coef, "ctn_pit_sisdr_fwd: null pointer");
    CTN_REQUIRE(batch > 0 && T > 0 && S >= 1 && S <= 4, "ctn_pit_sisdr_fwd: supports 1..4 sources");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ctn_cdiv(T, CHUNK), batch);
#define CTN_PIT_CASE(SS)                                                                                        \
    case SS:                                                                                                    \
        hipLaunchKernelGGL(pit_pass1_kernel<SS>, grid, dim3(NTH), 0, s, est, tgt, T, ss, bs, work);             \
        hipLaunchKernelGGL(pit_pass2_kernel<SS>, grid, dim3(NTH), 0, s, est, tgt, T, ss, bs, eps, work);        \
        hipLaunchKernelGGL(pit_finalize_kernel<SS>, dim3(ctn_cdiv(batch, 64)), dim3(64), 0, s, work, batch, eps, \
                           fixed_perm, pair, loss_b, perm, coef);                                                           \
        break;
    switch (S) { CTN_PIT_CASE(1) CTN_PIT_CASE(2) CTN_PIT_CASE(3) CTN_PIT_CASE(4) }
#undef CTN_PIT_CASE
    CTN_CHECK_LAUNCH("ctn_pit_sisdr_fwd");
    return CTN_OK;
}

int ctn_pit_sisdr_bwd(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs,
                      const int64_t* perm, const float* coef, const float* gscale, int gs_bstride, int gs_istride,
                      float* dest, int64_t dss, int64_t dbs, void* stream) {
    CTN_REQUIRE(est && tgt && perm && coef && gscale && dest && batch > 0 && S >= 1 && T > 0,
                "ctn_pit_sisdr_bwd: bad arguments");
    dim3 grid(ctn_cdiv(T, NTH), S, batch);
    hipLaunchKernelGGL(pit_bwd_kernel, grid, dim3(NTH), 0, (hipStream_t)stream, est, tgt, S, T, ss, bs, perm, coef,
                       gscale, gs_bstride, gs_istride, dest, dss, dbs);
    CTN_CHECK_LAUNCH("ctn_pit_sisdr_bwd");
    return CTN_OK;
}

int ctn_pair_sdr_fwd(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs, float eps,
                     int kind, double* work, float* pair, float* paircoef, void* stream) {
    CTN_REQUIRE(est && tgt && work && pair && paircoef, "ctn_pair_sdr_fwd: null pointer");
    CTN_REQUIRE(batch > 0 && T > 0 && S >= 1 && S <= 4, "ctn_pair_sdr_fwd: supports 1..4 sources");
    CTN_REQUIRE(kind == 0 || kind == 1, "ctn_pair_sdr_fwd: kind is 0 (SI-SDR) or 1 (SDR)");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ctn_cdiv(T, CHUNK), batch);
#define CTN_PAIR_CASE(SS)                                                                                            \
    case SS:                                                                                                         \
        hipLaunchKernelGGL(pit_pass1_kernel<SS>, grid, dim3(NTH), 0, s, est, tgt, T, ss, bs, work);                  \
        hipLaunchKernelGGL(pit_pass2_kernel<SS>, grid, dim3(NTH), 0, s, est, tgt, T, ss, bs, eps, work, kind);       \
        hipLaunchKernelGGL(pair_finalize_kernel<SS>, dim3(ctn_cdiv(batch, 64)), dim3(64), 0, s, work, batch, eps,    \
                           kind, pair, paircoef);                                                                    \
        break;
    switch (S) { CTN_PAIR_CASE(1) CTN_PAIR_CASE(2) CTN_PAIR_CASE(3) CTN_PAIR_CASE(4) }
#undef CTN_PAIR_CASE
    CTN_CHECK_LAUNCH("ctn_pair_sdr_fwd");
    return CTN_OK;
}

int ctn_pair_sdr_bwd(const float* est, const float* tgt, int batch, int S, int T, int64_t ss, int64_t bs,
                     const float* A, const float* C, float* dest, int64_t dss, int64_t dbs, void* stream) {
    CTN_REQUIRE(est && tgt && A && C && dest && batch > 0 && S >= 1 && T > 0, "ctn_pair_sdr_bwd: bad arguments");
    dim3 grid(ctn_cdiv(T, NTH), S, batch);
    hipLaunchKernelGGL(pair_bwd_kernel, grid, dim3(NTH), 0, (hipStream_t)stream, est, tgt, S, T, ss, bs, A, C, dest,
                       dss, dbs);
    CTN_CHECK_LAUNCH("ctn_pair_sdr_bwd");
    return CTN_OK;
}

int ctn_sumsq(const float* g, int64_t n, double* acc, void* stream) {
    CTN_REQUIRE(g && acc && n > 0, "ctn_sumsq: bad arguments");
    int64_t blocks = (n + NTH * 8 - 1) / (NTH * 8);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(NTH), 0, (hipStream_t)stream, g, n, acc);
    CTN_CHECK_LAUNCH("ctn_sumsq");
    return CTN_OK;
}

int ctn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm,
                  float grad_scale, float lr, float beta1, float beta2, float eps, int step, void* stream) {
    CTN_REQUIRE(p && g && m && v && n > 0 && step >= 1, "ctn_adam_step: bad arguments");
    CTN_REQUIRE(!(max_norm > 0.f) || sumsq, "ctn_adam_step: clipping needs sumsq");
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    int64_t blocks = (n + NTH * 4 - 1) / (NTH * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(NTH), 0, (hipStream_t)stream, p, g, m, v, n, sumsq,
                       max_norm, grad_scale, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
    CTN_CHECK_LAUNCH("ctn_adam_step");
    return CTN_OK;
}

}  // extern "C"
