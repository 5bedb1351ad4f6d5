// Encoder (strided Conv1d), mask * decoder (ConvTranspose1d overlap-add) and their backward kernels, gfx950.
//
// All four kernels put the FRAME index on the lane, so reads and writes of the (batch, N, frames)
// latent tensors are fully coalesced 256-byte rows, and keep the tiny L-tap basis in scalar registers
// (the basis row of channel n is wave-uniform).  Weight gradients, which reduce over frames, go through
// a transposed LDS tile so that a thread owns a channel and walks frames.
#include <algorithm>

#include "ctn_common.h"

namespace {

constexpr int NTH = 256;
constexpr int FT = 64;         // frames per tile (one per lane)
constexpr int LDV = FT + 1;    // LDS row stride of the transposed value tiles

// ------------------------------------------------------------------------------------------ encoder fwd
// The L-tap basis rows are staged in LDS zero-padded to ML taps: the inner products then run over a
// compile-time tap count with wave-uniform (broadcast) ds_read_b128 operands.  Reading the rows from global
// memory inside the channel loop makes every tap a dependent load.
template <int ML>
__global__ __launch_bounds__(NTH) void encoder_fwd_kernel(const float* __restrict__ x, int64_t xbs, int Tin,
                                                          int pad_left, const float* __restrict__ W, int N, int L,
                                                          int S, int relu, float* __restrict__ w, int Tf,
                                                          int64_t ldw, double* sums, float* col, int64_t col_ld,
                                                          int NC) {
    __shared__ float red[2 * (NTH / 64)];
    __shared__ __attribute__((aligned(16))) float Wl[64 * ML];       // NC <= 64 rows
    const int b = blockIdx.z, n0 = blockIdx.y * NC, t = blockIdx.x * NTH + threadIdx.x;
    const int nend = min(n0 + NC, N);
    for (int i = threadIdx.x; i < NC * ML; i += NTH) {
        const int n = n0 + i / ML, l = i % ML;
        Wl[i] = (n < N && l < L) ? W[(int64_t)n * L + l] : 0.f;
    }
    const bool tv = t < Tf;
    float xv[ML];
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        const int tau = t * S + l - pad_left;
        xv[l] = (tv && l < L && tau >= 0 && tau < Tin) ? x[(int64_t)b * xbs + tau] : 0.f;
    }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    float* wout = w + ((int64_t)b * N + n0) * ldw + t;
#pragma unroll 4
    for (int n = n0; n < nend; ++n) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(&Wl[(n - n0) * ML]);
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < ML / 4; ++q) {
            const f32x4 c = wr[q];
            acc += c[0] * xv[4 * q] + c[1] * xv[4 * q + 1] + c[2] * xv[4 * q + 2] + c[3] * xv[4 * q + 3];
        }
        if (relu) acc = fmaxf(acc, 0.f);
        if (tv) {
            wout[(int64_t)(n - n0) * ldw] = acc;
            s1 += acc;
            s2 += acc * acc;
        }
    }
    if (col) {
        if (tv) {
            atomicAdd(col + ((int64_t)b * 2 + 0) * col_ld + t, s1);
            atomicAdd(col + ((int64_t)b * 2 + 1) * col_ld + t, s2);
        }
    } else if (sums) {
        float v[2] = {s1, s2};
        const float tot = ctn_block_sum<2>(v, red);
        if (threadIdx.x < 2) atomicAdd(sums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + threadIdx.x, (double)tot);
    }
}

// ------------------------------------------------------------------------------------------ mask * decode fwd
// One block = (frame tile, batch element), all NS sources (the latent w is read once).  Lane = frame (FT - R1
// new frames plus R1 halo frames on the left, R1 = (L-1)/S, so every output sample of the tile is complete),
// wave = quarter of the channels of the current 256-channel chunk, whose basis rows sit in LDS.
constexpr int MDC = 256;       // channels per staged basis chunk
template <int ML, int NS, bool MASKED>
__global__ __launch_bounds__(NTH) void maskdecode_fwd_kernel(const float* __restrict__ w, int64_t ldw,
                                                             const float* __restrict__ mask, int64_t ldm,
                                                             int N, int Tf, const float* __restrict__ Wd, int L,
                                                             int S, float* __restrict__ out, int64_t out_ss,
                                                             int64_t out_bs, int Tout, int trim_left, int R1) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* part = lds;                                 // [4][NS][FT * (ML + 1)]
    float* Wl = lds + 4 * NS * FT * (ML + 1);          // [MDC][ML]
    const int tile = blockIdx.x, b = blockIdx.y;
    const int TTo = FT - R1;
    const int f0 = tile * TTo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = f0 - R1 + lane;
    const bool fv = f >= 0 && f < Tf;
    const int fc = min(max(f, 0), Tf - 1);             // clamped: loads stay unconditional, values are masked
    float acc[NS][ML];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int l = 0; l < ML; ++l) acc[s][l] = 0.f;
    const float* wp = w + (int64_t)b * N * ldw + fc;
    const float* mp = MASKED ? mask + (int64_t)b * NS * N * ldm + fc : nullptr;
    for (int nc = 0; nc < N; nc += MDC) {
        __syncthreads();
        for (int i = threadIdx.x; i < MDC * ML; i += NTH) {
            const int n = nc + i / ML, l = i % ML;
            Wl[i] = (n < N && l < L) ? Wd[(int64_t)n * L + l] : 0.f;
        }
        __syncthreads();
        const int nb = nc + wave * (MDC / 4), ne = min(nb + MDC / 4, N);
        // 4 channels per step: all loads first (unconditional, clamped addresses; out-of-range values are
        // zeroed by selects), so 4 (1 + NS) rows are in flight per wave instead of one dependent load at a time
        for (int n = nb; n < ne; n += 4) {
            float wv[4], mv[NS][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int nn = min(n + u, ne - 1);
                wv[u] = wp[(int64_t)nn * ldw];
#pragma unroll
                for (int s = 0; s < NS; ++s) mv[s][u] = MASKED ? mp[((int64_t)s * N + nn) * ldm] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = fv && (n + u < ne);
                float h[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) h[s] = ok ? wv[u] * mv[s][u] : 0.f;
                const f32x4* dr = reinterpret_cast<const f32x4*>(&Wl[(min(n + u, ne - 1) - nc) * ML]);
#pragma unroll
                for (int q = 0; q < ML / 4; ++q) {
                    const f32x4 c = dr[q];
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        acc[s][4 * q] += c[0] * h[s];
                        acc[s][4 * q + 1] += c[1] * h[s];
                        acc[s][4 * q + 2] += c[2] * h[s];
                        acc[s][4 * q + 3] += c[3] * h[s];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int l = 0; l < ML; ++l) part[((wave * NS + s) * FT + lane) * (ML + 1) + l] = acc[s][l];
    __syncthreads();
    // overlap-add: sample tau collects frame fr, tap tau - fr*S
    const int Tpad = (Tf - 1) * S + L;
    const bool last = (f0 + TTo >= Tf);
    const int tau0 = f0 * S;
    const int tau1 = last ? Tpad : (f0 + TTo) * S;
    const int span = tau1 - tau0;
    for (int i = threadIdx.x; i < span * NS; i += NTH) {
        const int s = i / span, tau = tau0 + i % span;
        float v = 0.f;
        int fr = tau / S;                  // newest contributing frame
        for (int r = 0; r <= R1; ++r, --fr) {
            const int l = tau - fr * S;
            const int li = fr - (f0 - R1);
            if (l < L && fr >= 0 && fr < Tf && li >= 0 && li < FT) {
                const int o = li * (ML + 1) + l;
                const int ps = FT * (ML + 1);
                v += (part[(0 * NS + s) * ps + o] + part[(1 * NS + s) * ps + o]) +
                     (part[(2 * NS + s) * ps + o] + part[(3 * NS + s) * ps + o]);
            }
        }
        const int to = tau - trim_left;
        if (to >= 0 && to < Tout) out[(int64_t)b * out_bs + (int64_t)s * out_ss + to] = v;
    }
}

// ------------------------------------------------------------------------------------------ shared phase B
// V[ns][FT rows n_local][LDV] holds values per (channel, frame); F[ns][FLEN] the tile's stretch of the
// time-domain signal, stored linearly (frame f, tap l -> F[f * S + l]; filled by coalesced loads, zero outside
// the signal).  Thread (n_local, q) adds sum_f V[n][f] * F[f * S + l] over its quarter of the frames to acc.
// corr_flush folds the four quarters through LDS (scratch: 4 x 64 x ML floats) and issues one atomic per (n, l).
constexpr int FLEN = FT * 16 + 32;      // samples of one tile for stride <= 16 and <= 32 taps

template <int ML, int NS>
__device__ __forceinline__ void corr_accumulate(const float* V, const float* F, int S, float (&acc)[ML]) {
    const int nl = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int s = 0; s < NS; ++s) {
        const float* Vs = V + (size_t)s * FT * LDV + nl * LDV;
        const float* Fs = F + (size_t)s * FLEN;
#pragma unroll 4
        for (int fl = q * (FT / 4); fl < (q + 1) * (FT / 4); ++fl) {
            const float h = Vs[fl];
#pragma unroll
            for (int l = 0; l < ML; ++l) acc[l] += h * Fs[fl * S + l];     // taps >= L: basis rows are never read
        }
    }
}

// Coalesced fill of F: samples tau0 .. tau0 + FT*S + ML - 1 of sig (zero outside [0, Tlen)).
__device__ __forceinline__ void fill_samples(float* F, const float* __restrict__ sig, int tau0, int Tlen, int n) {
    for (int i = threadIdx.x; i < n; i += NTH) {
        const int tau = tau0 + i;
        const float v = sig[min(max(tau, 0), Tlen - 1)];
        F[i] = (tau >= 0 && tau < Tlen) ? v : 0.f;
    }
}

template <int ML>
__device__ __forceinline__ void corr_flush(float* scratch, const float (&acc)[ML], int L, int n0, int N, float* dW) {
    const int nl = threadIdx.x & 63, q = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int l = 0; l < ML; ++l) scratch[(q * 64 + nl) * ML + l] = acc[l];
    __syncthreads();
    const int n = n0 + nl;
    if (q == 0 && n < N) {
#pragma unroll
        for (int l = 0; l < ML; ++l) {
            if (l < L) {
                const float v = (scratch[nl * ML + l] + scratch[(64 + nl) * ML + l]) +
                                (scratch[(128 + nl) * ML + l] + scratch[(192 + nl) * ML + l]);
                atomicAdd(dW + (int64_t)n * L + l, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ mask * decode bwd
// Block = (frame tile of FT, chunk of 64 channels, batch element).
template <int ML, int NS, bool MASKED>
__global__ __launch_bounds__(NTH) void maskdecode_bwd_kernel(const float* __restrict__ dout, int64_t dout_ss,
                                                             int64_t dout_bs, int Tout, int trim_left,
                                                             const float* __restrict__ w, int64_t ldw,
                                                             const float* __restrict__ mask, int64_t ldm, int N,
                                                             int Tf, const float* __restrict__ Wd, int L, int S,
                                                             int sigmoid_bwd, float* __restrict__ dz,
                                                             int64_t lddz, float* __restrict__ dw_dec,
                                                             int64_t lddw, float* dWd, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // V: [NS][64][LDV] value tiles, and corr_flush's scratch of 4 * 64 * ML floats -- sized for the larger of the two
    constexpr size_t VSZ = (size_t)NS * FT * LDV > (size_t)4 * 64 * ML ? (size_t)NS * FT * LDV : (size_t)4 * 64 * ML;
    float* V = lds;
    float* F = lds + VSZ;                         // [NS][FLEN] output-gradient samples of the tile
    float* Wl = F + (size_t)NS * FLEN;            // [64][ML] basis rows of this block's channels, zero-padded
    const int b = blockIdx.z, n0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * ML; i += NTH) {
        const int n = n0 + i / ML, l = i % ML;
        Wl[i] = (n < N && l < L) ? Wd[(int64_t)n * L + l] : 0.f;
    }
    __syncthreads();
    float acc[ML];
#pragma unroll
    for (int l = 0; l < ML; ++l) acc[l] = 0.f;
    const int ntiles = (Tf + FT - 1) / FT;
    const int tile_end = min((int)(blockIdx.x + 1) * tiles_per_block, ntiles);
  for (int tile = blockIdx.x * tiles_per_block; tile < tile_end; ++tile) {
    const int f0 = tile * FT;
    const int f = f0 + lane;
    const bool fv = f < Tf;
    // the tile's stretch of the output gradient, staged once with coalesced loads
#pragma unroll
    for (int s = 0; s < NS; ++s)
        fill_samples(F + (size_t)s * FLEN, dout + (int64_t)b * dout_bs + (int64_t)s * dout_ss, f0 * S - trim_left, Tout,
                     FT * S + ML);
    __syncthreads();
    float dx[NS][ML];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int l = 0; l < ML; ++l) dx[s][l] = (fv && l < L) ? F[(size_t)s * FLEN + lane * S + l] : 0.f;
    // global loads below are unconditional with clamped addresses and selects on the values: a guarded load
    // compiles to a branch plus a full vmcnt wait per element, i.e. one dependent memory round trip each
    const int fc = min(f, Tf - 1);
    for (int i0 = 0; i0 < 16; i0 += 4) {
        float wv4[4], m4[NS][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = min(n0 + wave * 16 + i0 + u, N - 1);
            wv4[u] = w[((int64_t)b * N + n) * ldw + fc];
#pragma unroll
            for (int s = 0; s < NS; ++s)
                m4[s][u] = MASKED ? mask[(((int64_t)b * NS + s) * N + n) * ldm + fc] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int nl = wave * 16 + i0 + u, n = n0 + nl;
            const bool ok = (n < N) && fv;
            const float wv = ok ? wv4[u] : 0.f;
            float dwsum = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float dwh = 0.f;
                const f32x4* dr = reinterpret_cast<const f32x4*>(&Wl[nl * ML]);
#pragma unroll
                for (int q = 0; q < ML / 4; ++q) {
                    const f32x4 c = dr[q];
                    dwh += c[0] * dx[s][4 * q] + c[1] * dx[s][4 * q + 1] + c[2] * dx[s][4 * q + 2] +
                           c[3] * dx[s][4 * q + 3];
                }
                const float m = ok ? m4[s][u] : 1.f;
                if (dz && ok) {
                    const float dzv = dwh * wv * (sigmoid_bwd ? m * (1.f - m) : 1.f);
                    dz[(((int64_t)b * NS + s) * N + n) * lddz + f] = dzv;
                }
                dwsum += dwh * m;
                V[((size_t)s * FT + nl) * LDV + lane] = wv * m;
            }
            if (dw_dec && ok) dw_dec[((int64_t)b * N + n) * lddw + f] = dwsum;
        }
    }
    __syncthreads();
    corr_accumulate<ML, NS>(V, F, S, acc);
    __syncthreads();
  }
    corr_flush<ML>(V, acc, L, n0, N, dWd);
}

// ------------------------------------------------------------------------------------------ encoder bwd
template <int ML, bool HAS_DWA, bool HAS_DN>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(2, 4))) void encoder_bwd_kernel(const float* __restrict__ x, int64_t xbs, int Tin,
                                                          int pad_left, int N, int L, int S, int relu,
                                                          const float* __restrict__ w, int Tf, int64_t ldw,
                                                          const float* __restrict__ dw_a, int64_t lddwa,
                                                          const float* __restrict__ dn, int64_t lddn,
                                                          const float* mean, const float* rstd, const float* c1,
                                                          const float* c2, int64_t ssb, int64_t sst,
                                                          const float* __restrict__ gamma, float* dW,
                                                          float* dgamma, float* dbeta, float* dwt_out,
                                                          int64_t lddwt, int tiles_per_block) {
    // 64 x 65 value tile, and corr_flush's scratch of 4 * 64 * ML floats -- sized for the larger of the two
    __shared__ float V[FT * LDV > 4 * 64 * ML ? FT * LDV : 4 * 64 * ML];
    __shared__ float F[FLEN];
    const int b = blockIdx.z, n0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[ML];
#pragma unroll
    for (int l = 0; l < ML; ++l) acc[l] = 0.f;
    float gxa[16], gda[16];                       // per-channel norm-gradient partials of this wave's 16 channels
#pragma unroll
    for (int i = 0; i < 16; ++i) gxa[i] = gda[i] = 0.f;
    const int ntiles = (Tf + FT - 1) / FT;
    const int tile_end = min((int)(blockIdx.x + 1) * tiles_per_block, ntiles);
  for (int tile = blockIdx.x * tiles_per_block; tile < tile_end; ++tile) {
    const int f0 = tile * FT;
    const int f = f0 + lane;
    const bool fv = f < Tf;
    // the tile's stretch of the input signal, staged once with coalesced loads; the global loads below are
    // unconditional with clamped addresses and selects on the values (a guarded load costs a branch and a full
    // vmcnt wait per element)
    fill_samples(F, x + (int64_t)b * xbs, f0 * S - pad_left, Tin, FT * S + ML);
    const int fc = min(f, Tf - 1);
    float mu = 0.f, rs = 1.f, k1 = 0.f, k2 = 0.f;
    if (HAS_DN) {
        const int64_t si = (int64_t)b * ssb + (int64_t)fc * sst;
        mu = mean[si]; rs = rstd[si]; k1 = c1[si]; k2 = c2[si];
    }
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 4) {       // 4 channels per step, all their loads first
        float wv4[4], da4[4], dn4[4], g4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = min(n0 + wave * 16 + i0 + u, N - 1);
            const int64_t row = (int64_t)b * N + n;
            wv4[u] = w[row * ldw + fc];
            da4[u] = HAS_DWA ? dw_a[row * lddwa + fc] : 0.f;
            dn4[u] = HAS_DN ? dn[row * lddn + fc] : 0.f;
            g4[u] = HAS_DN ? gamma[n] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            const int nl = wave * 16 + i, n = n0 + nl;
            const bool ok = (n < N) && fv;
            const float wv = wv4[u];
            float dwt = da4[u], gx = 0.f, gd = 0.f;
            if (HAS_DN) {
                const float d = dn4[u];
                const float xc = wv - mu;
                dwt += g4[u] * d * rs + k1 + xc * k2;
                gx = ok ? d * xc * rs : 0.f;
                gd = ok ? d : 0.f;
            }
            if (relu && !(wv > 0.f)) dwt = 0.f;
            if (!ok) dwt = 0.f;
            if (dwt_out && ok) dwt_out[((int64_t)b * N + n) * lddwt + f] = dwt;
            V[nl * LDV + lane] = dwt;
            gxa[i] += gx;
            gda[i] += gd;
        }
    }
    __syncthreads();
    corr_accumulate<ML, 1>(V, F, S, acc);
    __syncthreads();
  }
    if (dn) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = n0 + wave * 16 + i;
            const float gx = ctn_wave_sum(gxa[i]);
            const float gd = ctn_wave_sum(gda[i]);
            if (lane == 0 && n < N) {
                atomicAdd(dgamma + n, gx);
                atomicAdd(dbeta + n, gd);
            }
        }
    }
    corr_flush<ML>(V, acc, L, n0, N, dW);
}

// Frame tiles walked by one block of the weight-gradient kernels: enough blocks to fill the chip (~1500),
// as few as possible beyond that, because every block ends in N*L contended atomics.
int tiles_per_block(int ntiles, int other_blocks) {
    int tpb = (int)(((int64_t)ntiles * other_blocks + 1535) / 1536);
    if (tpb < 1) tpb = 1;
    if (tpb > ntiles) tpb = ntiles;
    return tpb;
}

}  // namespace

extern "C" {

int ctn_encoder_fwd(const float* x, int64_t xbs, int Tin, int pad_left, int batch, const float* W, int N, int L,
                    int S, int relu, float* w, int Tf, int64_t ldw, double* sums, float* col, int64_t col_ld,
                    void* stream) {
    CTN_REQUIRE(x && W && w && batch > 0 && N > 0 && L > 0 && L <= 64 && S > 0 && Tf > 0 && Tin > 0,
                "ctn_encoder_fwd: bad arguments (L <= 64)");
    CTN_REQUIRE(ldw >= Tf, "ctn_encoder_fwd: ldw < Tf");
    CTN_REQUIRE(!(sums && col), "ctn_encoder_fwd: sums and col are exclusive");
    CTN_REQUIRE(!col || col_ld >= Tf, "ctn_encoder_fwd: col_ld < Tf");
    const int NC = N >= 64 ? 64 : N;
    dim3 grid(ctn_cdiv(Tf, NTH), ctn_cdiv(N, NC), batch);
    hipStream_t s = (hipStream_t)stream;
    if (L <= 16)
        hipLaunchKernelGGL(encoder_fwd_kernel<16>, grid, dim3(NTH), 0, s, x, xbs, Tin, pad_left, W, N, L, S, relu, w,
                           Tf, ldw, sums, col, col_ld, NC);
    else
        hipLaunchKernelGGL(encoder_fwd_kernel<64>, grid, dim3(NTH), 0, s, x, xbs, Tin, pad_left, W, N, L, S, relu, w,
                           Tf, ldw, sums, col, col_ld, NC);
    CTN_CHECK_LAUNCH("ctn_encoder_fwd");
    return CTN_OK;
}

int ctn_maskdecode_fwd(const float* w, int64_t ldw, const float* mask, int64_t ldm, int batch, int n_src, int N,
                       int Tf, const float* Wd, int L, int S, float* out, int64_t out_ss, int64_t out_bs, int Tout,
                       int trim_left, void* stream) {
    CTN_REQUIRE(w && Wd && out && batch > 0 && n_src > 0 && N > 0 && Tf > 0 && L > 0 && L <= 32 && S > 0 && Tout > 0,
                "ctn_maskdecode_fwd: bad arguments (L <= 32)");
    CTN_REQUIRE(ldw >= Tf && (!mask || ldm >= Tf), "ctn_maskdecode_fwd: ld < Tf");
    const int R1 = (L - 1) / S;
    CTN_REQUIRE(R1 < FT / 2, "ctn_maskdecode_fwd: kernel_size / stride too large");
    CTN_REQUIRE(n_src <= (L <= 16 ? 4 : 3), "ctn_maskdecode_fwd: n_src <= 4 (<= 3 for more than 16 taps)");
    const int TTo = FT - R1;
    dim3 grid(ctn_cdiv(Tf, TTo), batch);
    hipStream_t s = (hipStream_t)stream;
    const int ML = L <= 16 ? 16 : 32;
    const size_t shmem = ((size_t)4 * n_src * FT * (ML + 1) + (size_t)MDC * ML) * sizeof(float);
#define CTN_MDF_(MLV, NS, MK)                                                                                         \
    do {                                                                                                           \
        if (shmem > 48 * 1024)                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&maskdecode_fwd_kernel<MLV, NS, MK>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                     \
        hipLaunchKernelGGL((maskdecode_fwd_kernel<MLV, NS, MK>), grid, dim3(NTH), shmem, s, w, ldw, mask, ldm, N,  \
                           Tf, Wd, L, S, out, out_ss, out_bs, Tout, trim_left, R1);                                \
    } while (0)
    if (ML == 16) {
#define CTN_MDF(MLV, NS)              \
    do {                               \
        if (mask) CTN_MDF_(MLV, NS, true); \
        else CTN_MDF_(MLV, NS, false);     \
    } while (0)
        if (n_src == 1) CTN_MDF(16, 1);
        else if (n_src == 2) CTN_MDF(16, 2);
        else if (n_src == 3) CTN_MDF(16, 3);
        else CTN_MDF(16, 4);
    } else {
        if (n_src == 1) CTN_MDF(32, 1);
        else if (n_src == 2) CTN_MDF(32, 2);
        else CTN_MDF(32, 3);
    }
#undef CTN_MDF
#undef CTN_MDF_
    CTN_CHECK_LAUNCH("ctn_maskdecode_fwd");
    return CTN_OK;
}

int ctn_maskdecode_bwd(const float* dout, int64_t dout_ss, int64_t dout_bs, int Tout, int trim_left, const float* w,
                       int64_t ldw, const float* mask, int64_t ldm, int batch, int n_src, int N, int Tf,
                       const float* Wd, int L, int S, int sigmoid_bwd, float* dz, int64_t lddz, float* dw_dec,
                       int64_t lddw, float* dWd, void* stream) {
    CTN_REQUIRE(dout && w && Wd && dWd && batch > 0 && n_src >= 1 && N > 0 && Tf > 0 && L > 0 && L <= 32 && S > 0,
                "ctn_maskdecode_bwd: bad arguments (L <= 32)");
    CTN_REQUIRE(n_src <= (L <= 16 ? 4 : 3), "ctn_maskdecode_bwd: n_src <= 4 (<= 3 for more than 16 taps)");
    CTN_REQUIRE(ldw >= Tf && (!mask || ldm >= Tf) && (!dz || lddz >= Tf) && (!dw_dec || lddw >= Tf),
                "ctn_maskdecode_bwd: ld < Tf");
    CTN_REQUIRE(!mask || dz, "ctn_maskdecode_bwd: dz required with a mask");
    CTN_REQUIRE(S <= 16, "ctn_maskdecode_bwd: stride <= 16");
    const int ntiles = ctn_cdiv(Tf, FT);
    const int tpb = tiles_per_block(ntiles, ctn_cdiv(N, 64) * batch);
    dim3 grid(ctn_cdiv(ntiles, tpb), ctn_cdiv(N, 64), batch);
    hipStream_t s = (hipStream_t)stream;
    const int ML = L <= 16 ? 16 : 32;
    const size_t vsz = std::max((size_t)n_src * FT * LDV, (size_t)4 * 64 * ML);
    const size_t shmem = (vsz + (size_t)n_src * FLEN + (size_t)64 * ML) * sizeof(float);
#define CTN_MD_LAUNCH(MLV, NS, MK)                                                                                  \
    do {                                                                                                            \
        if (shmem > 48 * 1024)                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&maskdecode_bwd_kernel<MLV, NS, MK>),           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                      \
        hipLaunchKernelGGL((maskdecode_bwd_kernel<MLV, NS, MK>), grid, dim3(NTH), shmem, s, dout, dout_ss, dout_bs,   \
                           Tout, trim_left, w, ldw, mask, ldm, N, Tf, Wd, L, S, sigmoid_bwd, dz, lddz, dw_dec, lddw,  \
                           dWd, tpb);                                                                               \
    } while (0)
#define CTN_MD_CASE(MLV, NS)                      \
    case NS:                                      \
        if (mask) CTN_MD_LAUNCH(MLV, NS, true);   \
        else CTN_MD_LAUNCH(MLV, NS, false);       \
        break;
    if (ML == 16) {
        switch (n_src) { CTN_MD_CASE(16, 1) CTN_MD_CASE(16, 2) CTN_MD_CASE(16, 3) CTN_MD_CASE(16, 4) }
    } else {
        switch (n_src) { CTN_MD_CASE(32, 1) CTN_MD_CASE(32, 2) CTN_MD_CASE(32, 3) }
    }
#undef CTN_MD_CASE
#undef CTN_MD_LAUNCH
    CTN_CHECK_LAUNCH("ctn_maskdecode_bwd");
    return CTN_OK;
}

int ctn_encoder_bwd(const float* x, int64_t xbs, int Tin, int pad_left, int batch, int N, int L, int S, int relu,
                    const float* w, int Tf, int64_t ldw, const float* dw_a, int64_t lddwa, const float* dn,
                    int64_t lddn, const float* mean, const float* rstd, const float* c1, const float* c2, int64_t ssb,
                    int64_t sst, const float* gamma, float* dW, float* dgamma, float* dbeta, float* dwt_out,
                    int64_t lddwt, void* stream) {
    CTN_REQUIRE(x && w && dW && batch > 0 && N > 0 && L > 0 && L <= 32 && S > 0 && Tf > 0 && Tin > 0,
                "ctn_encoder_bwd: bad arguments (L <= 32)");
    CTN_REQUIRE(ldw >= Tf && (!dw_a || lddwa >= Tf) && (!dn || lddn >= Tf) && (!dwt_out || lddwt >= Tf),
                "ctn_encoder_bwd: ld < Tf");
    CTN_REQUIRE(dw_a || dn, "ctn_encoder_bwd: no incoming gradient");
    CTN_REQUIRE(S <= 16, "ctn_encoder_bwd: stride <= 16");
    if (dn) CTN_REQUIRE(mean && rstd && c1 && c2 && gamma && dgamma && dbeta, "ctn_encoder_bwd: norm arguments");
    const int ntiles = ctn_cdiv(Tf, FT);
    const int tpb = tiles_per_block(ntiles, ctn_cdiv(N, 64) * batch);
    dim3 grid(ctn_cdiv(ntiles, tpb), ctn_cdiv(N, 64), batch);
#define CTN_EB_(MLV, DA, DN)                                                                                      \
    hipLaunchKernelGGL((encoder_bwd_kernel<MLV, DA, DN>), grid, dim3(NTH), 0, (hipStream_t)stream, x, xbs, Tin,       \
                       pad_left, N, L, S, relu, w, Tf, ldw, dw_a, lddwa, dn, lddn, mean, rstd, c1, c2, ssb, sst, gamma, \
                       dW, dgamma, dbeta, dwt_out, lddwt, tpb)
#define CTN_EB(DA, DN)                  \
    do {                                \
        if (L <= 16) CTN_EB_(16, DA, DN); \
        else CTN_EB_(32, DA, DN);         \
    } while (0)
    if (dw_a && dn) CTN_EB(true, true);
    else if (dw_a) CTN_EB(true, false);
    else CTN_EB(false, true);
#undef CTN_EB_
#undef CTN_EB
    CTN_CHECK_LAUNCH("ctn_encoder_bwd");
    return CTN_OK;
}

}  // extern "C"
