// Encoder (strided Conv1d), mask * decoder (ConvTranspose1d overlap-add) and their backward kernels, gfx950.
//
// All four kernels put the FRAME index on the lane, so reads and writes of the (batch, N, frames)
// latent tensors are fully coalesced 256-byte rows, and keep the tiny L-tap basis in scalar registers
// (the basis row of channel n is wave-uniform).  Weight gradients, which reduce over frames, go through
// a transposed LDS tile so that a thread owns a channel and walks frames.
#include "ctn_common.h"

namespace {

constexpr int NTH = 256;
constexpr int FT = 64;         // frames per tile (one per lane)
constexpr int LDV = FT + 1;    // LDS row stride of the transposed value tiles

// ------------------------------------------------------------------------------------------ encoder fwd
template <int ML>
__global__ __launch_bounds__(NTH) void encoder_fwd_kernel(const float* __restrict__ x, int64_t xbs, int Tin,
                                                          int pad_left, const float* __restrict__ W, int N, int L,
                                                          int S, int relu, float* __restrict__ w, int Tf,
                                                          int64_t ldw, double* sums, float* col, int64_t col_ld,
                                                          int NC) {
    __shared__ float red[2 * (NTH / 64)];
    const int b = blockIdx.z, n0 = blockIdx.y * NC, t = blockIdx.x * NTH + threadIdx.x;
    const bool tv = t < Tf;
    float xv[ML];
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        const int tau = t * S + l - pad_left;
        xv[l] = (tv && l < L && tau >= 0 && tau < Tin) ? x[(int64_t)b * xbs + tau] : 0.f;
    }
    float s1 = 0.f, s2 = 0.f;
    const int nend = min(n0 + NC, N);
    for (int n = n0; n < nend; ++n) {
        const float* wr = W + (int64_t)n * L;
        float acc = 0.f;
#pragma unroll
        for (int l = 0; l < ML; ++l)
            if (l < L) acc += wr[l] * xv[l];
        if (relu) acc = fmaxf(acc, 0.f);
        if (tv) {
            w[((int64_t)b * N + n) * ldw + t] = acc;
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
// One block = (frame tile, batch element, source).  Lane = frame (FT - R1 new frames plus R1 halo frames on
// the left, R1 = (L-1)/S, so every output sample of the tile is complete), wave = quarter of the channels.
template <int ML>
__global__ __launch_bounds__(NTH) void maskdecode_fwd_kernel(const float* __restrict__ w, int64_t ldw,
                                                             const float* __restrict__ mask, int64_t ldm,
                                                             int n_src, int N, int Tf,
                                                             const float* __restrict__ Wd, int L, int S,
                                                             float* __restrict__ out, int64_t out_ss,
                                                             int64_t out_bs, int Tout, int trim_left, int R1) {
    __shared__ float part[4][FT * (ML + 1)];
    const int tile = blockIdx.x, b = blockIdx.y / n_src, s = blockIdx.y % n_src;
    const int TTo = FT - R1;
    const int f0 = tile * TTo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = f0 - R1 + lane;
    const bool fv = f >= 0 && f < Tf;
    float acc[ML];
#pragma unroll
    for (int l = 0; l < ML; ++l) acc[l] = 0.f;
    const int nq = (N + 3) / 4;
    const int nb = wave * nq, ne = min(nb + nq, N);
    const float* wp = w + (int64_t)b * N * ldw + f;
    const float* mp = mask ? mask + ((int64_t)b * n_src + s) * N * ldm + f : nullptr;
#pragma unroll 4
    for (int n = nb; n < ne; ++n) {
        float h = fv ? wp[(int64_t)n * ldw] : 0.f;
        if (mp && fv) h *= mp[(int64_t)n * ldm];
        const float* dr = Wd + (int64_t)n * L;
#pragma unroll
        for (int l = 0; l < ML; ++l)
            if (l < L) acc[l] += dr[l] * h;
    }
#pragma unroll
    for (int l = 0; l < ML; ++l) part[wave][lane * (ML + 1) + l] = acc[l];
    __syncthreads();
    // overlap-add: sample tau collects frame fr, tap tau - fr*S
    const int Tpad = (Tf - 1) * S + L;
    const bool last = (f0 + TTo >= Tf);
    const int tau0 = f0 * S;
    const int tau1 = last ? Tpad : (f0 + TTo) * S;
    for (int tau = tau0 + (int)threadIdx.x; tau < tau1; tau += NTH) {
        float v = 0.f;
        int fr = tau / S;                  // newest contributing frame
        for (int r = 0; r <= R1; ++r, --fr) {
            const int l = tau - fr * S;
            const int li = fr - (f0 - R1);
            if (l < L && fr >= 0 && fr < Tf && li >= 0 && li < FT) {
                const int o = li * (ML + 1) + l;
                v += (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
            }
        }
        const int to = tau - trim_left;
        if (to >= 0 && to < Tout) out[(int64_t)b * out_bs + (int64_t)s * out_ss + to] = v;
    }
}

// ------------------------------------------------------------------------------------------ shared phase B
// V[ns][FT rows n_local][LDV] holds values per (channel, frame); F[ns][FT][ML] the matching frames of the
// time-domain signal.  Thread (n_local, q) adds sum_f V[n][f] * F[f][l] over its quarter of the frames to acc.
// corr_flush folds the four quarters through LDS (scratch: 4 x 64 x ML floats) and issues one atomic per (n, l).
template <int ML, int NS>
__device__ __forceinline__ void corr_accumulate(const float* V, const float* F, int L, float (&acc)[ML]) {
    const int nl = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int s = 0; s < NS; ++s) {
        const float* Vs = V + (size_t)s * FT * LDV + nl * LDV;
        const float* Fs = F + (size_t)s * FT * ML;
        for (int fl = q * (FT / 4); fl < (q + 1) * (FT / 4); ++fl) {
            const float h = Vs[fl];
#pragma unroll
            for (int l = 0; l < ML; ++l)
                if (l < L) acc[l] += h * Fs[fl * ML + l];
        }
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
template <int ML, int NS>
__global__ __launch_bounds__(NTH) void maskdecode_bwd_kernel(const float* __restrict__ dout, int64_t dout_ss,
                                                             int64_t dout_bs, int Tout, int trim_left,
                                                             const float* __restrict__ w, int64_t ldw,
                                                             const float* __restrict__ mask, int64_t ldm, int N,
                                                             int Tf, const float* __restrict__ Wd, int L, int S,
                                                             int sigmoid_bwd, float* __restrict__ dz,
                                                             int64_t lddz, float* __restrict__ dw_dec,
                                                             int64_t lddw, float* dWd, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* V = lds;                               // [NS][64][LDV]   (>= 4 * 64 * ML floats: also corr_flush scratch)
    float* F = lds + (size_t)NS * FT * LDV;       // [NS][FT][ML]
    const int b = blockIdx.z, n0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[ML];
#pragma unroll
    for (int l = 0; l < ML; ++l) acc[l] = 0.f;
    const int ntiles = (Tf + FT - 1) / FT;
    const int tile_end = min((int)(blockIdx.x + 1) * tiles_per_block, ntiles);
  for (int tile = blockIdx.x * tiles_per_block; tile < tile_end; ++tile) {
    const int f0 = tile * FT;
    const int f = f0 + lane;
    const bool fv = f < Tf;
    float dx[NS][ML];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int l = 0; l < ML; ++l) {
            const int to = f * S + l - trim_left;
            dx[s][l] = (fv && l < L && to >= 0 && to < Tout) ? dout[(int64_t)b * dout_bs + (int64_t)s * dout_ss + to]
                                                            : 0.f;
        }
    if (wave == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int l = 0; l < ML; ++l) F[((size_t)s * FT + lane) * ML + l] = dx[s][l];
    }
    for (int i = 0; i < 16; ++i) {
        const int nl = wave * 16 + i, n = n0 + nl;
        const bool nv = n < N;
        const float wv = (nv && fv) ? w[((int64_t)b * N + n) * ldw + f] : 0.f;
        float dwsum = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float dwh = 0.f;
            if (nv) {
                const float* dr = Wd + (int64_t)n * L;
#pragma unroll
                for (int l = 0; l < ML; ++l)
                    if (l < L) dwh += dr[l] * dx[s][l];
            }
            float m = 1.f;
            if (mask && nv && fv) m = mask[(((int64_t)b * NS + s) * N + n) * ldm + f];
            if (dz && nv && fv) {
                const float dzv = dwh * wv * (sigmoid_bwd ? m * (1.f - m) : 1.f);
                dz[(((int64_t)b * NS + s) * N + n) * lddz + f] = dzv;
            }
            dwsum += dwh * m;
            V[((size_t)s * FT + nl) * LDV + lane] = wv * m;
        }
        if (dw_dec && nv && fv) dw_dec[((int64_t)b * N + n) * lddw + f] = dwsum;
    }
    __syncthreads();
    corr_accumulate<ML, NS>(V, F, L, acc);
    __syncthreads();
  }
    corr_flush<ML>(V, acc, L, n0, N, dWd);
}

// ------------------------------------------------------------------------------------------ encoder bwd
template <int ML>
__global__ __launch_bounds__(NTH) void encoder_bwd_kernel(const float* __restrict__ x, int64_t xbs, int Tin,
                                                          int pad_left, int N, int L, int S, int relu,
                                                          const float* __restrict__ w, int Tf, int64_t ldw,
                                                          const float* __restrict__ dw_a, int64_t lddwa,
                                                          const float* __restrict__ dn, int64_t lddn,
                                                          const float* mean, const float* rstd, const float* c1,
                                                          const float* c2, int64_t ssb, int64_t sst,
                                                          const float* __restrict__ gamma, float* dW,
                                                          float* dgamma, float* dbeta, float* dwt_out,
                                                          int64_t lddwt, int tiles_per_block) {
    __shared__ float V[FT * LDV];                 // 64 x 65 >= 4 * 64 * ML floats: also corr_flush scratch
    __shared__ float F[FT * ML];
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
    if (wave == 0) {
#pragma unroll
        for (int l = 0; l < ML; ++l) {
            const int tau = f * S + l - pad_left;
            F[lane * ML + l] = (fv && l < L && tau >= 0 && tau < Tin) ? x[(int64_t)b * xbs + tau] : 0.f;
        }
    }
    float mu = 0.f, rs = 1.f, k1 = 0.f, k2 = 0.f;
    if (dn && fv) {
        const int64_t si = (int64_t)b * ssb + (int64_t)f * sst;
        mu = mean[si]; rs = rstd[si]; k1 = c1[si]; k2 = c2[si];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int nl = wave * 16 + i, n = n0 + nl;
        const bool ok = (n < N) && fv;
        float dwt = 0.f, gx = 0.f, gd = 0.f;
        if (ok) {
            const float wv = w[((int64_t)b * N + n) * ldw + f];
            if (dw_a) dwt = dw_a[((int64_t)b * N + n) * lddwa + f];
            if (dn) {
                const float d = dn[((int64_t)b * N + n) * lddn + f];
                const float xc = wv - mu;
                dwt += gamma[n] * d * rs + k1 + xc * k2;
                gx = d * xc * rs;
                gd = d;
            }
            if (relu && !(wv > 0.f)) dwt = 0.f;
            if (dwt_out) dwt_out[((int64_t)b * N + n) * lddwt + f] = dwt;
        }
        V[nl * LDV + lane] = dwt;
        gxa[i] += gx;
        gda[i] += gd;
    }
    __syncthreads();
    corr_accumulate<ML, 1>(V, F, L, acc);
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
    const int TTo = FT - R1;
    dim3 grid(ctn_cdiv(Tf, TTo), batch * n_src);
    hipStream_t s = (hipStream_t)stream;
    if (L <= 16)
        hipLaunchKernelGGL(maskdecode_fwd_kernel<16>, grid, dim3(NTH), 0, s, w, ldw, mask, ldm, n_src, N, Tf, Wd, L,
                           S, out, out_ss, out_bs, Tout, trim_left, R1);
    else
        hipLaunchKernelGGL(maskdecode_fwd_kernel<32>, grid, dim3(NTH), 0, s, w, ldw, mask, ldm, n_src, N, Tf, Wd, L,
                           S, out, out_ss, out_bs, Tout, trim_left, R1);
    CTN_CHECK_LAUNCH("ctn_maskdecode_fwd");
    return CTN_OK;
}

int ctn_maskdecode_bwd(const float* dout, int64_t dout_ss, int64_t dout_bs, int Tout, int trim_left, const float* w,
                       int64_t ldw, const float* mask, int64_t ldm, int batch, int n_src, int N, int Tf,
                       const float* Wd, int L, int S, int sigmoid_bwd, float* dz, int64_t lddz, float* dw_dec,
                       int64_t lddw, float* dWd, void* stream) {
    CTN_REQUIRE(dout && w && Wd && dWd && batch > 0 && n_src >= 1 && n_src <= 3 && N > 0 && Tf > 0 && L > 0 &&
                    L <= 16 && S > 0,
                "ctn_maskdecode_bwd: bad arguments (L <= 16, n_src <= 3)");
    CTN_REQUIRE(ldw >= Tf && (!mask || ldm >= Tf) && (!dz || lddz >= Tf) && (!dw_dec || lddw >= Tf),
                "ctn_maskdecode_bwd: ld < Tf");
    CTN_REQUIRE(!mask || dz, "ctn_maskdecode_bwd: dz required with a mask");
    const int ntiles = ctn_cdiv(Tf, FT);
    const int tpb = tiles_per_block(ntiles, ctn_cdiv(N, 64) * batch);
    dim3 grid(ctn_cdiv(ntiles, tpb), ctn_cdiv(N, 64), batch);
    hipStream_t s = (hipStream_t)stream;
    const size_t shmem = ((size_t)n_src * FT * LDV + (size_t)n_src * FT * 16) * sizeof(float);
#define CTN_MD_CASE(NS)                                                                                         \
    case NS:                                                                                                    \
        hipLaunchKernelGGL((maskdecode_bwd_kernel<16, NS>), grid, dim3(NTH), shmem, s, dout, dout_ss, dout_bs, Tout, \
                           trim_left, w, ldw, mask, ldm, N, Tf, Wd, L, S, sigmoid_bwd, dz, lddz, dw_dec, lddw, dWd, tpb);  \
        break;
    switch (n_src) { CTN_MD_CASE(1) CTN_MD_CASE(2) CTN_MD_CASE(3) }
#undef CTN_MD_CASE
    CTN_CHECK_LAUNCH("ctn_maskdecode_bwd");
    return CTN_OK;
}

int ctn_encoder_bwd(const float* x, int64_t xbs, int Tin, int pad_left, int batch, int N, int L, int S, int relu,
                    const float* w, int Tf, int64_t ldw, const float* dw_a, int64_t lddwa, const float* dn,
                    int64_t lddn, const float* mean, const float* rstd, const float* c1, const float* c2, int64_t ssb,
                    int64_t sst, const float* gamma, float* dW, float* dgamma, float* dbeta, float* dwt_out,
                    int64_t lddwt, void* stream) {
    CTN_REQUIRE(x && w && dW && batch > 0 && N > 0 && L > 0 && L <= 16 && S > 0 && Tf > 0 && Tin > 0,
                "ctn_encoder_bwd: bad arguments (L <= 16)");
    CTN_REQUIRE(ldw >= Tf && (!dw_a || lddwa >= Tf) && (!dn || lddn >= Tf) && (!dwt_out || lddwt >= Tf),
                "ctn_encoder_bwd: ld < Tf");
    CTN_REQUIRE(dw_a || dn, "ctn_encoder_bwd: no incoming gradient");
    if (dn) CTN_REQUIRE(mean && rstd && c1 && c2 && gamma && dgamma && dbeta, "ctn_encoder_bwd: norm arguments");
    const int ntiles = ctn_cdiv(Tf, FT);
    const int tpb = tiles_per_block(ntiles, ctn_cdiv(N, 64) * batch);
    dim3 grid(ctn_cdiv(ntiles, tpb), ctn_cdiv(N, 64), batch);
    hipLaunchKernelGGL(encoder_bwd_kernel<16>, grid, dim3(NTH), 0, (hipStream_t)stream, x, xbs, Tin, pad_left, N, L,
                       S, relu, w, Tf, ldw, dw_a, lddwa, dn, lddn, mean, rstd, c1, c2, ssb, sst, gamma, dW, dgamma,
                       dbeta, dwt_out, lddwt, tpb);
    CTN_CHECK_LAUNCH("ctn_encoder_bwd");
    return CTN_OK;
}

}  // extern "C"
