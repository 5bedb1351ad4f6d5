// Persistent LSTM recurrence kernels (forward and backward through time), gfx950.
//
// The dual-path RNN blocks run an LSTM over hundreds of time steps on thousands of short, independent sequences
// (src/models/dprnn.py:51-148: hidden size 128, 250 steps x ~2000 sequences inside chunks, ~500 steps x 1000 sequences
// across chunks).  A library LSTM launches a small GEMM and a pointwise kernel per time step -- some 30 000 launches per
// training step at the recipe's defaults.  Here ONE launch walks all time steps:
//
//   * the part of a step that does not depend on the previous step -- W_ih x_t + b_ih + b_hh for every t -- is hoisted
//     out as one large GEMM by the caller (`xproj`);
//   * a workgroup owns 16 sequences of one direction and keeps them for the whole walk; its four waves split the hidden
//     units (wave w: units [HP/4 w, HP/4 (w+1)), all four gates of each), and every wave holds ITS rows of the recurrent
//     matrix W_hh in registers for the whole kernel as MFMA A-fragments (hidden size 128: 256 registers per lane -- the
//     kernel runs one wave per SIMD on purpose);
//   * per step: h_{t-1} (16 columns, from LDS) is the B operand of v_mfma_f32_16x16x4_f32, the accumulators start from
//     xproj_t (prefetched one step ahead), and the accumulator layout -- four consecutive hidden units of one sequence
//     per lane, the same in all four gate tiles -- lets the cell update c = f c + i g, h = o tanh(c) run in registers;
//     h_t goes to LDS for the next step and, with the gate values the backward pass needs, to HBM as 16-byte stores.
//   * one barrier per step (forward), two (backward: the gate gradients go through LDS to become the B operand of
//     W_hh^T).
// Exact fp32 arithmetic (f32 MFMA), so the recurrence accumulates no error beyond fp32 rounding over hundreds of steps.
//
// Hidden sizes up to 128 (multiple of 4) are padded to HP = 64 or 128 with zero weights: padded units stay at
// c = h = 0 and are never stored.
#include <stdlib.h>

#include "ctn_common.h"

namespace {

constexpr int NTH = 256;
constexpr int NSEQ = 16;                  // sequences per workgroup = the N of the 16x16x4 MFMA

// Gate non-linearities on the hardware exponential and reciprocal (v_exp_f32, v_rcp_f32: about 1 ulp each).  A step of
// the recurrence evaluates 40 of them per lane on its critical path -- nothing else runs on a SIMD that holds one wave --
// and the library versions cost about four times as many instructions.  Absolute error below 3e-7, far inside the fp32
// noise the recurrence carries anyway (the kernel tests hold h and every gradient to 2e-5 / 5e-5 of float64).
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {          // 2 sigmoid(2x) - 1
    return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)), -1.f);
}

// ------------------------------------------------------------------------------------------ forward
// NT: 16-unit tiles per gate and wave (HP = 64 NT padded hidden units)
template <int NT>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_kernel(
    const float* __restrict__ xproj, const float* __restrict__ whh, float* __restrict__ hout, float* __restrict__ save,
    const int N, const int T, const int H, const int D) {
    constexpr int HP = 64 * NT, KS = HP / 4, UPW = 16 * NT;          // padded units, k-steps, units per wave
    __shared__ float hbuf[2][HP * NSEQ];                               // h_{t-1}: [unit][sequence], double-buffered
    const int dir = blockIdx.y, n0 = blockIdx.x * NSEQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, rg = lane >> 4;
    const int n = n0 + col;
    const bool nv = n < N;
    const int nc = min(n, N - 1);
    const float* W = whh + (size_t)dir * 4 * H * H;

    // A fragments of this wave's gate rows: tile (gate g, sub-tile s), k-step kk: A[m = lane & 15][k = 4 kk + (lane >> 4)]
    float wreg[4][NT][KS];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < NT; ++s)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int u = wave * UPW + 16 * s + (lane & 15), k = 4 * kk + rg;
                wreg[g][s][kk] = (u < H && k < H) ? W[((size_t)g * H + u) * H + k] : 0.f;
            }
    // this lane's units of sub-tile s: ub[s] .. ub[s] + 3 (C/D layout: row = 4 (lane >> 4) + reg, column = lane & 15)
    int ub[NT];
    bool uv[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        ub[s] = wave * UPW + 16 * s + 4 * rg;
        uv[s] = ub[s] < H;                                            // H % 4 == 0: the four units are valid together
    }
    float c[NT][4];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) c[s][e] = 0.f;
    for (int i = tid; i < HP * NSEQ; i += NTH) hbuf[0][i] = 0.f;

    const int64_t row_x = (int64_t)D * 4 * H;                          // floats per (sequence, step) of xproj
    auto load_x = [&](int t, f32x4 (&xp)[4][NT]) {
        const float* base = xproj + ((int64_t)nc * T + t) * row_x + (int64_t)dir * 4 * H;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                const f32x4 v = ctn_ld4(base + g * H + (uv[s] ? ub[s] : 0));      // clamped address, masked value
                xp[g][s] = uv[s] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };
    f32x4 xp[4][NT];
    load_x(dir == 0 ? 0 : T - 1, xp);
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? step : T - 1 - step;
        f32x4 acc[4][NT];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[g][s] = xp[g][s];
        if (step + 1 < T) load_x(dir == 0 ? t + 1 : t - 1, xp);       // in flight during this step's MFMAs
        const float* hb = hbuf[step & 1];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const float b = hb[(4 * kk + rg) * NSEQ + col];            // B[k = 4 kk + (lane >> 4)][n = lane & 15]
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int s = 0; s < NT; ++s)
                    acc[g][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[g][s][kk], b, acc[g][s], 0, 0, 0);
        }
        float* hn = hbuf[(step + 1) & 1];
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            f32x4 iv, fv, gv, ov, cv, hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#ifndef CTN_DBG_LSTM_NOPOINT
                iv[e] = sigmoidf_(acc[0][s][e]);
                fv[e] = sigmoidf_(acc[1][s][e]);
                gv[e] = tanhf_(acc[2][s][e]);
                ov[e] = sigmoidf_(acc[3][s][e]);
                c[s][e] = fv[e] * c[s][e] + iv[e] * gv[e];
                cv[e] = c[s][e];
                hv[e] = ov[e] * tanhf_(c[s][e]);
#else
                iv[e] = acc[0][s][e]; fv[e] = acc[1][s][e]; gv[e] = acc[2][s][e]; ov[e] = acc[3][s][e];
                c[s][e] = fv[e] * c[s][e] + iv[e] * gv[e];
                cv[e] = c[s][e];
                hv[e] = ov[e] * 0.001f;
#endif
                hn[(ub[s] + e) * NSEQ + col] = hv[e];
            }
#ifndef CTN_DBG_LSTM_NOSTORE
            if (nv && uv[s]) {
#else
            if (nv && uv[s] && t == 12345678) {
#endif
                const int64_t o = ((int64_t)n * T + t) * D + dir;
                ctn_st4(hout + o * H + ub[s], hv);
                float* sv = save + o * 5 * H + ub[s];
                ctn_st4(sv, iv);
                ctn_st4(sv + H, fv);
                ctn_st4(sv + 2 * H, gv);
                ctn_st4(sv + 3 * H, ov);
                ctn_st4(sv + 4 * H, cv);
            }
        }
        __syncthreads();              // h_t complete in hn; every wave is done reading hb
    }
}

// ------------------------------------------------------------------------------------------ forward, bf16 matrix unit
// The same walk with the recurrent product on v_mfma_f32_16x16x32_bf16 in the exact three-piece arithmetic of the
// pointwise GEMMs (CTN_ARITH_BF16X3): W_hh and h are split as x = hi + mid + lo, each piece the next eight significant
// bits of the fp32 value, and the six piece products above 2^-22 of the largest are accumulated in fp32 on top of xproj_t.
// A step's matrix work drops from 256 instructions of 32 cycles to 192 of 16.  W_hh's hi and mid pieces stay in
// registers as A fragments (256 registers per lane, as the fp32 weights did); the lo pieces, used by one product in six,
// live in LDS (128 KB at hidden size 128) and are read per step; h_t is split by the wave that computes it and written to
// LDS as B fragments (three 16-byte reads per 32 units and wave in the next step instead of 32 four-byte reads).
typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));

// Two floats -> the packed pieces of both: hi, mid, lo (low half-word = first value)
__device__ __forceinline__ void lstm_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned ru0 = __float_as_uint(r0), ru1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(ru1, ru0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(ru0 & 0xffff0000u), q1 = r1 - __uint_as_float(ru1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

union lstm_frag {
    lbf16x8 v;
    unsigned u[4];
};

// FULL: H fills the padded tile (H == 64 NT); the stores then need no condition: a padded sequence repeats the block's last
// real one, input and all, and writes the same values to the same places.  Without a branch around the stores the
// compiler counts them exactly when it waits for the prefetched input of the next step, so that wait does not include them.
template <int NT, bool FULL>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_b3_kernel(
    const float* __restrict__ xproj, const float* __restrict__ whh, float* __restrict__ hout, float* __restrict__ save,
    const int N, const int T, const int H, const int D) {
    constexpr int HP = 64 * NT, KS = HP / 32, UPW = 16 * NT;          // padded units, 32-unit k-steps, units per wave
    constexpr int WL16 = 4 * 4 * NT * KS * 64;                         // 16-byte fragments of the lo plane (all waves)
    constexpr int HB16 = 3 * KS * 64;                                  // ... of one buffer of h fragments
    extern __shared__ __attribute__((aligned(16))) unsigned char lstm_lds[];
    uint4* const wl = reinterpret_cast<uint4*>(lstm_lds);              // [wave][gate][sub-tile][k-step][lane]
    uint4* const hp = wl + WL16;                                       // [buffer][piece][k-step][slot = 16 kgroup + sequence]
    const int dir = blockIdx.y, n0 = blockIdx.x * NSEQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, rg = lane >> 4;
    const int n = n0 + col;
    const bool nv = n < N;
    const int nc = min(n, N - 1);
    const float* W = whh + (size_t)dir * 4 * H * H;

    // A fragments of this wave's gate rows: tile (gate g, sub-tile s), k-step ks: A[m = lane & 15][k = 32 ks + 8 (lane >> 4) + j]
    lstm_frag wh[4][NT][KS], wm[4][NT][KS];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < NT; ++s)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int u = wave * UPW + 16 * s + (lane & 15);
                lstm_frag lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 32 * ks + 8 * rg + 2 * j;
                    const float w0 = (u < H && k < H) ? W[((size_t)g * H + u) * H + k] : 0.f;
                    const float w1 = (u < H && k + 1 < H) ? W[((size_t)g * H + u) * H + k + 1] : 0.f;
                    lstm_split2(w0, w1, wh[g][s][ks].u[j], wm[g][s][ks].u[j], lo.u[j]);
                }
                wl[(((wave * 4 + g) * NT + s) * KS + ks) * 64 + lane] = uint4{lo.u[0], lo.u[1], lo.u[2], lo.u[3]};
            }
    // The weights take the accumulation registers, all 256 of them (the matrix instruction reads its A operand from
    // either file); left to itself the allocator fills the vector registers with them and then shuffles everything
    // else of the step through the accumulation registers, waiting for each prefetched load right behind its issue.
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < NT; ++s)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                asm volatile("" : "+a"(wh[g][s][ks].v));
                asm volatile("" : "+a"(wm[g][s][ks].v));
            }
    int ub[NT];
    bool uv[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        ub[s] = wave * UPW + 16 * s + 4 * rg;
        uv[s] = ub[s] < H;
    }
    float c[NT][4];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) c[s][e] = 0.f;
    for (int i = tid; i < HB16; i += NTH) hp[i] = uint4{0u, 0u, 0u, 0u};          // h_{-1} = 0

    const int64_t row_x = (int64_t)D * 4 * H;
    auto load_x = [&](int t, f32x4 (&xp)[4][NT]) {
        const float* base = xproj + ((int64_t)nc * T + t) * row_x + (int64_t)dir * 4 * H;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                xp[g][s] = ctn_ld4(base + g * H + (uv[s] ? ub[s] : 0));           // units past H: zeroed where it is used
            }
    };
    f32x4 xp[4][NT];
    load_x(dir == 0 ? 0 : T - 1, xp);
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? step : T - 1 - step;
        f32x4 acc[4][NT];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[g][s] = f32x4{0.f, 0.f, 0.f, 0.f};
        // next step's input projection, a whole step ahead of its use (the last step reads its own row again)
        f32x4 xn[4][NT];
#ifndef CTN_DBG_LSTM_NO_LOADX
        load_x(step + 1 < T ? (dir == 0 ? t + 1 : t - 1) : t, xn);
#else
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < NT; ++s) xn[g][s] = xp[g][s];
#endif
        const uint4* hb = hp + (step & 1) * HB16;
        // h fragments of k-step ks + 1 are read from LDS while the matrix unit works on k-step ks; the lo plane of W_hh
        // feeds the first product only, so one set of registers serves: it is refilled right behind that product
        lstm_frag bh[2], bm[2], bl[2], lo[4][NT];
        auto read_h = [&](int ks, int b) {
            const uint4 v0 = hb[(0 * KS + ks) * 64 + lane], v1 = hb[(1 * KS + ks) * 64 + lane], v2 = hb[(2 * KS + ks) * 64 + lane];
            bh[b].u[0] = v0.x, bh[b].u[1] = v0.y, bh[b].u[2] = v0.z, bh[b].u[3] = v0.w;
            bm[b].u[0] = v1.x, bm[b].u[1] = v1.y, bm[b].u[2] = v1.z, bm[b].u[3] = v1.w;
            bl[b].u[0] = v2.x, bl[b].u[1] = v2.y, bl[b].u[2] = v2.z, bl[b].u[3] = v2.w;
        };
        auto read_lo = [&](int ks) {
#ifndef CTN_DBG_LSTM_NO_LO
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    const uint4 v = wl[(((wave * 4 + g) * NT + s) * KS + ks) * 64 + lane];
                    lo[g][s].u[0] = v.x, lo[g][s].u[1] = v.y, lo[g][s].u[2] = v.z, lo[g][s].u[3] = v.w;
                }
#endif
        };
        read_h(0, 0);
        read_lo(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cb = ks & 1;
            // smallest products first; the eight tiles alternate, so consecutive instructions are independent
#define CTN_LSTM_MFMA(A_, B_)                                                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int s = 0; s < NT; ++s)                  \
        acc[g][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[g][s] CTN_LSTM_KS .v, B_[cb].v, acc[g][s], 0, 0, 0)
            __builtin_amdgcn_sched_barrier(0);
#define CTN_LSTM_KS
#ifndef CTN_DBG_LSTM_NO_LO
            CTN_LSTM_MFMA(lo, bh);
#endif
#undef CTN_LSTM_KS
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < KS) {
                read_lo(ks + 1);
                read_h(ks + 1, cb ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#define CTN_LSTM_KS [ks]
#ifndef CTN_DBG_LSTM_NO_MMA
            CTN_LSTM_MFMA(wh, bl);
            CTN_LSTM_MFMA(wm, bm);
            CTN_LSTM_MFMA(wm, bh);
            CTN_LSTM_MFMA(wh, bm);
#endif
            CTN_LSTM_MFMA(wh, bh);
#undef CTN_LSTM_KS
#undef CTN_LSTM_MFMA
        }
        __builtin_amdgcn_sched_barrier(0);
        uint4* hn = hp + ((step + 1) & 1) * HB16;
        // the input projection joins after the matrix work
        f32x4 xv[4][NT];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < NT; ++s) xv[g][s] = (FULL || uv[s]) ? xp[g][s] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            f32x4 iv, fv, gv, ov, cv, hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#ifndef CTN_DBG_LSTM_NO_GATE
                iv[e] = sigmoidf_(acc[0][s][e] + xv[0][s][e]);
                fv[e] = sigmoidf_(acc[1][s][e] + xv[1][s][e]);
                gv[e] = tanhf_(acc[2][s][e] + xv[2][s][e]);
                ov[e] = sigmoidf_(acc[3][s][e] + xv[3][s][e]);
                c[s][e] = fv[e] * c[s][e] + iv[e] * gv[e];
                cv[e] = c[s][e];
                hv[e] = ov[e] * tanhf_(c[s][e]);
#else
                iv[e] = acc[0][s][e]; fv[e] = acc[1][s][e]; gv[e] = acc[2][s][e]; ov[e] = acc[3][s][e];
                c[s][e] = 0.5f * c[s][e] + 0.01f * iv[e] * gv[e];
                cv[e] = c[s][e];
                hv[e] = 0.01f * ov[e] * c[s][e];
#endif
            }
            // this lane's four units of h_t as half a B fragment per piece: units ub[s] .. ub[s] + 3 of sequence `col`
            {
                unsigned h0, m0, l0, h1, m1, l1;
                lstm_split2(hv[0], hv[1], h0, m0, l0);
                lstm_split2(hv[2], hv[3], h1, m1, l1);
                const int u = ub[s], ks = u >> 5, slot = ((u & 31) >> 3) * 16 + col, half8 = (u & 7) >> 2;
                uint2* d0 = reinterpret_cast<uint2*>(hn + (0 * KS + ks) * 64 + slot) + half8;
                uint2* d1 = reinterpret_cast<uint2*>(hn + (1 * KS + ks) * 64 + slot) + half8;
                uint2* d2 = reinterpret_cast<uint2*>(hn + (2 * KS + ks) * 64 + slot) + half8;
                *d0 = uint2{h0, h1};
                *d1 = uint2{m0, m1};
                *d2 = uint2{l0, l1};
            }
#ifdef CTN_DBG_LSTM_NO_STORE
            if (nv && uv[s] && step == T - 1) {
#else
            if (FULL || (nv && uv[s])) {
#endif
                const int64_t o = ((int64_t)(FULL ? nc : n) * T + t) * D + dir;
                ctn_st4(hout + o * H + ub[s], hv);
                float* sv = save + o * 5 * H + ub[s];
                ctn_st4(sv, iv);
                ctn_st4(sv + H, fv);
                ctn_st4(sv + 2 * H, gv);
                ctn_st4(sv + 3 * H, ov);
                ctn_st4(sv + 4 * H, cv);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < NT; ++s) xp[g][s] = xn[g][s];
        __syncthreads();              // h_t complete in hn; every wave is done reading hb
    }
}

// Bias gradient of one direction: each lane's per-step sums (its sequence, its four units of every gate) are added over
// the block's 16 sequences (lanes that differ in their low four bits) and one lane in 16 adds the block's share to
// dbias[gate * H + unit] (vector-memory float atomics).  Padded sequences do not take part.
template <int NT>
__device__ __forceinline__ void lstm_bias_grad(f32x4 (&dbs)[NT][4], float* __restrict__ dbias, const bool nv,
                                               const bool (&uv)[NT], const int (&ub)[NT], const int H, const int lane) {
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = nv ? dbs[s][g][e] : 0.f;
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                if ((lane & 15) == 0 && uv[s]) atomicAdd(dbias + g * H + ub[s] + e, v);
            }
}

// ------------------------------------------------------------------------------------------ backward through time
template <int NT>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_kernel(
    const float* __restrict__ dhout, const float* __restrict__ save, const float* __restrict__ whh,
    float* __restrict__ dgates, float* __restrict__ dbias, const int N, const int T, const int H, const int D) {
    constexpr int HP = 64 * NT, ROWS = 4 * HP, KS = ROWS / 4, UPW = 16 * NT;
    __shared__ float dgbuf[ROWS * NSEQ];                               // gate gradients of this step: [padded row][sequence]
    const int dir = blockIdx.y, n0 = blockIdx.x * NSEQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, rg = lane >> 4;
    const int n = n0 + col;
    const bool nv = n < N;
    const int nc = min(n, N - 1);
    const float* W = whh + (size_t)dir * 4 * H * H;

    // A fragments of W_hh^T for this wave's output units: A[m = unit][k = padded gate row g HP + u']
    float wt[NT][KS];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int u = wave * UPW + 16 * s + (lane & 15);
            const int r = 4 * kk + rg, g = r / HP, ur = r % HP;
            wt[s][kk] = (u < H && ur < H) ? W[((size_t)g * H + ur) * H + u] : 0.f;
        }
    int ub[NT];
    bool uv[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        ub[s] = wave * UPW + 16 * s + 4 * rg;
        uv[s] = ub[s] < H;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 dh_rec[NT], dc_carry[NT], dbs[NT][4];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        dh_rec[s] = dc_carry[s] = zero4;
#pragma unroll
        for (int g = 0; g < 4; ++g) dbs[s][g] = zero4;
    }

    // what one step reads: the five saved tensors at t, the cell state of the step before t in forward order, dL/dh_t
    struct StepIn {
        f32x4 sv[5][NT], cp[NT], dho[NT];
    };
    auto load_step = [&](int t, StepIn& in) {
        const int tp = dir == 0 ? t - 1 : t + 1;                       // forward-order predecessor
        const bool has_prev = tp >= 0 && tp < T;
        const int64_t o = ((int64_t)nc * T + t) * D + dir;
        const int64_t op = ((int64_t)nc * T + (has_prev ? tp : t)) * D + dir;
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int u0 = uv[s] ? ub[s] : 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const f32x4 v = ctn_ld4(save + o * 5 * H + q * H + u0);
                in.sv[q][s] = uv[s] ? v : zero4;
            }
            const f32x4 cpv = ctn_ld4(save + op * 5 * H + 4 * H + u0);
            in.cp[s] = (uv[s] && has_prev) ? cpv : zero4;
            const f32x4 d = ctn_ld4(dhout + o * H + u0);
            in.dho[s] = uv[s] ? d : zero4;
        }
    };
    StepIn cur;
    load_step(dir == 0 ? T - 1 : 0, cur);

    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? T - 1 - step : step;                  // the reverse of the forward walk
        StepIn nxt = cur;
        if (step + 1 < T) load_step(dir == 0 ? t - 1 : t + 1, nxt);   // in flight during this step
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            f32x4 di, df, dg, dob;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float iv = cur.sv[0][s][e], fv = cur.sv[1][s][e], gv = cur.sv[2][s][e], ov = cur.sv[3][s][e];
                const float tc = tanhf_(cur.sv[4][s][e]);
                const float dh = cur.dho[s][e] + dh_rec[s][e];
                const float dc = dh * ov * (1.f - tc * tc) + dc_carry[s][e];
                dob[e] = dh * tc * ov * (1.f - ov);
                di[e] = dc * gv * iv * (1.f - iv);
                dg[e] = dc * iv * (1.f - gv * gv);
                df[e] = dc * cur.cp[s][e] * fv * (1.f - fv);
                dc_carry[s][e] = dc * fv;
                dgbuf[(0 * HP + ub[s] + e) * NSEQ + col] = di[e];
                dgbuf[(1 * HP + ub[s] + e) * NSEQ + col] = df[e];
                dgbuf[(2 * HP + ub[s] + e) * NSEQ + col] = dg[e];
                dgbuf[(3 * HP + ub[s] + e) * NSEQ + col] = dob[e];
            }
            dbs[s][0] += di;
            dbs[s][1] += df;
            dbs[s][2] += dg;
            dbs[s][3] += dob;
            if (nv && uv[s]) {
                float* o = dgates + (((int64_t)n * T + t) * D + dir) * 4 * H + ub[s];
                ctn_st4(o, di);
                ctn_st4(o + H, df);
                ctn_st4(o + 2 * H, dg);
                ctn_st4(o + 3 * H, dob);
            }
        }
        __syncthreads();              // the step's gate gradients are complete in LDS
        // dL/dh of the forward-order predecessor through the recurrence: W_hh^T . dgates; two partial sums per tile keep
        // four independent MFMA chains in flight for NT = 2
        f32x4 acc[NT][2];
#pragma unroll
        for (int s = 0; s < NT; ++s) acc[s][0] = acc[s][1] = zero4;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const float b = dgbuf[(4 * kk + rg) * NSEQ + col];
#pragma unroll
            for (int s = 0; s < NT; ++s)
                acc[s][kk & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[s][kk], b, acc[s][kk & 1], 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < NT; ++s) dh_rec[s] = acc[s][0] + acc[s][1];
        cur = nxt;
        __syncthreads();              // every wave is done reading dgbuf before the next step overwrites it
    }
    if (dbias) lstm_bias_grad<NT>(dbs, dbias + (size_t)dir * 4 * H, nv, uv, ub, H, lane);
}

// ------------------------------------------------------------------------------------------ backward, bf16 matrix unit
// The reverse walk with dL/dh_{t-1} = W_hh^T . dgates_t on v_mfma_f32_16x16x32_bf16, same exact three-piece arithmetic as
// the forward kernel.  W_hh^T's hi and mid pieces fill the accumulation registers as A fragments; of the lo pieces the
// first KR k-steps stay in vector registers and the rest live in LDS (96 KB at hidden size 128), because the step's gate
// gradients, split by the lanes that compute them, need 48 KB of LDS as B fragments.  One set of registers holds a
// step's saved tensors: the next step's are requested right behind the gate arithmetic and arrive during the matrix phase.
template <int NT, bool FULL>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_b3_kernel(
    const float* __restrict__ dhout, const float* __restrict__ save, const float* __restrict__ whh,
    float* __restrict__ dgates, float* __restrict__ dbias, const int N, const int T, const int H, const int D) {
    constexpr int HP = 64 * NT, ROWS = 4 * HP, KS = ROWS / 32, UPW = 16 * NT;
    constexpr int KR = NT == 2 ? 4 : 0;                                // k-steps whose lo fragments stay in registers
    constexpr int KL = KS - KR;
    constexpr int WL16 = 4 * NT * KL * 64;                             // 16-byte fragments of the lo plane in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char lstm_lds[];
    uint4* const wl = reinterpret_cast<uint4*>(lstm_lds);              // [wave][sub-tile][k-step - KR][lane]
    uint4* const dgp = wl + WL16;                                      // [piece][k-step][slot = 16 kgroup + sequence]
    const int dir = blockIdx.y, n0 = blockIdx.x * NSEQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, rg = lane >> 4;
    const int n = n0 + col;
    const bool nv = n < N;
    const int nc = min(n, N - 1);
    const float* W = whh + (size_t)dir * 4 * H * H;

    // A fragments of W_hh^T: tile s (units wave UPW + 16 s + (lane & 15)), k-step ks: k = padded gate row 32 ks + 8 (lane >> 4) + j
    lstm_frag wh[NT][KS], wm[NT][KS], lor[NT][KR > 0 ? KR : 1];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int u = wave * UPW + 16 * s + (lane & 15);
            lstm_frag lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 32 * ks + 8 * rg + 2 * j, g = r / HP, ur = r % HP;      // r even: r and r + 1 share the gate
                const float w0 = (u < H && ur < H) ? W[((size_t)g * H + ur) * H + u] : 0.f;
                const float w1 = (u < H && ur + 1 < H) ? W[((size_t)g * H + ur + 1) * H + u] : 0.f;
                lstm_split2(w0, w1, wh[s][ks].u[j], wm[s][ks].u[j], lo.u[j]);
            }
            if (ks < KR)
                lor[s][ks < KR ? ks : 0] = lo;
            else
                wl[((wave * NT + s) * KL + (ks - KR)) * 64 + lane] = uint4{lo.u[0], lo.u[1], lo.u[2], lo.u[3]};
        }
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            asm volatile("" : "+a"(wh[s][ks].v));
            asm volatile("" : "+a"(wm[s][ks].v));
        }
    int ub[NT];
    bool uv[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        ub[s] = wave * UPW + 16 * s + 4 * rg;
        uv[s] = ub[s] < H;
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 dh_rec[NT], dc_carry[NT], dbs[NT][4];                        // dbs: this lane's gate gradients summed over the steps
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        dh_rec[s] = dc_carry[s] = zero4;
#pragma unroll
        for (int g = 0; g < 4; ++g) dbs[s][g] = zero4;
    }

    // what one step reads: the five saved tensors at t, the cell state of the step before t in forward order, dL/dh_t.
    // Raw loads only: what must be zero (units past H, no predecessor) is zeroed where it is used, so nothing waits here.
    f32x4 sv[5][NT], cp[NT], dho[NT];
    auto load_step = [&](int t) {
        const int tp = dir == 0 ? t - 1 : t + 1;                       // forward-order predecessor
        const bool has_prev = tp >= 0 && tp < T;
        const int64_t o = ((int64_t)nc * T + t) * D + dir;
        const int64_t op = ((int64_t)nc * T + (has_prev ? tp : t)) * D + dir;
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int u0 = uv[s] ? ub[s] : 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) sv[q][s] = ctn_ld4(save + o * 5 * H + q * H + u0);
            cp[s] = ctn_ld4(save + op * 5 * H + 4 * H + u0);
            dho[s] = ctn_ld4(dhout + o * H + u0);
        }
    };
    load_step(dir == 0 ? T - 1 : 0);
    __syncthreads();                  // lo plane in place

    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? T - 1 - step : step;                  // the reverse of the forward walk
        const int tp = dir == 0 ? t - 1 : t + 1;
        const bool has_prev = tp >= 0 && tp < T;
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            f32x4 dgv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = FULL || uv[s];
                const float iv = sv[0][s][e], fv = sv[1][s][e], gv = sv[2][s][e], ov = sv[3][s][e];
                const float tc = tanhf_(sv[4][s][e]);
                const float dh = (ok ? dho[s][e] : 0.f) + dh_rec[s][e];
                const float dc = dh * ov * (1.f - tc * tc) + dc_carry[s][e];
                const float cpe = has_prev ? cp[s][e] : 0.f;
                dgv[3][e] = ok ? dh * tc * ov * (1.f - ov) : 0.f;
                dgv[0][e] = ok ? dc * gv * iv * (1.f - iv) : 0.f;
                dgv[2][e] = ok ? dc * iv * (1.f - gv * gv) : 0.f;
                dgv[1][e] = ok ? dc * cpe * fv * (1.f - fv) : 0.f;
                dc_carry[s][e] = ok ? dc * fv : 0.f;
            }
            // this lane's four units of each gate's gradient as half a B fragment per piece
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dbs[s][g] += dgv[g];
                unsigned h0, m0, l0, h1, m1, l1;
                lstm_split2(dgv[g][0], dgv[g][1], h0, m0, l0);
                lstm_split2(dgv[g][2], dgv[g][3], h1, m1, l1);
                const int r = g * HP + ub[s], ks = r >> 5, slot = ((r & 31) >> 3) * 16 + col, half8 = (r & 7) >> 2;
                *(reinterpret_cast<uint2*>(dgp + (0 * KS + ks) * 64 + slot) + half8) = uint2{h0, h1};
                *(reinterpret_cast<uint2*>(dgp + (1 * KS + ks) * 64 + slot) + half8) = uint2{m0, m1};
                *(reinterpret_cast<uint2*>(dgp + (2 * KS + ks) * 64 + slot) + half8) = uint2{l0, l1};
            }
            if (FULL || (nv && uv[s])) {
                float* o = dgates + (((int64_t)(FULL ? nc : n) * T + t) * D + dir) * 4 * H + ub[s];
                ctn_st4(o, dgv[0]);
                ctn_st4(o + H, dgv[1]);
                ctn_st4(o + 2 * H, dgv[2]);
                ctn_st4(o + 3 * H, dgv[3]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next step's tensors into the registers this step is done with (the last step reads its own again)
        load_step(step + 1 < T ? (dir == 0 ? t - 1 : t + 1) : t);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();              // the step's gate gradients are complete in LDS

        // dL/dh of the forward-order predecessor: W_hh^T . dgates; two partial sums per tile
        f32x4 acc[NT][2];
#pragma unroll
        for (int s = 0; s < NT; ++s) acc[s][0] = acc[s][1] = zero4;
        lstm_frag bh[2], bm[2], bl[2], lo[2][NT];
        auto read_b = [&](int ks, int b) {
            const uint4 v0 = dgp[(0 * KS + ks) * 64 + lane], v1 = dgp[(1 * KS + ks) * 64 + lane], v2 = dgp[(2 * KS + ks) * 64 + lane];
            bh[b].u[0] = v0.x, bh[b].u[1] = v0.y, bh[b].u[2] = v0.z, bh[b].u[3] = v0.w;
            bm[b].u[0] = v1.x, bm[b].u[1] = v1.y, bm[b].u[2] = v1.z, bm[b].u[3] = v1.w;
            bl[b].u[0] = v2.x, bl[b].u[1] = v2.y, bl[b].u[2] = v2.z, bl[b].u[3] = v2.w;
        };
        auto read_lo = [&](int ks, int b) {
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                if (ks < KR) {
                    lo[b][s] = lor[s][ks < KR ? ks : 0];
                } else {
                    const uint4 v = wl[((wave * NT + s) * KL + (ks - KR)) * 64 + lane];
                    lo[b][s].u[0] = v.x, lo[b][s].u[1] = v.y, lo[b][s].u[2] = v.z, lo[b][s].u[3] = v.w;
                }
            }
        };
        read_b(0, 0);
        read_lo(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cb = ks & 1;
            if (ks + 1 < KS) {
                read_b(ks + 1, cb ^ 1);
                read_lo(ks + 1, cb ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            // smallest products first; consecutive instructions alternate over the tiles and the two partial sums
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo[cb][s].v, bh[cb].v, acc[s][0], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s][ks].v, bl[cb].v, acc[s][1], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s][ks].v, bm[cb].v, acc[s][0], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s][ks].v, bh[cb].v, acc[s][1], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s][ks].v, bm[cb].v, acc[s][0], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s][ks].v, bh[cb].v, acc[s][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < NT; ++s) dh_rec[s] = acc[s][0] + acc[s][1];
        __syncthreads();              // every wave is done reading the fragments before the next step overwrites them
    }
    if (dbias) lstm_bias_grad<NT>(dbs, dbias + (size_t)dir * 4 * H, nv, uv, ub, H, lane);
}

bool lstm_args_ok(int N, int T, int H, int D) { return N > 0 && T > 0 && H > 0 && H <= 128 && H % 4 == 0 && (D == 1 || D == 2); }

}  // namespace

extern "C" {

int ctn_lstm_fwd(const float* xproj, const float* whh, int N, int T, int H, int D, float* hout, float* save,
                 void* stream) {
    CTN_REQUIRE(xproj && whh && hout && save, "ctn_lstm_fwd: null pointer");
    CTN_REQUIRE(lstm_args_ok(N, T, H, D), "ctn_lstm_fwd: needs hidden size <= 128 and a multiple of 4, 1 or 2 directions");
    CTN_REQUIRE(ctn_aligned16(xproj) && ctn_aligned16(hout) && ctn_aligned16(save), "ctn_lstm_fwd: 16-byte alignment");
    dim3 grid(ctn_cdiv(N, NSEQ), D);
    static const bool f32_only = [] { const char* e = getenv("CTN_LSTM_F32"); return e && e[0] == '1'; }();
    if (!f32_only) {
        // recurrent product on the bf16 matrix instruction, exact three-piece split (dynamic LDS: the lo plane of W_hh
        // and two buffers of h fragments)
        const int NTv = H <= 64 ? 1 : 2;
        const size_t shm = ((size_t)4 * 4 * NTv * (2 * NTv) * 64 + 2 * (size_t)3 * (2 * NTv) * 64) * 16;
        static bool asked[64][2] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!asked[dev][NTv - 1]) {
            const void* fns[3] = {reinterpret_cast<const void*>(lstm_fwd_b3_kernel<1, false>),
                                  reinterpret_cast<const void*>(lstm_fwd_b3_kernel<2, false>),
                                  reinterpret_cast<const void*>(lstm_fwd_b3_kernel<2, true>)};
            for (int i = NTv - 1; i < (NTv == 1 ? 1 : 3); ++i)
                CTN_REQUIRE(hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) == hipSuccess,
                            "ctn_lstm_fwd: cannot raise the dynamic LDS limit");
            asked[dev][NTv - 1] = true;
        }
        if (NTv == 1)
            hipLaunchKernelGGL((lstm_fwd_b3_kernel<1, false>), grid, dim3(NTH), shm, (hipStream_t)stream, xproj, whh, hout, save, N, T, H, D);
        else if (H == 128)
            hipLaunchKernelGGL((lstm_fwd_b3_kernel<2, true>), grid, dim3(NTH), shm, (hipStream_t)stream, xproj, whh, hout, save, N, T, H, D);
        else
            hipLaunchKernelGGL((lstm_fwd_b3_kernel<2, false>), grid, dim3(NTH), shm, (hipStream_t)stream, xproj, whh, hout, save, N, T, H, D);
        CTN_CHECK_LAUNCH("ctn_lstm_fwd");
        return CTN_OK;
    }
    if (H <= 64)
        hipLaunchKernelGGL(lstm_fwd_kernel<1>, grid, dim3(NTH), 0, (hipStream_t)stream, xproj, whh, hout, save, N, T, H, D);
    else
        hipLaunchKernelGGL(lstm_fwd_kernel<2>, grid, dim3(NTH), 0, (hipStream_t)stream, xproj, whh, hout, save, N, T, H, D);
    CTN_CHECK_LAUNCH("ctn_lstm_fwd");
    return CTN_OK;
}

int ctn_lstm_bwd(const float* dhout, const float* save, const float* whh, int N, int T, int H, int D, float* dgates,
                 float* dbias, void* stream) {
    CTN_REQUIRE(dhout && save && whh && dgates, "ctn_lstm_bwd: null pointer");
    CTN_REQUIRE(lstm_args_ok(N, T, H, D), "ctn_lstm_bwd: needs hidden size <= 128 and a multiple of 4, 1 or 2 directions");
    CTN_REQUIRE(ctn_aligned16(dhout) && ctn_aligned16(dgates) && ctn_aligned16(save), "ctn_lstm_bwd: 16-byte alignment");
    dim3 grid(ctn_cdiv(N, NSEQ), D);
    if (dbias)
        CTN_REQUIRE(hipMemsetAsync(dbias, 0, sizeof(float) * D * 4 * H, (hipStream_t)stream) == hipSuccess,
                    "ctn_lstm_bwd: cannot clear the bias gradient");
    static const bool f32_only = [] { const char* e = getenv("CTN_LSTM_F32"); return e && e[0] == '1'; }();
    if (!f32_only) {
        const int NTv = H <= 64 ? 1 : 2;
        const int KSv = 8 * NTv, KRv = NTv == 2 ? 4 : 0;
        const size_t shm = ((size_t)4 * NTv * (KSv - KRv) * 64 + (size_t)3 * KSv * 64) * 16;
        static bool asked[64][2] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!asked[dev][NTv - 1]) {
            const void* fns[3] = {reinterpret_cast<const void*>(lstm_bwd_b3_kernel<1, false>),
                                  reinterpret_cast<const void*>(lstm_bwd_b3_kernel<2, false>),
                                  reinterpret_cast<const void*>(lstm_bwd_b3_kernel<2, true>)};
            for (int i = NTv - 1; i < (NTv == 1 ? 1 : 3); ++i)
                CTN_REQUIRE(hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) == hipSuccess,
                            "ctn_lstm_bwd: cannot raise the dynamic LDS limit");
            asked[dev][NTv - 1] = true;
        }
        if (NTv == 1)
            hipLaunchKernelGGL((lstm_bwd_b3_kernel<1, false>), grid, dim3(NTH), shm, (hipStream_t)stream, dhout, save, whh, dgates, dbias, N, T, H, D);
        else if (H == 128)
            hipLaunchKernelGGL((lstm_bwd_b3_kernel<2, true>), grid, dim3(NTH), shm, (hipStream_t)stream, dhout, save, whh, dgates, dbias, N, T, H, D);
        else
            hipLaunchKernelGGL((lstm_bwd_b3_kernel<2, false>), grid, dim3(NTH), shm, (hipStream_t)stream, dhout, save, whh, dgates, dbias, N, T, H, D);
        CTN_CHECK_LAUNCH("ctn_lstm_bwd");
        return CTN_OK;
    }
    if (H <= 64)
        hipLaunchKernelGGL(lstm_bwd_kernel<1>, grid, dim3(NTH), 0, (hipStream_t)stream, dhout, save, whh, dgates, dbias, N, T, H, D);
    else
        hipLaunchKernelGGL(lstm_bwd_kernel<2>, grid, dim3(NTH), 0, (hipStream_t)stream, dhout, save, whh, dgates, dbias, N, T, H, D);
    CTN_CHECK_LAUNCH("ctn_lstm_bwd");
    return CTN_OK;
}

}  // extern "C"
