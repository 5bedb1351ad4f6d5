// Dense products over feature-last rows on the f32 MFMA, gfx950: the input projection of the dual-path blocks' LSTMs,
// their linear layers and every weight / input gradient of both (src/models/dprnn.py:60-64,85-88 inside chunks,
// :104-116,137-140 across chunks of the reference: nn.LSTM's W_ih x_t + b, nn.Linear, and their backward passes).
//
// The recurrences keep their tensors as (sequence, step) rows with the features contiguous -- R = sequences x steps rows,
// 205 200 of them at the recipe's size, against 64 to 1024 features -- so the two products have their own shapes:
//
//   rows_gemm   Y[r][m]  = sum_k X[r][k] * W(k, m) (+ bias[m])       tall output, short contraction; the output write is
//                                                                    what it costs (R x M floats)
//   rows_wgrad  dW[m][k] += sign * sum_r dY[r][m] * X[r][k]          small output, contraction over ALL rows: split over row
//                                                                    chunks (one generation of resident blocks), every block
//                                                                    ends with float atomics of its tile; optional column
//                                                                    sums of dY (the bias gradient) on the way
//
// Arithmetic: v_mfma_f32_32x32x2_f32, i.e. exact fp32 products accumulated in fp32.  Both kernels are bound by their
// operand / result streams at these shapes (the projection of one LSTM: 27 GFLOP against 0.9 GB), so the f32 matrix
// instruction's rate is not what limits them.
//
// In rows_wgrad the contraction index is the ROW index of both operands, so the MFMA fragments (one value per lane:
// A[m = lane % 32][kk = lane / 32], B[kk][n = lane % 32]) are plain conflict-free reads of row-major LDS tiles.
// rows_gemm contracts along the rows' contiguous axis: its X tile is staged k-major in LDS.
#include "ctn_common.h"

namespace {

constexpr int RNTH = 256;

// four consecutive floats of row `p` starting at column c of a row with `ncols` valid columns; zero beyond the row
// (vec: p + c is 16-byte aligned and c + 3 < ncols is decided by the caller's layout: row pitch and base multiples of 4)
__device__ __forceinline__ f32x4 rows_ld4(const float* __restrict__ p, const int c, const int ncols, const bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec && c + 3 < ncols) return ctn_ld4(p + c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c + e < ncols) v[e] = p[c + e];
    return v;
}

// ------------------------------------------------------------------------------------------------ Y = X W (+ b)
// Block: 128 rows x 128 output columns, four waves of 32 rows x 128 columns (4 accumulator tiles); contraction in
// chunks of 32.  X tile k-major in LDS (Xs[k][row], pitch 132: a fragment read takes 32 consecutive rows of one k).
// W tile: W_KM (W stored [K][M], the product with the transposed forward weight) lands as it lies, Ws[k][m];
// otherwise (W stored [M][K], torch's (out, in) layout) it is staged k-major like X.
template <bool W_KM>
__global__ __launch_bounds__(RNTH) void rows_gemm_kernel(const float* __restrict__ X, const int64_t ldx,
                                                         const float* __restrict__ W, const int64_t ldw,
                                                         const float* __restrict__ bias, const int64_t R, const int K,
                                                         const int M, float* __restrict__ Y, const int64_t ldy,
                                                         const bool xvec, const bool wvec, const unsigned tiles_m) {
    constexpr int BR = 128, BM = 128, KC = 32, PITCH = 132;
    __shared__ __attribute__((aligned(16))) float Xs[KC * PITCH];
    __shared__ __attribute__((aligned(16))) float Ws[KC * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // column tile fastest: the blocks in flight together write whole rows of Y (every memory channel) and share X tiles
    const int64_t r0 = (int64_t)(blockIdx.x / tiles_m) * BR;
    const int m0 = (int)(blockIdx.x % tiles_m) * BM;

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // loaders: a 128 x 32 tile is 1024 float4, four per thread; thread -> (row = tid / 8 + 32 q, 4 columns at (tid % 8) * 4)
    const int lc = (tid & 7) * 4, lr = tid >> 3;
    f32x4 rx[4], rw[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t r = r0 + lr + 32 * q;
            rx[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < R) rx[q] = rows_ld4(X + r * ldx + k0, lc, K - k0, xvec);
            if (W_KM) {      // rows of W are contraction indices: tile rows k0 + (tid / 32 + 8 q), columns m0 + (tid % 32) * 4
                const int kk = k0 + (tid >> 5) + 8 * q;
                rw[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (kk < K) rw[q] = rows_ld4(W + (int64_t)kk * ldw + m0, (tid & 31) * 4, M - m0, wvec);
            } else {
                const int m = m0 + lr + 32 * q;
                rw[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < M) rw[q] = rows_ld4(W + (int64_t)m * ldw + k0, lc, K - k0, wvec);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Xs[(lc + e) * PITCH + lr + 32 * q] = rx[q][e];
            if (W_KM) {
                ctn_st4(&Ws[((tid >> 5) + 8 * q) * PITCH + (tid & 31) * 4], rw[q]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) Ws[(lc + e) * PITCH + lr + 32 * q] = rw[q][e];
            }
        }
    };

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KC) {
        stage();
        __syncthreads();
        if (k0 + KC < K) fetch(k0 + KC);
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const int kk = 2 * s + half;
            const float av = Xs[kk * PITCH + wave * 32 + l31];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bv = Ws[kk * PITCH + j * 32 + l31];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // accumulators -> Y: register e of lane (l31, half) is row (e & 3) + 8 (e >> 2) + 4 half, column l31 of its tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + j * 32 + l31;
        if (m >= M) continue;
        const float bj = bias ? bias[m] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t r = r0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (r < R) Y[r * ldy + m] = acc[j][e] + bj;
        }
    }
}

// ------------------------------------------------------------------------------------------------ Y = X W^T (+ b), six bf16 products
// The same product in the exact three-piece arithmetic of the pointwise GEMMs (CTN_ARITH_BF16X3): every fp32 element of
// both tiles is split ONCE, by the thread that stages it, into three bf16 pieces (hi + mid + lo = x exactly, eight
// significant bits each), written to three LDS planes [row][k] of bf16 -- both operands are contiguous along the
// contraction, which is the matrix instruction's fragment order, so a fragment is one 16-byte LDS read per plane -- and
// the product tile is hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 (what is dropped is
// below one fp32 rounding of the product).  Six bf16 instructions cost 6/16 of one f32 instruction's time for the same
// product, which takes these launches from the matrix unit's roof to the memory roof.  Rows of 40 bf16 (80 bytes): the
// 16 lanes the LDS serves together for a 16-byte read then cover all 64 banks.  The result leaves through LDS, half a
// wave's strip at a time, as 16-byte stores of 256 contiguous bytes per row.
typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ru32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void rows_split3(const f32x4 v, ru32x2& hi, ru32x2& mid, ru32x2& lo) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a0 = v[2 * i], a1 = v[2 * i + 1];
        const unsigned x0 = __float_as_uint(a0), x1 = __float_as_uint(a1);
        hi[i] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        const float r0 = a0 - __uint_as_float(x0 & 0xffff0000u), r1 = a1 - __uint_as_float(x1 & 0xffff0000u);
        const unsigned ru0 = __float_as_uint(r0), ru1 = __float_as_uint(r1);
        mid[i] = __builtin_amdgcn_perm(ru1, ru0, 0x07060302u);
        const float q0 = r0 - __uint_as_float(ru0 & 0xffff0000u), q1 = r1 - __uint_as_float(ru1 & 0xffff0000u);
        lo[i] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
    }
}

// BM: output columns per block, 128 or 64 (narrow outputs -- the 64 features of the dual-path blocks -- waste half of the
// matrix instructions and LDS of a 128-column tile).
template <int BM>
__global__ __launch_bounds__(RNTH) void rows_gemm_b3_kernel(const float* __restrict__ X, const int64_t ldx,
                                                            const float* __restrict__ W, const int64_t ldw,
                                                            const float* __restrict__ bias, const int64_t R, const int K,
                                                            const int M, float* __restrict__ Y, const int64_t ldy,
                                                            const bool xvec, const bool wvec, const bool yvec,
                                                            const unsigned tiles_m) {
    constexpr int BR = 128, KC = 32, PB = 40;                 // PB: bf16 per LDS row (32 + 8 of padding)
    constexpr int PLANE = BR * PB, WPLANE = BM * PB;          // bf16 per plane of X and of W
    constexpr int NJ = BM / 32;                               // accumulator tiles per wave
    constexpr int OUTF = 4 * 32 * 68;                         // floats of the epilogue staging
    constexpr int LDSB = (3 * PLANE + 3 * WPLANE) * 2 > OUTF * 4 ? (3 * PLANE + 3 * WPLANE) * 2 : OUTF * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDSB];
    __bf16* const planes = reinterpret_cast<__bf16*>(lds_raw);
    __bf16* const Xp = planes;
    __bf16* const Wp = planes + 3 * PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // column tile fastest: the blocks in flight together write whole rows of Y (every memory channel) and share X tiles
    const int64_t r0 = (int64_t)(blockIdx.x / tiles_m) * BR;
    const int m0 = (int)(blockIdx.x % tiles_m) * BM;

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    const int lc = (tid & 7) * 4, lr = tid >> 3;     // thread -> (row lr + 32 q, 4 contraction indices at lc)
    f32x4 rx[4], rw[NJ];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t r = r0 + lr + 32 * q;
            const int m = m0 + lr + 32 * q;
            rx[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < R) rx[q] = rows_ld4(X + r * ldx + k0, lc, K - k0, xvec);
            if (q < NJ) {
                rw[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < M) rw[q] = rows_ld4(W + (int64_t)m * ldw + k0, lc, K - k0, wvec);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = (lr + 32 * q) * PB + lc;
            ru32x2 h, m, l;
            rows_split3(rx[q], h, m, l);
            *reinterpret_cast<ru32x2*>(&Xp[o]) = h;
            *reinterpret_cast<ru32x2*>(&Xp[PLANE + o]) = m;
            *reinterpret_cast<ru32x2*>(&Xp[2 * PLANE + o]) = l;
            if (q < NJ) {
                rows_split3(rw[q], h, m, l);
                *reinterpret_cast<ru32x2*>(&Wp[o]) = h;
                *reinterpret_cast<ru32x2*>(&Wp[WPLANE + o]) = m;
                *reinterpret_cast<ru32x2*>(&Wp[2 * WPLANE + o]) = l;
            }
        }
    };

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KC) {
        stage();
        __syncthreads();
        if (k0 + KC < K) fetch(k0 + KC);
#pragma unroll
        for (int s = 0; s < KC / 16; ++s) {
            const int ao = (wave * 32 + l31) * PB + 16 * s + 8 * half;
            const rbf16x8 ah = *reinterpret_cast<const rbf16x8*>(&Xp[ao]);
            const rbf16x8 am = *reinterpret_cast<const rbf16x8*>(&Xp[PLANE + ao]);
            const rbf16x8 al = *reinterpret_cast<const rbf16x8*>(&Xp[2 * PLANE + ao]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int bo = (j * 32 + l31) * PB + 16 * s + 8 * half;
                const rbf16x8 bh = *reinterpret_cast<const rbf16x8*>(&Wp[bo]);
                const rbf16x8 bm = *reinterpret_cast<const rbf16x8*>(&Wp[WPLANE + bo]);
                const rbf16x8 bl = *reinterpret_cast<const rbf16x8*>(&Wp[2 * WPLANE + bo]);
                // smallest terms first
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // accumulators -> Y through LDS (the operand planes are free behind the loop's last barrier): each wave owns 8 KB,
    // a 32-row x 64-column half of its strip at a time, rows of 68 floats; then every lane stores 16 bytes, 16 lanes a row.
    float* out = reinterpret_cast<float*>(lds_raw) + wave * (32 * 68);
#pragma unroll
    for (int hc = 0; hc < BM / 64; ++hc) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * hc + jj;
#pragma unroll
            for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * half) * 68 + jj * 32 + l31] = acc[j][e];
        }
        // (a wave reads back only what it wrote itself: no barrier, the LDS executes a wave's accesses in order)
        const int c4 = (lane & 15) * 4;                  // 4 columns of the 64
        const int mcol = m0 + hc * 64 + c4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (mcol + e < M) bv[e] = bias[mcol + e];
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = rr * 4 + (lane >> 4);
            const int64_t r = r0 + wave * 32 + row;
            f32x4 v = ctn_ld4(&out[row * 68 + c4]);
            if (r < R) {
                float* dst = Y + r * ldy + mcol;
                if (yvec && mcol + 3 < M) {
                    ctn_st4(dst, v + bv);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (mcol + e < M) dst[e] = v[e] + bv[e];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ short contractions (K <= 64)
// Y = X W^T + b for K <= 64 (the LSTMs' input projection, 64 -> 1024, and the linear layers' input gradient), where the
// output write is everything and the general kernel above spends its time on staging and barriers: here the weight
// slab of a block's 128 output columns is split ONCE into LDS planes (all K at once) and the block then walks row tiles
// without a barrier.  Each wave takes 32 rows at a time straight from global memory -- lane (row, half) loads the 32
// contraction indices 32 half .. 32 half + 31 of its row, one full 128-byte line, and splits them in registers -- with
// the roles of the operands swapped: the weights are the A operand (fragments from LDS), the rows the B operand, so the
// accumulators hold four CONSECUTIVE output columns of one row per register group and leave as 16-byte stores
// without a pass through LDS.  The next tile's rows are requested before the current tile's matrix instructions.
// Contraction index of slot (k-step s, half h, element j): 32 h + 8 s + j, on both operands.
__global__ __launch_bounds__(RNTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rows_gemm_b3_k64_kernel(const float* __restrict__ X, const int64_t ldx, const float* __restrict__ W,
                             const int64_t ldw, const float* __restrict__ bias, const int64_t R, const int K, const int M,
                             float* __restrict__ Y, const int64_t ldy, const unsigned tiles_m, const bool wvec) {
    constexpr int BM = 128, PB = 72;                      // 64 + 8 bf16 per LDS row: 16-byte reads of 16 rows cover all banks
    constexpr int WPLANE = BM * PB;
    __shared__ __attribute__((aligned(16))) __bf16 Wp[3 * WPLANE];
    __shared__ __attribute__((aligned(16))) float bs[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = (int)(blockIdx.x % tiles_m) * BM;
    const int64_t group = blockIdx.x / tiles_m, ngroups = gridDim.x / tiles_m;

    // ---- the weight slab: 128 rows x 64 contraction indices, zero beyond M and K
    {
        const int lc = (tid & 15) * 4, lr = tid >> 4;     // thread -> (row lr + 16 q, 4 contraction indices at lc)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = m0 + lr + 16 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < M) v = rows_ld4(W + (int64_t)m * ldw, lc, K, wvec);
            ru32x2 h, mi, l;
            rows_split3(v, h, mi, l);
            const int o = (lr + 16 * q) * PB + lc;
            *reinterpret_cast<ru32x2*>(&Wp[o]) = h;
            *reinterpret_cast<ru32x2*>(&Wp[WPLANE + o]) = mi;
            *reinterpret_cast<ru32x2*>(&Wp[2 * WPLANE + o]) = l;
        }
        if (tid < BM) bs[tid] = (bias && m0 + tid < M) ? bias[m0 + tid] : 0.f;
    }
    __syncthreads();

    const int64_t ntiles = (R + 31) / 32;
    const int64_t stride = ngroups * 4;
    int64_t tile = group * 4 + wave;
    const int kbase = 32 * half;
    f32x4 xv[8], xn[8];
    auto fetch = [&](int64_t tl, f32x4 (&dst)[8]) {
        int64_t r = tl * 32 + l31;
        if (r > R - 1) r = R - 1;                          // clamped: rows beyond R are computed and not stored
        const float* p = X + r * ldx + kbase;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = kbase + 4 * q;
            dst[q] = ctn_ld4(p + (k < K ? 4 * q : 0));
            if (!(k < K)) dst[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (tile < ntiles) fetch(tile, xv);
    for (; tile < ntiles; tile += stride) {
        if (tile + stride < ntiles) fetch(tile + stride, xn);
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            union { rbf16x8 v; ru32x2 p[2]; } bh, bm, bl;
            rows_split3(xv[2 * s], bh.p[0], bm.p[0], bl.p[0]);
            rows_split3(xv[2 * s + 1], bh.p[1], bm.p[1], bl.p[1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ao = (j * 32 + l31) * PB + kbase + 8 * s;
                const rbf16x8 ah = *reinterpret_cast<const rbf16x8*>(&Wp[ao]);
                const rbf16x8 am = *reinterpret_cast<const rbf16x8*>(&Wp[WPLANE + ao]);
                const rbf16x8 al = *reinterpret_cast<const rbf16x8*>(&Wp[2 * WPLANE + ao]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh.v, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl.v, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm.v, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh.v, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm.v, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh.v, acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // one k-step's fragments at a time: hoisted reads spill registers
        }
        // accumulator register 4 g + i of lane (l31, half) in tile j: output column m0 + 32 j + 8 g + 4 half + i of row l31
        const int64_t r = tile * 32 + l31;
        if (r < R) {
            float* yrow = Y + r * ldy + m0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int mc = 32 * j + 8 * g + 4 * half;
                    if (m0 + mc < M) {
                        const f32x4 bv = ctn_ld4(&bs[mc]);
                        ctn_st4(yrow + mc, f32x4{acc[j][4 * g] + bv[0], acc[j][4 * g + 1] + bv[1], acc[j][4 * g + 2] + bv[2],
                                                 acc[j][4 * g + 3] + bv[3]});
                    }
                }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[q] = xn[q];
    }
}

// ------------------------------------------------------------------------------------------------ dW += dY^T X
// Block: TM x 64 output tile and one chunk of rows; stages of 32 rows: dYs[32][TM], Xs[32][64] row-major as they lie in
// memory.  TM = 128: wave w owns output rows 32 w .. 32 w + 31 and both 32-column tiles; TM = 64: wave w owns row tile
// w / 2 and column tile w % 2.  Blocks of column tile 0 also sum dY's columns (the bias gradient).
template <int TM>
__global__ __launch_bounds__(RNTH) void rows_wgrad_kernel(const float* __restrict__ dY, const int64_t lddy,
                                                          const float* __restrict__ X, const int64_t ldx, const int64_t R,
                                                          const int M, const int K, float* __restrict__ dW,
                                                          const int64_t lddw, float* __restrict__ dbias, const float sign,
                                                          const int64_t rchunk, const bool yvec, const bool xvec) {
    constexpr int TK = 64, RS = 32;
    constexpr int NY = TM * RS / 4 / RNTH;       // float4 per thread per stage: 4 (TM = 128) or 2
    constexpr int NX = TK * RS / 4 / RNTH;       // 2
    constexpr int TJ = TM == 128 ? 2 : 1;        // column tiles per wave
    __shared__ __attribute__((aligned(16))) float Ys[RS * TM];
    __shared__ __attribute__((aligned(16))) float Xs[RS * TK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * TM, k0 = blockIdx.y * TK;
    // the 32-row stages are dealt round-robin over the grid's z blocks (stage G to block G % gridDim.z), so that the
    // resident blocks read one contiguous stretch of the operands at a time instead of streams a chunk apart
    const int64_t re = R, rstep = (int64_t)gridDim.z * RS, rb = (int64_t)blockIdx.z * RS;
    const int wm = TM == 128 ? wave * 32 : (wave >> 1) * 32;
    const int wk = TM == 128 ? 0 : (wave & 1) * 32;

    f32x16 acc[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float colsum = 0.f;
    const bool do_bias = dbias != nullptr && blockIdx.y == 0;

    constexpr int YC = TM / 4;                   // float4 per tile row of dY
    f32x4 ry[NY], rx[NX];
    auto fetch = [&](int64_t r_stage) {
#pragma unroll
        for (int q = 0; q < NY; ++q) {
            const int i = tid + RNTH * q;
            const int64_t r = r_stage + i / YC;
            ry[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < re) ry[q] = rows_ld4(dY + r * lddy + m0, (i % YC) * 4, M - m0, yvec);
        }
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int i = tid + RNTH * q;
            const int64_t r = r_stage + i / (TK / 4);
            rx[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < re) rx[q] = rows_ld4(X + r * ldx + k0, (i % (TK / 4)) * 4, K - k0, xvec);
        }
    };

    if (rb < re) {
        fetch(rb);
        for (int64_t rs = rb; rs < re; rs += rstep) {
#pragma unroll
            for (int q = 0; q < NY; ++q) ctn_st4(&Ys[(tid + RNTH * q) * 4], ry[q]);
#pragma unroll
            for (int q = 0; q < NX; ++q) ctn_st4(&Xs[(tid + RNTH * q) * 4], rx[q]);
            __syncthreads();
            if (rs + rstep < re) fetch(rs + rstep);
#pragma unroll
            for (int s = 0; s < RS / 2; ++s) {
                const int rr = 2 * s + half;
                const float av = Ys[rr * TM + wm + l31];
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const float bv = Xs[rr * TK + wk + j * 32 + l31];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                }
            }
            if (do_bias && tid < TM) {
#pragma unroll 8
                for (int rr = 0; rr < RS; ++rr) colsum += Ys[rr * TM + tid];
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int k = k0 + wk + j * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (m < M && k < K) atomicAdd(dW + (int64_t)m * lddw + k, sign * acc[j][e]);
        }
    }
    if (do_bias && tid < TM && m0 + tid < M) atomicAdd(dbias + m0 + tid, sign * colsum);
}

// ------------------------------------------------------------------------------------------------ dW += dY^T X, six bf16 products
// The weight gradient in the exact three-piece arithmetic.  The contraction runs over the ROWS of both operands, so the
// matrix instruction wants, per lane, eight consecutive rows of one column: a transposed view of tiles that arrive
// row-major.  The tiles are staged as they lie -- each thread splits the float4 it loaded and stores 8 bytes per piece
// plane, [row][column] of bf16 -- and the fragments are fetched with ds_read_b64_tr_b16, the LDS's transposing read (per
// group of 16 lanes a block of 4 rows x 16 columns comes back column-major: lane i receives column i of the four rows),
// two reads per fragment and plane.  LDS image: rows of 256 bytes (128 bf16), the 16-byte chunk ch of row r stored at
// chunk ch ^ (((r & 3) << 2) | ((r >> 2) & 3)), which keeps both the 8-byte stores and the transposing reads off each
// other's banks.  Block: 128 x 64 output tile (wave w: rows 32 w .. 32 w + 31, both column tiles), stages of 32 rows.
typedef short rs16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int rows_img_off(const int row, const int ch) {      // byte offset of a 16-byte chunk
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

__global__ __launch_bounds__(RNTH) void rows_wgrad_b3_kernel(const float* __restrict__ dY, const int64_t lddy,
                                                             const float* __restrict__ X, const int64_t ldx, const int64_t R,
                                                             const int M, const int K, float* __restrict__ dW,
                                                             const int64_t lddw, float* __restrict__ dbias, const float sign,
                                                             const int64_t rchunk, const bool yvec, const bool xvec) {
    constexpr int TM = 128, TK = 64, RS = 32;
    constexpr int PLANEB = RS * 256;                      // bytes of one piece plane (32 rows x 256 bytes)
    __shared__ __attribute__((aligned(16))) unsigned char Yp[3 * PLANEB];
    __shared__ __attribute__((aligned(16))) unsigned char Xp[3 * PLANEB];
    __shared__ float csum[4][TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31;
    const int m0 = blockIdx.x * TM, k0 = blockIdx.y * TK;
    const int64_t rb = (int64_t)blockIdx.z * rchunk, re = rb + rchunk < R ? rb + rchunk : R;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = dbias != nullptr && blockIdx.y == 0;

    // loaders: dY tile 32 x 128 = 1024 float4 (four per thread: row tid / 32 + 8 q, columns 4 (tid % 32)); X tile 32 x 64 =
    // 512 float4 (two per thread: row tid / 16 + 16 q, columns 4 (tid % 16))
    const int yc = (tid & 31) * 4, yr = tid >> 5;
    const int xc = (tid & 15) * 4, xr = tid >> 4;
    f32x4 ry[4], rx[2];
    auto fetch = [&](int64_t r_stage) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t r = r_stage + yr + 8 * q;
            ry[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < re) ry[q] = rows_ld4(dY + r * lddy + m0, yc, M - m0, yvec);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t r = r_stage + xr + 16 * q;
            rx[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r < re) rx[q] = rows_ld4(X + r * ldx + k0, xc, K - k0, xvec);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ru32x2 h, m, l;
            rows_split3(ry[q], h, m, l);
            const int o = rows_img_off(yr + 8 * q, yc >> 3) + (yc & 4) * 2;
            *reinterpret_cast<ru32x2*>(Yp + o) = h;
            *reinterpret_cast<ru32x2*>(Yp + PLANEB + o) = m;
            *reinterpret_cast<ru32x2*>(Yp + 2 * PLANEB + o) = l;
            if (do_bias) colsum += ry[q];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ru32x2 h, m, l;
            rows_split3(rx[q], h, m, l);
            const int o = rows_img_off(xr + 16 * q, xc >> 3) + (xc & 4) * 2;
            *reinterpret_cast<ru32x2*>(Xp + o) = h;
            *reinterpret_cast<ru32x2*>(Xp + PLANEB + o) = m;
            *reinterpret_cast<ru32x2*>(Xp + 2 * PLANEB + o) = l;
        }
    };
    // transposing read of one fragment: rows r0 .. r0 + 7 of 32 columns starting at c0 (a multiple of 32), this lane's
    // column being c0 + l31.  Lane 4 q + p of a 16-lane group supplies row q, columns 4 p .. 4 p + 3 of its group's block.
    const int gq = (lane & 15) >> 2, gp = lane & 3, gblk = ((lane >> 4) & 1) * 16, ghalf = lane >> 5;
    auto frag = [&](const unsigned char* plane, const int s, const int c0) -> rbf16x8 {
        const int r0 = 16 * s + 8 * ghalf;
        const int col = c0 + gblk + 4 * gp;                 // first of this lane's four address columns
        union { rbf16x8 v; rs16x4 h[2]; } u;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int row = r0 + 4 * x + gq;
            const unsigned char* ptr = plane + rows_img_off(row, col >> 3) + (col & 4) * 2;
            u.h[x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rs16x4*)(ptr));
        }
        return u.v;
    };

    if (rb < re) {
        fetch(rb);
        for (int64_t rs = rb; rs < re; rs += RS) {
            stage();
            __syncthreads();
            if (rs + RS < re) fetch(rs + RS);
#pragma unroll
            for (int s = 0; s < RS / 16; ++s) {
                const rbf16x8 ah = frag(Yp, s, wave * 32), am = frag(Yp + PLANEB, s, wave * 32),
                              al = frag(Yp + 2 * PLANEB, s, wave * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const rbf16x8 bh = frag(Xp, s, j * 32), bm = frag(Xp + PLANEB, s, j * 32),
                                  bl = frag(Xp + 2 * PLANEB, s, j * 32);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }

    const int half = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + j * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (m < M && k < K) atomicAdd(dW + (int64_t)m * lddw + k, sign * acc[j][e]);
        }
    }
    if (do_bias) {          // a thread summed columns yc .. yc + 3 over its rows: fold the 8 row groups through LDS
        // (csum is not an operand plane: no barrier is needed in front of these stores)
        if (yr < 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[yr][yc + e] = 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&csum[yr & 3][yc + e], colsum[e]);
        __syncthreads();
        if (tid < TM && m0 + tid < M)
            atomicAdd(dbias + m0 + tid, sign * ((csum[0][tid] + csum[1][tid]) + (csum[2][tid] + csum[3][tid])));
    }
}


// ------------------------------------------------------------------------------------------------ dW += dY^T X, whole output per block
// The three-piece weight gradient for the recurrent layers' shapes (512 x 128, 512 x 64 per block): ONE block holds the
// whole output in its accumulators (eight waves, 128 or 64 registers each) and walks a chunk of rows, so every element of
// dY and X is fetched, split and staged exactly once on the chip, and each wave's 4 x 2 (or 2 x 2) accumulator tiles
// reuse a fragment set 2 to 4 times: 18 KB of LDS fragments per 48 matrix instructions where the tile-per-block kernel
// above reads 9 KB per 12.  Stages of 16 rows (one k-step of the matrix instruction), planes double-buffered, operands
// requested two stages ahead by buffer loads whose range check supplies the zeros behind the chunk's last row; the
// barriers order LDS traffic only.  LDS image of a plane: rows of MB bf16, as segments of 256 bytes with the chunk swizzle
// of rows_img_off inside each segment (segments and rows alias on the banks alike, so the 256-byte analysis carries over).
template <int WM, int WK, int NWM, int NWK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rows_wgrad_b3_full_kernel(const float* __restrict__ dY, const int64_t lddy, const float* __restrict__ X,
                               const int64_t ldx, const int64_t R, const int M, const int K, float* __restrict__ dW,
                               const int64_t lddw, float* __restrict__ dbias, const float sign, const int64_t rchunk) {
    static_assert(NWM * NWK == 8, "eight waves");
    constexpr int NTHF = 512, RS = 16;
    constexpr int MB = 32 * WM * NWM, KB = 32 * WK * NWK;          // the block's output: MB x KB
    constexpr int YPITCH = 2 * MB, XPITCH = KB >= 128 ? 2 * KB : 256;   // bytes per LDS row (whole 256-byte segments)
    constexpr int YPLANE = RS * YPITCH, XPLANE = RS * XPITCH;
    constexpr int YBUF = 3 * YPLANE, XBUF = 3 * XPLANE;
    constexpr int YQ = RS * (MB / 4) / NTHF;                       // float4 of dY per thread and stage
    constexpr int XTH = RS * (KB / 4);                             // threads that carry a float4 of X
    static_assert(YQ >= 1 && XTH <= NTHF, "tile / thread mapping");
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];
    unsigned char* const Yp = wlds;                                // two buffers
    unsigned char* const Xp = wlds + 2 * YBUF;
    float* const csum = reinterpret_cast<float*>(wlds + 2 * YBUF + 2 * XBUF);      // [NTHF / (MB / 4)][MB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wv % NWM, wk = wv / NWM;                        // this wave's output tiles: rows 32 WM wm .., columns 32 WK wk ..
    const int m0 = blockIdx.x * MB, k0 = blockIdx.y * KB;
    // The row stages are dealt round-robin over the blocks (stage G of the matrix goes to block G % gridDim.z): at any
    // moment the resident blocks read one contiguous stretch of the operands.  With a contiguous chunk per block the
    // blocks' streams lie a chunk apart, advance in lock-step and keep landing on the same memory channels (measured:
    // 2.4 TB/s).
    const int64_t nstages = (R + RS - 1) / RS;
    const int nz = gridDim.z, bz = blockIdx.z;
    const int nst = (int)((nstages - bz + nz - 1) / nz);          // stages bz, bz + nz, ...

    f32x16 acc[WM][WK];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WK; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = dbias != nullptr && blockIdx.y == 0;

    // loaders: float4 i = tid + NTHF q of the 16 x MB tile of dY (row i / (MB / 4), columns 4 (i % (MB / 4))); X likewise
    typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
    constexpr int YF4 = MB / 4, XF4 = KB / 4;
    const int yrow = tid / YF4, ycol = (tid % YF4) * 4;            // + (NTHF / YF4) rows per q
    const int xrow = tid / XF4, xcol = (tid % XF4) * 4;
    constexpr int YRQ = NTHF / YF4;
    const int voY = (yrow * (int)lddy + ycol) * 4, voX = (xrow * (int)ldx + xcol) * 4;
    const int qY = YRQ * (int)lddy * 4;
    // a stage's descriptors start at its first row and end behind its last valid one: rows beyond the matrix (the ragged
    // last stage, and stages requested past the block's last) come back as zeros and touch no memory
    auto fetch = [&](const int g, wu32x4 (&ry)[YQ], wu32x4& rx) {
        const int64_t row0 = ((int64_t)g * nz + bz) * RS;
        const int64_t left = R - row0;
        const int rows = left >= RS ? RS : (left > 0 ? (int)left : 0);
        const int64_t r0c = rows > 0 ? row0 : 0;
        const auto rY = __builtin_amdgcn_make_buffer_rsrc((void*)(dY + r0c * lddy + m0), 0,
                                                          rows > 0 ? (int)(((int64_t)(rows - 1) * lddy + MB) * 4) : 0, 0x00020000);
        const auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)(X + r0c * ldx + k0), 0,
                                                          rows > 0 ? (int)(((int64_t)(rows - 1) * ldx + KB) * 4) : 0, 0x00020000);
#pragma unroll
        for (int q = 0; q < YQ; ++q) ry[q] = __builtin_amdgcn_raw_buffer_load_b128(rY, voY + q * qY, 0, 0);
        if (tid < XTH) rx = __builtin_amdgcn_raw_buffer_load_b128(rX, voX, 0, 0);
    };
    auto img = [](const int row, const int col, const int pitch) -> int {      // byte offset of 4 bf16 at (row, col)
        return row * pitch + (col >> 7) * 256 + 16 * ((((col & 127) >> 3)) ^ (((row & 3) << 2) | ((row >> 2) & 3))) + (col & 4) * 2;
    };
    auto stage = [&](const wu32x4 (&ry)[YQ], const wu32x4& rx, const int buf) {
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            const f32x4 v = __builtin_bit_cast(f32x4, ry[q]);
            ru32x2 h, m, l;
            rows_split3(v, h, m, l);
            unsigned char* o = Yp + buf * YBUF + img(yrow + YRQ * q, ycol, YPITCH);
            *reinterpret_cast<ru32x2*>(o) = h;
            *reinterpret_cast<ru32x2*>(o + YPLANE) = m;
            *reinterpret_cast<ru32x2*>(o + 2 * YPLANE) = l;
            if (do_bias) colsum += v;
        }
        if (tid < XTH) {
            ru32x2 h, m, l;
            rows_split3(__builtin_bit_cast(f32x4, rx), h, m, l);
            unsigned char* o = Xp + buf * XBUF + img(xrow, xcol, XPITCH);
            *reinterpret_cast<ru32x2*>(o) = h;
            *reinterpret_cast<ru32x2*>(o + XPLANE) = m;
            *reinterpret_cast<ru32x2*>(o + 2 * XPLANE) = l;
        }
    };
    // transposing read of one fragment: rows 8 (lane / 32) .. + 7 of the stage, column c0 + lane % 32 (c0 a multiple of 32)
    const int gq = (lane & 15) >> 2, gp = lane & 3, gblk = ((lane >> 4) & 1) * 16, ghalf = lane >> 5;
    auto frag = [&](const unsigned char* plane, const int c0, const int pitch) -> rbf16x8 {
        const int col = c0 + gblk + 4 * gp;
        union { rbf16x8 v; rs16x4 h[2]; } u;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const unsigned char* ptr = plane + img(8 * ghalf + 4 * x + gq, col, pitch);
            u.h[x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rs16x4*)(ptr));
        }
        return u.v;
    };
    auto compute = [&](const int buf) {
        const unsigned char* Yb = Yp + buf * YBUF;
        const unsigned char* Xb = Xp + buf * XBUF;
        rbf16x8 bf[WK][3];
#pragma unroll
        for (int j = 0; j < WK; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[j][pl] = frag(Xb + pl * XPLANE, 32 * (WK * wk + j), XPITCH);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            rbf16x8 af[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[pl] = frag(Yb + pl * YPLANE, 32 * (WM * wm + i), YPITCH);
            // smallest products first; the k-tiles alternate so that consecutive instructions are independent
#define ROWS_W_MFMA(a_, b_)                                                                         \
    _Pragma("unroll") for (int j = 0; j < WK; ++j)                                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a_], bf[j][b_], acc[i][j], 0, 0, 0)
            ROWS_W_MFMA(2, 0);
            ROWS_W_MFMA(0, 2);
            ROWS_W_MFMA(1, 1);
            ROWS_W_MFMA(1, 0);
            ROWS_W_MFMA(0, 1);
            ROWS_W_MFMA(0, 0);
#undef ROWS_W_MFMA
        }
    };
#define ROWS_LDS_BARRIER()                                  \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                       \
        asm volatile("" ::: "memory");                      \
    } while (0)

    if (nst > 0) {
        // stage g: [barrier] request stage g + 2 -> matrix instructions of g (buffer g & 1) -> split and store g + 1
        wu32x4 ry0[YQ], ry1[YQ], rx0 = {0, 0, 0, 0}, rx1 = {0, 0, 0, 0};
        fetch(0, ry0, rx0);
        fetch(1, ry1, rx1);                            // beyond the chunk: zeros
        stage(ry0, rx0, 0);
        for (int g = 0; g < nst; g += 2) {
            ROWS_LDS_BARRIER();
            fetch(g + 2, ry0, rx0);
            compute(0);
            stage(ry1, rx1, 1);                        // stage g + 1 (zeros behind the chunk's end: harmless)
            ROWS_LDS_BARRIER();
            fetch(g + 3, ry1, rx1);
            if (g + 1 < nst) compute(1);
            stage(ry0, rx0, 0);
        }
    }
#undef ROWS_LDS_BARRIER

    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WK; ++j) {
            const int k = k0 + 32 * (WK * wk + j) + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * (WM * wm + i) + (e & 3) + 8 * (e >> 2) + 4 * half;
                atomicAdd(dW + (int64_t)m * lddw + k, sign * acc[i][j][e]);
            }
        }
    if (do_bias) {          // a thread summed columns ycol .. ycol + 3 over its rows: fold the row groups through LDS
        __syncthreads();    // the planes are no longer read; csum lies behind them
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[yrow * MB + ycol + e] = colsum[e];
        __syncthreads();
        if (tid < MB) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < YRQ; ++q) t += csum[q * MB + tid];
            atomicAdd(dbias + m0 + tid, sign * t);
        }
    }
}

template <int WM, int WK, int NWM, int NWK>
constexpr size_t rows_wgrad_full_lds() {
    return (size_t)2 * 3 * 16 * 2 * (32 * WM * NWM) + (size_t)2 * 3 * 16 * (32 * WK * NWK >= 128 ? 2 * 32 * WK * NWK : 256) +
           (size_t)(512 / (32 * WM * NWM / 4)) * (32 * WM * NWM) * 4;
}

}  // namespace

extern "C" int ctn_rows_gemm(const float* X, int64_t ldx, const float* W, int64_t ldw, int w_is_km, const float* bias,
                             int64_t R, int K, int M, float* Y, int64_t ldy, int arith, void* stream) {
    CTN_REQUIRE(X && W && Y, "ctn_rows_gemm: null pointer");
    CTN_REQUIRE(R > 0 && K > 0 && M > 0, "ctn_rows_gemm: empty shape");
    CTN_REQUIRE(ldx >= K && ldy >= M && ldw >= (w_is_km ? M : K), "ctn_rows_gemm: a row stride shorter than the row");
    // the three-piece kernel has a 64-column form for narrow outputs
    const bool b3 = !w_is_km && arith != CTN_ARITH_F32;
    const int bm = (b3 && M <= 64) ? 64 : 128;
    const unsigned tiles_m = (unsigned)ctn_cdiv(M, bm);
    CTN_REQUIRE((R + 127) / 128 * tiles_m <= 0x7fffffff, "ctn_rows_gemm: too many tiles");
    CTN_REQUIRE(arith == CTN_ARITH_DEFAULT || arith == CTN_ARITH_F32 || arith == CTN_ARITH_BF16X3,
                "ctn_rows_gemm: arith is CTN_ARITH_DEFAULT, CTN_ARITH_F32 or CTN_ARITH_BF16X3");
    const bool xvec = ctn_aligned16(X) && ldx % 4 == 0;
    const bool wvec = ctn_aligned16(W) && ldw % 4 == 0;
    const bool yvec = ctn_aligned16(Y) && ldy % 4 == 0;
    dim3 grid((unsigned)((R + 127) / 128 * tiles_m));
    if (b3 && K <= 64 && K % 4 == 0 && M % 4 == 0 && xvec && yvec) {
        // short contraction: resident blocks walk the row tiles (two blocks per CU, every column tile in flight at once)
        const unsigned tm = (unsigned)ctn_cdiv(M, 128);
        int64_t groups = 512 / tm;
        if (groups < 1) groups = 1;
        const int64_t max_groups = ((R + 31) / 32 + 3) / 4;       // a group's four waves take a row tile each
        if (groups > max_groups) groups = max_groups;
        hipLaunchKernelGGL(rows_gemm_b3_k64_kernel, dim3((unsigned)(groups * tm)), dim3(RNTH), 0, (hipStream_t)stream, X, ldx,
                           W, ldw, bias, R, K, M, Y, ldy, tm, wvec);
        CTN_CHECK_LAUNCH("ctn_rows_gemm");
        return CTN_OK;
    }
    if (w_is_km)        // the weight's rows are the contraction: f32 matrix instructions (both arithmetic modes)
        hipLaunchKernelGGL(rows_gemm_kernel<true>, grid, dim3(RNTH), 0, (hipStream_t)stream, X, ldx, W, ldw, bias, R, K, M,
                           Y, ldy, xvec, wvec, tiles_m);
    else if (arith == CTN_ARITH_F32)
        hipLaunchKernelGGL(rows_gemm_kernel<false>, grid, dim3(RNTH), 0, (hipStream_t)stream, X, ldx, W, ldw, bias, R, K, M,
                           Y, ldy, xvec, wvec, tiles_m);
    else if (bm == 64)
        hipLaunchKernelGGL(rows_gemm_b3_kernel<64>, grid, dim3(RNTH), 0, (hipStream_t)stream, X, ldx, W, ldw, bias, R, K, M,
                           Y, ldy, xvec, wvec, yvec, tiles_m);
    else
        hipLaunchKernelGGL(rows_gemm_b3_kernel<128>, grid, dim3(RNTH), 0, (hipStream_t)stream, X, ldx, W, ldw, bias, R, K, M,
                           Y, ldy, xvec, wvec, yvec, tiles_m);
    CTN_CHECK_LAUNCH("ctn_rows_gemm");
    return CTN_OK;
}

extern "C" int ctn_rows_wgrad(const float* dY, int64_t lddy, const float* X, int64_t ldx, int64_t R, int M, int K,
                              float* dW, int64_t lddw, float* dbias, float sign, int arith, void* stream) {
    CTN_REQUIRE(dY && X && dW, "ctn_rows_wgrad: null pointer");
    CTN_REQUIRE(R > 0 && K > 0 && M > 0, "ctn_rows_wgrad: empty shape");
    CTN_REQUIRE(lddy >= M && ldx >= K && lddw >= K, "ctn_rows_wgrad: a row stride shorter than the row");
    CTN_REQUIRE(arith == CTN_ARITH_DEFAULT || arith == CTN_ARITH_F32 || arith == CTN_ARITH_BF16X3,
                "ctn_rows_wgrad: arith is CTN_ARITH_DEFAULT, CTN_ARITH_F32 or CTN_ARITH_BF16X3");
    const bool yvec = ctn_aligned16(dY) && lddy % 4 == 0;
    const bool xvec = ctn_aligned16(X) && ldx % 4 == 0;
    // The three-piece kernel runs on explicit request only (128-row output tiles: M > 64).  At the dual-path blocks'
    // shapes it is the slower of the two -- 1024 x 64 over 511 104 rows: 850 against 780 us, 512 x 128: 772 against 686 --
    // because with a 32-row stage between two barriers its 24 matrix instructions per wave (768 cycles) no longer cover
    // the split, the LDS traffic and the barriers, which the f32 form's 32 instructions (2048 cycles) do.
    // whole-output blocks for the recurrent layers' shapes: M a multiple of 512, K a multiple of 64 up to 128 per block
    // (also what CTN_ARITH_DEFAULT means at these shapes: 405 against 655 us and 539 against 841 us over 511 104 rows)
    if (arith != CTN_ARITH_F32 && M % 512 == 0 && K % 64 == 0 && K <= 128 && yvec && xvec && lddy < (1 << 20) &&
        ldx < (1 << 20)) {
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
                n = 256;
            return n;
        }();
        const int tiles_m = M / 512;
        // one block per CU, all resident at once; the 16-row stages are dealt round-robin over them
        int64_t nchunks = cus / tiles_m;
        if (nchunks < 1) nchunks = 1;
        const int64_t nstages = (R + 15) / 16;
        if (nchunks > (nstages + 3) / 4) nchunks = (nstages + 3) / 4;     // at least four stages per block
        const int64_t rchunk = 0;
        if (nchunks <= 65535) {
            dim3 grid(tiles_m, 1, (unsigned)nchunks);
            if (K > 64) {
                constexpr size_t shm = rows_wgrad_full_lds<4, 2, 4, 2>();
                const void* fn = reinterpret_cast<const void*>(&rows_wgrad_b3_full_kernel<4, 2, 4, 2>);
                CTN_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess,
                            "ctn_rows_wgrad: cannot raise the dynamic LDS limit");
                hipLaunchKernelGGL((rows_wgrad_b3_full_kernel<4, 2, 4, 2>), grid, dim3(512), shm, (hipStream_t)stream, dY, lddy,
                                   X, ldx, R, M, K, dW, lddw, dbias, sign, rchunk);
            } else {
                constexpr size_t shm = rows_wgrad_full_lds<2, 2, 8, 1>();
                const void* fn = reinterpret_cast<const void*>(&rows_wgrad_b3_full_kernel<2, 2, 8, 1>);
                CTN_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess,
                            "ctn_rows_wgrad: cannot raise the dynamic LDS limit");
                hipLaunchKernelGGL((rows_wgrad_b3_full_kernel<2, 2, 8, 1>), grid, dim3(512), shm, (hipStream_t)stream, dY, lddy,
                                   X, ldx, R, M, K, dW, lddw, dbias, sign, rchunk);
            }
            CTN_CHECK_LAUNCH("ctn_rows_wgrad");
            return CTN_OK;
        }
    }
    const bool b3 = arith == CTN_ARITH_BF16X3 && M > 64;
    const int TM = M > 64 ? 128 : 64;
    const int tiles = ctn_cdiv(M, TM) * ctn_cdiv(K, 64);
    CTN_REQUIRE(ctn_cdiv(K, 64) <= 65535, "ctn_rows_wgrad: too many column tiles");
    // one generation of resident blocks (about four per CU; three for the three-piece kernel's 48 KB of planes), chunks of
    // whole 32-row stages, at least 256 rows each
    int64_t nchunks = (b3 ? 768 : 1024) / tiles;
    if (nchunks < 1) nchunks = 1;
    int64_t rchunk = (R + nchunks - 1) / nchunks;
    rchunk = (rchunk + 31) / 32 * 32;
    if (rchunk < 256) rchunk = 256;
    nchunks = (R + rchunk - 1) / rchunk;
    CTN_REQUIRE(nchunks <= 65535, "ctn_rows_wgrad: too many row chunks");
    dim3 grid(ctn_cdiv(M, TM), ctn_cdiv(K, 64), (unsigned)nchunks);
    if (b3)
        hipLaunchKernelGGL(rows_wgrad_b3_kernel, grid, dim3(RNTH), 0, (hipStream_t)stream, dY, lddy, X, ldx, R, M, K, dW, lddw,
                           dbias, sign, rchunk, yvec, xvec);
    else if (TM == 128)
        hipLaunchKernelGGL(rows_wgrad_kernel<128>, grid, dim3(RNTH), 0, (hipStream_t)stream, dY, lddy, X, ldx, R, M, K, dW,
                           lddw, dbias, sign, rchunk, yvec, xvec);
    else
        hipLaunchKernelGGL(rows_wgrad_kernel<64>, grid, dim3(RNTH), 0, (hipStream_t)stream, dY, lddy, X, ldx, R, M, K, dW,
                           lddw, dbias, sign, rchunk, yvec, xvec);
    CTN_CHECK_LAUNCH("ctn_rows_wgrad");
    return CTN_OK;
}
