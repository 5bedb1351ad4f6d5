// Backward pass of a residual layer's two pointwise heads in ONE launch, gfx950 (reference: the output / skip
// Conv1d(H, B, 1) pair of src/models/tdcn.py:172-175,190-194 behind the norm of :187).  Given dY = [dout; dskip]
// (M = M1 + M2 rows), the pre-activation tensor y2 (H channels) and the heads' weights Wh (M x H):
//
//   input gradient    dn2[c][t] = sum_m Wh[m][c] dY[m][t]                         (what ctn_gemm_nn's heads launch forms)
//   its norm sums     S1_b = sum gamma_c dn2,  S2_b = sum gamma_c dn2 xhat2       (that launch's CTN_EPI_NORMBWD epilogue)
//   weight gradient   dWh[m][c] += sum_t dY[m][t] n2[c][t],  n2 = gamma xhat2 + beta   (ctn_gemm_nt's heads launch)
//   bias gradients    db[m] += sum_t dY[m][t]
//
// As two launches the pair streams y2 twice (the input gradient reads it only for S2) and dY twice.  Here a block owns a
// slice of 128 channels and walks time tiles of 64 frames: per tile it forms n2 of its slice ONCE (transformed, split
// into the three bf16 pieces of CTN_ARITH_BF16X3, kept as LDS planes), then streams dY through in k-tiles of 32 rows
// (split once per block, LDS planes, double-buffered) and uses every k-tile TWICE -- as the B operand of the input
// gradient (contraction over m: a transposed view of the [m][t] planes, fetched with ds_read_b64_tr_b16) and as the A
// operand of the weight gradient (contraction over t: plain 16-byte row reads of the same planes).  The weights arrive
// as prepared planes (ctn_weight_prep, the "input gradient" matrix of the heads), 24 KB per k-tile out of L2.
// Every operand is staged through registers one step of the (tile, k-tile) stream ahead, and the barriers between the
// steps order LDS traffic only (a bare s_barrier behind s_waitcnt lgkmcnt(0)): the compiler then waits for each load
// where its registers are first used, and loads and stores in flight cross the barriers.
// The 256 x 128 weight-gradient tile of the slice stays in registers for ALL the block's tiles (64 registers per lane
// over 8 waves) and is added to dWh once, when the block ends: the grid is one block per CU, each with about 31 tiles.
// S2 needs no second pass either: gamma xhat2 = n2 - beta, and n2 is in LDS when the tile's dn2 leaves.
//
// Measured (batch 32, 3999 frames, H = 512, two heads of 128 rows; isolated launches): 598 us against 535-558 us for
// the two launches it replaces (463 us inside a training step); the training step takes 35.3 ms with it, 34.7 ms
// without.  Hence opt-in.  Cycle stamps of one block (a build with -DHB_TRACE, tools/trace_heads_bwd.py) say where the
// time goes: a tile takes about 31 k cycles; its eight k-tile steps take about 2.9 k each although the 48 matrix
// instructions a SIMD issues per step occupy 1.5 k -- with one block per CU the two waves of a SIMD move in lock-step
// between the barriers, so every step's lead-in (first LDS reads, 0.4 k), the staging behind the owner waves' matrix
// work (0.3 k) and the barrier skew are exposed; the n2 transform (1.7 k, bound by vector issue), the wait in front of
// the first barrier (1 k) and the epilogue (5-7 k: accumulator drain, four barriers, the join with n2) run with the
// matrix pipe idle.  Counters: matrix pipe busy 33 % of the launch (44 % over the pair), LDS active 28 %, LDS bank
// conflicts 0.4 %.  Neither a second accumulation chain nor reading the fragments a sub-step ahead changed the time:
// the loss is in the exposed phases.  A second resident block does not cover them either: with slices of 64 channels,
// blocks of four waves and 75 KB of LDS (two blocks per CU, the staging rows aliased onto the idle weight buffer) the
// same stamps show owner-wave steps of 2.5-2.9 k cycles again and the launch takes 659 us.  The wave that owns a
// k-tile's weight gradient issues 36 matrix instructions per step from 32 x 32 tiles that need one 6 KB set of LDS
// fragments per 6 instructions; with both accumulator sets resident (16 + 64 registers) no operand can stay in
// registers across tiles as in ctn_gemm_nn / ctn_gemm_nt, and the fragment reads' latency paces the step.
//
// LDS image of the [row][64 frames] bf16 planes: rows of 128 bytes, 16-byte chunk ch of row r at chunk
// ch ^ (((r >> 2) & 3) | (((r >> 1) & 1) << 2)).  With it the 16-byte row reads (16 lanes = 16 rows of one chunk), the
// transposing reads (a 32-lane half = 4 rows x 4 chunks) and the 8-byte staging stores are all free of bank conflicts.
#include "ctn_common.h"
#ifdef HB_TRACE
__device__ unsigned long long hb_trace_buf[8 * 4096];
#define HB_STAMP()                                                              \
    do {                                                                        \
        if (blockIdx.x == 9) {                                                  \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();         \
            if ((threadIdx.x & 63) == 0 && ntr < 4096) hb_trace_buf[(threadIdx.x >> 6) * 4096 + ntr] = t_; \
            ++ntr;                                                              \
        }                                                                       \
    } while (0)
extern "C" int ctn_debug_hb_trace(void* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(hb_trace_buf), sizeof(hb_trace_buf)) == hipSuccess ? 0 : 1;
}
#else
#define HB_STAMP() do { } while (0)
#endif

namespace {

constexpr int HB_TT = 64;        // frames per time tile
constexpr int HB_CS = 128;       // channels per slice
constexpr int HB_KT = 32;        // rows of dY per k-tile
constexpr int HB_NTH = 512;      // eight waves

typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned hu32x2 __attribute__((ext_vector_type(2)));
typedef short hs16x4 __attribute__((ext_vector_type(4)));

// barrier that orders LDS accesses only (__syncthreads() also waits for every outstanding global load and store)
#define HB_BARRIER()                                        \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                       \
        asm volatile("" ::: "memory");                      \
    } while (0)

// x = hi + mid + lo exactly, each piece the next eight significant bits (truncation): the split of ctn_gemm_nn
__device__ __forceinline__ void hb_split3(const f32x4 v, hu32x2& hi, hu32x2& mid, hu32x2& lo) {
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

__device__ __forceinline__ int hb_off(const int row, const int ch) {      // byte offset of a 16-byte chunk of a plane
    return 128 * row + 16 * (ch ^ (((row >> 2) & 3) | (((row >> 1) & 1) << 2)));
}

// four consecutive bf16 (8 bytes) of each of the three planes at byte offset o -> the four fp32 values they sum to
__device__ __forceinline__ f32x4 hb_join3(const unsigned char* planes, const int plane_bytes, const int o) {
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 2; p >= 0; --p) {          // lo + mid + hi: every partial sum is exact
        const hu32x2 w = *reinterpret_cast<const hu32x2*>(planes + p * plane_bytes + o);
        r[0] += __uint_as_float(w[0] << 16);
        r[1] += __uint_as_float(w[0] & 0xffff0000u);
        r[2] += __uint_as_float(w[1] << 16);
        r[3] += __uint_as_float(w[1] & 0xffff0000u);
    }
    return r;
}

// LDS map (bytes)
constexpr int HB_NPLANE = HB_CS * 128;                 // one n2 plane: 128 rows x 128 bytes
constexpr int HB_DPLANE = HB_KT * 128;                 // one dY k-tile plane
constexpr int HB_WBUF = 3 * 4 * HB_CS * 16;            // prepared weights of one k-tile: 3 planes x 4 k-groups x 128 rows x 16 B
constexpr int HB_SROW = 68;                            // floats per staging row (64 + 4)
constexpr int HB_OFF_N = 0;
constexpr int HB_OFF_D = HB_OFF_N + 3 * HB_NPLANE;                 // two buffers of three planes
constexpr int HB_OFF_W = HB_OFF_D + 2 * 3 * HB_DPLANE;             // two buffers
constexpr int HB_OFF_S = HB_OFF_W + 2 * HB_WBUF;                   // epilogue staging: 64 rows x 68 floats
constexpr int HB_OFF_G = HB_OFF_S + 64 * HB_SROW * 4;              // gamma, beta of the slice
constexpr int HB_OFF_R = HB_OFF_G + 2 * HB_CS * 4;                 // row sums of dY: 256 floats
constexpr int HB_OFF_X = HB_OFF_R + 256 * 4;                       // block-sum scratch: 2 x 8 doubles
constexpr int HB_LDS = HB_OFF_X + 2 * 8 * 8;

// NKT: k-tiles of dY (M / 32): 8 for the two heads of the paper's layers, 4 for a single head of 128 rows
template <int NKT>
__global__ __launch_bounds__(HB_NTH) __attribute__((amdgpu_waves_per_eu(2, 2)))
void heads_bwd_kernel(const ctn_heads_bwd_args a, const int nslices, const int ntt, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TPW = NKT / 2;                 // weight-gradient row tiles (32 rows of dWh) per wave
    static_assert(TPW % 2 == 0, "the register sets of the stream alternate with the k-tile's parity");
    unsigned char* const Np = lds + HB_OFF_N;
    unsigned char* const Dp = lds + HB_OFF_D;
    unsigned char* const Wb = lds + HB_OFF_W;
    float* const Sg = reinterpret_cast<float*>(lds + HB_OFF_S);
    float* const Gs = reinterpret_cast<float*>(lds + HB_OFF_G);
    float* const Bes = Gs + HB_CS;
    float* const Rs = reinterpret_cast<float*>(lds + HB_OFF_R);
    double* const Xs = reinterpret_cast<double*>(lds + HB_OFF_X);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef HB_TRACE
    int ntr = 0;
#endif
    const int l31 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int ct = wv & 3, tt = wv >> 2;         // input-gradient tile of this wave: channels 32 ct.., frames 32 tt..
    const int wc = wv & 3, wm = wv >> 2;         // weight-gradient tiles: channels 32 wc.., rows 32 (wm TPW + j)
    const int slice = blockIdx.x % nslices, part = blockIdx.x / nslices, nparts = gridDim.x / nslices;
    const int c0 = slice * HB_CS;
    const int tile_lo = (int)((int64_t)part * ntiles / nparts), tile_hi = (int)((int64_t)(part + 1) * ntiles / nparts);
    const int T = a.T, M1 = a.M1;
    const float alpha = a.alpha[0];
    const float plim = ctn_prelu_lim(alpha);

    if (tid < HB_CS) {
        Gs[tid] = a.gamma[c0 + tid];
        Bes[tid] = a.beta[c0 + tid];
    }
    if (tid < 256) Rs[tid] = 0.f;

    // staging roles: thread -> row tid / 16 (+ 32 q) of a tile, frames 4 (tid % 16) .. + 3
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    const int sch = scol >> 3, ssub = (scol & 4) * 2;

    // Global operands through buffer loads: one per-lane byte offset per operand (loop-invariant), everything that
    // changes with the tile or the k-tile in the scalar offset -- no vector registers for addresses.  A lane whose
    // float4 starts at or beyond frame T gets an offset beyond the descriptor's range: the hardware returns zeros
    // and touches no memory (rows hold at least T rounded up to 4 floats, so every other float4 lies inside its row).
    constexpr int HB_OOB = (int)0x80000000u;
    typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
    const int KG = 2 * ((a.M1 + a.M2 + 15) / 16), Mp = (a.H + 127) / 128 * 128;
    const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W_prep, 0, 3 * KG * Mp * 16, 0x00020000);
    const auto rD1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY1, 0, 0x7ffffffc, 0x00020000);
    const auto rD2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY2, 0, 0x7ffffffc, 0x00020000);
    const auto rY = __builtin_amdgcn_make_buffer_rsrc((void*)a.y2, 0, 0x7ffffffc, 0x00020000);
    int voW[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {                // fragment i = tid + 512 j of a k-tile: (plane, k-group) i / 128, row i % 128
        const int i = tid + HB_NTH * j, pg = i >> 7;
        voW[j] = (((pg >> 2) * KG + (pg & 3)) * Mp + (i & 127)) * 16;
    }
    const int voD1 = (srow * (int)a.lddy1 + scol) * 4, voD2 = (srow * (int)a.lddy2 + scol) * 4;
    const int voY = (srow * (int)a.ldy2 + scol) * 4;
    auto w_load = [&](const int kt, hu32x4 (&dst)[3]) {
        const int so = (4 * kt * Mp + c0) * 16;
#pragma unroll
        for (int j = 0; j < 3; ++j) dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rW, voW[j], so, 0);
    };
    auto w_stage = [&](const hu32x4 (&src)[3], const int buf) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            *reinterpret_cast<hu32x4*>(Wb + buf * HB_WBUF + (tid + HB_NTH * j) * 16) = src[j];
    };
    // ---- the dY stream: (tile, k-tile) pairs in order
    auto dy_load = [&](const int tile, const int kt) -> hu32x4 {
        const int b = tile / ntt, t0 = (tile - b * ntt) * HB_TT;
        const int m = kt * HB_KT;
        const bool in = t0 + scol < T;
        if (m < M1)
            return __builtin_amdgcn_raw_buffer_load_b128(rD1, in ? voD1 : HB_OOB,
                                                         (int)(((int64_t)b * a.dy1bs + (int64_t)m * a.lddy1 + t0) * 4), 0);
        return __builtin_amdgcn_raw_buffer_load_b128(rD2, in ? voD2 : HB_OOB,
                                                     (int)(((int64_t)b * a.dy2bs + (int64_t)(m - M1) * a.lddy2 + t0) * 4), 0);
    };
    auto dy_stage = [&](const hu32x4 raw, const int tile, const int buf, const int kt, const bool sum_rows) {
        const int b = tile / ntt, t0 = (tile - b * ntt) * HB_TT;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = t0 + scol + e < T ? __uint_as_float(raw[e]) : 0.f;
        hu32x2 h, m, l;
        hb_split3(v, h, m, l);
        unsigned char* base = Dp + buf * 3 * HB_DPLANE + hb_off(srow, sch) + ssub;
        *reinterpret_cast<hu32x2*>(base) = h;
        *reinterpret_cast<hu32x2*>(base + HB_DPLANE) = m;
        *reinterpret_cast<hu32x2*>(base + 2 * HB_DPLANE) = l;
        if (sum_rows) {             // bias gradients: the 16 threads of a row fold their sums, one of them owns the slot
            float s = (v[0] + v[1]) + (v[2] + v[3]);
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
            if ((tid & 15) == 0) Rs[kt * HB_KT + srow] += s;
        }
    };
    auto y2_load = [&](const int tile, hu32x4 (&dst)[4]) {
        const int b = tile / ntt, t0 = (tile - b * ntt) * HB_TT;
        const int vo = t0 + scol < T ? voY : HB_OOB;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q] = __builtin_amdgcn_raw_buffer_load_b128(
                rY, vo, (int)(((int64_t)b * a.y2bs + (int64_t)(c0 + 32 * q) * a.ldy2 + t0) * 4), 0);
    };

    f32x16 accw[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[j][e] = 0.f;
    float s1 = 0.f, s2 = 0.f;            // norm sums of the utterance in progress
    int cur_b = tile_lo < tile_hi ? tile_lo / ntt : 0;
    const bool sum_rows = slice == 0 && (a.rowsum1 != nullptr || a.rowsum2 != nullptr);

    // transposing-read lane roles: 16-lane group gi, lane 4 q + p of it
    const int gq = (lane & 15) >> 2, gp = lane & 3, gblk = ((lane >> 4) & 1) * 16;

    auto flush_sums = [&](const int b) {
        // block sum of (s1, s2) in fp64, one atomic pair per utterance and block
        const double d1 = ctn_wave_sum((double)s1), d2 = ctn_wave_sum((double)s2);
        if (lane == 0) {
            Xs[wave] = d1;
            Xs[8 + wave] = d2;
        }
        __syncthreads();
        if (tid < 2 && a.esums) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += Xs[tid * 8 + w];
            atomicAdd(a.esums + ((int64_t)b * CTN_STAT_SLOTS + ctn_slot()) * 2 + tid, tot);
        }
        __syncthreads();
        s1 = s2 = 0.f;
    };

    if (tile_lo < tile_hi) {
        // Stream position g = (tile, kt).  Step g: [barrier] weight load of g + 1, dY load of g + 2 -> matrix instructions
        // of g (LDS buffers g & 1) -> staging of g + 1 into LDS buffers (g + 1) & 1.  The weights come from L2 and have
        // the step's matrix work to arrive; dY comes from HBM and has that of two steps.  A step's staging waits behind
        // its matrix instructions, not in front of them.
        hu32x4 yv[4], wq[3], dq[2];
        auto next_pos = [&](int tile, int kt, int& ntile, int& nkt) {     // position behind (tile, kt)
            nkt = kt + 1;
            ntile = tile;
            if (nkt == NKT) {
                nkt = 0;
                ntile = tile + 1;
            }
        };
        y2_load(tile_lo, yv);
        w_load(0, wq);
        dq[0] = dy_load(tile_lo, 0);
        {
            int t1, k1;
            next_pos(tile_lo, 0, t1, k1);
            dq[1] = t1 < tile_hi ? dy_load(t1, k1) : dq[0];
        }
        __syncthreads();                                         // gamma / beta / row-sum slots are in place
        dy_stage(dq[0], tile_lo, 0, 0, sum_rows);
        w_stage(wq, 0);
        for (int tile = tile_lo; tile < tile_hi; ++tile) {
            const int b = tile / ntt, t0 = (tile - b * ntt) * HB_TT;
            if (b != cur_b) {
                flush_sums(cur_b);
                cur_b = b;
            }
            HB_STAMP();
            // ---- n2 of this tile: transform, split, planes
            {
                const float mean = a.mean[(int64_t)b * a.ssb], rstd = a.rstd[(int64_t)b * a.ssb];
                const float nmr = -mean * rstd;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = srow + 32 * q;
                    const float g = Gs[r], be = Bes[r];
                    f32x4 n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = ctn_prelu_med(__uint_as_float(yv[q][e]), alpha, plim);
                        const float v = __builtin_fmaf(__builtin_fmaf(u, rstd, nmr), g, be);
                        n[e] = t0 + scol + e < T ? v : 0.f;
                    }
                    hu32x2 h, m, l;
                    hb_split3(n, h, m, l);
                    unsigned char* base = Np + hb_off(r, sch) + ssub;
                    *reinterpret_cast<hu32x2*>(base) = h;
                    *reinterpret_cast<hu32x2*>(base + HB_NPLANE) = m;
                    *reinterpret_cast<hu32x2*>(base + 2 * HB_NPLANE) = l;
                }
            }
            HB_STAMP();
            f32x16 accd;
#pragma unroll
            for (int e = 0; e < 16; ++e) accd[e] = 0.f;
            const bool more = tile + 1 < tile_hi;
            // k-tiles in two halves of TPW: the half is a rolled loop (the wave pair wm that owns the half's row tiles of the
            // weight gradient changes with it), the k-tiles inside a half are unrolled (their accumulators and the register
            // sets of the stream are named)
#pragma unroll 1
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int kj = 0; kj < TPW; ++kj) {
                const int kt = kh * TPW + kj;
                // planes of (tile, kt) and its weights are complete; everything of the step before is free
                HB_BARRIER();
                HB_STAMP();
                // loads of the position two steps ahead, into the register set this position's staging has just freed
                {
                    int t1, k1, t2, k2;
                    next_pos(tile, kt, t1, k1);
                    next_pos(t1, k1, t2, k2);
                    w_load(k1, wq);
                    // (unconditional, so that the counter waits below are exact: past the block's last position the
                    // load repeats the last tile's and is not used)
                    dq[kj & 1] = dy_load(t2 < tile_hi ? t2 : tile_hi - 1, k2);
                }
                const unsigned char* D0 = Dp + (kj & 1) * 3 * HB_DPLANE;
                const unsigned char* W0 = Wb + (kj & 1) * HB_WBUF;
                // Fragment reads run one sub-step ahead of the matrix instructions that consume them (two register sets);
                // the scheduling fences keep that order, the counter waits are the compiler's.
                hbf16x8 fa[2][3], fb[2][3];
                // input gradient, sub-step s: accd[c][t] += Wh^T[c][m] dY[m][t] over rows 16 s .. 16 s + 15 of the k-tile
                auto ld_d = [&](const int s, hbf16x8 (&af)[3], hbf16x8 (&bf)[3]) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        af[p] = *reinterpret_cast<const hbf16x8*>(W0 + ((p * 4 + 2 * s + half) * HB_CS + 32 * ct + l31) * 16);
                        union { hbf16x8 v; hs16x4 h[2]; } u;
#pragma unroll
                        for (int x = 0; x < 2; ++x) {
                            const int row = 16 * s + 8 * half + 4 * x + gq;
                            const int col = 32 * tt + gblk + 4 * gp;
                            const unsigned char* ptr = D0 + p * HB_DPLANE + hb_off(row, col >> 3) + (col & 4) * 2;
                            u.h[x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(ptr));
                        }
                        bf[p] = u.v;
                    }
                };
                // weight gradient of row tile kt, sub-step s: accw[m][c] += dY[m][t] n2[c][t] over frames 16 s .. 16 s + 15
                auto ld_w = [&](const int s, hbf16x8 (&af)[3], hbf16x8 (&bf)[3]) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        af[p] = *reinterpret_cast<const hbf16x8*>(D0 + p * HB_DPLANE + hb_off(l31, 2 * s + half));
                        bf[p] = *reinterpret_cast<const hbf16x8*>(Np + p * HB_NPLANE + hb_off(32 * wc + l31, 2 * s + half));
                    }
                };
                auto mm = [&](f32x16& acc, const hbf16x8 (&af)[3], const hbf16x8 (&bf)[3]) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc, 0, 0, 0);
                };
                ld_d(0, fa[0], fb[0]);
                ld_d(1, fa[1], fb[1]);
                __builtin_amdgcn_sched_barrier(0);
                mm(accd, fa[0], fb[0]);
                __builtin_amdgcn_sched_barrier(0);
                if (kh == wm) {             // the waves that own row tile kt
                    ld_w(0, fa[0], fb[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(accd, fa[1], fb[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_w(1, fa[1], fb[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(accw[kj], fa[0], fb[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_w(2, fa[0], fb[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(accw[kj], fa[1], fb[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    ld_w(3, fa[1], fb[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(accw[kj], fa[0], fb[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(accw[kj], fa[1], fb[1]);
                } else {
                    mm(accd, fa[1], fb[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- staging of the next position (its registers were filled during the step before this one)
                {
                    int t1, k1;
                    next_pos(tile, kt, t1, k1);
                    if (t1 < tile_hi) {
                        dy_stage(dq[(kj + 1) & 1], t1, (kj + 1) & 1, k1, sum_rows);
                        w_stage(wq, (kj + 1) & 1);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                HB_STAMP();
            }
            // the next tile's y2 arrives while this tile's result leaves (its registers are free of the k loop's pressure)
            HB_STAMP();
            if (more) y2_load(tile + 1, yv);
            // ---- the tile's dn2 leaves through LDS, 64 channels at a time; the norm sums ride along
#pragma unroll
            for (int hc = 0; hc < 2; ++hc) {
                if ((ct >> 1) == hc) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        Sg[(32 * (ct & 1) + (e & 3) + 8 * (e >> 2) + 4 * half) * HB_SROW + 32 * tt + l31] = accd[e];
                }
#ifdef HB_TRACE_EPI
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                HB_STAMP();
#endif
                HB_BARRIER();
#ifdef HB_TRACE_EPI
                HB_STAMP();
#endif
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = srow + 32 * q;                   // row of the staged half
                    const int c = 64 * hc + r;                     // channel within the slice
                    const f32x4 v = ctn_ld4(&Sg[r * HB_SROW + scol]);
                    const f32x4 n = hb_join3(Np, HB_NPLANE, hb_off(c, sch) + ssub);
                    const float g = Gs[c], be = Bes[c];
                    const int t = t0 + scol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (t + e < T) {
                            s1 = __builtin_fmaf(g, v[e], s1);
                            s2 = __builtin_fmaf(v[e], n[e] - be, s2);
                        }
                    }
                    if (t < T) ctn_st4(a.dn2 + (int64_t)b * a.dn2bs + (int64_t)(c0 + c) * a.lddn2 + t, v);
                }
#ifdef HB_TRACE_EPI
                HB_STAMP();
#endif
                HB_BARRIER();
            }
            HB_STAMP();
        }
        flush_sums(cur_b);
    }

    // ---- the slice's weight gradient and the bias gradients
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int mt = wm * TPW + j;
        const int c = c0 + 32 * wc + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * half;
            float* dst = m < M1 ? a.dW1 + (int64_t)m * a.lddw1 + c : a.dW2 + (int64_t)(m - M1) * a.lddw2 + c;
            atomicAdd(dst, accw[j][e]);
        }
    }
    if (sum_rows) {
        __syncthreads();
        if (tid < NKT * HB_KT) {
            if (tid < M1) {
                if (a.rowsum1) atomicAdd(a.rowsum1 + tid, Rs[tid]);
            } else if (a.rowsum2) {
                atomicAdd(a.rowsum2 + (tid - M1), Rs[tid]);
            }
        }
    }
}

}  // namespace

extern "C" int ctn_sizeof_heads_bwd_args(void) { return (int)sizeof(ctn_heads_bwd_args); }

extern "C" int ctn_heads_bwd(const ctn_heads_bwd_args* args, void* stream) {
    CTN_REQUIRE(args, "ctn_heads_bwd: null args");
    ctn_heads_bwd_args a = *args;
    const int M = a.M1 + a.M2;
    CTN_REQUIRE(a.batch > 0 && a.T >= 4 && a.H > 0 && a.H % HB_CS == 0, "ctn_heads_bwd: H must be a multiple of 128, T >= 4");
    CTN_REQUIRE(a.M1 >= 0 && a.M2 > 0 && a.M1 % 32 == 0 && a.M2 % 32 == 0 && (M == 128 || M == 256),
                "ctn_heads_bwd: M1 + M2 must be 128 or 256, both multiples of 32");
    auto act_ok = [&](const float* p, int64_t ld, int64_t bs) {
        return p && ctn_aligned16(p) && ld % 4 == 0 && bs % 4 == 0 && ld >= ((a.T + 3) & ~3);
    };
    CTN_REQUIRE((a.M1 == 0 || act_ok(a.dY1, a.lddy1, a.dy1bs)) && act_ok(a.dY2, a.lddy2, a.dy2bs) &&
                    act_ok(a.y2, a.ldy2, a.y2bs) && act_ok(a.dn2, a.lddn2, a.dn2bs),
                "ctn_heads_bwd: dY / y2 / dn2 layout (16-byte aligned, strides multiples of 4, rows of at least T rounded up to 4)");
    // the kernel addresses these tensors with 32-bit byte offsets
    auto span_ok = [&](int64_t ld, int64_t bs, int rows) {
        return ((int64_t)(a.batch - 1) * bs + (int64_t)rows * ld) * 4 < 0x7ffffff0ll && ld > 0 && bs >= 0;
    };
    CTN_REQUIRE((a.M1 == 0 || span_ok(a.lddy1, a.dy1bs, a.M1)) && span_ok(a.lddy2, a.dy2bs, a.M2) && span_ok(a.ldy2, a.y2bs, a.H),
                "ctn_heads_bwd: dY / y2 must span less than 2 GiB each");
    CTN_REQUIRE(a.W_prep && ctn_aligned16(a.W_prep) && a.alpha && a.mean && a.rstd && a.gamma && a.beta,
                "ctn_heads_bwd: prepared weights, slope, statistics, gamma, beta");
    CTN_REQUIRE((a.M1 == 0 || a.dW1) && a.dW2, "ctn_heads_bwd: weight-gradient targets");
    if (a.M1 == 0) {            // one head: its rows are the second tensor's
        a.dY1 = a.dY2; a.lddy1 = a.lddy2; a.dy1bs = a.dy2bs;
        a.dW1 = a.dW2; a.lddw1 = a.lddw2;
    }
    a.ldmin = (int32_t)(a.lddy1 < a.lddy2 ? a.lddy1 : a.lddy2);
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    const int nslices = a.H / HB_CS;
    const int ntt = ctn_cdiv(a.T, HB_TT);
    const int64_t ntiles = (int64_t)a.batch * ntt;
    CTN_REQUIRE(ntiles < 0x7fffffff, "ctn_heads_bwd: too many tiles");
    // one block per CU: every block ends with atomic adds of its whole 256 x 128 weight-gradient tile
    int64_t nparts = cus / nslices;
    if (nparts < 1) nparts = 1;
    if (nparts > ntiles) nparts = ntiles;
    const dim3 grid((unsigned)(nparts * nslices));
    hipStream_t s = (hipStream_t)stream;
    const void* kern = M == 256 ? reinterpret_cast<const void*>(&heads_bwd_kernel<8>)
                                : reinterpret_cast<const void*>(&heads_bwd_kernel<4>);
    // more dynamic LDS than a kernel gets by default (a host-side call per launch; no state is kept here)
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, HB_LDS) != hipSuccess) {
        ctn_set_error("ctn_heads_bwd: cannot raise the kernel's dynamic LDS limit to %d bytes", HB_LDS);
        return CTN_ERR_LAUNCH;
    }
    if (M == 256)
        hipLaunchKernelGGL(heads_bwd_kernel<8>, grid, dim3(HB_NTH), HB_LDS, s, a, nslices, ntt, (int)ntiles);
    else
        hipLaunchKernelGGL(heads_bwd_kernel<4>, grid, dim3(HB_NTH), HB_LDS, s, a, nslices, ntt, (int)ntiles);
    CTN_CHECK_LAUNCH("ctn_heads_bwd");
    return CTN_OK;
}
