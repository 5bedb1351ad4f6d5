// Weight gradients of the residual layer's pointwise convolutions with the WHOLE output resident in one block, gfx950
// (same products as ctn_gemm_nt in ctn_gemm_nt.hip: the weight-gradient half of autograd for the Conv1d(B, H, 1) and
// the output / skip Conv1d(H, B, 1) pair of src/models/tdcn.py:118-120,172-175 of the reference).
//
//   C[m][n] += sum_b sum_t A'[b][m][t] * B'[b][n][t]        exact three-piece arithmetic (CTN_ARITH_BF16X3)
//
// ctn_gemm_nt's tile-per-block kernels split every operand fragment in each wave that reads it and spend 16.7 vector
// instructions per matrix instruction on it: they are bound by vector issue (72 % busy against 42 % for the matrix
// pipe).  Here one 512-thread block keeps a 256 x 256 (heads: dY against n2, two blocks for the 256 x 512 output) or
// 256 x 128 (bottleneck: the norm-backward operand against the layer input) output in its accumulators and walks
// stages of 16 (heads) or 32 (bottleneck) frames of the (batch, time) axis: every operand element is fetched,
// transformed and split ONCE on the chip (by the thread that loaded it) and stored as three bf16 planes, [row][frames],
// whose 16-byte chunks are the matrix instruction's fragments (lane: row lane % 32, frames 8 (lane / 32) .. + 7 of a
// k-step; chunk c of row r sits at chunk c ^ f(r), which keeps the fragment reads and the 8-byte staging stores free of
// bank conflicts).  4.2 vector instructions per matrix instruction remain.  Stages are dealt round-robin over the
// blocks of a launch (one block per CU, all resident), planes are double-buffered, the operands of the next stage are
// requested behind the barrier, and the barriers order LDS traffic only.  Half of the waves run a stage's two halves in
// the other order, so that on each SIMD one wave converts while the other multiplies.  Row sums of A' (bias gradients)
// and the PReLU-slope gradient of the norm-backward operand are formed by the staging threads, which see every element
// of A exactly once.
//
// With a workspace from the caller the blocks store their partial outputs (fragment order, 256-byte stores) and a
// second launch adds the 128 partials of each element in a fixed order: 211 + 15 us where the atomics route takes 257
// (every CU ends with a full-size partial, 256 KB, and 16.8 M float atomics at about one per clock and L2 channel have
// nothing running beneath them with one block per CU), and the same bits in C on every run.
//
// Measured on MI355X over 32 x 3999 frames (16 x 3999 in brackets), launch to launch: heads 239 us against 291 for the
// tile kernels (113 against 134), bottleneck 163 against 191 (94 against 104); the training step 943.6 -> 979.5
// utterances/s on one stream.  How it got there, from cycle stamps of one block (-DWH_TRACE, tools/trace_nt_whole.py):
// a first version (all waves in one order, 16-frame stages in both forms, two accumulation chains, atomics) ran at
// 282 / 235 us.  (1) 16-frame stages read 64 contiguous bytes per row and request, which this memory system serves at
// 3.3 TB/s at most against 6.7 for 128 bytes and more (tools/probe/rowseg_read.hip): the bottleneck form, three input
// streams, was at that limit, 32-frame stages took it to 191.  (2) The two loop bodies unrolled over the buffer pair
// kept two stages' registers live in the second wave order and spilled 17 registers, whose reloads wait on the
// prefetch in flight; one body with a run-time buffer index: 208 registers, 166 us.  (3) The wave orders, four
// accumulation chains and the request position were then set per form by measurement (see the loop).  What remains: a
// wave on its own issues a matrix instruction every 52 - 70 cycles, not 32 (fragment reads in between), the bottleneck
// form's conversions (2 x 1750 cycles of vector work per stage and SIMD against 3072 of matrix work) bound it, and
// the stored partials still cost 15 - 20 % of a launch.
#include <stdlib.h>

#include <atomic>

#include "ctn_common.h"

#ifdef WH_TRACE
// cycle stamps of block (0, 0, 0): kept in LDS while the kernel runs, copied out at its end
#define WH_TRN 192
__device__ unsigned long long wh_trace_buf[8 * WH_TRN];
#define WH_STAMP()                                                                   \
    do {                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                           \
        if (tracing) {                                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();              \
            if ((threadIdx.x & 63) == 0 && ntr < WH_TRN) trl[(threadIdx.x >> 6) * WH_TRN + ntr] = t_; \
            ++ntr;                                                                   \
        }                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                           \
    } while (0)
#define WH_WAITV() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
extern "C" int ctn_debug_whole_trace(void* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(wh_trace_buf), sizeof(wh_trace_buf)) == hipSuccess ? 0 : 1;
}
#else
#define WH_STAMP() do { } while (0)
#define WH_WAITV() do { } while (0)
#endif

namespace {

std::atomic<int64_t> whole_launches{0};      // host side: calls that took these kernels
constexpr int WNT = 512;          // threads per block
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));

#define WH_BARRIER()                                        \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                       \
        asm volatile("" ::: "memory");                      \
    } while (0)

// Pipeline settings, fixed per form by measurement (the macros exist for diagnostic builds):
//   WH_IP_*:     row tiles of a wave whose accumulations alternate (1: two independent chains, 2: four)
//   WH_AORDER_*: where the first half of the waves requests the next operands (0: in front of its matrix
//                instructions, the stage after the current one; 1: behind its stores, two stages ahead)
//   WH_SHIFT:    which waves run a stage's two halves in the other order (0: none; 1: waves 4..7; 2: odd waves; 3: all)
#ifndef WH_IP_BWD
#define WH_IP_BWD 1
#endif
#ifndef WH_IP_HEADS
#define WH_IP_HEADS 2
#endif
#ifndef WH_AORDER_BWD
#define WH_AORDER_BWD 0
#endif
#ifndef WH_AORDER_HEADS
#define WH_AORDER_HEADS 1
#endif
#ifndef WH_SHIFT
#define WH_SHIFT 1
#endif
#if WH_SHIFT == 1
#define WH_SHIFTED(w) ((w) >= 4)
#elif WH_SHIFT == 3
#define WH_SHIFTED(w) true
#elif WH_SHIFT == 2
#define WH_SHIFTED(w) ((w)&1)
#else
#define WH_SHIFTED(w) false
#endif

// x = hi + mid + lo exactly, each piece the next eight significant bits (truncation): the split of ctn_gemm_nn / _nt
__device__ __forceinline__ void wh_split3(const f32x4 v, wu32x2& hi, wu32x2& mid, wu32x2& lo) {
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

// BWD: A is the norm-backward operand (A, A2 = pre-activation, per-batch statistics, gamma), B plain: the bottleneck
// convolution's gradient.  !BWD: A plain (rows below M1 from A, the others from Ab), B = gamma (prelu(B) - mean) rstd + beta:
// the heads' gradient.  NWM x NWN waves, each WM x WN output tiles of 32 x 32; FR frames per stage (16 or 32).
template <bool BWD, int FR, int WM, int WN, int NWM, int NWN>
__global__ __launch_bounds__(WNT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_nt_whole_kernel(const ctn_gemm_nt_args a, const bool ws_ok) {
    static_assert(NWM * NWN == 8 && WM % 2 == 0, "eight waves, row tiles in pairs");
    static_assert(FR == 16 || FR == 32, "one or two k-steps per stage");
    constexpr int MB = 32 * WM * NWM, NB = 32 * WN * NWN;          // the block's output
    constexpr int RB = 2 * FR;                                     // bytes of a plane row
    constexpr int CH = FR / 8;                                     // 16-byte chunks of a plane row
    constexpr int SWS = CH == 2 ? 3 : 2;                           // chunk c of row r lives at chunk c ^ ((r >> SWS) % CH)
    constexpr int KS = FR / 16;                                    // k-steps of a stage
    constexpr int IP = (BWD ? WH_IP_BWD : WH_IP_HEADS);            // row tiles whose accumulations alternate
    constexpr bool LATE = (BWD ? WH_AORDER_BWD : WH_AORDER_HEADS) != 0;   // first half of the waves: request at the end of a step
    constexpr int PPR = FR / 4;                                    // float4 pieces of a row and stage
    constexpr int RPP = WNT / PPR;                                 // rows of one staging pass
    constexpr int APLANE = MB * RB, BPLANE = NB * RB;              // bytes of one piece plane of a stage
    constexpr int ABUF = 3 * APLANE, BBUF = 3 * BPLANE;
    constexpr int AQ = MB / RPP, BQ = NB / RPP;                    // float4 per thread and stage: rows tid / PPR + RPP q
    static_assert(AQ >= 1 && BQ >= 1, "tile / thread mapping");
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
    unsigned char* const Ap = wl;                                  // two buffers each
    unsigned char* const Bp = wl + 2 * ABUF;
    __shared__ double dred[WNT / 64];
#ifdef WH_TRACE
    __shared__ unsigned long long trl[8 * WH_TRN];
    int ntr = 0;
    const bool tracing = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wv % NWM, wn = wv / NWM;
    const int m0 = blockIdx.x * MB, n0 = blockIdx.y * NB;
    const int T = a.T, M1 = a.M1;
    const int spb = (T + FR - 1) / FR;                             // stages per batch element
    const int64_t nstages = (int64_t)a.batch * spb;
    const int nz = gridDim.z, bz = blockIdx.z;
    const int nst = (int)((nstages - bz + nz - 1) / nz);          // stages bz, bz + nz, ... of the (batch, time) axis

    // staging roles: float4 i = tid + WNT q of a stage: row i / PPR, frames 4 (i % PPR) .. + 3
    const int srow = tid / PPR, sfq = tid % PPR;
    const float a_alpha = BWD ? a.a_alpha[0] : 1.f;
    const float b_alpha = (!BWD && a.b_alpha) ? a.b_alpha[0] : 1.f;
    const float bplim = ctn_prelu_lim(b_alpha);
    float ag[AQ], bg[BQ], bb[BQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) ag[q] = BWD ? a.a_gamma[m0 + srow + RPP * q] : 0.f;
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
        bg[q] = (!BWD && (a.bflags & CTN_XF_NORM)) ? a.b_gamma[n0 + srow + RPP * q] : 1.f;
        bb[q] = (!BWD && (a.bflags & CTN_XF_NORM)) ? a.b_beta[n0 + srow + RPP * q] : 0.f;
    }
    // byte offsets inside a batch element's tensors: the thread's part (row srow of a pass, its piece) per leading
    // dimension, and the uniform part of each pass, added to the stage's frame offset as a scalar
    const int voAl = (srow * (int)a.lda + 4 * sfq) * 4, voAh = (srow * (int)a.ldab + 4 * sfq) * 4;
    const int voA2 = BWD ? (srow * (int)a.lda2 + 4 * sfq) * 4 : 0;
    const int voB = (srow * (int)a.ldb + 4 * sfq) * 4;
    bool a_low[AQ];                                              // uniform: M1 lies on the grid of the staging passes
    int soA[AQ], soA2[AQ], soB[BQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        const int m = m0 + RPP * q;
        a_low[q] = BWD || m < M1;                                // BWD: M1 = M
        soA[q] = (int)((a_low[q] ? (int64_t)m * a.lda : (int64_t)(m - M1) * a.ldab) * 4);
        soA2[q] = BWD ? (int)((int64_t)m * a.lda2 * 4) : 0;
    }
#pragma unroll
    for (int q = 0; q < BQ; ++q) soB[q] = (int)((int64_t)(n0 + RPP * q) * a.ldb * 4);
    const int sswz = (srow >> SWS) & (CH - 1), lswz = (l31 >> SWS) & (CH - 1);
    const int st_off = srow * RB + (((sfq >> 1) ^ sswz) * 16) + (sfq & 1) * 8;      // the thread's 8 bytes of a plane pass

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float rsum[AQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) rsum[q] = 0.f;
    double dalpha = 0.0;
    const bool reducer = blockIdx.y == 0;                          // blocks of the first column tile own the A-side sums

    // one stage's operands in registers, with the scalars its transform needs
    struct Stage {
        wu32x4 va[AQ], va2[BWD ? AQ : 1], vb[BQ];
        float s0, s1, s2;        // BWD: rstd, c2, c1 - mean c2;  !BWD: rstd, mean (of B)
        int t0;
    };
    // stages bz, bz + nz, ...: (batch element, stage inside it) of the next request, advanced without divisions
    int nb = bz / spb, ns = bz % spb;
    auto fetch = [&](const int /* stage index: the cursor below keeps it */, Stage& st) {
        const bool live = nb < a.batch;
        const int b = live ? nb : 0;
        const int t0 = live ? ns * FR : T;
        st.t0 = t0;                                              // a stage behind the last one: every frame masked
        const int so = live ? t0 * 4 : 0;
        ns += nz;
        while (ns >= spb) ns -= spb, ++nb;
        // pieces that start at or behind frame T are not read: their offset lies outside the descriptor's range and
        // the load returns zeros without touching memory (rows are padded to whole float4 only)
        const bool in = t0 + 4 * sfq < T;
        const int oAl = in ? voAl : (int)0x80000000, oAh = in ? voAh : (int)0x80000000;
        const int oA2 = in ? voA2 : (int)0x80000000, oB = in ? voB : (int)0x80000000;
        // descriptors of this batch element's tensors (2 GiB windows; every offset used lies inside the tensors)
        const auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A ? a.A + (int64_t)b * a.abs_ : a.Ab), 0, 0x7ffffffc, 0x00020000);
        const auto rAb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Ab ? a.Ab + (int64_t)b * a.abbs : a.A), 0, 0x7ffffffc, 0x00020000);
        const auto rB = __builtin_amdgcn_make_buffer_rsrc((void*)(a.B + (int64_t)b * a.bbs), 0, 0x7ffffffc, 0x00020000);
#pragma unroll
        for (int q = 0; q < AQ; ++q)
            st.va[q] = a_low[q] ? __builtin_amdgcn_raw_buffer_load_b128(rA, oAl, soA[q] + so, 0)
                                : __builtin_amdgcn_raw_buffer_load_b128(rAb, oAh, soA[q] + so, 0);
        if (BWD) {
            const auto rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A2 + (int64_t)b * a.a2bs), 0, 0x7ffffffc, 0x00020000);
#pragma unroll
            for (int q = 0; q < AQ; ++q) st.va2[q] = __builtin_amdgcn_raw_buffer_load_b128(rA2, oA2, soA2[q] + so, 0);
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) st.vb[q] = __builtin_amdgcn_raw_buffer_load_b128(rB, oB, soB[q] + so, 0);
        if (BWD) {
            const int64_t si = (int64_t)b * a.a_ssb;
            const float mean = a.a_mean[si], c1 = a.a_c1[si];
            st.s0 = a.a_rstd[si];
            st.s1 = a.a_c2[si];
            st.s2 = c1 - mean * st.s1;
        } else if (a.bflags & CTN_XF_NORM) {
            const int64_t si = (int64_t)b * a.b_ssb;
            st.s0 = a.b_rstd[si];
            st.s1 = a.b_mean[si];
            st.s2 = 0.f;
        } else {
            st.s0 = 1.f, st.s1 = 0.f, st.s2 = 0.f;
        }
    };
    auto stage = [&](const Stage& st, const int buf) {
        const int tf = st.t0 + 4 * sfq;                         // first frame of this thread's float4
        unsigned char* Ab_ = Ap + buf * ABUF + st_off;
        unsigned char* Bb_ = Bp + buf * BBUF + st_off;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            f32x4 v = __builtin_bit_cast(f32x4, st.va[q]);
            if (BWD) {
                const f32x4 y = __builtin_bit_cast(f32x4, st.va2[q]);
                const float xg = ag[q] * st.s0;
                float da = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sl = y[e] > 0.f ? 1.f : a_alpha;                     // PReLU slope at y
                    const float du = __builtin_fmaf(xg, v[e], __builtin_fmaf(sl * y[e], st.s1, st.s2));
                    const bool valid = tf + e < T;
                    v[e] = valid ? sl * du : 0.f;
                    if (valid && !(y[e] > 0.f)) da += du * y[e];
                }
                if (reducer) dalpha += (double)da;          // four terms in fp32, the signed terms above them in fp64
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tf + e < T ? v[e] : 0.f;
            }
            rsum[q] += (v[0] + v[1]) + (v[2] + v[3]);
            wu32x2 h, m, l;
            wh_split3(v, h, m, l);
            unsigned char* o = Ab_ + q * RPP * RB;
            *reinterpret_cast<wu32x2*>(o) = h;
            *reinterpret_cast<wu32x2*>(o + APLANE) = m;
            *reinterpret_cast<wu32x2*>(o + 2 * APLANE) = l;
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            f32x4 v = __builtin_bit_cast(f32x4, st.vb[q]);
            const float g = bg[q] * st.s0, be = bb[q] - st.s1 * g;       // (u - mean) rstd gamma + beta = u g + be
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w = v[e];
                if (!BWD) {
                    if (a.bflags & CTN_XF_PRELU) w = ctn_prelu_med(w, b_alpha, bplim);
                    if (a.bflags & CTN_XF_NORM) w = __builtin_fmaf(w, g, be);
                }
                v[e] = tf + e < T ? w : 0.f;
            }
            wu32x2 h, m, l;
            wh_split3(v, h, m, l);
            unsigned char* o = Bb_ + q * RPP * RB;
            *reinterpret_cast<wu32x2*>(o) = h;
            *reinterpret_cast<wu32x2*>(o + BPLANE) = m;
            *reinterpret_cast<wu32x2*>(o + 2 * BPLANE) = l;
        }
    };
    auto compute = [&](const int buf) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks) __builtin_amdgcn_sched_barrier(0);           // keep one k-step's fragments live at a time
            const int coff = ((2 * ks + half) ^ lswz) * 16;      // this lane's eight frames of the k-step
            const unsigned char* Ab_ = Ap + buf * ABUF + (32 * WM * wm + l31) * RB + coff;
            const unsigned char* Bb_ = Bp + buf * BBUF + (32 * WN * wn + l31) * RB + coff;
            wbf16x8 bf[WN][3];
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const wbf16x8*>(Bb_ + pl * BPLANE + j * 32 * RB);
#pragma unroll
            for (int i0 = 0; i0 < WM; i0 += IP) {
                wbf16x8 af[IP][3];
#pragma unroll
                for (int ii = 0; ii < IP; ++ii)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        af[ii][pl] = *reinterpret_cast<const wbf16x8*>(Ab_ + pl * APLANE + (i0 + ii) * 32 * RB);
                // smallest products first; the tiles of two rows alternate so that four independent accumulations are
                // in flight (a wave on its own otherwise waits on each instruction's predecessor in its chain)
#define WH_MFMA(a_, b_)                                                                                   \
    _Pragma("unroll") for (int ii = 0; ii < IP; ++ii)                                                     \
        _Pragma("unroll") for (int j = 0; j < WN; ++j)                                                    \
            acc[i0 + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ii][a_], bf[j][b_], acc[i0 + ii][j], 0, 0, 0)
                WH_MFMA(2, 0);
                WH_MFMA(0, 2);
                WH_MFMA(1, 1);
                WH_MFMA(1, 0);
                WH_MFMA(0, 1);
                WH_MFMA(0, 0);
#undef WH_MFMA
            }
        }
    };

    if (nst > 0) {
        // step g: [barrier] matrix instructions of g (buffer g & 1) -> transform, split, store g + 1, with the request
        // for g + 1 in front of the matrix instructions or (LATE) the one for g + 2 behind the stores.  The second half
        // of the waves does the same work in the other order (store g + 1, request g + 2, matrix instructions of g), so
        // that on each SIMD one wave's conversions run under the other's matrix instructions.  Which request position
        // and how many alternating accumulation chains is measured per form: heads 275 us (early, two chains) -> 257
        // (late, four); bottleneck 163 us (early, two) -> 185 (late, four).
        Stage s0, s1;
        fetch(0, s0);
        stage(s0, 0);
        if (!WH_SHIFTED(wv)) {
            if (LATE) {
                fetch(1, s1);
                for (int g = 0; g < nst; ++g) {
                    const int buf = g & 1;
                    WH_BARRIER();
                    WH_STAMP();
                    compute(buf);
                    WH_STAMP();
                    WH_WAITV();
                    WH_STAMP();
                    stage(s1, buf ^ 1);                  // behind the block's last stage: all zeros, never multiplied
                    WH_STAMP();
                    __builtin_amdgcn_sched_barrier(0);           // the requests reuse the registers just stored
                    fetch(g + 2, s1);
                }
            } else {
                for (int g = 0; g < nst; ++g) {
                    const int buf = g & 1;
                    WH_BARRIER();
                    WH_STAMP();
                    fetch(g + 1, s1);
                    compute(buf);
                    WH_STAMP();
                    WH_WAITV();
                    WH_STAMP();
                    stage(s1, buf ^ 1);
                    WH_STAMP();
                }
            }
        } else {
            fetch(1, s0);
            for (int g = 0; g < nst; ++g) {
                const int buf = g & 1;
                WH_BARRIER();
                WH_STAMP();
                WH_WAITV();
                WH_STAMP();
                stage(s0, buf ^ 1);
                WH_STAMP();
                __builtin_amdgcn_sched_barrier(0);               // the requests reuse the registers just stored
                fetch(g + 2, s0);
                compute(buf);
                WH_STAMP();
            }
        }
    }
    WH_STAMP();

    // ---- partial output: to the workspace in fragment order (64 consecutive floats per store; the reduction kernel
    // below adds the blocks' partials in a fixed order), or with float atomics straight into C
    constexpr int64_t PS = (int64_t)MB * NB + MB;                 // floats of one block's partial: tile, then row sums
    float* const wsb = ws_ok ? a.ws + ((int64_t)bz * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * PS : nullptr;
    if (wsb != nullptr) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) wsb[(((wv * WM + i) * WN + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    } else {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = n0 + 32 * (WN * wn + j) + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + 32 * (WM * wm + i) + (e & 3) + 8 * (e >> 2) + 4 * half;
                    float* p = (m < M1) ? a.C + (int64_t)m * a.ldc + n : a.C2 + (int64_t)(m - M1) * a.ldc2 + n;
                    atomicAdd(p, acc[i][j][e]);
                }
            }
    }
    // ---- A-side sums: the PPR threads of a row hold its pieces
    if (reducer && (a.a_rowsum != nullptr || a.a_rowsum2 != nullptr)) {
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            float sv = rsum[q];
#pragma unroll
            for (int d = 1; d < PPR; d <<= 1) sv += __shfl_xor(sv, d);
            const int m = m0 + srow + RPP * q;
            if (sfq == 0) {
                if (wsb != nullptr) {
                    wsb[(int64_t)MB * NB + srow + RPP * q] = sv;
                } else if (m < M1) {
                    if (a.a_rowsum) atomicAdd(a.a_rowsum + m, sv);
                } else if (a.a_rowsum2) {
                    atomicAdd(a.a_rowsum2 + (m - M1), sv);
                }
            }
        }
    }
    if (BWD && reducer && a.a_dalpha != nullptr) {
        const double tot = ctn_block_sum_f64(dalpha, dred);
        if (tid == 0) atomicAdd(a.a_dalpha + ctn_slot(), tot);
    }
#ifdef WH_TRACE
    WH_STAMP();
    if (tracing) {
        __syncthreads();
        for (int i = tid; i < 8 * WH_TRN; i += WNT) wh_trace_buf[i] = (i % WH_TRN) < ntr ? trl[i] : 0ull;
    }
#endif
}

// Adds the nz partial outputs of every tile in a fixed order -- four threads take a quarter of the partials each, in
// the order of the blocks' indices, and the quarters are added in order -- and the sum to C (one float atomic per
// output element: another launch may be adding to the same weight gradient from another stream).  A block takes 256
// consecutive elements of the partials' fragment order as 64 float4; the last MB / 64 blocks of a tile take its row sums.
template <int WM, int WN, int NWM, int NWN>
__global__ __launch_bounds__(256) void gemm_nt_whole_reduce_kernel(const ctn_gemm_nt_args a, const int nz, const int tiles_m,
                                                                   const int tiles) {
    constexpr int MB = 32 * WM * NWM, NB = 32 * WN * NWN;
    constexpr int64_t PS = (int64_t)MB * NB + MB;
    constexpr int NBLK = MB * NB / 256;
    __shared__ float part[4][256];
    const int tile = blockIdx.y, tm = tile % tiles_m, tn = tile / tiles_m;
    const int m0 = tm * MB, n0 = tn * NB;
    const float* src = a.ws + (int64_t)tile * PS;
    const int64_t zs = (int64_t)tiles * PS;
    const int tid = threadIdx.x, zg = tid >> 6, l = tid & 63;
    const int zper = (nz + 3) / 4, z0 = zg * zper, z1 = min(z0 + zper, nz);
    if (blockIdx.x < NBLK) {
        const float* sp = src + blockIdx.x * 256 + l * 4;
        f32x4 sv = {0.f, 0.f, 0.f, 0.f};
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ctn_ld4(sp + (z + u) * zs);
#pragma unroll
            for (int u = 0; u < 8; ++u) sv += v[u];
        }
        for (; z < z1; ++z) sv += ctn_ld4(sp + z * zs);
#pragma unroll
        for (int c = 0; c < 4; ++c) part[zg][l * 4 + c] = sv[c];
        __syncthreads();
        const float tot = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
        const int p = blockIdx.x * 256 + tid;
        const int lane = p & 63, e = (p >> 6) & 15, t = p >> 10;
        const int j = t % WN, i = (t / WN) % WM, wave = t / (WN * WM);
        const int wm = wave % NWM, wn = wave / NWM;
        const int m = m0 + 32 * (WM * wm + i) + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int n = n0 + 32 * (WN * wn + j) + (lane & 31);
        float* dst = (m < a.M1) ? a.C + (int64_t)m * a.ldc + n : a.C2 + (int64_t)(m - a.M1) * a.ldc2 + n;
        atomicAdd(dst, tot);
    } else if (tn == 0) {
        const int r = (blockIdx.x - NBLK) * 64 + l;               // row of the tile
        const float* sp = src + (int64_t)MB * NB + r;
        float sv = 0.f;
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sp[(z + u) * zs];
#pragma unroll
            for (int u = 0; u < 8; ++u) sv += v[u];
        }
        for (; z < z1; ++z) sv += sp[z * zs];
        part[zg][l] = sv;
        __syncthreads();
        if (zg == 0) {
            const float tot = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
            const int m = m0 + r;
            if (m < a.M1) {
                if (a.a_rowsum) atomicAdd(a.a_rowsum + m, tot);
            } else if (a.a_rowsum2) {
                atomicAdd(a.a_rowsum2 + (m - a.M1), tot);
            }
        }
    }
}

// blocks along the (batch, time) axis of a launch, and the floats of workspace that lets them store their partials
template <int FR, int MB, int NB>
int64_t whole_nz(const ctn_gemm_nt_args& a, int cus, int64_t* ws_floats) {
    const int tiles = (a.M / MB) * (a.N / NB);
    const int64_t nstages = (int64_t)a.batch * ((a.T + FR - 1) / FR);
    int64_t nz = cus / tiles;                               // one block per CU, all resident at once
    if (nz < 1) nz = 1;
    if (nz > (nstages + 3) / 4) nz = (nstages + 3) / 4;       // at least four stages per block
    if (nz > 65535) nz = 65535;
    *ws_floats = nz * tiles * ((int64_t)MB * NB + MB);
    return nz;
}

template <bool BWD, int FR, int WM, int WN, int NWM, int NWN>
int launch_whole(const ctn_gemm_nt_args& a, hipStream_t s, int cus, int64_t* ws_query) {
    constexpr int MB = 32 * WM * NWM, NB = 32 * WN * NWN;
    constexpr size_t shm = (size_t)2 * 3 * 2 * FR * (MB + NB);
    const int tiles_m = a.M / MB, tiles = tiles_m * (a.N / NB);
    int64_t need = 0;
    const int64_t nz = whole_nz<FR, MB, NB>(a, cus, &need);
    if (ws_query) {
        *ws_query = need;
        return 1;
    }
    const bool ws_ok = a.ws != nullptr && a.ws_floats >= need && ctn_aligned16(a.ws);
    const void* fn = reinterpret_cast<const void*>(&gemm_nt_whole_kernel<BWD, FR, WM, WN, NWM, NWN>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) return 0;
    hipLaunchKernelGGL((gemm_nt_whole_kernel<BWD, FR, WM, WN, NWM, NWN>), dim3(a.M / MB, a.N / NB, (unsigned)nz), dim3(WNT), shm, s, a, ws_ok);
    if (ws_ok)
        hipLaunchKernelGGL((gemm_nt_whole_reduce_kernel<WM, WN, NWM, NWN>), dim3(MB * NB / 256 + MB / 64, tiles), dim3(256), 0, s, a,
                           (int)nz, tiles_m, tiles);
    ++whole_launches;
    return 1;
}

}  // namespace

extern "C" int64_t ctn_gemm_nt_whole_launches(void) { return whole_launches.load(); }

// Launches the whole-output kernel when the call is one of its two forms (CTN_NT_WHOLE in the environment, read at
// every call: 0 never, 2 the bottleneck form only, 3 the heads' form only); returns 0 (nothing launched) otherwise, and
// the caller goes on to the tile-per-block kernels.  With ws_query the call launches nothing and reports the workspace
// the launch would use.
int ctn_gemm_nt_whole_try(const ctn_gemm_nt_args& a, hipStream_t s, int64_t* ws_query) {
    const char* const on = getenv("CTN_NT_WHOLE");
    const char mode = on ? on[0] : '1';
    if (mode < '1' || mode > '3') return 0;
    const bool want_bwd = mode != '3', want_heads = mode != '2';
    if (a.arith != CTN_ARITH_DEFAULT && a.arith != CTN_ARITH_BF16X3) return 0;
    if (a.a_rowsum_b || a.wdot_sums) return 0;
    const bool bwd = (a.aflags & CTN_XF_NORMBWD) != 0;
    if (bwd ? !want_bwd : !want_heads) return 0;
    if ((a.aflags & ~CTN_XF_NORMBWD) || (a.bflags & ~(CTN_XF_PRELU | CTN_XF_NORM))) return 0;
    if (bwd && (a.bflags != 0 || a.a_sst != 0 || a.M1 != a.M || !a.a_alpha)) return 0;
    if (!bwd && (a.bflags & CTN_XF_NORM) && a.b_sst != 0) return 0;          // per-frame statistics: the tile kernels
    if (!bwd && (a.bflags & CTN_XF_PRELU) && !a.b_alpha) return 0;
    // layout: 16-byte aligned rows, and every tensor of a batch element inside a 2 GiB window
    auto ok = [&](const float* p, int64_t ld, int64_t bs, int rows) {
        return p == nullptr ? rows == 0
                            : (ctn_aligned16(p) && ld % 4 == 0 && bs % 4 == 0 && ld >= ((a.T + 3) & ~3) &&
                               (int64_t)rows * ld * 4 < 0x7ff00000ll);
    };
    if (!ok(a.A, a.lda, a.abs_, a.M1) || !ok(a.Ab, a.ldab, a.abbs, a.M - a.M1) || !ok(a.B, a.ldb, a.bbs, a.N)) return 0;
    if (bwd && !ok(a.A2, a.lda2, a.a2bs, a.M)) return 0;
    if (a.M1 % 128 != 0 && a.M1 != a.M) return 0;              // a staging thread's rows lie 128 apart: keep the split on that grid
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    // enough stages to give every CU work: below that the tile kernels' finer split does better
    if (bwd) {
        const int64_t nstages = (int64_t)a.batch * ((a.T + 31) / 32);
        if (a.M % 256 != 0 || a.N != 128) return 0;
        if (nstages < 4 * (int64_t)(cus / (a.M / 256))) return 0;
        return launch_whole<true, 32, 2, 2, 4, 2>(a, s, cus, ws_query);
    }
    if (a.M % 256 == 0 && a.N == 128) {                        // tall outputs (the mask convolution, 1024 x 128: 266 us against 310)
        const int64_t nstages = (int64_t)a.batch * ((a.T + 31) / 32);
        if (nstages < 4 * (int64_t)(cus / (a.M / 256))) return 0;
        return launch_whole<false, 32, 2, 2, 4, 2>(a, s, cus, ws_query);
    }
    // (128 x 512 outputs -- a single head, the entry bottleneck -- stay with the tile kernels: a 128 x 256 block form
    // of this kernel measured 154 - 165 us against their 145)
    const int64_t nstages = (int64_t)a.batch * ((a.T + 15) / 16);
    if (a.M != 256 || a.N % 256 != 0) return 0;
    if (nstages < 8 * (int64_t)(cus / (a.N / 256))) return 0;
    return launch_whole<false, 16, 4, 2, 2, 4>(a, s, cus, ws_query);
}

extern "C" int64_t ctn_gemm_nt_ws_floats(const ctn_gemm_nt_args* args) {
    int64_t need = 0;
    if (args == nullptr || !ctn_gemm_nt_whole_try(*args, nullptr, &need)) return 0;
    return need;
}
