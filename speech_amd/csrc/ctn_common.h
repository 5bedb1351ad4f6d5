// Internal helpers shared by the gfx950 kernels of libctn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ctn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void ctn_set_error(const char* fmt, ...);

#define CTN_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            ctn_set_error(__VA_ARGS__);   \
            return CTN_ERR_ARG;           \
        }                                 \
    } while (0)

#define CTN_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            ctn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return CTN_ERR_LAUNCH;                                               \
        }                                                                        \
    } while (0)

static inline bool ctn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int ctn_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#define CTN_WAVE 64

// Scalar fp64 accumulators (norm sums, PReLU-slope gradients) are CTN_STAT_SLOTS-way replicated: thousands of
// blocks adding to ONE address serialise at the memory side, so each block adds to the slot picked by its
// block index and the (tiny) consumer kernels / the host sum the slots.
__device__ __forceinline__ int ctn_slot() {
    return (int)((blockIdx.x + 5u * blockIdx.y + 11u * blockIdx.z) & (CTN_STAT_SLOTS - 1));
}

__device__ __forceinline__ float ctn_prelu(float v, float a) { return v > 0.f ? v : a * v; }
// PReLU in two vector instructions for MFMA loops, where every vector slot counts: for a slope a <= 1 the result is
// max(v, a*v), for a > 1 it is min(v, a*v); both are the median of (v, a*v, lim) with lim = +inf / -inf.  The value
// is one of v and a*v, bit-identical to ctn_prelu.
__device__ __forceinline__ float ctn_prelu_lim(float a) { return a <= 1.f ? __builtin_inff() : -__builtin_inff(); }
__device__ __forceinline__ float ctn_prelu_med(float v, float a, float lim) { return __builtin_amdgcn_fmed3f(v, a * v, lim); }

__device__ __forceinline__ float ctn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double ctn_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum NV per-thread values over the block.  After the call threads 0..NV-1 hold total i in their
// return value (thread i -> v[i] summed); smem needs NV * (blockDim.x / 64) floats.
template <int NV>
__device__ __forceinline__ float ctn_block_sum(float (&v)[NV], float* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = ctn_wave_sum(v[i]);
        if (lane == 0) smem[wave * NV + i] = s;
    }
    __syncthreads();
    float tot = 0.f;
    if ((int)threadIdx.x < NV) {
        for (int w = 0; w < nw; ++w) tot += smem[w * NV + threadIdx.x];
    }
    __syncthreads();
    return tot;
}

// fp64 block sum of one value per thread (slope gradients: signed terms that cancel, so fp32 partial sums cost
// 1e-4 of the result).  Thread 0 holds the total; smem needs blockDim.x / 64 doubles.
__device__ __forceinline__ double ctn_block_sum_f64(double v, double* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const double s = ctn_wave_sum(v);
    if (lane == 0) smem[wave] = s;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < nw; ++w) tot += smem[w];
    __syncthreads();
    return tot;
}

// Finishing a gLN reduction in the producing launch (ctn_finish in include/ctn.h).  EVERY thread of EVERY block that
// belongs to batch element b calls this once, after the block's additions to sums_b (the element's
// [CTN_STAT_SLOTS][2] doubles) have been issued; `nblocks` is the number of such blocks.  All values travel through
// device-scope atomics -- additions, the ticket, the final loads -- so "my additions have been performed" (vmcnt(0))
// before "I take my ticket" is all the ordering the hand-over needs: no cache has to be written back or invalidated.
__device__ __forceinline__ void ctn_finish_sums(const ctn_finish& f, const double* sums_b, int b, unsigned nblocks,
                                                unsigned* flag /* one word of LDS */) {
    if (f.count == nullptr) return;                        // uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(f.count + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = ticket == nblocks - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u || threadIdx.x != 0) return;
    double S1 = 0.0, S2 = 0.0;                             // slot order and arithmetic of the stand-alone kernels
    for (int s = 0; s < CTN_STAT_SLOTS; ++s) {
        S1 += __hip_atomic_load(sums_b + 2 * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S2 += __hip_atomic_load(sums_b + 2 * s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (f.kind == CTN_FIN_STATS) {
        const double m = S1 / f.n;
        double var = S2 / f.n - m * m;
        if (var < 0.0) var = 0.0;
        f.out0[b] = (float)m;
        f.out1[b] = (float)(1.0 / sqrt(var + (double)f.eps));
    } else {
        const double r = f.rstd[b];
        f.out0[b] = (float)(-r * S1 / f.n);
        f.out1[b] = (float)(-r * r * S2 / f.n);
    }
}

// Host-side check of a ctn_finish request.
static inline bool ctn_finish_ok(const ctn_finish& f) {
    if (f.count == nullptr) return true;
    return f.out0 && f.out1 && f.n > 0 && (f.kind == CTN_FIN_STATS || (f.kind == CTN_FIN_COEFS && f.rstd));
}

__device__ __forceinline__ f32x4 ctn_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void ctn_st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// streaming variants (touched once by this kernel): non-temporal hint, the lines do not displace reusable data in L2
__device__ __forceinline__ f32x4 ctn_ld4_nt(const float* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}
__device__ __forceinline__ void ctn_st4_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }

// Per-frame statistics of 4 consecutive frames starting at t (gLN: stride 0 -> one value).
struct Stat4 {
    float mean[4], rstd[4], c1[4], c2[4];
};
