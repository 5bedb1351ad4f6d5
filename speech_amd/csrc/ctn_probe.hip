// Measurement aids: what the chip itself sustains, measured with the simplest kernels that can reach it, so that the
// roofline fractions reported for the real kernels have a yardstick taken on the same GPU in the same process.
//   copy / 2-read-1-write : 16-byte non-temporal loads and stores, one block per contiguous 16 KiB, nothing else
//   bare MFMA loop        : v_mfma_f32_32x32x2_f32 back to back on four independent accumulators per wave, operands in
//                           registers (loaded once from random data), no LDS or memory traffic inside the loop
#include "ctn_common.h"

namespace {

// One block per contiguous 16 KiB chunk (1024 float4), every thread's four loads issued before its four stores: the
// access pattern of a whole-row streaming kernel, and the fastest of the plain-copy patterns tried on this GPU
// (grid-stride loops over the same buffers are 10-20 % slower).
__global__ __launch_bounds__(256) void probe_copy_kernel(const f32x4* __restrict__ a, f32x4* __restrict__ y, int64_t n4) {
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(a + (base + u * 256 < n4 ? base + u * 256 : n4 - 1));
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + u * 256 < n4) __builtin_nontemporal_store(v[u], y + base + u * 256);
}

// Plain (temporal) 16-byte copy, one block per contiguous 16 KiB chunk, chunks taken in ascending or descending address
// order by block index: what a consumer gains from reading a producer's output last-written-first (the die-level cache
// holds the most recently touched 256 MiB).
__global__ __launch_bounds__(256) void probe_stream_kernel(const f32x4* __restrict__ a, f32x4* __restrict__ y, int64_t n4,
                                                           const int reverse) {
    const int64_t blk = reverse ? (int64_t)(gridDim.x - 1 - blockIdx.x) : (int64_t)blockIdx.x;
    const int64_t base = blk * 1024 + threadIdx.x;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = a[base + u * 256 < n4 ? base + u * 256 : n4 - 1];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + u * 256 < n4) y[base + u * 256] = v[u];
}

__global__ __launch_bounds__(256) void probe_2r1w_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                         f32x4* __restrict__ y, int64_t n4) {
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    f32x4 va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * 256 < n4 ? base + u * 256 : n4 - 1;
        va[u] = __builtin_nontemporal_load(a + i);
        vb[u] = __builtin_nontemporal_load(b + i);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + u * 256 < n4) __builtin_nontemporal_store(va[u] + vb[u], y + base + u * 256);
}

__global__ __launch_bounds__(256) void probe_mfma_f32_kernel(const float* __restrict__ in, float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(tid * 8 + i) & 65535];
        b[i] = in[(tid * 8 + 4 + i) & 65535];
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c3, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[tid] = s;
}

// The same with v_mfma_f32_32x32x16_bf16: operands are bf16 pieces of the random input (8 per lane and operand).
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void probe_mfma_bf16_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              const int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    union { pbf16x8 v; unsigned u[4]; } a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[i].u[q] = __float_as_uint(in[(tid * 32 + i * 8 + q) & 65535]);          // two arbitrary bf16 bit patterns
            b[i].u[q] = __float_as_uint(in[(tid * 32 + i * 8 + 4 + q) & 65535]) & 0x7f7f7f7fu;   // (finite)
            a[i].u[q] &= 0x7f7f7f7fu;
        }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[2].v, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3].v, b[3].v, c3, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[tid] = s;
}

}  // namespace

extern "C" {

int ctn_probe_copy(const float* a, const float* b, float* y, int64_t n, void* stream) {
    CTN_REQUIRE(a && y && n > 0 && n % 4 == 0 && ctn_aligned16(a) && ctn_aligned16(y) && (!b || ctn_aligned16(b)),
                "ctn_probe_copy: needs 16-byte aligned buffers and n %% 4 == 0");
    CTN_REQUIRE(n / 4 <= (int64_t)1024 * 0x7fffffff, "ctn_probe_copy: n too large");
    const unsigned blocks = (unsigned)((n / 4 + 1023) / 1024);
    if (b)
        hipLaunchKernelGGL(probe_2r1w_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)a,
                           (const f32x4*)b, (f32x4*)y, n / 4);
    else
        hipLaunchKernelGGL(probe_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)a, (f32x4*)y,
                           n / 4);
    CTN_CHECK_LAUNCH("ctn_probe_copy");
    return CTN_OK;
}

int ctn_probe_stream(const float* a, float* y, int64_t n, int reverse, void* stream) {
    CTN_REQUIRE(a && y && n > 0 && n % 4 == 0 && ctn_aligned16(a) && ctn_aligned16(y),
                "ctn_probe_stream: needs 16-byte aligned buffers and n %% 4 == 0");
    CTN_REQUIRE(n / 4 <= (int64_t)1024 * 0x7fffffff, "ctn_probe_stream: n too large");
    const unsigned blocks = (unsigned)((n / 4 + 1023) / 1024);
    hipLaunchKernelGGL(probe_stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)a, (f32x4*)y,
                       n / 4, reverse);
    CTN_CHECK_LAUNCH("ctn_probe_stream");
    return CTN_OK;
}

int ctn_probe_mfma_f32(const float* in, float* out, int blocks, int iters, void* stream) {
    CTN_REQUIRE(in && out && blocks > 0 && iters > 0, "ctn_probe_mfma_f32: bad arguments");
    hipLaunchKernelGGL(probe_mfma_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, iters);
    CTN_CHECK_LAUNCH("ctn_probe_mfma_f32");
    return CTN_OK;
}

int ctn_probe_mfma_bf16(const float* in, float* out, int blocks, int iters, void* stream) {
    CTN_REQUIRE(in && out && blocks > 0 && iters > 0, "ctn_probe_mfma_bf16: bad arguments");
    hipLaunchKernelGGL(probe_mfma_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, iters);
    CTN_CHECK_LAUNCH("ctn_probe_mfma_bf16");
    return CTN_OK;
}

}  // extern "C"
