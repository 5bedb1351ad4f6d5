// Version / error reporting and the weight-layout helper of libctn_hip.so.
#include <stdarg.h>

#include "ctn_common.h"

static thread_local char g_err[512] = "";

void ctn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

// dst[k*ldd + coff + m] = src[m*K + k], through a 32x33 LDS tile so both sides are coalesced.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int M, int K, float* dst,
                                                        int64_t ldd, int64_t coff) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r, k = k0 + tx;
        tile[r][tx] = (m < M && k < K) ? src[(int64_t)m * K + k] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, m = m0 + tx;
        if (k < K && m < M) dst[(int64_t)k * ldd + coff + m] = tile[tx][r];
    }
}

constexpr int TB_ITEMS = 64;
struct transpose_batch {
    ctn_transpose_item item[TB_ITEMS];
    int32_t first_block[TB_ITEMS + 1];
    int32_t n;
};

// One 32 x 32 tile per block; the block finds its matrix in the (by-value) table.
__global__ __launch_bounds__(256) void transpose_batch_kernel(const transpose_batch tb) {
    __shared__ float tile[32][33];
    int it = 0;
    while (it + 1 < tb.n && (int)blockIdx.x >= tb.first_block[it + 1]) ++it;
    const ctn_transpose_item& d = tb.item[it];
    const int local = (int)blockIdx.x - tb.first_block[it];
    const int tiles_k = (d.K + 31) >> 5;
    const int m0 = (local / tiles_k) * 32, k0 = (local % tiles_k) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r, k = k0 + tx;
        tile[r][tx] = (m < d.M && k < d.K) ? d.src[(int64_t)m * d.K + k] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, m = m0 + tx;
        if (k < d.K && m < d.M) d.dst[(int64_t)k * d.ldd + d.coff + m] = tile[tx][r];
    }
}

}  // namespace

extern "C" {

const char* ctn_version(void) { return "ctn-hip 0.1 (gfx950)"; }
const char* ctn_last_error(void) { return g_err; }

int ctn_transpose(const float* src, int M, int K, float* dst, int64_t ldd, int64_t coff, void* stream) {
    CTN_REQUIRE(src && dst && M > 0 && K > 0 && ldd >= coff + M, "ctn_transpose: bad arguments");
    dim3 grid(ctn_cdiv(K, 32), ctn_cdiv(M, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, M, K, dst, ldd, coff);
    CTN_CHECK_LAUNCH("ctn_transpose");
    return CTN_OK;
}

int ctn_sizeof_transpose_item(void) { return (int)sizeof(ctn_transpose_item); }

int ctn_transpose_batch(const ctn_transpose_item* items, int n_items, void* stream) {
    CTN_REQUIRE(items && n_items > 0, "ctn_transpose_batch: no items");
    for (int i = 0; i < n_items; ++i) {
        const ctn_transpose_item& d = items[i];
        CTN_REQUIRE(d.src && d.dst && d.M > 0 && d.K > 0 && d.coff >= 0 && d.ldd >= d.coff + d.M,
                    "ctn_transpose_batch: item %d: bad arguments", i);
    }
    for (int base = 0; base < n_items; base += TB_ITEMS) {
        transpose_batch tb;
        tb.n = n_items - base < TB_ITEMS ? n_items - base : TB_ITEMS;
        int64_t blocks = 0;
        for (int i = 0; i < tb.n; ++i) {
            tb.item[i] = items[base + i];
            tb.first_block[i] = (int32_t)blocks;
            blocks += (int64_t)ctn_cdiv(items[base + i].M, 32) * ctn_cdiv(items[base + i].K, 32);
            CTN_REQUIRE(blocks < 0x7fffffff, "ctn_transpose_batch: too much work for one launch");
        }
        for (int i = tb.n; i <= TB_ITEMS; ++i) tb.first_block[i] = (int32_t)blocks;
        hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tb);
        CTN_CHECK_LAUNCH("ctn_transpose_batch");
    }
    return CTN_OK;
}

}  // extern "C"

// Struct sizes, so a foreign-language binding can verify its mirror of include/ctn.h at load time.
extern "C" {
int ctn_sizeof_gemm_nn_args(void) { return (int)sizeof(ctn_gemm_nn_args); }
int ctn_sizeof_gemm_nt_args(void) { return (int)sizeof(ctn_gemm_nt_args); }
int ctn_sizeof_dw_fwd_args(void) { return (int)sizeof(ctn_dw_fwd_args); }
int ctn_sizeof_dw_bwd_args(void) { return (int)sizeof(ctn_dw_bwd_args); }
}
