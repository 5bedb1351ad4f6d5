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

}  // extern "C"

// Struct sizes, so a foreign-language binding can verify its mirror of include/ctn.h at load time.
extern "C" {
int ctn_sizeof_gemm_nn_args(void) { return (int)sizeof(ctn_gemm_nn_args); }
int ctn_sizeof_gemm_nt_args(void) { return (int)sizeof(ctn_gemm_nt_args); }
int ctn_sizeof_dw_fwd_args(void) { return (int)sizeof(ctn_dw_fwd_args); }
int ctn_sizeof_dw_bwd_args(void) { return (int)sizeof(ctn_dw_bwd_args); }
}
