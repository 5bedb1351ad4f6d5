// Batched strided 2-D transposes of fp32 activations, optionally with an addend in the destination's layout: the axis
// shuffles between the dual-path blocks' (batch, features, S, K) tensors and the (sequences, steps, features) rows of
// their recurrences (src/models/dprnn.py:83,88 inside chunks, :135,140,144-146 across chunks of the reference).
#include "ctn_common.h"

namespace {

// dst[b2][b1][k][m] = src[b2][b1][m][k] (+ addend[b2][b1][k][m]): rows m of the source are lds apart, rows k of the
// destination ldd apart, the two batch levels have their own strides on either side.  32 x 32 tiles through LDS (33 floats
// a row), so reads run along k and writes along m whatever the strides are.
__global__ __launch_bounds__(256) void permute_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                      const float* __restrict__ addend, const int M, const int K,
                                                      const int64_t lds, const int64_t ldd, const int nb1,
                                                      const int64_t sb1, const int64_t db1, const int64_t sb2,
                                                      const int64_t db2) {
    __shared__ float tile[32][33];
    const int b1 = (int)(blockIdx.z % (unsigned)nb1), b2 = (int)(blockIdx.z / (unsigned)nb1);
    const int64_t so = (int64_t)b1 * sb1 + (int64_t)b2 * sb2, dof = (int64_t)b1 * db1 + (int64_t)b2 * db2;
    const int m0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r, k = k0 + tx;
        tile[r][tx] = (m < M && k < K) ? src[so + (int64_t)m * lds + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, m = m0 + tx;
        if (k < K && m < M) {
            const int64_t o = dof + (int64_t)k * ldd + m;
            float v = tile[tx][r];
            if (addend) v += addend[o];
            dst[o] = v;
        }
    }
}

}  // namespace

extern "C" int ctn_permute(const float* src, float* dst, const float* addend, int M, int K, int64_t lds, int64_t ldd,
                           int nb1, int64_t sb1, int64_t db1, int nb2, int64_t sb2, int64_t db2, void* stream) {
    CTN_REQUIRE(src && dst, "ctn_permute: null pointer");
    CTN_REQUIRE(M > 0 && K > 0 && nb1 > 0 && nb2 > 0, "ctn_permute: empty shape");
    CTN_REQUIRE(lds >= K && ldd >= M, "ctn_permute: a row stride shorter than the row");
    CTN_REQUIRE(sb1 >= 0 && db1 >= 0 && sb2 >= 0 && db2 >= 0, "ctn_permute: negative batch stride");
    CTN_REQUIRE((int64_t)nb1 * nb2 <= 65535 && ctn_cdiv(M, 32) <= 65535, "ctn_permute: more than 65535 batches or row tiles");
    dim3 grid(ctn_cdiv(K, 32), ctn_cdiv(M, 32), nb1 * nb2);
    hipLaunchKernelGGL(permute_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, addend, M, K, lds, ldd, nb1, sb1,
                       db1, sb2, db2);
    CTN_CHECK_LAUNCH("ctn_permute");
    return CTN_OK;
}
