// Chunking of a frame sequence for dual-path models and its inverse (overlap-add).
//
//   ctn_segment:     y[b][c][s][k] = xpad[b][c][s*P + k],  xpad = x zero-padded by pad_left frames in front (and by
//                    whatever the S chunks need behind)            -- Segment1d.forward, src/models/transform.py:15-31,
//                    with the F.pad of src/models/dprnn_tasnet.py:339-341 folded in
//   ctn_overlap_add: x[b][c][t] = sum_s y[b][c][s][t + pad_left - s*P]   (0 <= t + pad_left - s*P < K)
//                    -- OverlapAdd1d.forward, transform.py:45-61, with the trim of dprnn_tasnet.py:344 folded in
//
// Each is the other's adjoint, so the pair also serves as each other's backward.  Both are gathers (no atomics): a
// thread owns output elements, consecutive lanes own consecutive k (or t), so every load and store of a wave covers
// one contiguous run of floats.  HBM-bound: segment reads T and writes S*K floats per row, overlap-add the reverse.
#include "ctn_common.h"

namespace {

constexpr int NTH = 256;

__global__ __launch_bounds__(NTH) void segment_kernel(const float* __restrict__ x, const int64_t xbs, const int64_t ldx,
                                                      const int C, const int T, const int pad_left, const int K, const int P,
                                                      const int S, float* __restrict__ y) {
    const int c = blockIdx.y, b = blockIdx.z;
    const float* xr = x + (int64_t)b * xbs + (int64_t)c * ldx;
    float* yr = y + ((int64_t)b * C + c) * (int64_t)S * K;
    const int n = S * K;
    for (int i = blockIdx.x * NTH + threadIdx.x; i < n; i += gridDim.x * NTH) {
        const int s = i / K, k = i - s * K;
        const int t = s * P + k - pad_left;
        yr[i] = (t >= 0 && t < T) ? xr[t] : 0.f;
    }
}

__global__ __launch_bounds__(NTH) void overlap_add_kernel(const float* __restrict__ y, const int C, const int S, const int K,
                                                          const int P, const int pad_left, const int T,
                                                          float* __restrict__ x, const int64_t xbs, const int64_t ldx) {
    const int c = blockIdx.y, b = blockIdx.z;
    const float* yr = y + ((int64_t)b * C + c) * (int64_t)S * K;
    float* xr = x + (int64_t)b * xbs + (int64_t)c * ldx;
    for (int t = blockIdx.x * NTH + threadIdx.x; t < T; t += gridDim.x * NTH) {
        const int u = t + pad_left;                      // position in the padded sequence
        // chunks s with 0 <= u - s*P < K
        const int s_hi = min(u / P, S - 1);
        const int s_lo = max((u - K + P) / P, 0);        // ceil((u - K + 1) / P) for u - K + 1 > 0
        float acc = 0.f;
        for (int s = s_lo; s <= s_hi; ++s) acc += yr[(int64_t)s * K + (u - s * P)];
        xr[t] = acc;
    }
}

}  // namespace

extern "C" int ctn_segment(const float* x, int64_t xbs, int64_t ldx, int batch, int C, int T, int pad_left, int K, int P,
                           int S, float* y, void* stream) {
    CTN_REQUIRE(x && y, "ctn_segment: null pointer");
    CTN_REQUIRE(batch > 0 && C > 0 && T > 0 && K > 0 && P > 0 && S > 0 && pad_left >= 0, "ctn_segment: bad sizes");
    CTN_REQUIRE(ldx >= T && xbs >= (int64_t)C * ldx, "ctn_segment: strides smaller than the data");
    CTN_REQUIRE((int64_t)S * K < (1ll << 31) && C <= 65535 && batch <= 65535, "ctn_segment: problem too large");
    const int gx = min(ctn_cdiv((int64_t)S * K, NTH), 1024);
    hipLaunchKernelGGL(segment_kernel, dim3(gx, C, batch), dim3(NTH), 0, (hipStream_t)stream, x, xbs, ldx, C, T, pad_left, K,
                       P, S, y);
    CTN_CHECK_LAUNCH("ctn_segment");
    return CTN_OK;
}

extern "C" int ctn_overlap_add(const float* y, int batch, int C, int S, int K, int P, int pad_left, int T, float* x,
                               int64_t xbs, int64_t ldx, void* stream) {
    CTN_REQUIRE(x && y, "ctn_overlap_add: null pointer");
    CTN_REQUIRE(batch > 0 && C > 0 && T > 0 && K > 0 && P > 0 && S > 0 && pad_left >= 0, "ctn_overlap_add: bad sizes");
    CTN_REQUIRE(ldx >= T && xbs >= (int64_t)C * ldx, "ctn_overlap_add: strides smaller than the data");
    CTN_REQUIRE((int64_t)S * K < (1ll << 31) && C <= 65535 && batch <= 65535, "ctn_overlap_add: problem too large");
    const int gx = min(ctn_cdiv(T, NTH), 1024);
    hipLaunchKernelGGL(overlap_add_kernel, dim3(gx, C, batch), dim3(NTH), 0, (hipStream_t)stream, y, C, S, K, P, pad_left, T,
                       x, xbs, ldx);
    CTN_CHECK_LAUNCH("ctn_overlap_add");
    return CTN_OK;
}
