// Depthwise Conv1d (groups = channels) with arbitrary stride, dilation and zero padding, forward and backward, gfx950:
// the depthwise half of the generic DepthwiseSeparableConv1d (src/modules/conv.py:22, default stride = kernel_size).
// The TCN's own depthwise layers (stride 1, 'same' length, fused with PReLU and the norm statistics) are in
// ctn_dwconv.hip; this file covers every other geometry.  HBM-bound and small: one output (or input-gradient) element
// per thread, unconditional loads from clamped addresses masked by selects, consecutive lanes on consecutive frames.
#include "ctn_common.h"

namespace {

constexpr int NTH = 256;

__global__ __launch_bounds__(NTH) void dwg_fwd_kernel(const float* __restrict__ x, int64_t xbs, int64_t ldx, int Tin,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      int K, int stride, int dilation, int pad_left,
                                                      float* __restrict__ y, int64_t ybs, int64_t ldy, int Tout) {
    const int t = blockIdx.x * NTH + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= Tout) return;
    const float* xr = x + (int64_t)b * xbs + (int64_t)c * ldx;
    float acc = bias ? bias[c] : 0.f;
    const int base = t * stride - pad_left;
    for (int k = 0; k < K; ++k) {
        const int i = base + k * dilation;
        const float v = xr[min(max(i, 0), Tin - 1)];
        acc = __builtin_fmaf(w[c * K + k], (i >= 0 && i < Tin) ? v : 0.f, acc);
    }
    y[(int64_t)b * ybs + (int64_t)c * ldy + t] = acc;
}

// dx[i] = sum_k w[k] * dy[(i + pad_left - k*dilation) / stride] over the taps for which the division is exact and the
// frame exists
__global__ __launch_bounds__(NTH) void dwg_bwd_dx_kernel(const float* __restrict__ dy, int64_t dybs, int64_t lddy,
                                                         int Tout, const float* __restrict__ w, int K, int stride,
                                                         int dilation, int pad_left, float* __restrict__ dx,
                                                         int64_t dxbs, int64_t lddx, int Tin) {
    const int i = blockIdx.x * NTH + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (i >= Tin) return;
    const float* dyr = dy + (int64_t)b * dybs + (int64_t)c * lddy;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int num = i + pad_left - k * dilation;
        const int t = num / stride;
        const bool ok = num >= 0 && t * stride == num && t < Tout;
        const float v = dyr[min(max(t, 0), Tout - 1)];
        acc = __builtin_fmaf(w[c * K + k], ok ? v : 0.f, acc);
    }
    dx[(int64_t)b * dxbs + (int64_t)c * lddx + i] = acc;
}

// dW[c][k] += sum_t dy[t] * x[t*stride + k*dilation - pad_left],  dbias[c] += sum_t dy[t]; one block per (c, b)
__global__ __launch_bounds__(NTH) void dwg_bwd_dw_kernel(const float* __restrict__ x, int64_t xbs, int64_t ldx, int Tin,
                                                         const float* __restrict__ dy, int64_t dybs, int64_t lddy,
                                                         int Tout, int K, int stride, int dilation, int pad_left,
                                                         float* dW, float* dbias) {
    __shared__ float red[NTH / 64];
    const int c = blockIdx.x, b = blockIdx.y;
    const float* xr = x + (int64_t)b * xbs + (int64_t)c * ldx;
    const float* dyr = dy + (int64_t)b * dybs + (int64_t)c * lddy;
    for (int k = -1; k < K; ++k) {                        // k = -1: the bias gradient
        if (k < 0 && !dbias) continue;
        float v[1] = {0.f};
        for (int t = threadIdx.x; t < Tout; t += NTH) {
            float xv = 1.f;
            if (k >= 0) {
                const int i = t * stride + k * dilation - pad_left;
                xv = xr[min(max(i, 0), Tin - 1)];
                xv = (i >= 0 && i < Tin) ? xv : 0.f;
            }
            v[0] = __builtin_fmaf(dyr[t], xv, v[0]);
        }
        const float tot = ctn_block_sum<1>(v, red);
        if (threadIdx.x == 0) atomicAdd(k < 0 ? dbias + c : dW + c * K + k, tot);
    }
}

bool rows_ok(const float* p, int64_t bs, int64_t ld, int C, int T) { return p && ld >= T && bs >= (int64_t)(C - 1) * ld + T; }

}  // namespace

extern "C" {

int ctn_dwconv_generic_fwd(const float* x, int64_t xbs, int64_t ldx, int batch, int C, int Tin, const float* w,
                           const float* bias, int K, int stride, int dilation, int pad_left, float* y, int64_t ybs,
                           int64_t ldy, int Tout, void* stream) {
    CTN_REQUIRE(batch > 0 && C > 0 && Tin > 0 && Tout > 0 && w, "ctn_dwconv_generic_fwd: empty problem");
    CTN_REQUIRE(K >= 1 && K <= 1024 && stride >= 1 && dilation >= 1 && pad_left >= 0, "ctn_dwconv_generic_fwd: geometry");
    CTN_REQUIRE(C <= 65535 && batch <= 65535, "ctn_dwconv_generic_fwd: at most 65535 channels and batch entries");
    CTN_REQUIRE(rows_ok(x, xbs, ldx, C, Tin) && rows_ok(y, ybs, ldy, C, Tout), "ctn_dwconv_generic_fwd: layout");
    // any right padding is implied by Tout: taps beyond the input read zeros
    hipLaunchKernelGGL(dwg_fwd_kernel, dim3(ctn_cdiv(Tout, NTH), C, batch), dim3(NTH), 0, (hipStream_t)stream, x, xbs,
                       ldx, Tin, w, bias, K, stride, dilation, pad_left, y, ybs, ldy, Tout);
    CTN_CHECK_LAUNCH("ctn_dwconv_generic_fwd");
    return CTN_OK;
}

int ctn_dwconv_generic_bwd(const float* x, int64_t xbs, int64_t ldx, int batch, int C, int Tin, const float* w, int K,
                           int stride, int dilation, int pad_left, const float* dy, int64_t dybs, int64_t lddy, int Tout,
                           float* dx, int64_t dxbs, int64_t lddx, float* dW, float* dbias, void* stream) {
    CTN_REQUIRE(batch > 0 && C > 0 && Tin > 0 && Tout > 0 && w, "ctn_dwconv_generic_bwd: empty problem");
    CTN_REQUIRE(K >= 1 && K <= 1024 && stride >= 1 && dilation >= 1 && pad_left >= 0, "ctn_dwconv_generic_bwd: geometry");
    CTN_REQUIRE(C <= 65535 && batch <= 65535, "ctn_dwconv_generic_bwd: at most 65535 channels and batch entries");
    CTN_REQUIRE(rows_ok(dy, dybs, lddy, C, Tout), "ctn_dwconv_generic_bwd: dy layout");
    CTN_REQUIRE(!dx || rows_ok(dx, dxbs, lddx, C, Tin), "ctn_dwconv_generic_bwd: dx layout");
    CTN_REQUIRE(!dW || rows_ok(x, xbs, ldx, C, Tin), "ctn_dwconv_generic_bwd: x layout");
    CTN_REQUIRE(dx || dW || dbias, "ctn_dwconv_generic_bwd: nothing to compute");
    hipStream_t s = (hipStream_t)stream;
    if (dx)
        hipLaunchKernelGGL(dwg_bwd_dx_kernel, dim3(ctn_cdiv(Tin, NTH), C, batch), dim3(NTH), 0, s, dy, dybs, lddy, Tout, w,
                           K, stride, dilation, pad_left, dx, dxbs, lddx, Tin);
    if (dW || dbias)
        hipLaunchKernelGGL(dwg_bwd_dw_kernel, dim3(C, batch), dim3(NTH), 0, s, x, xbs, ldx, Tin, dy, dybs, lddy, Tout,
                           dW ? K : 0, stride, dilation, pad_left, dW, dbias);
    CTN_CHECK_LAUNCH("ctn_dwconv_generic_bwd");
    return CTN_OK;
}

}  // extern "C"
