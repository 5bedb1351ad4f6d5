// Prepared weights: fp32 weight matrices split once into three bf16 pieces (hi + mid + lo = x exactly) and laid out as
// the bf16 matrix instruction takes its A operand -- planes[piece][k / 8][m][8] -- so that the pointwise GEMM fetches
// finished fragments with 16-byte LDS reads instead of splitting the same weights again in every wave of every block.
#include "ctn_common.h"

namespace {

constexpr int WP_ITEMS = 56;          // items per launch (passed by value: no table in device memory)
constexpr int WP_THREADS = 256;

struct wprep_batch {
    ctn_wprep_item item[WP_ITEMS];
    int32_t first_block[WP_ITEMS + 1];
    int32_t n;
};

// One thread per (k-group, m) of an item: reads the 8 source values A[8g .. 8g+7][m], writes three 16-byte fragments.
__global__ __launch_bounds__(WP_THREADS) void wprep_kernel(const wprep_batch tb) {
    int it = 0;
    while (it + 1 < tb.n && (int)blockIdx.x >= tb.first_block[it + 1]) ++it;        // uniform: scalar loop
    const ctn_wprep_item& d = tb.item[it];
    const int kext = d.transpose ? d.cols : d.rows, mext = d.transpose ? d.rows : d.cols;
    const int groups = (kext + 7) >> 3;
    const int64_t o = (int64_t)((int)blockIdx.x - tb.first_block[it]) * WP_THREADS + threadIdx.x;
    if (o >= (int64_t)groups * mext) return;
    const int g = (int)(o / mext), m = (int)(o % mext);
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = 8 * g + q;
        const int kc = min(k, kext - 1);
        const float v = d.transpose ? d.src[(int64_t)m * d.ld + kc] : d.src[(int64_t)kc * d.ld + m];
        x[q] = k < kext ? v : 0.f;
    }
    uint4 hi, mid, lo;
    unsigned* hp = reinterpret_cast<unsigned*>(&hi);
    unsigned* mp = reinterpret_cast<unsigned*>(&mid);
    unsigned* lp = reinterpret_cast<unsigned*>(&lo);
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // the same truncating split as the GEMM kernels apply to their fp32 operands
        const unsigned x0 = __float_as_uint(x[2 * i]), x1 = __float_as_uint(x[2 * i + 1]);
        hp[i] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
        const float r0 = x[2 * i] - __uint_as_float(x0 & 0xffff0000u), r1 = x[2 * i + 1] - __uint_as_float(x1 & 0xffff0000u);
        const unsigned ru0 = __float_as_uint(r0), ru1 = __float_as_uint(r1);
        mp[i] = __builtin_amdgcn_perm(ru1, ru0, 0x07060302u);
        const float q0 = r0 - __uint_as_float(ru0 & 0xffff0000u), q1 = r1 - __uint_as_float(ru1 & 0xffff0000u);
        lp[i] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
    }
    const int KG = ((d.K + 15) >> 4) * 2, Mp = ((d.M + 127) >> 7) * 128;
    uint4* dst = reinterpret_cast<uint4*>(d.dst);
    const int64_t at = (int64_t)((d.k_off >> 3) + g) * Mp + d.m_off + m;
    dst[at] = hi;
    dst[(int64_t)KG * Mp + at] = mid;
    dst[2 * (int64_t)KG * Mp + at] = lo;
}

}  // namespace

extern "C" {

int ctn_sizeof_wprep_item(void) { return (int)sizeof(ctn_wprep_item); }

int64_t ctn_weight_prep_bytes(int K, int M) {
    if (K <= 0 || M <= 0) return 0;
    const int64_t Kp = ((int64_t)K + 15) / 16 * 16, Mp = ((int64_t)M + 127) / 128 * 128;
    return 3 * Kp * Mp * 2;
}

int ctn_weight_prep(const ctn_wprep_item* items, int n_items, void* stream) {
    CTN_REQUIRE(items && n_items > 0, "ctn_weight_prep: no items");
    for (int i = 0; i < n_items; ++i) {
        const ctn_wprep_item& d = items[i];
        CTN_REQUIRE(d.src && d.dst && ctn_aligned16(d.dst), "ctn_weight_prep: item %d: null source or unaligned destination", i);
        CTN_REQUIRE(d.rows > 0 && d.cols > 0 && d.ld >= d.cols, "ctn_weight_prep: item %d: bad source shape", i);
        CTN_REQUIRE(d.transpose == 0 || d.transpose == 1, "ctn_weight_prep: item %d: transpose is 0 or 1", i);
        const int kext = d.transpose ? d.cols : d.rows, mext = d.transpose ? d.rows : d.cols;
        CTN_REQUIRE(d.K > 0 && d.M > 0 && d.k_off >= 0 && d.m_off >= 0 && d.k_off % 8 == 0 &&
                        (int64_t)d.k_off + kext <= d.K && (int64_t)d.m_off + mext <= d.M,
                    "ctn_weight_prep: item %d does not fit its prepared matrix (k_off %% 8 == 0)", i);
    }
    for (int base = 0; base < n_items; base += WP_ITEMS) {
        wprep_batch tb;
        tb.n = n_items - base < WP_ITEMS ? n_items - base : WP_ITEMS;
        int64_t blocks = 0;
        for (int i = 0; i < tb.n; ++i) {
            const ctn_wprep_item& d = items[base + i];
            tb.item[i] = d;
            tb.first_block[i] = (int32_t)blocks;
            const int kext = d.transpose ? d.cols : d.rows, mext = d.transpose ? d.rows : d.cols;
            blocks += ((int64_t)((kext + 7) / 8) * mext + WP_THREADS - 1) / WP_THREADS;
            CTN_REQUIRE(blocks < 0x7fffffff, "ctn_weight_prep: too much work for one launch");
        }
        for (int i = tb.n; i <= WP_ITEMS; ++i) tb.first_block[i] = (int32_t)blocks;
        hipLaunchKernelGGL(wprep_kernel, dim3((unsigned)blocks), dim3(WP_THREADS), 0, (hipStream_t)stream, tb);
        CTN_CHECK_LAUNCH("ctn_weight_prep");
    }
    return CTN_OK;
}

}  // extern "C"
