// Sweep of plain copy kernels on 1 GiB buffers: which access pattern reaches the highest HBM rate on this GPU.
//   hipcc -O3 --offload-arch=gfx950 -o copy_sweep copy_sweep.hip && ./copy_sweep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_gs(const f32x4* __restrict__ a, f32x4* __restrict__ y, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * stride); else y[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) y[i] = a[i];
}

// each block owns one contiguous chunk
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_chunk(const f32x4* __restrict__ a, f32x4* __restrict__ y, int64_t n4) {
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    int64_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * 256) : a[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], y + i + u * 256); else y[i + u * 256] = v[u]; }
    }
    for (; i < hi; i += 256) y[i] = a[i];
}

__global__ __launch_bounds__(256) void read_only(const f32x4* __restrict__ a, float* out, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    f32x4 s = {};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) s += __builtin_nontemporal_load(a + i);
    if (s[0] + s[1] + s[2] + s[3] == 1.2345f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void write_only(f32x4* y, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(v, y + i);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int64_t bytes = 1ll << 30, n4 = bytes / 16;
    f32x4 *a, *y; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(y, 0, bytes));
    const int grids[] = {512, 1024, 2048, 4096, 8192, 16384, 65536};
    for (int g : grids) {
#define RUN(name, K) { float ms = timeit([&] { hipLaunchKernelGGL(K, dim3(g), dim3(256), 0, 0, a, y, n4); }, 5); \
        printf("%-22s grid %6d  %7.1f GB/s\n", name, g, 2.0 * bytes / ms / 1e6); }
        RUN("gs U1 nt", (copy_gs<1, true>)) RUN("gs U2 nt", (copy_gs<2, true>)) RUN("gs U4 nt", (copy_gs<4, true>))
        RUN("gs U8 nt", (copy_gs<8, true>)) RUN("gs U4 plain", (copy_gs<4, false>)) RUN("gs U1 plain", (copy_gs<1, false>))
        RUN("chunk U4 nt", (copy_chunk<4, true>)) RUN("chunk U4 plain", (copy_chunk<4, false>)) RUN("chunk U8 nt", (copy_chunk<8, true>))
        fflush(stdout);
    }
    {
        float ms = timeit([&] { hipLaunchKernelGGL(read_only, dim3(4096), dim3(256), 0, 0, a, out, n4); }, 5);
        printf("read only   %7.1f GB/s\n", 1.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(write_only, dim3(4096), dim3(256), 0, 0, y, n4); }, 5);
        printf("write only  %7.1f GB/s\n", 1.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipMemcpyAsync(y, a, bytes, hipMemcpyDeviceToDevice, 0); }, 5);
        printf("hipMemcpy D2D %7.1f GB/s\n", 2.0 * bytes / ms / 1e6);
    }
    // smaller footprints (what a 32-utterance depthwise launch touches: 2 x 262 MB)
    for (int64_t mb : {64, 128, 256, 512}) {
        const int64_t m4 = mb * (1 << 20) / 16;
        float ms = timeit([&] { hipLaunchKernelGGL((copy_gs<4, true>), dim3(4096), dim3(256), 0, 0, a, y, m4); }, 10);
        printf("gs U4 nt footprint 2x%lld MB: %7.1f GB/s\n", (long long)mb, 2.0 * mb * (1 << 20) / ms / 1e6);
    }
    return 0;
}
