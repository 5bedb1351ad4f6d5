// Read throughput of [batch][rows][pitch] fp32 activations when every block of a full-chip launch reads, per stage,
// SEG contiguous bytes from each of 32768 / SEG rows (16 KB apart), stages dealt round-robin over the blocks: the access
// pattern of a weight-gradient kernel that walks the time axis of time-major operands.  usage: rowseg_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// rows R = 512 per batch element, pitch P floats; a stage = SEG bytes of every row of one row group of 32768 / SEG rows;
// thread t of 512: 4 loads of 16 bytes per stage: piece i = t + 512 q: row i / (SEG / 16), byte 16 (i % (SEG / 16))
template <int SEG>
__global__ __launch_bounds__(512) void probe(const float* x, int batch, int rows, int pitch, int T, unsigned* sink) {
    constexpr int PPR = SEG / 16;                 // 16-byte pieces per row segment
    constexpr int RPS = 2048 / PPR;               // rows per stage
    const int tid = threadIdx.x;
    const int segs = T * 4 / SEG;                 // whole segments per row
    const int groups = rows / RPS;
    const int64_t nstages = (int64_t)batch * groups * segs;
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t G = blockIdx.x; G < nstages; G += gridDim.x) {
        const int b = (int)(G / ((int64_t)groups * segs));
        const int r = (int)(G - (int64_t)b * groups * segs);
        const int grp = r / segs, sg = r - grp * segs;          // consecutive blocks: consecutive segments of the same rows
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)b * rows * pitch), 0, 0x7ffffffc, 0x00020000);
        u32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 512 * q;
            const int row = grp * RPS + i / PPR;
            v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, row * pitch * 4 + (i % PPR) * 16, sg * SEG, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc ^= v[q];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int SEG>
int run(const float* x, int batch, int rows, int pitch, int T, unsigned* sink, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<SEG>, dim3(blocks), dim3(512), 0, 0, x, batch, rows, pitch, T, sink);
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<SEG>, dim3(blocks), dim3(512), 0, 0, x, batch, rows, pitch, T, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)batch * rows * (T * 4 / SEG) * SEG;
    printf("segment %4d B, %d blocks: %.1f us, %.2f TB/s\n", SEG, blocks, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    return 0;
}

int main() {
    const int batch = 32, rows = 512, pitch = 4000, T = 3968;     // T a multiple of 256 frames: whole segments at every width
    float* x; unsigned* sink;
    CK(hipMalloc(&x, (size_t)batch * rows * pitch * 4));
    CK(hipMemset(x, 0, (size_t)batch * rows * pitch * 4));
    CK(hipMalloc(&sink, 4));
    for (int blocks : {256, 512, 1024}) {
        if (run<64>(x, batch, rows, pitch, T, sink, blocks)) return 1;
        if (run<128>(x, batch, rows, pitch, T, sink, blocks)) return 1;
        if (run<256>(x, batch, rows, pitch, T, sink, blocks)) return 1;
        if (run<512>(x, batch, rows, pitch, T, sink, blocks)) return 1;
        if (run<1024>(x, batch, rows, pitch, T, sink, blocks)) return 1;
    }
    CK(hipDeviceSynchronize());
    return 0;
}
