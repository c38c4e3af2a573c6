// Sustained fp32 / bf16 MFMA rate and in-kernel clock on this device (diagnostic, not part of the library).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, unsigned long long* clk, int iters) {
    const int lane = threadIdx.x;
    float a = in[lane], b = in[lane + 256];
    f32x16 acc[5];
    for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void mfma_loop_bf16(const float* in, float* out, unsigned long long* clk, int iters) {
    const int lane = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[(lane + e) & 255]; b[e] = (__bf16)in[256 + ((lane + 3 * e) & 255)]; }
    f32x16 acc[5];
    for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 20000;
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 4096); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
    hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
        double fl = (double)blocks * 4 * iters * 40 * 4096.0;
        printf("blocks %d: %.2f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n", blocks, ms, fl / ms / 1e9,
               (double)hc[0] / (double)hc[1] * 100.0);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop_bf16, dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
        double fl = (double)blocks * 4 * iters * 40 * 32768.0;
        printf("bf16 blocks %d: %.2f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n", blocks, ms, fl / ms / 1e9,
               (double)hc[0] / (double)hc[1] * 100.0);
    }
    return 0;
}
