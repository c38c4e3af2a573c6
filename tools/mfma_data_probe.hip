// Diagnostic: does the fp32 MFMA rate / clock depend on the SHAPE when the operands change every instruction
// (random data), as they do in a real kernel?  Each MFMA takes its A and B from a rotating set of 16 random registers.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_data_probe.hip -o tools/mfma_data_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 32x32x2, 1: 16x16x4
__global__ __launch_bounds__(256) void probe(const float* in, float* out, unsigned long long* clk, int iters) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + i + 2048) & 4095]; }
    f32x16 acc[4];
    f32x4 acs[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) acs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(4 * k + i) & 15], b[(4 * k + i + 5) & 15], acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acs[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + i) & 15], b[(5 * k + i + 3) & 15], acs[i], 0, 0, 0);
        }
        // keep magnitudes bounded (rarely taken)
        if ((it & 1023) == 1023) {
            for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;
            for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acs[i][r] *= 1e-3f;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acs[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 40000;
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        if (zero) for (int i = 0; i < 4096; ++i) h[i] = 0.f;
        hipMemcpy(in, h, 4096 * 4, hipMemcpyHostToDevice);
        for (int shape = 0; shape < 2; ++shape)
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL((probe<0>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
                else hipLaunchKernelGGL((probe<1>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
                const double fl = (double)blocks * 4 * iters * (shape == 0 ? 32 * 4096.0 : 64 * 2048.0);
                if (rep == 2)
                    printf("%s data, %s: %8.2f ms  %6.1f TFLOP/s  clock %4.0f MHz  cycles/iter %.0f (2048 = issue-bound)\n",
                           zero ? "zero  " : "random", shape == 0 ? "32x32x2" : "16x16x4", ms, fl / ms / 1e9,
                           (double)hc[0] / (double)hc[1] * 100.0, (double)hc[0] / iters);
            }
    }
    return 0;
}
