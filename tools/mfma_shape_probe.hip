// Diagnostic (not part of the library): fp32 MFMA shapes and VALU co-issue on gfx950.
//   mode 0: v_mfma_f32_32x32x2_f32 only          mode 1: v_mfma_f32_16x16x4_f32 only
//   mode 2: 32x32x2 + NV independent v_fma per MFMA in the SAME wave
//   mode 3: 16x16x4 + NV independent v_fma per MFMA in the SAME wave
//   mode 4: 512-thread blocks: waves 0-3 run mode 0, waves 4-7 (their SIMD partners) run v_fma only
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ __launch_bounds__(512) void probe(const float* in, float* out, unsigned long long* clk, int iters) {
    const int lane = threadIdx.x & 255;
    float a = in[lane], b = in[lane + 256];
    f32x16 acc[4];
    f32x4 acs[16];
    float v[8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) acs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) v[i] = in[(lane + i) & 255];
    const bool valu_wave = MODE == 4 && threadIdx.x >= 256;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (valu_wave) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < 64; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (MODE == 0 || MODE == 2 || MODE == 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < (MODE == 2 ? NV : 0); ++j) v[j & 7] = __builtin_fmaf(v[j & 7], a, b);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acs[2 * (i & 7) + (k & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acs[2 * (i & 7) + (k & 1)], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < (MODE == 3 ? NV : 0); ++j) v[j & 7] = __builtin_fmaf(v[j & 7], a, b);
                    }
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acs[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 255) == 0) { clk[4 * blockIdx.x + 2 * (threadIdx.x >> 8)] = t1 - t0; clk[4 * blockIdx.x + 2 * (threadIdx.x >> 8) + 1] = r1 - r0; }
}

template <int MODE, int NV>
void run(const char* name, int threads, const float* in, float* out, unsigned long long* clk, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, NV>), dim3(blocks), dim3(threads), 0, 0, in, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[4]; hipMemcpy(hc, clk, 32, hipMemcpyDeviceToHost);
        // flops: 32 MFMAs of 4096 flop (32x32x2) or 64 MFMAs of 2048 flop (16x16x4) per iteration per MFMA wave
        double fl = (double)blocks * 4 * iters * 32 * 4096.0;
        if (rep == 1)
            printf("%-44s %8.2f ms  %6.1f TFLOP/s (MFMA)  clock %4.0f MHz  cycles/iter %.0f%s\n", name, ms, fl / ms / 1e9,
                   (double)hc[0] / (double)hc[1] * 100.0, (double)hc[0] / iters,
                   threads == 512 ? "" : "");
        if (rep == 1 && threads == 512)
            printf("%-44s   VALU partner wave: cycles/iter %.0f (512 v_fma per iter)\n", "", (double)hc[2] / iters);
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 20000;
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 4096); hipMalloc(&out, blocks * 512 * 4); hipMalloc(&clk, blocks * 32);
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
    hipMemcpy(in, h, 2048, hipMemcpyHostToDevice);
    run<0, 0>("32x32x2 only", 256, in, out, clk, blocks, iters);
    run<1, 0>("16x16x4 only", 256, in, out, clk, blocks, iters);
    run<2, 2>("32x32x2 + 2 v_fma per MFMA (same wave)", 256, in, out, clk, blocks, iters);
    run<2, 8>("32x32x2 + 8 v_fma per MFMA (same wave)", 256, in, out, clk, blocks, iters);
    run<3, 2>("16x16x4 + 2 v_fma per MFMA (same wave)", 256, in, out, clk, blocks, iters);
    run<3, 4>("16x16x4 + 4 v_fma per MFMA (same wave)", 256, in, out, clk, blocks, iters);
    run<4, 0>("32x32x2 waves + VALU-only partner waves", 512, in, out, clk, blocks, iters);
    run<1, 0>("16x16x4 only, TWO waves per SIMD (2x work)", 512, in, out, clk, blocks, iters);
    run<0, 0>("32x32x2 only, TWO waves per SIMD (2x work)", 512, in, out, clk, blocks, iters);
    return 0;
}
