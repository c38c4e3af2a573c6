// Diagnostic: which operand change slows v_mfma_f32_16x16x4_f32 below its 32-cycle issue rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VA, int VB, int NACC>   // VA/VB: operand A / B rotates over 16 registers (1) or stays fixed (0)
__global__ __launch_bounds__(256) void probe(const float* in, float* out, unsigned long long* clk, int iters) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(threadIdx.x * 16 + i + 2048) & 4095]; }
    f32x4 acs[NACC];
    for (int i = 0; i < NACC; ++i) acs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64 / NACC; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acs[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[VA ? (k + i) & 15 : 0], b[VB ? (5 * k + i + 3) & 15 : 0], acs[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acs[i][r];
    for (int i = 0; i < 16; ++i) s += a[i] + b[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int VA, int VB, int NACC>
void run(const float* in, float* out, unsigned long long* clk, int blocks, int iters) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<VA, VB, NACC>), dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
        hipDeviceSynchronize();
    }
    unsigned long long hc; hipMemcpy(&hc, clk, 8, hipMemcpyDeviceToHost);
    printf("A %s  B %s  %2d accumulators: %.1f cycles per MFMA\n", VA ? "rotates" : "fixed  ", VB ? "rotates" : "fixed  ", NACC,
           (double)hc / iters / 64.0);
}

int main() {
    const int blocks = 256, iters = 20000;
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 8);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
    hipMemcpy(in, h, 4096 * 4, hipMemcpyHostToDevice);
    run<0, 0, 16>(in, out, clk, blocks, iters);
    run<1, 0, 16>(in, out, clk, blocks, iters);
    run<0, 1, 16>(in, out, clk, blocks, iters);
    run<1, 1, 16>(in, out, clk, blocks, iters);
    run<1, 1, 4>(in, out, clk, blocks, iters);
    run<1, 1, 8>(in, out, clk, blocks, iters);
    run<0, 0, 4>(in, out, clk, blocks, iters);
    run<0, 0, 2>(in, out, clk, blocks, iters);
    return 0;
}
