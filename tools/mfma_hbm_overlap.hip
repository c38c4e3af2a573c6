// Do the fp32 matrix cores and HBM streaming overlap on this device, or do their times add?  (diagnostic, not part of
// the library)  One block of 8 waves per CU: waves 0-3 (one per SIMD) run register-only
// v_mfma_f32_32x32x2_f32, waves 4-7 copy a large buffer (16-byte loads and stores, grid-strided).  Three launches:
// MFMA waves only, copy waves only, both - the same work per role each time.  argv[3] = 1: the copy waves raise their
// priority (s_setprio 3).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_hbm_overlap.hip -o tools/mfma_hbm_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int YIELD>       // independent accumulators in the MFMA stream; YIELD: 1 = s_nop 1 after each MFMA, 2 = s_sleep 1 per 8 MFMAs
__global__ __launch_bounds__(512) void both(const float* in, float* out, const f32x4* src, f32x4* dst, size_t n4,
                                            int iters, int do_mfma, int do_copy, unsigned long long* clk, int prio) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!do_mfma) return;
        float a = in[lane], b = in[lane + 64];
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 40 / NACC; ++k) {
#pragma unroll
                for (int i = 0; i < NACC; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    if (YIELD == 1) { asm volatile("s_nop 1"); }
                    if (YIELD == 3 && (i & 1)) { asm volatile("s_nop 1"); }
                    if (YIELD == 4 && i == NACC - 1) { asm volatile("s_nop 1"); }
                    if (YIELD == 5) { asm volatile("s_nop 0"); }
                    if (YIELD == 6) { asm volatile("s_nop 4"); }
                }
                if (YIELD == 2 && (k * NACC) % 8 == 0) __builtin_amdgcn_s_sleep(1);
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float s = 0.f;
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[(blockIdx.x * 4 + wave) * 64 + lane] = s;
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
    } else {
        if (!do_copy) return;
        if (prio) __builtin_amdgcn_s_setprio(3);            // the arbiter prefers this wave whenever it has an instruction ready
        const size_t stride = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + (threadIdx.x - 256); i < n4; i += 4 * stride) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + u * stride < n4) v[u] = src[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + u * stride < n4) dst[i + u * stride] = v[u];
        }
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;
    const size_t gb = argc > 2 ? atoi(argv[2]) : 8;            // GB read (and as many written) by the copy
    const int blocks = 256;
    const int prio = argc > 3 ? atoi(argv[3]) : 0;
    const int variant = argc > 4 ? atoi(argv[4]) : 0;       // 0: 5 accumulators; 1: one (a dependent chain); 2: 5 + s_nop 1 after every MFMA; 3: 5 + s_sleep 1 per 8; 4: s_nop 1 after every 2nd; 5: after every 5th; 6: s_nop 0 after each; 7: s_nop 4 after each
    float *in, *out; unsigned long long* clk; f32x4 *src, *dst;
    const size_t n4 = gb * (1ull << 30) / 16;
    if (hipMalloc(&in, 4096) || hipMalloc(&out, blocks * 4 * 64 * 4) || hipMalloc(&clk, blocks * 16) ||
        hipMalloc(&src, n4 * 16) || hipMalloc(&dst, n4 * 16)) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 0, n4 * 16);
    float h[128]; for (int i = 0; i < 128; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
    hipMemcpy(in, h, 512, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = 2.0 * 32 * 32 * 2 * 40.0 * iters * blocks * 4;
    const char* names[3] = {"mfma only", "copy only", "both     "};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            if (rep == 0 && mode < 2) continue;
            const int dm = mode != 1, dc = mode != 0;
            hipEventRecord(e0);
            if (variant == 1) hipLaunchKernelGGL((both<1, 0>), dim3(blocks), dim3(512), 0, 0, in, out, src, dst, n4, iters, dm, dc, clk, prio);
            else if (variant == 2) hipLaunchKernelGGL((both<5, 1>), dim3(blocks), dim3(512), 0, 0, in, out, src, dst, n4, iters, dm, dc, clk, prio);
            else if (variant == 3) hipLaunchKernelGGL((both<5, 2>), dim3(blocks), dim3(512), 0, 0, in, out, src, dst, n4, iters, dm, dc, clk, prio);
            else if (variant == 4) hipLaunchKernelGGL((both<5, 3>), dim3(blocks), dim3(512), 0, 0, in, out, src, dst, n4, iters, dm, dc, clk, prio);
            else if (variant == 5) hipLaunchKernelGGL((both<5, 4>), dim3(blocks), dim3(512), 0, 0, in, out, src, dst, n4, iters, dm, dc, clk, prio);
            else if (variant == 6) hipLaunchKernelGGL((both<5, 5>), dim3(blocks), dim3(512), 0, 0, in, out, src, dst, n4, iters, dm, dc, clk, prio);
            else if (variant == 7) hipLaunchKernelGGL((both<5, 6>), dim3(blocks), dim3(512), 0, 0, in, out, src, dst, n4, iters, dm, dc, clk, prio);
            else hipLaunchKernelGGL((both<5, 0>), dim3(blocks), dim3(512), 0, 0, in, out, src, dst, n4, iters, dm, dc, clk, prio);
            hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[2] = {0, 0};
            if (dm) hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
            // s_memtime counts at the shader clock, s_memrealtime at 100 MHz
            const double ghz = dm && hc[1] ? (double)hc[0] / ((double)hc[1] * 10.0) : 0.0;
            const double mfma_ms = dm && hc[1] ? hc[1] / 1e5 : 0.0;
            printf("%s  kernel %.3f ms", names[mode], ms);
            if (dm) printf("  | mfma waves %.3f ms = %.1f TFLOP/s, in-kernel clock %.2f GHz", mfma_ms, flops / mfma_ms / 1e9, ghz);
            if (dc) printf("  | copy %.2f TB/s (read + write) over the kernel", 2.0 * n4 * 16 / ms / 1e9);
            printf("\n");
        }
    return 0;
}
