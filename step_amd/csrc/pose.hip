// Input pipeline: root-relative similarity alignment of a batch of gaits, fused with the loader's
// layout change.
//
// Replaces the per-sample / per-frame Python loops of
// classifier_stgcn_real_only/utils/common.py:105-112 (first 48 features of get_affective_features:
// (c R) (x[s,t,j] - x[s,t,root]) + t with the per-gait transform of get_transformation, :230-279)
// and the reshape + moveaxis of utils/loader.py:65-66 that turns (N, T, V*C) into the model's
// (N, C, T, V, 1).  HBM-bound: reads each coordinate once (12-byte joint records), writes it once
// in NCTV order (64-byte rows).  The 3x3 SVD per gait stays on the host (step_amd/pipeline.py),
// as in the reference; its result arrives as M[n] = [cR | t] (3 x 4, row-major).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void pose_align_kernel(const float* __restrict__ gaits, const float* __restrict__ M,
                                                         float* __restrict__ out, int N, int T, int V) {
    // one thread per (n, t, v)
    const long long total = (long long)N * T * V;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(e % V);
        const long long nt = e / V;
        const int t = (int)(nt % T);
        const int n = (int)(nt / T);
        const float* fr = gaits + nt * V * 3;                 // frame (n, t): V joints x 3
        const float x = fr[v * 3] - fr[0], y = fr[v * 3 + 1] - fr[1], z = fr[v * 3 + 2] - fr[2];
        const float* m = M + (size_t)n * 12;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            out[(((size_t)n * 3 + c) * T + t) * V + v] = m[c * 4] * x + m[c * 4 + 1] * y + m[c * 4 + 2] * z + m[c * 4 + 3];
    }
}

// Ragged form: gait n is frames [off[n], off[n+1]) of `frames`; output frame t of gait n is source frame
// (t mod time_steps) mod len_n - the np.tile to the longest gait and the `cycles` repeats of
// utils/loader.py:33-36, never materialised.
__global__ __launch_bounds__(256) void pose_align_ragged_kernel(const float* __restrict__ frames,
                                                                const int* __restrict__ off,
                                                                const float* __restrict__ M, float* __restrict__ out,
                                                                int N, int time_steps, int T, int V) {
    const long long total = (long long)N * T * V;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(e % V);
        const long long nt = e / V;
        const int t = (int)(nt % T);
        const int n = (int)(nt / T);
        const int o = off[n], len = off[n + 1] - o;
        const float* fr = frames + ((size_t)o + (size_t)((t % time_steps) % len)) * V * 3;
        const float x = fr[v * 3] - fr[0], y = fr[v * 3 + 1] - fr[1], z = fr[v * 3 + 2] - fr[2];
        const float* m = M + (size_t)n * 12;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            out[(((size_t)n * 3 + c) * T + t) * V + v] = m[c * 4] * x + m[c * 4 + 1] * y + m[c * 4 + 2] * z + m[c * 4 + 3];
    }
}

}  // namespace

extern "C" int step_pose_align_ragged(const float* frames, const int* offsets, const float* M, float* out, int N,
                                      int time_steps, int cycles, int V, void* stream) {
    STEP_REQUIRE(frames && offsets && M && out, "step_pose_align_ragged: null pointer");
    STEP_REQUIRE(N >= 1 && time_steps >= 1 && cycles >= 1 && V >= 1,
                 "step_pose_align_ragged: bad shape N=%d time_steps=%d cycles=%d V=%d", N, time_steps, cycles, V);
    const int T = time_steps * cycles;
    const long long total = (long long)N * T * V;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pose_align_ragged_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, frames,
                       offsets, M, out, N, time_steps, T, V);
    return step::check_launch("step_pose_align_ragged");
}

extern "C" int step_pose_align(const float* gaits, const float* M, float* out, int N, int T, int V, void* stream) {
    STEP_REQUIRE(gaits && M && out, "step_pose_align: null pointer");
    STEP_REQUIRE(N >= 1 && T >= 1 && V >= 1, "step_pose_align: bad shape N=%d T=%d V=%d", N, T, V);
    const long long total = (long long)N * T * V;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pose_align_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gaits, M, out, N,
                       T, V);
    return step::check_launch("step_pose_align");
}
