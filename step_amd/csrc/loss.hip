// CVAE training loss, value and gradients in one pass.
//
// Replaces generator_cvae/utils/losses.py:7-42 `between_frame_loss` - an O(B*T*V*C) Python loop of
// scalar mse_loss calls in the reference - and the KLD term of vae_loss
// (generator_cvae/utils/processor_stgcn.py:34-40), plus what autograd runs for them.
// With D = x_in - x_out, F = C*V, mid = int(T/2) - 1 (negative wraps):
//   L = mean(D^2) + SUM_b SUM_t SUM_{a in {0,mid,T-1}} mean_f (D[t]-D[a])^2
//       + SUM_b SUM_{t>=1} SUM_f (D[t]-D[t-1])^2 + SUM_b SUM_{t>=2} SUM_f (D[t]-2D[t-1]+D[t-2])^2
//   KLD = -0.5 SUM (1 + lsig - mean^2 - exp(lsig))
// One block per sequence: D lives in LDS in the tensor's (C, T, V) order, so every stencil
// neighbour is V floats away; HBM-bound (read x_in, x_out once, write d_x_out once).  Per-sequence
// partial sums are fp64 and are added in sequence order by the finalize block (reproducible).
#include "common.h"

namespace {

constexpr int LOSS_MAX_E = 15360;          // C*T*V floats of one sequence held in LDS (60 KiB)

__global__ __launch_bounds__(256) void vae_loss_seq_kernel(const float* __restrict__ x_in,
                                                           const float* __restrict__ x_out,
                                                           float* __restrict__ d_x_out, double* __restrict__ part,
                                                           int N, int C, int T, int V) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int F = C * V, E = C * T * V, TV = T * V;
    float* D = sm;                 // [C][T][V]
    float* S = sm + E;             // [C][V]  SUM_t D
    __shared__ double red[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)n * E;
    for (int e = tid; e < E; e += 256) D[e] = x_in[base + e] - x_out[base + e];
    __syncthreads();
    for (int f = tid; f < F; f += 256) {
        const int c = f / V, v = f % V;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += D[c * TV + t * V + v];
        S[f] = s;
    }
    __syncthreads();
    int mid = T / 2 - 1;
    if (mid < 0) mid += T;
    const float invF = 1.f / (float)F;
    const float invAll = 1.f / ((float)N * (float)T * (float)F);
    double lsum = 0.0;
    for (int e = tid; e < E; e += 256) {
        const int c = e / TV, r = e - c * TV, t = r / V, v = r - t * V;
        const float* col = D + c * TV + v;                 // + u*V for frame u
        const float d = D[e];
        const float d0 = col[0], dm = col[mid * V], dl = col[(T - 1) * V];
        const float a0 = d - d0, am = d - dm, al = d - dl;
        float l = d * d * invAll + (a0 * a0 + am * am + al * al) * invF;
        float g = 2.f * d * invAll + 2.f * invF * (a0 + am + al);
        const float st = S[c * V + v];
        if (t == 0) g -= 2.f * invF * (st - (float)T * d0);
        if (t == mid) g -= 2.f * invF * (st - (float)T * dm);
        if (t == T - 1) g -= 2.f * invF * (st - (float)T * dl);
        const float dp1 = t >= 1 ? col[(t - 1) * V] : 0.f, dp2 = t >= 2 ? col[(t - 2) * V] : 0.f;
        const float dn1 = t + 1 < T ? col[(t + 1) * V] : 0.f, dn2 = t + 2 < T ? col[(t + 2) * V] : 0.f;
        if (t >= 1) { const float f1 = d - dp1; l += f1 * f1; g += 2.f * f1; }
        if (t + 1 < T) g -= 2.f * (dn1 - d);
        if (t >= 2) { const float s2 = d - 2.f * dp1 + dp2; l += s2 * s2; g += 2.f * s2; }
        if (t >= 1 && t + 1 < T) g -= 4.f * (dn1 - 2.f * d + dp1);
        if (t + 2 < T) g += 2.f * (dn2 - 2.f * dn1 + d);
        if (d_x_out) d_x_out[base + e] = -g;
        lsum += (double)l;
    }
    red[tid] = lsum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) part[n] = red[0];
}

__global__ __launch_bounds__(256) void vae_loss_finalize_kernel(const double* __restrict__ part,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ lsig, float beta,
                                                                float* __restrict__ loss, float* __restrict__ d_mean,
                                                                float* __restrict__ d_lsig, int N, int NZ) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double k = 0.0;
    if (mean && lsig)
        for (int i = tid; i < NZ; i += 256) {
            const float m = mean[i], s = lsig[i], ex = expf(s);
            k += (double)(1.f + s - m * m - ex);
            if (d_mean) d_mean[i] = beta * m;
            if (d_lsig) d_lsig[i] = -0.5f * beta * (1.f - ex);
        }
    red[tid] = k;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        double bf = 0.0;
        for (int n = 0; n < N; ++n) bf += part[n];
        const double kld = -0.5 * red[0];
        loss[0] = (float)(bf + (double)beta * kld);
        loss[1] = (float)bf;
        loss[2] = (float)kld;
    }
}

}  // namespace

extern "C" size_t step_vae_loss_ws_bytes(int N) { return N > 0 ? (size_t)N * sizeof(double) : 0; }

extern "C" int step_vae_loss(const float* x_in, const float* x_out, const float* mean, const float* lsig,
                             float beta, float* loss, float* d_x_out, float* d_mean, float* d_lsig, int N,
                             int C, int T, int V, int Z, void* ws, size_t ws_bytes, void* stream) {
    STEP_REQUIRE(x_in && x_out && loss && ws, "step_vae_loss: null pointer");
    STEP_REQUIRE(N >= 1 && C >= 1 && T >= 1 && V >= 1 && Z >= 0, "step_vae_loss: bad shape N=%d C=%d T=%d V=%d Z=%d",
                 N, C, T, V, Z);
    STEP_REQUIRE((mean == nullptr) == (lsig == nullptr), "step_vae_loss: mean and lsig come together");
    STEP_REQUIRE((long long)C * T * V <= LOSS_MAX_E,
                 "step_vae_loss: a sequence of C*T*V=%lld floats exceeds the %d kept in LDS", (long long)C * T * V,
                 LOSS_MAX_E);
    if (ws_bytes < (size_t)N * sizeof(double)) {
        step::set_error("step_vae_loss: workspace %zu < %zu bytes", ws_bytes, (size_t)N * sizeof(double));
        return STEP_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = ((size_t)C * T * V + (size_t)C * V) * sizeof(float);
    hipLaunchKernelGGL(vae_loss_seq_kernel, dim3(N), dim3(256), lds, st, x_in, x_out, d_x_out, (double*)ws, N, C, T, V);
    hipLaunchKernelGGL(vae_loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, mean, lsig, beta, loss,
                       d_mean, d_lsig, N, N * Z);
    return step::check_launch("step_vae_loss");
}
