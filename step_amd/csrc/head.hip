// Global average pooling, the small dense heads and softmax.
//
// Replaces F.avg_pool2d(x, x.size()[2:]) + mean over M (classifier.py:81-82), the 1x1 `fcn`
// Conv2d on the pooled (N,64,1,1) feature (classifier.py:63,86), F.softmax (classifier.py:87),
// z_mean / z_lsig (CVAE_stgcn.py:105-106,131-134), Decoder.fcn (CVAE_stgcn.py:171,208) and the
// hybrid head's two 1x1 convs (classifier_hybrid/net/classifier.py:47-50,84-85).
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// one wavefront per (n, c): mean over `inner` contiguous floats
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x,
                                                       float* __restrict__ feat, int NC, int inner) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= NC) return;
    const int lane = threadIdx.x & 63;
    const float* src = x + (size_t)row * inner;
    float s = 0.f;
    const int n4 = inner >> 2;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = *reinterpret_cast<const float4*>(src + i * 4);
        s += (v.x + v.y) + (v.z + v.w);
    }
    for (int i = n4 * 4 + lane; i < inner; i += 64) s += src[i];
    s = wave_sum(s);
    if (lane == 0) feat[row] = s / (float)inner;
}

// Inference head in ONE launch: pooling, the 1x1 `fcn` and (real_only) the softmax of classifier.py:81-87 for one
// sequence per block.  Every sum runs in the order of pool_fwd_kernel / dense_fwd_kernel / softmax_fwd_kernel, so the
// results are bit-identical to the three separate launches the autograd path keeps.
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, int wt, int softmax,
                                                       float* __restrict__ feat, float* __restrict__ out,
                                                       int C, int inner, int O) {
    extern __shared__ float sh[];                       // [C] pooled feature, then [O] logits
    float* logit = sh + C;
    const int n = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n4 = inner >> 2;
    for (int c = wave; c < C; c += 4) {
        const float* src = x + ((size_t)n * C + c) * inner;
        float s = 0.f;
        for (int i = lane; i < n4; i += 64) {
            const float4 v = *reinterpret_cast<const float4*>(src + i * 4);
            s += (v.x + v.y) + (v.z + v.w);
        }
        for (int i = n4 * 4 + lane; i < inner; i += 64) s += src[i];
        s = wave_sum(s);
        if (lane == 0) {
            const float f = s / (float)inner;
            sh[c] = f;
            feat[(size_t)n * C + c] = f;
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < O; o += 256) {
        float s = b ? b[o] : 0.f;
        if (wt) for (int i = 0; i < C; ++i) s = fmaf(sh[i], w[(size_t)i * O + o], s);
        else    for (int i = 0; i < C; ++i) s = fmaf(sh[i], w[(size_t)o * C + i], s);
        logit[o] = s;
        if (!softmax) out[(size_t)n * O + o] = s;
    }
    if (!softmax) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = logit[0];
        for (int o = 1; o < O; ++o) m = fmaxf(m, logit[o]);
        float s = 0.f;
        for (int o = 0; o < O; ++o) s += expf(logit[o] - m);
        const float inv = 1.f / s;
        for (int o = 0; o < O; ++o) out[(size_t)n * O + o] = expf(logit[o] - m) * inv;
    }
}

// one float4 per thread (inner % 4 == 0: a float4 never straddles two (n, c) rows), written once: non-temporal
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dfeat,
                                                       float* __restrict__ dx, size_t total4,
                                                       int inner4) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const float inv = 1.f / (float)(inner4 * 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (size_t)gridDim.x * blockDim.x) {
        const float v = dfeat[i / inner4] * inv;
        __builtin_nontemporal_store(f4v{v, v, v, v}, reinterpret_cast<f4v*>(dx) + i);
    }
}

// y[n,o] = b[o] + SUM_i x[n,i] * W(o,i); one thread per output
__global__ void dense_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                 const float* __restrict__ b, int wt, float* __restrict__ y, int N,
                                 int I, int O) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * O) return;
    const int n = e / O, o = e % O;
    float s = b ? b[o] : 0.f;
    const float* xr = x + (size_t)n * I;
    if (wt) for (int i = 0; i < I; ++i) s = fmaf(xr[i], w[(size_t)i * O + o], s);
    else    for (int i = 0; i < I; ++i) s = fmaf(xr[i], w[(size_t)o * I + i], s);
    y[e] = s;
}

// dx[n,i] = SUM_o dy[n,o] * W(o,i)
__global__ void dense_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, int wt,
                                    float* __restrict__ dx, int N, int I, int O) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * I) return;
    const int n = e / I, i = e % I;
    float s = 0.f;
    for (int o = 0; o < O; ++o)
        s = fmaf(dy[(size_t)n * O + o], wt ? w[(size_t)i * O + o] : w[(size_t)o * I + i], s);
    dx[e] = s;
}

// dW(o,i) = SUM_n dy[n,o]*x[n,i]; db[o] = SUM_n dy[n,o].  One block per output element: thread t sums rows
// t, t + 256, ... in fp64 and the 256 partials are combined in a fixed order (reproducible).  (One THREAD per
// element, serial over n, took 0.9 ms of every training step at N = 4096 for the 64 x 4 fcn.)
__global__ __launch_bounds__(256) void dense_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           int wt, float* __restrict__ dw, float* __restrict__ db,
                                                           int N, int I, int O) {
    const int e = blockIdx.x, tid = threadIdx.x;
    const bool is_w = e < O * I;
    const int o = is_w ? e / I : e - O * I, i = is_w ? e % I : 0;
    if (is_w ? dw == nullptr : db == nullptr) return;
    double s = 0.0;
    if (is_w) {
        for (int n = tid; n < N; n += 256) s += (double)dy[(size_t)n * O + o] * (double)x[(size_t)n * I + i];
    } else {
        for (int n = tid; n < N; n += 256) s += (double)dy[(size_t)n * O + o];
    }
    __shared__ double red[256];
    red[tid] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) red[tid] += red[tid + h];
        __syncthreads();
    }
    if (tid == 0) {
        if (is_w) dw[wt ? (size_t)i * O + o : (size_t)o * I + i] = (float)red[0];
        else db[o] = (float)red[0];
    }
}

__global__ void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int O) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* xr = x + (size_t)n * O;
    float m = xr[0];
    for (int o = 1; o < O; ++o) m = fmaxf(m, xr[o]);
    float s = 0.f;
    for (int o = 0; o < O; ++o) s += expf(xr[o] - m);
    const float inv = 1.f / s;
    for (int o = 0; o < O; ++o) y[(size_t)n * O + o] = expf(xr[o] - m) * inv;
}

__global__ void softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                   float* __restrict__ dx, int N, int O) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float dot = 0.f;
    for (int o = 0; o < O; ++o) dot = fmaf(dy[(size_t)n * O + o], y[(size_t)n * O + o], dot);
    for (int o = 0; o < O; ++o)
        dx[(size_t)n * O + o] = y[(size_t)n * O + o] * (dy[(size_t)n * O + o] - dot);
}

// BatchNorm1d on a (N, G) matrix: one wavefront per column g (the hybrid head's data_bn2)
__global__ __launch_bounds__(64) void bn_rows_fwd_kernel(const float* __restrict__ x,
                                                         const float* gamma, const float* beta,
                                                         float* running_mean, float* running_var,
                                                         int training, float momentum, float eps,
                                                         float* __restrict__ y, float* mean_out,
                                                         float* invstd_out, int N, int G) {
    const int g = blockIdx.x, lane = threadIdx.x;
    float m, inv;
    if (training) {
        // two passes in fp64 (mean, then SUM (x - mean)^2): a near-constant column such as 100 + 1e-2*randn
        // would lose its variance to cancellation in a one-pass fp32 ss/N - m*m (N is small here)
        double s = 0.0;
        for (int n = lane; n < N; n += 64) s += (double)x[(size_t)n * G + g];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const double md = s / N;
        double ss = 0.0;
        for (int n = lane; n < N; n += 64) { const double d = (double)x[(size_t)n * G + g] - md; ss += d * d; }
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const double vard = ss / N;
        m = (float)md;
        inv = (float)(1.0 / sqrt(vard + (double)eps));
        if (lane == 0 && running_mean) {
            const double unb = N > 1 ? vard * N / (N - 1) : vard;
            running_mean[g] = (float)((1.0 - momentum) * running_mean[g] + momentum * md);
            running_var[g] = (float)((1.0 - momentum) * running_var[g] + momentum * unb);
        }
    } else {
        m = running_mean[g];
        inv = 1.f / sqrtf(running_var[g] + eps);
    }
    if (lane == 0) { mean_out[g] = m; invstd_out[g] = inv; }
    const float sc = (gamma ? gamma[g] : 1.f) * inv, sh = (beta ? beta[g] : 0.f) - m * sc;
    for (int n = lane; n < N; n += 64) y[(size_t)n * G + g] = x[(size_t)n * G + g] * sc + sh;
}

__global__ __launch_bounds__(64) void bn_rows_bwd_kernel(const float* __restrict__ dy,
                                                         const float* __restrict__ x,
                                                         const float* mean, const float* invstd,
                                                         const float* gamma, int training,
                                                         float* __restrict__ dx, float* dgamma,
                                                         float* dbeta, int N, int G) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const float m = mean[g], inv = invstd[g], ga = gamma ? gamma[g] : 1.f;
    float s1 = 0.f, s2 = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float d = dy[(size_t)n * G + g];
        s1 += d; s2 += d * (x[(size_t)n * G + g] - m) * inv;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    s1 = __shfl(s1, 0, 64); s2 = __shfl(s2, 0, 64);
    if (lane == 0) { if (dgamma) dgamma[g] = s2; if (dbeta) dbeta[g] = s1; }
    if (!dx) return;
    for (int n = lane; n < N; n += 64) {
        const float d = dy[(size_t)n * G + g];
        float r = d;
        if (training) r = d - s1 / N - (x[(size_t)n * G + g] - m) * inv * s2 / N;
        dx[(size_t)n * G + g] = ga * inv * r;
    }
}

// ---- the same BatchNorm1d with its batch statistics taken over SEVERAL ranks (synchronised BatchNorm): the sums
// leave as doubles (exact products of fp32 inputs, so the one-pass variance keeps a near-constant column), the
// host all-reduces [sums, count], and the apply kernels take the combined buffer (count in slot 2G).
__global__ __launch_bounds__(64) void bn_rows_stats_kernel(const float* __restrict__ x, double* __restrict__ sums,
                                                           int N, int G) {
    const int g = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, ss = 0.0;
    for (int n = lane; n < N; n += 64) { const double v = (double)x[(size_t)n * G + g]; s += v; ss += v * v; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
    if (lane == 0) { sums[2 * g] = s; sums[2 * g + 1] = ss; }
}

__global__ __launch_bounds__(64) void bn_rows_apply_kernel(const float* __restrict__ x, const double* __restrict__ sums,
                                                           const float* gamma, const float* beta, float* running_mean,
                                                           float* running_var, float momentum, float eps,
                                                           float* __restrict__ y, float* mean_out, float* invstd_out,
                                                           int N, int G) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const double cnt = sums[2 * G];
    const double md = sums[2 * g] / cnt;
    double vard = sums[2 * g + 1] / cnt - md * md;
    vard = vard > 0.0 ? vard : 0.0;
    const float m = (float)md, inv = (float)(1.0 / sqrt(vard + (double)eps));
    if (lane == 0) {
        mean_out[g] = m; invstd_out[g] = inv;
        if (running_mean) {
            const double unb = cnt > 1.0 ? vard * cnt / (cnt - 1.0) : vard;
            running_mean[g] = (float)((1.0 - momentum) * running_mean[g] + momentum * md);
            running_var[g] = (float)((1.0 - momentum) * running_var[g] + momentum * unb);
        }
    }
    const float sc = (gamma ? gamma[g] : 1.f) * inv, sh = (beta ? beta[g] : 0.f) - m * sc;
    for (int n = lane; n < N; n += 64) y[(size_t)n * G + g] = x[(size_t)n * G + g] * sc + sh;
}

// sums[2g] = SUM dy, sums[2g+1] = SUM dy * xhat over this rank's rows (doubles); dgamma / dbeta from them
__global__ __launch_bounds__(64) void bn_rows_bwd_sums_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* mean, const float* invstd,
                                                              double* __restrict__ sums, float* dgamma, float* dbeta,
                                                              int N, int G) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const float m = mean[g], inv = invstd[g];
    double s1 = 0.0, s2 = 0.0;
    for (int n = lane; n < N; n += 64) {
        const float d = dy[(size_t)n * G + g];
        s1 += (double)d; s2 += (double)(d * (x[(size_t)n * G + g] - m) * inv);
    }
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) {
        sums[2 * g] = s1; sums[2 * g + 1] = s2;
        if (dgamma) dgamma[g] = (float)s2;
        if (dbeta) dbeta[g] = (float)s1;
    }
}

__global__ __launch_bounds__(64) void bn_rows_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* mean, const float* invstd, const float* gamma,
                                                               const double* __restrict__ sums, float* __restrict__ dx,
                                                               int N, int G) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const float m = mean[g], inv = invstd[g], ga = gamma ? gamma[g] : 1.f;
    const double cnt = sums[2 * G];
    const float a = (float)(sums[2 * g] / cnt), b = (float)(sums[2 * g + 1] / cnt);
    for (int n = lane; n < N; n += 64) {
        const float d = dy[(size_t)n * G + g];
        dx[(size_t)n * G + g] = ga * inv * (d - a - (x[(size_t)n * G + g] - m) * inv * b);
    }
}

}  // namespace

extern "C" int step_bn_rows_stats(const float* x, double* sums, int N, int G, void* stream) {
    STEP_REQUIRE(x && sums && N > 0 && G > 0, "step_bn_rows_stats: bad args");
    hipLaunchKernelGGL(bn_rows_stats_kernel, dim3(G), dim3(64), 0, (hipStream_t)stream, x, sums, N, G);
    return step::check_launch("step_bn_rows_stats");
}

extern "C" int step_bn_rows_apply(const float* x, const double* sums, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, float* y,
                                  float* mean, float* invstd, int N, int G, void* stream) {
    STEP_REQUIRE(x && sums && y && mean && invstd && N > 0 && G > 0, "step_bn_rows_apply: bad args");
    STEP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "step_bn_rows_apply: running stats come in pairs");
    hipLaunchKernelGGL(bn_rows_apply_kernel, dim3(G), dim3(64), 0, (hipStream_t)stream, x, sums, gamma, beta,
                       running_mean, running_var, momentum, eps, y, mean, invstd, N, G);
    return step::check_launch("step_bn_rows_apply");
}

extern "C" int step_bn_rows_bwd_sums(const float* dy, const float* x, const float* mean, const float* invstd,
                                     double* sums, float* dgamma, float* dbeta, int N, int G, void* stream) {
    STEP_REQUIRE(dy && x && mean && invstd && sums && N > 0 && G > 0, "step_bn_rows_bwd_sums: bad args");
    hipLaunchKernelGGL(bn_rows_bwd_sums_kernel, dim3(G), dim3(64), 0, (hipStream_t)stream, dy, x, mean, invstd, sums,
                       dgamma, dbeta, N, G);
    return step::check_launch("step_bn_rows_bwd_sums");
}

extern "C" int step_bn_rows_bwd_apply(const float* dy, const float* x, const float* mean, const float* invstd,
                                      const float* gamma, const double* sums, float* dx, int N, int G, void* stream) {
    STEP_REQUIRE(dy && x && mean && invstd && sums && dx && N > 0 && G > 0, "step_bn_rows_bwd_apply: bad args");
    hipLaunchKernelGGL(bn_rows_bwd_apply_kernel, dim3(G), dim3(64), 0, (hipStream_t)stream, dy, x, mean, invstd, gamma,
                       sums, dx, N, G);
    return step::check_launch("step_bn_rows_bwd_apply");
}

extern "C" int step_bn_rows_fwd(const float* x, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int training,
                                float momentum, float eps, float* y, float* mean, float* invstd,
                                int N, int G, void* stream) {
    STEP_REQUIRE(x && y && mean && invstd && N > 0 && G > 0, "step_bn_rows_fwd: bad args");
    STEP_REQUIRE(training || (running_mean && running_var), "step_bn_rows_fwd: eval needs running stats");
    hipLaunchKernelGGL(bn_rows_fwd_kernel, dim3(G), dim3(64), 0, (hipStream_t)stream, x, gamma, beta,
                       running_mean, running_var, training, momentum, eps, y, mean, invstd, N, G);
    return step::check_launch("step_bn_rows_fwd");
}

extern "C" int step_bn_rows_bwd(const float* dy, const float* x, const float* mean,
                                const float* invstd, const float* gamma, int training, float* dx,
                                float* dgamma, float* dbeta, int N, int G, void* stream) {
    STEP_REQUIRE(dy && x && mean && invstd && N > 0 && G > 0, "step_bn_rows_bwd: bad args");
    hipLaunchKernelGGL(bn_rows_bwd_kernel, dim3(G), dim3(64), 0, (hipStream_t)stream, dy, x, mean,
                       invstd, gamma, training, dx, dgamma, dbeta, N, G);
    return step::check_launch("step_bn_rows_bwd");
}

extern "C" int step_pool_fwd(const float* x, float* feat, int N, int C, int inner, void* stream) {
    STEP_REQUIRE(x && feat && N > 0 && C > 0 && inner > 0 && (inner % 4) == 0,
                 "step_pool_fwd: bad args N=%d C=%d inner=%d", N, C, inner);
    const int NC = N * C;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(step::ceil_div(NC, 4)), dim3(256), 0, (hipStream_t)stream,
                       x, feat, NC, inner);
    return step::check_launch("step_pool_fwd");
}

extern "C" int step_head_fwd(const float* x, const float* w, const float* b, int w_transposed, int softmax,
                             float* feat, float* out, int N, int C, int inner, int O, void* stream) {
    STEP_REQUIRE(x && w && feat && out && N > 0 && C > 0 && C <= 1024 && O > 0 && O <= 1024 && inner > 0,
                 "step_head_fwd: bad args N=%d C=%d inner=%d O=%d", N, C, inner, O);
    hipLaunchKernelGGL(head_fwd_kernel, dim3(N), dim3(256), (C + O) * sizeof(float), (hipStream_t)stream, x, w, b,
                       w_transposed, softmax, feat, out, C, inner, O);
    return step::check_launch("step_head_fwd");
}

extern "C" int step_pool_bwd(const float* dfeat, float* dx, int N, int C, int inner, void* stream) {
    STEP_REQUIRE(dfeat && dx && N > 0 && C > 0 && inner > 0 && (inner % 4) == 0,
                 "step_pool_bwd: bad args N=%d C=%d inner=%d", N, C, inner);
    const size_t total4 = (size_t)N * C * (inner / 4);
    size_t blocks = (total4 + 255) / 256;
    if (blocks > 0x7fffffffu) blocks = 0x7fffffffu;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dfeat,
                       dx, total4, inner / 4);
    return step::check_launch("step_pool_bwd");
}

extern "C" int step_dense_fwd(const float* x, const float* w, const float* b, int w_transposed,
                              float* y, int N, int I, int O, void* stream) {
    STEP_REQUIRE(x && w && y && N > 0 && I > 0 && O > 0, "step_dense_fwd: bad args");
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(step::ceil_div(N * O, 128)), dim3(128), 0,
                       (hipStream_t)stream, x, w, b, w_transposed, y, N, I, O);
    return step::check_launch("step_dense_fwd");
}

extern "C" int step_dense_bwd(const float* dy, const float* x, const float* w, int w_transposed,
                              float* dx, float* dw, float* db, int N, int I, int O, void* stream) {
    STEP_REQUIRE(dy && x && w && N > 0 && I > 0 && O > 0, "step_dense_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dx)
        hipLaunchKernelGGL(dense_bwd_dx_kernel, dim3(step::ceil_div(N * I, 128)), dim3(128), 0, st, dy,
                           w, w_transposed, dx, N, I, O);
    if (dw || db)
        hipLaunchKernelGGL(dense_bwd_dw_kernel, dim3(O * I + O), dim3(256), 0, st, dy, x, w_transposed, dw, db,
                           N, I, O);
    return step::check_launch("step_dense_bwd");
}

extern "C" int step_softmax_fwd(const float* x, float* y, int N, int O, void* stream) {
    STEP_REQUIRE(x && y && N > 0 && O > 0, "step_softmax_fwd: bad args");
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(step::ceil_div(N, 128)), dim3(128), 0,
                       (hipStream_t)stream, x, y, N, O);
    return step::check_launch("step_softmax_fwd");
}

extern "C" int step_softmax_bwd(const float* dy, const float* y, float* dx, int N, int O,
                                void* stream) {
    STEP_REQUIRE(dy && y && dx && N > 0 && O > 0, "step_softmax_bwd: bad args");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(step::ceil_div(N, 128)), dim3(128), 0,
                       (hipStream_t)stream, dy, y, dx, N, O);
    return step::check_launch("step_softmax_bwd");
}
