// Temporal (Kt x 1) convolution of an input that does not depend on the frame.
//
// The CVAE decoder repeats one latent vector over all T frames and V joints before its first st_gctn block
// (generator_cvae/net/CVAE_stgcn.py Decoder.forward: `z.repeat([1, 1, T, V])`), so that block's graph conv and
// BN + ReLU give the same 16 joint values at every frame, and its 75-tap temporal conv collapses to one Co x Ci
// matrix per OUTPUT FRAME - the sum of the taps that fall inside the sequence there (zero padding cuts taps off
// near the ends, so the matrix depends on t):
//   u[n, co, t, w] = SUM_ci wcum[t][co][ci] * a0[n, ci, w],   wcum[t] = SUM_{j : frame(t, j) in [0, T)} w[.., j]
// Kt times fewer multiply-adds than the convolution and nothing of size N x C x T x V is read: the kernel is bound
// by writing its output.  Eval-mode epilogue as step_tcn_conv (bias, BN2 scale/shift, residual, ReLU); stride 1.
#include "common.h"

namespace {

// wcum[t][co][ci]; Conv2d: in frame t - pad + j, w (Co,Ci,Kt); ConvTranspose2d: in frame t + pad - j, w (Ci,Co,Kt)
__global__ void tcn_const_weights_kernel(const float* __restrict__ w, float* __restrict__ wcum, int Co, int Ci,
                                         int Kt, int T, int transposed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= T * Co * Ci) return;
    const int ci = e % Ci, co = (e / Ci) % Co, t = e / (Ci * Co);
    const int pad = (Kt - 1) / 2;
    const float* wr = transposed ? w + ((size_t)ci * Co + co) * Kt : w + ((size_t)co * Ci + ci) * Kt;
    float s = 0.f;
    for (int j = 0; j < Kt; ++j) {
        const int tin = transposed ? t + pad - j : t - pad + j;
        if (tin >= 0 && tin < T) s += wr[j];
    }
    wcum[e] = s;
}

struct TcParams {
    const float* a0;         // (N, Ci, 16)
    const float* wcum;       // (T, Co, Ci)
    const float* bias;
    const float* out_scale;
    const float* out_shift;
    const float* res0;       // (N, Co, 16) or null
    const float* res_scale;
    const float* res_shift;
    float* out;              // (N, Co, T, 16)
    int out_relu, N, Ci, Co, T, ntb;
};

constexpr int CF = 8;        // frames per block: 512 contiguous bytes per channel row leave as 16-byte stores

__global__ __launch_bounds__(256) void tcn_const_fwd_kernel(TcParams p) {
    __shared__ float a_s[64 * 16];
    __shared__ __attribute__((aligned(16))) float o_s[64 * CF * 16];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / p.ntb, t0 = (blockIdx.x % p.ntb) * CF;
    for (int i = tid; i < p.Ci * 16; i += 256) a_s[i] = p.a0[(size_t)n * p.Ci * 16 + i];
    __syncthreads();
    const int w = tid & 15;
    for (int co = tid >> 4; co < p.Co; co += 16) {
        const float bia = p.bias ? p.bias[co] : 0.f;
        float osc = 1.f, osh = 0.f, resv = 0.f;
        if (p.out_scale) { osc = p.out_scale[co]; osh = p.out_shift[co]; }
        if (p.res0) {
            resv = p.res0[((size_t)n * p.Co + co) * 16 + w];
            if (p.res_scale) resv = resv * p.res_scale[co] + p.res_shift[co];
        }
#pragma unroll
        for (int f = 0; f < CF; ++f) {
            const int t = t0 + f;
            if (t >= p.T) break;
            const float4* wr = reinterpret_cast<const float4*>(p.wcum + ((size_t)t * p.Co + co) * p.Ci);
            float acc = 0.f;
            for (int c4 = 0; c4 < p.Ci / 4; ++c4) {
                const float4 q = wr[c4];
                acc = fmaf(q.x, a_s[(4 * c4) * 16 + w], acc);
                acc = fmaf(q.y, a_s[(4 * c4 + 1) * 16 + w], acc);
                acc = fmaf(q.z, a_s[(4 * c4 + 2) * 16 + w], acc);
                acc = fmaf(q.w, a_s[(4 * c4 + 3) * 16 + w], acc);
            }
            float v = (acc + bia) * osc + osh + resv;
            if (p.out_relu) v = fmaxf(v, 0.f);
            o_s[(co * CF + f) * 16 + w] = v;
        }
    }
    __syncthreads();
    for (int i = tid; i < p.Co * CF * 4; i += 256) {
        const int co = i / (CF * 4), piece = i % (CF * 4);
        const int t = t0 + (piece >> 2);
        if (t < p.T)
            *reinterpret_cast<float4*>(p.out + (((size_t)n * p.Co + co) * p.T + t) * 16 + (piece & 3) * 4) =
                *reinterpret_cast<const float4*>(o_s + co * (CF * 16) + piece * 4);
    }
}

}  // namespace

extern "C" int step_tcn_const_weights(const float* w, float* wcum, int Co, int Ci, int Kt, int T, int transposed,
                                      void* stream) {
    STEP_REQUIRE(w && wcum, "step_tcn_const_weights: null pointer");
    STEP_REQUIRE(Co >= 1 && Ci >= 1 && Kt >= 1 && (Kt & 1) && T >= 1 && (long long)T * Co * Ci <= 0x7fffffffLL,
                 "step_tcn_const_weights: bad shape Co=%d Ci=%d Kt=%d T=%d", Co, Ci, Kt, T);
    const int total = T * Co * Ci;
    hipLaunchKernelGGL(tcn_const_weights_kernel, dim3(step::ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, wcum, Co, Ci, Kt, T, transposed);
    return step::check_launch("step_tcn_const_weights");
}

extern "C" int step_tcn_const_fwd(const float* a0, const float* wcum, const float* bias, const float* out_scale,
                                  const float* out_shift, const float* res0, const float* res_scale,
                                  const float* res_shift, int out_relu, float* out, int N, int Ci, int Co, int T,
                                  void* stream) {
    STEP_REQUIRE(a0 && wcum && out, "step_tcn_const_fwd: null pointer");
    STEP_REQUIRE(N >= 1 && T >= 1 && Ci >= 4 && Ci <= 64 && Ci % 4 == 0 && Co >= 1 && Co <= 64,
                 "step_tcn_const_fwd: N=%d Ci=%d Co=%d T=%d (Ci a multiple of 4, both <= 64)", N, Ci, Co, T);
    STEP_REQUIRE((out_scale == nullptr) == (out_shift == nullptr) && (res_scale == nullptr) == (res_shift == nullptr),
                 "step_tcn_const_fwd: scale/shift must come in pairs");
    TcParams p{};
    p.a0 = a0; p.wcum = wcum; p.bias = bias; p.out_scale = out_scale; p.out_shift = out_shift; p.res0 = res0;
    p.res_scale = res_scale; p.res_shift = res_shift; p.out = out; p.out_relu = out_relu;
    p.N = N; p.Ci = Ci; p.Co = Co; p.T = T; p.ntb = step::ceil_div(T, CF);
    const long long blocks = (long long)N * p.ntb;
    STEP_REQUIRE(blocks <= 0x7fffffffLL, "step_tcn_const_fwd: too many blocks");
    hipLaunchKernelGGL(tcn_const_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return step::check_launch("step_tcn_const_fwd");
}
