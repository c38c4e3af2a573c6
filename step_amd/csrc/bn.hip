// BatchNorm statistics / affine / backward kernels (HBM-bound, fp32 data, fp64 partial sums).
//
// Replaces nn.BatchNorm2d (tcn[0], tcn[3], residual[1]: classifier.py:154,163,180) and the
// `data_bn` BatchNorm1d on the (N, V*C, T) view (classifier.py:68-74) in both modes, the
// add + ReLU tail of st_gcn.forward (classifier.py:189-191) and their autograd backward.
//
// Layout: x is (N, C, T, V=16) contiguous.  A block owns one channel c and one chunk of
// the batch; 4 lanes cover one 64-byte (t, :) row with float4 loads, so a wavefront reads
// 1 KiB contiguous.  Per-lane partial sums are reduced through LDS in a fixed order and
// written as fp64 partials [chunk][group][2]; a one-block finalize sums the chunks in order,
// so results are bit-reproducible run to run.
#include "common.h"
#include <cstdlib>
#include <mutex>

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_CHUNKS = 256;

__device__ __forceinline__ int group_of(int c, int v, int C, int mode) {
    return mode ? v * C + c : c;
}

// g = dy * mask ; returns g for 4 consecutive v
template <int MASK>
__device__ __forceinline__ float4 masked(const float4 dy, const float4 x, const float4 m,
                                         const float4 sc, const float4 sh) {
    float4 g = dy;
    if (MASK == 1) {
        g.x = m.x > 0.f ? dy.x : 0.f; g.y = m.y > 0.f ? dy.y : 0.f;
        g.z = m.z > 0.f ? dy.z : 0.f; g.w = m.w > 0.f ? dy.w : 0.f;
    } else if (MASK == 2) {
        g.x = (x.x * sc.x + sh.x) > 0.f ? dy.x : 0.f; g.y = (x.y * sc.y + sh.y) > 0.f ? dy.y : 0.f;
        g.z = (x.z * sc.z + sh.z) > 0.f ? dy.z : 0.f; g.w = (x.w * sc.w + sh.w) > 0.f ? dy.w : 0.f;
    }
    return g;
}

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void store4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Non-temporal variants for tensors touched once.  Measured (N = 4096, tools/gbench.py): the read-only statistics
// passes 0.73 -> 0.82 of the HBM peak with non-temporal loads; the backward apply passes 0.66 -> 0.69-0.73 with
// non-temporal loads AND stores (loads alone: no gain); affine_act is best with plain accesses (0.58-0.63).
__device__ __forceinline__ float4 load4nt(const float* p) {
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store4nt(float* p, const float4 v) {
    __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(p));
}

__device__ __forceinline__ float4 group4(const float* tab, int c, int v0, int C, int mode) {
    if (!tab) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == 0) { const float s = tab[c]; return make_float4(s, s, s, s); }
    return make_float4(tab[(v0 + 0) * C + c], tab[(v0 + 1) * C + c], tab[(v0 + 2) * C + c],
                       tab[(v0 + 3) * C + c]);
}

// Block (c, chunk): accumulates A = SUM a, B = SUM b per joint v over rows (n in chunk, t).
//   STATS: a = x, b = x*x.     BWD: a = g, b = g*x  with g = dy*mask.
template <bool BWD, int MASK>
__global__ __launch_bounds__(BN_THREADS) void bn_reduce_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ msrc,
    const float* __restrict__ scale, const float* __restrict__ shift, double* __restrict__ part,
    int N, int C, int T, int V, int mode, int nchunks) {
    const int c = blockIdx.x;
    const int chunk = blockIdx.y;
    const int n0 = (int)((long long)N * chunk / nchunks);
    const int n1 = (int)((long long)N * (chunk + 1) / nchunks);
    const int tid = threadIdx.x;
    const int v0 = (tid & 3) * 4;
    const int slot = tid >> 2;                       // 64 row slots
    const float4 sc = group4(scale, c, v0, C, mode);
    const float4 sh = group4(shift, c, v0, C, mode);
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = n0; n < n1; ++n) {
        const size_t base = ((size_t)n * C + c) * T * 16;
        for (int t = slot; t < T; t += 64) {
            const size_t o = base + (size_t)t * 16 + v0;
            const float4 xv = load4nt(x + o);
            if (!BWD) {
                a[0] += xv.x; a[1] += xv.y; a[2] += xv.z; a[3] += xv.w;
                b[0] += xv.x * xv.x; b[1] += xv.y * xv.y; b[2] += xv.z * xv.z; b[3] += xv.w * xv.w;
            } else {
                const float4 d = load4nt(dy + o);
                float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MASK == 1) m = load4nt(msrc + o);
                const float4 g = masked<MASK>(d, xv, m, sc, sh);
                a[0] += g.x; a[1] += g.y; a[2] += g.z; a[3] += g.w;
                b[0] += g.x * xv.x; b[1] += g.y * xv.y; b[2] += g.z * xv.z; b[3] += g.w * xv.w;
            }
        }
    }
    __shared__ double red[2][16][65];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[0][v0 + k][slot] = (double)a[k];
        red[1][v0 + k][slot] = (double)b[k];
    }
    __syncthreads();
    if (tid < 32) {
        const int which = tid >> 4, v = tid & 15;
        double s = 0.0;
        for (int k = 0; k < 64; ++k) s += red[which][v][k];
        red[which][v][64] = s;
    }
    __syncthreads();
    const int G = mode ? 16 * C : C;
    double* dst = part + (size_t)chunk * G * 2;
    if (mode) {
        if (tid < 32) {
            const int which = tid >> 4, v = tid & 15;
            dst[(v * C + c) * 2 + which] = red[which][v][64];
        }
    } else if (tid < 2) {
        double s = 0.0;
        for (int v = 0; v < 16; ++v) s += red[tid][v][64];
        dst[c * 2 + tid] = s;
    }
}

// sums the chunk partials in a fixed order; writes fp64 totals [G][2].  A block is 64 entries x 4 chunk
// quarters, 4 independent chains per thread (the loads are latency-bound: 256 chunks in 16 rounds instead of
// 64), the quarters combined through LDS
__global__ __launch_bounds__(256) void bn_sum_chunks_kernel(const double* __restrict__ part, double* __restrict__ tot,
                                                            int G, int nchunks) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int ka = nchunks * g / 4, kb = nchunks * (g + 1) / 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (e < 2 * G) {
        int k = ka;
        for (; k + 3 < kb; k += 4) {
            s0 += part[(size_t)k * G * 2 + e];
            s1 += part[(size_t)(k + 1) * G * 2 + e];
            s2 += part[(size_t)(k + 2) * G * 2 + e];
            s3 += part[(size_t)(k + 3) * G * 2 + e];
        }
        for (; k < kb; ++k) s0 += part[(size_t)k * G * 2 + e];
    }
    red[g][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < 2 * G) tot[e] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

__global__ void bn_train_finalize_kernel(const double* __restrict__ tot, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float* running_mean,
                                         float* running_var, float momentum, float eps,
                                         float* mean, float* invstd, float* scale, float* shift,
                                         int G, double count) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    if (count <= 0.0) count = tot[2 * G];        // combined element count behind the sums (synchronised BatchNorm)
    const double m = tot[2 * g] / count;
    double var = tot[2 * g + 1] / count - m * m;
    var = var > 0.0 ? var : 0.0;
    const double inv = 1.0 / sqrt(var + (double)eps);
    const float ga = gamma ? gamma[g] : 1.f, be = beta ? beta[g] : 0.f;
    mean[g] = (float)m;
    invstd[g] = (float)inv;
    const float sc = (float)(ga * inv);
    scale[g] = sc;
    shift[g] = (float)((double)be - m * (double)ga * inv);
    if (running_mean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[g] = (float)((1.0 - momentum) * running_mean[g] + momentum * m);
        running_var[g] = (float)((1.0 - momentum) * running_var[g] + momentum * unb);
    }
}

__global__ void bn_eval_prepare_kernel(const float* gamma, const float* beta, const float* rm,
                                       const float* rv, float eps, float* scale, float* shift, int G) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const float inv = (float)(1.0 / sqrt((double)rv[g] + (double)eps));
    const float sc = (gamma ? gamma[g] : 1.f) * inv;
    scale[g] = sc;
    shift[g] = (beta ? beta[g] : 0.f) - rm[g] * sc;
}

// out = act(x*scale+shift + res*rscale+rshift), float4 over v
__global__ __launch_bounds__(256) void affine_act_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ res, const float* __restrict__ rscale, const float* __restrict__ rshift,
    int relu, float* __restrict__ out, size_t n4, int C, int T, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int v0 = (int)(i & 3) * 4;
        const int c = (int)((i >> 2) / T % C);
        float4 xv = load4(x + i * 4);
        if (scale) {
            const float4 sc = group4(scale, c, v0, C, mode), sh = group4(shift, c, v0, C, mode);
            xv.x = xv.x * sc.x + sh.x; xv.y = xv.y * sc.y + sh.y;
            xv.z = xv.z * sc.z + sh.z; xv.w = xv.w * sc.w + sh.w;
        }
        if (res) {
            float4 r = load4(res + i * 4);
            if (rscale) {
                const float4 sc = group4(rscale, c, v0, C, mode), sh = group4(rshift, c, v0, C, mode);
                r.x = r.x * sc.x + sh.x; r.y = r.y * sc.y + sh.y;
                r.z = r.z * sc.z + sh.z; r.w = r.w * sc.w + sh.w;
            }
            xv.x += r.x; xv.y += r.y; xv.z += r.z; xv.w += r.w;
        }
        if (relu) {
            xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f);
            xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f);
        }
        store4(out + i * 4, xv);
    }
}

// per-group coefficients for the apply pass: dx = g*ka + x*kb + kc
//   training: ka = gamma*invstd, kb = -ka*invstd*Sgx/M, kc = -ka*S1/M - kb*mean   (Sgx = S2 - mean*S1)
//   eval    : ka = scale,        kb = 0,                 kc = 0
__global__ void bn_bwd_coef_kernel(const double* __restrict__ tot, const float* __restrict__ mean,
                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                   int training, double count, float* __restrict__ coef,
                                   float* dgamma, float* dbeta, int G) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const double S1 = tot[2 * g], S2 = tot[2 * g + 1];
    const double m = mean[g], inv = invstd[g], ga = gamma ? gamma[g] : 1.0;
    const double sgx = S2 - m * S1;                 // SUM g*(x-mean)
    if (dgamma) dgamma[g] = (float)(sgx * inv);
    if (dbeta) dbeta[g] = (float)S1;
    const double ka = ga * inv;
    double kb = 0.0, kc = 0.0;
    if (training) {
        kb = -ka * inv * inv * sgx / count;
        kc = -ka * S1 / count - kb * m;
    }
    coef[3 * g] = (float)ka;
    coef[3 * g + 1] = (float)kb;
    coef[3 * g + 2] = (float)kc;
}

template <int MASK>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ msrc,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ coef,
    int accumulate, float* __restrict__ dx, size_t n4, int C, int T, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int v0 = (int)(i & 3) * 4;
        const int c = (int)((i >> 2) / T % C);
        const float4 xv = load4nt(x + i * 4);
        const float4 d = load4nt(dy + i * 4);
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MASK == 1) m = load4nt(msrc + i * 4);
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MASK == 2) { sc = group4(scale, c, v0, C, mode); sh = group4(shift, c, v0, C, mode); }
        const float4 g = masked<MASK>(d, xv, m, sc, sh);
        float r[4];
        const float gg[4] = {g.x, g.y, g.z, g.w};
        const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int grp = group_of(c, v0 + k, C, mode);
            r[k] = gg[k] * coef[3 * grp] + xx[k] * coef[3 * grp + 1] + coef[3 * grp + 2];
        }
        float4 o = make_float4(r[0], r[1], r[2], r[3]);
        if (accumulate) {
            const float4 old = load4nt(dx + i * 4);
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        store4nt(dx + i * 4, o);
    }
}

// ---- block tail with a residual conv: BN2 (of u) and the residual BN (of r) share dout and the ReLU
// mask of `out`, so their backward passes run as ONE reduce and ONE apply (4 tensor reads instead of 6 in
// each).  Same per-lane summation order as bn_reduce_kernel<true, 1>, so the sums are bit-identical.
__global__ __launch_bounds__(BN_THREADS) void bn_reduce_pair_kernel(
    const float* __restrict__ dy, const float* __restrict__ msrc, const float* __restrict__ xu,
    const float* __restrict__ xr, double* __restrict__ part_u, double* __restrict__ part_r,
    int N, int C, int T, int nchunks) {
    const int c = blockIdx.x;
    const int chunk = blockIdx.y;
    const int n0 = (int)((long long)N * chunk / nchunks);
    const int n1 = (int)((long long)N * (chunk + 1) / nchunks);
    const int tid = threadIdx.x;
    const int v0 = (tid & 3) * 4;
    const int slot = tid >> 2;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, bu[4] = {0.f, 0.f, 0.f, 0.f}, br[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = n0; n < n1; ++n) {
        const size_t base = ((size_t)n * C + c) * T * 16;
        for (int t = slot; t < T; t += 64) {
            const size_t o = base + (size_t)t * 16 + v0;
            const float4 d = load4nt(dy + o), m = load4nt(msrc + o), u = load4nt(xu + o), r = load4nt(xr + o);
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 g = masked<1>(d, z, m, z, z);
            a[0] += g.x; a[1] += g.y; a[2] += g.z; a[3] += g.w;
            bu[0] += g.x * u.x; bu[1] += g.y * u.y; bu[2] += g.z * u.z; bu[3] += g.w * u.w;
            br[0] += g.x * r.x; br[1] += g.y * r.y; br[2] += g.z * r.z; br[3] += g.w * r.w;
        }
    }
    __shared__ double red[3][16][65];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[0][v0 + k][slot] = (double)a[k];
        red[1][v0 + k][slot] = (double)bu[k];
        red[2][v0 + k][slot] = (double)br[k];
    }
    __syncthreads();
    if (tid < 48) {
        const int which = tid >> 4, v = tid & 15;
        double s = 0.0;
        for (int k = 0; k < 64; ++k) s += red[which][v][k];
        red[which][v][64] = s;
    }
    __syncthreads();
    if (tid < 3) {
        double s = 0.0;
        for (int v = 0; v < 16; ++v) s += red[tid][v][64];
        const size_t e = ((size_t)chunk * C + c) * 2;
        if (tid == 0) { part_u[e] = s; part_r[e] = s; }
        else if (tid == 1) part_u[e + 1] = s;
        else part_r[e + 1] = s;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_pair_kernel(
    const float* __restrict__ dy, const float* __restrict__ msrc, const float* __restrict__ xu,
    const float* __restrict__ xr, const float* __restrict__ coef_u, const float* __restrict__ coef_r,
    float* __restrict__ du, float* __restrict__ dr, size_t n4, int C, int T) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i >> 2) / T % C);
        const float4 d = load4nt(dy + i * 4), m = load4nt(msrc + i * 4), u = load4nt(xu + i * 4), r = load4nt(xr + i * 4);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 g = masked<1>(d, z, m, z, z);
        const float au = coef_u[3 * c], bu = coef_u[3 * c + 1], cu = coef_u[3 * c + 2];
        const float ar = coef_r[3 * c], br = coef_r[3 * c + 1], cr = coef_r[3 * c + 2];
        store4nt(du + i * 4, make_float4(g.x * au + u.x * bu + cu, g.y * au + u.y * bu + cu,
                                       g.z * au + u.z * bu + cu, g.w * au + u.w * bu + cu));
        store4nt(dr + i * 4, make_float4(g.x * ar + r.x * br + cr, g.y * ar + r.y * br + cr,
                                       g.z * ar + r.z * br + cr, g.w * ar + r.w * br + cr));
    }
}

__global__ __launch_bounds__(256) void masked_add_kernel(const float* __restrict__ dy,
                                                         const float* __restrict__ msrc,
                                                         float* __restrict__ dx, int accumulate,
                                                         size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        const float4 d = load4nt(dy + i * 4), m = load4nt(msrc + i * 4);
        float4 o = make_float4(m.x > 0.f ? d.x : 0.f, m.y > 0.f ? d.y : 0.f, m.z > 0.f ? d.z : 0.f,
                               m.w > 0.f ? d.w : 0.f);
        if (accumulate) {
            const float4 old = load4nt(dx + i * 4);
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        store4nt(dx + i * 4, o);
    }
}

int elem_grid(size_t n4) {
    // one float4 per thread: blocks then walk memory in dispatch order, which measured 0.59 -> 0.76 of the HBM
    // peak for affine_act and 0.69 -> 0.79 for the backward apply against a 4096-block grid-stride launch
    // (the kernels keep their grid-stride loop for tensors beyond 2^31 blocks)
    size_t b = (n4 + 255) / 256;
    const size_t cap = 0x7fffffffu;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int chunks_for(int N) { return N < BN_MAX_CHUNKS ? N : BN_MAX_CHUNKS; }

// workspace: [chunk partials: MAX_CHUNKS*G*2 doubles][totals: 2G doubles][coef: 3G floats]
size_t ws_bytes(int G) { return ((size_t)BN_MAX_CHUNKS * G * 2 + 2 * (size_t)G) * 8 + 3 * (size_t)G * 4; }

}  // namespace

extern "C" size_t step_bn_ws_bytes(int G) { return G > 0 ? ws_bytes(G) : 0; }

// ordered sum of per-block fp64 partials [nchunks][G][2] (written by a producer kernel's epilogue, gcn.hip) into
// sums [G][2]: the second stage of step_bn_train_stats for statistics taken where the tensor is produced
int step::bn_sum_partials(const double* part, double* sums, int G, int nchunks, hipStream_t st) {
    hipLaunchKernelGGL(bn_sum_chunks_kernel, dim3(step::ceil_div(2 * G, 64)), dim3(256), 0, st, part, sums, G, nchunks);
    return step::check_launch("bn_sum_partials");
}

extern "C" int step_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, float* scale, float* shift,
                                    int G, void* stream) {
    STEP_REQUIRE(running_mean && running_var && scale && shift && G > 0, "step_bn_eval_prepare: bad args");
    hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(step::ceil_div(G, 128)), dim3(128), 0,
                       (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, scale, shift, G);
    return step::check_launch("step_bn_eval_prepare");
}

#define BN_SHAPE_CHECK(name)                                                                  \
    STEP_REQUIRE(N > 0 && C > 0 && T > 0 && V == STEP_V, name ": bad shape N=%d C=%d T=%d V=%d (V must be 16)", \
                 N, C, T, V);                                                                 \
    STEP_REQUIRE(bn_mode == 0 || bn_mode == 1, name ": bn_mode must be 0 or 1")

extern "C" int step_bn_train_stats(const float* x, int N, int C, int T, int V, int bn_mode, double* sums,
                                   void* ws, size_t wsb, void* stream) {
    BN_SHAPE_CHECK("step_bn_train_stats");
    STEP_REQUIRE(x && sums && ws, "step_bn_train_stats: null pointer");
    const int G = bn_mode ? 16 * C : C;
    if (wsb < ws_bytes(G)) {
        step::set_error("step_bn_train_stats: workspace %zu < %zu bytes", wsb, ws_bytes(G));
        return STEP_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = chunks_for(N);
    double* part = (double*)ws;
    hipLaunchKernelGGL((bn_reduce_kernel<false, 0>), dim3(C, nch), dim3(BN_THREADS), 0, st, x, nullptr,
                       nullptr, nullptr, nullptr, part, N, C, T, V, bn_mode, nch);
    hipLaunchKernelGGL(bn_sum_chunks_kernel, dim3(step::ceil_div(2 * G, 64)), dim3(256), 0, st, part,
                       sums, G, nch);
    return step::check_launch("step_bn_train_stats");
}

extern "C" int step_bn_train_finalize(const double* sums, double count, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps,
                                      float* mean, float* invstd, float* scale, float* shift, int G,
                                      void* stream) {
    STEP_REQUIRE(sums && mean && invstd && scale && shift && G > 0 && (count >= 1.0 || count == 0.0),
                 "step_bn_train_finalize: bad args (G=%d count=%g)", G, count);
    STEP_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
                 "step_bn_train_finalize: running stats come in pairs");
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(step::ceil_div(G, 128)), dim3(128), 0, (hipStream_t)stream,
                       sums, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, G,
                       count);
    return step::check_launch("step_bn_train_finalize");
}

extern "C" int step_bn_train_prepare(const float* x, int N, int C, int T, int V, int bn_mode,
                                     const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float momentum, float eps, float* mean,
                                     float* invstd, float* scale, float* shift, void* ws,
                                     size_t wsb, void* stream) {
    BN_SHAPE_CHECK("step_bn_train_prepare");
    STEP_REQUIRE(x && mean && invstd && scale && shift && ws, "step_bn_train_prepare: null pointer");
    const int G = bn_mode ? 16 * C : C;
    if (wsb < ws_bytes(G)) {
        step::set_error("step_bn_train_prepare: workspace %zu < %zu bytes", wsb, ws_bytes(G));
        return STEP_ERR_WORKSPACE;
    }
    double* tot = (double*)ws + (size_t)BN_MAX_CHUNKS * G * 2;
    const int rc = step_bn_train_stats(x, N, C, T, V, bn_mode, tot, ws, wsb, stream);
    if (rc != STEP_OK) return rc;
    const double count = bn_mode ? (double)N * T : (double)N * T * 16;
    return step_bn_train_finalize(tot, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd,
                                  scale, shift, G, stream);
}

// ---- the entry point that allocates nothing (no counterpart in the reference) ----------------------------
// Same launches as step_bn_train_prepare with the workspace in a file-scope device array.  In host code the
// array's NAME is only a stand-in: its device address is fetched once per device with hipGetSymbolAddress
// (each GPU holds its own copy at its own address) and every launch below takes that pointer.
constexpr int BN_STATIC_MAX_G = STEP_BN_STATIC_MAX_GROUPS;
constexpr size_t BN_STATIC_BYTES =
    ((size_t)BN_MAX_CHUNKS * BN_STATIC_MAX_G * 2 + 2 * (size_t)BN_STATIC_MAX_G) * 8 + 3 * (size_t)BN_STATIC_MAX_G * 4;
static __device__ __align__(256) unsigned char scratch[BN_STATIC_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[MAX_DEVICES] = {};

// device address of `scratch` on the current device (looked up on the first call per device)
int static_scratch(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
        step::set_error("step_bn_train_prepare_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
        return STEP_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            step::set_error("step_bn_train_prepare_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return STEP_ERR_LAUNCH;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return STEP_OK;
}
}  // namespace

extern "C" int step_bn_train_prepare_static(const float* x, int N, int C, int T, int V, int bn_mode,
                                            const float* gamma, const float* beta, float* running_mean,
                                            float* running_var, float momentum, float eps, float* mean,
                                            float* invstd, float* scale, float* shift, void* stream) {
    BN_SHAPE_CHECK("step_bn_train_prepare_static");
    const long long G = bn_mode ? 16LL * C : (long long)C;
    STEP_REQUIRE(G <= BN_STATIC_MAX_G,
                 "step_bn_train_prepare_static: %lld statistic groups exceed the static workspace's maximum of %d "
                 "(use step_bn_train_prepare with a caller workspace)", G, BN_STATIC_MAX_G);
    void* ws = nullptr;
    const int rc = static_scratch(&ws);
    if (rc != STEP_OK) return rc;
    // every partial, total and coefficient the kernels read is written earlier in the same call: nothing in
    // the array has to start at zero, so no memset is issued
    return step_bn_train_prepare(x, N, C, T, V, bn_mode, gamma, beta, running_mean, running_var, momentum,
                                 eps, mean, invstd, scale, shift, ws, ws_bytes((int)G), stream);
}

extern "C" int step_affine_act(const float* x, const float* scale, const float* shift,
                               const float* res, const float* rscale, const float* rshift, int relu,
                               float* out, int N, int C, int T, int V, int bn_mode, void* stream) {
    BN_SHAPE_CHECK("step_affine_act");
    STEP_REQUIRE(x && out, "step_affine_act: null pointer");
    STEP_REQUIRE((scale == nullptr) == (shift == nullptr) && (rscale == nullptr) == (rshift == nullptr),
                 "step_affine_act: scale/shift must come in pairs");
    const size_t n4 = (size_t)N * C * T * 4;
    hipLaunchKernelGGL(affine_act_kernel, dim3(elem_grid(n4)), dim3(256), 0, (hipStream_t)stream, x,
                       scale, shift, res, rscale, rshift, relu, out, n4, C, T, bn_mode);
    return step::check_launch("step_affine_act");
}

extern "C" int step_bn_bwd_reduce(const float* dy, const float* x, const float* mask_src,
                                  const float* scale, const float* shift, int mask_mode, int N, int C,
                                  int T, int V, int bn_mode, double* sums, void* ws, size_t wsb,
                                  void* stream) {
    BN_SHAPE_CHECK("step_bn_bwd_reduce");
    STEP_REQUIRE(dy && x && sums && ws, "step_bn_bwd_reduce: null pointer");
    STEP_REQUIRE(mask_mode >= 0 && mask_mode <= 2, "step_bn_bwd_reduce: mask_mode %d", mask_mode);
    STEP_REQUIRE(mask_mode != 1 || mask_src, "step_bn_bwd_reduce: mask_mode 1 needs mask_src");
    STEP_REQUIRE(mask_mode != 2 || (scale && shift), "step_bn_bwd_reduce: mask_mode 2 needs scale/shift");
    const int G = bn_mode ? 16 * C : C;
    if (wsb < ws_bytes(G)) {
        step::set_error("step_bn_bwd_reduce: workspace %zu < %zu bytes", wsb, ws_bytes(G));
        return STEP_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = chunks_for(N);
    double* part = (double*)ws;
    dim3 grid(C, nch), blk(BN_THREADS);
    if (mask_mode == 0)
        hipLaunchKernelGGL((bn_reduce_kernel<true, 0>), grid, blk, 0, st, x, dy, mask_src, scale, shift,
                           part, N, C, T, V, bn_mode, nch);
    else if (mask_mode == 1)
        hipLaunchKernelGGL((bn_reduce_kernel<true, 1>), grid, blk, 0, st, x, dy, mask_src, scale, shift,
                           part, N, C, T, V, bn_mode, nch);
    else
        hipLaunchKernelGGL((bn_reduce_kernel<true, 2>), grid, blk, 0, st, x, dy, mask_src, scale, shift,
                           part, N, C, T, V, bn_mode, nch);
    hipLaunchKernelGGL(bn_sum_chunks_kernel, dim3(step::ceil_div(2 * G, 64)), dim3(256), 0, st, part,
                       sums, G, nch);
    return step::check_launch("step_bn_bwd_reduce");
}

extern "C" int step_bn_bwd_apply(const float* dy, const float* x, const float* mask_src,
                                 const float* scale, const float* shift, int mask_mode,
                                 const float* mean, const float* invstd, const float* gamma,
                                 const double* sums, int training, int accumulate, float* dx,
                                 float* dgamma, float* dbeta, int N, int C, int T, int V, int bn_mode,
                                 void* ws, size_t wsb, void* stream) {
    BN_SHAPE_CHECK("step_bn_bwd_apply");
    STEP_REQUIRE(dy && x && mean && invstd && sums && ws, "step_bn_bwd_apply: null pointer");
    STEP_REQUIRE(mask_mode >= 0 && mask_mode <= 2, "step_bn_bwd_apply: mask_mode %d", mask_mode);
    STEP_REQUIRE(mask_mode != 1 || mask_src, "step_bn_bwd_apply: mask_mode 1 needs mask_src");
    STEP_REQUIRE(mask_mode != 2 || (scale && shift), "step_bn_bwd_apply: mask_mode 2 needs scale/shift");
    const int G = bn_mode ? 16 * C : C;
    if (wsb < ws_bytes(G)) {
        step::set_error("step_bn_bwd_apply: workspace %zu < %zu bytes", wsb, ws_bytes(G));
        return STEP_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* coef = (float*)((double*)ws + (size_t)BN_MAX_CHUNKS * G * 2 + 2 * (size_t)G);
    const double count = bn_mode ? (double)N * T : (double)N * T * 16;
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(step::ceil_div(G, 128)), dim3(128), 0, st, sums, mean,
                       invstd, gamma, training, count, coef, dgamma, dbeta, G);
    if (dx) {
        const size_t n4 = (size_t)N * C * T * 4;
        dim3 grid(elem_grid(n4)), blk(256);
        if (mask_mode == 0)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<0>), grid, blk, 0, st, dy, x, mask_src, scale, shift,
                               coef, accumulate, dx, n4, C, T, bn_mode);
        else if (mask_mode == 1)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), grid, blk, 0, st, dy, x, mask_src, scale, shift,
                               coef, accumulate, dx, n4, C, T, bn_mode);
        else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<2>), grid, blk, 0, st, dy, x, mask_src, scale, shift,
                               coef, accumulate, dx, n4, C, T, bn_mode);
    }
    return step::check_launch("step_bn_bwd_apply");
}

extern "C" int step_bn_bwd_tail_pair(const float* dout, const float* out, const float* u, const float* r,
                                     const float* mean_u, const float* invstd_u, const float* gamma_u,
                                     const float* mean_r, const float* invstd_r, const float* gamma_r,
                                     int training, float* du, float* dr, float* dgamma_u, float* dbeta_u,
                                     float* dgamma_r, float* dbeta_r, int N, int C, int T, int V, void* ws,
                                     size_t wsb, void* stream) {
    const int bn_mode = 0;
    BN_SHAPE_CHECK("step_bn_bwd_tail_pair");
    STEP_REQUIRE(dout && out && u && r && mean_u && invstd_u && mean_r && invstd_r && du && dr && ws,
                 "step_bn_bwd_tail_pair: null pointer");
    const size_t one = ws_bytes(C);
    if (wsb < 2 * one) {
        step::set_error("step_bn_bwd_tail_pair: workspace %zu < %zu bytes", wsb, 2 * one);
        return STEP_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = chunks_for(N);
    const double count = (double)N * T * 16;
    char* wsu = (char*)ws;
    char* wsr = wsu + one;
    auto part = [](char* w) { return (double*)w; };
    auto tot = [&](char* w) { return (double*)w + (size_t)BN_MAX_CHUNKS * C * 2; };
    auto coef = [&](char* w) { return (float*)((double*)w + (size_t)BN_MAX_CHUNKS * C * 2 + 2 * (size_t)C); };
    hipLaunchKernelGGL(bn_reduce_pair_kernel, dim3(C, nch), dim3(BN_THREADS), 0, st, dout, out, u, r, part(wsu),
                       part(wsr), N, C, T, nch);
    hipLaunchKernelGGL(bn_sum_chunks_kernel, dim3(step::ceil_div(2 * C, 64)), dim3(256), 0, st, part(wsu), tot(wsu), C, nch);
    hipLaunchKernelGGL(bn_sum_chunks_kernel, dim3(step::ceil_div(2 * C, 64)), dim3(256), 0, st, part(wsr), tot(wsr), C, nch);
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(step::ceil_div(C, 128)), dim3(128), 0, st, tot(wsu), mean_u, invstd_u,
                       gamma_u, training, count, coef(wsu), dgamma_u, dbeta_u, C);
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(step::ceil_div(C, 128)), dim3(128), 0, st, tot(wsr), mean_r, invstd_r,
                       gamma_r, training, count, coef(wsr), dgamma_r, dbeta_r, C);
    const size_t n4 = (size_t)N * C * T * 4;
    hipLaunchKernelGGL(bn_bwd_apply_pair_kernel, dim3(elem_grid(n4)), dim3(256), 0, st, dout, out, u, r, coef(wsu),
                       coef(wsr), du, dr, n4, C, T);
    return step::check_launch("step_bn_bwd_tail_pair");
}

extern "C" int step_masked_add(const float* dy, const float* mask_src, float* dx, int accumulate,
                               size_t n, void* stream) {
    STEP_REQUIRE(dy && mask_src && dx && n > 0 && (n % 4) == 0, "step_masked_add: bad args");
    hipLaunchKernelGGL(masked_add_kernel, dim3(elem_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, dy,
                       mask_src, dx, accumulate, n / 4);
    return step::check_launch("step_masked_add");
}
