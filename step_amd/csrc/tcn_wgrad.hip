// Weight gradient of the temporal (Kt x 1) convolution as an fp32 MFMA GEMM.
//
// Replaces the `convolution_backward` (weight part) that autograd runs for tcn[2]
// (classifier_stgcn_real_only/net/classifier.py:156-162, CVAE_stgcn.py:359-365):
//   G[co][ci][j] = SUM_{n,t,v} dout[n,co,t,v] * a[n,ci,s*t + j - pad, v],  a = relu(bn1(y))
// recomputed from y on load (the activated tensor is never stored).
//
// GEMM view: M = co, N = q = ci*Kt + j (flattened, so 32-wide column tiles are always full),
// K = (n, t, v).  v_mfma_f32_32x32x2_f32: a k-step is the joint pair (v, v+1) of one frame.
// A operand: dout tile [co][pos] in LDS with an odd row stride.  B operand: the activation
// strip transposed in LDS to [v][channel][row] so that the 32 lanes of a column tile (taps
// j, j+1, ...) read consecutive rows; the channel stride is == Kt (mod 32) so a tile that
// straddles two channels is still conflict-free.
// A block owns a q range x all co for a slice of the batch and keeps its accumulators in
// registers over the whole slice; slices are combined by a second, ordered pass
// (bit-reproducible, no float atomics).
#include "common.h"

namespace {

constexpr int WG_TT = 8;                    // output frames per staged chunk
constexpr int WG_POS = WG_TT * 16;          // 128 positions
constexpr int WG_DS = WG_POS + 1;           // odd row stride of the dout tile

struct WgParams {
    const float* dout;
    const float* in;
    const float* in_scale;
    const float* in_shift;
    float* slab;        // [nsplit][Co][Q]
    float* dbslab;      // [nsplit][Co]
    int in_relu;
    int N, Ci, Co, Tin, Tout, Kt, stride, pad;
    int Q, nqb, nsplit, RW, RS, VS, NCH;
};

template <int MW, int NQ>
__global__ __launch_bounds__(256) void tcn_wgrad_kernel(WgParams p) {
    constexpr int QW = 4 / MW;                       // column groups (waves) per block
    constexpr int QB = QW * NQ * 32;                 // q's per block
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ds = smem;                                // [32*MW][WG_DS]
    float* aT = smem + 32 * MW * WG_DS;              // [16][NCH][RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int mw = wave / QW;
    const int qw = wave % QW;
    const int l31 = lane & 31;
    const int half = lane >> 5;

    const int qb = blockIdx.x % p.nqb;
    const int split = blockIdx.x / p.nqb;
    const int q0 = qb * QB;
    const int ch0 = q0 / p.Kt;                       // first channel this block touches
    const int n0 = (int)((long long)p.N * split / p.nsplit);
    const int n1 = (int)((long long)p.N * (split + 1) / p.nsplit);

    // per-lane B base offsets (channel slot * RS + tap) of its column in each tile
    int cbase[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int q = q0 + (qw * NQ + i) * 32 + l31;
        if (q >= p.Q) q = p.Q - 1;                   // clamp: result discarded at the store
        const int ci = q / p.Kt;
        cbase[i] = (ci - ch0) * p.RS + (q - ci * p.Kt) + half * p.VS;
    }

    f32x16 acc[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float dbsum = 0.f;

    const int nchunks = (p.Tout + WG_TT - 1) / WG_TT;
    for (int n = n0; n < n1; ++n) {
        for (int ck = 0; ck < nchunks; ++ck) {
            const int t0 = ck * WG_TT;
            __syncthreads();
            // dout tile: 32*MW channels x 128 positions
            for (int f = tid; f < 32 * MW * 32; f += 256) {
                const int co = f >> 5;
                const int r4 = f & 31;                         // float4 index within 128 positions
                const int t = t0 + (r4 >> 2);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (co < p.Co && t < p.Tout)
                    v = *reinterpret_cast<const float4*>(
                        p.dout + (((size_t)n * p.Co + co) * p.Tout + t) * 16 + (r4 & 3) * 4);
                float* d = ds + co * WG_DS + r4 * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
            // activation strip, transposed: NCH channels x RW rows x 16 joints
            const int row0 = p.stride * t0 - p.pad;
            for (int f = tid; f < p.NCH * p.RW * 4; f += 256) {
                const int cil = f / (p.RW * 4);
                const int rem = f - cil * p.RW * 4;
                const int row = rem >> 2, vq = rem & 3;
                const int ci = ch0 + cil;
                const int tin = row0 + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci < p.Ci && tin >= 0 && tin < p.Tin) {
                    v = *reinterpret_cast<const float4*>(
                        p.in + (((size_t)n * p.Ci + ci) * p.Tin + tin) * 16 + vq * 4);
                    if (p.in_scale) {
                        const float sc = p.in_scale[ci], sh = p.in_shift[ci];
                        v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
                    }
                    if (p.in_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
                float* d = aT + (vq * 4) * p.VS + cil * p.RS + row;
                d[0] = v.x; d[p.VS] = v.y; d[2 * p.VS] = v.z; d[3 * p.VS] = v.w;
            }
            __syncthreads();

            const float* ap = ds + (mw * 32 + l31) * WG_DS + half;
            for (int tl = 0; tl < WG_TT; ++tl) {
                const float* bp = aT + p.stride * tl;
#pragma unroll
                for (int v2 = 0; v2 < 8; ++v2) {
                    const float a = ap[tl * 16 + v2 * 2];
                    dbsum += a;
                    const float* bq = bp + (2 * v2) * p.VS;
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[cbase[i]], acc[i], 0, 0, 0);
                }
            }
        }
    }

    // partial results of this batch slice
    float* slab = p.slab + (size_t)split * p.Co * p.Q;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = q0 + (qw * NQ + i) * 32 + l31;
        if (q < p.Q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mw * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < p.Co) slab[(size_t)co * p.Q + q] = acc[i][r];
            }
        }
    }
    if (qb == 0 && qw == 0 && p.dbslab) {
        const float other = __shfl_down(dbsum, 32, 64);          // joint-parity halves
        const int co = mw * 32 + l31;
        if (half == 0 && co < p.Co) p.dbslab[(size_t)split * p.Co + co] = dbsum + other;
    }
}

// ordered sum over the batch slices; writes dw in (Co,Ci,Kt) or, transposed, (Ci,Co,Kt) with
// the tap order reversed (ConvTranspose2d weight layout, see step_tcn_pack_weight mode 1)
__global__ void tcn_wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ dbslab,
                                        float* __restrict__ dw, float* __restrict__ db, int Co, int Ci,
                                        int Kt, int nsplit, int transposed) {
    const int Q = Ci * Kt;
    const int total = Co * Q;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += slab[(size_t)k * total + e];
        if (!transposed) {
            dw[e] = s;
        } else {
            const int co = e / Q, q = e % Q;
            const int ci = q / Kt, j = q % Kt;
            dw[((size_t)ci * Co + co) * Kt + (Kt - 1 - j)] = s;
        }
    } else if (e < total + Co && db) {
        const int co = e - total;
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += dbslab[(size_t)k * Co + co];
        db[co] = s;
    }
}

struct WgGeom {
    int mw, nq, qb, nqb, nsplit, RW, RS, VS, NCH;
    size_t lds, ws;
};

bool wg_geom(int N, int Ci, int Co, int Kt, int stride, WgGeom* g) {
    if (Co > 64 || Co < 1 || Ci < 1 || Kt < 1 || N < 1) return false;
    g->mw = Co > 32 ? 2 : 1;
    g->nq = g->mw == 2 ? 5 : 4;
    g->qb = (4 / g->mw) * g->nq * 32;
    const int Q = Ci * Kt;
    g->nqb = step::ceil_div(Q, g->qb);
    int ns = step::ceil_div(768, g->nqb);
    g->nsplit = ns < N ? ns : N;
    g->RW = stride * (WG_TT - 1) + Kt;
    int rs = g->RW;
    while ((rs & 31) != (Kt & 31)) ++rs;               // channel stride == Kt (mod 32)
    g->RS = rs;
    g->NCH = (g->qb + Kt - 1) / Kt + 1;
    if (g->NCH > Ci) g->NCH = Ci;                      // a block never touches more than Ci channels
    g->VS = g->NCH * g->RS + 1;
    g->lds = ((size_t)32 * g->mw * WG_DS + (size_t)16 * g->VS) * 4;
    g->ws = ((size_t)g->nsplit * Co * Q + (size_t)g->nsplit * Co) * 4;
    return g->lds <= 160 * 1024;
}

template <int MW, int NQ>
int launch_wg(const WgParams& p, const WgGeom& g, hipStream_t st) {
    auto kern = tcn_wgrad_kernel<MW, NQ>;
    if (g.lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) {
            step::set_error("step_tcn_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return STEP_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(kern, dim3(g.nqb * g.nsplit), dim3(256), g.lds, st, p);
    return step::check_launch("step_tcn_wgrad");
}

}  // namespace

extern "C" size_t step_tcn_wgrad_ws_bytes(int N, int Ci, int Co, int Tin, int Tout, int Kt) {
    WgGeom g;
    (void)Tin; (void)Tout;
    if (!wg_geom(N, Ci, Co, Kt, 1, &g)) return 0;
    return g.ws;                                        // independent of the stride
}

extern "C" int step_tcn_wgrad(const float* dout, const float* in, const float* in_scale,
                              const float* in_shift, int in_relu, float* dw, float* db, int N, int Ci,
                              int Co, int Tin, int Tout, int Kt, int stride, int transposed, void* ws,
                              size_t ws_bytes, void* stream) {
    STEP_REQUIRE(dout && in && dw && ws, "step_tcn_wgrad: null pointer");
    STEP_REQUIRE(N > 0 && Tin > 0 && Tout == (Tin - 1) / stride + 1 && (stride == 1 || stride == 2),
                 "step_tcn_wgrad: bad shape Tin=%d Tout=%d stride=%d", Tin, Tout, stride);
    STEP_REQUIRE((Kt & 1) == 1, "step_tcn_wgrad: Kt=%d must be odd", Kt);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "step_tcn_wgrad: scale/shift pair");
    STEP_REQUIRE(!transposed || stride == 1, "step_tcn_wgrad: transposed needs stride 1");
    WgGeom g;
    STEP_REQUIRE(wg_geom(N, Ci, Co, Kt, stride, &g), "step_tcn_wgrad: unsupported Ci=%d Co=%d Kt=%d", Ci,
                 Co, Kt);
    if (ws_bytes < g.ws) {
        step::set_error("step_tcn_wgrad: workspace %zu < %zu bytes", ws_bytes, g.ws);
        return STEP_ERR_WORKSPACE;
    }
    WgParams p{};
    p.dout = dout; p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.in_relu = in_relu;
    p.slab = (float*)ws;
    p.dbslab = p.slab + (size_t)g.nsplit * Co * Ci * Kt;
    p.N = N; p.Ci = Ci; p.Co = Co; p.Tin = Tin; p.Tout = Tout; p.Kt = Kt; p.stride = stride;
    p.pad = (Kt - 1) / 2;
    p.Q = Ci * Kt; p.nqb = g.nqb; p.nsplit = g.nsplit; p.RW = g.RW; p.RS = g.RS; p.VS = g.VS;
    p.NCH = g.NCH;
    hipStream_t st = (hipStream_t)stream;
    int rc = g.mw == 2 ? launch_wg<2, 5>(p, g, st) : launch_wg<1, 4>(p, g, st);
    if (rc) return rc;
    const int total = Co * Ci * Kt + Co;
    hipLaunchKernelGGL(tcn_wgrad_reduce_kernel, dim3(step::ceil_div(total, 256)), dim3(256), 0, st,
                       p.slab, p.dbslab, dw, db, Co, Ci, Kt, g.nsplit, transposed);
    return step::check_launch("step_tcn_wgrad(reduce)");
}
