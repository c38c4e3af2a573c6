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
#include <cstdlib>

namespace {

constexpr int WG_TT = 8;                    // output frames per staged chunk
constexpr int WG_POS = WG_TT * 16;          // 128 positions
constexpr int WG_DS = WG_POS + 4;           // dout tile row stride: 16-B aligned rows (float4 writes),
                                            // 4-way conflict on the (rare) A reads
constexpr int WG_VS = 18;                   // floats per staged activation frame: 16 joints + 2 pad
                                            // (8-B aligned rows, 2-way conflict on B reads)

struct WgParams {
    const float* dout;
    const float* in;
    const float* in_scale;
    const float* in_shift;
    float* slab;        // [nsplit][Co][Q]
    float* dbslab;      // [nsplit][Co]
    int in_relu;
    int N, Ci, Co, Tin, Tout, pad;
    int Q, nqb, nsplit;
};

// compile-time geometry of one (MW, NQ, Kt, stride) variant
template <int MW, int NQ, int KT, int STRIDE>
struct WgCfg {
    static constexpr int QW = 4 / MW;                         // column groups (waves) per block
    static constexpr int QB = QW * NQ * 32;                   // q's per block
    static constexpr int RW = STRIDE * (WG_TT - 1) + KT;      // staged input rows per chunk
    // channel stride of the strip image [channel][frame][WG_VS]: >= RW frames and
    // == KT*WG_VS (mod 32) so a column tile straddling two channels keeps its bank sequence
    static constexpr int RS = RW * WG_VS + (((KT - RW) * WG_VS) % 32 + 32) % 32;
    static constexpr int NCH0 = (QB + KT - 1) / KT + 1;
    static constexpr int NCH = NCH0 < 32 * MW ? NCH0 : 32 * MW;  // never more than Ci channels
    static constexpr int A_ITEMS = NCH * RW * 4;              // float4 items of the strip
    static constexpr int D_ITEMS = 32 * MW * 32;
    static constexpr int LDS_FLOATS = (32 * MW * WG_DS + NCH * RS + 3) / 4 * 4;
    static_assert(RS % 2 == 0, "8-byte aligned channel stride");
};

// Software-pipelined variant (DB = true, used when two LDS images fit): while the 320 MFMAs of
// chunk i run from LDS image i&1, the same waves fetch chunk i+1 in 7 small batches (global ->
// registers at frame step tl, registers -> the other LDS image at step tl+1), so the staging
// latency hides under the matrix pipe and one barrier per chunk remains.
template <int MW, int NQ, int KT, int STRIDE, bool DB>
__global__ __launch_bounds__(256) void tcn_wgrad_kernel(WgParams p) {
    using C = WgCfg<MW, NQ, KT, STRIDE>;
    constexpr int DN = C::D_ITEMS / 256;                         // dout float4 per thread (4 or 8)
    constexpr int AN = (C::A_ITEMS + 255) / 256;                 // strip float4 per thread
    constexpr int NSTEP = 7;                                     // batches per chunk
    constexpr int DPS = (DN + NSTEP - 1) / NSTEP;                // items per batch
    constexpr int APS = (AN + NSTEP - 1) / NSTEP;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int mw = wave / C::QW;
    const int qw = wave % C::QW;
    const int l31 = lane & 31;
    const int half = lane >> 5;

    const int qb = blockIdx.x % p.nqb;
    const int split = blockIdx.x / p.nqb;
    const int q0 = qb * C::QB;
    const int ch0 = q0 / KT;                         // first channel this block touches
    const int n0 = (int)((long long)p.N * split / p.nsplit);
    const int n1 = (int)((long long)p.N * (split + 1) / p.nsplit);

    // per-lane B offset (floats, inside the aT image) of its column in each tile
    int bcol[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        int q = q0 + (qw * NQ + i) * 32 + l31;
        if (q >= p.Q) q = p.Q - 1;                   // clamp: result discarded at the store
        const int ci = q / KT;
        bcol[i] = 32 * MW * WG_DS + (ci - ch0) * C::RS + (q - ci * KT) * WG_VS + half;
    }
    const int aoff = (mw * 32 + l31) * WG_DS + half;

    f32x16 acc[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float dbsum = 0.f;

    const int nchunks = (p.Tout + WG_TT - 1) / WG_TT;
    const int total = (n1 - n0) * nchunks;

    float4 dv[DPS], av[APS];
    // issue the global loads of batch `step` of chunk (n, t0)
    auto batch_load = [&](int step, int n, int t0) {
        const float* dbase = p.dout + (size_t)n * p.Co * p.Tout * 16;          // wave-uniform
        const float* abase = p.in + ((size_t)n * p.Ci + ch0) * p.Tin * 16;
        const int row0 = STRIDE * t0 - p.pad;
#pragma unroll
        for (int k = 0; k < DPS; ++k) {
            const int item = step * DPS + k;
            dv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item < DN) {
                const int f = tid + item * 256;
                const int co = f >> 5, r4 = f & 31;
                const int t = t0 + (r4 >> 2);
                if (co < p.Co && t < p.Tout)
                    dv[k] = *reinterpret_cast<const float4*>(dbase + (co * p.Tout + t) * 16 + (r4 & 3) * 4);
            }
        }
#pragma unroll
        for (int k = 0; k < APS; ++k) {
            const int item = step * APS + k;
            av[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item < AN) {
                const int f = tid + item * 256;
                const int cil = f / (C::RW * 4);
                const int rem = f - cil * (C::RW * 4);
                const int tin = row0 + (rem >> 2);
                if (f < C::A_ITEMS && ch0 + cil < p.Ci && tin >= 0 && tin < p.Tin)
                    av[k] = *reinterpret_cast<const float4*>(abase + (cil * p.Tin + tin) * 16 + (rem & 3) * 4);
            }
        }
    };
    // write batch `step` into the LDS image `img` (BN1 + ReLU applied to the strip here)
    auto batch_store = [&](int step, float* img, int t0) {
        const int row0 = STRIDE * t0 - p.pad;
#pragma unroll
        for (int k = 0; k < DPS; ++k) {
            const int item = step * DPS + k;
            if (item < DN) {
                const int f = tid + item * 256;
                *reinterpret_cast<float4*>(img + (f >> 5) * WG_DS + (f & 31) * 4) = dv[k];
            }
        }
#pragma unroll
        for (int k = 0; k < APS; ++k) {
            const int item = step * APS + k;
            const int f = tid + item * 256;
            if (item < AN && f < C::A_ITEMS) {
                const int cil = f / (C::RW * 4);
                const int rem = f - cil * (C::RW * 4);
                const int row = rem >> 2, vq = rem & 3;
                const int tin = row0 + row;
                float4 v = av[k];
                if (ch0 + cil < p.Ci && tin >= 0 && tin < p.Tin) {
                    if (p.in_scale) {
                        const float sc = p.in_scale[ch0 + cil], sh = p.in_shift[ch0 + cil];
                        v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
                    }
                    if (p.in_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
                float2* d = reinterpret_cast<float2*>(img + 32 * MW * WG_DS + cil * C::RS + row * WG_VS + vq * 4);
                d[0] = make_float2(v.x, v.y);
                d[1] = make_float2(v.z, v.w);
            }
        }
    };
    auto mfma_frame = [&](const float* img, int tl) {
#pragma unroll
        for (int v2 = 0; v2 < 8; ++v2) {
            const float a = img[aoff + tl * 16 + v2 * 2];
            dbsum += a;
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    a, img[bcol[i] + STRIDE * tl * WG_VS + 2 * v2], acc[i], 0, 0, 0);
        }
    };

    if (DB) {
        if (total > 0) {
#pragma unroll 1
            for (int s = 0; s < NSTEP; ++s) {         // prologue: chunk 0, not overlapped
                batch_load(s, n0, 0);
                batch_store(s, smem, 0);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < total; ++it) {
            const float* cur = smem + (it & 1) * C::LDS_FLOATS;
            float* nxt = smem + ((it + 1) & 1) * C::LDS_FLOATS;
            const bool more = it + 1 < total;
            const int nn = n0 + (it + 1) / nchunks;
            const int nt0 = ((it + 1) % nchunks) * WG_TT;
#pragma unroll
            for (int tl = 0; tl < WG_TT; ++tl) {
                if (more) {
                    if (tl >= 1) batch_store(tl - 1, nxt, nt0);
                    if (tl < NSTEP) batch_load(tl, nn, nt0);
                }
                mfma_frame(cur, tl);
            }
            __syncthreads();
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < total; ++it) {
            const int n = n0 + it / nchunks;
            const int t0 = (it % nchunks) * WG_TT;
            __syncthreads();
#pragma unroll 1
            for (int s = 0; s < NSTEP; ++s) {
                batch_load(s, n, t0);
                batch_store(s, smem, t0);
            }
            __syncthreads();
#pragma unroll 2
            for (int tl = 0; tl < WG_TT; ++tl) mfma_frame(smem, tl);
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
    int mw, qb, nqb, nsplit;
    size_t lds, ws;
};

template <int MW, int NQ, int KT, int STRIDE>
void fill_geom(WgGeom* g) {
    using C = WgCfg<MW, NQ, KT, STRIDE>;
    g->qb = C::QB;
    g->lds = (size_t)C::LDS_FLOATS * 4 * (KT == 75 ? 2 : 1);   // Kt=75: two LDS images (pipelined)
}

bool wg_geom(int N, int Ci, int Co, int Kt, int stride, WgGeom* g) {
    if (Co > 64 || Co < 1 || Ci < 1 || N < 1 || (stride != 1 && stride != 2)) return false;
    g->mw = Co > 32 ? 2 : 1;
    if (Ci > 32 * g->mw) return false;               // the temporal conv of the stack is square
    const int key = Kt * 100 + stride * 10 + g->mw;
    switch (key) {
        case 7511: fill_geom<1, 4, 75, 1>(g); break;
        case 7512: fill_geom<2, 5, 75, 1>(g); break;
        case 7521: fill_geom<1, 4, 75, 2>(g); break;
        case 7522: fill_geom<2, 5, 75, 2>(g); break;
        case 911: fill_geom<1, 4, 9, 1>(g); break;
        case 912: fill_geom<2, 5, 9, 1>(g); break;
        case 921: fill_geom<1, 4, 9, 2>(g); break;
        case 922: fill_geom<2, 5, 9, 2>(g); break;
        default: return false;
    }
    const int Q = Ci * Kt;
    g->nqb = step::ceil_div(Q, g->qb);
    // batch slices: nqb * nsplit blocks should fill whole rounds of the 256 CUs (one pipelined
    // block per CU at Kt=75), or the last round runs mostly empty
    int ns = (256 * 3) / g->nqb;
    ns = ns < 1 ? 1 : ns;
    g->nsplit = ns < N ? ns : N;
    g->ws = ((size_t)g->nsplit * Co * Q + (size_t)g->nsplit * Co) * 4;
    return g->lds <= 160 * 1024;
}

template <int MW, int NQ, int KT, int STRIDE>
int launch_wg(const WgParams& p, const WgGeom& g, hipStream_t st) {
    auto kern = tcn_wgrad_kernel<MW, NQ, KT, STRIDE, KT == 75>;
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

int step::wgrad_reduce(const float* slab, const float* dbslab, float* dw, float* db, int Co, int Ci, int Kt,
                       int nsplit, int transposed, hipStream_t st, const char* what) {
    const int total = Co * Ci * Kt + Co;
    hipLaunchKernelGGL(tcn_wgrad_reduce_kernel, dim3(step::ceil_div(total, 256)), dim3(256), 0, st, slab,
                       dbslab, dw, db, Co, Ci, Kt, nsplit, transposed);
    return step::check_launch(what);
}

extern "C" size_t step_tcn_wgrad_ws_bytes(int N, int Ci, int Co, int Tin, int Tout, int Kt) {
    WgGeom g;
    (void)Tin; (void)Tout;
    // wide layers (Co > 64) exist in the sliding-window form only
    size_t need = wg_geom(N, Ci, Co, Kt, 1, &g) ? g.ws : 0;   // nsplit does not depend on the stride
    for (int s = 1; s <= 2; ++s) {                      // the sliding-window variant, where it applies
        const size_t sw = step::wgrad_sw_ws_bytes(N, Ci, Co, Kt, s);
        need = sw > need ? sw : need;
    }
    return need;
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
    if (!wg_geom(N, Ci, Co, Kt, stride, &g)) {
        // wide layers (Co > 64): the sliding-window form only
        const size_t sw = step::wgrad_sw_ws_bytes(N, Ci, Co, Kt, stride);
        STEP_REQUIRE(sw != 0, "step_tcn_wgrad: unsupported Ci=%d Co=%d Kt=%d", Ci, Co, Kt);
        if (ws_bytes < sw) {
            step::set_error("step_tcn_wgrad: workspace %zu < %zu bytes", ws_bytes, sw);
            return STEP_ERR_WORKSPACE;
        }
        const int rc_sw = step::wgrad_sw(dout, in, in_scale, in_shift, in_relu, dw, db, N, Ci, Co, Tin, Tout, Kt,
                                         stride, transposed, ws, ws_bytes, (hipStream_t)stream);
        STEP_REQUIRE(rc_sw != 1, "step_tcn_wgrad: unsupported Ci=%d Co=%d Kt=%d", Ci, Co, Kt);
        return rc_sw;
    }
    if (ws_bytes < g.ws) {
        step::set_error("step_tcn_wgrad: workspace %zu < %zu bytes", ws_bytes, g.ws);
        return STEP_ERR_WORKSPACE;
    }
    if (!getenv("STEP_WGRAD_OLD")) {
        const int rc_sw = step::wgrad_sw(dout, in, in_scale, in_shift, in_relu, dw, db, N, Ci, Co, Tin, Tout, Kt,
                                         stride, transposed, ws, ws_bytes, (hipStream_t)stream);
        if (rc_sw != 1) return rc_sw;
    }
    WgParams p{};
    p.dout = dout; p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.in_relu = in_relu;
    p.slab = (float*)ws;
    p.dbslab = p.slab + (size_t)g.nsplit * Co * Ci * Kt;
    p.N = N; p.Ci = Ci; p.Co = Co; p.Tin = Tin; p.Tout = Tout;
    p.pad = (Kt - 1) / 2;
    p.Q = Ci * Kt; p.nqb = g.nqb; p.nsplit = g.nsplit;
    hipStream_t st = (hipStream_t)stream;
    int rc = STEP_ERR_ARG;
    switch (Kt * 100 + stride * 10 + g.mw) {
        case 7511: rc = launch_wg<1, 4, 75, 1>(p, g, st); break;
        case 7512: rc = launch_wg<2, 5, 75, 1>(p, g, st); break;
        case 7521: rc = launch_wg<1, 4, 75, 2>(p, g, st); break;
        case 7522: rc = launch_wg<2, 5, 75, 2>(p, g, st); break;
        case 911: rc = launch_wg<1, 4, 9, 1>(p, g, st); break;
        case 912: rc = launch_wg<2, 5, 9, 1>(p, g, st); break;
        case 921: rc = launch_wg<1, 4, 9, 2>(p, g, st); break;
        case 922: rc = launch_wg<2, 5, 9, 2>(p, g, st); break;
    }
    if (rc) return rc;
    return step::wgrad_reduce(p.slab, p.dbslab, dw, db, Co, Ci, Kt, g.nsplit, transposed, st,
                              "step_tcn_wgrad(reduce)");
}
