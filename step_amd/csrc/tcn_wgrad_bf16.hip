// Weight gradient of the temporal (Kt x 1) convolution as a bf16 MFMA GEMM, fp32 tensors in HBM.
//
// Same operator as tcn_wgrad.hip (the weight part of autograd's convolution_backward for tcn[2],
// classifier_stgcn_real_only/net/classifier.py:156-162):
//   G[co][ci][j] = SUM_{n,t,v} dout[n,co,t,v] * a[n,ci,s*t + j - pad, v],  a = relu(bn1(y))
// with dout and a rounded to bf16 on their way into LDS and fp32 accumulation.
//
// GEMM view: M = co, N = (tap j, ci), K = (n, t, v).  One v_mfma_f32_32x32x16_bf16 contracts the
// 16 joints of ONE output frame t:
//   A[m][k]   = dout[co = m][t][v = k]                 (LDS image [co][frame][16], row read)
//   B[k][col] = a[ci = col][s*t + j - pad][v = k]      (LDS image [ci][frame][16], row read)
// Both operands are plain 16-byte row reads of images kept in the tensors' own order.  With the
// 32 lanes of a column tile on 32 input channels, the B fragment of (frame t, tap j) depends only
// on s*t + j: an MFMA wave owns NJ consecutive taps and slides a window of fragments along t, so
// each LDS read feeds MW*NJ MFMAs (s = 1).
//
// A block = 4 MFMA waves (4*NJ taps x 32 input channels x all output channels, accumulators in
// registers over the block's whole batch slice) + 4 staging waves that bring the next chunk of
// TT frames into the other LDS image (HBM -> registers -> BN + ReLU -> bf16 -> LDS) and keep the
// exact fp32 bias-gradient sums.  Slices are combined by the ordered reduce of tcn_wgrad.hip
// (bit-reproducible, no float atomics).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct WbParams {
    const float* dout;
    const float* in;
    const float* in_scale;
    const float* in_shift;
    float* slab;        // [nsplit][Co][Q]
    float* dbslab;      // [nsplit][Co]
    int in_relu;
    int N, Ci, Co, Tin, Tout, Kt, pad;
    int Q, ntb, nqb, nsplit;
};

template <int MW, int NJ, int STRIDE, int TT>
struct WbCfg {
    static constexpr int NCONS = 4, NPROD = 4;
    static constexpr int NTHR = 64 * (NCONS + NPROD), PTHR = 64 * NPROD;
    static constexpr int TAPS = NCONS * NJ;                       // taps per block
    static constexpr int AF = STRIDE * (TT - 1) + TAPS;           // activation frames per chunk
    // row strides (bytes): an odd number of 16-byte chunks, so the 16-byte row reads of 32
    // channels fall on distinct bank groups
    static constexpr int CHS = AF * 32 + 16;
    static constexpr int DS = TT * 32 + 16;
    static constexpr int D_BYTES = 32 * MW * DS;
    static constexpr int IMG = D_BYTES + 32 * CHS;
    static constexpr int LDS_BYTES = 2 * IMG;
    static constexpr int D_ITEMS = 32 * MW * TT * 4;              // float4 items of the dout tile
    static constexpr int A_ITEMS = 32 * AF * 4;
    static constexpr int DN = D_ITEMS / PTHR;
    static constexpr int AN = (A_ITEMS + PTHR - 1) / PTHR;
    static constexpr int NBT = STRIDE * (TT - 1) + NJ;            // B fragments of one chunk (per wave)
    static_assert(D_ITEMS % PTHR == 0, "dout items");
    static_assert(TT * 4 == 32 || TT * 4 == 64, "bias sums");
    static_assert(IMG % 16 == 0 && LDS_BYTES <= 160 * 1024, "LDS");
};

template <int MW, int NJ, int STRIDE, int TT>
__global__ __launch_bounds__(512) void tcn_wgrad_bf16_kernel(WbParams p) {
    using C = WbCfg<MW, NJ, STRIDE, TT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // blocks that share an XCD (blockIdx % 8) take the q-blocks of the same batch slices: they
    // stream the same dout / activation chunks at the same time, out of that XCD's L2
    int lb = blockIdx.x;
    if ((gridDim.x & 7) == 0) lb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int qb = lb % p.nqb;
    const int split = lb / p.nqb;
    const int tb = qb % p.ntb;                        // tap block
    const int ci0 = (qb / p.ntb) * 32;                // input-channel block
    const int j0 = tb * C::TAPS;
    const int n0 = (int)((long long)p.N * split / p.nsplit);
    const int n1 = (int)((long long)p.N * (split + 1) / p.nsplit);
    const int nchunks = (p.Tout + TT - 1) / TT;
    const int total = (n1 - n0) * nchunks;

    if (wave >= C::NCONS) {
        // ------------------------------------------------------------ staging waves
        const int ptid = tid - 64 * C::NCONS;
        // the MFMA wave on this SIMD always has an instruction ready and would win every
        // arbitration; the staging wave issues rarely (it mostly waits on memory) but must not queue
        __builtin_amdgcn_s_setprio(3);
        float4 dv[C::DN], av[C::AN];
        float dbacc[C::DN];
#pragma unroll
        for (int k = 0; k < C::DN; ++k) dbacc[k] = 0.f;

        auto load_regs = [&](int item) {
            const int n = n0 + item / nchunks;
            const int t0 = (item % nchunks) * TT;
            const float* dbase = p.dout + (size_t)n * p.Co * p.Tout * 16;
            const float* abase = p.in + ((size_t)n * p.Ci + ci0) * p.Tin * 16;
            const int row0 = STRIDE * t0 + j0 - p.pad;
            int pt = ptid;
            asm volatile("" : "+v"(pt));            // keep the per-item index math inside the loop
#pragma unroll
            for (int k = 0; k < C::DN; ++k) {
                const int f = pt + k * C::PTHR;
                const int co = f / (TT * 4), w = f % (TT * 4);
                const int t = t0 + (w >> 2);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (co < p.Co && t < p.Tout)
                    v = *reinterpret_cast<const float4*>(dbase + ((co * p.Tout + t) * 16 + (w & 3) * 4));
                dv[k] = v;
            }
#pragma unroll
            for (int k = 0; k < C::AN; ++k) {
                const int f = pt + k * C::PTHR;
                const int cil = f / (C::AF * 4);
                const int rem = f - cil * (C::AF * 4);
                const int tin = row0 + (rem >> 2);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < C::A_ITEMS && ci0 + cil < p.Ci && tin >= 0 && tin < p.Tin)
                    v = *reinterpret_cast<const float4*>(abase + ((cil * p.Tin + tin) * 16 + (rem & 3) * 4));
                av[k] = v;
            }
        };
        auto store_regs = [&](int item, char* img) {
            const int t0 = (item % nchunks) * TT;
            const int row0 = STRIDE * t0 + j0 - p.pad;
            int pt = ptid;
            asm volatile("" : "+v"(pt));
#pragma unroll
            for (int k = 0; k < C::DN; ++k) {
                const int f = pt + k * C::PTHR;
                const int co = f / (TT * 4), w = f % (TT * 4);
                const f32x4 v = {dv[k].x, dv[k].y, dv[k].z, dv[k].w};
                dbacc[k] += (v[0] + v[1]) + (v[2] + v[3]);
                *reinterpret_cast<bf16x4*>(img + co * C::DS + (w >> 2) * 32 + (w & 3) * 8) =
                    __builtin_convertvector(v, bf16x4);
            }
#pragma unroll
            for (int k = 0; k < C::AN; ++k) {
                const int f = pt + k * C::PTHR;
                if (f < C::A_ITEMS) {
                    const int cil = f / (C::AF * 4);
                    const int rem = f - cil * (C::AF * 4);
                    const int fr = rem >> 2;
                    const int tin = row0 + fr;
                    f32x4 v = {av[k].x, av[k].y, av[k].z, av[k].w};
                    if (ci0 + cil < p.Ci && tin >= 0 && tin < p.Tin) {
                        if (p.in_scale) {
                            const float sc = p.in_scale[ci0 + cil], sh = p.in_shift[ci0 + cil];
                            v = v * sc + sh;
                        }
                        if (p.in_relu) {
                            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                            v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                        }
                    }
                    *reinterpret_cast<bf16x4*>(img + C::D_BYTES + cil * C::CHS + fr * 32 + (rem & 3) * 8) =
                        __builtin_convertvector(v, bf16x4);
                }
            }
        };

        if (total > 0) { load_regs(0); store_regs(0, smem); }
        if (total > 1) load_regs(1);                   // in flight across the first barrier
        __syncthreads();
        for (int it = 0; it < total; ++it) {
            // the MFMA waves work on chunk `it`; chunk it+1 goes to the other image, and the loads
            // of chunk it+2 are issued before the barrier so they fly during chunk it+1
            if (it + 1 < total) store_regs(it + 1, smem + ((it + 1) & 1) * C::IMG);
            if (it + 2 < total) load_regs(it + 2);
            __syncthreads();
        }
        // bias gradient: exact fp32 sums of dout, one block per batch slice writes them
        if (qb == 0 && p.dbslab) {
            constexpr int W = TT * 4;                  // lanes that share an output channel (32 | 64)
#pragma unroll
            for (int k = 0; k < C::DN; ++k) {
                float s = dbacc[k];
#pragma unroll
                for (int o = 1; o < W; o <<= 1) s += __shfl_xor(s, o, 64);
                const int f = ptid + k * C::PTHR;
                const int co = f / W;
                if (f % W == 0 && co < p.Co) p.dbslab[(size_t)split * p.Co + co] = s;
            }
        }
        return;
    }

    // ---------------------------------------------------------------- MFMA waves
    const int l31 = lane & 31;
    const int half = lane >> 5;
    f32x16 acc[MW][NJ];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;

    const int aoff = l31 * C::DS + half * 16;                                   // + mt*32*DS + tl*32
    const int boff = C::D_BYTES + l31 * C::CHS + (wave * NJ) * 32 + half * 16;  // + q*32

    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const char* img = smem + (it & 1) * C::IMG;
        auto read_a = [&](int mt, int tl) -> bf16x8 {
            return *reinterpret_cast<const bf16x8*>(img + aoff + mt * 32 * C::DS + tl * 32);
        };
        auto read_b = [&](int q) -> bf16x8 {
            return *reinterpret_cast<const bf16x8*>(img + boff + q * 32);
        };
        bf16x8 a[TT][MW];
        bf16x8 b[C::NBT];
        constexpr int PF = 1;                        // frames of LDS prefetch
#pragma unroll
        for (int tl = 0; tl < PF; ++tl)
#pragma unroll
            for (int m = 0; m < MW; ++m) a[tl][m] = read_a(m, tl);
#pragma unroll
        for (int q = 0; q < STRIDE * PF + NJ - STRIDE && q < C::NBT; ++q) b[q] = read_b(q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tl = 0; tl < TT; ++tl) {
            if (tl + PF < TT) {
#pragma unroll
                for (int m = 0; m < MW; ++m) a[tl + PF][m] = read_a(m, tl + PF);
            }
#pragma unroll
            for (int s = 0; s < STRIDE; ++s) {
                const int q = STRIDE * (tl + PF) + NJ - STRIDE + s;
                if (q < C::NBT) b[q] = read_b(q);
            }
#pragma unroll
            for (int i = 0; i < NJ; ++i)
#pragma unroll
                for (int m = 0; m < MW; ++m)
                    acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tl][m], b[STRIDE * tl + i],
                                                                        acc[m][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // partial results of this batch slice: column = input channel (lane), row = output channel
    float* slab = p.slab + (size_t)split * p.Co * p.Q;
    const int ci = ci0 + l31;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = j0 + wave * NJ + i;
        if (j < p.Kt && ci < p.Ci) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < p.Co) slab[(size_t)co * p.Q + ci * p.Kt + j] = acc[m][i][r];
                }
        }
    }
}

struct WbGeom {
    int mw, taps, ntb, nqb, nsplit;
    size_t ws;
};

// variants: (Co <= 32: MW 1, NJ 10) (Co <= 64: MW 2, NJ 5); stride 2 only with MW 2
bool wb_geom(int N, int Ci, int Co, int Kt, int stride, WbGeom* g) {
    if (Kt != 75 || N < 1 || (Ci != 32 && Ci != 64) || Ci != Co) return false;
    g->mw = Co / 32;
    if (stride == 1) g->taps = (g->mw == 1) ? 40 : 20;
    else if (stride == 2 && g->mw == 2) g->taps = 20;
    else return false;
    g->ntb = step::ceil_div(Kt, g->taps);
    g->nqb = g->ntb * (Ci / 32);
    int ns = 256 / g->nqb;                     // one block per CU
    ns = ns < 1 ? 1 : ns;
    g->nsplit = ns < N ? ns : N;
    g->ws = ((size_t)g->nsplit * Co * Ci * Kt + (size_t)g->nsplit * Co) * 4;
    return true;
}

template <int MW, int NJ, int STRIDE, int TT>
int launch_wb(const WbParams& p, hipStream_t st) {
    using C = WbCfg<MW, NJ, STRIDE, TT>;
    auto kern = tcn_wgrad_bf16_kernel<MW, NJ, STRIDE, TT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
        step::set_error("step_tcn_wgrad_bf16: hipFuncSetAttribute(%d): %s", C::LDS_BYTES, hipGetErrorString(e));
        return STEP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(p.nqb * p.nsplit), dim3(C::NTHR), C::LDS_BYTES, st, p);
    return step::check_launch("step_tcn_wgrad_bf16");
}

}  // namespace

extern "C" size_t step_tcn_wgrad_bf16_ws_bytes(int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride) {
    WbGeom g;
    (void)Tin; (void)Tout;
    if (!wb_geom(N, Ci, Co, Kt, stride, &g)) return 0;
    return g.ws;
}

extern "C" int step_tcn_wgrad_bf16(const float* dout, const float* in, const float* in_scale,
                                   const float* in_shift, int in_relu, float* dw, float* db, int N, int Ci,
                                   int Co, int Tin, int Tout, int Kt, int stride, void* ws, size_t ws_bytes,
                                   void* stream) {
    STEP_REQUIRE(dout && in && dw && ws, "step_tcn_wgrad_bf16: null pointer");
    STEP_REQUIRE(N > 0 && Tin > 0 && (stride == 1 || stride == 2) && Tout == (Tin - 1) / stride + 1,
                 "step_tcn_wgrad_bf16: bad shape Tin=%d Tout=%d stride=%d", Tin, Tout, stride);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "step_tcn_wgrad_bf16: scale/shift pair");
    WbGeom g;
    STEP_REQUIRE(wb_geom(N, Ci, Co, Kt, stride, &g),
                 "step_tcn_wgrad_bf16: unsupported Ci=%d Co=%d Kt=%d stride=%d", Ci, Co, Kt, stride);
    if (ws_bytes < g.ws) {
        step::set_error("step_tcn_wgrad_bf16: workspace %zu < %zu bytes", ws_bytes, g.ws);
        return STEP_ERR_WORKSPACE;
    }
    WbParams p{};
    p.dout = dout; p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.in_relu = in_relu;
    p.slab = (float*)ws;
    p.dbslab = p.slab + (size_t)g.nsplit * Co * Ci * Kt;
    p.N = N; p.Ci = Ci; p.Co = Co; p.Tin = Tin; p.Tout = Tout; p.Kt = Kt;
    p.pad = (Kt - 1) / 2;
    p.Q = Ci * Kt; p.ntb = g.ntb; p.nqb = g.nqb; p.nsplit = g.nsplit;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (g.mw == 1) rc = launch_wb<1, 10, 1, 16>(p, st);
    else if (stride == 1) rc = launch_wb<2, 5, 1, 16>(p, st);
    else rc = launch_wb<2, 5, 2, 8>(p, st);
    if (rc) return rc;
    return step::wgrad_reduce(p.slab, p.dbslab, dw, db, Co, Ci, Kt, g.nsplit, 0, st,
                              "step_tcn_wgrad_bf16(reduce)");
}
