// Temporal (Kt x 1) convolution of the ST-GCN block as an fp32 MFMA implicit GEMM.
//
// Replaces tcn[2] = Conv2d(C,C,(Kt,1),(s,1),((Kt-1)//2,0)) of the reference
// (classifier_stgcn_real_only/net/classifier.py:156-162), its ConvTranspose2d twin
// (generator_cvae/net/CVAE_stgcn.py:359-365) and the input gradients of both.
//
// One generic form covers all four:
//   out[n, c, PH*rho + ph, v] = epi( SUM_ci SUM_jj wp[ci][jj][ph*Cpad + c] *
//                                    pre(in[n, ci, S*rho + jj - padL, v]) )
// GEMM view (per sequence n): M = PH*Cpad output rows, N = (rho, v) positions,
// K = Cin * 2*KS taps.  Because v (16 joints) is the innermost axis, the flattened
// (t, v) axis makes the temporal convolution a 1-D correlation with dilation 16, so the
// B operand of tap jj is just the LDS-resident input strip of channel ci shifted by jj
// rows.  With v_mfma_f32_32x32x2_f32 a column tile is 2 output rows x 16 joints and one
// k-step is a pair of adjacent taps; the B register of (column tile i, tap pair m)
// depends only on i+m (stride 1) or 2i+m (stride 2, even/odd row planes), so each wave
// keeps a sliding window of B registers and re-uses every LDS read NT times.
//
// Block = MW x NW waves: MW 32-row M tiles (all output channels) x NW column groups of
// NT column tiles.  The K loop runs over input channels; per channel the block stages
// the channel's input strip (BN1 + ReLU applied on the way in, padding rows zero) and
// the channel's packed weights into double-buffered LDS, one barrier per channel.
#include "common.h"
#include "tcn_pack.h"

namespace {

using step::PackGeom;
using step::pack_geom;
using step::pack_value;

struct TcnParams {
    const float* in;
    const float* in_scale;
    const float* in_shift;
    const float* wp;
    const float* bias;
    const float* out_scale;
    const float* out_shift;
    const float* res;
    const float* res_scale;
    const float* res_shift;
    float* out;
    int in_relu, out_relu;
    int N, Cin, Cout, Cpad, PH, Tin, Tout, R, padL, nRB;
    int NG;             // groups of 32*MW output rows (wide layers: PH*Cpad > 128); blockIdx.y
    float* stat_part;   // forward, train mode: per-block [SUM, SUM of squares] of the conv output per channel,
                        // [blocks.x][Cout][2] floats (NULL: none) - the batch statistics of tcn[3]
};

template <int NT, int KS, int STRIDE, int MW, int NW>
struct TcnCfg {
    static constexpr int NTHR = 64 * MW * NW;
    static constexpr int RB = NW * NT * 2;                    // rho rows per block
    static constexpr int ROWS = STRIDE * (RB - 1) + 2 * KS;   // staged input rows
    static constexpr int STRIP = ROWS * 16;                   // floats
    static constexpr int PLANE = (ROWS / 2) * 16;             // stride 2: even | odd rows
    static constexpr int MPAD = 32 * MW;
    static constexpr int WCH = 2 * KS * MPAD;                 // packed weights of 1 channel
    static constexpr int NB = (STRIDE == 1) ? (NT + KS - 1) : (2 * (NT - 1) + KS);
    static constexpr int SF4 = (ROWS * 4 + NTHR - 1) / NTHR;  // float4 per thread
    static constexpr int WCHP = (WCH + 255) / 256 * 256;      // LDS image, whole 1-KiB DMA pieces
    static constexpr int WPIECES = WCHP / 256;                // 64 lanes x 16 B each
    static constexpr int WPW = (WPIECES + MW * NW - 1) / (MW * NW);   // pieces per wave
    static constexpr int BUF = STRIP + WCHP;
    // two operand images; the epilogue re-uses them as wave-private output stages of 32 rows x 64 positions
    static constexpr int SMEM = (2 * BUF > MW * NW * 2048) ? 2 * BUF : MW * NW * 2048;
    static_assert(STRIDE == 1 || ROWS % 2 == 0, "planes");
    static_assert(WCH % 4 == 0, "float4 weights");
};

template <int NT, int KS, int STRIDE, int MW, int NW>
__global__ __launch_bounds__(64 * MW * NW, (TcnCfg<NT, KS, STRIDE, MW, NW>::SMEM * 4 > 80 * 1024) ? 1 : 2) void tcn_conv_kernel(TcnParams p) {
    using Cfg = TcnCfg<NT, KS, STRIDE, MW, NW>;
    __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int mw = wave / NW;
    const int wn = wave % NW;
    const int l31 = lane & 31;
    const int half = lane >> 5;

    const int n = blockIdx.x / p.nRB;
    const int rho0 = (blockIdx.x % p.nRB) * Cfg::RB;
    const int tin0 = STRIDE * rho0 - p.padL;                  // input row of strip row 0
    const bool wave_live = (rho0 + 2 * wn * NT) < p.R;

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float4 sreg[Cfg::SF4];

    // global -> registers: raw loads only, nothing waits on them before the MFMA phase
    auto load_regs = [&](int c) {
        const float* src = p.in + ((size_t)n * p.Cin + c) * p.Tin * 16;
#pragma unroll
        for (int k = 0; k < Cfg::SF4; ++k) {
            const int f = tid + k * Cfg::NTHR;
            const int tin = tin0 + (f >> 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < Cfg::ROWS * 4 && tin >= 0 && tin < p.Tin)
                v = *reinterpret_cast<const float4*>(src + tin * 16 + (f & 3) * 4);
            sreg[k] = v;
        }
    };

    // packed weights of channel c: global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
    // wave-instruction, no VGPRs, no ds_write).  The image is a straight copy, so the lane-linear
    // destination is exactly right; the tail piece reads into the pad of the packed buffer.
    auto load_w = [&](int c, float* buf) {
        const float* wsrc = p.wp + ((size_t)blockIdx.y * p.Cin + c) * Cfg::WCH + lane * 4;
#pragma unroll
        for (int k = 0; k < Cfg::WPW; ++k) {
            const int piece = k * (MW * NW) + wave;
            if (piece < Cfg::WPIECES)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(wsrc + piece * 256),
                    (__attribute__((address_space(3))) void*)(buf + Cfg::STRIP + piece * 256), 16, 0, 0);
        }
    };

    // registers -> LDS, one float4 item at a time so the MFMA phase can carry the writes between
    // its k-steps; tcn[0] scale/shift and tcn[1] ReLU are applied here (padding rows stay exactly
    // zero: the reference pads after the activation)
    auto store_item = [&](float* buf, int item, float sc, float sh) {
        if (item < Cfg::SF4) {
            const int f = tid + item * Cfg::NTHR;
            if (f < Cfg::ROWS * 4) {
                const int lr = f >> 2;
                const int tin = tin0 + lr;
                float4 v = sreg[item];
                if (tin >= 0 && tin < p.Tin) {
                    v.x = v.x * sc + sh; v.y = v.y * sc + sh;
                    v.z = v.z * sc + sh; v.w = v.w * sc + sh;
                    if (p.in_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
                int dst;
                if (STRIDE == 1) dst = lr * 16 + (f & 3) * 4;
                else dst = (lr & 1) * Cfg::PLANE + (lr >> 1) * 16 + (f & 3) * 4;
                *reinterpret_cast<float4*>(buf + dst) = v;
            }
        }
    };
    constexpr int NITEM = Cfg::SF4;

    // One channel: KS k-steps of NT MFMAs.  LDS operand reads run PF k-steps ahead of the MFMAs
    // that use them, and the register -> LDS writes of the NEXT channel's staging ride between the
    // last k-steps, so the wave's instruction stream stays MFMA-dense from the first k-step to
    // the barrier (the scheduler may not move anything across the per-step fences).
    constexpr int PF = 4;
    auto compute = [&](const float* buf, float* nxt, bool more, float sc, float sh) {
        const float* W = buf + Cfg::STRIP + mw * 32 + l31 + half * Cfg::MPAD;
        const float* bs = (STRIDE == 1) ? buf + (2 * wn * NT) * 16 + l31 + half * 16
                                        : buf + half * Cfg::PLANE + (2 * wn * NT) * 16 + l31;
        constexpr int BSTEP = (STRIDE == 1) ? 32 : 16;
        constexpr int NB0 = (STRIDE == 1) ? NT : 2 * (NT - 1) + 1;     // B registers of k-step 0
        float b[Cfg::NB];
        float a[KS];
#pragma unroll
        for (int q = 0; q < NB0 + PF - 1 && q < Cfg::NB; ++q) b[q] = bs[q * BSTEP];
#pragma unroll
        for (int m = 0; m < PF && m < KS; ++m) a[m] = W[2 * m * Cfg::MPAD];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < KS; ++m) {
            if (m + PF < KS) {
                a[m + PF] = W[2 * (m + PF) * Cfg::MPAD];
                const int q = NB0 + PF - 1 + m;
                if (q < Cfg::NB) b[q] = bs[q * BSTEP];
            }
            const int item = m - (KS - NITEM);                          // staging rides the last k-steps
            if (more && item >= 0 && item < NITEM) store_item(nxt, item, sc, sh);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int q = (STRIDE == 1) ? (i + m) : (2 * i + m);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[q], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(NITEM <= KS, "staging items must fit the k-steps");

    auto scale_of = [&](int c, float& sc, float& sh) {
        sc = 1.f; sh = 0.f;
        if (p.in_scale && c < p.Cin) { sc = p.in_scale[c]; sh = p.in_shift[c]; }
    };

    {
        float sc, sh;
        scale_of(0, sc, sh);
        load_regs(0);
        load_w(0, smem);
#pragma unroll
        for (int it = 0; it < NITEM; ++it) store_item(smem, it, sc, sh);
    }
    __syncthreads();
    if (wave_live) {
        for (int c = 0; c < p.Cin; ++c) {
            float* cur = smem + (c & 1) * Cfg::BUF;
            float* nxt = smem + ((c + 1) & 1) * Cfg::BUF;
            const bool more = (c + 1) < p.Cin;
            float sc, sh;
            scale_of(c + 1, sc, sh);
            if (more) { load_regs(c + 1); load_w(c + 1, nxt); }
            compute(cur, nxt, more, sc, sh);
            __syncthreads();
        }
    } else {
        // a wave whose frames lie past the end of the sequence only helps with the staging
        for (int c = 0; c + 1 < p.Cin; ++c) {
            float* nxt = smem + ((c + 1) & 1) * Cfg::BUF;
            float sc, sh;
            scale_of(c + 1, sc, sh);
            load_regs(c + 1);
            load_w(c + 1, nxt);
#pragma unroll
            for (int it = 0; it < NITEM; ++it) store_item(nxt, it, sc, sh);
            __syncthreads();
        }
        if (p.Cin >= 1) __syncthreads();
    }

    __shared__ float stat_x[MW * NW][64];                     // per wave: [32 channels][sum | sum of squares]
    float st1[8], st2[8];                                     // this lane's sums of channels k * 4 + lch
#pragma unroll
    for (int k = 0; k < 8; ++k) { st1[k] = 0.f; st2[k] = 0.f; }
    if (wave_live) {
    // epilogue.  C/D layout: col = lane&31 = (frame parity, joint), row = (r&3) + 8*(r>>2) + 4*(lane>>5) = channel,
    // i.e. a lane holds ONE float of a 64-byte (channel, frame) row, and 4-byte global stores are issue-bound (a
    // wave-instruction costs about the same whatever its width).  So two column tiles at a time go through a
    // wave-private LDS stage [32 rows][64 positions] (the operand images are dead: the channel loop ended with a
    // barrier) and come back as float4 along the frame axis: bias, tcn[3] affine, residual and ReLU are applied
    // there, the residual is read 16 bytes wide as well, and one wave-instruction stores 1 KiB.
    float* stg = smem + wave * 2048;
    const int lch = lane >> 4, lq = lane & 15;
    const int fp = lq >> 2, v0 = (lq & 3) * 4;            // frame of the pair of tiles, first joint
#pragma unroll
    for (int i0 = 0; i0 < NT; i0 += 2) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            if (i0 + jj < NT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + jj * 32 + l31] = acc[i0 + jj][r];
            }
        const int jt = fp >> 1;                               // column tile of this lane's frame
        const int rho = rho0 + 2 * (wn * NT + i0 + jt) + (fp & 1);
        const bool col_ok = (i0 + jt < NT) && rho < p.R;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int chl = k * 4 + lch;
            const int m = blockIdx.y * Cfg::MPAD + mw * 32 + chl;
            const int ph = m / p.Cpad;
            const int c = m - ph * p.Cpad;
            const int t = p.PH * rho + ph;
            if (col_ok && c < p.Cout && ph < p.PH && t < p.Tout) {
                float4 v = *reinterpret_cast<const float4*>(stg + chl * 64 + lq * 4);
                const float bia = p.bias ? p.bias[c] : 0.f;
                float osc = 1.f, osh = 0.f;
                if (p.out_scale) { osc = p.out_scale[c]; osh = p.out_shift[c]; }
                v.x = (v.x + bia) * osc + osh; v.y = (v.y + bia) * osc + osh;
                v.z = (v.z + bia) * osc + osh; v.w = (v.w + bia) * osc + osh;
                const size_t o = (((size_t)n * p.Cout + c) * p.Tout + t) * 16 + v0;
                if (p.res) {
                    float rsc = 1.f, rsh = 0.f;
                    if (p.res_scale) { rsc = p.res_scale[c]; rsh = p.res_shift[c]; }
                    const float4 rv = *reinterpret_cast<const float4*>(p.res + o);
                    v.x += rv.x * rsc + rsh; v.y += rv.y * rsc + rsh; v.z += rv.z * rsc + rsh; v.w += rv.w * rsc + rsh;
                }
                if (p.out_relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *reinterpret_cast<float4*>(p.out + o) = v;
                if (p.stat_part) {
                    st1[k] += (v.x + v.y) + (v.z + v.w);
                    st2[k] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
            }
        }
    }
    }
    // batch statistics of the output for the BatchNorm behind it (train-mode forward only; uniform branch): the 16
    // lanes that share a channel are summed by DPP row rotations, the NW column groups through LDS in a fixed
    // order, and the block leaves one fp32 partial per channel (at most 40 frames x 16 joints of one sequence);
    // tcn_stat_fold_kernel carries on in fp64.  Dead waves (frames past the sequence end) contribute zeros.
    if (p.stat_part) {
        auto rsum = [](float v) {
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));
            return v;
        };
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float a = rsum(st1[k]), b = rsum(st2[k]);
            if ((lane & 15) == 0) {
                stat_x[wave][(k * 4 + (lane >> 4)) * 2] = a;
                stat_x[wave][(k * 4 + (lane >> 4)) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < 64 * MW) {
            const int m = tid >> 1, which = tid & 1;            // row of this block's M extent
            const int mwi = m >> 5, chl = m & 31;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NW; ++j) sum += stat_x[mwi * NW + j][chl * 2 + which];
            const int c = blockIdx.y * Cfg::MPAD + m;            // forward: PH == 1, row = output channel
            if (c < p.Cout) p.stat_part[((size_t)blockIdx.x * p.Cout + c) * 2 + which] = sum;
        }
    }
}

// [B][2C] fp32 block partials -> [256][2C] fp64 chunk partials (bn.hip's ordered finalize takes it from there): chunk j
// sums blocks [j*B/256, (j+1)*B/256) in order, two half ranges per entry combined through LDS
__global__ __launch_bounds__(256) void tcn_stat_fold_kernel(const float* __restrict__ part, double* __restrict__ out,
                                                            int B, int E) {
    __shared__ double red[256];
    const int j = blockIdx.x;
    const int b0 = (int)((long long)B * j / 256), b1 = (int)((long long)B * (j + 1) / 256);
    for (int e0 = 0; e0 < E; e0 += 128) {
        const int e = e0 + (threadIdx.x & 127), h = threadIdx.x >> 7;
        const int mid = b0 + (b1 - b0 + 1) / 2;
        const int lo = h ? mid : b0, hi = h ? b1 : mid;
        double s = 0.0;
        if (e < E)
            for (int b = lo; b < hi; ++b) s += (double)part[(size_t)b * E + e];
        red[threadIdx.x] = s;
        __syncthreads();
        if (h == 0 && e < E) out[(size_t)j * E + e] = red[threadIdx.x] + red[threadIdx.x + 128];
        __syncthreads();
    }
}

// ------------------------------------------------------------------ weight re-layout
__global__ void tcn_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co,
                                int Ci, int Kt, int stride, int mode, PackGeom g) {
    const int mpad = g.ph * g.cpad;
    const int mg = mpad > 128 ? 128 : mpad;            // rows per group (one block's M extent)
    const int total = g.cin_g * 2 * g.ks * mpad;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int ml = e % mg;
        const int jj = (e / mg) % (2 * g.ks);
        const int cg = (e / (mg * 2 * g.ks)) % g.cin_g;     // GEMM input channel
        const int grp = e / (mg * 2 * g.ks * g.cin_g);
        wp[e] = pack_value(w, Co, Ci, Kt, stride, mode, g, cg, jj, grp * mg + ml);
    }
}

template <int NT, int KS, int STRIDE, int MW, int NW>
int launch_tcn(const TcnParams& p, hipStream_t st, int* nblocks = nullptr) {
    using Cfg = TcnCfg<NT, KS, STRIDE, MW, NW>;
    TcnParams q = p;
    q.nRB = step::ceil_div(p.R, Cfg::RB);
    const long long blocks = (long long)p.N * q.nRB;
    if (nblocks) *nblocks = (int)blocks;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        step::set_error("step_tcn_conv: grid of %lld blocks out of range", blocks);
        return STEP_ERR_ARG;
    }
    hipLaunchKernelGGL((tcn_conv_kernel<NT, KS, STRIDE, MW, NW>), dim3((unsigned)blocks, (unsigned)p.NG),
                       dim3(Cfg::NTHR), 0, st, q);
    return step::check_launch("step_tcn_conv");
}

}  // namespace

extern "C" size_t step_tcn_packed_elems(int Co, int Ci, int Kt, int stride, int pack_mode) {
    PackGeom g;
    if (Co < 1 || Ci < 1 || !pack_geom(Co, Ci, Kt, stride, pack_mode, &g)) return 0;
    return (size_t)g.cin_g * 2 * g.ks * g.ph * g.cpad + 256;   // + pad for the last DMA piece
}

extern "C" int step_tcn_pack_weight(const float* w, float* wp, int Co, int Ci, int Kt, int stride,
                                    int pack_mode, void* stream) {
    PackGeom g;
    STEP_REQUIRE(w && wp, "step_tcn_pack_weight: null pointer");
    STEP_REQUIRE(Co >= 1 && Ci >= 1 && pack_geom(Co, Ci, Kt, stride, pack_mode, &g),
                 "step_tcn_pack_weight: unsupported Co=%d Ci=%d Kt=%d stride=%d mode=%d", Co, Ci, Kt,
                 stride, pack_mode);
    const int total = g.cin_g * 2 * g.ks * g.ph * g.cpad;
    const int blocks = step::ceil_div(total, 256) < 1024 ? step::ceil_div(total, 256) : 1024;
    hipLaunchKernelGGL(tcn_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, Co, Ci,
                       Kt, stride, pack_mode, g);
    return step::check_launch("step_tcn_pack_weight");
}

namespace {
// rows (rho) per block of the forward kernels: NT * 4 for every MW x NW shape in use (RB = NW * NT * 2, NW = 2 or 4
// with NT = 5 -> 20 or 40).  Upper bound of the block count for the statistics workspace: RB >= 16.
int tcn_conv(const float* in, const float* in_scale, const float* in_shift,
             int in_relu, const float* wp, const float* bias,
             const float* out_scale, const float* out_shift, const float* res,
             const float* res_scale, const float* res_shift, int out_relu,
             float* out, int N, int Ci, int Co, int Tin, int Tout, int Kt,
             int stride, int pack_mode, void* stream, float* stat_part, int* nblocks) {
    PackGeom g;
    STEP_REQUIRE(in && wp && out, "step_tcn_conv: null pointer");
    STEP_REQUIRE(N >= 1 && Ci >= 1 && Co >= 1 && Tin >= 1 && Tout >= 1,
                 "step_tcn_conv: bad shape N=%d Ci=%d Co=%d Tin=%d Tout=%d", N, Ci, Co, Tin, Tout);
    STEP_REQUIRE(pack_geom(Co, Ci, Kt, stride, pack_mode, &g),
                 "step_tcn_conv: unsupported Kt=%d stride=%d mode=%d", Kt, stride, pack_mode);
    STEP_REQUIRE(Tout == (Tin - 1) / stride + 1, "step_tcn_conv: Tout=%d != (Tin-1)/stride+1 (Tin=%d)",
                 Tout, Tin);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) &&
                     (out_scale == nullptr) == (out_shift == nullptr) &&
                     (res_scale == nullptr) == (res_shift == nullptr),
                 "step_tcn_conv: scale/shift must come in pairs");
    TcnParams p{};
    p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.in_relu = in_relu;
    p.wp = wp; p.bias = bias; p.out_scale = out_scale; p.out_shift = out_shift;
    p.res = res; p.res_scale = res_scale; p.res_shift = res_shift; p.out_relu = out_relu;
    p.out = out; p.N = N; p.Cin = g.cin_g; p.Cout = g.cout_g; p.Cpad = g.cpad; p.PH = g.ph;
    p.padL = g.padL;
    p.stat_part = stat_part;
    STEP_REQUIRE(!stat_part || (pack_mode < 2 && g.ph == 1), "step_tcn_conv_stats: forward pack modes only");
    const bool dgrad = pack_mode >= 2;
    p.Tin = dgrad ? Tout : Tin;                 // rows of the tensor being read
    p.Tout = dgrad ? Tin : Tout;                // rows of the tensor being written
    p.R = step::ceil_div(p.Tout, g.ph);
    int mw = g.ph * g.cpad / 32;
    p.NG = 1;
    if (mw > 4) {                               // wide layers: groups of 128 output rows on blockIdx.y
        STEP_REQUIRE(mw % 4 == 0 && mw / 4 <= 65535, "step_tcn_conv: %d output rows are not a multiple of 128",
                     g.ph * g.cpad);
        p.NG = mw / 4;
        mw = 4;
    }
    const int key = g.ks * 100 + g.stride_g * 10 + mw;
    hipStream_t st = (hipStream_t)stream;
    // 128-row blocks cover 4*NT rho rows each: NT = 4 where that leaves fewer dead rows in the last block of a
    // sequence than NT = 5 (R = 30: 32 instead of 40 rows of MFMA work), ties go to NT = 5 (one A read per 5 MFMAs)
    const bool nt4 = step::ceil_div(p.R, 16) * 16 < step::ceil_div(p.R, 20) * 20;
    switch (key) {
        case 3814: return nt4 ? launch_tcn<4, 38, 1, 4, 2>(p, st, nblocks) : launch_tcn<5, 38, 1, 4, 2>(p, st, nblocks);
        case 3824: return nt4 ? launch_tcn<4, 38, 2, 4, 2>(p, st, nblocks) : launch_tcn<5, 38, 2, 4, 2>(p, st, nblocks);
        case 514: return launch_tcn<5, 5, 1, 4, 2>(p, st, nblocks);
        case 524: return launch_tcn<5, 5, 2, 4, 2>(p, st, nblocks);
        case 3811: return launch_tcn<5, 38, 1, 1, 4>(p, st, nblocks);
        case 3812: return launch_tcn<5, 38, 1, 2, 2>(p, st, nblocks);
        case 3821: return launch_tcn<5, 38, 2, 1, 4>(p, st, nblocks);
        case 3822: return launch_tcn<5, 38, 2, 2, 2>(p, st, nblocks);
        case 1912: return launch_tcn<5, 19, 1, 2, 2>(p, st, nblocks);
        case 1914: return nt4 ? launch_tcn<4, 19, 1, 4, 2>(p, st, nblocks) : launch_tcn<5, 19, 1, 4, 2>(p, st, nblocks);
        case 511: return launch_tcn<5, 5, 1, 1, 4>(p, st, nblocks);
        case 512: return launch_tcn<5, 5, 1, 2, 2>(p, st, nblocks);
        case 521: return launch_tcn<5, 5, 2, 1, 4>(p, st, nblocks);
        case 522: return launch_tcn<5, 5, 2, 2, 2>(p, st, nblocks);
        case 312: return launch_tcn<5, 3, 1, 2, 2>(p, st, nblocks);
        case 314: return launch_tcn<5, 3, 1, 4, 2>(p, st, nblocks);
        default: break;
    }
    step::set_error("step_tcn_conv: no kernel for Kt=%d stride=%d channels=%d (ks=%d mw=%d)", Kt,
                    stride, g.cout_g, g.ks, mw);
    return STEP_ERR_ARG;
}
}  // namespace

extern "C" int step_tcn_conv(const float* in, const float* in_scale, const float* in_shift,
                             int in_relu, const float* wp, const float* bias,
                             const float* out_scale, const float* out_shift, const float* res,
                             const float* res_scale, const float* res_shift, int out_relu,
                             float* out, int N, int Ci, int Co, int Tin, int Tout, int Kt,
                             int stride, int pack_mode, void* stream) {
    return tcn_conv(in, in_scale, in_shift, in_relu, wp, bias, out_scale, out_shift, res, res_scale, res_shift,
                    out_relu, out, N, Ci, Co, Tin, Tout, Kt, stride, pack_mode, stream, nullptr, nullptr);
}

// workspace of step_tcn_conv_stats: fp32 block partials (a block covers at least 16 output frames of one sequence),
// then 256 fp64 chunk partials
static size_t tcn_stats_part_bytes(int N, int Co, int Tout) {
    return (size_t)N * step::ceil_div(Tout, 16) * Co * 2 * sizeof(float);
}
extern "C" size_t step_tcn_conv_stats_ws_bytes(int N, int Co, int Tout) {
    if (N < 1 || Co < 1 || Tout < 1) return 0;
    const size_t a = (tcn_stats_part_bytes(N, Co, Tout) + 255) / 256 * 256 + (size_t)256 * Co * 2 * sizeof(double);
    const size_t b = step_bn_ws_bytes(Co);
    return a > b ? a : b;
}

extern "C" int step_tcn_conv_stats(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                                   const float* wp, const float* bias, float* out,
                                   int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride, int pack_mode,
                                   double* sums, void* ws, size_t ws_bytes, void* stream) {
    STEP_REQUIRE(sums && ws, "step_tcn_conv_stats: null pointer");
    STEP_REQUIRE(pack_mode == 0 || pack_mode == 1, "step_tcn_conv_stats: forward pack modes only (got %d)", pack_mode);
    if (ws_bytes < step_tcn_conv_stats_ws_bytes(N, Co, Tout)) {
        step::set_error("step_tcn_conv_stats: workspace %zu < %zu bytes", ws_bytes, step_tcn_conv_stats_ws_bytes(N, Co, Tout));
        return STEP_ERR_WORKSPACE;
    }
    float* part = (float*)ws;
    double* chunks = (double*)((char*)ws + (tcn_stats_part_bytes(N, Co, Tout) + 255) / 256 * 256);
    int blocks = 0;
    int rc = tcn_conv(in, in_scale, in_shift, in_relu, wp, bias, nullptr, nullptr, nullptr, nullptr, nullptr, 0, out,
                      N, Ci, Co, Tin, Tout, Kt, stride, pack_mode, stream, part, &blocks);
    if (rc != STEP_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tcn_stat_fold_kernel, dim3(256), dim3(256), 0, st, part, chunks, blocks, 2 * Co);
    rc = step::check_launch("step_tcn_conv_stats(fold)");
    if (rc != STEP_OK) return rc;
    return step::bn_sum_partials(chunks, sums, Co, 256, st);
}
