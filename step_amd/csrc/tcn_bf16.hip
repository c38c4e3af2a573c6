// Temporal (Kt x 1) convolution as a bf16 MFMA implicit GEMM with fp32 tensors in HBM.
//
// Same operator, generic form and epilogue as tcn.hip (tcn[2] of st_gcn,
// classifier_stgcn_real_only/net/classifier.py:156-162, and its input gradient); the
// activations and weights enter v_mfma_f32_32x32x16_bf16 as bf16, accumulation is fp32
// (BASELINE config "classifier_stgcn_real_and_synth training, bf16").  Activations stay fp32
// in HBM: they are converted on their way into LDS, after tcn[0]/tcn[1] (BN + ReLU).
//
// GEMM view per sequence: M = PH*Cpad output rows, N = (rho, v) positions, K = Cin * KT2
// taps.  One MFMA contracts 16 INPUT CHANNELS of one tap:
//   A[m][k]   = wgen[16g + k][jj][m]            packed per lane, read straight from L2
//   B[k][col] = strip[16g + k][S*rho + jj][v]   col = (frame parity, joint)
// The strip of a 16-channel group lies in LDS in the tensor's own order ([channel][frame][v],
// bf16), so staging is a float4 load -> 4 x bf16 -> one 8-byte LDS write, and the B operand
// is gathered channel-major by the transposing LDS read (ds_read_b64_tr_b16: 4 channels x 16
// joints per 16 lanes).  As in tcn.hip the B fragment of (column tile i, tap jj) depends only
// on 2*S*i + jj, so a wave keeps a sliding window of fragments and reads each once.
#include "common.h"
#include "tcn_pack.h"
#include <type_traits>

namespace {

using step::PackGeom;
using step::pack_geom;
using step::pack_value;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 pk_bf16(float a, float b) {
    return __builtin_bit_cast(s16x2, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

struct TbParams {
    const float* in;
    const float* in_scale;
    const float* in_shift;
    const bf16x8* wp;
    const float* bias;
    const float* out_scale;
    const float* out_shift;
    const float* res;
    const float* res_scale;
    const float* res_shift;
    float* out;
    double* stat_part;   // [gridDim.x * NW][Cout][2]: SUM out, SUM out^2 per MFMA wave row (train-mode forward), or null
    int in_relu, out_relu;
    int N, Cin, Cout, Cpad, PH, Tin, Tout, R, padL, nRB, G;
};

constexpr int NPROD = 4;          // staging waves per block

template <int NT, int KT2, int STRIDE, int MW, int NW, bool SPLIT = false>
struct TbCfg {
    static constexpr int NCONS = MW * NW;                     // MFMA waves
    static constexpr int NTHR = 64 * (NCONS + NPROD);
    static constexpr int PTHR = 64 * NPROD;
    static constexpr int RB = NW * NT * 2;                    // rho rows per tile
    static constexpr int ROWS = STRIDE * (RB - 1) + KT2;      // staged input frames
    // bytes between channels of the strip: == 64 (mod 256), which spreads the 4 channel rows
    // of a transposing read (and the neighbouring frame's 4) over all 64 banks
    static constexpr int CHS = ((ROWS * 32 - 64 + 255) / 256) * 256 + 64;
    static constexpr int STRIP_BYTES = 16 * CHS;
    static constexpr int STAGE_BYTES = 4 * 32 * 32 * 4;       // output stages: one 32 x 32 tile per MFMA wave
    static constexpr int STAT_BYTES = 4 * 64 * 8;             // per MFMA wave: [32 channels][sum | sum of squares], fp64
    static constexpr int LDS_BYTES = 2 * STRIP_BYTES + STAGE_BYTES + STAT_BYTES;   // double-buffered strips + stages
    static constexpr int GCH = SPLIT ? 8 : 16;                // input channels of one group (one MFMA's K = 16)
    static constexpr int NBTOT = STRIDE * 2 * (NT - 1) + KT2; // B fragments of one group
    static constexpr int NB0 = STRIDE * 2 * (NT - 1) + 1;     // ... needed by tap 0
    static_assert(CHS >= ROWS * 32, "strip row");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(NCONS == 4, "one MFMA wave per SIMD");
};

// Work item = (tile, 16-channel group).  A block is NCONS MFMA waves (one per SIMD, MW M tiles
// x NW column groups of NT column tiles) plus NPROD staging waves; it walks its tiles
// persistently.  While the MFMA waves contract item i out of one LDS strip, the staging waves
// bring item i+1 (next group, or the next tile's first group) into the other: HBM -> registers
// -> BN + ReLU -> bf16 -> LDS.  One barrier per item; the two roles never touch the same buffer
// between barriers, and the staging loads never sit in front of the MFMA waves' operand loads.
// Round 3 ablations (64 -> 64, stride 1, T = 120, N = 4096, 4.2 ms): every tap reading the same A fragment (all L1
// hits): -2 %; no staging at all: -10 % (3.7 ms = 1.3 PFLOP/s, against the 1.58-1.84 PFLOP/s this box sustains on
// register-only bf16 MFMAs at the 1.7-1.87 GHz they pull the clock to).  The staging rewrite below (6-10 VALU
// instructions per float4 instead of ~35) is worth 5 % in the split mode and nothing measurable in the bf16 mode.
//
// SPLIT (fp32 operands at bf16-MFMA speed): a group is 8 channels, each as TWO strip rows - hi = bf16(v) and
// lo = bf16(v - hi), k = 2*channel + part - and a tap takes two MFMAs: A = [w_hi, w_hi] pairs against (hi, lo) gives
// w_hi*(v_hi + v_lo), A = [w_lo, 0] gives w_lo*v_hi.  The dropped w_lo*v_lo is 2^-16 relative; fp32 accumulate.
template <int NT, int KT2, int STRIDE, int MW, int NW, bool SPLIT>
__global__ __launch_bounds__(64 * (MW * NW + NPROD)) void tcn_bf16_kernel(TbParams p) {
    using Cfg = TbCfg<NT, KT2, STRIDE, MW, NW, SPLIT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.N * p.nRB;
    // blocks that share an XCD (blockIdx % 8) take neighbouring tiles: their halos overlap in L2
    int lb = blockIdx.x;
    if ((gridDim.x & 7) == 0) lb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    if (wave >= Cfg::NCONS) {
        // ------------------------------------------------------------ staging waves
        const int ptid = tid - 64 * Cfg::NCONS;
        // the MFMA wave on this SIMD always has an instruction ready and would win every
        // arbitration; the staging wave issues rarely (it mostly waits on memory) but must not queue
        __builtin_amdgcn_s_setprio(3);
        // Thread -> (channel row c of the group, position li among the row's TPCH threads); the row's float4 items
        // li, li + TPCH, ... are a linear walk of the tensor's channel row AND of the strip row, so both addresses are
        // one per-item base plus compile-time offsets, scale / shift are two values per thread, and only tiles that
        // touch an end of the sequence test positions.  (An instruction a staging wave issues costs the MFMA wave on
        // its SIMD time - tcn_wgrad_sw.hip; the first version spent ~35 VALU instructions per float4 here.)
        constexpr int TPCH = Cfg::PTHR / Cfg::GCH;
        constexpr int RI = Cfg::ROWS * 4;                        // float4 items of one channel row
        constexpr int NK = (RI + TPCH - 1) / TPCH;
        const int c = ptid / TPCH, li = ptid % TPCH;
        f32x4 sreg[NK];
        auto tile_t0 = [&](int tile) { return STRIDE * (tile % p.nRB) * Cfg::RB - p.padL; };   // input frame of strip row 0
        // HBM -> registers (raw); the loads stay in flight across the barrier
        auto load_regs = [&](int tile, int g) {
            const int n = tile / p.nRB;
            const int tin0 = tile_t0(tile);
            const f32x4* row = reinterpret_cast<const f32x4*>(p.in + ((size_t)n * p.Cin + Cfg::GCH * g + c) * p.Tin * 16);
            const int e0 = tin0 * 4 + li;
            if (tin0 >= 0 && tin0 + Cfg::ROWS <= p.Tin) {        // (uniform) every frame of the strip exists
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    if ((k + 1) * TPCH <= RI || li + k * TPCH < RI) sreg[k] = row[e0 + k * TPCH];
                }
            } else {
                const unsigned lim = (unsigned)p.Tin * 4u;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int e = e0 + k * TPCH;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if ((unsigned)e < lim && ((k + 1) * TPCH <= RI || li + k * TPCH < RI)) v = row[e];
                    sreg[k] = v;
                }
            }
        };
        // registers -> tcn[0] scale/shift, tcn[1] ReLU -> bf16 -> LDS; padding frames stay exactly
        // zero (the reference pads after the activation)
        auto store_body = [&](auto act_tag, auto relu_tag, auto edge_tag, int tile, int g, char* buf) {
            constexpr bool ACT = decltype(act_tag)::value, RELU = decltype(relu_tag)::value, EDGE = decltype(edge_tag)::value;
            float sc = 1.f, sh = 0.f;
            if constexpr (ACT) { sc = p.in_scale[Cfg::GCH * g + c]; sh = p.in_shift[Cfg::GCH * g + c]; }
            const int e0 = tile_t0(tile) * 4 + li;
            const unsigned lim = (unsigned)p.Tin * 4u;
            char* dst = buf + (SPLIT ? 2 * c : c) * Cfg::CHS + li * 8;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if ((k + 1) * TPCH > RI && li + k * TPCH >= RI) continue;      // (only the row's last, partial step tests)
                f32x4 v = sreg[k];
                if constexpr (ACT) {
                    float s1 = sc, s0 = sh;
                    if constexpr (EDGE) {                        // outside the sequence: 0 * 0 + 0 (the load gave zeros)
                        const bool in = (unsigned)(e0 + k * TPCH) < lim;
                        s1 = in ? sc : 0.f; s0 = in ? sh : 0.f;
                    }
                    v = v * s1 + s0;
                }
                if constexpr (SPLIT) {
                    if constexpr (RELU) {
                        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                    }
                    const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                    const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
                    *reinterpret_cast<bf16x4*>(dst + k * TPCH * 8) = hi;
                    *reinterpret_cast<bf16x4*>(dst + Cfg::CHS + k * TPCH * 8) = lo;
                } else {
                    // rounding is monotonic and keeps the sign, so ReLU commutes with it: on the packed bf16 pair it is
                    // a signed 16-bit max with 0 (one instruction per two values)
                    s16x2 h0 = pk_bf16(v[0], v[1]), h1 = pk_bf16(v[2], v[3]);     // (one v_cvt_pk_bf16_f32 each)
                    if constexpr (RELU) {
                        h0 = __builtin_elementwise_max(h0, s16x2{0, 0});
                        h1 = __builtin_elementwise_max(h1, s16x2{0, 0});
                    }
                    *reinterpret_cast<s16x4*>(dst + k * TPCH * 8) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
                }
            }
        };
        auto store_regs = [&](int tile, int g, char* buf) {
            using T = std::true_type; using F = std::false_type;
            const int tin0 = tile_t0(tile);
            const bool edge = !(tin0 >= 0 && tin0 + Cfg::ROWS <= p.Tin);
            if (p.in_scale) {
                if (p.in_relu) { if (edge) store_body(T{}, T{}, T{}, tile, g, buf); else store_body(T{}, T{}, F{}, tile, g, buf); }
                else           { if (edge) store_body(T{}, F{}, T{}, tile, g, buf); else store_body(T{}, F{}, F{}, tile, g, buf); }
            } else {
                if (p.in_relu) store_body(F{}, T{}, F{}, tile, g, buf); else store_body(F{}, F{}, F{}, tile, g, buf);
            }
        };
        // item sequence of this block: (tile, g) with g fastest
        auto advance = [&](int& tile, int& g) {
            if (++g == p.G) { g = 0; tile += gridDim.x; }
        };
        int t1 = lb, g1 = 0;                   // item whose registers are loaded / being stored
        if (t1 < ntiles) { load_regs(t1, g1); store_regs(t1, g1, smem); }
        advance(t1, g1);
        if (t1 < ntiles) load_regs(t1, g1);    // item 1 in flight across the first barrier
        __syncthreads();
        int it = 0;
        for (int tile = lb; tile < ntiles; tile += gridDim.x) {
            for (int g = 0; g < p.G; ++g) {
                // the MFMA waves work on item `it`; item it+1 goes to the other strip, and the
                // loads of item it+2 are issued before the barrier so they fly during item it+1
                if (t1 < ntiles) store_regs(t1, g1, smem + ((it + 1) & 1) * Cfg::STRIP_BYTES);
                advance(t1, g1);
                if (t1 < ntiles) load_regs(t1, g1);
                __syncthreads();
                ++it;
            }
        }
        return;
    }

    // ---------------------------------------------------------------- MFMA waves
    const int mw = wave / NW;
    const int wn = wave % NW;
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // transposing-read address of this lane: channel row 8*half + q, frame (lane>>4)&1 of the
    // column tile, joints 4p..4p+3  (q = (lane>>2)&3, p = lane&3)
    const int boff = (8 * half + ((lane >> 2) & 3)) * Cfg::CHS +
                     STRIDE * (2 * wn * NT + ((lane >> 4) & 1)) * 32 + (lane & 3) * 8;

    f32x16 acc[NT];
    constexpr int NS = SPLIT ? 2 : 1;         // A fragments (MFMAs) per tap and column tile
    constexpr int PFA = SPLIT ? 3 : 6;        // taps of A-fragment prefetch (L2 latency)
    constexpr int PFB = 2;      // taps of B-fragment prefetch (LDS latency)
    auto compute = [&](int g, const char* buf) {
        // A fragments: [group][tap][set][M tile][lane]
        const bf16x8* wa = p.wp + ((size_t)g * KT2 * NS * MW + mw) * 64 + lane;
        const char* bb = buf + boff;
        auto read_b = [&](int q) -> bf16x8 {
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(bb + q * 32));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (lds_bf16x4_ptr)(bb + 4 * Cfg::CHS + q * 32));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        bf16x8 a[KT2][NS];
        bf16x8 b[Cfg::NBTOT];
#pragma unroll
        for (int m = 0; m < PFA && m < KT2; ++m)
#pragma unroll
            for (int t = 0; t < NS; ++t) a[m][t] = wa[(m * NS + t) * MW * 64];
#pragma unroll
        for (int q = 0; q < Cfg::NB0 + PFB - 1 && q < Cfg::NBTOT; ++q) b[q] = read_b(q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KT2; ++j) {
            if (j + PFA < KT2) {
#pragma unroll
                for (int t = 0; t < NS; ++t) a[j + PFA][t] = wa[((j + PFA) * NS + t) * MW * 64];
            }
            {
                const int q = Cfg::NB0 + PFB - 1 + j;
                if (q < Cfg::NBTOT) b[q] = read_b(q);
            }
            // (split: the small term w_lo * v_hi first, over all column tiles, then w_hi * (v_hi + v_lo))
#pragma unroll
            for (int t = NS - 1; t >= 0; --t)
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j][t], b[STRIDE * 2 * i + j], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Batch statistics of the output for the BatchNorm behind it (train-mode forward, p.stat_part != NULL; as
    // step_tcn_conv_stats of tcn.hip): per tile a lane sums its values in fp32 (NT x 4 per channel), the 8 lanes that
    // share a channel row are combined by DPP row rotations and added to this wave's fp64 accumulators in LDS; after
    // the last tile the wave writes one fp64 partial per channel into its own row for bn.hip's ordered finalize.
    double* dacc = reinterpret_cast<double*>(smem + 2 * Cfg::STRIP_BYTES + Cfg::STAGE_BYTES) + wave * 64;
    if (p.stat_part) dacc[lane] = 0.0;
    int it = 0;
    __syncthreads();
    for (int tile = lb; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int g = 0; g < p.G; ++g) {
            compute(g, smem + (it & 1) * Cfg::STRIP_BYTES);
            __syncthreads();
            ++it;
        }

        // epilogue.  C/D layout: col = lane&31 = (frame parity, joint), row = (r&3) + 8*(r>>2) + 4*(lane>>5); a lane
        // holds one float of a (channel, frame) row and 4-byte stores are issue-bound, so each column tile goes
        // through this wave's LDS stage [32 rows][32 positions] and leaves (the residual arrives) as float4
        const int n = tile / p.nRB;
        const int rho0 = (tile % p.nRB) * Cfg::RB;
        float* stg = reinterpret_cast<float*>(smem + 2 * Cfg::STRIP_BYTES) + wave * 1024;
        const int lch = lane >> 3, lq = lane & 7;
        const int fp = lq >> 2, v0 = (lq & 3) * 4;
        float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[i][r];
            const int rho = rho0 + 2 * (wn * NT + i) + fp;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int chl = k * 8 + lch;
                const int m = mw * 32 + chl;
                const int ph = m / p.Cpad;
                const int c = m - ph * p.Cpad;
                const int t = p.PH * rho + ph;
                if (c < p.Cout && ph < p.PH && rho < p.R && t < p.Tout) {
                    float4 v = *reinterpret_cast<const float4*>(stg + chl * 32 + lq * 4);
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
        if (p.stat_part) {
            auto rsum8 = [](float v) {            // every lane gets the sum of its group of 8 (one channel row)
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
                return v;
            };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = rsum8(st1[k]), b = rsum8(st2[k]);
                if (lq == 0) {
                    dacc[(k * 8 + lch) * 2] += (double)a;
                    dacc[(k * 8 + lch) * 2 + 1] += (double)b;
                }
            }
        }
    }
    if (p.stat_part) {
        // forward pack modes only (PH == 1: GEMM row = output channel); row of this wave: (block, column group)
        const int c = mw * 32 + (lane >> 1);
        if (c < p.Cout) p.stat_part[((size_t)(blockIdx.x * NW + wn) * p.Cout + c) * 2 + (lane & 1)] = dacc[lane];
    }
}

// A fragments: [group g][tap jj][M tile][lane = 32*h + r][8] = wgen[16g + 8h + e][jj][32*mt + r]
// split: [group of 8 channels][tap][set][M tile][lane][8], k = 8h + e = 2*channel + part:
//   set 0 = w_hi for both parts, set 1 = w_lo for part 0 and 0 for part 1
__global__ void tcn_pack_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int Co,
                                     int Ci, int Kt, int stride, int mode, PackGeom g, int split) {
    const int mtiles = g.ph * g.cpad / 32;
    const int kt2 = 2 * g.ks;
    const int ns = split ? 2 : 1;
    const int total = (g.cin_g / (split ? 8 : 16)) * kt2 * ns * mtiles * 512;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int el = e & 7;
        const int lane = (e >> 3) & 63;
        const int mt = (e >> 9) % mtiles;
        const int set = ((e >> 9) / mtiles) % ns;
        const int jj = ((e >> 9) / (mtiles * ns)) % kt2;
        const int grp = (e >> 9) / (mtiles * ns * kt2);
        const int k = 8 * (lane >> 5) + el;
        const int m = 32 * mt + (lane & 31);
        if (!split) {
            wp[e] = (__bf16)pack_value(w, Co, Ci, Kt, stride, mode, g, 16 * grp + k, jj, m);
        } else {
            const float v = pack_value(w, Co, Ci, Kt, stride, mode, g, 8 * grp + (k >> 1), jj, m);
            const __bf16 hi = (__bf16)v;
            const __bf16 lo = (__bf16)(v - (float)hi);
            wp[e] = set == 0 ? hi : ((k & 1) == 0 ? lo : (__bf16)0.f);
        }
    }
}

bool bf16_supported(const PackGeom& g, int mode) {
    if (mode < 0 || mode > 3) return false;          // (1 / 3: ConvTranspose2d, stride 1 - same geometry as 0 / 2 there)
    if (g.cin_g % 16 != 0) return false;
    const int mw = g.ph * g.cpad / 32;
    const int key = (2 * g.ks) * 100 + g.stride_g * 10 + mw;
    switch (key) {
        case 7611: case 7612: case 7622: case 3814: return true;
        default: return false;
    }
}

template <int NT, int KT2, int STRIDE, int MW, int NW, bool SPLIT>
int launch_tb(const TbParams& p, hipStream_t st, int* nrows) {
    using Cfg = TbCfg<NT, KT2, STRIDE, MW, NW, SPLIT>;
    TbParams q = p;
    q.nRB = step::ceil_div(p.R, Cfg::RB);
    const long long tiles = (long long)p.N * q.nRB;
    if (tiles <= 0 || tiles > 0x7fffffffLL) {
        step::set_error("step_tcn_conv_bf16: %lld tiles out of range", tiles);
        return STEP_ERR_ARG;
    }
    auto kern = tcn_bf16_kernel<NT, KT2, STRIDE, MW, NW, SPLIT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    int dev = 0, cus = 0;
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cus < 1) {
        step::set_error("step_tcn_conv_bf16: launch set-up (%d bytes of LDS): %s", Cfg::LDS_BYTES,
                        hipGetErrorString(e));
        return STEP_ERR_LAUNCH;
    }
    const unsigned blocks = tiles < cus ? (unsigned)tiles : (unsigned)cus;   // one block per CU (LDS)
    if (nrows) {
        if ((int)blocks * NW > *nrows) {
            step::set_error("step_tcn_conv_stats_bf16: %u x %d partial rows > %d", blocks, NW, *nrows);
            return STEP_ERR_WORKSPACE;
        }
        *nrows = (int)blocks * NW;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(Cfg::NTHR), Cfg::LDS_BYTES, st, q);
    return step::check_launch("step_tcn_conv_bf16");
}

}  // namespace

namespace {

size_t packed_bytes(int Co, int Ci, int Kt, int stride, int pack_mode, int split) {
    PackGeom g;
    if (Co < 1 || Ci < 1 || !pack_geom(Co, Ci, Kt, stride, pack_mode, &g)) return 0;
    if (!bf16_supported(g, pack_mode)) return 0;
    return (size_t)(g.cin_g / (split ? 8 : 16)) * 2 * g.ks * (split ? 2 : 1) * (g.ph * g.cpad / 32) * 512 * 2;
}

int pack_weight(const float* w, void* wp, int Co, int Ci, int Kt, int stride, int pack_mode, int split,
                void* stream) {
    PackGeom g;
    STEP_REQUIRE(w && wp, "step_tcn_pack_weight_bf16: null pointer");
    STEP_REQUIRE(Co >= 1 && Ci >= 1 && pack_geom(Co, Ci, Kt, stride, pack_mode, &g) &&
                     bf16_supported(g, pack_mode),
                 "step_tcn_pack_weight_bf16: unsupported Co=%d Ci=%d Kt=%d stride=%d mode=%d", Co, Ci, Kt,
                 stride, pack_mode);
    const int total = (int)(packed_bytes(Co, Ci, Kt, stride, pack_mode, split) / 2);
    const int blocks = step::ceil_div(total, 256) < 1024 ? step::ceil_div(total, 256) : 1024;
    hipLaunchKernelGGL(tcn_pack_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<__bf16*>(wp), Co, Ci, Kt, stride, pack_mode, g, split);
    return step::check_launch("step_tcn_pack_weight_bf16");
}

template <bool SPLIT>
int conv(const float* in, const float* in_scale, const float* in_shift, int in_relu, const void* wp,
         const float* bias, const float* out_scale, const float* out_shift, const float* res,
         const float* res_scale, const float* res_shift, int out_relu, float* out, int N, int Ci, int Co,
         int Tin, int Tout, int Kt, int stride, int pack_mode, void* stream, double* stat_part = nullptr,
         int* nrows = nullptr) {
    PackGeom g;
    STEP_REQUIRE(in && wp && out, "step_tcn_conv_bf16: null pointer");
    STEP_REQUIRE(N >= 1 && Ci >= 1 && Co >= 1 && Tin >= 1 && Tout >= 1,
                 "step_tcn_conv_bf16: bad shape N=%d Ci=%d Co=%d Tin=%d Tout=%d", N, Ci, Co, Tin, Tout);
    STEP_REQUIRE(pack_geom(Co, Ci, Kt, stride, pack_mode, &g) && bf16_supported(g, pack_mode),
                 "step_tcn_conv_bf16: unsupported Co=%d Ci=%d Kt=%d stride=%d mode=%d", Co, Ci, Kt, stride,
                 pack_mode);
    STEP_REQUIRE(Tout == (Tin - 1) / stride + 1,
                 "step_tcn_conv_bf16: Tout=%d != (Tin-1)/stride+1 (Tin=%d)", Tout, Tin);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) &&
                     (out_scale == nullptr) == (out_shift == nullptr) &&
                     (res_scale == nullptr) == (res_shift == nullptr),
                 "step_tcn_conv_bf16: scale/shift must come in pairs");
    TbParams p{};
    p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.in_relu = in_relu;
    p.wp = reinterpret_cast<const bf16x8*>(wp); p.bias = bias; p.out_scale = out_scale;
    p.out_shift = out_shift; p.res = res; p.res_scale = res_scale; p.res_shift = res_shift;
    p.stat_part = stat_part;
    STEP_REQUIRE(!stat_part || (pack_mode < 2 && g.ph == 1), "step_tcn_conv_stats_bf16: forward pack modes only");
    p.out_relu = out_relu; p.out = out; p.N = N; p.Cin = g.cin_g; p.Cout = g.cout_g;
    p.Cpad = g.cpad; p.PH = g.ph; p.padL = g.padL; p.G = g.cin_g / (SPLIT ? 8 : 16);
    const bool dgrad = pack_mode >= 2;
    p.Tin = dgrad ? Tout : Tin;                 // frames of the tensor being read
    p.Tout = dgrad ? Tin : Tout;                // frames of the tensor being written
    p.R = step::ceil_div(p.Tout, g.ph);
    const int mw = g.ph * g.cpad / 32;
    const int key = (2 * g.ks) * 100 + g.stride_g * 10 + mw;
    hipStream_t st = (hipStream_t)stream;
    switch (key) {
        case 7611: return launch_tb<6, 76, 1, 1, 4, SPLIT>(p, st, nrows);
        case 7612: return launch_tb<6, 76, 1, 2, 2, SPLIT>(p, st, nrows);
        case 7622: return launch_tb<5, 76, 2, 2, 2, SPLIT>(p, st, nrows);
        case 3814: return launch_tb<6, 38, 1, 4, 1, SPLIT>(p, st, nrows);
        default: break;
    }
    step::set_error("step_tcn_conv_bf16: no kernel for Kt=%d stride=%d channels=%d", Kt, stride, g.cout_g);
    return STEP_ERR_ARG;
}

}  // namespace

extern "C" size_t step_tcn_packed_bytes_bf16(int Co, int Ci, int Kt, int stride, int pack_mode) {
    return packed_bytes(Co, Ci, Kt, stride, pack_mode, 0);
}
extern "C" size_t step_tcn_packed_bytes_split(int Co, int Ci, int Kt, int stride, int pack_mode) {
    return packed_bytes(Co, Ci, Kt, stride, pack_mode, 1);
}

extern "C" int step_tcn_pack_weight_bf16(const float* w, void* wp, int Co, int Ci, int Kt, int stride,
                                         int pack_mode, void* stream) {
    return pack_weight(w, wp, Co, Ci, Kt, stride, pack_mode, 0, stream);
}
extern "C" int step_tcn_pack_weight_split(const float* w, void* wp, int Co, int Ci, int Kt, int stride,
                                          int pack_mode, void* stream) {
    return pack_weight(w, wp, Co, Ci, Kt, stride, pack_mode, 1, stream);
}

extern "C" int step_tcn_conv_bf16(const float* in, const float* in_scale, const float* in_shift,
                                  int in_relu, const void* wp, const float* bias,
                                  const float* out_scale, const float* out_shift, const float* res,
                                  const float* res_scale, const float* res_shift, int out_relu,
                                  float* out, int N, int Ci, int Co, int Tin, int Tout, int Kt,
                                  int stride, int pack_mode, void* stream) {
    return conv<false>(in, in_scale, in_shift, in_relu, wp, bias, out_scale, out_shift, res, res_scale, res_shift,
                       out_relu, out, N, Ci, Co, Tin, Tout, Kt, stride, pack_mode, stream);
}
// step_tcn_conv_stats on the bf16 matrix cores (split == 0: bf16-rounded operands, else fp32 operands as hi + lo)
constexpr int STAT_ROWS = 1024;         // >= blocks (one per CU) x column groups (<= 4)
extern "C" size_t step_tcn_conv_stats_bf16_ws_bytes(int Co) {
    return Co < 1 ? 0 : (size_t)STAT_ROWS * Co * 2 * sizeof(double);
}
extern "C" int step_tcn_conv_stats_bf16(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                                        const void* wp, const float* bias, float* out, int N, int Ci, int Co, int Tin,
                                        int Tout, int Kt, int stride, int pack_mode, int split, double* sums, void* ws,
                                        size_t ws_bytes, void* stream) {
    STEP_REQUIRE(sums && ws, "step_tcn_conv_stats_bf16: null pointer");
    STEP_REQUIRE(pack_mode == 0 || pack_mode == 1, "step_tcn_conv_stats_bf16: forward pack modes only (got %d)", pack_mode);
    if (ws_bytes < step_tcn_conv_stats_bf16_ws_bytes(Co)) {
        step::set_error("step_tcn_conv_stats_bf16: workspace %zu < %zu bytes", ws_bytes, step_tcn_conv_stats_bf16_ws_bytes(Co));
        return STEP_ERR_WORKSPACE;
    }
    int rows = STAT_ROWS;
    const int rc = split ? conv<true>(in, in_scale, in_shift, in_relu, wp, bias, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                                      out, N, Ci, Co, Tin, Tout, Kt, stride, pack_mode, stream, (double*)ws, &rows)
                         : conv<false>(in, in_scale, in_shift, in_relu, wp, bias, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                                       out, N, Ci, Co, Tin, Tout, Kt, stride, pack_mode, stream, (double*)ws, &rows);
    if (rc != STEP_OK) return rc;
    return step::bn_sum_partials((const double*)ws, sums, Co, rows, (hipStream_t)stream);
}

extern "C" int step_tcn_conv_split(const float* in, const float* in_scale, const float* in_shift,
                                   int in_relu, const void* wp, const float* bias,
                                   const float* out_scale, const float* out_shift, const float* res,
                                   const float* res_scale, const float* res_shift, int out_relu,
                                   float* out, int N, int Ci, int Co, int Tin, int Tout, int Kt,
                                   int stride, int pack_mode, void* stream) {
    return conv<true>(in, in_scale, in_shift, in_relu, wp, bias, out_scale, out_shift, res, res_scale, res_shift,
                      out_relu, out, N, Ci, Co, Tin, Tout, Kt, stride, pack_mode, stream);
}
