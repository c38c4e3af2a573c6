// Backward of the fused graph convolution (gcn.hip).
//
// Replaces what autograd runs for ConvTemporalGraphical.forward (tgcn.py:58-67, :121-130)
// and the residual 1x1 strided conv (classifier.py:175-179): conv dgrad/wgrad, the two bmm's
// of the einsum backward and the permute copies around them.
//
// With  y = Wg * z + bterm,  z[(k,ci),(t,w)] = SUM_v x[ci,t,v] A[k,v,w]  (see gcn.hip):
//   dz[(k,ci),(t,w)] = SUM_c W[k*Co+c,ci] dy[c,(t,w)]                        (MFMA, kernel 1)
//   dx[ci,t,v]       = SUM_k SUM_w A[k,v,w] dz[(k,ci),(t,w)]  (+ Wr^T dres)   (LDS scatter, kernel 1)
//   dA[k,v,w]        = SUM_{n,ci,t} x[ci,t,v] dz[(k,ci),(t,w)] + SUM_c b[k*Co+c] S[c,w]   on supp(A)
//   dW[k*Co+c,ci]    = SUM_{n,t,w} dy[c,(t,w)] z[(k,ci),(t,w)]               (MFMA, kernel 2)
//   db[k*Co+c]       = SUM_w colsum_k[w] S[c,w],   S[c,w] = SUM_{n,t} dy[n,c,t,w]
// Both kernels are persistent over (n, 8-frame) tiles and keep their reductions in registers;
// per-block partials are summed in block order by a finalize kernel (bit-reproducible).
#include "common.h"
#include "split_mfma.h"
#include <type_traits>

namespace {

constexpr int TT = 8;
constexpr int P = TT * 16;        // 128 positions per tile
constexpr int PS = P + 1;         // odd LDS row stride (kernel 2)

struct GbParams {
    const float* dy;
    const float* dres;
    const float* x;
    const float* in_scale;
    const float* in_shift;
    const float* w;
    const float* b;
    const float* A;
    const float* res_w;
    float* dx;
    const float* id_dout;   // identity residual of the block (Ci == Co, stride 1): dx += id_dout * [id_out > 0]
    const float* id_out;
    float* part1;     // kernel 1 partials: [nblk][3][EMAX][16]
    float* part2;     // kernel 2 partials: [nblk][3*COPAD*CIP + COPAD*16 + COPAD*CIP + COPAD]
    int w_transposed, res_transposed;
    int N, Ci, Co, K, T, To, stride, tiles_per_seq, ntiles;
};

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float4 load_x4(const GbParams& p, int n, int ci, int t, int vq) {
    float4 v = *reinterpret_cast<const float4*>(p.x + (((size_t)n * p.Ci + ci) * p.T + t) * 16 + vq * 4);
    if (p.in_scale) {
        const int g = vq * 4 * p.Ci + ci;
        v.x = v.x * p.in_scale[g] + p.in_shift[g];
        v.y = v.y * p.in_scale[g + p.Ci] + p.in_shift[g + p.Ci];
        v.z = v.z * p.in_scale[g + 2 * p.Ci] + p.in_shift[g + 2 * p.Ci];
        v.w = v.w * p.in_scale[g + 3 * p.Ci] + p.in_shift[g + 3 * p.Ci];
    }
    return v;
}

// ------------------------------------------------------------------ kernel 1: dx, dA
// dz stays in the MFMA accumulators (column = (t, w) on the lane, row = (k, ci) in the
// register).
//  dx needs SUM_w A[k,v,w] dz[.., w], a 16 x 16 product along the LANE axis inside each frame's
//  16 lanes: it is taken as a sum over the 16 rotations of a DPP row (row_ror), each lane
//  holding the coefficient A[k][its v][the w that rotation brings]; rotations whose coefficient
//  vector is all zero (most of them: the skeleton's A has 46 non-zeros) are skipped with a
//  block-uniform branch.  One VALU instruction per (register, live rotation), no LDS traffic,
//  and dx comes out in the accumulator layout (128-byte rows to HBM).
//  dA[k] = SUM x[ci,t,v] dz[(k,ci),(t,w)] is a (16 x K') x (K' x 16) product with K' = (ci, t):
//  for one accumulator register the lane's (half, frame) pair is exactly the k index of
//  v_mfma_f32_16x16x4_f32's B operand, so dz feeds it as it lies and the A operand is the
//  lane's own x element; the dense 16 x 16 dA[k] accumulates in 4 registers per k.
template <int R>
__device__ __forceinline__ float row_ror(float x) {
    if (R == 0) return x;
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x120 + (R & 15), 0xF, 0xF, true));
}

template <int R, int END, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (R < END) {
        f(std::integral_constant<int, R>{});
        static_for<R + 1, END>(f);
    }
}

template <int CIP, int COPAD, bool RES>
__global__ __launch_bounds__(256, CIP <= 8 ? 2 : 1) void gcn_bwd_data_kernel(GbParams p) {
    constexpr int KQ = 3 * CIP;
    constexpr int MQT = (KQ + 31) / 32;
    constexpr int KQP = MQT * 32;
    constexpr int MCI = (CIP + 31) / 32;            // ci tiles of the residual product
    constexpr int CIRP = MCI * 32;
    constexpr int NDX = CIP >= 32 ? (CIP / 32) * 16 : 4;   // dx registers per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wq = smem;                                // [COPAD][KQP]   W[(k,c),ci] as [c][k*CIP+ci]
    float* dys = Wq + COPAD * KQP;                   // [COPAD][P]     dy tile, later the dres tile
    float* xs = dys + COPAD * P;                     // [CIP][P]       x tile, later Wr^T dres
    float* cfs = xs + CIP * P;                       // [3][16 rotations][16 joints]
    float* Wrt = cfs + 768;                          // [COPAD][CIRP]  residual conv weight (RES)
    float* drs = Wrt + COPAD * CIRP;                 // [COPAD][DW]    dres tile (RES), DW = 128 / stride
    float* red = dys;                                // [4][3*16*16], aliases dys after the loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5, w16 = lane & 15;
    const int trow = 2 * wave + (l31 >> 4);

    for (int e = tid; e < COPAD * KQP; e += 256) {
        const int q = e % KQP, c = e / KQP;
        const int k = q / CIP, ci = q % CIP;
        float v = 0.f;
        if (q < KQ && k < p.K && c < p.Co && ci < p.Ci)
            v = p.w_transposed ? p.w[(size_t)ci * (p.K * p.Co) + k * p.Co + c]
                               : p.w[((size_t)k * p.Co + c) * p.Ci + ci];
        Wq[e] = v;
    }

    const int DW = p.stride == 1 ? P : P / 2;        // dres frames under a tile: 8 or 4
    if (RES) {
        for (int e = tid; e < COPAD * CIRP; e += 256) {
            const int ci = e % CIRP, c = e / CIRP;
            float v = 0.f;
            if (c < p.Co && ci < p.Ci)
                v = p.res_transposed ? p.res_w[(size_t)ci * p.Co + c] : p.res_w[(size_t)c * p.Ci + ci];
            Wrt[e] = v;
        }
    }

    // rotation coefficients: cf[k][R] = A[k][v = this lane's joint][w = joint of the lane that
    // row_ror:R brings here] (the source lane is found by rotating the lane ids themselves)
    unsigned rmask[3] = {0u, 0u, 0u};
    static_for<0, 16>([&](auto rt) {
        constexpr int R = decltype(rt)::value;
        const int src = __float_as_int(row_ror<R>(__int_as_float(w16)));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = (k < p.K) ? p.A[(k * 16 + w16) * 16 + src] : 0.f;
            if (tid < 16) cfs[(k * 16 + R) * 16 + w16] = a;
            if (__ballot(a != 0.f) != 0ull) rmask[k] |= 1u << R;
        }
    });
#pragma unroll
    for (int k = 0; k < 3; ++k) rmask[k] = __builtin_amdgcn_readfirstlane(rmask[k]);

    f32x4 dAk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dAk[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // global -> registers a tile ahead (raw loads), so HBM latency hides under the MFMA + epilogue
    // of the tile before; data_bn's scale/shift is applied when the registers go to LDS
    constexpr int DYN = COPAD * 32 / 256, XN = (CIP * 32 + 255) / 256;
    constexpr int DRN = RES ? DYN : 1;
    float4 dyr[DYN], xr4[XN], drr[DRN];
    auto load_tile = [&](int tile) {
        const int n = tile / p.tiles_per_seq;
        const int t0 = (tile % p.tiles_per_seq) * TT;
        if (RES) {
            const int q4 = DW / 4;                   // float4 per channel row of the dres tile
            const int tr0 = t0 / p.stride;
#pragma unroll
            for (int i = 0; i < DRN; ++i) {
                const int f = tid + i * 256;
                const int c = f / q4, r4 = f - c * q4;
                const int to = tr0 + (r4 >> 2);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < COPAD * q4 && c < p.Co && to < p.To)
                    v = *reinterpret_cast<const float4*>(
                        p.dres + (((size_t)n * p.Co + c) * p.To + to) * 16 + (r4 & 3) * 4);
                drr[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < DYN; ++i) {
            const int f = tid + i * 256;
            const int c = f >> 5, row = (f & 31) >> 2, vq = f & 3;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < p.Co && t0 + row < p.T)
                v = *reinterpret_cast<const float4*>(
                    p.dy + (((size_t)n * p.Co + c) * p.T + t0 + row) * 16 + vq * 4);
            dyr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int f = tid + i * 256;
            const int ci = f >> 5, row = (f & 31) >> 2, vq = f & 3;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < CIP * 32 && ci < p.Ci && t0 + row < p.T)
                v = *reinterpret_cast<const float4*>(
                    p.x + (((size_t)n * p.Ci + ci) * p.T + t0 + row) * 16 + vq * 4);
            xr4[i] = v;
        }
    };

    if ((int)blockIdx.x < p.ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int n = tile / p.tiles_per_seq;
        const int t0 = (tile % p.tiles_per_seq) * TT;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DYN; ++i) {
            const int f = tid + i * 256;
            *reinterpret_cast<float4*>(dys + (f >> 5) * P + (f & 31) * 4) = dyr[i];
        }
        if (RES) {
#pragma unroll
            for (int i = 0; i < DRN; ++i) {
                const int f = tid + i * 256;
                if (f < COPAD * (DW / 4)) *reinterpret_cast<float4*>(drs + f * 4) = drr[i];
            }
        }
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int f = tid + i * 256;
            if (f < CIP * 32) {
                const int ci = f >> 5, row = (f & 31) >> 2, vq = f & 3;
                float4 v = xr4[i];
                if (p.in_scale && ci < p.Ci && t0 + row < p.T) {
                    const int g = vq * 4 * p.Ci + ci;
                    v.x = v.x * p.in_scale[g] + p.in_shift[g];
                    v.y = v.y * p.in_scale[g + p.Ci] + p.in_shift[g + p.Ci];
                    v.z = v.z * p.in_scale[g + 2 * p.Ci] + p.in_shift[g + 2 * p.Ci];
                    v.w = v.w * p.in_scale[g + 3 * p.Ci] + p.in_shift[g + 3 * p.Ci];
                }
                *reinterpret_cast<float4*>(xs + ci * P + (f & 31) * 4) = v;
            }
        }
        if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x);
        __syncthreads();

        // dz = Wq^T . dy : rows q = (k,ci), columns = this wave's 32 positions
        f32x16 acc[MQT];
#pragma unroll
        for (int m = 0; m < MQT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        {
            const float* ap = Wq + half * KQP + l31;
            const float* bp = dys + half * P + wave * 32 + l31;
            for (int s = 0; s < COPAD / 2; ++s) {
                const float bv = bp[s * 2 * P];
#pragma unroll
                for (int m = 0; m < MQT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s * 2 * KQP + m * 32], bv, acc[m], 0, 0, 0);
            }
        }
        float dxr[NDX];
#pragma unroll
        for (int i = 0; i < NDX; ++i) dxr[i] = 0.f;
        // register (m, r) holds row rr + 4*half of M tile m: partition kq, channel ci
        static_for<0, 3>([&](auto kt) {
            constexpr int kq = decltype(kt)::value;
            // dA[kq] += x^T dz  (16x16x4: k index = the lane's (half, frame))
#pragma unroll
            for (int m = 0; m < MQT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    if ((m * 32 + rr) / CIP == kq) {
                        const int ci = m * 32 + rr + 4 * half - kq * CIP;
                        dAk[kq] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[ci * P + (trow & 7) * 16 + w16],
                                                                       acc[m][r], dAk[kq], 0, 0, 0);
                    }
                }
            // dx += A_kq dz along the lanes of each frame
            static_for<0, 16>([&](auto rt) {
                constexpr int R = decltype(rt)::value;
                if ((rmask[kq] >> R) & 1u) {
                    const float c = cfs[(kq * 16 + R) * 16 + w16];
#pragma unroll
                    for (int m = 0; m < MQT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rr = (r & 3) + 8 * (r >> 2);
                            if ((m * 32 + rr) / CIP == kq) {
                                const int slot = CIP >= 32 ? ((m * 32) % CIP) / 32 * 16 + r : (r & 3);
                                dxr[slot] = fmaf(c, row_ror<R>(acc[m][r]), dxr[slot]);
                            }
                        }
                }
            });
        });

        // residual branch: dx[ci, s*t', w] += SUM_c Wr[c,ci] dres[c, t', w].  Its product has the
        // layout of dz (row = ci in the register, column = (t, w) on the lane), so it is one more
        // MFMA chain on the same columns and adds straight into dx; at stride 2 only the even
        // frames of the tile receive a contribution.
        if (RES) {
            f32x16 ar[MCI];
#pragma unroll
            for (int m = 0; m < MCI; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) ar[m][r] = 0.f;
            const bool mine = p.stride == 1 || (trow & 1) == 0;
            const int col = p.stride == 1 ? trow * 16 + w16 : (trow >> 1) * 16 + w16;
            const float* bp = drs + half * DW + col;
            const float* ap = Wrt + half * CIRP + l31;
            for (int s = 0; s < COPAD / 2; ++s) {
                const float bv = mine ? bp[s * 2 * DW] : 0.f;
#pragma unroll
                for (int m = 0; m < MCI; ++m)
                    ar[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[s * 2 * CIRP + m * 32], bv, ar[m], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NDX; ++i) dxr[i] += CIP >= 32 ? ar[i / 16][i & 15] : ar[0][i];
        }
        if (p.dx && t0 + trow < p.T) {
#pragma unroll
            for (int i = 0; i < NDX; ++i) {
                const int ci = CIP >= 32 ? (i / 16) * 32 + rowmap(i & 15, half) : i + 4 * half;
                if (ci < p.Ci)
                    p.dx[(((size_t)n * p.Ci + ci) * p.T + t0 + trow) * 16 + w16] = dxr[i];
            }
        }
    }

    // block partial of dA (dense 16 x 16 per k): D register i of lane l = dA[k][4*(l>>4)+i][l&15]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            red[wave * 768 + (k * 16 + 4 * (lane >> 4) + i) * 16 + w16] = dAk[k][i];
    __syncthreads();
    for (int e = tid; e < 768; e += 256)
        p.part1[(size_t)blockIdx.x * 768 + e] = (red[e] + red[768 + e]) + (red[2 * 768 + e] + red[3 * 768 + e]);
}

// -------------------------------------------------------------- kernel 2: dW, S, dWr, dbr
// dW_k[c][ci] = SUM_p dy[c][p] z_k[ci][p],  z_k[ci,(t,w)] = SUM_v x[ci,t,v] A[k,v,w].
// Per tile of 8 frames: every thread takes x elements straight from HBM with its lane on the
// joint axis, forms z_0, z_1, z_2 with DPP row rotations (the column twin of kernel 1's
// coefficients) and writes them to three LDS images; the dy tile goes to LDS next to them (its
// per-(c,w) frame sums S stay in registers on the way).  One barrier later the MFMA waves run
// all three products from one A read per k-step.  Two barriers per tile (four with a residual
// conv); when fewer than four 32x32 output tiles exist the waves split the position axis.
__device__ __forceinline__ float load_x1(const GbParams& p, int n, int ci, int t, int v) {
    float x = p.x[(((size_t)n * p.Ci + ci) * p.T + t) * 16 + v];
    if (p.in_scale) x = x * p.in_scale[v * p.Ci + ci] + p.in_shift[v * p.Ci + ci];
    return x;
}

template <int CIP, int COPAD>
__global__ __launch_bounds__(256, CIP <= 8 ? 2 : 1) void gcn_bwd_weight_kernel(GbParams p) {
    constexpr int MC = COPAD / 32;                   // c tiles
    constexpr int NCI = (CIP + 31) / 32;             // ci tiles
    constexpr int CIR = NCI * 32;                    // padded ci rows of the z images
    constexpr int NTILE = MC * NCI;                  // output tiles per k (1, 2 or 4)
    constexpr int KSPLIT = 4 / NTILE;                // waves sharing one output tile
    constexpr int SSTEPS = (P / 2) / KSPLIT;         // k-steps per wave
    constexpr int PART = 3 * COPAD * CIR + COPAD * 16 + COPAD * CIR + COPAD;
    constexpr int CPT = (CIP + 1) / 2;               // z rows per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dys = smem;                               // [COPAD][PS]
    float* zk = dys + COPAD * PS;                    // [3][CIR][PS]; zk[0] doubles as the x tile
    float* czs = zk + 3 * CIR * PS;                  // [3][16 rotations][16 joints]
    // MFMA aggregation (CIP >= 32): the x tile lies in LDS as [ci][PS]; at 32 channels in a region of its own,
    // at 64 in the z_2 image (LDS is full), which is why the z fragments then wait in registers for a barrier
    constexpr bool ZMFMA = CIP >= 32;
    constexpr bool XT_OWN = CIP == 32;
    float* xt = XT_OWN ? czs + 768 : zk + 2 * CIR * PS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5, w16 = lane & 15;
    const int tile_id = wave % NTILE, ks = wave / NTILE;
    const int mt = tile_id / NCI, nt = tile_id % NCI;

    // B operand of the aggregation MFMA (v_mfma_f32_16x16x4_f32): A_k[v = 4*c4 + (lane >> 4)][w = lane & 15]
    float areg[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
            areg[k][c4] = (ZMFMA && k < p.K) ? p.A[(k * 16 + 4 * c4 + (lane >> 4)) * 16 + w16] : 0.f;

    // column coefficients: cz[k][R] = A[k][v = joint of the lane row_ror:R brings][w = own joint]
    unsigned zmask[3] = {0u, 0u, 0u};
    static_for<0, 16>([&](auto rt) {
        constexpr int R = decltype(rt)::value;
        const int src = __float_as_int(row_ror<R>(__int_as_float(w16)));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = (k < p.K) ? p.A[(k * 16 + src) * 16 + w16] : 0.f;
            if (tid < 16) czs[(k * 16 + R) * 16 + w16] = a;
            if (__ballot(a != 0.f) != 0ull) zmask[k] |= 1u << R;
        }
    });
#pragma unroll
    for (int k = 0; k < 3; ++k) zmask[k] = __builtin_amdgcn_readfirstlane(zmask[k]);

    for (int e = tid; e < 3 * CIR * PS; e += 256) zk[e] = 0.f;

    f32x16 accw[3];
    f32x16 accr;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accw[0][r] = 0.f; accw[1][r] = 0.f; accw[2][r] = 0.f; accr[r] = 0.f; }
    float4 Sacc = make_float4(0.f, 0.f, 0.f, 0.f);   // thread (c, vq): SUM_{n,t} dy[c][t][4vq..4vq+3]
    float4 Bacc = make_float4(0.f, 0.f, 0.f, 0.f);   // same for dres
    const int cdy = tid >> 2, vq = tid & 3;          // this thread's dy row segment
    const int pz = tid & 127;                        // position handled in the z pass
    const int tz = pz >> 4;

    // global -> registers for one tile (raw loads; issued a tile ahead so HBM latency hides under
    // the MFMA phase of the tile before)
    float xr[CPT];
    float4 dyr[TT], drr[TT];
    auto load_tile = [&](int tile) {
        const int n = tile / p.tiles_per_seq;
        const int t0 = (tile % p.tiles_per_seq) * TT;
        if (p.dres) {                                    // the dres frames under this tile, a tile ahead as well
            const int nfr = p.stride == 1 ? TT : TT / 2;
            const int tr0 = t0 / p.stride;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t < nfr && cdy < p.Co && tr0 + t < p.To)
                    v = *reinterpret_cast<const float4*>(
                        p.dres + (((size_t)n * p.Co + cdy) * p.To + tr0 + t) * 16 + vq * 4);
                drr[t] = v;
            }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cdy < p.Co && t0 + t < p.T)
                v = *reinterpret_cast<const float4*>(
                    p.dy + (((size_t)n * p.Co + cdy) * p.T + t0 + t) * 16 + vq * 4);
            dyr[t] = v;
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int ci = (tid >> 7) + 2 * i;
            float x = 0.f;
            if (ci < p.Ci && t0 + tz < p.T) x = p.x[(((size_t)n * p.Ci + ci) * p.T + t0 + tz) * 16 + w16];
            xr[i] = x;
        }
    };

    if ((int)blockIdx.x < p.ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int t0 = (tile % p.tiles_per_seq) * TT;
        const int next = tile + gridDim.x;
        __syncthreads();                                  // previous tile's MFMA reads are done
        if (cdy < COPAD) {
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const float4 v = dyr[t];
                Sacc.x += v.x; Sacc.y += v.y; Sacc.z += v.z; Sacc.w += v.w;
                float* d = dys + cdy * PS + t * 16 + vq * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        if (p.in_scale) {
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int ci = (tid >> 7) + 2 * i;
                if (ci < p.Ci && t0 + tz < p.T)
                    xr[i] = xr[i] * p.in_scale[w16 * p.Ci + ci] + p.in_shift[w16 * p.Ci + ci];
            }
        }
        if constexpr (ZMFMA) {
            // z_k = x A_k on the matrix cores: a 16-row z tile is 8 channels x 2 frames (row 4q + r = channel
            // 2r + (q >> 1), frame q & 1), k = 4 joints per instruction; lane 16q + w of result register r holds
            // z_k[channel 8g + 2r + (q >> 1)][frame 2fp + (q & 1)][w], written to the [ci][position] images the
            // main product reads.  A wave takes every fourth (channel group, frame pair).
            constexpr int NG = (CIP / 8) * 4 / 4;                 // groups per wave
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int ci = (tid >> 7) + 2 * i;
                if (ci < CIP) xt[ci * PS + pz] = xr[i];
            }
            __syncthreads();
            const int ai = lane & 15;
            const float* xa = xt + (2 * (ai & 3) + ((ai >> 3) & 1)) * PS + ((ai >> 2) & 1) * 16 + (lane >> 4);
            const int zq = lane >> 4;
            float* zw = zk + (zq >> 1) * PS + (zq & 1) * 16 + w16;
            f32x4 zr[NG][3];
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int grp = wave + 4 * gi, g = grp >> 2, fp = grp & 3;
                float a4[4];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) a4[c4] = xa[8 * g * PS + fp * 32 + c4 * 4];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    zr[gi][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4)
                        zr[gi][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[c4], areg[k][c4], zr[gi][k], 0, 0, 0);
                }
                if constexpr (XT_OWN) {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int r = 0; r < 4; ++r) zw[(k * CIR + 8 * g + 2 * r) * PS + fp * 32] = zr[gi][k][r];
                }
            }
            if constexpr (!XT_OWN) {
                __syncthreads();                                  // every wave has read the x tile out of z_2's image
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const int grp = wave + 4 * gi, g = grp >> 2, fp = grp & 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int r = 0; r < 4; ++r) zw[(k * CIR + 8 * g + 2 * r) * PS + fp * 32] = zr[gi][k][r];
                }
            }
        } else
        {
            // rotations outside, rows inside: one uniform branch per live (k, rotation) pair
            float z[3][CPT];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int i = 0; i < CPT; ++i) z[k][i] = 0.f;
            static_for<0, 16>([&](auto rt) {
                constexpr int R = decltype(rt)::value;
                static_for<0, 3>([&](auto kt) {
                    constexpr int k = decltype(kt)::value;
                    if ((zmask[k] >> R) & 1u) {
                        const float c = czs[(k * 16 + R) * 16 + w16];
#pragma unroll
                        for (int i = 0; i < CPT; ++i) z[k][i] = fmaf(c, row_ror<R>(xr[i]), z[k][i]);
                    }
                });
            });
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int ci = (tid >> 7) + 2 * i;
                if (ci < CIP) {
                    zk[ci * PS + pz] = z[0][i];
                    zk[(CIR + ci) * PS + pz] = z[1][i];
                    zk[(2 * CIR + ci) * PS + pz] = z[2][i];
                }
            }
        }
        if (!p.dres && next < p.ntiles) load_tile(next);
        __syncthreads();
        {
            const float* ap = dys + (mt * 32 + l31) * PS + half + 2 * ks * SSTEPS;
            const float* bp = zk + (nt * 32 + l31) * PS + half + 2 * ks * SSTEPS;
#pragma unroll 4
            for (int s = 0; s < SSTEPS; ++s) {
                const float a = ap[2 * s];
                accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[2 * s], accw[0], 0, 0, 0);
                accw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[CIR * PS + 2 * s], accw[1], 0, 0, 0);
                accw[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[2 * CIR * PS + 2 * s], accw[2], 0, 0, 0);
            }
        }
        if (p.dres) {
            // dWr[c,ci] += SUM_{t',w} dres[c,t',w] x[ci, s*t', w]: dres tile -> dys, x tile -> zk[0]
            const int nfr = p.stride == 1 ? TT : TT / 2;        // dres frames under this tile
            __syncthreads();
            if (cdy < COPAD) {
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const float4 v = drr[t];
                    Bacc.x += v.x; Bacc.y += v.y; Bacc.z += v.z; Bacc.w += v.w;
                    float* d = dys + cdy * PS + t * 16 + vq * 4;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
            }
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int ci = (tid >> 7) + 2 * i;
                if (ci < CIP) zk[ci * PS + pz] = xr[i];
            }
            if (next < p.ntiles) load_tile(next);
            __syncthreads();
            // frames past nfr of the dres tile are zero; k-steps are (frame tl, joint pair v2)
            const float* ap = dys + (mt * 32 + l31) * PS + half;
            const float* bp = zk + (nt * 32 + l31) * PS + half;
            for (int q = ks * SSTEPS; q < (ks + 1) * SSTEPS; ++q) {
                const int tl = q >> 3, v2 = q & 7;
                if (tl < nfr)
                    accr = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[tl * 16 + 2 * v2],
                                                                bp[p.stride * tl * 16 + 2 * v2], accr, 0, 0, 0);
            }
        }
    }

    // combine the KSPLIT waves of each output tile through LDS, then write the block partial
    float* part = p.part2 + (size_t)blockIdx.x * PART;
    __syncthreads();
    float* red = smem;                                   // [4 waves][4 tiles of 16 regs][64 lanes]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 64 + k * 16 + r) * 64 + lane] = accw[k][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 64 + 48 + r) * 64 + lane] = accr[r];
    __syncthreads();
    if (ks == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < KSPLIT; ++j) s += red[((tile_id + j * NTILE) * 64 + k * 16 + r) * 64 + lane];
                const int c = mt * 32 + rowmap(r, half);
                if (k < 3) part[(k * COPAD + c) * CIR + nt * 32 + l31] = s;
                else part[3 * COPAD * CIR + COPAD * 16 + c * CIR + nt * 32 + l31] = s;
            }
    }
    if (cdy < COPAD) {
        float* ps = part + 3 * COPAD * CIR + cdy * 16 + vq * 4;
        ps[0] = Sacc.x; ps[1] = Sacc.y; ps[2] = Sacc.z; ps[3] = Sacc.w;
        float b = (Bacc.x + Bacc.y) + (Bacc.z + Bacc.w);
        b += __shfl_xor(b, 1, 64);
        b += __shfl_xor(b, 2, 64);
        if (vq == 0) part[3 * COPAD * CIR + COPAD * 16 + COPAD * CIR + cdy] = b;
    }
}


// ==========================================================================================================
// Second form of the backward for the wide blocks (Ci, Co in {32, 64}; no data_bn on the load).
//
// Every product runs on v_mfma_f32_16x16x4_f32 with one (n, t) FRAME as the column tile (16 joints), and every
// operand reaches the matrix cores in the layout the producing instruction (a global load or an MFMA) left it in:
// no LDS tiles, no barriers inside the loops, waves are autonomous.  Two waves per SIMD (the kernels stay under
// 256 VGPRs), so one wave's VALU / load / store phase runs beside its partner's MFMAs.  Lane = 16 q + l.
//
// data kernel, wave = one frame at a time:
//   dz_k[ci,w]   = SUM_c W_k[c,ci] dy[c,t,w]        A[i=ci][q] = W_k[4s+q][ci] (LDS, read-only),  B[q][j=w] = dy[4s+q][t][w]
//                                                   (the lane's own 4-byte load); D register r of lane (q,w) = dz_k[4q+r][w]
//   dA_k[v,w]   += SUM_ci x[ci,t,v] dz_k[ci,w]      A[i=v][q] = x[4q+r][t][v] (own load),  B = D register r of dz as it lies
//   dx[ci,t,v]   = SUM_k SUM_w A_k[v,w] dz_k[ci,w]  along the 16 lanes of a row: DPP row rotations, live rotations only
//                  (+ SUM_c Wr[c,ci] dres[c,t',w]: one more MFMA chain with the dz layout)
// weight kernel, wave = (16 input channels, one frame at a time):
//   z_k[ci,w]    = SUM_v x[ci,t,v] A_k[v,w]         A[i=w][q] = A_k[4q+e][w] (registers),  B[q][j=ci] = component e of the
//                                                   lane's float4 x[ci][t][4q..4q+3];  D register r of lane (q,ci) = z_k[ci][4q+r]
//   dW_k[c,ci]  += SUM_w dy[c,t,w] z_k[ci,w]        A[i=c][q] = component r of the lane's float4 dy[c][t][4q..4q+3],  B = D register
//                                                   r of z_k as it lies;  dWr the same with x itself as B
// Per-block partials go through the same ordered finalize as the first form (bit-reproducible).
// Measured (N = 4096): block 1 (32 -> 64, stride-2 residual) 8.4 -> 6.0 ms, block 2 (64 -> 64) 6.5 -> 4.5 ms for the two
// kernels.  What bounds them now: in these kernels a v_mfma_f32_16x16x4_f32 takes ~38-41 cycles instead of its 32-cycle
// issue rate (MFMA-only ablation of the data kernel: 78 % of the pipe; a register-only probe of the same instruction
// reaches 32.0, tools/mfma_data_probe2.hip), every VALU instruction of a wave ADDS to its fp32 MFMA time
// (tools/mfma_shape_probe.hip: same wave additive, a partner wave's VALU does overlap), and the clock sits at
// 1.9-2.05 GHz while both kernels stream x and dy from HBM.  Staggering / prioritising the SIMD partners changed nothing.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// store target that may be absent: a NULL base gets num_records = 0, so the hardware drops every store (no branch
// around the stores - see the note on s_waitcnt at gcn_bwd_data16_kernel)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_opt(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, base ? 0x7ffffffc : 0, 0x00020000);
}

template <int CT>
__device__ __forceinline__ void lds_frag(const float* p, float (&out)[CT]) {
    if constexpr (CT == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
    } else {
        const f32x2 v = *reinterpret_cast<const f32x2*>(p);
        out[0] = v[0]; out[1] = v[1];
    }
}

// acc += c * (src rotated right by R lanes inside its row of 16) as ONE instruction.  hipcc leaves
// __builtin_amdgcn_mov_dpp + fmaf as v_mov_b32_dpp + (SLP-packed) v_pk_fma_f32 - twice the VALU work, and every VALU
// cycle of this wave adds to its fp32 MFMA time.  `first` pads the SALU-writes-EXEC -> DPP hazard (5 wait states)
// that the compiler does not see through an asm statement; the DPP source is an MFMA result that is dozens of MFMAs
// old by the time these run (the callers fence with sched_barrier).
template <int R>
__device__ __forceinline__ void fma_ror(float& acc, float c, float src, bool first) {
    if constexpr (R == 0) {
        acc = fmaf(c, src, acc);
    } else if (first) {
        asm volatile("s_nop 4\n\tv_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(acc) : "v"(src), "v"(c), "i"(R));
    } else {
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(acc) : "v"(src), "v"(c), "i"(R));
    }
}

// A note on s_waitcnt here (measured while building the split-operand twin below): vmcnt counts in order, holds 63 at
// most, and at a control-flow join the compiler waits for the path with the FEWEST newer operations - with the next
// row's dy and the identity residual's rows requested under `if`s, the dz chain of every other row starts with
// vmcnt(31 - s), i.e. it waits for loads it has just issued.  Making every request unconditional (IDRES as a template
// parameter, the last row prefetching itself, stores to a NULL dx dropped through num_records = 0) gives exact waits
// and changed this kernel by -6 % at 64 -> 64 and +5 % at 32 -> 64 (stride-2 residual): a wash, so this kernel keeps
// its form.  What bounds both forms is NOT settled.  The split twin does a quarter of the MFMA work in the same time
// (2.64 ms at 64 -> 64, N = 4096; TA busy 81 %, L2 hit rate 2.7 %, FETCH_SIZE 1.9x the algorithmic bytes, waves 43 % of
// their time in s_waitcnt), and none of these moved it: exact waits, whole-row prefetch, whole-line requests (frame
// pairs through v_permlane16_swap: -2 % / +10 % on this kernel), 16-byte accesses with an LDS transpose (a quarter of
// the vector-memory instructions: +2 %), store cache policies.  Removing a tensor does (0.75 ms for the identity
// residual's two rows, 1.05 ms for dx).  What is left untested is latency inside the wave (every MFMA pair of the
// twin waits for its own LDS read; two waves per SIMD): DESIGN section 9.
template <int CI, int CO, int RES>
__global__ __launch_bounds__(512, 2) void gcn_bwd_data16_kernel(GbParams p) {
    constexpr int CT = CI / 16;                       // 16-channel tiles of ci
    constexpr int S = CO / 4;                         // k-steps over c (4 output channels each)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                 // [3][S][64 lanes][CT]
    float* Wrl = Wl + 3 * S * 64 * CT;                // [S][64][CT]   (RES)
    float* cfs = Wrl + (RES ? S * 64 * CT : 0);       // [3][16 joints][16 rotations]
    float* red = smem;                                // [8][768], after the loop

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably uniform: rows, descriptors, offsets in SGPRs
    const int q = lane >> 4, l16 = lane & 15;

    for (int e = tid; e < 3 * S * 64 * CT; e += 512) {
        const int tile = e % CT, ln = (e / CT) & 63, s = (e / (CT * 64)) % S, k = e / (CT * 64 * S);
        const int c = 4 * s + (ln >> 4), ci = 16 * tile + (ln & 15);
        float v = 0.f;
        if (k < p.K)
            v = p.w_transposed ? p.w[(size_t)ci * (p.K * CO) + k * CO + c] : p.w[((size_t)k * CO + c) * CI + ci];
        Wl[e] = v;
    }
    if constexpr (RES != 0) {
        for (int e = tid; e < S * 64 * CT; e += 512) {
            const int tile = e % CT, ln = (e / CT) & 63, s = e / (CT * 64);
            const int c = 4 * s + (ln >> 4), ci = 16 * tile + (ln & 15);
            Wrl[e] = p.res_transposed ? p.res_w[(size_t)ci * CO + c] : p.res_w[(size_t)c * CI + ci];
        }
    }
    // rotation coefficients: cf[k][R] = A[k][v = this lane's joint][w = joint of the lane row_ror:R brings here]
    unsigned rmask[3] = {0u, 0u, 0u};
    static_for<0, 16>([&](auto rt) {
        constexpr int R = decltype(rt)::value;
        const int src = __float_as_int(row_ror<R>(__int_as_float(l16)));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = (k < p.K) ? p.A[(k * 16 + l16) * 16 + src] : 0.f;
            if (tid < 16) cfs[(k * 16 + l16) * 16 + R] = a;
            if (__ballot(a != 0.f) != 0ull) rmask[k] |= 1u << R;
        }
    });
#pragma unroll
    for (int k = 0; k < 3; ++k) rmask[k] = __builtin_amdgcn_readfirstlane(rmask[k]);
    __syncthreads();

    f32x4 dAk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dAk[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's contiguous run of frames
    const long long R = (long long)p.N * p.T;
    const long long nw = (long long)gridDim.x * 8, gw = (long long)blockIdx.x * 8 + wave;
    const long long r0 = R * gw / nw, r1 = R * (gw + 1) / nw;
    const int nrows = (int)(r1 - r0);

    // Addresses: per-sample buffer descriptors (SGPRs), ONE 32-bit lane offset per tensor, and the channel-step /
    // frame part as a uniform SGPR offset computed on the scalar unit.  Bytes.
    const unsigned TS = (unsigned)p.T * 64u, TOS = (unsigned)p.To * 64u;      // bytes per channel row
    const unsigned lane_dy = (unsigned)q * TS + l16 * 4u, lane_x = 4u * q * TS + l16 * 4u;
    const unsigned lane_dr = (unsigned)q * TOS + l16 * 4u;
    int n = (int)(r0 / p.T), t = (int)(r0 - (long long)n * p.T);

    float dyA[S], dyB[S];
    auto load_dy = [&](int nn, int tt, float (&dyv)[S]) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.dy + (size_t)nn * CO * p.T * 16);
        const unsigned so = (unsigned)tt * 64u;
#pragma unroll
        for (int s = 0; s < S; ++s) dyv[s] = bload(rs, lane_dy, so + (unsigned)s * 4u * TS);
    };
    auto do_row = [&](int nn, int tt, bool have_next, float (&dyc)[S], float (&dyn)[S]) {
        const bool res_row = RES == 1 || (RES == 2 && (tt & 1) == 0);
        float xc[CT * 4], drc[RES ? S : 1];
        {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.x + (size_t)nn * CI * p.T * 16);
            const unsigned so = (unsigned)tt * 64u;
#pragma unroll
            for (int i = 0; i < CT * 4; ++i) xc[i] = bload(rs, lane_x, so + (unsigned)((i >> 2) * 16 + (i & 3)) * TS);
        }
        if constexpr (RES != 0) {
            if (res_row) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.dres + (size_t)nn * CO * p.To * 16);
                const unsigned so = (unsigned)(RES == 1 ? tt : tt >> 1) * 64u;
#pragma unroll
                for (int s = 0; s < S; ++s) drc[s] = bload(rs, lane_dr, so + (unsigned)s * 4u * TOS);
            }
        }
        // the block's identity residual (out = relu(.. + x)): its gradient dout * [out > 0] has dx's layout; the two
        // rows are requested with the row's other loads (HBM latency is several times the row's MFMA time)
        float idg[RES == 0 ? CT * 4 : 1], ido[RES == 0 ? CT * 4 : 1];
        const bool idres = RES == 0 && p.dx && p.id_dout != nullptr;
        if constexpr (RES == 0) {
            if (idres) {
                const __amdgpu_buffer_rsrc_t rg = make_rsrc(p.id_dout + (size_t)nn * CI * p.T * 16);
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.id_out + (size_t)nn * CI * p.T * 16);
                const unsigned so = (unsigned)tt * 64u;
#pragma unroll
                for (int i = 0; i < CT * 4; ++i) {
                    idg[i] = bload(rg, lane_x, so + (unsigned)((i >> 2) * 16 + (i & 3)) * TS);
                    ido[i] = bload(ro, lane_x, so + (unsigned)((i >> 2) * 16 + (i & 3)) * TS);
                }
            }
        }
        if (have_next) {
            int n2 = nn, t2 = tt + 1;
            if (t2 == p.T) { t2 = 0; ++n2; }
            load_dy(n2, t2, dyn);
        }

        // dz_k = W_k^T dy; the W fragments of step s + 1 are read while step s multiplies
        f32x4 dz[3][CT];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < CT; ++j) dz[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float wf[3][CT], wn[3][CT];
        const float* wl = Wl + (size_t)lane * CT;
#pragma unroll
        for (int k = 0; k < 3; ++k) lds_frag<CT>(wl + (size_t)(k * S) * 64 * CT, wf[k]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s + 1 < S) {
#pragma unroll
                for (int k = 0; k < 3; ++k) lds_frag<CT>(wl + (size_t)(k * S + s + 1) * 64 * CT, wn[k]);
            }
            __builtin_amdgcn_sched_barrier(0);        // the reads of step s + 1 stay AHEAD of step s's MFMAs
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    dz[k][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[k][j], dyc[s], dz[k][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < CT; ++j) wf[k][j] = wn[k][j];
        }
        // residual branch: SUM_c Wr[c,ci] dres[c,t',w] has dz's layout and adds straight into dx
        f32x4 dxa[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) dxa[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (RES != 0) {
            if (res_row) {
                const float* wr = Wrl + (size_t)lane * CT;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    float wv[CT];
                    lds_frag<CT>(wr + (size_t)s * 64 * CT, wv);
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        dxa[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], drc[s], dxa[j], 0, 0, 0);
                    if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // the lane's rotation coefficients (16 per partition, 12 wide LDS reads) arrive under the dA chain
        f32x4 cf[3][4];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) cf[k][g] = *reinterpret_cast<const f32x4*>(cfs + (k * 16 + l16) * 16 + 4 * g);
        // dA_k += x^T dz_k
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    dAk[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[j * 4 + r], dz[k][j][r], dAk[k], 0, 0, 0);
        // dx = SUM_k A_k dz_k along the lanes of the frame
        if (p.dx) {
            float dxr[CT * 4];
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) dxr[j * 4 + r] = dxa[j][r];
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 3>([&](auto kt) {
                constexpr int kq = decltype(kt)::value;
                static_for<0, 16>([&](auto rt) {
                    constexpr int RR = decltype(rt)::value;
                    if ((rmask[kq] >> RR) & 1u) {
                        const float c = cf[kq][RR >> 2][RR & 3];
#pragma unroll
                        for (int j = 0; j < CT; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                fma_ror<RR>(dxr[j * 4 + r], c, dz[kq][j][r], j == 0 && r == 0);
                    }
                });
            });
            if constexpr (RES == 0) {
                if (idres) {
#pragma unroll
                    for (int i = 0; i < CT * 4; ++i) dxr[i] += ido[i] > 0.f ? idg[i] : 0.f;
                }
            }
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.dx + (size_t)nn * CI * p.T * 16);
            const unsigned so = (unsigned)tt * 64u;
#pragma unroll
            for (int i = 0; i < CT * 4; ++i) bstore(rs, lane_x, so + (unsigned)((i >> 2) * 16 + (i & 3)) * TS, dxr[i]);
        }
    };

    if (nrows > 0) load_dy(n, t, dyA);
    for (int i = 0; i < nrows; i += 2) {
        do_row(n, t, i + 1 < nrows, dyA, dyB);
        if (++t == p.T) { t = 0; ++n; }
        if (i + 1 < nrows) {
            do_row(n, t, i + 2 < nrows, dyB, dyA);
            if (++t == p.T) { t = 0; ++n; }
        }
    }

    // block partial of dA (dense 16 x 16 per k): D register i of lane l = dA[k][4*(l>>4)+i][l&15]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave * 768 + (k * 16 + 4 * q + i) * 16 + l16] = dAk[k][i];
    __syncthreads();
    for (int e = tid; e < 768; e += 512) {
        float sum = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) sum += red[w8 * 768 + e];
        p.part1[(size_t)blockIdx.x * 768 + e] = sum;
    }
}

// Split-operand twin of gcn_bwd_data16_kernel (step_gcn_bwd_split): the two MFMA products on v_mfma_f32_16x16x32_bf16
// with every fp32 operand as bf16 hi + lo (split_mfma.h).  One instruction pair covers 16 values of the contracted
// index, lane group q holding 4 of them, so the fragments are regrouped:
//   dz_k[ci,w]  = SUM_c W_k[c,ci] dy[c,t,w]     A = W (LDS, split once per block: lane (q, ci) holds c = 16 s + 4q + e),
//                                               B = the lane's four 4-byte loads dy[16 s + 4q + e][t][w], split per row
//   dA_k[v,w]  += SUM_ci x[ci,t,v] dz_k[ci,w]   A = x[16 j + 4q + e][t][v] (own loads),  B = dz's D registers, rounded
//                                               to hi + lo here
// dx (DPP rotations of the fp32 dz), the residual conv term, the identity-residual fusion and the dA reduction are
// the fp32 kernel's.
template <int CI, int CO, int RES, bool IDRES>
__global__ __launch_bounds__(512, 2) void gcn_bwd_data16s_kernel(GbParams p) {
    constexpr int CT = CI / 16;
    constexpr int S4 = CO / 16;                       // contraction steps over c (16 output channels each)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SplitA* Wl = reinterpret_cast<SplitA*>(smem);     // [3][S4][CT][64 lanes]
    SplitA* Wrl = Wl + 3 * S4 * CT * 64;              // [S4][CT][64]   (RES)
    float* cfs = reinterpret_cast<float*>(Wrl + (RES ? S4 * CT * 64 : 0));    // [3][16 joints][16 rotations]
    float* red = smem;                                // [8][768], after the loop

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;

    for (int e = tid; e < (3 + (RES ? 1 : 0)) * S4 * CT * 64; e += 512) {
        const int ln = e & 63, j = (e >> 6) % CT, s = (e / (64 * CT)) % S4, k = e / (64 * CT * S4);
        const int ci = 16 * j + (ln & 15);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * s + 4 * (ln >> 4) + r;
            float x = 0.f;
            if (k < 3) {
                if (k < p.K) x = p.w_transposed ? p.w[(size_t)ci * (p.K * CO) + k * CO + c] : p.w[((size_t)k * CO + c) * CI + ci];
            } else {
                x = p.res_transposed ? p.res_w[(size_t)ci * CO + c] : p.res_w[(size_t)c * CI + ci];
            }
            v[r] = x;
        }
        Wl[e] = split_a(v);                           // (k == 3 lands in Wrl)
    }
    unsigned rmask[3] = {0u, 0u, 0u};
    static_for<0, 16>([&](auto rt) {
        constexpr int R = decltype(rt)::value;
        const int src = __float_as_int(row_ror<R>(__int_as_float(l16)));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = (k < p.K) ? p.A[(k * 16 + l16) * 16 + src] : 0.f;
            if (tid < 16) cfs[(k * 16 + l16) * 16 + R] = a;
            if (__ballot(a != 0.f) != 0ull) rmask[k] |= 1u << R;
        }
    });
#pragma unroll
    for (int k = 0; k < 3; ++k) rmask[k] = __builtin_amdgcn_readfirstlane(rmask[k]);
    __syncthreads();

    f32x4 dAk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dAk[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's contiguous run of frames.  (Measured: FETCH_SIZE is 1.9x the algorithmic bytes - every 128-byte
    // line holds two frames of a channel row and the second request comes a whole row-step later, when the L2 has
    // dropped it.  Giving the two frames of a line to a pair of waves brought that to 1.1x at 32 -> 64 and changed
    // the time by -2 % / +11 %, so it is not what bounds the kernel; the run per wave stays.)
    const long long R = (long long)p.N * p.T;
    const long long nw = (long long)gridDim.x * 8, gw = (long long)blockIdx.x * 8 + wave;
    const long long r0 = R * gw / nw, r1 = R * (gw + 1) / nw;
    const int nrows = (int)(r1 - r0);
    int n = (int)(r0 / p.T), t = (int)(r0 - (long long)n * p.T);
    const unsigned TS = (unsigned)p.T * 64u, TOS = (unsigned)p.To * 64u;      // bytes per channel row
    const unsigned lane_x = 4u * q * TS + l16 * 4u, lane_dr = 4u * q * TOS + l16 * 4u;

    // With the MFMA time of a row down to a fraction of an HBM round trip, every operand of row r + 1 is requested
    // while row r computes: dy / x / dres at its start, the identity residual's two rows after the dA chain (so that
    // fewer than 63 operations are newer than anything a chain waits for), all unconditionally (the last row
    // requests itself again) - see the note on s_waitcnt at gcn_bwd_data16_kernel.
    struct Row { f32x4 dy[S4], x[CT], dr[RES ? S4 : 1], idg[IDRES ? CT : 1], ido[IDRES ? CT : 1]; };
    auto load_main = [&](int nn, int tt, Row& rw) {
        const unsigned so = (unsigned)tt * 64u;
        {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.dy + (size_t)nn * CO * p.T * 16);
#pragma unroll
            for (int s = 0; s < S4; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) rw.dy[s][e] = bload(rs, lane_x, so + (unsigned)(16 * s + e) * TS);
        }
        if constexpr (RES != 0) {
            if (RES == 1 || (tt & 1) == 0) {          // (stride 2: only even frames have a dres row)
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.dres + (size_t)nn * CO * p.To * 16);
                const unsigned sr = (unsigned)(RES == 1 ? tt : tt >> 1) * 64u;
#pragma unroll
                for (int s = 0; s < S4; ++s)
#pragma unroll
                    for (int e = 0; e < 4; ++e) rw.dr[s][e] = bload(rs, lane_dr, sr + (unsigned)(16 * s + e) * TOS);
            }
        }
        {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.x + (size_t)nn * CI * p.T * 16);
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) rw.x[j][e] = bload(rs, lane_x, so + (unsigned)(16 * j + e) * TS);
        }
    };
    auto load_id = [&](int nn, int tt, Row& rw) {
        if constexpr (IDRES) {
            const unsigned so = (unsigned)tt * 64u;
            const __amdgpu_buffer_rsrc_t rg = make_rsrc(p.id_dout + (size_t)nn * CI * p.T * 16);
            const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.id_out + (size_t)nn * CI * p.T * 16);
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rw.idg[j][e] = bload(rg, lane_x, so + (unsigned)(16 * j + e) * TS);
                    rw.ido[j][e] = bload(ro, lane_x, so + (unsigned)(16 * j + e) * TS);
                }
        }
    };
    auto do_row = [&](int nn, int tt, bool have_next, Row& cur, Row& nxt) {
        const bool res_row = RES == 1 || (RES == 2 && (tt & 1) == 0);
        int t2 = tt + 1, n2 = nn;
        if (t2 == p.T) { t2 = 0; ++n2; }
        n2 = have_next ? n2 : nn;
        t2 = have_next ? t2 : tt;
        load_main(n2, t2, nxt);

        f32x4 dz[3][CT];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < CT; ++j) dz[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const SplitA* wl = Wl + lane;
#pragma unroll
        for (int s = 0; s < S4; ++s) {
            const SplitB ds = split_b(cur.dy[s]);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int j = 0; j < CT; ++j) dz[k][j] = mfma_split(wl[((k * S4 + s) * CT + j) * 64], ds, dz[k][j]);
        }
        f32x4 dxa[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) dxa[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (RES != 0) {
            if (res_row) {
                const SplitA* wr = Wrl + lane;
#pragma unroll
                for (int s = 0; s < S4; ++s) {
                    const SplitB ds = split_b(cur.dr[s]);
#pragma unroll
                    for (int j = 0; j < CT; ++j) dxa[j] = mfma_split(wr[(s * CT + j) * 64], ds, dxa[j]);
                }
            }
        }
        // dA_k += x^T dz_k
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const SplitA xs = split_a(cur.x[j]);
#pragma unroll
            for (int k = 0; k < 3; ++k) dAk[k] = mfma_split(xs, split_b(dz[k][j]), dAk[k]);
        }
        __builtin_amdgcn_sched_barrier(0);            // requested HERE, not hoisted to the row's other loads
        load_id(n2, t2, nxt);
        __builtin_amdgcn_sched_barrier(0);
        {                                             // (dx == NULL: computed all the same, stores dropped)
            float dxr[CT * 4];
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) dxr[j * 4 + r] = dxa[j][r];
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 3>([&](auto kt) {
                constexpr int kq = decltype(kt)::value;
                static_for<0, 16>([&](auto rt) {
                    constexpr int RR = decltype(rt)::value;
                    if ((rmask[kq] >> RR) & 1u) {
                        const float c = cfs[(kq * 16 + l16) * 16 + RR];       // (one LDS dword per live rotation)
#pragma unroll
                        for (int j = 0; j < CT; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                fma_ror<RR>(dxr[j * 4 + r], c, dz[kq][j][r], j == 0 && r == 0);
                    }
                });
            });
            if constexpr (IDRES) {
#pragma unroll
                for (int i = 0; i < CT * 4; ++i) dxr[i] += cur.ido[i >> 2][i & 3] > 0.f ? cur.idg[i >> 2][i & 3] : 0.f;
            }
            // (1 GB of dx costs 0.85 ms of this kernel at N = 4096, 64 -> 64: dword stores, 64 bytes per channel row.
            // Writing the two frames of each 128-byte line back to back did not change that.)
            const __amdgpu_buffer_rsrc_t rs = make_rsrc_opt(p.dx ? p.dx + (size_t)nn * CI * p.T * 16 : nullptr);
            const unsigned so = (unsigned)tt * 64u;
#pragma unroll
            for (int i = 0; i < CT * 4; ++i) bstore(rs, lane_x, so + (unsigned)((i >> 2) * 16 + (i & 3)) * TS, dxr[i]);
        }
    };

    Row ra, rb;
    if (nrows > 0) {
        load_main(n, t, ra);
        load_id(n, t, ra);
    }
    for (int i = 0; i < nrows; i += 2) {
        do_row(n, t, i + 1 < nrows, ra, rb);
        if (++t == p.T) { t = 0; ++n; }
        if (i + 1 < nrows) {
            do_row(n, t, i + 2 < nrows, rb, ra);
            if (++t == p.T) { t = 0; ++n; }
        }
    }

    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave * 768 + (k * 16 + 4 * q + i) * 16 + l16] = dAk[k][i];
    __syncthreads();
    for (int e = tid; e < 768; e += 512) {
        float sum = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) sum += red[w8 * 768 + e];
        p.part1[(size_t)blockIdx.x * 768 + e] = sum;
    }
}

// Data kernel of the wide layers: a block owns a group of CIG = 16 * CT input channels (blockIdx.y) with the matching
// slice of W (all Co output channels) in LDS and walks the output channels in chunks of 64, so dz for its group stays
// in 12 * CT registers whatever Co is.  Otherwise gcn_bwd_data16_kernel: one frame per wave at a time, no barriers.
template <int CT>
__global__ __launch_bounds__(512, 2) void gcn_bwd_data16w_kernel(GbParams p) {
    constexpr int S = 16;                             // k-steps of one 64-channel chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CO = p.Co, CI = p.Ci, NCH = CO / 64;
    const bool has_res = p.dres != nullptr;
    float* Wl = smem;                                 // [3][CO / 4][64 lanes][CT]
    float* Wrl = Wl + 3 * (CO / 4) * 64 * CT;         // [CO / 4][64][CT]   (residual)
    float* cfs = Wrl + (has_res ? (CO / 4) * 64 * CT : 0);   // [3][16 joints][16 rotations]
    float* red = smem;                                // [8][768], after the loop

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int ci0 = blockIdx.y * 16 * CT;

    for (int e = tid; e < 3 * (CO / 4) * 64 * CT; e += 512) {
        const int tile = e % CT, ln = (e / CT) & 63, s = (e / (CT * 64)) % (CO / 4), k = e / (CT * 64 * (CO / 4));
        const int c = 4 * s + (ln >> 4), ci = ci0 + 16 * tile + (ln & 15);
        float v = 0.f;
        if (k < p.K)
            v = p.w_transposed ? p.w[(size_t)ci * (p.K * CO) + k * CO + c] : p.w[((size_t)k * CO + c) * CI + ci];
        Wl[e] = v;
    }
    if (has_res) {
        for (int e = tid; e < (CO / 4) * 64 * CT; e += 512) {
            const int tile = e % CT, ln = (e / CT) & 63, s = e / (CT * 64);
            const int c = 4 * s + (ln >> 4), ci = ci0 + 16 * tile + (ln & 15);
            Wrl[e] = p.res_transposed ? p.res_w[(size_t)ci * CO + c] : p.res_w[(size_t)c * CI + ci];
        }
    }
    unsigned rmask[3] = {0u, 0u, 0u};
    static_for<0, 16>([&](auto rt) {
        constexpr int R = decltype(rt)::value;
        const int src = __float_as_int(row_ror<R>(__int_as_float(l16)));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float a = (k < p.K) ? p.A[(k * 16 + l16) * 16 + src] : 0.f;
            if (tid < 16) cfs[(k * 16 + l16) * 16 + R] = a;
            if (__ballot(a != 0.f) != 0ull) rmask[k] |= 1u << R;
        }
    });
#pragma unroll
    for (int k = 0; k < 3; ++k) rmask[k] = __builtin_amdgcn_readfirstlane(rmask[k]);
    __syncthreads();

    f32x4 dAk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) dAk[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long R = (long long)p.N * p.T;
    const long long nw = (long long)gridDim.x * 8, gw = (long long)blockIdx.x * 8 + wave;
    const long long r0 = R * gw / nw, r1 = R * (gw + 1) / nw;
    const int nrows = (int)(r1 - r0);
    const unsigned TS = (unsigned)p.T * 64u, TOS = (unsigned)p.To * 64u;
    const unsigned lane_dy = (unsigned)q * TS + l16 * 4u, lane_x = (unsigned)(ci0 + 4 * q) * TS + l16 * 4u;
    const unsigned lane_dr = (unsigned)q * TOS + l16 * 4u;
    int n = (int)(r0 / p.T), t = (int)(r0 - (long long)n * p.T);
    const float* wl = Wl + (size_t)lane * CT;
    const float* wr = Wrl + (size_t)lane * CT;

    for (int i = 0; i < nrows; ++i) {
        const bool res_row = has_res && (p.stride == 1 || (t & 1) == 0);
        const unsigned so = (unsigned)t * 64u;
        const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy + (size_t)n * CO * p.T * 16);
        const __amdgpu_buffer_rsrc_t rdr = make_rsrc(has_res ? p.dres + (size_t)n * CO * p.To * 16 : p.dy);
        const unsigned sr = (unsigned)(p.stride == 1 ? t : t >> 1) * 64u;
        float xc[CT * 4];
        {
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.x + (size_t)n * CI * p.T * 16);
#pragma unroll
            for (int j = 0; j < CT * 4; ++j) xc[j] = bload(rs, lane_x, so + (unsigned)((j >> 2) * 16 + (j & 3)) * TS);
        }
        f32x4 dz[3][CT], dxa[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            dxa[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k) dz[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float dyc[S];
#pragma unroll
        for (int s = 0; s < S; ++s) dyc[s] = bload(rdy, lane_dy, so + (unsigned)s * 4u * TS);
        for (int ch = 0; ch < NCH; ++ch) {
            float dyn[S], drc[S];
            if (ch + 1 < NCH) {
#pragma unroll
                for (int s = 0; s < S; ++s) dyn[s] = bload(rdy, lane_dy, so + (unsigned)(16 * (ch + 1) + s) * 4u * TS);
            }
            if (res_row) {
#pragma unroll
                for (int s = 0; s < S; ++s) drc[s] = bload(rdr, lane_dr, sr + (unsigned)(16 * ch + s) * 4u * TOS);
            }
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float wf[CT];
                    lds_frag<CT>(wl + (size_t)(k * (CO / 4) + 16 * ch + s) * 64 * CT, wf);
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        dz[k][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], dyc[s], dz[k][j], 0, 0, 0);
                }
            if (res_row) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    float wv[CT];
                    lds_frag<CT>(wr + (size_t)(16 * ch + s) * 64 * CT, wv);
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        dxa[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], drc[s], dxa[j], 0, 0, 0);
                }
            }
            if (ch + 1 < NCH) {
#pragma unroll
                for (int s = 0; s < S; ++s) dyc[s] = dyn[s];
            }
        }
        f32x4 cf[3][4];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g) cf[k][g] = *reinterpret_cast<const f32x4*>(cfs + (k * 16 + l16) * 16 + 4 * g);
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    dAk[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[j * 4 + r], dz[k][j][r], dAk[k], 0, 0, 0);
        if (p.dx) {
            float dxr[CT * 4];
#pragma unroll
            for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) dxr[j * 4 + r] = dxa[j][r];
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 3>([&](auto kt) {
                constexpr int kq = decltype(kt)::value;
                static_for<0, 16>([&](auto rt) {
                    constexpr int RR = decltype(rt)::value;
                    if ((rmask[kq] >> RR) & 1u) {
                        const float c = cf[kq][RR >> 2][RR & 3];
#pragma unroll
                        for (int j = 0; j < CT; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                fma_ror<RR>(dxr[j * 4 + r], c, dz[kq][j][r], j == 0 && r == 0);
                    }
                });
            });
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.dx + (size_t)n * CI * p.T * 16);
#pragma unroll
            for (int j = 0; j < CT * 4; ++j) bstore(rs, lane_x, so + (unsigned)((j >> 2) * 16 + (j & 3)) * TS, dxr[j]);
        }
        if (++t == p.T) { t = 0; ++n; }
    }

    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave * 768 + (k * 16 + 4 * q + i) * 16 + l16] = dAk[k][i];
    __syncthreads();
    for (int e = tid; e < 768; e += 512) {
        float sum = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) sum += red[w8 * 768 + e];
        p.part1[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 768 + e] = sum;
    }
}

// weight kernel: 512-thread blocks = 8 waves = (CI / 16 input-channel tiles) x (8 / CIT frame streams).  CI / CO are
// the channels ONE BLOCK covers: wide layers (Ci or Co > 64) run as blockIdx.y = (input group, output group) with the
// same code, each group writing its own sub-block of the partial row.
// SPLIT: both products on v_mfma_f32_16x16x32_bf16 with every fp32 operand (A_k, x, z_k, dy, dres) as bf16 hi + lo
// (split_mfma.h): two instructions where the fp32 form issues four 16x16x4, same fragment layouts.
template <int CI, int CO, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void gcn_bwd_weight16_kernel(GbParams p) {
    constexpr int CIT = CI / 16, COT = CO / 16;
    constexpr int SPB = 8 / CIT;                      // frame streams per block (waves sharing a ci tile)
    const int CIA = p.Ci, COA = p.Co;                 // all channels
    const int PART = 3 * COA * CIA + COA * 16 + COA * CIA + COA;
    const int ngi = CIA / CI;
    const int ci0 = ((int)blockIdx.y % ngi) * CI, co0 = ((int)blockIdx.y / ngi) * CO;
    constexpr int PW = (4 * COT) * 4 + 2 * COT * 4;   // accumulator floats per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [SPB - 1][CIT][64][PW]: the other streams' sums

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably uniform: rows, descriptors, offsets in SGPRs
    const int q = lane >> 4, l16 = lane & 15;
    const int cit = wave % CIT, sib = wave / CIT;

    // A operand of the aggregation: A[i = w][k' = q] of step e = A_k[v = 4q + e][w]
    float akt[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) akt[k][e] = (k < p.K) ? p.A[(k * 16 + 4 * q + e) * 16 + l16] : 0.f;
    SplitA aks[SPLIT ? 3 : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int k = 0; k < 3; ++k) aks[k] = split_a(f32x4{akt[k][0], akt[k][1], akt[k][2], akt[k][3]});
    }

    f32x4 acc[3][COT], accr[COT];
    f32x4 Sacc[COT], Bacc[COT];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        accr[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        Sacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        Bacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const long long R = (long long)p.N * p.T;
    const long long ns = (long long)gridDim.x * SPB, gs = (long long)blockIdx.x * SPB + sib;
    const long long r0 = R * gs / ns, r1 = R * (gs + 1) / ns;
    const int nrows = (int)(r1 - r0);
    const bool have_res = p.dres != nullptr;
    const unsigned TS = (unsigned)p.T * 64u, TOS = (unsigned)p.To * 64u;
    const unsigned lane_x = (unsigned)(ci0 + 16 * cit + l16) * TS + q * 16u;
    const unsigned lane_dy = (unsigned)(co0 + l16) * TS + q * 16u, lane_dr = (unsigned)(co0 + l16) * TOS + q * 16u;
    int n = (int)(r0 / p.T), t = (int)(r0 - (long long)n * p.T);

    struct Row { f32x4 x, dy[COT], dr[COT]; };
    auto load_row = [&](int nn, int tt, Row& rw) {
        const unsigned so = (unsigned)tt * 64u;
        rw.x = bload4(make_rsrc(p.x + (size_t)nn * CIA * p.T * 16), lane_x, so);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.dy + (size_t)nn * COA * p.T * 16);
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) rw.dy[ct] = bload4(rs, lane_dy, so + (unsigned)ct * 16u * TS);
        if (have_res && (p.stride == 1 || (tt & 1) == 0)) {
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.dres + (size_t)nn * COA * p.To * 16);
            const unsigned sr = (unsigned)(p.stride == 1 ? tt : tt >> 1) * 64u;
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) rw.dr[ct] = bload4(rr, lane_dr, sr + (unsigned)ct * 16u * TOS);
        }
    };
    auto do_row = [&](int nn, int tt, bool have_next, Row& cur, Row& nxt) {
        if (have_next) {
            int n2 = nn, t2 = tt + 1;
            if (t2 == p.T) { t2 = 0; ++n2; }
            load_row(n2, t2, nxt);
        }
        if constexpr (SPLIT) {
            const SplitB xs = split_b(cur.x);
            SplitB zs[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) zs[k] = split_b(mfma_split(aks[k], xs, f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
                if (cit == 0 && ci0 == 0) Sacc[ct] += cur.dy[ct];
                const SplitA ds = split_a(cur.dy[ct]);
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k][ct] = mfma_split(ds, zs[k], acc[k][ct]);
            }
            if (have_res && (p.stride == 1 || (tt & 1) == 0)) {
#pragma unroll
                for (int ct = 0; ct < COT; ++ct) {
                    if (cit == 0 && ci0 == 0) Bacc[ct] += cur.dr[ct];
                    accr[ct] = mfma_split(split_a(cur.dr[ct]), xs, accr[ct]);
                }
            }
            return;
        }
        // z_k = x A_k
        f32x4 z[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            z[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) z[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(akt[k][e], cur.x[e], z[k], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
            if (cit == 0 && ci0 == 0) Sacc[ct] += cur.dy[ct];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    acc[k][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.dy[ct][r], z[k][r], acc[k][ct], 0, 0, 0);
        }
        if (have_res && (p.stride == 1 || (tt & 1) == 0)) {
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
                if (cit == 0 && ci0 == 0) Bacc[ct] += cur.dr[ct];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    accr[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.dr[ct][r], cur.x[r], accr[ct], 0, 0, 0);
            }
        }
    };

    Row ra, rb;
    if (nrows > 0) load_row(n, t, ra);
    for (int i = 0; i < nrows; i += 2) {
        do_row(n, t, i + 1 < nrows, ra, rb);
        if (++t == p.T) { t = 0; ++n; }
        if (i + 1 < nrows) {
            do_row(n, t, i + 2 < nrows, rb, ra);
            if (++t == p.T) { t = 0; ++n; }
        }
    }

    // the streams of a block that share a ci tile are summed through LDS in stream order
    if (sib > 0) {
        float* st = smem + (((size_t)(sib - 1) * CIT + cit) * 64 + lane) * PW;
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
            for (int k = 0; k < 3; ++k) *reinterpret_cast<f32x4*>(st + (k * COT + ct) * 4) = acc[k][ct];
            *reinterpret_cast<f32x4*>(st + (3 * COT + ct) * 4) = accr[ct];
            *reinterpret_cast<f32x4*>(st + 16 * COT + ct * 8) = Sacc[ct];
            *reinterpret_cast<f32x4*>(st + 16 * COT + ct * 8 + 4) = Bacc[ct];
        }
    }
    __syncthreads();
    if (sib == 0) {
#pragma unroll 1
        for (int o = 1; o < SPB; ++o) {
            const float* st = smem + (((size_t)(o - 1) * CIT + cit) * 64 + lane) * PW;
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k][ct] += *reinterpret_cast<const f32x4*>(st + (k * COT + ct) * 4);
                accr[ct] += *reinterpret_cast<const f32x4*>(st + (3 * COT + ct) * 4);
                Sacc[ct] += *reinterpret_cast<const f32x4*>(st + 16 * COT + ct * 8);
                Bacc[ct] += *reinterpret_cast<const f32x4*>(st + 16 * COT + ct * 8 + 4);
            }
        }
        float* part = p.part2 + (size_t)blockIdx.x * PART;
        const int ci = ci0 + 16 * cit + l16;
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = co0 + 16 * ct + 4 * q + r;
#pragma unroll
                for (int k = 0; k < 3; ++k) part[(k * COA + c) * CIA + ci] = acc[k][ct][r];
                part[3 * COA * CIA + COA * 16 + c * CIA + ci] = accr[ct][r];
            }
            if (cit == 0 && ci0 == 0) {
                const int c = co0 + 16 * ct + l16;                // the lane's dy row
                float* ps = part + 3 * COA * CIA + c * 16 + 4 * q;
                ps[0] = Sacc[ct][0]; ps[1] = Sacc[ct][1]; ps[2] = Sacc[ct][2]; ps[3] = Sacc[ct][3];
                float b = (Bacc[ct][0] + Bacc[ct][1]) + (Bacc[ct][2] + Bacc[ct][3]);
                b += __shfl_xor(b, 16, 64);
                b += __shfl_xor(b, 32, 64);
                if (q == 0) part[3 * COA * CIA + COA * 16 + COA * CIA + c] = b;
            }
        }
    }
}

// Backward of the FIRST block's graph conv (Ci <= 4: the 3 coordinate channels; data_bn folded into the load,
// residual 1x1 conv, Co = 32): HBM-bound (dy and dres are 2 GB each at N = 4096), so ONE kernel reads them once
// for both the data and the weight gradients - the two-kernel forms read everything twice.  Frame-per-wave, operands in
// registers; the three partitions share one 16-row tile, row 4k + ci:
//   dz[4k+ci][w]   = SUM_c W_k[c,ci] dy[c,t,w]           A = W (registers),  B = the lane's 4-byte loads of dy
//   dA_k[v,w]     += SUM_ci x'[ci,v] dz[4k+ci][w]        B = dz's D register where the lane's row group q == k, else 0
//   dx[ci,v]       = SUM_k SUM_w A_k[v,w] dz[4k+ci][w]   DPP rotations with the lane's OWN partition's coefficients
//                    (row group q = k), then a sum over the row groups; + Wr^T dres (same layout, rows ci)
//   z_k[ci,w]      = SUM_v x'[ci,v] A_k[v,w]             lane j of a row holds channel j & 3, so after selecting
//                    k = j >> 2 the three results form ONE B operand with column j = 4k + ci
//   dW[c][4k+ci]  += SUM_w dy[c,t,w] z[4k+ci][w],   dWr[c][ci] += SUM_w dres[c,t',w] x'[ci,s t',w]
// x' = data_bn(x) (scale / shift per (v, ci): per-lane constants in both load layouts).  Every wave writes its own
// partial row through LDS at the end (ordered sums as everywhere); no barrier inside the frame loop.
template <int COT>
__global__ __launch_bounds__(512, 2) void gcn_bwd_small_kernel(GbParams p) {
    constexpr int CO = 16 * COT, S = CO / 4;
    constexpr int PART = 3 * CO * 4 + CO * 16 + CO * 4 + CO;       // partial row, Ci padded to 4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int Ci = p.Ci;
    const bool has_res = p.dres != nullptr;

    // A operands held in registers: row i = l16 = 4k + ci of step s: W_k[c = 4s + q][ci]; residual: row i = ci
    float wreg[S], wrreg[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int c = 4 * s + q, k = l16 >> 2, ci = l16 & 3;
        float v = 0.f, vr = 0.f;
        if (k < p.K && ci < Ci)
            v = p.w_transposed ? p.w[(size_t)ci * (p.K * CO) + k * CO + c] : p.w[((size_t)k * CO + c) * Ci + ci];
        if (has_res && l16 < Ci) vr = p.res_transposed ? p.res_w[(size_t)l16 * CO + c] : p.res_w[(size_t)c * Ci + l16];
        wreg[s] = v;
        wrreg[s] = vr;
    }
    // aggregation: A[i = w][k' = q] of step e = A_k[v = 4q + e][w]
    float akt[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) akt[k][e] = (k < p.K) ? p.A[(k * 16 + 4 * q + e) * 16 + l16] : 0.f;
    // rotation coefficients of the lane's own partition (row group q): A[q][v = l16][w = source lane of row_ror:R]
    float cfl[16];
    unsigned rmask = 0u;
    static_for<0, 16>([&](auto rt) {
        constexpr int R = decltype(rt)::value;
        const int src = __float_as_int(row_ror<R>(__int_as_float(l16)));
        const float a = (q < p.K) ? p.A[(q * 16 + l16) * 16 + src] : 0.f;
        cfl[R] = a;
        if (__ballot(a != 0.f) != 0ull) rmask |= 1u << R;
    });
    rmask = __builtin_amdgcn_readfirstlane(rmask);
    // data_bn on the two load layouts of x: float4 of channel l16 & 3 at joints 4q..4q+3; dword of channel r at joint l16
    const int cj = l16 & 3;
    float sq[4] = {1.f, 1.f, 1.f, 1.f}, hq[4] = {0.f, 0.f, 0.f, 0.f}, sd[4] = {1.f, 1.f, 1.f, 1.f}, hd[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.in_scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (cj < Ci) { sq[e] = p.in_scale[(4 * q + e) * Ci + cj]; hq[e] = p.in_shift[(4 * q + e) * Ci + cj]; }
            if (e < Ci) { sd[e] = p.in_scale[l16 * Ci + e]; hd[e] = p.in_shift[l16 * Ci + e]; }
        }
    }

    f32x4 dAk[3], acc[COT], accr[COT], Sacc[COT], Bacc[COT];
#pragma unroll
    for (int k = 0; k < 3; ++k) dAk[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
        acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; accr[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        Sacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; Bacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const long long R = (long long)p.N * p.T;
    const long long nw = (long long)gridDim.x * 8, gw = (long long)blockIdx.x * 8 + wave;
    const long long r0 = R * gw / nw, r1 = R * (gw + 1) / nw;
    const int nrows = (int)(r1 - r0);
    const unsigned TS = (unsigned)p.T * 64u, TOS = (unsigned)p.To * 64u;
    const unsigned lane_dyd = (unsigned)q * TS + l16 * 4u, lane_dyq = (unsigned)l16 * TS + q * 16u;
    const unsigned lane_drd = (unsigned)q * TOS + l16 * 4u, lane_drq = (unsigned)l16 * TOS + q * 16u;
    const unsigned lane_xq = (unsigned)(cj < Ci ? cj : 0) * TS + q * 16u, lane_xd = l16 * 4u;
    int n = (int)(r0 / p.T), t = (int)(r0 - (long long)n * p.T);

    struct Row { float dyd[S], drd[S], xd[4]; f32x4 dyq[COT], drq[COT], xq; };
    auto load_row = [&](int nn, int tt, Row& rw) {
        const unsigned so = (unsigned)tt * 64u;
        const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy + (size_t)nn * CO * p.T * 16);
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)nn * Ci * p.T * 16);
#pragma unroll
        for (int s = 0; s < S; ++s) rw.dyd[s] = bload(rdy, lane_dyd, so + (unsigned)s * 4u * TS);
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) rw.dyq[ct] = bload4(rdy, lane_dyq, so + (unsigned)ct * 16u * TS);
        rw.xq = bload4(rx, lane_xq, so);
#pragma unroll
        for (int r = 0; r < 4; ++r) rw.xd[r] = r < Ci ? bload(rx, lane_xd, so + (unsigned)r * TS) : 0.f;
        if (has_res && (p.stride == 1 || (tt & 1) == 0)) {
            const __amdgpu_buffer_rsrc_t rdr = make_rsrc(p.dres + (size_t)nn * CO * p.To * 16);
            const unsigned sr = (unsigned)(p.stride == 1 ? tt : tt >> 1) * 64u;
#pragma unroll
            for (int s = 0; s < S; ++s) rw.drd[s] = bload(rdr, lane_drd, sr + (unsigned)s * 4u * TOS);
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) rw.drq[ct] = bload4(rdr, lane_drq, sr + (unsigned)ct * 16u * TOS);
        }
    };
    auto do_row = [&](int nn, int tt, bool have_next, Row& cur, Row& nxt) {
        if (have_next) {
            int n2 = nn, t2 = tt + 1;
            if (t2 == p.T) { t2 = 0; ++n2; }
            load_row(n2, t2, nxt);
        }
        const bool res_row = has_res && (p.stride == 1 || (tt & 1) == 0);
        // x' = data_bn(x) in both layouts (channels past Ci stay 0)
        float xq[4], xd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xq[e] = cj < Ci ? cur.xq[e] * sq[e] + hq[e] : 0.f;
            xd[e] = e < Ci ? cur.xd[e] * sd[e] + hd[e] : 0.f;
        }
        // ---- data side
        f32x4 dz = f32x4{0.f, 0.f, 0.f, 0.f}, dxa = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s) dz = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], cur.dyd[s], dz, 0, 0, 0);
        if (res_row) {
#pragma unroll
            for (int s = 0; s < S; ++s) dxa = __builtin_amdgcn_mfma_f32_16x16x4f32(wrreg[s], cur.drd[s], dxa, 0, 0, 0);
        }
        // ---- weight side (independent of dz: fills the pipe while the dz chain drains)
        f32x4 z[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            z[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) z[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(akt[k][e], xq[e], z[k], 0, 0, 0);
        }
        const int kj = l16 >> 2;
        f32x4 zs;
#pragma unroll
        for (int r = 0; r < 4; ++r) zs[r] = kj == 0 ? z[0][r] : (kj == 1 ? z[1][r] : (kj == 2 ? z[2][r] : 0.f));
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
            Sacc[ct] += cur.dyq[ct];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.dyq[ct][r], zs[r], acc[ct], 0, 0, 0);
        }
        if (res_row) {
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
                Bacc[ct] += cur.drq[ct];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    accr[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.drq[ct][r], xq[r], accr[ct], 0, 0, 0);
            }
        }
        // ---- dA_k += x'^T dz_k (the lane's row group q is partition q)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < Ci) dAk[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xd[r], q == k ? dz[r] : 0.f, dAk[k], 0, 0, 0);
        // ---- dx
        if (p.dx) {
            float dxr[4] = {0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 16>([&](auto rt) {
                constexpr int RR = decltype(rt)::value;
                if ((rmask >> RR) & 1u) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r < Ci) fma_ror<RR>(dxr[r], cfl[RR], dz[r], r == 0);
                }
            });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r >= Ci) continue;
                float v = dxr[r];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                dxr[r] = v + dxa[r];                      // (row group 0 holds the residual product's rows ci)
            }
            if (q == 0) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.dx + (size_t)nn * Ci * p.T * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < Ci) bstore(rs, lane_xd, (unsigned)tt * 64u + (unsigned)r * TS, dxr[r]);
            }
        }
    };

    Row ra, rb;
    if (nrows > 0) load_row(n, t, ra);
    for (int i = 0; i < nrows; i += 2) {
        do_row(n, t, i + 1 < nrows, ra, rb);
        if (++t == p.T) { t = 0; ++n; }
        if (i + 1 < nrows) {
            do_row(n, t, i + 2 < nrows, rb, ra);
            if (++t == p.T) { t = 0; ++n; }
        }
    }

    // partials: dA (dense 16 x 16 per k) and dW / S / dWr / dbr in the finalize's layout (Ci padded to 4); the 8 waves'
    // rows are summed in order through LDS, one row per block leaves
    __shared__ float red[8][768 + PART];
    float* p1 = red[wave];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) p1[(k * 16 + 4 * q + i) * 16 + l16] = dAk[k][i];
    float* part = p1 + 768;
    const int kj = l16 >> 2, cij = l16 & 3;
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * ct + 4 * q + r;
            if (kj < 3) part[(kj * CO + c) * 4 + cij] = acc[ct][r];
            if (l16 < 4) part[3 * CO * 4 + CO * 16 + c * 4 + l16] = accr[ct][r];
        }
        const int c = 16 * ct + l16;
        float* ps = part + 3 * CO * 4 + c * 16 + 4 * q;
        ps[0] = Sacc[ct][0]; ps[1] = Sacc[ct][1]; ps[2] = Sacc[ct][2]; ps[3] = Sacc[ct][3];
        float b = (Bacc[ct][0] + Bacc[ct][1]) + (Bacc[ct][2] + Bacc[ct][3]);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (q == 0) part[3 * CO * 4 + CO * 16 + CO * 4 + c] = b;
    }
    __syncthreads();
    for (int e = tid; e < 768 + PART; e += 512) {
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) sum += red[wv][e];
        if (e < 768) p.part1[(size_t)blockIdx.x * 768 + e] = sum;
        else p.part2[(size_t)blockIdx.x * PART + (e - 768)] = sum;
    }
}

// ---------------------------------------------------------------------- finalize
struct FinParams {
    const float* part1; const float* part2; const float* A; const float* b;
    float* dw; float* db; float* dA; float* dres_w; float* dres_b;
    int nblk1, nblk2, emax, copad, cir, Ci, Co, K, w_transposed, res_transposed, have_res;
};

__global__ void gcn_bwd_finalize_kernel(FinParams f) {
    const int PART = 3 * f.copad * f.cir + f.copad * 16 + f.copad * f.cir + f.copad;
    const int offS = 3 * f.copad * f.cir, offR = offS + f.copad * 16, offB = offR + f.copad * f.cir;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nW = f.K * f.Co * f.Ci;
    if (gtid < nW) {                                       // dW
        if (!f.dw) return;
        const int ci = gtid % f.Ci, c = (gtid / f.Ci) % f.Co, k = gtid / (f.Ci * f.Co);
        float s = 0.f;
        for (int b = 0; b < f.nblk2; ++b) s += f.part2[(size_t)b * PART + (k * f.copad + c) * f.cir + ci];
        if (f.w_transposed) f.dw[(size_t)ci * (f.K * f.Co) + k * f.Co + c] = s;
        else f.dw[((size_t)k * f.Co + c) * f.Ci + ci] = s;
        return;
    }
    int e = gtid - nW;
    if (e < f.Co * f.Ci) {                                 // dWr
        if (!f.have_res || !f.dres_w) return;
        const int ci = e % f.Ci, c = e / f.Ci;
        float s = 0.f;
        for (int b = 0; b < f.nblk2; ++b) s += f.part2[(size_t)b * PART + offR + c * f.cir + ci];
        if (f.res_transposed) f.dres_w[(size_t)ci * f.Co + c] = s;
        else f.dres_w[(size_t)c * f.Ci + ci] = s;
        return;
    }
    e -= f.Co * f.Ci;
    if (e < f.Co) {                                        // dbr
        if (!f.have_res || !f.dres_b) return;
        float s = 0.f;
        for (int b = 0; b < f.nblk2; ++b) s += f.part2[(size_t)b * PART + offB + e];
        f.dres_b[e] = s;
        return;
    }
    e -= f.Co;
    if (e < f.K * f.Co) {                                  // db[k*Co+c] = SUM_w colsum_k[w] S[c,w]
        if (!f.db) return;
        const int c = e % f.Co, k = e / f.Co;
        float tot = 0.f;
        for (int w = 0; w < 16; ++w) {
            float S = 0.f;
            for (int b = 0; b < f.nblk2; ++b) S += f.part2[(size_t)b * PART + offS + c * 16 + w];
            float cs = 0.f;
            for (int v = 0; v < 16; ++v) cs += f.A[(k * 16 + v) * 16 + w];
            tot += cs * S;
        }
        f.db[e] = tot;
        return;
    }
    e -= f.K * f.Co;
    if (e < f.K * 16) {                                    // dA column (k, w), support only
        if (!f.dA) return;
        const int w = e & 15, k = e >> 4;
        float bias_part = 0.f;
        if (f.b)
            for (int c = 0; c < f.Co; ++c) {
                float S = 0.f;
                for (int b = 0; b < f.nblk2; ++b) S += f.part2[(size_t)b * PART + offS + c * 16 + w];
                bias_part += f.b[k * f.Co + c] * S;
            }
        for (int v = 0; v < 16; ++v) {
            const float a = f.A[(k * 16 + v) * 16 + w];
            float out = 0.f;
            if (a != 0.f) {
                float s = 0.f;
                for (int b = 0; b < f.nblk1; ++b) s += f.part1[(size_t)b * 768 + (k * 16 + v) * 16 + w];
                out = s + bias_part;
            }
            f.dA[(k * 16 + v) * 16 + w] = out;
        }
    }
}

// dst[c] = SUM_r src[r][c] for the two partial arrays of a launch, in a fixed order (reproducible): a block is
// 64 columns x 4 row quarters, 4 independent chains per thread, the quarters combined through LDS
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ src1, float* __restrict__ dst1,
                                                       int nrows1, int ncols1, const float* __restrict__ src2,
                                                       float* __restrict__ dst2, int nrows2, int ncols2) {
    __shared__ float red[4][64];
    const int nb1 = (ncols1 + 63) / 64;
    const bool first = (int)blockIdx.x < nb1;
    const float* src = first ? src1 : src2;
    float* dst = first ? dst1 : dst2;
    const int nrows = first ? nrows1 : nrows2, ncols = first ? ncols1 : ncols2;
    const int c = ((int)blockIdx.x - (first ? 0 : nb1)) * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const int ra = (int)((long long)nrows * g / 4), rb = (int)((long long)nrows * (g + 1) / 4);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < ncols) {
        int r = ra;
        for (; r + 3 < rb; r += 4) {
            s0 += src[(size_t)r * ncols + c];
            s1 += src[(size_t)(r + 1) * ncols + c];
            s2 += src[(size_t)(r + 2) * ncols + c];
            s3 += src[(size_t)(r + 3) * ncols + c];
        }
        for (; r < rb; ++r) s0 += src[(size_t)r * ncols + c];
    }
    red[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && c < ncols) dst[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

struct GbGeom {
    int cip, copad, emax, cir, nblk1, nblk2;
    size_t lds1, lds2, part1, part2;
};

bool gb_geom(int N, int Ci, int Co, int T, int nnz, GbGeom* g) {
    if (Ci < 1 || Ci > 64 || Co < 1 || Co > 64 || N < 1 || T < 1) return false;
    g->cip = Ci <= 8 ? 8 : (Ci <= 32 ? 32 : 64);
    g->copad = Co <= 32 ? 32 : 64;
    g->emax = nnz <= 4 ? 4 : 16;
    g->cir = g->cip <= 32 ? 32 : 64;
    const int kqp = ((3 * g->cip + 31) / 32) * 32;
    g->lds1 = ((size_t)g->copad * kqp + (size_t)g->copad * P + (size_t)g->cip * P + 768) * 4;
    g->lds2 = ((size_t)g->copad * PS + 3 * (size_t)g->cir * PS + 768 + (g->cip == 32 ? 32 * PS : 0)) * 4;
    if (g->lds2 < 4 * 64 * 64 * 4) g->lds2 = 4 * 64 * 64 * 4;       // the end-of-kernel wave reduction
    const long long ntiles = (long long)N * ((T + TT - 1) / TT);
    auto blocks = [&](size_t lds) {
        int per = (int)((160 * 1024) / lds);
        per = per < 1 ? 1 : (per > 4 ? 4 : per);
        long long b = 256LL * per;
        return (int)(b < ntiles ? b : ntiles);
    };
    g->nblk1 = blocks(g->lds1);
    g->nblk2 = blocks(g->lds2);
    g->part1 = (size_t)3 * 16 * 16;
    g->part2 = (size_t)3 * g->copad * g->cir + (size_t)g->copad * 16 + (size_t)g->copad * g->cir + g->copad;
    return g->lds1 <= 160 * 1024 && g->lds2 <= 160 * 1024;
}

// second form: blocks for the data / weight kernels (two 256-thread blocks per CU, fewer for tiny inputs)
bool gb16_shape(int Ci, int Co, const float* in_scale) {
    return !in_scale && (Ci == 32 || Ci == 64) && (Co == 32 || Co == 64) && !getenv("STEP_GCN_BWD_V1");
}

// first-block form (Ci <= 4): one fused kernel, a partial row per wave
bool gbs_shape(int Ci, int Co) { return Ci >= 1 && Ci <= 4 && Co == 32 && !getenv("STEP_GCN_BWD_V1"); }
int gbs_blocks(int N, int T) {
    const long long rows = (long long)N * T;
    const long long b = (rows + 15) / 16;
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

// wide layers (Ci or Co > 64): channel groups per launch
struct Gb16w { int ct, gi1, cib, cob, gy2, nblk; size_t lds1; };
bool gb16w_geom(int N, int Ci, int Co, int T, bool res, Gb16w* g) {
    if (Ci % 32 || Co % 64 || Ci > 256 || Co > 256 || Ci < 32 || (Ci <= 64 && Co <= 64)) return false;
    // data kernel: input-channel group of 64 or 32 whose W slice (all Co) fits LDS
    g->ct = 0;
    for (int ct = 4; ct >= 2; ct >>= 1) {
        const size_t lds = ((size_t)(3 + (res ? 1 : 0)) * Co * 16 * ct + 768) * 4;
        if (Ci % (16 * ct) == 0 && lds <= 150 * 1024) { g->ct = ct; g->lds1 = lds < 8 * 768 * 4 ? 8 * 768 * 4 : lds; break; }
    }
    if (!g->ct) return false;
    g->gi1 = Ci / (16 * g->ct);
    g->cib = Ci % 128 == 0 ? 128 : 64;
    g->cob = 64;
    g->gy2 = (Ci / g->cib) * (Co / g->cob);
    const long long rows = (long long)N * T;
    long long b = (rows + 15) / 16;
    const int gmax = g->gi1 > g->gy2 ? g->gi1 : g->gy2;
    const long long cap = 512 / gmax > 0 ? 512 / gmax : 1;
    g->nblk = (int)(b < 1 ? 1 : (b > cap ? cap : b));
    return true;
}

int launch_gb16w(const GbParams& p, const Gb16w& g, hipStream_t st) {
    hipError_t e = hipSuccess;
    auto attr = [&](const void* k, size_t lds) {
        if (e == hipSuccess && lds > 64 * 1024)
            e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    };
    if (g.ct == 4) {
        attr(reinterpret_cast<const void*>(gcn_bwd_data16w_kernel<4>), g.lds1);
        if (e == hipSuccess) hipLaunchKernelGGL((gcn_bwd_data16w_kernel<4>), dim3(g.nblk, g.gi1), dim3(512), g.lds1, st, p);
    } else {
        attr(reinterpret_cast<const void*>(gcn_bwd_data16w_kernel<2>), g.lds1);
        if (e == hipSuccess) hipLaunchKernelGGL((gcn_bwd_data16w_kernel<2>), dim3(g.nblk, g.gi1), dim3(512), g.lds1, st, p);
    }
    if (e != hipSuccess) {
        step::set_error("step_gcn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return STEP_ERR_LAUNCH;
    }
    if (g.cib == 128) {
        hipLaunchKernelGGL((gcn_bwd_weight16_kernel<128, 64>), dim3(g.nblk, g.gy2), dim3(512), 16, st, p);
    } else {
        const size_t lds2 = (size_t)4 * 64 * 96 * 4;       // two frame streams per ci tile: see launch_gb16
        attr(reinterpret_cast<const void*>(gcn_bwd_weight16_kernel<64, 64>), lds2);
        if (e != hipSuccess) {
            step::set_error("step_gcn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return STEP_ERR_LAUNCH;
        }
        hipLaunchKernelGGL((gcn_bwd_weight16_kernel<64, 64>), dim3(g.nblk, g.gy2), dim3(512), lds2, st, p);
    }
    return step::check_launch("step_gcn_bwd(wide)");
}

int gb16_blocks(int N, int T) {
    const long long rows = (long long)N * T;
    const long long b = (rows + 15) / 16;             // 8 waves per block, at least two frames per wave
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

template <int CI, int CO, bool SPLIT>
int launch_gb16(const GbParams& p, int nblk, hipStream_t st) {
    const int res = p.dres ? p.stride : 0;
    size_t lds1 = ((size_t)(3 + (res ? 1 : 0)) * (CO / 4) * 64 * (CI / 16) + 768) * 4;
    if (lds1 < 8 * 768 * 4) lds1 = 8 * 768 * 4;
    constexpr int CIT = CI / 16, COT = CO / 16;
    const size_t lds2 = (size_t)(8 / CIT - 1) * CIT * 64 * (24 * COT) * 4;
    auto k2 = gcn_bwd_weight16_kernel<CI, CO, SPLIT>;
    hipError_t e = hipSuccess;
    auto attr = [&](const void* k, size_t lds) {
        if (e == hipSuccess && lds > 64 * 1024)
            e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    };
    attr(reinterpret_cast<const void*>(k2), lds2);
    auto data = [&](auto k1) {
        attr(reinterpret_cast<const void*>(k1), lds1);
        if (e == hipSuccess) hipLaunchKernelGGL(k1, dim3(nblk), dim3(512), lds1, st, p);
    };
    if constexpr (SPLIT) {
        if (res == 0 && p.id_dout && p.dx) data(gcn_bwd_data16s_kernel<CI, CO, 0, true>);
        else if (res == 0) data(gcn_bwd_data16s_kernel<CI, CO, 0, false>);
        else if (res == 1) data(gcn_bwd_data16s_kernel<CI, CO, 1, false>);
        else data(gcn_bwd_data16s_kernel<CI, CO, 2, false>);
    } else {
        if (res == 0) data(gcn_bwd_data16_kernel<CI, CO, 0>);
        else if (res == 1) data(gcn_bwd_data16_kernel<CI, CO, 1>);
        else data(gcn_bwd_data16_kernel<CI, CO, 2>);
    }
    if (e != hipSuccess) {
        step::set_error("step_gcn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return STEP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(k2, dim3(nblk), dim3(512), lds2, st, p);
    return step::check_launch("step_gcn_bwd");
}

template <bool SPLIT>
int dispatch_gb16(const GbParams& p, int nblk, hipStream_t st) {
    if (p.Ci == 32) return p.Co == 32 ? launch_gb16<32, 32, SPLIT>(p, nblk, st) : launch_gb16<32, 64, SPLIT>(p, nblk, st);
    return p.Co == 32 ? launch_gb16<64, 32, SPLIT>(p, nblk, st) : launch_gb16<64, 64, SPLIT>(p, nblk, st);
}

template <int CIP, int COPAD, bool RES>
int launch_gb(const GbParams& p, const GbGeom& g, hipStream_t st) {
    auto k1 = gcn_bwd_data_kernel<CIP, COPAD, RES>;
    auto k2 = gcn_bwd_weight_kernel<CIP, COPAD>;
    hipError_t e = hipSuccess;
    if (g.lds1 > 64 * 1024)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k1),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds1);
    if (e == hipSuccess && g.lds2 > 64 * 1024)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k2),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds2);
    if (e != hipSuccess) {
        step::set_error("step_gcn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return STEP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(k1, dim3(g.nblk1), dim3(256), g.lds1, st, p);
    hipLaunchKernelGGL(k2, dim3(g.nblk2), dim3(256), g.lds2, st, p);
    return step::check_launch("step_gcn_bwd");
}

int dispatch_gb(const GbParams& p, const GbGeom& g, hipStream_t st) {
    const int key = g.cip * 100 + g.copad;
    switch (key) {
        case 832: return p.dres ? launch_gb<8, 32, true>(p, g, st) : launch_gb<8, 32, false>(p, g, st);
        case 864: return p.dres ? launch_gb<8, 64, true>(p, g, st) : launch_gb<8, 64, false>(p, g, st);
        case 3232: return p.dres ? launch_gb<32, 32, true>(p, g, st) : launch_gb<32, 32, false>(p, g, st);
        case 3264: return p.dres ? launch_gb<32, 64, true>(p, g, st) : launch_gb<32, 64, false>(p, g, st);
        case 6432: return p.dres ? launch_gb<64, 32, true>(p, g, st) : launch_gb<64, 32, false>(p, g, st);
        case 6464: return p.dres ? launch_gb<64, 64, true>(p, g, st) : launch_gb<64, 64, false>(p, g, st);
    }
    step::set_error("step_gcn_bwd: no kernel for Ci=%d Co=%d", p.Ci, p.Co);
    return STEP_ERR_ARG;
}

}  // namespace

extern "C" size_t step_gcn_bwd_ws_bytes(int N, int Ci, int Co, int K, int T) {
    (void)K;
    size_t best = 0;
    {
        Gb16w gw;                                                                      // wide layers
        if (N >= 1 && T >= 1 && gb16w_geom(N, Ci, Co, T, true, &gw)) {
            const size_t part2 = 3 * (size_t)Co * Ci + (size_t)Co * 16 + (size_t)Co * Ci + Co;
            return (((size_t)gw.nblk * gw.gi1 + 1) * 768 + ((size_t)gw.nblk + 1) * part2) * 4;
        }
    }
    if (N >= 1 && T >= 1 && Ci >= 1 && Ci <= 4 && Co == 32) {                         // first-block form
        const size_t rows = (size_t)gbs_blocks(N, T);
        best = (rows + 1) * ((size_t)768 + 3 * (size_t)Co * 4 + (size_t)Co * 16 + (size_t)Co * 4 + Co) * 4;
    }
    if ((Ci == 32 || Ci == 64) && (Co == 32 || Co == 64) && N >= 1 && T >= 1) {      // second form
        const size_t nb = (size_t)gb16_blocks(N, T);
        best = (nb + 1) * ((size_t)768 + 3 * (size_t)Co * Ci + (size_t)Co * 16 + (size_t)Co * Ci + Co) * 4;
    }
    for (int nnz = 4; nnz <= 16; nnz += 12) {             // both kernel variants
        GbGeom g;
        if (!gb_geom(N, Ci, Co, T, nnz, &g)) continue;
        const size_t need = ((size_t)(g.nblk1 + 1) * g.part1 + (size_t)(g.nblk2 + 1) * g.part2) * 4;
        best = need > best ? need : best;
    }
    return best;
}

extern "C" int step_gcn_bwd(const float* dy, const float* dres, const float* x, const float* in_scale,
                            const float* in_shift, const float* w, const float* b, int w_transposed,
                            const float* A_eff, int a_nnz_max, const float* res_w, int res_transposed,
                            float* dx, float* dw, float* db, float* dA, float* dres_w, float* dres_b,
                            int N, int Ci, int Co, int K, int T, int stride, void* ws, size_t ws_bytes,
                            void* stream) {
    return step_gcn_bwd_idres(dy, dres, x, in_scale, in_shift, w, b, w_transposed, A_eff, a_nnz_max, res_w,
                              res_transposed, nullptr, nullptr, dx, dw, db, dA, dres_w, dres_b, N, Ci, Co, K, T,
                              stride, ws, ws_bytes, stream);
}

namespace {
int gcn_bwd(const float* dy, const float* dres, const float* x, const float* in_scale,
            const float* in_shift, const float* w, const float* b, int w_transposed,
            const float* A_eff, int a_nnz_max, const float* res_w, int res_transposed,
            const float* id_dout, const float* id_out,
            float* dx, float* dw, float* db, float* dA, float* dres_w, float* dres_b,
            int N, int Ci, int Co, int K, int T, int stride, void* ws, size_t ws_bytes, bool split,
            void* stream) {
    STEP_REQUIRE(dy && x && w && A_eff && ws, "step_gcn_bwd: null pointer");
    STEP_REQUIRE((id_dout == nullptr) == (id_out == nullptr), "step_gcn_bwd: id_dout and id_out come together");
    STEP_REQUIRE(!id_dout || (dx && !dres && Ci == Co && stride == 1),
                 "step_gcn_bwd: an identity residual needs dx, Ci == Co, stride 1 and no residual conv");
    STEP_REQUIRE(K >= 1 && K <= 3 && (stride == 1 || stride == 2), "step_gcn_bwd: K=%d stride=%d", K, stride);
    STEP_REQUIRE(a_nnz_max >= 1 && a_nnz_max <= 16, "step_gcn_bwd: a_nnz_max=%d", a_nnz_max);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "step_gcn_bwd: in_scale/in_shift pair");
    STEP_REQUIRE((dres == nullptr) == (res_w == nullptr), "step_gcn_bwd: dres and res_w come together");
    GbGeom g;
    Gb16w gw;
    const bool wide = N >= 1 && T >= 1 && !in_scale && gb16w_geom(N, Ci, Co, T, true, &gw);
    if (wide) {
        g = GbGeom{};
        g.emax = 16;
    } else {
        STEP_REQUIRE(gb_geom(N, Ci, Co, T, a_nnz_max, &g), "step_gcn_bwd: unsupported N=%d Ci=%d Co=%d T=%d", N,
                     Ci, Co, T);
    }
    const bool small = !wide && gbs_shape(Ci, Co);
    const bool v2 = wide || small || gb16_shape(Ci, Co, in_scale);
    if (wide) {
        g.nblk1 = gw.nblk * gw.gi1;
        g.nblk2 = gw.nblk;
    }
    if (small) g.nblk1 = g.nblk2 = gbs_blocks(N, T);
    if (v2) {
        if (!wide && !small) g.nblk1 = g.nblk2 = gb16_blocks(N, T);
        g.copad = Co; g.cir = small ? 4 : Ci;
        g.part1 = 768;
        g.part2 = 3 * (size_t)Co * g.cir + (size_t)Co * 16 + (size_t)Co * g.cir + Co;
    }
    const size_t need = ((size_t)(g.nblk1 + 1) * g.part1 + (size_t)(g.nblk2 + 1) * g.part2) * 4;
    if (ws_bytes < need) {
        step::set_error("step_gcn_bwd: workspace %zu < %zu bytes", ws_bytes, need);
        return STEP_ERR_WORKSPACE;
    }
    const int nblk1_ws = g.nblk1;                    // rows the workspace was sized for
    if (dres && !v2) {
        // the residual product keeps Wr and the dres tile in LDS too: fewer blocks per CU, never more
        const int cirp = g.cip <= 32 ? 32 : 64;
        g.lds1 += ((size_t)g.copad * cirp + (size_t)g.copad * (stride == 1 ? P : P / 2)) * 4;
        STEP_REQUIRE(g.lds1 <= 160 * 1024,
                     "step_gcn_bwd: residual conv with Ci=%d Co=%d stride=%d needs %zu bytes of LDS", Ci, Co, stride,
                     g.lds1);
        int per = (int)((160 * 1024) / g.lds1);
        per = per < 1 ? 1 : (per > 4 ? 4 : per);
        if (256 * per < g.nblk1) g.nblk1 = 256 * per;
    }
    GbParams p{};
    p.dy = dy; p.dres = dres; p.x = x; p.in_scale = in_scale; p.in_shift = in_shift; p.w = w; p.b = b;
    p.A = A_eff; p.res_w = res_w; p.dx = dx;
    // the second-form data kernel adds the identity residual's gradient in its dx store; every other form is
    // followed by the elementwise kernel
    const bool id_fused = id_dout && v2 && !wide && !small;
    if (id_fused) { p.id_dout = id_dout; p.id_out = id_out; }
    p.part1 = (float*)ws;
    p.part2 = p.part1 + (size_t)nblk1_ws * g.part1;
    p.w_transposed = w_transposed; p.res_transposed = res_transposed;
    p.N = N; p.Ci = Ci; p.Co = Co; p.K = K; p.T = T; p.stride = stride; p.To = (T - 1) / stride + 1;
    p.tiles_per_seq = (T + TT - 1) / TT;
    p.ntiles = N * p.tiles_per_seq;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (small) {
        hipLaunchKernelGGL((gcn_bwd_small_kernel<2>), dim3(g.nblk1), dim3(512), 0, st, p);
        rc = step::check_launch("step_gcn_bwd(first block)");
    } else {
        // (split operands: the 32 / 64-channel second form only; every other layer runs its fp32 kernels)
        rc = wide ? launch_gb16w(p, gw, st)
                  : (v2 ? (split ? dispatch_gb16<true>(p, g.nblk1, st) : dispatch_gb16<false>(p, g.nblk1, st))
                        : dispatch_gb(p, g, st));
    }
    if (rc) return rc;
    float* tot1 = p.part2 + (size_t)g.nblk2 * g.part2;   // (workspace rows: nblk1_ws + nblk2 + 2)
    float* tot2 = tot1 + g.part1;
    hipLaunchKernelGGL(sum_rows_kernel, dim3(step::ceil_div((int)g.part1, 64) + step::ceil_div((int)g.part2, 64)),
                       dim3(256), 0, st, p.part1, tot1, g.nblk1, (int)g.part1, p.part2, tot2, g.nblk2, (int)g.part2);
    FinParams f{};
    f.part1 = tot1; f.part2 = tot2; f.A = A_eff; f.b = b;
    f.dw = dw; f.db = db; f.dA = dA; f.dres_w = dres_w; f.dres_b = dres_b;
    f.nblk1 = 1; f.nblk2 = 1; f.emax = g.emax; f.copad = g.copad; f.cir = g.cir;
    f.Ci = Ci; f.Co = Co; f.K = K; f.w_transposed = w_transposed; f.res_transposed = res_transposed;
    f.have_res = dres != nullptr;
    const int total = K * Co * Ci + Co * Ci + Co + K * Co + K * 16;
    hipLaunchKernelGGL(gcn_bwd_finalize_kernel, dim3(step::ceil_div(total, 128)), dim3(128), 0, st, f);
    rc = step::check_launch("step_gcn_bwd(finalize)");
    if (rc == STEP_OK && id_dout && !id_fused)
        rc = step_masked_add(id_dout, id_out, dx, 1, (size_t)N * Ci * T * 16, stream);
    return rc;
}
}  // namespace

extern "C" int step_gcn_bwd_idres(const float* dy, const float* dres, const float* x, const float* in_scale,
                                  const float* in_shift, const float* w, const float* b, int w_transposed,
                                  const float* A_eff, int a_nnz_max, const float* res_w, int res_transposed,
                                  const float* id_dout, const float* id_out,
                                  float* dx, float* dw, float* db, float* dA, float* dres_w, float* dres_b,
                                  int N, int Ci, int Co, int K, int T, int stride, void* ws, size_t ws_bytes,
                                  void* stream) {
    return gcn_bwd(dy, dres, x, in_scale, in_shift, w, b, w_transposed, A_eff, a_nnz_max, res_w, res_transposed,
                   id_dout, id_out, dx, dw, db, dA, dres_w, dres_b, N, Ci, Co, K, T, stride, ws, ws_bytes, false, stream);
}

extern "C" int step_gcn_bwd_split(const float* dy, const float* dres, const float* x, const float* in_scale,
                                  const float* in_shift, const float* w, const float* b, int w_transposed,
                                  const float* A_eff, int a_nnz_max, const float* res_w, int res_transposed,
                                  const float* id_dout, const float* id_out,
                                  float* dx, float* dw, float* db, float* dA, float* dres_w, float* dres_b,
                                  int N, int Ci, int Co, int K, int T, int stride, void* ws, size_t ws_bytes,
                                  void* stream) {
    return gcn_bwd(dy, dres, x, in_scale, in_shift, w, b, w_transposed, A_eff, a_nnz_max, res_w, res_transposed,
                   id_dout, id_out, dx, dw, db, dA, dres_w, dres_b, N, Ci, Co, K, T, stride, ws, ws_bytes, true, stream);
}
