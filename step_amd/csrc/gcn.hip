// Fused graph convolution of the ST-GCN block (forward).
//
// Replaces ConvTemporalGraphical.forward (classifier_stgcn_real_only/net/utils/tgcn.py:58-67):
//     h = conv1x1(x)  (N, K*Co, T, V);  y = einsum('nkctv,kvw->nctw', h.view(N,K,Co,T,V), A)
// its transposed twin (tgcn.py:121-130), the `data_bn` affine that precedes block 0
// (classifier.py:68-74, folded into the load) and the residual 1x1 strided conv of
// st_gcn (classifier.py:175-179, shares the x tile).
//
// Aggregate-first identity:  y[c,(t,w)] = SUM_{k,ci} W[k*Co+c, ci] * z[(k,ci),(t,w)] + bterm[c,w]
//   z[(k,ci),(t,w)] = SUM_v x[ci,t,v] * A[k,v,w]        (A keeps the skeleton's sparsity)
//   bterm[c,w]      = SUM_k b[k*Co+c] * SUM_v A[k,v,w]
// The K*Co-channel intermediate `h` of the reference never exists.  The contraction over
// (k,ci) runs on the matrix cores (v_mfma_f32_32x32x2_f32: M = Co, N = 32 positions =
// 2 frames x 16 joints, k-step = 2 input channels); its B operand z is produced on the fly
// by each lane from the x tile in LDS with the <= EMAX non-zeros of column w of A_k held in
// registers, so A lives entirely on chip.
#include "common.h"
#include "split_mfma.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct GcnParams {
    const float* x;
    const float* in_scale;
    const float* in_shift;
    const float* w;
    const float* b;
    const float* A;
    float* y;
    const float* res_w;
    const float* res_b;
    float* res_out;
    int w_transposed, res_transposed;
    int N, Ci, Cip, Co, Copad, K, T, To, stride, tiles_per_seq, ntiles;
    double* ypart;      // fourth form only: per-block fp64 partials [blocks][Co][2] of SUM y, SUM y^2 (NULL: none)
    double* rpart;      // ... and of the residual conv's output
};

constexpr int GCN_TT = 8;             // frames per tile
constexpr int GCN_P = GCN_TT * 16;    // positions per tile

template <int EMAX, int MT, bool PF>
__global__ __launch_bounds__(256) void gcn_fwd_kernel(GcnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kq = p.K * p.Cip;
    float* Ws = smem;                                   // [Kq][Copad]
    float* xs = Ws + Kq * p.Copad;                      // [Cip][128]
    float* ellv = xs + p.Cip * GCN_P;                   // [K][EMAX][16]
    int* elli = reinterpret_cast<int*>(ellv + p.K * EMAX * 16);
    float* colsum = reinterpret_cast<float*>(elli + p.K * EMAX * 16);   // [K][16]
    __shared__ int kcnt[4];                             // widest column list of each partition k
    float* bterm = colsum + p.K * 16;                   // [Copad][16]
    float* Wr = bterm + p.Copad * 16;                   // [Cip][Copad]   (if residual conv)
    float* br = Wr + (p.res_w ? p.Cip * p.Copad : 0);   // [Copad]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int w16 = lane & 15;
    const int trow = 2 * wave + (l31 >> 4);

    // ---- one-time per block: weights, adjacency pattern, bias term -------------------
    for (int e = tid; e < Kq * p.Copad; e += 256) {
        const int c = e % p.Copad;
        const int q = e / p.Copad;
        const int ci = q % p.Cip, k = q / p.Cip;
        float v = 0.f;
        if (c < p.Co && ci < p.Ci)
            v = p.w_transposed ? p.w[(size_t)ci * (p.K * p.Co) + k * p.Co + c]
                               : p.w[((size_t)k * p.Co + c) * p.Ci + ci];
        Ws[e] = v;
    }
    if (p.res_w) {
        for (int e = tid; e < p.Cip * p.Copad; e += 256) {
            const int c = e % p.Copad, ci = e / p.Copad;
            float v = 0.f;
            if (c < p.Co && ci < p.Ci)
                v = p.res_transposed ? p.res_w[(size_t)ci * p.Co + c] : p.res_w[(size_t)c * p.Ci + ci];
            Wr[e] = v;
        }
        for (int c = tid; c < p.Copad; c += 256) br[c] = (c < p.Co && p.res_b) ? p.res_b[c] : 0.f;
    }
    if (tid < 4) kcnt[tid] = 1;
    __syncthreads();
    if (tid < p.K * 16) {
        const int k = tid >> 4, w = tid & 15;
        int cnt = 0;
        float cs = 0.f;
        for (int v = 0; v < 16; ++v) {
            const float a = p.A[(k * 16 + v) * 16 + w];
            cs += a;
            if (a != 0.f && cnt < EMAX) {
                ellv[(k * EMAX + cnt) * 16 + w] = a;
                elli[(k * EMAX + cnt) * 16 + w] = v;
                ++cnt;
            }
        }
        atomicMax(&kcnt[k], cnt);
        for (; cnt < EMAX; ++cnt) {
            ellv[(k * EMAX + cnt) * 16 + w] = 0.f;
            elli[(k * EMAX + cnt) * 16 + w] = 0;
        }
        colsum[k * 16 + w] = cs;
    }
    __syncthreads();
    for (int e = tid; e < p.Copad * 16; e += 256) {
        const int c = e >> 4, w = e & 15;
        float s = 0.f;
        if (c < p.Co && p.b)
            for (int k = 0; k < p.K; ++k) s += p.b[k * p.Co + c] * colsum[k * 16 + w];
        bterm[e] = s;
    }

    // ---- persistent loop over (n, 8-frame) tiles ---------------------------------------
    // the x tile of the NEXT tile is fetched into registers (raw) before this tile's MFMA phase, so
    // its HBM latency hides under the matrix work even at one block per CU
    constexpr int XN = PF ? 8 : 1;                        // float4 per thread: Cip <= 64
    float4 xr[XN];
    auto load_tile = [&](int tile) {
        const int n = tile / p.tiles_per_seq;
        const int t0 = (tile % p.tiles_per_seq) * GCN_TT;
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int f = tid + i * 256;
            const int ci = f >> 5;
            const int row = (f & 31) >> 2, vq = f & 3;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < p.Cip * 32 && ci < p.Ci && t0 + row < p.T)
                v = *reinterpret_cast<const float4*>(
                    p.x + (((size_t)n * p.Ci + ci) * p.T + t0 + row) * 16 + vq * 4);
            xr[i] = v;
        }
    };
    if (PF && (int)blockIdx.x < p.ntiles) load_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int n = tile / p.tiles_per_seq;
        const int t0 = (tile % p.tiles_per_seq) * GCN_TT;
        __syncthreads();                                  // previous tile's readers are done
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < XN; ++i) {
                const int f = tid + i * 256;
                if (f < p.Cip * 32) {
                    const int ci = f >> 5;
                    const int row = (f & 31) >> 2, vq = f & 3;
                    float4 v = xr[i];
                    if (p.in_scale && ci < p.Ci && t0 + row < p.T) {
                        const int g = vq * 4 * p.Ci + ci;
                        v.x = v.x * p.in_scale[g] + p.in_shift[g];
                        v.y = v.y * p.in_scale[g + p.Ci] + p.in_shift[g + p.Ci];
                        v.z = v.z * p.in_scale[g + 2 * p.Ci] + p.in_shift[g + 2 * p.Ci];
                        v.w = v.w * p.in_scale[g + 3 * p.Ci] + p.in_shift[g + 3 * p.Ci];
                    }
                    *reinterpret_cast<float4*>(xs + ci * GCN_P + row * 16 + vq * 4) = v;
                }
            }
            if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x);
        } else {
            for (int f = tid; f < p.Cip * 32; f += 256) {
                const int ci = f >> 5;
                const int row = (f & 31) >> 2, vq = f & 3;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ci < p.Ci && t0 + row < p.T) {
                    v = *reinterpret_cast<const float4*>(
                        p.x + (((size_t)n * p.Ci + ci) * p.T + t0 + row) * 16 + vq * 4);
                    if (p.in_scale) {
                        const int g = vq * 4 * p.Ci + ci;
                        v.x = v.x * p.in_scale[g] + p.in_shift[g];
                        v.y = v.y * p.in_scale[g + p.Ci] + p.in_shift[g + p.Ci];
                        v.z = v.z * p.in_scale[g + 2 * p.Ci] + p.in_shift[g + 2 * p.Ci];
                        v.w = v.w * p.in_scale[g + 3 * p.Ci] + p.in_shift[g + 3 * p.Ci];
                    }
                }
                *reinterpret_cast<float4*>(xs + ci * GCN_P + row * 16 + vq * 4) = v;
            }
        }
        __syncthreads();

        const bool live = (t0 + 2 * wave) < p.T;
        if (live) {
            f32x16 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            const float* xrow = xs + half * GCN_P + trow * 16;      // + 2*cs*128 per k-step
            const float* wrow = Ws + half * p.Copad + l31;
            for (int k = 0; k < p.K; ++k) {
                float av[EMAX];
                int ai[EMAX];
#pragma unroll
                for (int e = 0; e < EMAX; ++e) {
                    av[e] = ellv[(k * EMAX + e) * 16 + w16];
                    ai[e] = elli[(k * EMAX + e) * 16 + w16];
                }
                const float* wk = wrow + (size_t)k * p.Cip * p.Copad;
                // NE = the longest column list of A_k (block-uniform): the skeleton's partitions
                // have 1 / 3 / 1 entries, so most of the EMAX slots are never touched
                auto kphase = [&](auto ne_tag) {
                    constexpr int NE = decltype(ne_tag)::value;
                    for (int cs = 0; cs < p.Cip / 2; ++cs) {
                        const float* xr = xrow + cs * 2 * GCN_P;
                        float z = 0.f;
#pragma unroll
                        for (int e = 0; e < NE; ++e) z = fmaf(av[e], xr[ai[e]], z);
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const float a = wk[cs * 2 * p.Copad + m * 32];
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, z, acc[m], 0, 0, 0);
                        }
                    }
                };
                const int ne = kcnt[k];
                if (EMAX > 4 && ne > 4) kphase(std::integral_constant<int, EMAX>{});
                else if (ne <= 1) kphase(std::integral_constant<int, 1>{});
                else if (ne == 2) kphase(std::integral_constant<int, 2>{});
                else if (ne == 3) kphase(std::integral_constant<int, 3>{});
                else kphase(std::integral_constant<int, 4>{});
            }
            const int t = t0 + trow;
            if (t < p.T) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (c < p.Co)
                            p.y[(((size_t)n * p.Co + c) * p.T + t) * 16 + w16] =
                                acc[m][r] + bterm[c * 16 + w16];
                    }
            }
        }

        // ---- residual 1x1 conv with temporal stride, same x tile -------------------------
        if (p.res_w) {
            const int ncol = (p.stride == 1) ? 4 : 2;               // column tiles in this tile
            const int lrow = (p.stride == 1) ? trow : 2 * trow;     // local input frame
            if (wave < ncol && (t0 + p.stride * 2 * wave) < p.T) {
                f32x16 acc[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
                const float* xr = xs + half * GCN_P + lrow * 16 + w16;
                const float* wr = Wr + half * p.Copad + l31;
                for (int cs = 0; cs < p.Cip / 2; ++cs) {
                    const float bv = xr[cs * 2 * GCN_P];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float a = wr[cs * 2 * p.Copad + m * 32];
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[m], 0, 0, 0);
                    }
                }
                if (t0 + lrow < p.T) {
                    const int to = (t0 + lrow) / p.stride;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int c = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (c < p.Co)
                                p.res_out[(((size_t)n * p.Co + c) * p.To + to) * 16 + w16] =
                                    acc[m][r] + br[c];
                        }
                }
            }
        }
    }
}

size_t gcn_lds_bytes(const GcnParams& p, int emax) {
    size_t f = (size_t)p.K * p.Cip * p.Copad + (size_t)p.Cip * GCN_P + 2 * (size_t)p.K * emax * 16 +
               (size_t)p.K * 16 + (size_t)p.Copad * 16;
    if (p.res_w) f += (size_t)p.Cip * p.Copad;
    f += p.Copad;
    return f * 4;
}

template <int EMAX, int MT, bool PF>
int launch_gcn_pf(const GcnParams& p, hipStream_t st) {
    const size_t lds = gcn_lds_bytes(p, EMAX);
    if (lds > 160 * 1024) {
        step::set_error("step_gcn_fwd: needs %zu bytes of LDS (> 160 KiB) for Ci=%d Co=%d", lds, p.Ci, p.Co);
        return STEP_ERR_ARG;
    }
    auto kern = gcn_fwd_kernel<EMAX, MT, PF>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            step::set_error("step_gcn_fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return STEP_ERR_LAUNCH;
        }
    }
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    int blocks = 256 * per_cu;
    if (blocks > p.ntiles) blocks = p.ntiles;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, p);
    return step::check_launch("step_gcn_fwd");
}

// one block per CU (big LDS image): prefetch the next x tile into registers
template <int EMAX, int MT>
int launch_gcn(const GcnParams& p, hipStream_t st) {
    const bool one_per_cu = gcn_lds_bytes(p, EMAX) * 2 > 160 * 1024;
    return one_per_cu ? launch_gcn_pf<EMAX, MT, true>(p, st) : launch_gcn_pf<EMAX, MT, false>(p, st);
}


// ------------------------------------------------------------------------------------------
// Second form of the forward kernel for the wide blocks (Ci = 32 | 64, no data_bn on the load).
// Both contractions run on the matrix cores, and the first feeds the second without moving a byte:
//   z_k[(ci,t), w] = SUM_v x[ci,t,v] A[k,v,w]      v_mfma_f32_16x16x4_f32: a 16-row z tile = 8 channels x 2 frames,
//                                                  4 instructions (4 joints of v each); B operand = A_k (registers)
//   y[c,(t,w)]    += SUM_ci W[k*Co+c, ci] z_k[..]  v_mfma_f32_32x32x2_f32
// With row i = 4q + r of the z tile standing for (channel 2r + (q >> 1), frame q & 1), result register r of lane
// 16q + w IS the B operand of the main product's k-step r of that channel group (k index = lane >> 5 = channel
// parity, column = (frame, joint) = lane & 31).  The aggregation costs one 32-cycle MFMA per main k-step (shared
// by nothing, so +25 % MFMA cycles at two column tiles per wave) and replaces every gather / DPP / VALU step:
// any A, dense included, costs the same.  Where the operands live:
//   * W (and the residual conv's Wr) in REGISTERS for the life of the persistent block: a wave owns one
//     32-row M tile, so its A operand of (k, k-step s) is one register; A_k's fragments are 12 registers;
//   * the x tile arrives by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write) into a double buffer,
//     tile i+1 in flight while tile i is contracted - one barrier per tile, no exposed HBM latency.
// Wave w: M tile w % MT, position group w / MT = 2*MT frames = MT column tiles of 2 frames x 16 joints.
// RES: 0 none, 1 residual 1x1 conv at stride 1, 2 at stride 2 (MT == 2: the group's 2 even frames).
// OCC blocks per CU (registers and LDS allow two at Ci = 32)
template <int CIP, int MT, int RES>
constexpr int gcn2_occ() { return CIP == 32 ? 2 : 1; }   // Ci = 64: the two x images alone are 64 KiB of LDS

template <int CIP, int MT, int RES>
__global__ __launch_bounds__(256, (gcn2_occ<CIP, MT, RES>())) void gcn_fwd2_kernel(GcnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = CIP / 2;                          // k-steps (pairs of input channels)
    float* xs = smem;                                    // [2][CIP][128]
    float* colsum = xs + 2 * CIP * GCN_P;                // [3][16]
    float* bterm = colsum + 3 * 16;                      // [32*MT][16]
    float* brs = bterm + 32 * MT * 16;                   // [32*MT]
    float* stage = brs + 32 * MT;                        // [4 waves][32 rows][32*MT]: output transpose

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int w16 = lane & 15;
    const int mt = wave % MT;
    const int fbase = (wave / MT) * 2 * MT;              // first local frame of the wave's group
    const int frow = fbase + (l31 >> 4);                 // + 2*j for column tile j

    // x tile by LDS-DMA: one wave-instruction moves two channel rows (2 x 512 B = 8 frames x 16 joints)
    auto issue_dma = [&](int tile, int buf) {
        const int n = tile / p.tiles_per_seq;
        const int t0 = (tile % p.tiles_per_seq) * GCN_TT;
        int tf = t0 + ((lane & 31) >> 2);
        tf = tf < p.T ? tf : p.T - 1;                    // frames past the end: any valid address, never stored
#pragma unroll
        for (int i = 0; i < CIP / 8; ++i) {
            const int piece = wave * (CIP / 8) + i;
            const int ch = 2 * piece + half;
            const float* src = p.x + (((size_t)n * CIP + ch) * p.T + tf) * 16 + (lane & 3) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xs + buf * CIP * GCN_P + piece * 256),
                                             16, 0, 0);
        }
    };
    if ((int)blockIdx.x < p.ntiles) issue_dma(blockIdx.x, 0);

    // ---- one-time per block: adjacency fragments, bias term, operands in registers ----------------
    // B operand of the aggregation MFMA: A_k[v = 4*c4 + (lane >> 4)][w = lane & 15]
    float areg[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
            areg[k][c4] = (k < p.K) ? p.A[(k * 16 + 4 * c4 + (lane >> 4)) * 16 + w16] : 0.f;
    if (tid < 48) {
        const int k = tid >> 4, w = tid & 15;
        float cs = 0.f;
        if (k < p.K)
            for (int v = 0; v < 16; ++v) cs += p.A[(k * 16 + v) * 16 + w];
        colsum[tid] = cs;
    }
    __syncthreads();
    for (int e = tid; e < 32 * MT * 16; e += 256) {
        const int c = e >> 4, w = e & 15;
        float s = 0.f;
        if (c < p.Co && p.b)
            for (int k = 0; k < p.K; ++k) s += p.b[k * p.Co + c] * colsum[k * 16 + w];
        bterm[e] = s;
    }
    for (int c = tid; c < 32 * MT; c += 256) brs[c] = (RES != 0 && c < p.Co && p.res_b) ? p.res_b[c] : 0.f;

    float wreg[3][KS];
    {
        const int c = mt * 32 + l31;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ci = 2 * s + half;
                float v = 0.f;
                if (k < p.K && c < p.Co)
                    v = p.w_transposed ? p.w[(size_t)ci * (p.K * p.Co) + k * p.Co + c]
                                       : p.w[((size_t)k * p.Co + c) * CIP + ci];
                wreg[k][s] = v;
            }
    }
    float wres[RES ? KS : 1];
    if constexpr (RES != 0) {
        const int c = mt * 32 + l31;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ci = 2 * s + half;
            float v = 0.f;
            if (c < p.Co) v = p.res_transposed ? p.res_w[(size_t)ci * p.Co + c] : p.res_w[(size_t)c * CIP + ci];
            wres[s] = v;
        }
    }
    __syncthreads();                                     // tables visible; the first x tile has landed (vmcnt 0)

    constexpr int NRA = RES == 1 ? MT : 1;
    constexpr int PW = 32 * MT;                          // positions per wave (2*MT frames x 16 joints)
    constexpr int LPC = PW / 4;                          // lanes per channel row in a wide store
    const int lch = lane / LPC, lq = lane % LPC;
    // rows of PW floats from the wave's stage -> out[n, mt*32 + ch, tfirst .., :] as float4
    auto wide_store = [&](const float* stg, float* out, int Tout, int tfirst, int pw, bool tok, int n) {
#pragma unroll
        for (int i = 0; i < LPC / 2; ++i) {
            const int ch = i * (64 / LPC) + lch;
            const float4 v = *reinterpret_cast<const float4*>(stg + ch * pw + lq * 4);
            if (tok && mt * 32 + ch < p.Co)
                *reinterpret_cast<float4*>(out + (((size_t)n * p.Co + mt * 32 + ch) * Tout + tfirst) * 16 + lq * 4) = v;
        }
    };
    int buf = 0;
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int n = tile / p.tiles_per_seq;
        const int t0 = (tile % p.tiles_per_seq) * GCN_TT;
        if (tile + (int)gridDim.x < p.ntiles) issue_dma(tile + gridDim.x, buf ^ 1);
        const float* xb = xs + buf * CIP * GCN_P;

        f32x16 acc[MT], ra[NRA];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[j][r] = 0.f;
#pragma unroll
            for (int j = 0; j < NRA; ++j) ra[j][r] = 0.f;
        }
        // A operand of the aggregation MFMA (row i = lane & 15 of a 16-row z tile, k = lane >> 4): row
        // i = 4q + r is (channel 2r + (q >> 1) of the 8-channel group, frame q & 1), k is the joint v
        const float* xagg = xb + (2 * (lane & 3) + ((lane >> 3) & 1)) * GCN_P + (fbase + ((lane >> 2) & 1)) * 16 + (lane >> 4);
        const float* xlane = xb + half * GCN_P + frow * 16 + w16;          // residual conv: B operand = x as it lies
        const float* xeven = xb + half * GCN_P + (fbase + 2 * (l31 >> 4)) * 16 + w16;   // RES == 2: the even frames
#pragma unroll
        for (int g = 0; g < CIP / 8; ++g) {
            // the x fragments of this 8-channel group: one per (column tile, 4-joint chunk), shared by the
            // three partitions
            float xa[MT][4];
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) xa[j][c4] = xagg[g * 8 * GCN_P + j * 32 + c4 * 4];
            float xr[RES ? 4 : 1][NRA];
            if constexpr (RES == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < MT; ++j) xr[r][j] = xlane[(4 * g + r) * 2 * GCN_P + j * 32];
            } else if constexpr (RES == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[r][0] = xeven[(4 * g + r) * 2 * GCN_P];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // z = x A_k for 8 channels x 2 frames on v_mfma_f32_16x16x4_f32; its 4 result registers ARE
                // the B operands of the main product's k-steps 4g .. 4g+3 (lane = half*32 + frame*16 + joint)
                f32x4 z[MT];
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    z[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4)
                        z[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j][c4], areg[k][c4], z[j], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[k][4 * g + r], z[j][r], acc[j], 0, 0, 0);
            }
            if constexpr (RES == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        ra[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wres[4 * g + r], xr[r][j], ra[j], 0, 0, 0);
            } else if constexpr (RES == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ra[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wres[4 * g + r], xr[r][0], ra[0], 0, 0, 0);
            }
        }

        // ---- bias term, store ------------------------------------------------------------------------
        // The next tile's DMA (issued before the k loop) is waited for HERE, before this tile's stores are
        // issued: __syncthreads() with an LDS-DMA in flight drains vmcnt(0), which at the end of the tile would
        // also wait for every store of the epilogue to reach L2 - several microseconds per tile at one wave per
        // SIMD.  The barrier at the end of the tile is then a raw s_barrier.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // Stores go through a wave-private LDS stage: in the accumulator layout a lane holds ONE float of a
        // (channel, frame) row, and 4-byte global stores are issue-bound (a wave-instruction costs about the
        // same whatever its width); read back as float4 along the frame axis, one wave-instruction stores
        // 1 KiB (64 / LPC channel rows of PW*4 contiguous bytes) instead of 256 B.
        float* stg = stage + wave * (32 * PW);
        {
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = (r & 3) + 8 * (r >> 2) + 4 * half;
                    stg[cl * PW + j * 32 + l31] = acc[j][r] + bterm[(mt * 32 + cl) * 16 + w16];
                }
            wide_store(stg, p.y, p.T, t0 + fbase, PW, (t0 + fbase + lq / 4) < p.T, n);
        }

        // ---- residual 1x1 conv (temporal stride 1 | 2): accumulated beside the main product -------------
        if constexpr (RES == 1) {
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cl = (r & 3) + 8 * (r >> 2) + 4 * half;
                    stg[cl * PW + j * 32 + l31] = ra[j][r] + brs[mt * 32 + cl];
                }
            wide_store(stg, p.res_out, p.To, t0 + fbase, PW, (t0 + fbase + lq / 4) < p.T, n);
        } else if constexpr (RES == 2) {
            static_assert(RES != 2 || MT == 2, "stride-2 residual: 4 frames per wave");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = (r & 3) + 8 * (r >> 2) + 4 * half;
                stg[cl * 32 + l31] = ra[0][r] + brs[mt * 32 + cl];
            }
            // 32 channel rows x 2 output frames: 8 lanes per row
            const int lch8 = lane >> 3, lq8 = lane & 7;
            const bool tok = (t0 + fbase + 2 * (lq8 >> 2)) < p.T;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = i * 8 + lch8;
                const float4 v = *reinterpret_cast<const float4*>(stg + ch * 32 + lq8 * 4);
                if (tok && mt * 32 + ch < p.Co)
                    *reinterpret_cast<float4*>(p.res_out + (((size_t)n * p.Co + mt * 32 + ch) * p.To + ((t0 + fbase) >> 1)) * 16 + lq8 * 4) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // every wave's DMA has landed; this buffer is free
        asm volatile("" ::: "memory");
        buf ^= 1;
    }
}

template <int CIP, int MT, int RES>
int launch_gcn2(const GcnParams& p, hipStream_t st) {
    const size_t lds = ((size_t)2 * CIP * GCN_P + 3 * 16 + 32 * MT * 17 + 4 * 32 * 32 * MT) * 4;
    auto kern = gcn_fwd2_kernel<CIP, MT, RES>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            step::set_error("step_gcn_fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return STEP_ERR_LAUNCH;
        }
    }
    int blocks = 256 * gcn2_occ<CIP, MT, RES>();
    if (blocks > p.ntiles) blocks = p.ntiles;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, st, p);
    return step::check_launch("step_gcn_fwd");
}

// returns -1 when the shape is not one of the register-resident forms
template <int CIP>
int dispatch_gcn2(const GcnParams& p, hipStream_t st) {
    const int mt = p.Copad / 32;
    const int res = p.res_w ? p.stride : 0;
    if (mt == 1) {
        if (res == 0) return launch_gcn2<CIP, 1, 0>(p, st);
        if (res == 1) return launch_gcn2<CIP, 1, 1>(p, st);
        return -1;
    }
    if (res == 0) return launch_gcn2<CIP, 2, 0>(p, st);
    if (res == 1) return launch_gcn2<CIP, 2, 1>(p, st);
    return launch_gcn2<CIP, 2, 2>(p, st);
}


// ------------------------------------------------------------------------------------------
// Third form: the wide layers (Ci or Co > 64: the C = 128 / 256 blocks of the original ST-GCN schedule that
// the reference leaves commented out, classifier.py:42-49; north_star's MFMA bench layers).  W no longer fits
// registers, and at 3*Ci*Co*4 bytes not LDS either, so a block owns ONE output-channel group of COG = 16*COT
// channels (blockIdx.y) with that slice of W (and of the residual conv's Wr, as a 4th "partition" whose adjacency
// is the identity) in LDS, and every wave walks frames on its own (no barriers in the loop), all products on
// v_mfma_f32_16x16x4_f32 with one frame as the column tile:
//   z_k[ci,w]  = SUM_v x[ci,t,v] A_k[v,w]     A[i=ci][q] = component e of the lane's float4 x[ci][t][4q..4q+3],
//                                             B[q][j=w] = A_k[4q+e][w] (registers); D reg r of lane (q,w) = z_k[4q+r][w]
//   y[c,w]    += SUM_ci W_k[c,ci] z_k[ci,w]   A[i=c][q] = W_k[c][16*tile + 4q + r] (LDS), B = D register r of z_k as it lies
// The aggregation is recomputed per output group (1 / COT of the main product's MFMAs).
template <int COT>
__global__ __launch_bounds__(512, 2) void gcn_fwd16_kernel(GcnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CIT = p.Ci / 16;
    const int NP = p.res_w ? 4 : 3;
    float* Wl = smem;                                   // [NP][CIT][4 r][64 lanes][COT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int co0 = blockIdx.y * 16 * COT;

    for (int e = tid; e < NP * CIT * 4 * 64 * COT; e += 512) {
        const int ct = e % COT, ln = (e / COT) & 63, r = (e / (COT * 64)) & 3, tile = (e / (COT * 256)) % CIT,
                  kk = e / (COT * 256 * CIT);
        const int c = co0 + 16 * ct + (ln & 15), ci = 16 * tile + 4 * (ln >> 4) + r;
        float v = 0.f;
        if (kk < 3) {
            if (kk < p.K)
                v = p.w_transposed ? p.w[(size_t)ci * (p.K * p.Co) + kk * p.Co + c] : p.w[((size_t)kk * p.Co + c) * p.Ci + ci];
        } else {
            v = p.res_transposed ? p.res_w[(size_t)ci * p.Co + c] : p.res_w[(size_t)c * p.Ci + ci];
        }
        Wl[e] = v;
    }
    // B operand of the aggregation: A_k[v = 4q + e][w = l16]; the residual's "adjacency" is the identity
    float akb[4][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) akb[k][e] = (k < p.K) ? p.A[(k * 16 + 4 * q + e) * 16 + l16] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) akb[3][e] = (4 * q + e == l16) ? 1.f : 0.f;
    // bias term of this lane's outputs: c = co0 + 16 ct + 4 q + r, w = l16
    float bt[COT][4], brs[COT][4];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = co0 + 16 * ct + 4 * q + r;
            float s = 0.f;
            if (p.b)
                for (int k = 0; k < p.K; ++k) {
                    float cs = 0.f;
                    for (int v = 0; v < 16; ++v) cs += p.A[(k * 16 + v) * 16 + l16];
                    s += p.b[k * p.Co + c] * cs;
                }
            bt[ct][r] = s;
            brs[ct][r] = (p.res_w && p.res_b) ? p.res_b[c] : 0.f;
        }
    __syncthreads();

    const long long R = (long long)p.N * p.T;
    const long long nw = (long long)gridDim.x * 8, gw = (long long)blockIdx.x * 8 + wave;
    const long long r0 = R * gw / nw, r1 = R * (gw + 1) / nw;
    const int nrows = (int)(r1 - r0);
    int n = (int)(r0 / p.T), t = (int)(r0 - (long long)n * p.T);
    const unsigned TS = (unsigned)p.T * 64u, TOS = (unsigned)p.To * 64u;
    const unsigned lane_x = (unsigned)l16 * TS + q * 16u;                      // float4 of channel l16 (+16 tile)
    const unsigned lane_y = (unsigned)(co0 + 4 * q) * TS + l16 * 4u;           // dword of channel co0 + 4q (+16 ct + r)
    const unsigned lane_r = (unsigned)(co0 + 4 * q) * TOS + l16 * 4u;
    const float* wl = Wl + (size_t)lane * COT;

    for (int i = 0; i < nrows; ++i) {
        const bool res_row = p.res_w && (p.stride == 1 || (t & 1) == 0);
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)n * p.Ci * p.T * 16);
        const unsigned so = (unsigned)t * 64u;
        f32x4 acc[COT], accr[COT];
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) { acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; accr[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        f32x4 xq = bload4(rx, lane_x, so);
        for (int tile = 0; tile < CIT; ++tile) {
            const f32x4 xc = xq;
            if (tile + 1 < CIT) xq = bload4(rx, lane_x, so + (unsigned)(tile + 1) * 16u * TS);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 3 && !res_row) break;
                f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) z = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[e], akb[k][e], z, 0, 0, 0);
                const float* wk = wl + (size_t)((k * CIT + tile) * 4) * 64 * COT;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float wf[COT];
                    if constexpr (COT == 8) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(wk + (size_t)r * 64 * COT);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(wk + (size_t)r * 64 * COT + 4);
                        wf[0] = a[0]; wf[1] = a[1]; wf[2] = a[2]; wf[3] = a[3]; wf[4] = b[0]; wf[5] = b[1]; wf[6] = b[2]; wf[7] = b[3];
                    } else if constexpr (COT == 4) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(wk + (size_t)r * 64 * COT);
                        wf[0] = a[0]; wf[1] = a[1]; wf[2] = a[2]; wf[3] = a[3];
                    } else {
                        wf[0] = wk[(size_t)r * 64 * COT]; wf[1] = wk[(size_t)r * 64 * COT + 1];
                    }
                    if (k < 3) {
#pragma unroll
                        for (int ct = 0; ct < COT; ++ct)
                            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct], z[r], acc[ct], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int ct = 0; ct < COT; ++ct)
                            accr[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct], z[r], accr[ct], 0, 0, 0);
                    }
                }
            }
        }
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + (size_t)n * p.Co * p.T * 16);
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) bstore(ry, lane_y, so + (unsigned)(16 * ct + r) * TS, acc[ct][r] + bt[ct][r]);
        if (res_row) {
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(p.res_out + (size_t)n * p.Co * p.To * 16);
            const unsigned sr = (unsigned)(p.stride == 1 ? t : t >> 1) * 64u;
#pragma unroll
            for (int ct = 0; ct < COT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) bstore(rr, lane_r, sr + (unsigned)(16 * ct + r) * TOS, accr[ct][r] + brs[ct][r]);
        }
        if (++t == p.T) { t = 0; ++n; }
    }
}

// Split-operand twin of gcn_fwd16_kernel (step_amd.set_mfma_dtype "f32_split"): the four k-steps of every
// 16x16x4 chain become TWO v_mfma_f32_16x16x32_bf16 on operands split into bf16 hi + lo (any pairing of k works as
// long as A and B agree).  W lies in LDS already split ([.][lane][COT] x (hi4, lo4), the fp32 image's size); x and
// the aggregation's result are split in registers once and reused by every partition / output tile.
// RES: 0 no residual conv, 1 residual conv at stride 1, 2 at stride 2 (even frames only)
template <int COT, int RES>
__global__ __launch_bounds__(512, 1) void gcn_fwd16s_kernel(GcnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int FR = 4;                                  // frames per work unit (256 contiguous bytes per stored row)
    const int CIT = p.Ci / 16;
    constexpr int NP = RES ? 4 : 3;
    constexpr int FRR = RES == 2 ? FR / 2 : (RES ? FR : 1);
    SplitA* Wl = reinterpret_cast<SplitA*>(smem);          // [NP][CIT][COT][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int co0 = blockIdx.y * 16 * COT;
    // per-wave output stage behind the W image: [16 channels][FR frames][16 joints]
    float* stg = smem + (size_t)NP * CIT * 64 * COT * 4 + (size_t)wave * 16 * FR * 16;

    for (int e = tid; e < NP * CIT * 64 * COT; e += 512) {
        const int ln = e & 63, ct = (e >> 6) % COT, tile = (e / (COT * 64)) % CIT, kk = e / (COT * 64 * CIT);
        const int c = co0 + 16 * ct + (ln & 15);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = 16 * tile + 4 * (ln >> 4) + r;
            float x = 0.f;
            if (kk < 3) {
                if (kk < p.K)
                    x = p.w_transposed ? p.w[(size_t)ci * (p.K * p.Co) + kk * p.Co + c] : p.w[((size_t)kk * p.Co + c) * p.Ci + ci];
            } else {
                x = p.res_transposed ? p.res_w[(size_t)ci * p.Co + c] : p.res_w[(size_t)c * p.Ci + ci];
            }
            v[r] = x;
        }
        Wl[e] = split_a(v);
    }
    SplitB ak[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = k < 3 ? ((k < p.K) ? p.A[(k * 16 + 4 * q + e) * 16 + l16] : 0.f) : ((4 * q + e == l16) ? 1.f : 0.f);
        ak[k] = split_b(v);
    }
    // bias term of output channel c, joint w: SUM_k b[k,c] colsum(A_k)[w]; kept in LDS, read at the flush
    float* bias = smem + (size_t)NP * CIT * 64 * COT * 4 + 8 * 16 * FR * 16;    // [16 COT][16], then res bias [16 COT]
    for (int e = tid; e < 16 * COT * 16; e += 512) {
        const int c = co0 + (e >> 4), w = e & 15;
        float s = 0.f;
        if (p.b)
            for (int kk = 0; kk < p.K; ++kk) {
                float cs = 0.f;
                for (int v = 0; v < 16; ++v) cs += p.A[(kk * 16 + v) * 16 + w];
                s += p.b[kk * p.Co + c] * cs;
            }
        bias[e] = s;
    }
    for (int e = tid; e < 16 * COT; e += 512) bias[16 * COT * 16 + e] = (RES && p.res_b) ? p.res_b[co0 + e] : 0.f;
    __syncthreads();

    // work unit: FR consecutive frames of one sequence (the last unit of a sequence may be short)
    const int gps = (p.T + FR - 1) / FR;
    const long long G = (long long)p.N * gps;
    const long long nw = (long long)gridDim.x * 8, gw = (long long)blockIdx.x * 8 + wave;
    const long long g0 = G * gw / nw, g1 = G * (gw + 1) / nw;
    const int ngrp = (int)(g1 - g0);
    int n = (int)(g0 / gps), tg = (int)(g0 - (long long)n * gps);
    const unsigned TS = (unsigned)p.T * 64u, TOS = (unsigned)p.To * 64u;
    const unsigned lane_x = (unsigned)l16 * TS + q * 16u;
    const SplitA* wl = Wl + lane;

    // stage -> HBM for one 16-channel tile: 16-byte pieces, FR * 64 contiguous bytes per channel row
    auto flush = [&](float* base, unsigned row_bytes, int c_first, int f0, int nf) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base);
#pragma unroll
        for (int it = 0; it < 16 * FR * 4 / 64; ++it) {
            const int idx = it * 64 + lane;
            const int c = idx >> 4, piece = idx & 15;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + c * (FR * 16) + piece * 4);
            if ((piece >> 2) < nf)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                       (int)((unsigned)(c_first + c) * row_bytes + (unsigned)piece * 16u),
                                                       (int)((unsigned)f0 * 64u), 0);
        }
    };

    for (int i = 0; i < ngrp; ++i) {
        const int t0 = tg * FR;
        const int nf = p.T - t0 < FR ? p.T - t0 : FR;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)n * p.Ci * p.T * 16);
        f32x4 acc[FR][COT], accr[FRR][COT];
#pragma unroll
        for (int f = 0; f < FR; ++f)
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) acc[f][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < FRR; ++f)
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) accr[f][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int tile = 0; tile < CIT; ++tile) {
            f32x4 xq[FR];
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                const int tt = t0 + f < p.T ? t0 + f : p.T - 1;
                xq[f] = bload4(rx, lane_x, (unsigned)tt * 64u + (unsigned)tile * 16u * TS);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k == 3 && !RES) break;
                SplitA w[COT];
#pragma unroll
                for (int ct = 0; ct < COT; ++ct) w[ct] = wl[(size_t)((k * CIT + tile) * COT + ct) * 64];
#pragma unroll
                for (int f = 0; f < FR; ++f) {
                    // (residual conv at stride 2: odd frames are not outputs; t0 is a multiple of 4)
                    if (k == 3 && RES == 2 && (f & 1)) continue;
                    const SplitB zs = split_b(mfma_split(split_a(xq[f]), ak[k], f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
                    for (int ct = 0; ct < COT; ++ct) {
                        if (k < 3) acc[f][ct] = mfma_split(w[ct], zs, acc[f][ct]);
                        else if constexpr (RES != 0) {
                            constexpr int sh = RES == 2 ? 1 : 0;
                            accr[f >> sh][ct] = mfma_split(w[ct], zs, accr[f >> sh][ct]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
            for (int f = 0; f < FR; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    stg[(4 * q + r) * (FR * 16) + f * 16 + l16] = acc[f][ct][r] + bias[(16 * ct + 4 * q + r) * 16 + l16];
            flush(p.y + (size_t)n * p.Co * p.T * 16, TS, co0 + 16 * ct, t0, nf);
        }
        if constexpr (RES != 0) {
            constexpr int st = RES;
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
                for (int f = 0; f < FRR; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stg[(4 * q + r) * (FR * 16) + f * 16 + l16] = accr[f][ct][r] + bias[16 * COT * 16 + 16 * ct + 4 * q + r];
                flush(p.res_out + (size_t)n * p.Co * p.To * 16, TOS, co0 + 16 * ct, t0 / st, (nf + st - 1) / st);
            }
        }
        if (++tg == gps) { tg = 0; ++n; }
    }
}

// ------------------------------------------------------------------------------------------
// Fourth form (round 3): gcn_fwd16_kernel's products (fp32, v_mfma_f32_16x16x4_f32, one frame = one column tile) in
// gcn_fwd16s_kernel's work shape: a wave owns FR = 4 consecutive frames of one sequence and ALL output channels of
// the block's group, so
//   * the aggregation z_k = x A_k is computed once per (frame, partition, 16 input channels) - the second form
//     recomputed it per M tile (+50 % MFMA cycles at 32 -> 64);
//   * a W fragment read (one ds_read_b128 = 4 output tiles) serves 4 frames = 16 MFMAs instead of 4;
//   * results leave through a wave-private LDS stage as 16-byte pieces of 256-BYTE channel rows: the direct dword
//     stores of the third form are 64-byte pieces and hold the write path at 2.4 TB/s (3.28 ms at 32 -> 64, N = 4096,
//     6 of the 8 GB written), staged rows reach ~3.9 TB/s (first block).
// Optional epilogue (p.ypart != NULL): per-channel SUM y, SUM y^2 (and of the residual conv's output) for the
// train-mode BatchNorm that follows (tcn[0], residual[1]: classifier.py:154,180) - per-lane fp32 sums over at most
// STAT_FLUSH work units, then fp64: 16-lane DPP row sums into per-wave LDS accumulators, an ordered sum over the 8
// waves, one fp64 partial per (block, channel) for bn.hip's ordered finalize.  Bit-reproducible run to run.
// RES: 0 no residual conv, 1 at stride 1, 2 at stride 2 (even frames).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {          // every lane of a row of 16 gets the row's sum
    v += dpp_f<0x128>(v);                                       // row_ror:8
    v += dpp_f<0x124>(v);                                       // row_ror:4
    v += dpp_f<0x122>(v);                                       // row_ror:2
    v += dpp_f<0x121>(v);                                       // row_ror:1
    return v;
}

#ifndef F16_STORE_AUX
#define F16_STORE_AUX 0
#endif
constexpr int F16_FR = 4;                 // frames per work unit
constexpr int F16_ROW = F16_FR * 16 + 4;  // stage row stride in floats: +4 keeps the four q groups on distinct banks
constexpr int STAT_FLUSH = 16;            // work units between two flushes of the fp32 per-lane sums

template <int CIT, int COT, int RES>
__global__ __launch_bounds__(512, 1) void gcn_fwd16f_kernel(GcnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int FR = F16_FR;
    constexpr int NP = RES ? 4 : 3;
    constexpr int FRR = RES == 2 ? FR / 2 : (RES ? FR : 1);
    constexpr int COG = 16 * COT;
    float* Wl = smem;                                        // [NP][CIT][4 r][64 lanes][COT]
    float* stg_all = Wl + NP * CIT * 4 * 64 * COT;           // [8 waves][16 channels][F16_ROW]
    float* bias = stg_all + 8 * 16 * F16_ROW;                // [COG][16], then the residual conv's bias [COG]
    double* dacc_all = reinterpret_cast<double*>(bias + COG * 16 + COG);   // [8 waves][2 (y | r)][COG][2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int co0 = blockIdx.y * COG;
    float* stg = stg_all + wave * 16 * F16_ROW;
    double* dacc = dacc_all + (size_t)wave * 2 * COG * 2;
    const bool stats = p.ypart != nullptr;

    // The main product is taken TRANSPOSED (the two MFMA operands swapped): lane (q, c) of an output tile holds
    // y[c][w = 4q .. 4q+3] of its frame - a float4 for the stage, and ONE channel per lane for the statistics.
    // W fragment = B operand: lane (q, j = c) holds W_k[c][16 tile + 4 q + r].
    for (int e = tid; e < NP * CIT * 4 * 64 * COT; e += 512) {
        const int ct = e % COT, ln = (e / COT) & 63, r = (e / (COT * 64)) & 3, tile = (e / (COT * 256)) % CIT,
                  kk = e / (COT * 256 * CIT);
        const int c = co0 + 16 * ct + (ln & 15), ci = 16 * tile + 4 * (ln >> 4) + r;
        float v = 0.f;
        if (kk < 3) {
            if (kk < p.K)
                v = p.w_transposed ? p.w[(size_t)ci * (p.K * p.Co) + kk * p.Co + c] : p.w[((size_t)kk * p.Co + c) * p.Ci + ci];
        } else {
            v = p.res_transposed ? p.res_w[(size_t)ci * p.Co + c] : p.res_w[(size_t)c * p.Ci + ci];
        }
        Wl[e] = v;
    }
    // B operand of the aggregation: A_k[v = 4q + e][w = l16]; the residual's "adjacency" is the identity
    float akb[NP][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) akb[k][e] = (k < p.K) ? p.A[(k * 16 + 4 * q + e) * 16 + l16] : 0.f;
    if constexpr (RES != 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) akb[NP - 1][e] = (4 * q + e == l16) ? 1.f : 0.f;
    }
    for (int e = tid; e < COG * 16; e += 512) {
        const int c = co0 + (e >> 4), w = e & 15;
        float s = 0.f;
        if (p.b)
            for (int kk = 0; kk < p.K; ++kk) {
                float cs = 0.f;
                for (int v = 0; v < 16; ++v) cs += p.A[(kk * 16 + v) * 16 + w];
                s += p.b[kk * p.Co + c] * cs;
            }
        bias[e] = s;
    }
    for (int e = tid; e < COG; e += 512) bias[COG * 16 + e] = (RES && p.res_b) ? p.res_b[co0 + e] : 0.f;
    for (int e = tid; e < 8 * 2 * COG * 2; e += 512) dacc_all[e] = 0.0;
    __syncthreads();

    // work unit: FR consecutive frames of one sequence (the last unit of a sequence may be short)
    const int gps = (p.T + FR - 1) / FR;
    const long long G = (long long)p.N * gps;
    const long long nw = (long long)gridDim.x * 8, gw = (long long)blockIdx.x * 8 + wave;
    const long long g0 = G * gw / nw, g1 = G * (gw + 1) / nw;
    const int ngrp = (int)(g1 - g0);
    int n = (int)(g0 / gps), tg = (int)(g0 - (long long)n * gps);
    const unsigned TS = (unsigned)p.T * 64u, TOS = (unsigned)p.To * 64u;
    const unsigned lane_x = (unsigned)l16 * TS + q * 16u;     // float4 (joints 4q .. 4q+3) of channel l16 (+16 tile)
    const float* wl = Wl + (size_t)lane * COT;

    // stage -> HBM for one 16-channel tile: 16-byte pieces, FR * 64 contiguous bytes per channel row
    auto flush = [&](float* base, unsigned row_bytes, int c_first, int f0, int nf) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base);
#pragma unroll
        for (int it = 0; it < 16 * FR * 4 / 64; ++it) {
            const int idx = it * 64 + lane;
            const int c = idx >> 4, piece = idx & 15;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + c * F16_ROW + piece * 4);
#ifdef F16_ABL_NOSTORE
            if (v[0] == 1.2345e30f)
#else
            if ((piece >> 2) < nf)
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                       (int)((unsigned)(c_first + c) * row_bytes + (unsigned)piece * 16u),
                                                       (int)((unsigned)f0 * 64u), F16_STORE_AUX);
        }
    };
    // per-lane fp32 sums of this lane's channel 16 ct + l16 (joints 4q .. 4q+3).  Plain v_add_f32 / v_fmac_f32 through
    // inline asm: left to the compiler the adjacent adds are SLP-packed into v_pk_add_f32 / v_pk_fma_f32, which cost
    // an MFMA-paced SIMD about three times what two scalar instructions do (MI355X_MICROARCH.md, issue-cost table).
    float s1[COT], s2[COT], t1[RES ? COT : 1], t2[RES ? COT : 1];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
        s1[ct] = 0.f; s2[ct] = 0.f;
        if (RES) { t1[ct] = 0.f; t2[ct] = 0.f; }
    }
    auto stat4 = [&](float& a, float& b, const f32x4 v) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "v"(v[e]));
            asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(b) : "v"(v[e]));
        }
    };
    auto quad_sum = [&](float v) {                            // over the 4 lanes (q) that share a channel
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
    };
    auto stat_flush = [&]() {
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
            const float a = quad_sum(s1[ct]), b = quad_sum(s2[ct]);
            s1[ct] = 0.f; s2[ct] = 0.f;
            const int c = 16 * ct + l16;
            if (q == 0) { dacc[c * 2] += (double)a; dacc[c * 2 + 1] += (double)b; }
            if constexpr (RES != 0) {
                const float ar = quad_sum(t1[ct]), br = quad_sum(t2[ct]);
                t1[ct] = 0.f; t2[ct] = 0.f;
                if (q == 0) { dacc[(COG + c) * 2] += (double)ar; dacc[(COG + c) * 2 + 1] += (double)br; }
            }
        }
    };

    f32x4 xn[FR];                                             // x fragments of the NEXT tile (or unit), in flight
    auto load_x = [&](int nn, int t0, int tile) {
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)nn * p.Ci * p.T * 16);
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            const int tt = t0 + f < p.T ? t0 + f : p.T - 1;
            xn[f] = bload4(rx, lane_x, (unsigned)tt * 64u + (unsigned)tile * 16u * TS);
        }
    };
    if (ngrp > 0) load_x(n, tg * FR, 0);

    for (int i = 0; i < ngrp; ++i) {
        const int t0 = tg * FR;
        const int nf = p.T - t0 < FR ? p.T - t0 : FR;
        int n2 = n, tg2 = tg + 1;
        if (tg2 == gps) { tg2 = 0; ++n2; }
        if (i + 1 == ngrp) { n2 = n; tg2 = tg; }              // the last unit prefetches itself: unconditional loads
        // the accumulators start from the bias term (lane (q, c): bias[c][4q .. 4q+3]): no add in the epilogue
        f32x4 acc[FR][COT], accr[FRR][COT];
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + (16 * ct + l16) * 16 + 4 * q);
#pragma unroll
            for (int f = 0; f < FR; ++f) acc[f][ct] = bv;
            if constexpr (RES != 0) {
                const float br = bias[COG * 16 + 16 * ct + l16];
#pragma unroll
                for (int f = 0; f < FRR; ++f) accr[f][ct] = f32x4{br, br, br, br};
            }
        }
#pragma unroll
        for (int tile = 0; tile < CIT; ++tile) {
            f32x4 xq[FR];
#pragma unroll
            for (int f = 0; f < FR; ++f) xq[f] = xn[f];
            if (tile + 1 < CIT) load_x(n, t0, tile + 1);
            else load_x(n2, tg2 * FR, 0);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                // W fragments of (partition, tile): 4 k-steps x COT output tiles, read once for the 4 frames
                float wf[4][COT];
                const float* wk = wl + (size_t)((k * CIT + tile) * 4) * 64 * COT;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (COT == 4) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(wk + (size_t)r * 64 * COT);
                        wf[r][0] = a[0]; wf[r][1] = a[1]; wf[r][2] = a[2]; wf[r][3] = a[3];
                    } else {
                        wf[r][0] = wk[(size_t)r * 64 * COT]; wf[r][1] = wk[(size_t)r * 64 * COT + 1];
                    }
                }
                // the aggregation of the 4 frames as 4 interleaved chains (D register r of lane (q, w) =
                // z_k[16 tile + 4 q + r][w]), then the main product with z as the A operand
                // (two frames at a time: the z registers of a pair are dead before the next pair's chains start)
#pragma unroll
                for (int fp = 0; fp < FR; fp += 2) {
                    f32x4 z[2];
#pragma unroll
                    for (int f = 0; f < 2; ++f) z[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int f = 0; f < 2; ++f) {
                            if (k == 3 && RES == 2 && (f & 1)) continue;     // (stride 2: odd frames are not outputs)
                            z[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[fp + f][e], akb[k][e], z[f], 0, 0, 0);
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int f = 0; f < 2; ++f) {
                            if (k == 3 && RES == 2 && (f & 1)) continue;
#pragma unroll
                            for (int ct = 0; ct < COT; ++ct) {
                                if (k < 3)
                                    acc[fp + f][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[f][r], wf[r][ct], acc[fp + f][ct], 0, 0, 0);
                                else if constexpr (RES != 0) {
                                    constexpr int sh = RES == 2 ? 1 : 0;
                                    accr[(fp + f) >> sh][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[f][r], wf[r][ct], accr[(fp + f) >> sh][ct], 0, 0, 0);
                                }
                            }
                        }
                }
            }
        }
#pragma unroll
        for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
            for (int f = 0; f < FR; ++f)
                *reinterpret_cast<f32x4*>(stg + l16 * F16_ROW + f * 16 + 4 * q) = acc[f][ct];
#ifndef F16_ABL_NOSTATS
            if (stats) {
#pragma unroll
                for (int f = 0; f < FR; ++f)
                    if (f < nf) stat4(s1[ct], s2[ct], acc[f][ct]);
            }
#endif
            flush(p.y + (size_t)n * p.Co * p.T * 16, TS, co0 + 16 * ct, t0, nf);
        }
        if constexpr (RES != 0) {
            constexpr int st = RES;
            const int nfr = (nf + st - 1) / st;
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
#pragma unroll
                for (int f = 0; f < FRR; ++f)
                    *reinterpret_cast<f32x4*>(stg + l16 * F16_ROW + f * 16 + 4 * q) = accr[f][ct];
#ifndef F16_ABL_NOSTATS
                if (stats) {
#pragma unroll
                    for (int f = 0; f < FRR; ++f)
                        if (f < nfr) stat4(t1[ct], t2[ct], accr[f][ct]);
                }
#endif
                flush(p.res_out + (size_t)n * p.Co * p.To * 16, TOS, co0 + 16 * ct, t0 / st, nfr);
            }
        }
        if (stats && ((i + 1) % STAT_FLUSH == 0)) stat_flush();
        n = n2; tg = tg2;
    }
    if (stats) {
        stat_flush();
        __syncthreads();
        // ordered sum over the 8 waves; partial layout of bn.hip: [block][channel of Co][2]
        for (int e = tid; e < 2 * COG * 2; e += 512) {
            double s = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) s += dacc_all[(size_t)w8 * 2 * COG * 2 + e];
            const int which = e / (COG * 2), ce = e % (COG * 2);          // y | r, (channel, sum | sum of squares)
            double* dst = which ? p.rpart : p.ypart;
            if (dst) dst[((size_t)blockIdx.x * p.Co + co0) * 2 + ce] = s;
        }
    }
}

size_t gcn16f_lds(int cit, int cot, int np) {
    return ((size_t)np * cit * 4 * 64 * cot + 8 * 16 * F16_ROW + 16 * cot * 17) * sizeof(float) +
           (size_t)8 * 2 * 16 * cot * 2 * sizeof(double);
}

// blocks of the fourth form (also the number of statistic partials per channel): one 512-thread block per CU
int gcn16f_blocks(int N, int T, int groups) {
    const long long units = (long long)N * ((T + F16_FR - 1) / F16_FR);
    long long b = (units + 7) / 8;
    const long long cap = 256 / groups > 0 ? 256 / groups : 1;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// Ci, Co in {32, 64}; -2: shape not covered (the other forms take it)
int launch_gcn16f(const GcnParams& p, hipStream_t st) {
    if ((p.Ci != 32 && p.Ci != 64) || (p.Co != 32 && p.Co != 64) || p.in_scale) return -2;
    const int res = p.res_w ? p.stride : 0;
    const int cot = p.Co / 16, cit = p.Ci / 16;
    if (cot == 4 && res == 1) return -2;                      // 4 frames x (4 + 4) accumulator tiles do not fit
    const size_t lds = gcn16f_lds(cit, cot, res ? 4 : 3);
    const int b = gcn16f_blocks(p.N, p.T, 1);
    auto go = [&](auto kern) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            step::set_error("step_gcn_fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
            return (int)STEP_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)b, 1), dim3(512), lds, st, p);
        return step::check_launch("step_gcn_fwd(4-frame form)");
    };
#define F16_CASE(CITV, COTV, RESV) if (cit == CITV && cot == COTV && res == RESV) return go(gcn_fwd16f_kernel<CITV, COTV, RESV>)
    F16_CASE(2, 2, 0); F16_CASE(2, 2, 1); F16_CASE(2, 2, 2);
    F16_CASE(2, 4, 0); F16_CASE(2, 4, 2);
    F16_CASE(4, 2, 0); F16_CASE(4, 2, 1); F16_CASE(4, 2, 2);
    F16_CASE(4, 4, 0); F16_CASE(4, 4, 2);
#undef F16_CASE
    return -2;
}

// output-channel group of a wide launch: the largest of 128 / 64 / 32 that divides Co and whose W slice fits LDS
int gcn16_cog(int Ci, int Co, bool res) {
    const int np = res ? 4 : 3;
    for (int cog = 128; cog >= 32; cog >>= 1)
        if (Co % cog == 0 && (size_t)np * Ci * cog * 4 <= 144 * 1024) return cog;
    return 0;
}

int launch_gcn16(const GcnParams& p, hipStream_t st) {
    const int cog = gcn16_cog(p.Ci, p.Co, p.res_w != nullptr);
    if (!cog) {
        step::set_error("step_gcn_fwd: no wide kernel for Ci=%d Co=%d", p.Ci, p.Co);
        return STEP_ERR_ARG;
    }
    const size_t lds = (size_t)(p.res_w ? 4 : 3) * p.Ci * cog * 4;
    const long long rows = (long long)p.N * p.T;
    long long b = (rows + 15) / 16;
    const int groups = p.Co / cog;
    const long long cap = 256 / groups > 0 ? 256 / groups : 1;      // ~one 512-thread block per CU over all groups
    b = b < 1 ? 1 : (b > cap ? cap : b);
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                step::set_error("step_gcn_fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
                return (int)STEP_ERR_LAUNCH;
            }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)b, groups), dim3(512), lds, st, p);
        return step::check_launch("step_gcn_fwd(wide)");
    };
    if (cog == 128) return go(gcn_fwd16_kernel<8>);
    if (cog == 64) return go(gcn_fwd16_kernel<4>);
    return go(gcn_fwd16_kernel<2>);
}

// split-operand launch: output-channel groups of 64 (or 32): 4-frame accumulators for wider groups do not fit
int launch_gcn16s(const GcnParams& p, hipStream_t st) {
    const int np = p.res_w ? 4 : 3;
    int cog = 0;
    for (int c = 64; c >= 32; c >>= 1)
        if (p.Co % c == 0 && (size_t)np * p.Ci * c * 4 + 40 * 1024 <= 160 * 1024) { cog = c; break; }
    if (cog == 64 && p.res_w && p.stride == 1) cog = 32;      // 4 frames x (4 + 4) accumulator tiles do not fit
    if (!cog) return -2;                                      // W slice + stage do not fit the LDS: the fp32 kernels take it
    const size_t lds = (size_t)np * p.Ci * cog * 4 + (8 * 16 * 4 * 16 + cog * 17) * sizeof(float);
    const long long units = (long long)p.N * ((p.T + 3) / 4);
    long long b = (units + 7) / 8;
    const int groups = p.Co / cog;
    const long long cap = 256 / groups > 0 ? 256 / groups : 1;
    b = b < 1 ? 1 : (b > cap ? cap : b);
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                step::set_error("step_gcn_fwd_split: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
                return (int)STEP_ERR_LAUNCH;
            }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)b, groups), dim3(512), lds, st, p);
        return step::check_launch("step_gcn_fwd_split");
    };
    const int res = p.res_w ? p.stride : 0;
    if (cog == 64) return res == 0 ? go(gcn_fwd16s_kernel<4, 0>) : go(gcn_fwd16s_kernel<4, 2>);
    return res == 0 ? go(gcn_fwd16s_kernel<2, 0>) : res == 1 ? go(gcn_fwd16s_kernel<2, 1>) : go(gcn_fwd16s_kernel<2, 2>);
}

// First block (Ci <= 4: the coordinate channels, data_bn folded into the load; Co = 32 | 64; residual 1x1 conv):
// HBM-bound on the y / res_out writes.  Frame-per-wave, operands in registers, no barriers:
//   z_k[ci][w] = SUM_v x'[ci,v] A_k[v,w]   16x16x4 with x' as the A operand, rows i -> channel i & 3, so every row
//                                          group of the result holds all (<= 4) channels in its 4 registers
//   y[c][w]    = SUM_(k,ci) W_k[c,ci] z_k[ci][w]   ONE contraction over k' = (k, ci): row group q supplies partition q
//                                          (B = z_q, selected per lane), step e supplies channel e
//   r[c][w]    = SUM_ci Wr[c,ci] x'[ci,w]  one MFMA per 16 output channels
template <int COT>
__global__ __launch_bounds__(512, 2) void gcn_fwd_small_kernel(GcnParams p) {
    constexpr int CO = 16 * COT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l16 = lane & 15;
    const int Ci = p.Ci;
    const bool has_res = p.res_w != nullptr;

    float akb[3][4], wA[COT][4], wrA[COT], by[COT][4], br[COT][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) akb[k][e] = k < p.K ? p.A[(k * 16 + 4 * q + e) * 16 + l16] : 0.f;
    float cs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.f;
        if (k < p.K)
            for (int v = 0; v < 16; ++v) s += p.A[(k * 16 + v) * 16 + l16];
        cs[k] = s;
    }
#pragma unroll
    for (int ct = 0; ct < COT; ++ct) {
        const int c = 16 * ct + l16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = 0.f;
            if (q < p.K && e < Ci)
                v = p.w_transposed ? p.w[(size_t)e * (p.K * CO) + q * CO + c] : p.w[((size_t)q * CO + c) * Ci + e];
            wA[ct][e] = v;
        }
        float vr = 0.f;
        if (has_res && q < Ci) vr = p.res_transposed ? p.res_w[(size_t)q * CO + c] : p.res_w[(size_t)c * Ci + q];
        wrA[ct] = vr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cc = 16 * ct + 4 * q + r;
            float bsum = 0.f;
            if (p.b)
                for (int k = 0; k < p.K; ++k) bsum += p.b[k * CO + cc] * cs[k];
            by[ct][r] = bsum;
            br[ct][r] = (has_res && p.res_b) ? p.res_b[cc] : 0.f;
        }
    }
    const int cj = l16 & 3;
    float sq[4] = {1.f, 1.f, 1.f, 1.f}, hq[4] = {0.f, 0.f, 0.f, 0.f}, sd = 1.f, hd = 0.f;
    if (p.in_scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cj < Ci) { sq[e] = p.in_scale[(4 * q + e) * Ci + cj]; hq[e] = p.in_shift[(4 * q + e) * Ci + cj]; }
        if (q < Ci) { sd = p.in_scale[l16 * Ci + q]; hd = p.in_shift[l16 * Ci + q]; }
    }

    // work unit: FR = 4 consecutive frames of one sequence.  The wave's results for them go through its own LDS
    // stage [channel][frame][joint] and leave as 16-byte stores, 256 contiguous bytes per channel row (64-byte
    // pieces scattered over 32 channel rows run the HBM write path at 2.4 TB/s: measured)
    constexpr int FR = 4;
    __shared__ __attribute__((aligned(16))) float stage_all[8][CO * FR * 16];
    float* stg = stage_all[wave];
    // optional BatchNorm statistics of y and of the residual conv's output (p.ypart != NULL), as in the fourth form:
    // per-lane fp32 sums of channel 16 ct + 4 q + r at joint l16 over at most STAT_FLUSH units, then fp64 per wave
    __shared__ double dacc_all[8][2][CO][2];
    double* dacc = &dacc_all[wave][0][0][0];
    const bool stats = p.ypart != nullptr;
    for (int e = tid; e < 8 * 2 * CO * 2; e += 512) (&dacc_all[0][0][0][0])[e] = 0.0;
    __syncthreads();
    float s1[COT][4], s2[COT][4], t1[COT][4], t2[COT][4];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[ct][r] = 0.f; s2[ct][r] = 0.f; t1[ct][r] = 0.f; t2[ct][r] = 0.f; }
    auto stat1 = [&](float& sa, float& sb, float v) {
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(sa) : "v"(v));
        asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(sb) : "v"(v));
    };
    auto stat_flush = [&]() {
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * ct + 4 * q + r;
                const float a = row16_sum(s1[ct][r]), b2 = row16_sum(s2[ct][r]);
                s1[ct][r] = 0.f; s2[ct][r] = 0.f;
                if (l16 == 0) { dacc[c * 2] += (double)a; dacc[c * 2 + 1] += (double)b2; }
                if (has_res) {
                    const float ar = row16_sum(t1[ct][r]), br2 = row16_sum(t2[ct][r]);
                    t1[ct][r] = 0.f; t2[ct][r] = 0.f;
                    if (l16 == 0) { dacc[(CO + c) * 2] += (double)ar; dacc[(CO + c) * 2 + 1] += (double)br2; }
                }
            }
    };
    const int gps = (p.T + FR - 1) / FR;
    const long long G = (long long)p.N * gps;
    const long long nw = (long long)gridDim.x * 8, gw = (long long)blockIdx.x * 8 + wave;
    const long long g0 = G * gw / nw, g1 = G * (gw + 1) / nw;
    const int ngrp = (int)(g1 - g0);
    const unsigned TS = (unsigned)p.T * 64u, TOS = (unsigned)p.To * 64u;
    const unsigned lane_xq = (unsigned)(cj < Ci ? cj : 0) * TS + q * 16u;
    const unsigned lane_xd = (unsigned)(q < Ci ? q : 0) * TS + l16 * 4u;
    int n = (int)(g0 / gps), tg = (int)(g0 - (long long)n * gps);

    struct Grp { f32x4 xq[FR]; float xd[FR]; };
    auto load_grp = [&](int nn, int t0, Grp& gr) {
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (size_t)nn * Ci * p.T * 16);
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            const int tt = t0 + f < p.T ? t0 + f : p.T - 1;
            gr.xq[f] = bload4(rx, lane_xq, (unsigned)tt * 64u);
            gr.xd[f] = bload(rx, lane_xd, (unsigned)tt * 64u);
        }
    };
    // stage -> HBM: piece = 16 bytes of a (channel, frame) row; nf frames of the group are real
    auto flush = [&](float* base, unsigned row_bytes, int f0, int nf) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base);
#pragma unroll
        for (int it = 0; it < CO * FR * 4 / 64; ++it) {
            const int idx = it * 64 + lane;
            const int c = idx >> 4, piece = idx & 15;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + c * (FR * 16) + piece * 4);
            if ((piece >> 2) < nf)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                       (int)((unsigned)c * row_bytes + (unsigned)piece * 16u),
                                                       (int)((unsigned)f0 * 64u), 0);
        }
    };
    auto do_grp = [&](int nn, int t0, const Grp& cur) {
        const int nf = p.T - t0 < FR ? p.T - t0 : FR;
        float xds[FR];
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            float xq[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) xq[e] = cj < Ci ? cur.xq[f][e] * sq[e] + hq[e] : 0.f;
            xds[f] = q < Ci ? cur.xd[f] * sd + hd : 0.f;
            f32x4 z[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                z[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    z[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[e], akb[k][e], z[k], 0, 0, 0);
            }
            f32x4 zs;
#pragma unroll
            for (int e = 0; e < 4; ++e) zs[e] = q == 0 ? z[0][e] : (q == 1 ? z[1][e] : (q == 2 ? z[2][e] : 0.f));
#pragma unroll
            for (int ct = 0; ct < COT; ++ct) {
                f32x4 acc = {by[ct][0], by[ct][1], by[ct][2], by[ct][3]};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < Ci) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[ct][e], zs[e], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[(16 * ct + 4 * q + r) * (FR * 16) + f * 16 + l16] = acc[r];
                if (stats && f < nf) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) stat1(s1[ct][r], s2[ct][r], acc[r]);
                }
            }
        }
        flush(p.y + (size_t)nn * CO * p.T * 16, TS, t0, nf);
        if (has_res) {
            // residual frames of the group: t0 is a multiple of 4, so they start at To index t0 / stride
            const int st = p.stride;
#pragma unroll
            for (int f = 0; f < FR; f += 1) {
                if (st == 2 && (f & 1)) continue;
                const int fo = st == 2 ? f >> 1 : f;
#pragma unroll
                for (int ct = 0; ct < COT; ++ct) {
                    f32x4 acc = {br[ct][0], br[ct][1], br[ct][2], br[ct][3]};
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wrA[ct], xds[f], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[(16 * ct + 4 * q + r) * (FR * 16) + fo * 16 + l16] = acc[r];
                    if (stats && f < nf) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) stat1(t1[ct][r], t2[ct][r], acc[r]);
                    }
                }
            }
            flush(p.res_out + (size_t)nn * CO * p.To * 16, TOS, t0 / st, (nf + st - 1) / st);
        }
    };

    Grp ga, gb;
    if (ngrp > 0) load_grp(n, tg * FR, ga);
    for (int i = 0; i < ngrp; i += 2) {
        int n1 = n, tg1 = tg + 1;
        if (tg1 == gps) { tg1 = 0; ++n1; }
        if (i + 1 < ngrp) load_grp(n1, tg1 * FR, gb);
        do_grp(n, tg * FR, ga);
        n = n1; tg = tg1 + 1;
        if (tg == gps) { tg = 0; ++n; }
        if (i + 1 < ngrp) {
            if (i + 2 < ngrp) load_grp(n, tg * FR, ga);
            do_grp(n1, tg1 * FR, gb);
        }
        if (stats && ((i / 2 + 1) % (STAT_FLUSH / 2) == 0)) stat_flush();
    }
    if (stats) {
        stat_flush();
        __syncthreads();
        for (int e = tid; e < 2 * CO * 2; e += 512) {
            double sum = 0.0;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) sum += (&dacc_all[w8][0][0][0])[e];
            const int which = e / (CO * 2), ce = e % (CO * 2);
            double* dst = which ? p.rpart : p.ypart;
            if (dst) dst[(size_t)blockIdx.x * CO * 2 + ce] = sum;
        }
    }
}

int gcn_small_blocks(int N, int T) {
    const long long grps = (long long)N * ((T + 3) / 4);
    long long b = (grps + 15) / 16;
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

template <int COT>
int launch_gcn_small(const GcnParams& p, hipStream_t st) {
    const int b = gcn_small_blocks(p.N, p.T);
    hipLaunchKernelGGL((gcn_fwd_small_kernel<COT>), dim3((unsigned)b), dim3(512), 0, st, p);
    return step::check_launch("step_gcn_fwd(first block)");
}

}  // namespace

namespace {
// *fused_stats: in: statistic partials wanted (ypart / rpart below); out: whether the launched form wrote them
int gcn_fwd(const float* x, const float* in_scale, const float* in_shift, const float* w, const float* b,
            int w_transposed, const float* A_eff, int a_nnz_max, float* y, const float* res_w, const float* res_b,
            int res_transposed, float* res_out, int N, int Ci, int Co, int K, int T, int stride, bool split,
            void* stream, double* ypart = nullptr, double* rpart = nullptr, int* stat_blocks = nullptr) {
    STEP_REQUIRE(x && w && A_eff && y, "step_gcn_fwd: null pointer");
    STEP_REQUIRE(N > 0 && Ci > 0 && Co > 0 && T > 0, "step_gcn_fwd: bad shape N=%d Ci=%d Co=%d T=%d", N,
                 Ci, Co, T);
    STEP_REQUIRE(K >= 1 && K <= 3, "step_gcn_fwd: K=%d partitions (1..3 supported)", K);
    // the frame-per-wave form also wins at 64 -> 64 (block 2: 2.62 -> 2.22 ms at N = 4096: a quarter of the main
    // product's MFMAs go to the aggregation instead of half); 32 -> 64 stays with the second form (3.16 vs 3.28 ms)
    const bool wide = Ci > 64 || Co > 64 || (Ci == 64 && Co == 64 && !in_scale && !getenv("STEP_GCN_FWD_V1"));
    STEP_REQUIRE(!wide || (Ci % 16 == 0 && Co % 32 == 0 && Ci <= 256 && Co <= 256 && !in_scale),
                 "step_gcn_fwd: wide layers need Ci %% 16 == 0, Co %% 32 == 0, both <= 256, no data_bn (Ci=%d Co=%d)", Ci, Co);
    STEP_REQUIRE(stride == 1 || stride == 2, "step_gcn_fwd: stride %d", stride);
    STEP_REQUIRE(a_nnz_max >= 1 && a_nnz_max <= 16, "step_gcn_fwd: a_nnz_max=%d", a_nnz_max);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "step_gcn_fwd: in_scale/in_shift pair");
    STEP_REQUIRE((res_w == nullptr) == (res_out == nullptr), "step_gcn_fwd: res_w/res_out pair");
    GcnParams p{};
    p.x = x; p.in_scale = in_scale; p.in_shift = in_shift; p.w = w; p.b = b; p.A = A_eff; p.y = y;
    p.res_w = res_w; p.res_b = res_b; p.res_out = res_out;
    p.w_transposed = w_transposed; p.res_transposed = res_transposed;
    p.N = N; p.Ci = Ci; p.Cip = step::round_up(Ci, 2); p.Co = Co; p.Copad = step::round_up(Co, 32);
    p.K = K; p.T = T; p.stride = stride; p.To = (T - 1) / stride + 1;
    p.tiles_per_seq = step::ceil_div(T, GCN_TT);
    const long long nt = (long long)N * p.tiles_per_seq;
    STEP_REQUIRE(nt <= 0x7fffffffLL, "step_gcn_fwd: too many tiles");
    p.ntiles = (int)nt;
    hipStream_t st = (hipStream_t)stream;
    // split-operand mode: every layer the frame-per-wave form covers; the first block (HBM-bound) stays as it is
    if (split && !in_scale && Ci % 16 == 0 && Co % 32 == 0 && Ci <= 256 && Co <= 256) {
        const int rc = launch_gcn16s(p, st);
        if (rc != -2) return rc;
    }
    // round 3: the 4-frame form for the reference's inner widths (Ci, Co in {32, 64}); STEP_GCN_FWD_OLD=1 keeps the
    // second / third forms (A/B runs and the cross-check test)
    if (stat_blocks) *stat_blocks = 0;
    if (!getenv("STEP_GCN_FWD_OLD") && !getenv("STEP_GCN_FWD_V1") && !split) {
        p.ypart = ypart; p.rpart = rpart;
        const int rc = launch_gcn16f(p, st);
        if (rc != -2) {
            if (stat_blocks && ypart) *stat_blocks = gcn16f_blocks(N, T, 1);
            return rc;
        }
        p.ypart = p.rpart = nullptr;
    }
    if (wide) return launch_gcn16(p, st);
    if (Ci <= 4 && (Co == 32 || Co == 64) && !getenv("STEP_GCN_FWD_V1")) {
        if (!split && !getenv("STEP_GCN_FWD_OLD")) {             // first block: statistics in the epilogue too
            p.ypart = ypart; p.rpart = rpart;
            if (stat_blocks && ypart) *stat_blocks = gcn_small_blocks(N, T);
        }
        return Co == 32 ? launch_gcn_small<2>(p, st) : launch_gcn_small<4>(p, st);
    }
    const int mt = p.Copad / 32;
    if (!in_scale && (Ci == 32 || Ci == 64) && !getenv("STEP_GCN_FWD_V1")) {
        const int rc = Ci == 32 ? dispatch_gcn2<32>(p, st) : dispatch_gcn2<64>(p, st);
        if (rc != -1) return rc;
    }
    if (a_nnz_max <= 4) return mt == 1 ? launch_gcn<4, 1>(p, st) : launch_gcn<4, 2>(p, st);
    return mt == 1 ? launch_gcn<16, 1>(p, st) : launch_gcn<16, 2>(p, st);
}
}  // namespace

extern "C" int step_gcn_fwd(const float* x, const float* in_scale, const float* in_shift,
                            const float* w, const float* b, int w_transposed, const float* A_eff,
                            int a_nnz_max, float* y, const float* res_w, const float* res_b,
                            int res_transposed, float* res_out, int N, int Ci, int Co, int K, int T,
                            int stride, void* stream) {
    return gcn_fwd(x, in_scale, in_shift, w, b, w_transposed, A_eff, a_nnz_max, y, res_w, res_b, res_transposed,
                   res_out, N, Ci, Co, K, T, stride, false, stream);
}

extern "C" int step_gcn_fwd_split(const float* x, const float* in_scale, const float* in_shift,
                                  const float* w, const float* b, int w_transposed, const float* A_eff,
                                  int a_nnz_max, float* y, const float* res_w, const float* res_b,
                                  int res_transposed, float* res_out, int N, int Ci, int Co, int K, int T,
                                  int stride, void* stream) {
    return gcn_fwd(x, in_scale, in_shift, w, b, w_transposed, A_eff, a_nnz_max, y, res_w, res_b, res_transposed,
                   res_out, N, Ci, Co, K, T, stride, true, stream);
}

// half a workspace: 512 blocks of partials [Co][2] doubles (the first block's form launches up to 512), which also
// covers step_bn_ws_bytes(Co) for the shapes that take the separate statistics pass
static size_t stats_half(int Co) {
    const size_t a = (size_t)512 * Co * 2 * sizeof(double) + 64 * (size_t)Co, b = step_bn_ws_bytes(Co);
    return a > b ? a : b;
}
extern "C" size_t step_gcn_fwd_stats_ws_bytes(int Co) { return Co > 0 ? 2 * stats_half(Co) : 0; }

extern "C" int step_gcn_fwd_stats(const float* x, const float* in_scale, const float* in_shift,
                                  const float* w, const float* b, int w_transposed, const float* A_eff,
                                  int a_nnz_max, float* y, const float* res_w, const float* res_b,
                                  int res_transposed, float* res_out, int N, int Ci, int Co, int K, int T,
                                  int stride, double* y_sums, double* r_sums, void* ws, size_t ws_bytes,
                                  void* stream) {
    STEP_REQUIRE(y_sums && ws, "step_gcn_fwd_stats: null pointer");
    STEP_REQUIRE((res_w == nullptr) == (r_sums == nullptr), "step_gcn_fwd_stats: r_sums comes with the residual conv");
    STEP_REQUIRE(Co > 0, "step_gcn_fwd_stats: Co=%d", Co);
    const size_t half = stats_half(Co);
    if (ws_bytes < 2 * half) {
        step::set_error("step_gcn_fwd_stats: workspace %zu < %zu bytes", ws_bytes, 2 * half);
        return STEP_ERR_WORKSPACE;
    }
    double* ypart = (double*)ws;
    double* rpart = res_w ? (double*)((char*)ws + half) : nullptr;
    int blocks = 0;
    int rc = gcn_fwd(x, in_scale, in_shift, w, b, w_transposed, A_eff, a_nnz_max, y, res_w, res_b, res_transposed,
                     res_out, N, Ci, Co, K, T, stride, false, stream, ypart, rpart, &blocks);
    if (rc != STEP_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (blocks > 0) {
        // statistics taken in the producer's epilogue: one fp64 partial per (block, channel), summed in order
        rc = step::bn_sum_partials(ypart, y_sums, Co, blocks, st);
        if (rc == STEP_OK && rpart) rc = step::bn_sum_partials(rpart, r_sums, Co, blocks, st);
        return rc;
    }
    // forms without the epilogue: the separate statistics passes, same results layout
    rc = step_bn_train_stats(y, N, Co, T, STEP_V, 0, y_sums, ws, half, stream);
    if (rc == STEP_OK && res_w)
        rc = step_bn_train_stats(res_out, N, Co, (T - 1) / stride + 1, STEP_V, 0, r_sums, (char*)ws + half, half, stream);
    return rc;
}
