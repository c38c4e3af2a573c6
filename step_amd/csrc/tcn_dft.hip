// Temporal (Kt x 1) convolution in the frequency domain, fp32, every stage a GEMM on v_mfma_f32_32x32x2_f32.
//
// Same operator as step_tcn_conv / step_tcn_wgrad (tcn[2] of st_gcn / st_gctn, classifier_stgcn_real_only/net/
// classifier.py:156-162, generator_cvae/net/CVAE_stgcn.py:359-365, and both gradients).  A 75-tap correlation along t
// costs 75 multiply-adds per output, input and output channel; with the frames of a (sample, channel, joint) row taken
// to L >= T + (Kt-1)/2 frequency bins the channel mixing costs 3 real multiply-adds per BIN (a complex product as three
// real ones), and the two transforms are themselves small dense GEMMs against fixed cos / sin matrices (a "DFT by
// matmul": any T, no butterflies, exact zero padding, strides and ConvTranspose are just other matrices) of HALF the
// dense size: the frames are paired about their centre, the cos rows act on the pair sums, the sin rows on the pair
// differences.  For 64 channels, T = 240, stride 2 that is 7x fewer multiply-adds than the direct kernels execute
// (14x at T = 120, stride 1), all of them still fp32 MFMA work.
//
//   r[co, tau] = SUM_ci SUM_j w[co,ci,j] a[ci, tau + j - P]      (P = (Kt-1)/2, a = 0 outside [0, T); u[t'] = r[s t'])
//   a^[ci,f]   = SUM_t a[ci,t] e^{-2 pi i f t / L}                              step_dft_fwd   (matrix F, [2 NF x T])
//   r^[co,f]   = SUM_ci w^[co,ci,f] a^[ci,f],  w^[f] = SUM_j w[j] e^{+2 pi i f (j-P)/L}   step_dft_mix   (per bin f; 3 real products)
//   u[co,t']   = SUM_f kappa_f / L Re(r^[co,f] e^{+2 pi i f s t'/L}) + b[co]     step_dft_inv   (matrix G, [To x 2 NF])
// L = 2 NF - 1 is odd (no Nyquist bin; kappa_0 = 1, kappa_f = 2), 2 NF = round_up(T + P + 1, 4) (NF even); the circular
// correlation equals the linear one on tau in [0, T) because L >= T + P.  Backward:
//   du^[co,f]  = SUM_t' du[co,t'] e^{-2 pi i f s t'/L}                          step_dft_fwd   (matrix with rows at s t')
//   da^[ci,f]  = SUM_co conj(w^[co,ci,f]) du^[co,f];  da = inverse at full rate  step_dft_mix + step_dft_inv
//   C^[co,ci,f]= SUM_(n,v) conj(du^[co,f]) a^[ci,f]                              step_dft_wgrad (K = N * 16 per bin)
//   dw[co,ci,j]= SUM_f kappa_f / L Re(C^[f] e^{+2 pi i f (j-P)/L});  db[co] = SUM_(n,v) Re du^[co,0]   step_dft_taps
// Spectra live in HBM as S[n][f2][c][v] fp32, f2 = f (real part) | NF + f (imaginary part): 1.2x the size of the
// tensor they come from; a^ of the forward pass is kept for the weight gradient.  The cos / sin matrices and the tap
// tables are built by the host in float64 with integer-reduced angles (step_amd/dft.py) and passed in.
#include "common.h"
#include <cstdlib>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_n(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
static __device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ------------------------------------------------------------------------------------------------------------
// Left-multiplication of every sample's [K x (C * 16)] matrix by a fixed matrix, HALF the dense transform's size in
// each direction: the frames are paired about their centre (hi = Kh + j, lo = Kh - 1 - j, Kh = ceil(K / 2); the
// spectrum's time origin is that centre, s (Kh - 1/2), which costs nothing: the difference between the origins of a
// conv's input and output is a phase folded into the weight spectrum and the tap tables, step_amd/dft.py), so that
//   Re X[f] = SUM_j (x[hi] + x[lo]) cos(th_fj),   Im X[f] = - SUM_j (x[hi] - x[lo]) sin(th_fj),   th_fj = 2 pi f s (j + 1/2) / L
//   x[hi], x[lo] = E[j] +- O[j],   E[j] = SUM_f kap_f / L cos(th_fj) Re X[f],   O[j] = - SUM_f kap_f / L sin(th_fj) Im X[f]
//   INV == 0 (to the frequency domain): a block takes MG row tiles of the cos half (on the pair sums) and the same MG
//            tiles of the sin half (on the pair differences): one read of the frames feeds both;
//            In(n,k,c,v) = act(in[n][c][k][v]),  Out = S[n][m][c][v]
//   INV == 1 (back): rows = pairs j; the first NF k (Re rows of S) accumulate E, the last NF (Im rows) O; the epilogue
//            writes frames hi and lo: Out = out[n][c][frame][v] (+ bias, statistics, eval-mode tail)
//            (taking the real and the imaginary row of a bin in the same k-step, as the forward form does with its pair
//            sums and differences, measured 20 % slower: two distant row streams per wave instead of one)
// The matrix is stored k-major and tiled (step_dft_table_geom); rows past the real extent and k past K are zero.
// Block = 4 waves = (sample, 8 channels = 4 column tiles of 2 channels x 16 joints, group of MG row tiles); wave w owns
// column tile w and all row tiles of the group.  The matrix streams through LDS in chunks of KC = 16 k (double-buffered,
// one barrier per chunk); the B operand is the lane's own 4-byte loads (64 contiguous bytes per 16 lanes), a chunk ahead.
// (A persistent form - one block of 8 waves per CU, the whole matrix resident in LDS, waves walking the column tiles
// without a barrier - measured 5-25 % SLOWER on the same box at every shape of the model, with or without its stores,
// staggered or not; so did three blocks per CU.  Not kept.)
constexpr int KC = 16;

struct DgParams {
    const float* in;
    const float* matt;       // [groups][Kpad][slots WP x 32]: per k the row tiles in groups of four, [q][32 rows][4 tiles], then a last pair [32][2] (step_dft_table_geom)
    const float* in_scale;   // INV == 0: BN + ReLU on the load (per channel), or null
    const float* in_shift;
    const float* bias;       // INV == 1: per channel, or null
    const float* out_scale;  // INV == 1, eval-mode epilogue: out = relu?((acc + bias) * out_scale + out_shift + res * res_scale + res_shift)
    const float* out_shift;
    const float* res;        // (N, C, Mrows, 16) or null
    const float* res_scale;
    const float* res_shift;
    int out_relu;
    float* out;
    double* stat_part;       // INV == 1: [N * groups][C][2] partial SUM out, SUM out^2, or null
    int in_relu;
    // Segments (long sequences, stride 1: overlap-save): block z = sample * S + segment.  INV == 0 reads the K frames
    // frame0 + segment * seg_step + [0, K) of a channel row of Tfull frames (frames outside [0, Tfull) are the conv's zero
    // padding: masked AFTER the activation); INV == 1 writes its Mrows frames at segment * seg_step of a row of Tfull
    // frames (frames past the row's end are dropped).  One segment: S = 1, seg_step = frame0 = 0, Tfull = K / Mrows.
    int S, seg_step, frame0, Tfull;
    // Phases (a stride-2 conv as two stride-1 convs on the even and the odd frames, step_amd/dft.py): the tensor holds
    // fs = 2 interleaved sequences of Tfull frames per channel row; the spectrum has fs * C channels, channel fs * c + ph
    // for phase ph of channel c (a wave's 2 x 16 lanes then touch 128 contiguous bytes of the tensor: frames 2 t and
    // 2 t + 1 of one channel), every one an ordinary spectrum of its Tfull frames.  fs = 1: one sequence per row.
    int fs;
    int N, C, K, Kpad, Mpad, Mrows;   // INV == 0: K frames, Kpad = round_up(ceil(K/2), KC), Mrows = 2 NF; INV == 1: K = 2 NF rows of the
                                      // spectrum, Kpad = 2 round_up(NF, KC): the table holds each half (Re, Im) padded with zero rows to whole chunks, Mrows = T_out
};

// slots per k in the table for W row tiles: fours, then a pair (3 -> 4)
constexpr int table_slots(int W) { return (W % 4 == 0) ? W : (W % 4 == 2 ? W : W + 1); }

template <int MG, bool INV, int FS>      // FS = frames of the tensor per frame of a sequence: 1, or 2 (two interleaved phases per channel row)
__global__ __launch_bounds__(256, 2) void dft_gemm_kernel(DgParams p) {
    constexpr int MGR = MG * 32;
    constexpr int NACC = 2 * MG;                       // INV: E tiles, then O tiles; else cos tiles, then sin tiles
    constexpr int TABW = INV ? MG : 2 * MG;            // row tiles per k in the table (INV == 0: cos, then sin)
    constexpr int WP = table_slots(TABW);
    constexpr int NQ = WP / 4;                         // 16-byte reads per k: lanes 16 bytes apart, conflict-free
    constexpr bool PAIR = (WP % 4) == 2;               // + one 8-byte read
    constexpr int KSTR = WP * 32;                      // floats per k
    constexpr int CHUNK = KC * KSTR;
    __shared__ __attribute__((aligned(16))) float fbuf[2 * CHUNK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int cpar = l31 >> 4, v = l31 & 15;
    const int g = blockIdx.x;                           // row-tile group (fastest: groups share the B operand in L2)
    const int m0 = g * MGR;                             // first row of the group within its half / first pair
    const int nchunk = p.Kpad / KC;

    const int nz = blockIdx.z;                          // sample * S + segment: the spectrum's "sample"
    const int n = nz / p.S, seg = nz - n * p.S;
    const int cp = blockIdx.y * 8 + wave * 2 + cpar;    // channel of the spectrum = fs * channel of the tensor + phase
    const int c = cp >> (FS - 1), ph = cp & (FS - 1);                       // FS is 1 or 2
    const int CP = p.C * FS;
    constexpr unsigned fb = 64u * FS;                   // bytes from a frame of a phase to its next (compile-time: the k offsets stay immediates)

    // B operand addressing: descriptor on the sample, lane offsets, uniform k offsets
    const size_t sample = INV ? (size_t)CP * p.K * 16 : (size_t)CP * p.Tfull * 16;
    const __amdgpu_buffer_rsrc_t rs = rsrc_n(p.in + (size_t)(INV ? nz : n) * sample, (unsigned)(sample * 4));
    const int Kh = (p.K + 1) >> 1;                      // INV == 0: pairs of frames
    const int mf = p.frame0 + seg * p.seg_step;         // INV == 0: the window's first frame in the row (may be negative)
    const unsigned kstride = INV ? (unsigned)CP * 64u : fb;                         // bytes per k
    const unsigned voff = INV ? (unsigned)cp * 64u + half * kstride + v * 4u
                              : (unsigned)(((c * p.Tfull + mf + Kh + half) * FS + ph) * 64 + v * 4);       // frame hi of pair `half`
    const unsigned voff_lo = (unsigned)(((c * p.Tfull + mf + Kh - 1 - half) * FS + ph) * 64 + v * 4);       // frame lo
    float sc = 1.f, sh = 0.f;
    const bool act = !INV && p.in_scale != nullptr;
    if (act) { sc = p.in_scale[c]; sh = p.in_shift[c]; }
    const float lo = (!INV && p.in_relu) ? 0.f : -__builtin_inff();
    // pairs (k + half) whose hi / lo frame exists: inside the window (K odd: one hi fewer) and inside the row
    const int hi_lo = max(0, -(mf + Kh)), hi_n = max(0, min(p.K - Kh, p.Tfull - mf - Kh) - hi_lo);
    const int lo_lo = max(0, mf + Kh - p.Tfull), lo_n = max(0, min(Kh, mf + Kh) - lo_lo);

    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // a chunk of the table is one contiguous run: a linear copy, registers in between
    constexpr int FPT = (CHUNK / 4 + 255) / 256;
    f32x4 freg[FPT];
    const float* fsrc = p.matt + (size_t)g * p.Kpad * KSTR + tid * 4;
    auto load_f = [&](int kc) {
#pragma unroll
        for (int i = 0; i < FPT; ++i)
            if (CHUNK % 1024 == 0 || tid * 4 + i * 1024 < CHUNK)
                freg[i] = *reinterpret_cast<const f32x4*>(fsrc + (size_t)kc * CHUNK + i * 1024);
    };
    auto store_f = [&](int b) {
#pragma unroll
        for (int i = 0; i < FPT; ++i)
            if (CHUNK % 1024 == 0 || tid * 4 + i * 1024 < CHUNK)
                *reinterpret_cast<f32x4*>(&fbuf[b * CHUNK + tid * 4 + i * 1024]) = freg[i];
    };
    constexpr int NB = INV ? KC / 2 : KC;              // INV == 0: pair sums, then pair differences
    float bcur[NB], bnxt[NB];
    // INV: table k -> row of the spectrum: the Im half starts kpadh = round_up(NF, KC) - NF rows early; rows of a half's
    // zero padding are read from the last real pair of rows instead (finite values times the table's zeros; NF is even,
    // so a pair of rows is never half padding)
    [[maybe_unused]] const int nre = nchunk >> 1;
    [[maybe_unused]] const int kpadh = (p.Kpad - p.K) >> 1;
    auto load_b = [&](int kc, float (&b)[NB]) {
        [[maybe_unused]] int rowbase = 0;
        if constexpr (INV) {
            // first row of the chunk in the spectrum; a chunk that lies inside its half needs no clamp (uniform branch):
            // one scalar base, compile-time steps
            const int kin = (kc >= nre ? kc - nre : kc) * KC;
            rowbase = kc * KC - (kc >= nre ? kpadh : 0);
            if (kin + KC <= (p.K >> 1)) {
                const unsigned sb = (unsigned)rowbase * kstride;
#pragma unroll
                for (int ks = 0; ks < KC / 2; ++ks) b[ks] = bload(rs, voff, sb + (unsigned)(2 * ks) * kstride);
                return;
            }
        }
        if constexpr (!INV) {
            // a chunk whose 16 pairs all have both frames (every chunk but the last, and the first of a segment at the
            // row's start) needs no masks: 64 VALU instructions fewer beside the MFMA stream (uniform branch)
            const int k0 = kc * KC;
            if (k0 >= hi_lo && k0 + KC <= hi_lo + hi_n && k0 >= lo_lo && k0 + KC <= lo_lo + lo_n) {
#pragma unroll
                for (int ks = 0; ks < KC / 2; ++ks) {
                    const int k = k0 + 2 * ks;
                    float xh = bload(rs, voff + (unsigned)k * fb, 0u);
                    float xl = bload(rs, voff_lo - (unsigned)k * fb, 0u);
                    if (act) { xh = xh * sc + sh; xl = xl * sc + sh; }
                    xh = fmaxf(xh, lo); xl = fmaxf(xl, lo);
                    b[ks] = xh + xl;
                    b[KC / 2 + ks] = xh - xl;
                }
                return;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            const int k = kc * KC + 2 * ks;                                  // (+ half: in the lane offsets)
            if constexpr (INV) {
                b[ks] = bload(rs, voff, (unsigned)min(rowbase + 2 * ks, p.K - 2) * kstride);       // (a chunk with padding rows)
            } else {
                float xh = bload(rs, voff + (unsigned)k * fb, 0u);      // (offsets in the VGPR: the range check sees all of them)
                float xl = bload(rs, voff_lo - (unsigned)k * fb, 0u);
                if (act) { xh = xh * sc + sh; xl = xl * sc + sh; }
                xh = fmaxf(xh, lo); xl = fmaxf(xl, lo);
                xh = ((unsigned)(k + half - hi_lo) < (unsigned)hi_n) ? xh : 0.f;    // the odd frame out / k padding (the matrix is 0 there
                xl = ((unsigned)(k + half - lo_lo) < (unsigned)lo_n) ? xl : 0.f;    // too) / the conv's zero padding outside the row
                b[ks] = xh + xl;
                b[KC / 2 + ks] = xh - xl;
            }
        }
    };
    auto chunk = [&](int kc, f32x16* A) {
        const bool more = kc + 1 < nchunk;
        if (more) { load_f(kc + 1); load_b(kc + 1, bnxt); }
        const float* fb = fbuf + (kc & 1) * CHUNK + half * KSTR;
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            if (!INV && kc * KC + 2 * ks >= Kh) break;                       // k padding of the last chunk (uniform)
            float a[WP];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(fb + (2 * ks) * KSTR + q * 128 + l31 * 4);
                a[4 * q] = t[0]; a[4 * q + 1] = t[1]; a[4 * q + 2] = t[2]; a[4 * q + 3] = t[3];
            }
            if constexpr (PAIR) {
                const f32x2 t = *reinterpret_cast<const f32x2*>(fb + (2 * ks) * KSTR + NQ * 128 + l31 * 2);
                a[4 * NQ] = t[0]; a[4 * NQ + 1] = t[1];
            }
#pragma unroll
            for (int i = 0; i < MG; ++i)
                A[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bcur[ks], A[i], 0, 0, 0);
            if constexpr (!INV) {
#pragma unroll
                for (int i = MG; i < 2 * MG; ++i)
                    A[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bcur[KC / 2 + ks], A[i], 0, 0, 0);
            }
        }
        if (more) {
            store_f((kc + 1) & 1);
#pragma unroll
            for (int ks = 0; ks < NB; ++ks) bcur[ks] = bnxt[ks];
        }
        __syncthreads();
    };

    load_f(0);
    load_b(0, bcur);
    store_f(0);
    __syncthreads();
    if constexpr (!INV) {
        for (int kc = 0; kc < nchunk; ++kc) chunk(kc, acc);
    } else {
        for (int kc = 0; kc < nre; ++kc) chunk(kc, acc);
        for (int kc = nre; kc < nchunk; ++kc) chunk(kc, acc + MG);
    }

    if constexpr (!INV) {
        // S[n][m][c][v]: 32 lanes (2 channels x 16 joints) write 128 contiguous bytes per row; the cos half's descriptor
        // ends with its NF rows (NF is not a multiple of 32: its padding rows must not land in the sin half), the sin
        // half's with the sample
        const int NF = p.Mrows >> 1;
        const size_t osample = (size_t)p.Mrows * CP * 16;
        const unsigned rowb = (unsigned)CP * 64u;
        const unsigned ol = ((unsigned)(m0 + 4 * half) * (unsigned)CP * 16u + (unsigned)cp * 16u + v) * 4u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const __amdgpu_buffer_rsrc_t ro = rsrc_n(p.out + (size_t)nz * osample, (unsigned)((size_t)(h + 1) * NF * CP * 64));
            const unsigned oh = ol + (unsigned)(h * NF) * rowb;
#pragma unroll
            for (int i = 0; i < MG; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    bstore(ro, oh + (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * rowb, 0u, acc[h * MG + i][r]);
        }
    } else {
        const float bia = p.bias ? p.bias[c] : 0.f;
        float osc = 1.f, osh = 0.f, rsc = 1.f, rsh = 0.f;
        if (p.out_scale) { osc = p.out_scale[c]; osh = p.out_shift[c]; }
        if (p.res && p.res_scale) { rsc = p.res_scale[c]; rsh = p.res_shift[c]; }
        const bool epi = p.out_scale != nullptr || p.res != nullptr || p.out_relu != 0;       // (uniform)
        // descriptor on the sample; pair j -> frames Th + j and Th - 1 - j of the lane's channel row, both masked
        // (padding pairs of the last tile, the odd frame out)
        const int Th = (p.Mrows + 1) >> 1;
        const size_t osample = (size_t)CP * p.Tfull * 16;
        const int fo = seg * p.seg_step;                    // the segment's first frame in the output row
        const __amdgpu_buffer_rsrc_t ro = rsrc_n(p.out + (size_t)n * osample, (unsigned)(osample * 4));
        const __amdgpu_buffer_rsrc_t rr = rsrc_n(p.res ? p.res + (size_t)n * osample : p.out, p.res ? (unsigned)(osample * 4) : 0u);
        const int j0 = m0 + 4 * half;
        const unsigned ol_hi = (unsigned)((((c * p.Tfull + fo + Th + j0) * FS + ph) * 16 + v) * 4);
        const unsigned ol_lo = (unsigned)((((c * p.Tfull + fo + Th - 1 - j0) * FS + ph) * 16 + v) * 4);
        // pairs (relative to the lane's first) whose lo / hi frame exists - in the segment and in the row
        const int nlo = Th - j0, nhi_o = min(p.Mrows, p.Tfull - fo) - Th - j0, nlo_min = fo + Th - j0 - p.Tfull;
        float s1 = 0.f, s2 = 0.f;
        auto finish = [&](float x, unsigned off) {
            x += bia;
            if (epi) {
                x = x * osc + osh;
                if (p.res) x += bload(rr, off, 0u) * rsc + rsh;
                if (p.out_relu) x = fmaxf(x, 0.f);
            }
            bstore(ro, off, 0u, x);
            s1 += x; s2 += x * x;
        };
#pragma unroll
        for (int i = 0; i < MG; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mr = i * 32 + (r & 3) + 8 * (r >> 2);
                const float e = acc[i][r], o = acc[MG + i][r];
                if (mr < nhi_o) finish(e + o, ol_hi + (unsigned)mr * fb);
                if (mr < nlo && mr >= nlo_min) finish(e - o, ol_lo - (unsigned)mr * fb);
            }
        if (p.stat_part) {
            // per (block, channel): the 16 joints by DPP row rotations, the two halves by a shuffle; fp64 from here on
            auto rsum = [](float x) {
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, true));
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, true));
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, true));
                x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, true));
                return x;
            };
            float a1 = rsum(s1), a2 = rsum(s2);
            a1 += __shfl_xor(a1, 32, 64);
            a2 += __shfl_xor(a2, 32, 64);
            if (half == 0 && v == 0) {
                double* d = p.stat_part + (((size_t)nz * gridDim.x + g) * p.C + c) * 2;
                d[0] = (double)a1; d[1] = (double)a2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Channel mixing per frequency bin: R[f] = W[f] X[f], complex (rows x K) times (K x columns), with THREE real products
// instead of four (Gauss): with W = a + i b per (row, k) and X = Xr + i Xi per (k, column)
//   k1 = (a + b) Xr,   k2 = a (Xi - Xr),   k3 = b (Xr + Xi);     Re R = k1 - k3,   Im R = k1 + k2
// The weights come as the triples (a + b, a, b, 0) per (row, k) (step_dft_wspec), the two operand combinations are two
// VALU instructions per loaded pair: 25 % fewer MFMAs than the real block form [[a, -b], [b, a]] - this kernel runs
// with its MFMA pipe ~80 % busy at a power-limited clock, so the multiply count is what its time follows.
// Block = (bin, slice of the batch): the bin's triples sit in LDS ([row tile][k / 2][k % 2][32 rows][4]: a lane's three
// A values for one k are one 16-byte read); a wave owns one tile of 32 output rows (both its real and its imaginary
// result) and walks pairs of samples (32 columns = 2 samples x 16 joints) - no barrier in the loop.  All operand values
// of a pair are requested at once, the next pair's while this one multiplies.
struct MxParams {
    const float* in;
    const float* wb;     // [NF][rows / 32][K / 2][2][32][4]
    float* out;
    int N, Ci, Co, NF;
};

// Rectangular bins (the two phases of a stride-2 conv, step_amd/dft.py: 64 x 128 forward, 128 x 64 backward) hold 128 KB
// of triples: one block of 8 waves per CU instead of two of 4; K = 128 goes through the registers in two halves (the
// second half of a pair is requested while the first multiplies, the next pair's first half while the second does).
constexpr int mix_waves(int CT, int CI) { return CT * CI > 128 ? 8 : 4; }

template <int CT, int CI>        // row tiles = Co / 32, input channels
__global__ __launch_bounds__(64 * mix_waves(CT, CI), 8 / mix_waves(CT, CI)) void dft_mix_kernel(MxParams p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];          // CT * CI * 128 floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int npar = l31 >> 4, v = l31 & 15;
    const int f = blockIdx.x;
    constexpr int WAVES = mix_waves(CT, CI);
    constexpr int STREAMS = WAVES / CT;                      // independent pair streams per block
    constexpr int WTOT = CT * CI * 128;
    constexpr int KH = CI > 64 ? CI / 64 : 1;                // K in halves of 64
    constexpr int CIH = CI / KH;
    static_assert(KH == 1 || KH == 2, "K = 128 at most");
    const int ct = wave % CT, stream = wave / CT;
    {
        const float* wsrc = p.wb + (size_t)f * WTOT;
        for (int i = tid * 4; i < WTOT; i += WAVES * 256)
            *reinterpret_cast<f32x4*>(&wl[i]) = *reinterpret_cast<const f32x4*>(wsrc + i);
    }
    __syncthreads();
    const float* wlane = wl + ((ct * (CI / 2)) * 2 + half) * 128 + l31 * 4;       // + kk * 256
    const int npairs = (p.N + 1) / 2;
    const int per = (npairs + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(npairs, p0 + per);
    const int rows = 2 * p.NF;
    const size_t in_sample = (size_t)rows * CI * 16, out_sample = (size_t)rows * p.Co * 16;
    // Addresses: a buffer descriptor on the PAIR (uniform; its size covers the one or two samples that exist, so the
    // missing half of a last odd pair loads zeros and drops its stores), 32-bit lane offsets computed once, compile-time
    // k offsets - no per-access VALU work.
    const unsigned in_lane = (unsigned)(npar * in_sample + (f * CI + half) * 16 + v) * 4u;
    const unsigned part_jump = (unsigned)(p.NF * CI * 16) * 4u;
    const unsigned im_jump = (unsigned)(p.NF * p.Co * 16) * 4u;
    unsigned out_lane[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = ct * 32 + rowmap(r, half);
        out_lane[r] = (unsigned)(npar * out_sample + ((size_t)f * p.Co + co) * 16 + v) * 4u;
    }

    float bc[CIH], bn[CIH];                                  // [0, CIH/2): real rows (k = 2 ks + half), then imaginary rows
    auto load_pair = [&](int pr, int h, float (&b)[CIH]) {   // K half h: channels h * CIH + [0, CIH)
        const int have = min(2, p.N - 2 * pr);
        const __amdgpu_buffer_rsrc_t rs = rsrc_n(p.in + (size_t)(2 * pr) * in_sample, (unsigned)(have * in_sample * 4));
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
            for (int ks = 0; ks < CIH / 2; ++ks)
                b[part * (CIH / 2) + ks] = bload(rs, in_lane + part * part_jump + (unsigned)(h * CIH * 64), (unsigned)(ks * 128));
    };
    int pr = p0 + stream;
    if (pr < p1) load_pair(pr, 0, bc);
    if constexpr (KH == 1) {
        // (the loop bodies are written out, not shared through lambdas: the compiler then packs the operand sums /
        // differences and the epilogue's adds in pairs - 8 VALU instructions fewer per pair)
        for (; pr < p1; pr += STREAMS) {
            if (pr + STREAMS < p1) load_pair(pr + STREAMS, 0, bn);
            f32x16 q1, q2, q3;
#pragma unroll
            for (int r = 0; r < 16; ++r) { q1[r] = 0.f; q2[r] = 0.f; q3[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < CI / 2; ++kk) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(wlane + kk * 256);
                const float xr = bc[kk], xi = bc[CI / 2 + kk];
                q1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], xr, q1, 0, 0, 0);
                q2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], xi - xr, q2, 0, 0, 0);
                q3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], xr + xi, q3, 0, 0, 0);
                if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);           // keep the LDS reads 4 k-steps ahead at most (registers)
            }
            const int have = min(2, p.N - 2 * pr);
            const __amdgpu_buffer_rsrc_t ro = rsrc_n(p.out + (size_t)(2 * pr) * out_sample, (unsigned)(have * out_sample * 4));
            const f32x16 re = q1 - q3, im = q1 + q2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bstore(ro, out_lane[r], 0u, re[r]);
                bstore(ro, out_lane[r], im_jump, im[r]);          // (the uniform jump in the SGPR offset: the pair's descriptor covers it)
            }
            if (pr + STREAMS < p1) {
#pragma unroll
                for (int kk = 0; kk < CI; ++kk) bc[kk] = bn[kk];
            }
        }
    } else {
        // K = 128: the second half of a pair is requested while its first half multiplies, the next pair's first half
        // while the second does - the two register sets swap roles, no copies
#define MIX_MAC(B, H)                                                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < CIH / 2; ++kk) {                                                       \
            const f32x4 w = *reinterpret_cast<const f32x4*>(wlane + ((H) * (CIH / 2) + kk) * 256);                     \
            const float xr = B[kk], xi = B[CIH / 2 + kk];                                                              \
            q1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], xr, q1, 0, 0, 0);                                          \
            q2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], xi - xr, q2, 0, 0, 0);                                     \
            q3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], xr + xi, q3, 0, 0, 0);                                     \
            if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);                                                      \
        }
        for (; pr < p1; pr += STREAMS) {
            load_pair(pr, 1, bn);
            f32x16 q1, q2, q3;
#pragma unroll
            for (int r = 0; r < 16; ++r) { q1[r] = 0.f; q2[r] = 0.f; q3[r] = 0.f; }
            MIX_MAC(bc, 0)
            if (pr + STREAMS < p1) load_pair(pr + STREAMS, 0, bc);
            MIX_MAC(bn, 1)
            const int have = min(2, p.N - 2 * pr);
            const __amdgpu_buffer_rsrc_t ro = rsrc_n(p.out + (size_t)(2 * pr) * out_sample, (unsigned)(have * out_sample * 4));
            const f32x16 re = q1 - q3, im = q1 + q2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bstore(ro, out_lane[r], 0u, re[r]);
                bstore(ro, out_lane[r], im_jump, im[r]);
            }
        }
#undef MIX_MAC
    }
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradient per bin: C^[f] = SUM_(n, v) conj(D) A (D = du^, A = a^), the same three-product form:
//   P1 = SUM (Dr - Di) Ar,   P2 = SUM Dr (Ai - Ar),   P3 = SUM Di (Ar + Ai);     Re C^ = P1 + P3,   Im C^ = P1 + P2
// part[f][split][3][Co][Ci].  A lane loads the 8 joints 8 half .. 8 half + 7 of its row as two float4; MFMA step e pairs
// joint e (k = 0) with joint 8 + e (k = 1) in BOTH operands.  Block = (bin, slice of the batch); wave (wm, wn) owns the
// (co tile wm, ci tile wn) pair of 32 x 32 tiles.
struct WgParams {
    const float* ds;     // du^  [N][2 NF][Co][16]
    const float* s;      // a^   [N][2 NF][Ci][16]
    float* part;         // [NF][nsplit][3][Co][Ci]
    int N, Ci, Co, NF, nsplit;
};

template <int CT, int CTN>    // tiles per side: Co / 32, Ci / 32 (Ci = 2 Co: the two phases of a stride-2 conv); block = CT * CTN waves
__global__ __launch_bounds__(64 * CT * CTN) __attribute__((amdgpu_num_vgpr(128))) void dft_wgrad_kernel(WgParams p) {      // 4 waves per SIMD
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int f = blockIdx.x, split = blockIdx.y;
    const int wm = wave / CTN, wn = wave % CTN;
    const int n0 = (int)((long long)p.N * split / p.nsplit), n1 = (int)((long long)p.N * (split + 1) / p.nsplit);
    const size_t ds_sample = (size_t)2 * p.NF * p.Co * 16, s_sample = (size_t)2 * p.NF * p.Ci * 16;
    // rows: real part at bin f, imaginary part NF rows further
    const unsigned aoff = (unsigned)((f * p.Co + wm * 32 + l31) * 16 + 8 * half), ajump = (unsigned)(p.NF * p.Co * 16);
    const unsigned boff = (unsigned)((f * p.Ci + wn * 32 + l31) * 16 + 8 * half), bjump = (unsigned)(p.NF * p.Ci * 16);
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    f32x4 x[8], xn[8];                   // Dr (2), Di (2), Ar (2), Ai (2)
    auto load = [&](int n, f32x4 (&y)[8]) {
        const float* da = p.ds + (size_t)n * ds_sample + aoff;
        const float* sb = p.s + (size_t)n * s_sample + boff;
        y[0] = *reinterpret_cast<const f32x4*>(da);
        y[1] = *reinterpret_cast<const f32x4*>(da + 4);
        y[2] = *reinterpret_cast<const f32x4*>(da + ajump);
        y[3] = *reinterpret_cast<const f32x4*>(da + ajump + 4);
        y[4] = *reinterpret_cast<const f32x4*>(sb);
        y[5] = *reinterpret_cast<const f32x4*>(sb + 4);
        y[6] = *reinterpret_cast<const f32x4*>(sb + bjump);
        y[7] = *reinterpret_cast<const f32x4*>(sb + bjump + 4);
    };
    // (the two operand sets swap roles from sample to sample: no register copies in the MFMA stream - every
    // instruction of a SIMD's waves costs its MFMA stream the issue time)
    auto do_sample = [&](int n, const f32x4 (&cur)[8], f32x4 (&nxt)[8]) {
        if (n + 1 < n1) load(n + 1, nxt);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 dr = cur[h], di = cur[2 + h], ar = cur[4 + h], ai = cur[6 + h];
            const f32x4 dm = dr - di, bd = ai - ar, bs = ar + ai;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dm[e], ar[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dr[e], bd[e], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(di[e], bs[e], acc[2], 0, 0, 0);
            }
        }
    };
    if (n0 < n1) load(n0, x);
    for (int n = n0; n < n1; n += 2) {
        do_sample(n, x, xn);
        if (n + 1 < n1) do_sample(n + 1, xn, x);
    }
    float* dst = p.part + ((size_t)f * p.nsplit + split) * 3 * p.Co * p.Ci;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wm * 32 + rowmap(r, half), ci = wn * 32 + l31;
            dst[((size_t)i * p.Co + co) * p.Ci + ci] = acc[i][r];
        }
}

// The same products with the operands shared through LDS (blocks of more than one wave).  In the form above every wave
// requests its own (co tile, ci tile) rows from memory: a du^ tile is requested by CTN waves, an a^ tile by CT, and most
// of those requests miss in L2 while the first is in flight - PMC at 64 x 128: 13.3 GB moved for 7.05 GB, 132 M TCC
// requests / 104 M misses for 55 M distinct lines.  Here the block copies the bin's rows of one sample - du^ re | im
// (2 Co rows), a^ re | im (2 Ci rows), 64 bytes each, every group contiguous in HBM - into LDS once (coalesced 16-byte
// loads, two samples ahead in registers), rows at a pitch of 80 bytes: a wave's 16-byte fragment reads (lane = row, the two
// halves 32 bytes apart) then touch every bank once per 16 lanes.  Double-buffered, one barrier per sample.
template <int CT, int CTN>
__global__ __launch_bounds__(64 * CT * CTN) __attribute__((amdgpu_num_vgpr(128))) void dft_wgrad_lds_kernel(WgParams p) {
    constexpr int NT = 64 * CT * CTN;
    constexpr int CO = 32 * CT, CI = 32 * CTN;
    constexpr int ROWS = 2 * (CO + CI);                 // du^ re, du^ im, a^ re, a^ im
    constexpr int PITCH = 20;                           // floats per row in LDS (16 + 4 of padding)
    constexpr int NL = (ROWS * 4 + NT - 1) / NT;        // 16-byte pieces per thread and sample
    static_assert((ROWS * 4) % NT == 0, "the copy divides evenly");
    __shared__ __attribute__((aligned(16))) float lbuf[2 * ROWS * PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int f = blockIdx.x, split = blockIdx.y;
    const int wm = wave / CTN, wn = wave % CTN;
    const int n0 = (int)((long long)p.N * split / p.nsplit), n1 = (int)((long long)p.N * (split + 1) / p.nsplit);
    const size_t ds_sample = (size_t)2 * p.NF * CO * 16, s_sample = (size_t)2 * p.NF * CI * 16;

    // copy: piece q = tid + j * NT -> row q / 4, quarter q % 4; the source group of a row is uniform per (j, wave) except
    // at group boundaries, so the offsets are computed per thread once
    size_t goff[NL];                                    // float offset within the sample of ds (rows < 2 CO) or s
    int loff[NL];
    bool from_s[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int q = tid + j * NT, row = q >> 2, quarter = q & 3;
        loff[j] = row * PITCH + quarter * 4;
        from_s[j] = row >= 2 * CO;
        if (!from_s[j]) {
            const int im = row >= CO, r = row - im * CO;
            goff[j] = ((size_t)(im * p.NF + f) * CO + r) * 16 + quarter * 4;
        } else {
            const int r2 = row - 2 * CO, im = r2 >= CI, r = r2 - im * CI;
            goff[j] = ((size_t)(im * p.NF + f) * CI + r) * 16 + quarter * 4;
        }
    }
    f32x4 sta[NL], stb[NL];                             // two samples in flight: the copy runs two samples ahead of the MFMAs
    auto gload = [&](int n, f32x4 (&st)[NL]) {
#pragma unroll
        for (int j = 0; j < NL; ++j)
            st[j] = *reinterpret_cast<const f32x4*>((from_s[j] ? p.s + (size_t)n * s_sample : p.ds + (size_t)n * ds_sample) + goff[j]);
    };
    auto lstore = [&](int b, const f32x4 (&st)[NL]) {
#pragma unroll
        for (int j = 0; j < NL; ++j) *reinterpret_cast<f32x4*>(&lbuf[b * ROWS * PITCH + loff[j]]) = st[j];
    };
    // fragments: row = tile * 32 + l31, joints 8 half .. 8 half + 7 as two float4
    const int fa = (wm * 32 + l31) * PITCH + 8 * half;                 // du^ re; im: + CO rows
    const int fb = (2 * CO + wn * 32 + l31) * PITCH + 8 * half;        // a^ re; im: + CI rows

    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // sample n: multiply from LDS buffer b while sample n + 2 is requested into `nxt2`; then `nxt1` (sample n + 1, requested
    // an iteration ago) goes to the other buffer; the two register sets swap roles from sample to sample
    auto do_sample = [&](int n, int b, const f32x4 (&nxt1)[NL], f32x4 (&nxt2)[NL]) {
        if (n + 2 < n1) gload(n + 2, nxt2);
        const float* L = lbuf + b * ROWS * PITCH;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 dr = *reinterpret_cast<const f32x4*>(L + fa + 4 * h);
            const f32x4 di = *reinterpret_cast<const f32x4*>(L + fa + CO * PITCH + 4 * h);
            const f32x4 ar = *reinterpret_cast<const f32x4*>(L + fb + 4 * h);
            const f32x4 ai = *reinterpret_cast<const f32x4*>(L + fb + CI * PITCH + 4 * h);
            const f32x4 dm = dr - di, bd = ai - ar, bs = ar + ai;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dm[e], ar[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dr[e], bd[e], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(di[e], bs[e], acc[2], 0, 0, 0);
            }
        }
        if (n + 1 < n1) lstore(b ^ 1, nxt1);
        __syncthreads();
    };
    if (n0 < n1) {
        gload(n0, sta);
        if (n0 + 1 < n1) gload(n0 + 1, stb);
        lstore(0, sta);
    }
    __syncthreads();
    for (int n = n0; n < n1; n += 2) {
        do_sample(n, 0, stb, sta);                      // stb holds n + 1; sta takes n + 2
        if (n + 1 < n1) do_sample(n + 1, 1, sta, stb);
    }
    float* dst = p.part + ((size_t)f * p.nsplit + split) * 3 * CO * CI;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wm * 32 + rowmap(r, half), ci = wn * 32 + l31;
            dst[((size_t)i * CO + co) * CI + ci] = acc[i][r];
        }
}

// C^[f][co][ci] = kap_f / L * ordered sum over the batch slices of (P1 + P3, P1 + P2); then
// dw[co][ci][j] = SUM_f (C^r[f] tc[f][j] - C^i[f] ts[f][j]);  db[co] = SUM_(n,v) du^[n][0][co][v] in a fixed order.
// tc / ts: [NF][Kt] cos / sin of 2 pi f (j - P) / L.
__global__ __launch_bounds__(256) void dft_cfold_kernel(const float* __restrict__ part, float* __restrict__ chat, int Ci,
                                                        int Co, int NF, int nsplit, int L) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= NF * Co * Ci) return;
    const int ci = e % Ci, co = (e / Ci) % Co, f = e / (Ci * Co);
    const size_t plane = (size_t)Co * Ci, o = (size_t)co * Ci + ci;
    float cr = 0.f, cim = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* P = part + ((size_t)f * nsplit + s) * 3 * plane;
        const float p1 = P[o];
        cr += p1 + P[2 * plane + o];
        cim += p1 + P[plane + o];
    }
    const float kap = (f == 0 ? 1.f : 2.f) / (float)L;
    chat[(size_t)(2 * f) * Co * Ci + co * Ci + ci] = cr * kap;
    chat[(size_t)(2 * f + 1) * Co * Ci + co * Ci + ci] = cim * kap;
}

constexpr int TAPS_PER_THREAD = 8;           // blockIdx.y = group of 8 taps: ~10x the threads of one-thread-per-(co, ci)
__global__ __launch_bounds__(64) void dft_taps_kernel(const float* __restrict__ chat, const float* __restrict__ tc,
                                                      const float* __restrict__ ts, float* __restrict__ dw, int Ci, int Co,
                                                      int NF, int Kt, int transposed) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= Co * Ci) return;
    const int co = e / Ci, ci = e - co * Ci;
    const int j0 = blockIdx.y * TAPS_PER_THREAD;
    float acc[TAPS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < TAPS_PER_THREAD; ++j) acc[j] = 0.f;
    for (int f = 0; f < NF; ++f) {
        const float cr = chat[(size_t)(2 * f) * Co * Ci + e], cim = chat[(size_t)(2 * f + 1) * Co * Ci + e];
        const float* c = tc + (size_t)f * Kt + j0;
        const float* s = ts + (size_t)f * Kt + j0;
#pragma unroll
        for (int j = 0; j < TAPS_PER_THREAD; ++j)
            if (j0 + j < Kt) acc[j] += cr * c[j] - cim * s[j];
    }
#pragma unroll
    for (int j = 0; j < TAPS_PER_THREAD; ++j)
        if (j0 + j < Kt) {
            if (!transposed) dw[((size_t)co * Ci + ci) * Kt + j0 + j] = acc[j];
            else dw[((size_t)ci * Co + co) * Kt + (Kt - 1 - j0 - j)] = acc[j];
        }
}

__global__ __launch_bounds__(256) void dft_db_kernel(const float* __restrict__ ds, float* __restrict__ db, int N, int Co,
                                                     int NF) {
    __shared__ double red[256];
    const int co = blockIdx.x;
    const size_t sample = (size_t)2 * NF * Co * 16;
    double s = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* r = ds + (size_t)n * sample + (size_t)co * 16;          // row f2 = 0: SUM_t du
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) q += r[v];
        s += (double)q;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[co] = (float)red[0];
}

// w^[f] from the taps, as the triples (a + b, a, b, 0) the mixing kernel takes, W[row][k] = a + i b:
//   forward  wf[f][co / 32][ci / 2][ci % 2][co % 32][4]: a = Wr, b = Wi
//   backward wb[f][ci / 32][co / 2][co % 2][ci % 32][4]: a = Wr, b = -Wi          (conj(w^)^T)
__global__ __launch_bounds__(256) void dft_wspec_kernel(const float* __restrict__ w, const float* __restrict__ tc,
                                                        const float* __restrict__ ts, float* __restrict__ wf,
                                                        float* __restrict__ wb, int Ci, int Co, int NF, int Kt,
                                                        int transposed) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= NF * Co * Ci) return;
    const int ci = e % Ci, co = (e / Ci) % Co, f = e / (Ci * Co);
    float wr = 0.f, wi = 0.f;
    for (int j = 0; j < Kt; ++j) {
        const float x = transposed ? w[((size_t)ci * Co + co) * Kt + (Kt - 1 - j)] : w[((size_t)co * Ci + ci) * Kt + j];
        wr += x * tc[(size_t)f * Kt + j];
        wi += x * ts[(size_t)f * Kt + j];
    }
    f32x4* F = reinterpret_cast<f32x4*>(wf) + ((((size_t)f * (Co / 32) + co / 32) * (Ci / 2) + ci / 2) * 2 + (ci & 1)) * 32 + (co & 31);
    f32x4* B = reinterpret_cast<f32x4*>(wb) + ((((size_t)f * (Ci / 32) + ci / 32) * (Co / 2) + co / 2) * 2 + (co & 1)) * 32 + (ci & 31);
    f32x4 tf, tb;
    tf[0] = wr + wi; tf[1] = wr; tf[2] = wi; tf[3] = 0.f;
    tb[0] = wr - wi; tb[1] = wr; tb[2] = -wi; tb[3] = 0.f;
    *F = tf;
    *B = tb;
}

int pick_mg(int mtiles, bool inverse) {
    // largest group that divides the tile count, else the one with the least padding; both directions keep two
    // accumulator sets (cos | sin rows, E | O), so groups stop at 5 / 4 tiles (160 / 128 accumulator registers)
    if (mtiles <= 1) return 2;
    const int cand_f[] = {5, 4, 3, 2}, cand_i[] = {4, 3, 2};
    const int* cand = inverse ? cand_i : cand_f;
    const int nc = inverse ? 3 : 4;
    for (int i = 0; i < nc; ++i)
        if (mtiles % cand[i] == 0) return cand[i];
    int best = cand[nc - 1], waste = 1 << 30;
    for (int i = 0; i < nc; ++i) {
        const int w = step::ceil_div(mtiles, cand[i]) * cand[i] - mtiles;
        if (w < waste) { waste = w; best = cand[i]; }
    }
    return best;
}

template <bool INV>
int launch_gemm(DgParams p, hipStream_t st, const char* what) {
    // INV == 0: Mpad = round_up(NF, 32) rows per half, two halves; INV == 1: Mpad = round_up(ceil(T_out / 2), 32) pairs
    const int mtiles = p.Mpad / 32;
    const int mg = pick_mg(mtiles, INV);
    const int groups = step::ceil_div(mtiles, mg);
    const dim3 grid((unsigned)groups, (unsigned)(p.C * p.fs / 8), (unsigned)(p.N * p.S));
#define DG_CASE(MGV) case MGV:                                                                                              \
        if (p.fs == 2) hipLaunchKernelGGL((dft_gemm_kernel<MGV, INV, 2>), grid, dim3(256), 0, st, p);                       \
        else hipLaunchKernelGGL((dft_gemm_kernel<MGV, INV, 1>), grid, dim3(256), 0, st, p);                                 \
        return step::check_launch(what)
    if constexpr (INV) {
        switch (mg) { DG_CASE(2); DG_CASE(3); DG_CASE(4); default: break; }
    } else {
        switch (mg) { DG_CASE(2); DG_CASE(3); DG_CASE(4); DG_CASE(5); default: break; }
    }
#undef DG_CASE
    step::set_error("%s: no kernel for %d row tiles", what, mtiles);
    return STEP_ERR_ARG;
}

}  // namespace

// rows of a spectrum for a tensor of T frames and a Kt-tap kernel: 2 NF = round_up(T + (Kt-1)/2 + 1, 4) (NF even: the
// inverse reads the rows in pairs); L = 2 NF - 1 >= T + (Kt-1)/2
extern "C" int step_dft_rows(int T, int Kt) {
    if (T < 1 || Kt < 1 || (Kt & 1) == 0) return 0;
    static const int gran = getenv("STEP_DFT_ROWS32") ? 32 : 4;       // (same-box A/B against the coarser grid of the first version)
    return step::round_up(T + (Kt - 1) / 2 + 1, gran);
}

// layout of the matrices step_dft_fwd / step_dft_inv read (one layout for both kernel forms).  `rows` = the matrix's output rows: NF for each half (cos,
// sin) of the forward matrix, ceil(T_out / 2) pairs for the inverse.  A matrix of `rows` rows is tiled
// [groups][Kpad][mgp x 32] floats: tile slot i of group g is M[row (g * mg + i) * 32 + l][k], zero where the row or k does
// not exist (Kpad = round_up(ceil(K / 2), 16) forward; inverse: 2 round_up(NF, 16), the Re rows and the Im rows each padded to whole chunks of 16); in the forward table slots i < mg hold the cos
// half's tiles and slots mg + i the sin half's.  Per k the slots are stored in fours, [q][l < 32][4], then (mgp % 4 == 2)
// a last pair [l][2]: every LDS read is 16 (8) bytes per lane at a lane stride of 16 (8) bytes.  Returns mg, *mgp = the
// slot count per k (3 tiles are padded to 4), *groups
extern "C" int step_dft_table_geom(int rows, int inverse, int* mgp, int* groups) {
    const int mtiles = step::ceil_div(rows, 32);
    if (mtiles < 1) return 0;
    const int mg = pick_mg(mtiles, inverse != 0);
    if (mgp) *mgp = table_slots(inverse ? mg : 2 * mg);
    if (groups) *groups = step::ceil_div(mtiles, mg);
    return mg;
}

// row-tile groups a launch of step_dft_inv uses for T_out output frames (statistics partial rows per sample)
extern "C" int step_dft_groups(int T_out) {
    const int mtiles = step::ceil_div((T_out + 1) / 2, 32);
    if (mtiles < 1) return 0;
    return step::ceil_div(mtiles, pick_mg(mtiles, true));
}

// Long sequences at stride 1 (overlap-save): the row of T frames is cut into S segments of seg_step output frames;
// segment s transforms the K = seg_step + Kt - 1 input frames [s * seg_step - P, s * seg_step + seg_step + P) (frames
// outside [0, T) are the conv's zero padding), and its circular product needs no more than L >= K bins because the
// outputs kept never wrap - the transforms cost S * K^2 instead of T^2.  The spectra are [N * S][rows][C][16];
// step_dft_mix / step_dft_wspec take them as N * S samples (tap tables with the segment's origin shift, dft.py).
extern "C" int step_dft_fwd_seg(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                                const float* matt, float* spec, int N, int C, int T, int K, int rows, int S, int seg_step,
                                int frame0, void* stream) {
    STEP_REQUIRE(in && matt && spec, "step_dft_fwd: null pointer");
    STEP_REQUIRE(N >= 1 && S >= 1 && (long long)N * S <= 65535 && C >= 8 && C % 8 == 0 && C / 8 <= 65535 && K >= 1 && T >= 1 &&
                 rows >= 4 && rows % 4 == 0 && seg_step >= 0,
                 "step_dft_fwd: bad shape N=%d C=%d T=%d K=%d rows=%d S=%d", N, C, T, K, rows, S);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "step_dft_fwd: scale/shift pair");
    STEP_REQUIRE(((size_t)C * T + K + (size_t)S * seg_step + 64) * 64 < (1ull << 31) && (size_t)rows * C * 64 < (1ull << 31),
                 "step_dft_fwd: sample too large");
    DgParams p{};
    p.in = in; p.matt = matt; p.in_scale = in_scale; p.in_shift = in_shift; p.in_relu = in_relu; p.out = spec;
    p.N = N; p.C = C; p.K = K; p.Kpad = step::round_up((K + 1) / 2, KC); p.Mpad = step::round_up(rows / 2, 32); p.Mrows = rows;
    p.S = S; p.seg_step = seg_step; p.frame0 = frame0; p.Tfull = T; p.fs = 1;
    return launch_gemm<false>(p, (hipStream_t)stream, "step_dft_fwd");
}

// Phases: in (N, C, phases * Tp, 16) holds `phases` interleaved sequences of Tp frames per channel row (frame t of phase
// ph at phases * t + ph); spec [N][rows][phases * C][16] takes the spectrum of sequence (c, ph) as channel phases * c + ph.
extern "C" int step_dft_fwd_ph(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                               const float* matt, float* spec, int N, int C, int Tp, int rows, int phases, void* stream) {
    STEP_REQUIRE(in && matt && spec, "step_dft_fwd_ph: null pointer");
    STEP_REQUIRE(N >= 1 && N <= 65535 && C >= 8 && C % 8 == 0 && (phases == 1 || phases == 2) && C * phases / 8 <= 65535 &&
                 Tp >= 1 && rows >= 4 && rows % 4 == 0,
                 "step_dft_fwd_ph: bad shape N=%d C=%d Tp=%d rows=%d phases=%d", N, C, Tp, rows, phases);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "step_dft_fwd_ph: scale/shift pair");
    STEP_REQUIRE(((size_t)C * Tp + Tp + 64) * phases * 64 < (1ull << 31) && (size_t)rows * C * phases * 64 < (1ull << 31),
                 "step_dft_fwd_ph: sample too large");
    DgParams p{};
    p.in = in; p.matt = matt; p.in_scale = in_scale; p.in_shift = in_shift; p.in_relu = in_relu; p.out = spec;
    p.N = N; p.C = C; p.K = Tp; p.Kpad = step::round_up((Tp + 1) / 2, KC); p.Mpad = step::round_up(rows / 2, 32); p.Mrows = rows;
    p.S = 1; p.seg_step = 0; p.frame0 = 0; p.Tfull = Tp; p.fs = phases;
    return launch_gemm<false>(p, (hipStream_t)stream, "step_dft_fwd_ph");
}

// The inverse of step_dft_fwd_ph: spec [N][rows_in][phases * C][16] -> out (N, C, phases * Tp, 16), channel phases * c + ph of
// the spectrum back to the frames phases * t + ph of channel c (no bias, statistics or tail: the input gradient of a strided conv).
extern "C" int step_dft_inv_ph(const float* spec, const float* matt, float* out, int N, int C, int rows_in, int Tp,
                               int phases, void* stream) {
    STEP_REQUIRE(spec && matt && out, "step_dft_inv_ph: null pointer");
    STEP_REQUIRE(N >= 1 && N <= 65535 && C >= 8 && C % 8 == 0 && (phases == 1 || phases == 2) && C * phases / 8 <= 65535 &&
                 Tp >= 1 && rows_in >= 4 && rows_in % 4 == 0,
                 "step_dft_inv_ph: bad shape N=%d C=%d rows=%d Tp=%d phases=%d", N, C, rows_in, Tp, phases);
    STEP_REQUIRE(((size_t)C * Tp + Tp + 64) * phases * 64 < (1ull << 31) && (size_t)rows_in * C * phases * 64 < (1ull << 31),
                 "step_dft_inv_ph: sample too large");
    DgParams p{};
    p.in = spec; p.matt = matt; p.out = out;
    p.N = N; p.C = C; p.K = rows_in; p.Kpad = 2 * step::round_up(rows_in / 2, KC); p.Mpad = step::round_up((Tp + 1) / 2, 32); p.Mrows = Tp;
    p.S = 1; p.seg_step = 0; p.frame0 = 0; p.Tfull = Tp; p.fs = phases;
    return launch_gemm<true>(p, (hipStream_t)stream, "step_dft_inv_ph");
}

extern "C" int step_dft_fwd(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                            const float* matt, float* spec, int N, int C, int K, int rows, void* stream) {
    return step_dft_fwd_seg(in, in_scale, in_shift, in_relu, matt, spec, N, C, K, K, rows, 1, 0, 0, stream);
}

extern "C" int step_dft_inv_seg(const float* spec, const float* matt, const float* bias, const float* out_scale,
                                const float* out_shift, const float* res, const float* res_scale, const float* res_shift,
                                int out_relu, float* out, int N, int C, int rows_in, int T_seg, int T_out, int S,
                                int seg_step, double* sums, void* ws, size_t ws_bytes, void* stream) {
    STEP_REQUIRE(spec && matt && out, "step_dft_inv: null pointer");
    STEP_REQUIRE(N >= 1 && S >= 1 && (long long)N * S <= 65535 && C >= 8 && C % 8 == 0 && C / 8 <= 65535 && T_out >= 1 &&
                 T_seg >= 1 && rows_in >= 4 && rows_in % 4 == 0 && seg_step >= 0,
                 "step_dft_inv: bad shape N=%d C=%d rows=%d T=%d (%d x %d)", N, C, rows_in, T_out, S, T_seg);
    STEP_REQUIRE(((size_t)C * T_out + T_seg + (size_t)S * seg_step + 64) * 64 < (1ull << 31) && (size_t)rows_in * C * 64 < (1ull << 31),
                 "step_dft_inv: sample too large");
    DgParams p{};
    STEP_REQUIRE((out_scale == nullptr) == (out_shift == nullptr) && (res_scale == nullptr) == (res_shift == nullptr),
                 "step_dft_inv: scale/shift must come in pairs");
    STEP_REQUIRE(!sums || (!out_scale && !res && !out_relu), "step_dft_inv: statistics are of the plain conv output");
    p.in = spec; p.matt = matt; p.bias = bias; p.out = out;
    p.out_scale = out_scale; p.out_shift = out_shift; p.res = res; p.res_scale = res_scale; p.res_shift = res_shift;
    p.out_relu = out_relu;
    p.N = N; p.C = C; p.K = rows_in; p.Kpad = 2 * step::round_up(rows_in / 2, KC); p.Mpad = step::round_up((T_seg + 1) / 2, 32); p.Mrows = T_seg;
    p.S = S; p.seg_step = seg_step; p.frame0 = 0; p.Tfull = T_out; p.fs = 1;
    const int groups = step_dft_groups(T_seg);
    if (sums) {
        const size_t need = (size_t)N * S * groups * C * 2 * sizeof(double);
        if (!ws || ws_bytes < need) {
            step::set_error("step_dft_inv: workspace %zu < %zu bytes", ws_bytes, need);
            return STEP_ERR_WORKSPACE;
        }
        p.stat_part = (double*)ws;
    }
    const int rc = launch_gemm<true>(p, (hipStream_t)stream, "step_dft_inv");
    if (rc != STEP_OK || !sums) return rc;
    return step::bn_sum_partials((const double*)ws, sums, C, N * S * groups, (hipStream_t)stream);
}

extern "C" int step_dft_inv(const float* spec, const float* matt, const float* bias, const float* out_scale,
                            const float* out_shift, const float* res, const float* res_scale, const float* res_shift,
                            int out_relu, float* out, int N, int C, int rows_in, int T_out, double* sums, void* ws,
                            size_t ws_bytes, void* stream) {
    return step_dft_inv_seg(spec, matt, bias, out_scale, out_shift, res, res_scale, res_shift, out_relu, out, N, C, rows_in,
                            T_out, T_out, 1, 0, sums, ws, ws_bytes, stream);
}

extern "C" size_t step_dft_inv_ws_bytes(int N, int C, int T_out) {
    if (N < 1 || C < 1 || T_out < 1) return 0;
    return (size_t)N * step_dft_groups(T_out) * C * 2 * sizeof(double);
}

extern "C" int step_dft_wspec(const float* w, const float* tc, const float* ts, float* wf, float* wb, int Ci, int Co,
                              int NF, int Kt, int transposed, void* stream) {
    STEP_REQUIRE(w && tc && ts && wf && wb, "step_dft_wspec: null pointer");
    STEP_REQUIRE(Ci >= 1 && Co >= 1 && NF >= 1 && Kt >= 1, "step_dft_wspec: bad shape");
    const int total = NF * Co * Ci;
    hipLaunchKernelGGL(dft_wspec_kernel, dim3(step::ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, tc, ts,
                       wf, wb, Ci, Co, NF, Kt, transposed);
    return step::check_launch("step_dft_wspec");
}

extern "C" int step_dft_mix(const float* in, const float* wmat, float* out, int N, int Ci, int Co, int NF, void* stream) {
    STEP_REQUIRE(in && wmat && out, "step_dft_mix: null pointer");
    const bool square = (Ci == 32 || Ci == 64) && (Co == 32 || Co == 64);
    const bool rect = (Ci == 128 && Co == 64) || (Ci == 64 && Co == 128);          // the two phases of a stride-2 conv
    STEP_REQUIRE(N >= 1 && (square || rect) && NF >= 1, "step_dft_mix: unsupported N=%d Ci=%d Co=%d NF=%d", N, Ci, Co, NF);
    MxParams p{in, wmat, out, N, Ci, Co, NF};
    int ny = step::ceil_div(rect ? 2048 : 4096, NF);     // ~8 rounds of 2 blocks (rect: 1 block of 8 waves) per CU: no tail to speak of
    const int npairs = (N + 1) / 2;
    ny = ny > step::ceil_div(npairs, 4) ? step::ceil_div(npairs, 4) : ny;
    ny = ny < 1 ? 1 : ny;
    const dim3 grid((unsigned)NF, (unsigned)ny);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)Co * Ci * 16;                     // (Co / 32) * Ci * 128 floats: 64 KB at 64 x 64
    hipError_t e = hipSuccess;
#define MX_CASE(CTV, CIV)                                                                                              \
    do {                                                                                                               \
        auto kern = dft_mix_kernel<CTV, CIV>;                                                                          \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL(kern, grid, dim3(64 * mix_waves(CTV, CIV)), lds, st, p);               \
    } while (0)
    if (Co == 64 && Ci == 128) MX_CASE(2, 128);
    else if (Co == 128) MX_CASE(4, 64);
    else if (Co == 64 && Ci == 64) MX_CASE(2, 64);
    else if (Co == 64) MX_CASE(2, 32);
    else if (Ci == 64) MX_CASE(1, 64);
    else MX_CASE(1, 32);
#undef MX_CASE
    if (e != hipSuccess) {
        step::set_error("step_dft_mix: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
        return STEP_ERR_LAUNCH;
    }
    return step::check_launch("step_dft_mix");
}

static int wgrad_nsplit(int N, int NF, int C, int Ci) {
    // 64 channels: the block slots of the chip in one round (4 blocks of 4 waves per CU; Ci = 2 Co: 2 blocks of 8 waves); 32: one wave per block
    int ns = C == 32 ? step::ceil_div(3072, NF) : (Ci == 2 * C ? 512 : 1024) / NF;     // 64 channels: the 1024 block slots of the chip (4 per CU) in one round; 32: one wave per block
    ns = ns > N ? N : ns;
    return ns < 1 ? 1 : ns;
}
extern "C" size_t step_dft_wgrad_ws_bytes(int N, int Ci, int Co, int NF) {
    if (N < 1 || Ci < 1 || Co < 1 || NF < 1) return 0;
    return ((size_t)NF * wgrad_nsplit(N, NF, Co, Ci) * 3 * Ci * Co + (size_t)NF * 2 * Ci * Co) * sizeof(float);
}

extern "C" int step_dft_wgrad(const float* dspec, const float* spec, const float* tc, const float* ts, float* dw,
                              float* db, int N, int Ci, int Co, int NF, int Kt, int transposed, void* ws,
                              size_t ws_bytes, void* stream) {
    STEP_REQUIRE(dspec && spec && tc && ts && dw && ws, "step_dft_wgrad: null pointer");
    STEP_REQUIRE(N >= 1 && (Co == 32 || Co == 64) && (Ci == Co || Ci == 2 * Co) && NF >= 1 && Kt >= 1,
                 "step_dft_wgrad: unsupported N=%d Ci=%d Co=%d NF=%d Kt=%d", N, Ci, Co, NF, Kt);
    if (ws_bytes < step_dft_wgrad_ws_bytes(N, Ci, Co, NF)) {
        step::set_error("step_dft_wgrad: workspace %zu < %zu bytes", ws_bytes, step_dft_wgrad_ws_bytes(N, Ci, Co, NF));
        return STEP_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    WgParams p{dspec, spec, (float*)ws, N, Ci, Co, NF, wgrad_nsplit(N, NF, Co, Ci)};
    const dim3 grid((unsigned)NF, (unsigned)p.nsplit);
    static const bool direct = getenv("STEP_DFT_WGRAD_DIRECT") != nullptr;      // (same-box A/B: every wave loads its own operands)
    if (Co == 64 && Ci == 128) {
        if (direct) hipLaunchKernelGGL((dft_wgrad_kernel<2, 4>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((dft_wgrad_lds_kernel<2, 4>), grid, dim3(512), 0, st, p);
    } else if (Co == 64) {
        if (direct) hipLaunchKernelGGL((dft_wgrad_kernel<2, 2>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((dft_wgrad_lds_kernel<2, 2>), grid, dim3(256), 0, st, p);
    } else if (Ci == 64) {
        if (direct) hipLaunchKernelGGL((dft_wgrad_kernel<1, 2>), grid, dim3(128), 0, st, p);
        else hipLaunchKernelGGL((dft_wgrad_lds_kernel<1, 2>), grid, dim3(128), 0, st, p);
    } else {
        hipLaunchKernelGGL((dft_wgrad_kernel<1, 1>), grid, dim3(64), 0, st, p);
    }
    int rc = step::check_launch("step_dft_wgrad");
    if (rc != STEP_OK) return rc;
    float* chat = (float*)ws + (size_t)NF * p.nsplit * 3 * Ci * Co;
    hipLaunchKernelGGL(dft_cfold_kernel, dim3(step::ceil_div(NF * Co * Ci, 256)), dim3(256), 0, st, (const float*)ws, chat,
                       Ci, Co, NF, p.nsplit, 2 * NF - 1);
    rc = step::check_launch("step_dft_wgrad(fold)");
    if (rc != STEP_OK) return rc;
    hipLaunchKernelGGL(dft_taps_kernel, dim3(step::ceil_div(Co * Ci, 64), step::ceil_div(Kt, TAPS_PER_THREAD)), dim3(64), 0, st, (const float*)chat, tc, ts, dw,
                       Ci, Co, NF, Kt, transposed);
    rc = step::check_launch("step_dft_wgrad(taps)");
    if (rc != STEP_OK || !db) return rc;
    hipLaunchKernelGGL(dft_db_kernel, dim3((unsigned)Co), dim3(256), 0, st, dspec, db, N, Co, NF);
    return step::check_launch("step_dft_wgrad(db)");
}
