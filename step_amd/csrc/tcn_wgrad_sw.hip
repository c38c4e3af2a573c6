// Weight gradient of the temporal (Kt x 1) convolution, fp32 MFMA, sliding-window form.
//
// Same operator and results as tcn_wgrad.hip (weight part of convolution_backward for tcn[2],
// classifier_stgcn_real_only/net/classifier.py:156-162), arranged like tcn_wgrad_bf16.hip:
//   G[co][ci][j] = SUM_{n,t,v} dout[n,co,t,v] * a[n,ci,s*t + j - pad, v],  a = relu(bn1(y))
// GEMM view: M = co, N = (tap j, ci) with the 32 lanes of a column tile on 32 INPUT CHANNELS,
// K = (n, t, v).  v_mfma_f32_32x32x2_f32: k-step s of a frame pairs joints (s, s + 8), so a
// lane's eight A (or B) values of one frame are 32 contiguous bytes of the LDS image
// [channel][frame][16 joints] - two 16-byte reads feed 8 MFMAs per accumulator tile.  The B
// fragment of (frame t, tap j) depends only on s*t + j: an MFMA wave owns NJ consecutive taps and
// slides a window along t, so one fragment read serves MW*NJ*8 = 80 MFMAs (tcn_wgrad.hip needs a
// 4-byte LDS read per MFMA, which is what holds it at 0.73 of the MFMA peak).
// Block = 4 MFMA waves + 4 staging waves whose only work is LDS-DMA: the dout tile (two images) and the raw y
// frames go global -> LDS with no VGPRs, and the MFMA waves apply BN + ReLU to the fragments they read.  The
// activation strip is a ring of frames (see WsCfg): per chunk only the S*TT new frames are fetched, the whole
// strip once per sequence.  Slices of the batch are combined by the ordered reduce of tcn_wgrad.hip
// (bit-reproducible).
// Round 3, what the ablations said (C = 64, stride 2, N = 4096; 36.7 ms = 0.837 of the fp32 MFMA peak before): feeding
// the MFMAs pre-activated fragments (no BN + ReLU VALU in the MFMA waves) changes nothing; the bias sums change nothing;
// no slab store, no sequence-start barrier: nothing; no per-chunk barrier: -1.5 %; no DMA at all: -6 %; and 256 extra
// v_fma per chunk in the STAGING waves: +23 %.  An instruction a staging wave issues costs the MFMA wave on its SIMD
// time, and the staging waves spent ~30 VALU instructions per 1-KiB DMA piece on addresses (a division among them).
// With per-lane constant offsets and uniform bases (one v_lshl_add_u64 per piece): 35.1 ms = 0.875 at stride 2,
// 68.5 ms = 0.897 at stride 1 (T = 240), 17.8 ms = 0.86 at 32 channels; with DMA and barriers off the MFMA loop alone
// runs at 0.887 of the peak in this two-launch form (75 of 76 tap slots useful, 252 of 256 CUs busy).
// Also tried: no staging waves, 8 MFMA waves (tap group x M tile, two per SIMD, 80 accumulator registers each) that
// issue the DMA themselves (through inline asm - the compiler waits vmcnt(0) before the first LDS read after an
// LDS-DMA it knows of): correct, but 37.1 ms against 35.0 - every B fragment is then read and activated by two waves.
#include "common.h"
#include <type_traits>
#include <cstdlib>

namespace {

struct WsParams {
    const float* dout;
    const float* in;
    const float* in_scale;
    const float* in_shift;
    float* slab;        // [nsplit][Co][Q]
    float* dbslab;      // [nsplit][Co]
    const float* zeros;  // 64 B of zeros in device memory (source of the frames past the end)
    int in_relu;
    int N, Ci, Co, Tin, Tout, Kt, pad;
    int Q, ntb, nqb, nsplit;
    int ncib;           // input-channel blocks of 32; q-block = (tap block, ci block, co group of 32*MW)
    int j_base, ks;     // this launch covers taps [j_base, j_base + ks); slab rows are [Ci][ks] (Q = Ci * ks)
};

template <int MW, int NJ, int STRIDE, int TT>
struct WsCfg {
    static constexpr int NCONS = 4, NPROD = 4;
    static constexpr int NTHR = 64 * (NCONS + NPROD), PTHR = 64 * NPROD;
    static constexpr int TAPS = NCONS * NJ;                       // taps per block
    static constexpr int AF = STRIDE * (TT - 1) + TAPS;           // activation frames a chunk reads
    static constexpr int NEWF = STRIDE * TT;                      // ... of which only these are new
    // The activation strip is a RING of frames in LDS: chunk c of a sequence reads slots [NEWF*c, NEWF*c + AF)
    // (mod RING) while the NEWF frames of chunk c+1 land in the slots behind them, so RING >= AF + NEWF.  A frame
    // is 2 KiB = [32 channels][4 quarters of 16 B] = two whole 1-KiB DMA pieces; quarter q of channel ch sits at
    // position (q + (ch >> 2)) & 3 of its 64-byte row (the swizzle is applied to the per-lane SOURCE address),
    // which puts the 16-byte reads of 16 consecutive channels on 16 distinct bank groups.
    static constexpr int RING = (AF + NEWF <= 32) ? 32 : AF + NEWF;
    static constexpr int FRAME_BYTES = 2048;
    // dout rows: an odd number of 16-byte chunks keeps the 16-byte row reads of 32 channels on distinct bank groups
    static constexpr int DS = TT * 64 + 16;
    static constexpr int D_BYTES = (32 * MW * DS + 1023) / 1024 * 1024;   // whole 1-KiB DMA pieces
    static constexpr int A_BASE = 2 * D_BYTES;                    // two dout images, then the ring
    static constexpr int LDS_BYTES = A_BASE + RING * FRAME_BYTES;
    static constexpr int NBT = STRIDE * (TT - 1) + NJ;            // B fragments of one chunk (per wave)
    static_assert(AF + NEWF <= RING && LDS_BYTES <= 160 * 1024, "LDS");
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

struct frag8 {
    float4 lo, hi;
    __device__ __forceinline__ float at(int s) const {
        switch (s) {
            case 0: return lo.x; case 1: return lo.y; case 2: return lo.z; case 3: return lo.w;
            case 4: return hi.x; case 5: return hi.y; case 6: return hi.z; default: return hi.w;
        }
    }
};

__device__ __forceinline__ bf16x8 to_bf16(const frag8& f) {
    const f32x8 v = {f.lo.x, f.lo.y, f.lo.z, f.lo.w, f.hi.x, f.hi.y, f.hi.z, f.hi.w};
    return __builtin_convertvector(v, bf16x8);
}

// v = hi + lo (+ a residual below 2^-17 |v|): hi = bf16(v), lo = bf16(v - hi); v - hi is exact in fp32
__device__ __forceinline__ void split_bf16(const frag8& f, bf16x8& hi, bf16x8& lo) {
    const f32x8 v = {f.lo.x, f.lo.y, f.lo.z, f.lo.w, f.hi.x, f.hi.y, f.hi.z, f.hi.w};
    hi = __builtin_convertvector(v, bf16x8);
    lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), bf16x8);
}

// logical id of a block among the `count` blocks [first, first + count) of a launch: blocks go to the XCDs round-robin
// (XCD = blockIdx % 8), and the blocks of one XCD get a contiguous range of ids, so that the q-blocks of the same
// batch slices (consecutive ids) stream their operands out of one L2
__device__ __forceinline__ int xcd_contiguous_id(int b, int first, int count) {
    const int x = b & 7;
    int before = 0;                                   // blocks of this range on the XCDs before x
    for (int y = 0; y < x; ++y) {
        const int f = first + ((y - first) & 7);      // first id >= `first` on XCD y
        before += f < first + count ? (first + count - 1 - f) / 8 + 1 : 0;
    }
    const int f = first + ((x - first) & 7);
    return before + (b - f) / 8;
}

// A uniform address the optimiser may not fold into the lanes' offsets: kept as an SGPR pair, so that `base + offset`
// stays (uniform base) + (32-bit lane offset) and the DMA takes its saddr + voffset form with no VALU instruction.
__device__ __forceinline__ const char* sgpr_base(const void* q) {
    unsigned long long v = reinterpret_cast<unsigned long long>(q);
    asm volatile("" : "+s"(v));
    return reinterpret_cast<const char*>(v);
}

template <int MW, int NJ, int STRIDE, int TT, int BF>
__device__ __forceinline__ void ws_body(const WsParams& p, const int lb, char* smem) {
    using C = WsCfg<MW, NJ, STRIDE, TT>;
    constexpr bool BF16 = BF == 1;       // operands rounded to bf16 (BASELINE's bf16 training config)
    constexpr bool X3 = BF == 2;         // fp32 operands split into bf16 hi + lo, three bf16 MFMAs per product

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int qb = lb % p.nqb;
    const int split = lb / p.nqb;
    const int tb = qb % p.ntb;                        // tap block
    const int ci0 = ((qb / p.ntb) % p.ncib) * 32;     // input-channel block
    const int co0 = (qb / (p.ntb * p.ncib)) * 32 * MW;   // output-channel group (wide layers: Co > 64)
    const int j0 = p.j_base + tb * C::TAPS;
    const int n0 = (int)((long long)p.N * split / p.nsplit);
    const int n1 = (int)((long long)p.N * (split + 1) / p.nsplit);
    const int nchunks = (p.Tout + TT - 1) / TT;
    const int total = (n1 - n0) * nchunks;

    if (wave >= C::NCONS) {
        // ------------------------------------------------------------ staging waves
        const int ptid = tid - 64 * C::NCONS;
        const int pwave = wave - C::NCONS;

        // dout tile: a straight copy, so it goes global -> LDS by LDS-DMA (1 KiB per wave
        // instruction, no VGPRs, no ds_write).  16-byte chunk c of the image is (co = c / CPR,
        // slot = c % CPR); slot CPR-1 is the row pad (it copies chunk 0 again), frames past the end
        // of the sequence copy from a zero line.
        constexpr int CPR = C::DS / 16;                          // chunks per channel row
        constexpr int DCH = 32 * MW * CPR;                        // chunks of the image
        constexpr int NPIECE = C::D_BYTES / 1024;
        constexpr int DPW = (NPIECE + C::NPROD - 1) / C::NPROD;   // DMA instructions per wave
        // Address work of these waves is kept off the vector ALU: an instruction a staging wave issues costs the MFMA
        // wave on its SIMD time (measured: 256 extra v_fma per chunk here = +23 % kernel time), and the first version
        // spent ~30 VALU instructions per DMA piece (a division by CPR among them).  The lane's byte offset inside an
        // item is a per-lane CONSTANT (doff), the item's (n, t0) part is uniform and stays in SGPRs: saddr + voffset.
        unsigned doff[DPW];
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int c = (k * C::NPROD + pwave) * 64 + lane;
            const int co = c / CPR, slot = c - co * CPR;
            const int w = slot < CPR - 1 ? slot : 0;
            // rows past Co (never with the channel counts ws_geom admits) and the tail pad of the image re-read the
            // item's first chunk: those bytes are never used
            doff[k] = (c < DCH && co0 + co < p.Co) ? (unsigned)((((co0 + co) * p.Tout + (w >> 2)) * 16 + (w & 3) * 4) * 4) : 0u;
        }
        auto dma_dout = [&](int item, char* img) {
            const int n = n0 + item / nchunks;
            const int t0 = (item % nchunks) * TT;
            const float* dbase = p.dout + (size_t)n * p.Co * p.Tout * 16;
            if (t0 + TT <= p.Tout) {
                const char* ibase = sgpr_base(dbase + (size_t)t0 * 16);
#pragma unroll
                for (int k = 0; k < DPW; ++k) {
                    const int piece = k * C::NPROD + pwave;
                    if (piece < NPIECE)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ibase + doff[k]),
                                                         (__attribute__((address_space(3))) void*)(img + piece * 1024),
                                                         16, 0, 0);
                }
                return;
            }
            // the sequence's last chunk when TT does not divide Tout: frames past the end copy from the zero line
#pragma unroll
            for (int k = 0; k < DPW; ++k) {
                const int piece = k * C::NPROD + pwave;           // 1 KiB piece of the image
                const int c = piece * 64 + lane;
                if (piece < NPIECE) {
                    const int co = c / CPR, slot = c - co * CPR;
                    const int w = slot < CPR - 1 ? slot : 0;
                    const int t = t0 + (w >> 2);
                    const float* src = (c < DCH && co0 + co < p.Co && t < p.Tout)
                                           ? dbase + (((co0 + co) * p.Tout + t) * 16 + (w & 3) * 4)
                                           : p.zeros;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(img + piece * 1024),
                                                     16, 0, 0);
                }
            }
        };
        // exact fp32 bias sums from the landed image (only the block that reports them): TPC
        // threads share an output channel and split its TT*4 chunks
        constexpr int TPC = C::PTHR / (32 * MW);
        float dbacc = 0.f;
        auto sum_dout = [&](const char* img) {
            const int co = ptid / TPC, part = ptid % TPC;
#pragma unroll
            for (int w = 0; w < (TT * 4 + TPC - 1) / TPC; ++w) {
                const int ch = part + w * TPC;
                if (ch < TT * 4) {
                    const float4 v = *reinterpret_cast<const float4*>(img + co * C::DS + ch * 16);
                    dbacc += (v.x + v.y) + (v.z + v.w);
                }
            }
        };

        // activation strip: the raw y frames go global -> LDS by DMA as well, into the ring; tcn[0]/tcn[1] (BN +
        // ReLU) are applied by the MFMA waves to the fragments they read (a lane of a B fragment is one input
        // channel, so scale/shift are per-lane constants there), and frames outside the sequence are zeroed there
        // too.  One piece = 16 channels of one frame: lane -> (channel, stored quarter); it fetches the source
        // quarter that the swizzle maps there.
        char* aring = smem + C::A_BASE;
        const int pch = lane >> 2;                                  // channel within the piece
        // per-lane constant byte offsets of the two half-frames (channels 0..15 | 16..31 of the block's 32; ws_geom
        // admits only channel counts that are multiples of 32, so every channel exists); the frame and sample part of
        // the address is uniform
        unsigned aoff[2];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            const int ch = 16 * hp + pch;
            const int q = ((lane & 3) - (ch >> 2)) & 3;
            aoff[hp] = (unsigned)(((ci0 + ch) * p.Tin * 16 + q * 4) * 4);
        }
        auto dma_frames = [&](auto count_tag, int n, int frame_first, int slot_first) {
            constexpr int COUNT = decltype(count_tag)::value;
            constexpr int PER = (2 * COUNT + C::NPROD - 1) / C::NPROD;
            const float* nbase = p.in + (size_t)n * p.Ci * p.Tin * 16;
#pragma unroll
            for (int k0 = 0; k0 < PER; ++k0) {
                const int k = k0 * C::NPROD + pwave;
                if (k < 2 * COUNT) {
                    const int fi = k >> 1, hp = k & 1;
                    int slot = slot_first + fi;
                    slot = slot >= C::RING ? slot - C::RING : slot;
                    // Frames outside the sequence: the MFMA waves multiply what lands by scale = shift = 0, so it only
                    // has to be finite - the nearest frame of the sequence is fetched instead (scalar min / max, no
                    // branch, no per-lane select; a non-finite activation anywhere makes dw non-finite either way).
                    int frame = frame_first + fi;
                    frame = frame < 0 ? 0 : frame;
                    frame = frame < p.Tin ? frame : p.Tin - 1;
                    const char* src = sgpr_base(nbase + (size_t)frame * 16);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + aoff[hp]),
                                                     (__attribute__((address_space(3))) void*)(aring + slot * C::FRAME_BYTES + hp * 1024),
                                                     16, 0, 0);
                }
            }
        };

        const bool want_db = (qb % (p.ntb * p.ncib) == 0) && p.dbslab;     // one q-block per output group
        auto dbuf = [&](int item) { return smem + (item & 1) * C::D_BYTES; };
        for (int it = 0; it < total; ++it) {
            const int n = n0 + it / nchunks, c = it % nchunks;
            if (c == 0) {
                // sequence start: the whole strip, the ring restarts at slot 0 (the MFMA waves wait at the same
                // barrier; once per sequence)
                dma_frames(std::integral_constant<int, C::AF>{}, n, j0 - p.pad, 0);
                if (it == 0) dma_dout(0, dbuf(0));
                __syncthreads();                   // (the compiler drains vmcnt before every barrier here)
            }
            // the MFMA waves work on chunk `it`; the dout tile of chunk it+1 goes to the other image and, within a
            // sequence, its NEWF new frames to the ring slots behind the ones being read
            if (it + 1 < total) {
                dma_dout(it + 1, dbuf(it + 1));
                if (c + 1 < nchunks)
                    dma_frames(std::integral_constant<int, C::NEWF>{}, n, STRIDE * TT * c + j0 - p.pad + C::AF,
                               (C::NEWF * c + C::AF) % C::RING);
            }
            if (want_db) sum_dout(dbuf(it));
            __syncthreads();
        }
        if (want_db) {
            float sdb = dbacc;
#pragma unroll
            for (int o = 1; o < TPC; o <<= 1) sdb += __shfl_xor(sdb, o, 64);
            const int co = ptid / TPC;
            if (ptid % TPC == 0 && co0 + co < p.Co) p.dbslab[(size_t)split * p.Co + co0 + co] = sdb;
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

    const int aoff = l31 * C::DS + half * 32;                                   // + mt*32*DS + tl*64
    // ring reads: this lane's channel row of a frame slot, the two quarters 2*half and 2*half + 1 where the
    // swizzle put them
    const int boffa = l31 * 64 + (((2 * half) + (l31 >> 2)) & 3) * 16;
    const int boffb = l31 * 64 + (((2 * half + 1) + (l31 >> 2)) & 3) * 16;
    const float relu_lo = p.in_relu ? 0.f : -__builtin_inff();
    float bsc = 1.f, bsh = 0.f;                       // tcn[0] of this lane's input channel
    if (p.in_scale && ci0 + l31 < p.Ci) { bsc = p.in_scale[ci0 + l31]; bsh = p.in_shift[ci0 + l31]; }

    // the window slides by NEWF = S*TT frames per chunk: the last NJ - S fragments a chunk holds (activated, in
    // registers) are the first NJ - S of the next chunk of the same sequence - carried over instead of being read
    // from the ring and activated again before the chunk's first MFMA can issue
    // (bf16 mode only: 6.6 / 5.9 -> 6.4 / 5.7 ms; the fp32 mode schedules worse with it: 36.7 -> 37.3 ms, measured)
    // (split mode: stride 1 only - 11.6 / 7.0 -> 11.3 / 6.7 ms there, 13.0 -> 13.6 ms at stride 2, measured)
    constexpr int NCARRY = ((BF == 1 || (BF == 2 && STRIDE == 1)) && NJ - STRIDE > 0 && NJ <= C::NBT) ? NJ - STRIDE : 0;
    frag8 bcar[NCARRY > 0 ? NCARRY : 1];
    for (int it = 0; it < total; ++it) {
        const int cc = it % nchunks;
        if (cc == 0) __syncthreads();                  // sequence start: the staging waves reload the whole strip
        const char* dimg = smem + (it & 1) * C::D_BYTES;
        const char* aring = smem + C::A_BASE;
        int slot0 = (C::NEWF * cc) % C::RING + wave * NJ;         // ring slot of this wave's fragment 0
        slot0 = slot0 >= C::RING ? slot0 - C::RING : slot0;
        auto read_a = [&](int mt, int tl) -> frag8 {
            const float4* q = reinterpret_cast<const float4*>(dimg + aoff + mt * 32 * C::DS + tl * 64);
            return frag8{q[0], q[1]};
        };
        auto read_b = [&](int qi) -> frag8 {
            int slot = slot0 + qi;
            slot = slot >= C::RING ? slot - C::RING : slot;
            const char* fr = aring + slot * C::FRAME_BYTES;
            return frag8{*reinterpret_cast<const float4*>(fr + boffa), *reinterpret_cast<const float4*>(fr + boffb)};
        };
        // tcn[0] scale/shift + tcn[1] ReLU on a fragment (this lane's channel); frames outside the
        // sequence are zero (the reference pads after the activation)
        const int t0c = (it % nchunks) * TT;
        const int frame0 = STRIDE * t0c + j0 - p.pad + wave * NJ;     // input frame of fragment 0
        auto activate = [&](frag8& f, int qi) {
            // frames outside the sequence: scale = shift = 0 gives an exact zero with one select per fragment
            // instead of one per element (the landed bytes there are the zero line, so 0 * x is 0, not NaN)
            const bool inside = (unsigned)(frame0 + qi) < (unsigned)p.Tin;      // wave-uniform
            float* v = reinterpret_cast<float*>(&f);
            const float sc = inside ? bsc : 0.f, sh = inside ? bsh : 0.f;
            // two VALU instructions per element (fma, max: ReLU is a lower bound of 0, or -inf without it); every
            // VALU instruction of an MFMA wave adds to its MFMA time (fp32 stride-2 wgrad 38.9 -> 37.5 ms against
            // the four of fma / max / select on in_relu / select on `inside`).  Consequence for in_relu == 0 (no model
            // path uses it): v_max(NaN, -inf) is -inf, so a NaN activation reaches dw as +-inf / NaN of the products
            // with -inf instead of as a NaN of its own - dw is non-finite either way
            // (tests/test_gpu_ops.py::test_wgrad_nan_input_without_relu_stays_non_finite).  Frames outside the
            // sequence and channels past Ci always land from the zero line, so 0 * x there is 0.
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * sc + sh, relu_lo);
        };
        frag8 a[TT][MW];
        frag8 b[C::NBT];
#pragma unroll
        for (int m = 0; m < MW; ++m) a[0][m] = read_a(m, 0);
        if (NCARRY > 0 && cc != 0) {
#pragma unroll
            for (int q = 0; q < NCARRY; ++q) b[q] = bcar[q];
#pragma unroll
            for (int q = NCARRY; q < NJ; ++q) { b[q] = read_b(q); }
#pragma unroll
            for (int q = NCARRY; q < NJ; ++q) activate(b[q], q);
        } else {
#pragma unroll
            for (int q = 0; q < NJ && q < C::NBT; ++q) { b[q] = read_b(q); }
#pragma unroll
            for (int q = 0; q < NJ && q < C::NBT; ++q) activate(b[q], q);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tl = 0; tl < TT; ++tl) {
            if (tl + 1 < TT) {
#pragma unroll
                for (int m = 0; m < MW; ++m) a[tl + 1][m] = read_a(m, tl + 1);
            }
#pragma unroll
            for (int s = 0; s < STRIDE; ++s) {
                const int q = STRIDE * (tl + 1) + NJ - STRIDE + s;
                if (q < C::NBT) b[q] = read_b(q);
            }
            if constexpr (X3) {
                // split operands: v = hi + lo with hi = bf16(v), lo = bf16(v - hi) (16 mantissa bits together);
                // a*b ~ a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, the dropped lo*lo term is 2^-16 relative.  Small terms first.
                bf16x8 ah[MW], al[MW], bh[NJ], bl[NJ];
#pragma unroll
                for (int m = 0; m < MW; ++m) split_bf16(a[tl][m], ah[m], al[m]);
#pragma unroll
                for (int i = 0; i < NJ; ++i) split_bf16(b[STRIDE * tl + i], bh[i], bl[i]);
#pragma unroll
                for (int i = 0; i < NJ; ++i)
#pragma unroll
                    for (int m = 0; m < MW; ++m) {
                        acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[i], acc[m][i], 0, 0, 0);
                        acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[i], acc[m][i], 0, 0, 0);
                        acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[i], acc[m][i], 0, 0, 0);
                    }
            } else if constexpr (BF16) {
                // bf16 operands: a lane's eight joints of a frame are one v_mfma_f32_32x32x16_bf16
                // fragment (k = 8*half + e), rounded here, in registers
                bf16x8 ab[MW], bb[NJ];
#pragma unroll
                for (int m = 0; m < MW; ++m) ab[m] = to_bf16(a[tl][m]);
#pragma unroll
                for (int i = 0; i < NJ; ++i) bb[i] = to_bf16(b[STRIDE * tl + i]);
#pragma unroll
                for (int i = 0; i < NJ; ++i)
#pragma unroll
                    for (int m = 0; m < MW; ++m)
                        acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[m], bb[i], acc[m][i], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int i = 0; i < NJ; ++i)
#pragma unroll
                        for (int m = 0; m < MW; ++m)
                            acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tl][m].at(s), b[STRIDE * tl + i].at(s),
                                                                            acc[m][i], 0, 0, 0);
            }
            // the fragments read during this frame are first used by the next one
#pragma unroll
            for (int s = 0; s < STRIDE; ++s) {
                const int q = STRIDE * (tl + 1) + NJ - STRIDE + s;
                if (q < C::NBT) activate(b[q], q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NCARRY > 0) {
#pragma unroll
            for (int q = 0; q < NCARRY; ++q) bcar[q] = b[C::NEWF + q];
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
                    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < p.Co) slab[(size_t)co * p.Q + ci * p.ks + (j - p.j_base)] = acc[m][i][r];
                }
        }
    }
}

template <int MW, int NJ, int STRIDE, int TT, int BF>
__global__ __launch_bounds__(512) void tcn_wgrad_sw_kernel(WsParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ws_body<MW, NJ, STRIDE, TT, BF>(p, xcd_contiguous_id(blockIdx.x, 0, gridDim.x), smem);
}

// The 75 taps are 15 groups of 5 on blocks of 4 MFMA waves: one launch pads them to 16 groups (the last wave of the
// last tap block multiplies zeros, 6 % of all MFMAs).  Where it pays, the taps go as TWO launches instead: taps 0..59 as
// 3 tap blocks of 4 x 5, taps 60..74 as one tap block of 4 x 4 (one padded tap in 76), each with its own number of batch
// slices so that both fill the 256 CUs (a wave with 4 taps does 4/5 of the work per frame).
constexpr int TAPS_A = 60, TAPS_B = 16;

struct WsGeom {
    int mw, taps, ntb, nqb, nsplit;
    bool two;                       // two launches (taps 0..59 | 60..74)
    int nqbA, nsA, nqbB, nsB;
    size_t ws;
};

bool ws_geom(int N, int Ci, int Co, int Kt, int stride, WsGeom* g) {
    // 32 / 64 channels (the reference's widths) or wider in multiples of 64 (output-channel groups of 64)
    if (Kt != 75 || N < 1 || Ci != Co || (Ci != 32 && Ci % 64 != 0) || (stride != 1 && stride != 2)) return false;
    g->mw = Co == 32 ? 1 : 2;
    g->taps = 20;
    g->ntb = step::ceil_div(Kt, g->taps);
    const int per_tb = (Ci / 32) * (Co / (32 * g->mw));
    g->nqb = g->ntb * per_tb;
    int ns = 256 / g->nqb;                     // one block per CU
    ns = ns < 1 ? 1 : ns;
    g->nsplit = ns < N ? ns : N;
    const size_t one = ((size_t)g->nsplit * Co * Ci * Kt + (size_t)g->nsplit * Co + 16) * 4;   // + a zero line
    g->nqbA = 3 * per_tb;
    g->nqbB = per_tb;
    int a = 256 / g->nqbA, b = 256 / g->nqbB;
    a = a < 1 ? 1 : a; b = b < 1 ? 1 : b;
    g->nsA = a < N ? a : N;
    g->nsB = b < N ? b : N;
    const size_t two = ((size_t)g->nsA * Co * Ci * TAPS_A + (size_t)g->nsB * Co * Ci * TAPS_B + (size_t)g->nsA * Co + 16) * 4;
    const double cost_one = 1.0 / g->nsplit, cost_two = 1.0 / g->nsA + 0.8 / g->nsB;
    g->two = cost_two < 0.99 * cost_one;
    g->ws = one > two ? one : two;
    return true;
}

template <int MW, int NJ, int STRIDE, int TT, int BF>
int launch_ws(const WsParams& p, hipStream_t st) {
    using C = WsCfg<MW, NJ, STRIDE, TT>;
    auto kern = tcn_wgrad_sw_kernel<MW, NJ, STRIDE, TT, BF>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
        step::set_error("step_tcn_wgrad: hipFuncSetAttribute(%d): %s", C::LDS_BYTES, hipGetErrorString(e));
        return STEP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(p.nqb * p.nsplit), dim3(C::NTHR), C::LDS_BYTES, st, p);
    return step::check_launch("step_tcn_wgrad(sliding)");
}

template <int NJ>
int dispatch_ws(const WsParams& p, int mw, int stride, int bf16, hipStream_t st) {
    if (bf16 == 2) {
        if (mw == 1) return stride == 1 ? launch_ws<1, NJ, 1, 8, 2>(p, st) : launch_ws<1, NJ, 2, 6, 2>(p, st);
        return stride == 1 ? launch_ws<2, NJ, 1, 6, 2>(p, st) : launch_ws<2, NJ, 2, 6, 2>(p, st);
    }
    if (bf16) {
        if (mw == 1) return stride == 1 ? launch_ws<1, NJ, 1, 8, 1>(p, st) : launch_ws<1, NJ, 2, 6, 1>(p, st);
        return stride == 1 ? launch_ws<2, NJ, 1, 6, 1>(p, st) : launch_ws<2, NJ, 2, 6, 1>(p, st);
    }
    if (mw == 1) return stride == 1 ? launch_ws<1, NJ, 1, 8, 0>(p, st) : launch_ws<1, NJ, 2, 6, 0>(p, st);
    return stride == 1 ? launch_ws<2, NJ, 1, 6, 0>(p, st) : launch_ws<2, NJ, 2, 6, 0>(p, st);
}

// ordered sums of the two launches' slabs ([nsA][Co][Ci][60] and [nsB][Co][Ci][16]) into dw / db; 4 chains per thread
__global__ void ws_reduce2_kernel(const float* __restrict__ slabA, const float* __restrict__ slabB,
                                  const float* __restrict__ dbslab, float* __restrict__ dw, float* __restrict__ db,
                                  int Co, int Ci, int Kt, int nsA, int nsB, int transposed) {
    const int total = Co * Ci * Kt;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < total) {
        const int j = e % Kt, cc = e / Kt;               // cc = co * Ci + ci
        const bool inA = j < TAPS_A;
        const float* src = inA ? slabA + (size_t)cc * TAPS_A + j : slabB + (size_t)cc * TAPS_B + (j - TAPS_A);
        const size_t stride = (size_t)Co * Ci * (inA ? TAPS_A : TAPS_B);
        const int ns = inA ? nsA : nsB;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < ns; k += 4) {
            s0 += src[(size_t)k * stride];
            s1 += src[(size_t)(k + 1) * stride];
            s2 += src[(size_t)(k + 2) * stride];
            s3 += src[(size_t)(k + 3) * stride];
        }
        for (; k < ns; ++k) s0 += src[(size_t)k * stride];
        const float sum = (s0 + s1) + (s2 + s3);
        if (!transposed) {
            dw[e] = sum;
        } else {
            const int co = cc / Ci, ci = cc % Ci;
            dw[((size_t)ci * Co + co) * Kt + (Kt - 1 - j)] = sum;
        }
    } else if (e < total + Co && db) {
        const int co = e - total;
        float sum = 0.f;
        for (int k = 0; k < nsA; ++k) sum += dbslab[(size_t)k * Co + co];
        db[co] = sum;
    }
}

}  // namespace

size_t step::wgrad_sw_ws_bytes(int N, int Ci, int Co, int Kt, int stride) {
    WsGeom g;
    return ws_geom(N, Ci, Co, Kt, stride, &g) ? g.ws : 0;
}

// returns 1 when this variant does not cover the shape (the caller runs tcn_wgrad.hip's kernel)
int step::wgrad_sw(const float* dout, const float* in, const float* in_scale, const float* in_shift,
                   int in_relu, float* dw, float* db, int N, int Ci, int Co, int Tin, int Tout, int Kt,
                   int stride, int transposed, void* ws, size_t ws_bytes, hipStream_t st, int bf16) {
    WsGeom g;
    if (!ws_geom(N, Ci, Co, Kt, stride, &g) || ws_bytes < g.ws) return 1;
    WsParams p{};
    p.dout = dout; p.in = in; p.in_scale = in_scale; p.in_shift = in_shift; p.in_relu = in_relu;
    p.N = N; p.Ci = Ci; p.Co = Co; p.Tin = Tin; p.Tout = Tout; p.Kt = Kt;
    p.pad = (Kt - 1) / 2;
    p.ncib = Ci / 32;
    auto zero_line = [&](float* zl) {
        if (hipMemsetAsync(zl, 0, 64, st) != hipSuccess) {
            step::set_error("step_tcn_wgrad: hipMemsetAsync failed");
            return false;
        }
        p.zeros = zl;
        return true;
    };
    // (fp32 only: 37.7 -> 36.6 ms at C = 64, stride 2; the bf16 mode is not bound by its MFMAs and loses: 6.1 -> 6.7 ms)
    const char* force = getenv("STEP_WS_LAUNCHES");            // "1" | "2": tests and A/B runs
    const bool two = force ? force[0] == '2' : (g.two && bf16 == 0);     // (split operands: 13.0 ms either way)
    if (!two) {
        p.slab = (float*)ws;
        p.dbslab = p.slab + (size_t)g.nsplit * Co * Ci * Kt;
        if (!zero_line(p.dbslab + (size_t)g.nsplit * Co)) return STEP_ERR_LAUNCH;
        p.j_base = 0; p.ks = Kt;
        p.Q = Ci * Kt; p.ntb = g.ntb; p.nqb = g.nqb; p.nsplit = g.nsplit;
        const int rc = dispatch_ws<5>(p, g.mw, stride, bf16, st);
        if (rc) return rc;
        return step::wgrad_reduce(p.slab, p.dbslab, dw, db, Co, Ci, Kt, g.nsplit, transposed, st,
                                  "step_tcn_wgrad(sliding, reduce)");
    }
    float* slabA = (float*)ws;
    float* slabB = slabA + (size_t)g.nsA * Co * Ci * TAPS_A;
    float* dbslab = slabB + (size_t)g.nsB * Co * Ci * TAPS_B;
    if (!zero_line(dbslab + (size_t)g.nsA * Co)) return STEP_ERR_LAUNCH;
    // taps 0..59: 3 tap blocks of 4 waves x 5 taps (these blocks also take the bias sums)
    p.slab = slabA; p.dbslab = dbslab; p.j_base = 0; p.ks = TAPS_A;
    p.Q = Ci * TAPS_A; p.ntb = 3; p.nqb = g.nqbA; p.nsplit = g.nsA;
    // taps 60..74: one tap block of 4 waves x 4 taps
    WsParams pb = p;
    pb.slab = slabB; pb.dbslab = nullptr; pb.j_base = TAPS_A; pb.ks = TAPS_B;
    pb.Q = Ci * TAPS_B; pb.ntb = 1; pb.nqb = g.nqbB; pb.nsplit = g.nsB;
    // (both ranges as ONE launch - the second behind the first in block order, so that the 4 CUs the 252-block range
    // leaves free start on it at once - measured the same 36.8 ms: not kept)
    int rc = dispatch_ws<5>(p, g.mw, stride, bf16, st);
    if (rc) return rc;
    rc = dispatch_ws<4>(pb, g.mw, stride, bf16, st);
    if (rc) return rc;
    const int total = Co * Ci * Kt + Co;
    hipLaunchKernelGGL(ws_reduce2_kernel, dim3(step::ceil_div(total, 256)), dim3(256), 0, st, slabA, slabB, dbslab,
                       dw, db, Co, Ci, Kt, g.nsA, g.nsB, transposed);
    return step::check_launch("step_tcn_wgrad(sliding, reduce of two launches)");
}
