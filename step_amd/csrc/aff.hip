// Affective features of the hybrid classifier: 27 (29) posture / movement scalars per
// gait and their min-max normalisation over the data set.
//
// Replaces compute_aff_features/compute_features.py:428-468 (per-gait Python loops over frames: box
// volume :12-45, five angles :49-144, four distances :148-216, two Heron areas :220-256, mean speed /
// acceleration / jerk magnitude of hands, head and feet :260-389, stride length and gait-cycle time
// :393-425) with helpers compute_aff_features/utils.py:3-34, and normalize_features.py:1-22.
//
// One block of 128 threads per gait walks the (T, 48) record in chunks of 128 frames: the chunk (plus
// 3 frames of history and 1 of lookahead) is copied coalesced into LDS as doubles, one thread takes one
// frame, and 27 double sums are reduced in a fixed order (bit-reproducible).  Arithmetic is float64 as
// in the reference.  HBM-bound: every coordinate is read once (48 values per frame), 27-29 doubles are
// written per gait.
#include "common.h"

namespace {

constexpr int AF_NT = 128;            // threads = frames per chunk
constexpr int AF_HIST = 3;            // jerk looks 3 frames back
constexpr int AF_ROWS = AF_NT + AF_HIST + 1;
constexpr int AF_LD = 49;             // padded row: frame-per-lane reads hit distinct banks
constexpr int AF_NF = 27;

struct Foot {                         // strike points of one foot so far (compute_features.py:393-404)
    int cnt, first, last, prev_strike;
};

__device__ inline double ang(double ax, double ay, double az, double bx, double by, double bz) {
    // utils.py:7-19: unit vectors first, then the dot product, clipped, arccos in degrees
    const double na = sqrt(ax * ax + ay * ay + az * az), nb = sqrt(bx * bx + by * by + bz * bz);
    ax /= na; ay /= na; az /= na;
    bx /= nb; by /= nb; bz /= nb;
    double d = ax * bx + ay * by + az * bz;
    d = d < -1.0 ? -1.0 : (d > 1.0 ? 1.0 : d);          // NaN passes through both comparisons
    return (180.0 / 3.14159265358979323846) * acos(d);
}

__device__ inline double dist(const double* a, const double* b) {
    const double x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return sqrt(x * x + y * y + z * z);
}

__device__ inline double heron(const double* p, const double* q, const double* r) {
    const double a = dist(p, q), b = dist(p, r), c = dist(q, r);
    const double s = (a + b + c) / 2;
    return sqrt(s * (s - a) * (s - b) * (s - c));         // negative radicand -> NaN, as `** 0.5` gives
}

template <typename TIN>
__global__ __launch_bounds__(AF_NT) void aff_features_kernel(const TIN* __restrict__ gaits, double* __restrict__ feat,
                                                             int T, double dt, int stride_feats, int F) {
    __shared__ double tile[AF_ROWS * AF_LD];
    __shared__ double red[2][AF_NF];
    __shared__ unsigned long long masks[2][2][2];          // [foot][E|M][wave]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const TIN* g = gaits + (size_t)blockIdx.x * T * 48;

    double acc[AF_NF];
#pragma unroll
    for (int i = 0; i < AF_NF; ++i) acc[i] = 0.0;
    Foot foot{0, -1, -1, 0};                               // thread 0: right foot (12), thread 64: left (15)
    const int movers[5] = {6, 9, 3, 12, 15};               // compute_features.py:272-294

    for (int t0 = 0; t0 < T; t0 += AF_NT) {
        __syncthreads();
        // rows r <-> frames t0 - 3 + r, r in [0, 132)
        const int fbeg = t0 - AF_HIST < 0 ? 0 : t0 - AF_HIST;
        const int fend = t0 + AF_NT + 1 > T ? T : t0 + AF_NT + 1;
        for (int e = fbeg * 48 + tid; e < fend * 48; e += AF_NT) {
            const int f = e / 48, k = e - f * 48;
            tile[(f - t0 + AF_HIST) * AF_LD + k] = (double)g[e];
        }
        __syncthreads();
        const int t = t0 + tid;
        const double* fr = tile + (tid + AF_HIST) * AF_LD;
        if (t < T) {
            // ---- posture features of frame t
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int d = 0; d < 3; ++d) {                // :20-34: the max is an `elif` of the min
                    const double x = fr[3 * i + d];
                    if (lo[d] > x) lo[d] = x;
                    else if (hi[d] < x) hi[d] = x;
                }
            acc[0] += (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]) / 1000;
            const double *root = fr, *spine = fr + 3, *neck = fr + 6, *head = fr + 9, *rsh = fr + 12, *lsh = fr + 21;
            acc[1] += ang(rsh[0] - neck[0], rsh[1] - neck[1], rsh[2] - neck[2], lsh[0] - neck[0], lsh[1] - neck[1], lsh[2] - neck[2]);
            acc[2] += ang(neck[0] - rsh[0], neck[1] - rsh[1], neck[2] - rsh[2], lsh[0] - rsh[0], lsh[1] - rsh[1], lsh[2] - rsh[2]);
            acc[3] += ang(neck[0] - lsh[0], neck[1] - lsh[1], neck[2] - lsh[2], rsh[0] - lsh[0], rsh[1] - lsh[1], rsh[2] - lsh[2]);
            acc[4] += ang(head[0] - root[0], head[1] - root[1], head[2] - root[2], 0.0, 1.0, 0.0);
            acc[5] += ang(head[0] - neck[0], head[1] - neck[1], head[2] - neck[2], spine[0] - neck[0], spine[1] - neck[1], spine[2] - neck[2]);
            acc[6] += dist(fr + 18, root) / 10;
            acc[7] += dist(fr + 27, root) / 10;
            acc[8] += dist(fr + 36, root) / 10;
            acc[9] += dist(fr + 45, root) / 10;
            acc[10] += heron(fr + 27, neck, fr + 18) / 100;
            acc[11] += heron(fr + 45, root, fr + 36) / 100;
            // ---- movement features that end at frame t
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const int o = 3 * movers[m];
                double v0[3], v1[3], v2[3];                 // velocities ending at t, t-1, t-2
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const double p0 = fr[o + d];
                    const double p1 = t >= 1 ? fr[o + d - AF_LD] : 0.0;
                    const double p2 = t >= 2 ? fr[o + d - 2 * AF_LD] : 0.0;
                    const double p3 = t >= 3 ? fr[o + d - 3 * AF_LD] : 0.0;
                    v0[d] = p0 - p1; v1[d] = p1 - p2; v2[d] = p2 - p3;
                }
                if (t >= 1)                                  // :260-269  |dp| / 10 / dt
                    acc[12 + m] += sqrt(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]) / 10 / dt;
                if (t >= 2) {                                // :298-313  |(v - v')/dt| / 10, v = dp/dt
                    double a[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) a[d] = (v0[d] / dt - v1[d] / dt) / dt;
                    acc[17 + m] += sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) / 10;
                    if (t >= 3) {                            // :342-364
                        double j[3];
#pragma unroll
                        for (int d = 0; d < 3; ++d) j[d] = (a[d] - (v1[d] / dt - v2[d] / dt) / dt) / dt;
                        acc[22 + m] += sqrt(j[0] * j[0] + j[1] * j[1] + j[2] * j[2]) / 10;
                    }
                }
            }
        }
        if (stride_feats) {
            // strike tests of frame t (1 <= t <= T-2): E = plateau (prev == curr == next), M = local minimum
            bool e[2] = {false, false}, m[2] = {false, false};
            if (t >= 1 && t < T - 1) {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int o = 3 * (f ? 15 : 12) + 1;
                    const double p = fr[o - AF_LD], c = fr[o], n = fr[o + AF_LD];
                    e[f] = (p == c) && (c == n);
                    m[f] = !e[f] && (p >= c) && (c <= n);
                }
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const unsigned long long be = __ballot(e[f]), bm = __ballot(m[f]);
                if (lane == 0) { masks[f][0][wave] = be; masks[f][1][wave] = bm; }
            }
            __syncthreads();
            if (tid == 0 || tid == 64) {
                const int f = tid >> 6;
                for (int w = 0; w < 2; ++w) {
                    unsigned long long be = masks[f][0][w], bm = masks[f][1][w];
                    const int base = t0 + 64 * w;
                    if (be == 0ull) {                        // no plateau in these 64 frames: strikes = M
                        if (bm) {
                            const int lo_i = base + __ffsll((long long)bm) - 1;
                            const int hi_i = base + 63 - __clzll((long long)bm);
                            if (foot.cnt == 0) foot.first = lo_i;
                            foot.last = hi_i;
                            foot.cnt += __popcll(bm);
                            foot.prev_strike = (bm >> 63) & 1ull;
                        } else {
                            foot.prev_strike = 0;
                        }
                    } else {                                 // :399-401: on a plateau every other frame counts
                        for (int b = 0; b < 64; ++b) {
                            const bool ee = (be >> b) & 1ull, mm = (bm >> b) & 1ull;
                            const bool s = mm || (ee && !foot.prev_strike);
                            if (s) {
                                if (foot.cnt == 0) foot.first = base + b;
                                foot.last = base + b;
                                ++foot.cnt;
                            }
                            foot.prev_strike = s;
                        }
                    }
                }
            }
        }
    }

    // ---- block reduction in a fixed order: lanes by xor-shuffle, then wave 0 + wave 1
#pragma unroll
    for (int i = 0; i < AF_NF; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __shared__ int fs[2][3];
    if (stride_feats && (tid == 0 || tid == 64)) {
        fs[tid >> 6][0] = foot.cnt; fs[tid >> 6][1] = foot.first; fs[tid >> 6][2] = foot.last;
    }
    __syncthreads();
    double* out = feat + (size_t)blockIdx.x * F;
    if (tid < AF_NF) {
        const int cnt = tid < 12 ? T : (tid < 17 ? T - 1 : (tid < 22 ? T - 2 : T - 3));
        out[tid] = (red[0][tid] + red[1][tid]) / (double)(cnt > 0 ? cnt : 0);     // empty mean -> NaN, as numpy
    }
    if (stride_feats && tid == 0) {                         // :407-420
        const int cr = fs[0][0], cl = fs[1][0];
        const double dr = fs[0][2] - fs[0][1], dl = fs[1][2] - fs[1][1];
        double len;
        if (cr < 2 && cl < 2) len = (double)T;
        else if (cr < 2) len = (dl + (cl - 1)) / (double)(cl - 1);             // mean(diff + 1)
        else if (cl < 2) len = (dr + (cr - 1)) / (double)(cr - 1);
        else len = (dr + dl) / (double)(cr - 1 + cl - 1);
        out[AF_NF] = len;
        out[AF_NF + 1] = len * dt;
    }
}

// normalize_features.py:1-22, one block per feature column; NaNs never win a comparison there
// (`min > x`, `max < x`), neither do they here.
template <typename TOUT>
__global__ __launch_bounds__(256) void aff_normalize_kernel(const double* __restrict__ feat, TOUT* __restrict__ out,
                                                            int N, int F) {
    __shared__ double slo[256], shi[256];
    const int j = blockIdx.x, tid = threadIdx.x;
    double lo = INFINITY, hi = -INFINITY;
    for (int n = tid; n < N; n += 256) {
        const double x = feat[(size_t)n * F + j];
        if (lo > x) lo = x;
        if (hi < x) hi = x;
    }
    slo[tid] = lo; shi[tid] = hi;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) {
            if (slo[tid] > slo[tid + s]) slo[tid] = slo[tid + s];
            if (shi[tid] < shi[tid + s]) shi[tid] = shi[tid + s];
        }
        __syncthreads();
    }
    const double a = (shi[0] + slo[0]) / 2, b = (shi[0] - slo[0]) / 2;
    for (int n = tid; n < N; n += 256) {
        const double x = feat[(size_t)n * F + j];
        out[(size_t)n * F + j] = (TOUT)(b == 0 ? 0.0 : (x - a) / b);
    }
}

}  // namespace

extern "C" int step_aff_features(const void* gaits, int in_f64, double* feat, int N, int T, double time_step,
                                 int add_stride_features, void* stream) {
    STEP_REQUIRE(gaits && feat, "step_aff_features: null pointer");
    STEP_REQUIRE(N >= 1 && T >= 1, "step_aff_features: bad shape N=%d T=%d", N, T);
    STEP_REQUIRE((long long)T * 48 <= 0x7fffffffLL, "step_aff_features: T=%d too long", T);
    STEP_REQUIRE(time_step > 0, "step_aff_features: time_step must be positive");
    const int F = AF_NF + (add_stride_features ? 2 : 0);
    hipStream_t st = (hipStream_t)stream;
    if (in_f64)
        hipLaunchKernelGGL(aff_features_kernel<double>, dim3(N), dim3(AF_NT), 0, st, (const double*)gaits, feat, T,
                           time_step, add_stride_features, F);
    else
        hipLaunchKernelGGL(aff_features_kernel<float>, dim3(N), dim3(AF_NT), 0, st, (const float*)gaits, feat, T,
                           time_step, add_stride_features, F);
    return step::check_launch("step_aff_features");
}

extern "C" int step_aff_normalize(const double* feat, void* out, int out_f64, int N, int F, void* stream) {
    STEP_REQUIRE(feat && out, "step_aff_normalize: null pointer");
    STEP_REQUIRE(N >= 1 && F >= 1, "step_aff_normalize: bad shape N=%d F=%d", N, F);
    hipStream_t st = (hipStream_t)stream;
    if (out_f64)
        hipLaunchKernelGGL(aff_normalize_kernel<double>, dim3(F), dim3(256), 0, st, feat, (double*)out, N, F);
    else
        hipLaunchKernelGGL(aff_normalize_kernel<float>, dim3(F), dim3(256), 0, st, feat, (float*)out, N, F);
    return step::check_launch("step_aff_normalize");
}
