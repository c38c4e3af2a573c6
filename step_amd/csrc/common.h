// Shared helpers for libstep_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/step_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace step {

void set_error(const char* fmt, ...);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// hipGetLastError after a launch -> status
int check_launch(const char* what);

// ordered sum of the per-slice weight-gradient slabs [nsplit][Co][Ci*Kt] (+ [nsplit][Co] bias
// partials) into dw / db; shared by the fp32 and bf16 wgrad kernels (tcn_wgrad.hip)
int wgrad_reduce(const float* slab, const float* dbslab, float* dw, float* db, int Co, int Ci, int Kt,
                 int nsplit, int transposed, hipStream_t st, const char* what);


// fp32 sliding-window wgrad (tcn_wgrad_sw.hip): workspace bytes (0 = shape not covered) and the
// launch; wgrad_sw returns 1 when the shape is not covered or the workspace is too small for it
size_t wgrad_sw_ws_bytes(int N, int Ci, int Co, int Kt, int stride);
int wgrad_sw(const float* dout, const float* in, const float* in_scale, const float* in_shift, int in_relu,
             float* dw, float* db, int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride,
             int transposed, void* ws, size_t ws_bytes, hipStream_t st, int bf16 = 0);

// bn.hip: ordered sum of producer-epilogue statistic partials [nchunks][G][2] -> sums [G][2]
int bn_sum_partials(const double* part, double* sums, int G, int nchunks, hipStream_t st);

}  // namespace step

// raw buffer access: descriptor (uniform base) + 32-bit lane offset + uniform SGPR offset - no 64-bit VALU address
// arithmetic per access (it would all add to the MFMA time: fp32 MFMAs and the same wave's VALU do not overlap)
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffffc, 0x00020000);
}
static __device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
static __device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}


#define STEP_REQUIRE(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            step::set_error(__VA_ARGS__);               \
            return STEP_ERR_ARG;                        \
        }                                               \
    } while (0)
