// bf16-operand weight gradient of the temporal convolution: entry points.
//
// Same operator as step_tcn_wgrad (weight part of convolution_backward for tcn[2],
// classifier_stgcn_real_only/net/classifier.py:156-162).  The kernel is the sliding-window kernel of
// tcn_wgrad_sw.hip in its BF16 mode: dout and the raw y strip are staged by LDS-DMA as fp32, the
// MFMA waves apply BN + ReLU to the fragments they read and round both operands to bf16 in
// registers (one v_mfma_f32_32x32x16_bf16 per frame and accumulator tile), accumulation is fp32 and
// the bias gradient stays an exact fp32 sum.
#include "common.h"

extern "C" size_t step_tcn_wgrad_bf16_ws_bytes(int N, int Ci, int Co, int Tin, int Tout, int Kt, int stride) {
    (void)Tin; (void)Tout;
    return step::wgrad_sw_ws_bytes(N, Ci, Co, Kt, stride);
}

static int wgrad_bf16(const float* dout, const float* in, const float* in_scale,
                                   const float* in_shift, int in_relu, float* dw, float* db, int N, int Ci,
                                   int Co, int Tin, int Tout, int Kt, int stride, int transposed, void* ws,
                                   size_t ws_bytes, void* stream) {
    STEP_REQUIRE(!transposed || stride == 1, "step_tcn_wgrad_bf16_t: transposed needs stride 1");
    STEP_REQUIRE(dout && in && dw && ws, "step_tcn_wgrad_bf16: null pointer");
    STEP_REQUIRE(N > 0 && Tin > 0 && (stride == 1 || stride == 2) && Tout == (Tin - 1) / stride + 1,
                 "step_tcn_wgrad_bf16: bad shape Tin=%d Tout=%d stride=%d", Tin, Tout, stride);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "step_tcn_wgrad_bf16: scale/shift pair");
    const size_t need = step::wgrad_sw_ws_bytes(N, Ci, Co, Kt, stride);
    STEP_REQUIRE(need > 0, "step_tcn_wgrad_bf16: unsupported Ci=%d Co=%d Kt=%d stride=%d", Ci, Co, Kt, stride);
    if (ws_bytes < need) {
        step::set_error("step_tcn_wgrad_bf16: workspace %zu < %zu bytes", ws_bytes, need);
        return STEP_ERR_WORKSPACE;
    }
    return step::wgrad_sw(dout, in, in_scale, in_shift, in_relu, dw, db, N, Ci, Co, Tin, Tout, Kt, stride,
                          transposed, ws, ws_bytes, (hipStream_t)stream, 1);
}

extern "C" int step_tcn_wgrad_bf16(const float* dout, const float* in, const float* in_scale,
                                   const float* in_shift, int in_relu, float* dw, float* db, int N, int Ci,
                                   int Co, int Tin, int Tout, int Kt, int stride, void* ws, size_t ws_bytes,
                                   void* stream) {
    return wgrad_bf16(dout, in, in_scale, in_shift, in_relu, dw, db, N, Ci, Co, Tin, Tout, Kt, stride, 0, ws,
                      ws_bytes, stream);
}

// the same with the ConvTranspose2d weight layout (Ci, Co, Kt), taps reversed (st_gctn blocks, CVAE_stgcn.py:359-365)
extern "C" int step_tcn_wgrad_bf16_t(const float* dout, const float* in, const float* in_scale,
                                     const float* in_shift, int in_relu, float* dw, float* db, int N, int Ci,
                                     int Co, int Tin, int Tout, int Kt, int stride, int transposed, void* ws,
                                     size_t ws_bytes, void* stream) {
    return wgrad_bf16(dout, in, in_scale, in_shift, in_relu, dw, db, N, Ci, Co, Tin, Tout, Kt, stride, transposed,
                      ws, ws_bytes, stream);
}

// fp32 operands split into bf16 hi + lo, three bf16 MFMAs per product (tcn_wgrad_sw.hip, mode 2): same arguments
extern "C" int step_tcn_wgrad_split(const float* dout, const float* in, const float* in_scale,
                                    const float* in_shift, int in_relu, float* dw, float* db, int N, int Ci,
                                    int Co, int Tin, int Tout, int Kt, int stride, int transposed, void* ws,
                                    size_t ws_bytes, void* stream) {
    STEP_REQUIRE(!transposed || stride == 1, "step_tcn_wgrad_split: transposed needs stride 1");
    STEP_REQUIRE(dout && in && dw && ws, "step_tcn_wgrad_split: null pointer");
    STEP_REQUIRE(N > 0 && Tin > 0 && (stride == 1 || stride == 2) && Tout == (Tin - 1) / stride + 1,
                 "step_tcn_wgrad_split: bad shape Tin=%d Tout=%d stride=%d", Tin, Tout, stride);
    STEP_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "step_tcn_wgrad_split: scale/shift pair");
    const size_t need = step::wgrad_sw_ws_bytes(N, Ci, Co, Kt, stride);
    STEP_REQUIRE(need > 0, "step_tcn_wgrad_split: unsupported Ci=%d Co=%d Kt=%d stride=%d", Ci, Co, Kt, stride);
    if (ws_bytes < need) {
        step::set_error("step_tcn_wgrad_split: workspace %zu < %zu bytes", ws_bytes, need);
        return STEP_ERR_WORKSPACE;
    }
    return step::wgrad_sw(dout, in, in_scale, in_shift, in_relu, dw, db, N, Ci, Co, Tin, Tout, Kt, stride,
                          transposed, ws, ws_bytes, (hipStream_t)stream, 2);
}
