// Diagnostic harness: times step_tcn_conv (C=64, stride 1, T=120) with parts of the kernel
// disabled through TCN_PROBE (see the #if blocks it enables in a scratch copy of tcn.hip).
#include "../step_amd/csrc/common.hip"
#include TCN_SRC
#include <cstdlib>
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2048, C = 64, T = 120, Kt = 75;
    size_t ne = (size_t)N * C * T * 16;
    float *x, *y, *w, *wp, *b, *sc, *sh;
    hipMalloc(&x, ne * 4); hipMalloc(&y, ne * 4); hipMalloc(&w, C * C * Kt * 4); hipMalloc(&b, C * 4);
    hipMalloc(&sc, C * 4); hipMalloc(&sh, C * 4);
    size_t np = step_tcn_packed_elems(C, C, Kt, 1, 0);
    hipMalloc(&wp, np * 4);
    float* h = (float*)malloc(ne * 4);
    for (size_t i = 0; i < ne; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
    hipMemcpy(x, h, ne * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h, C * C * Kt * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, h, C * 4, hipMemcpyHostToDevice); hipMemcpy(sc, h + 100, C * 4, hipMemcpyHostToDevice);
    hipMemcpy(sh, h + 200, C * 4, hipMemcpyHostToDevice);
    step_tcn_pack_weight(w, wp, C, C, Kt, 1, 0, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        int rc = step_tcn_conv(x, sc, sh, 1, wp, b, 0, 0, 0, 0, 0, 0, y, N, C, C, T, T, Kt, 1, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("probe %d rc %d: %.3f ms  %.1f TFLOP/s\n", TCN_PROBE, rc, ms, 2.0 * C * C * Kt * T * 16 * N / ms / 1e9);
    }
    return 0;
}
