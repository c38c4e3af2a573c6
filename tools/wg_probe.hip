// Diagnostic harness: times step_tcn_wgrad (C=64, stride 1, T=120) with WG_PROBE switches.
#include "../step_amd/csrc/common.hip"
#include WG_SRC
#include <cstdlib>
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2048, C = 64, T = 120, Kt = 75;
    size_t ne = (size_t)N * C * T * 16;
    float *x, *du, *dw, *db, *sc, *sh; void* ws;
    hipMalloc(&x, ne * 4); hipMalloc(&du, ne * 4); hipMalloc(&dw, C * C * Kt * 4); hipMalloc(&db, C * 4);
    hipMalloc(&sc, C * 4); hipMalloc(&sh, C * 4);
    size_t wsb = step_tcn_wgrad_ws_bytes(N, C, C, T, T, Kt);
    hipMalloc(&ws, wsb);
    float* h = (float*)malloc(ne * 4);
    for (size_t i = 0; i < ne; ++i) h[i] = (rand() / (float)RAND_MAX) * 2 - 1;
    hipMemcpy(x, h, ne * 4, hipMemcpyHostToDevice); hipMemcpy(du, h, ne * 4, hipMemcpyHostToDevice);
    hipMemcpy(sc, h + 100, C * 4, hipMemcpyHostToDevice); hipMemcpy(sh, h + 200, C * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        int rc = step_tcn_wgrad(du, x, sc, sh, 1, dw, db, N, C, C, T, T, Kt, 1, 0, ws, wsb, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("wg probe %d rc %d: %.3f ms  %.1f TFLOP/s\n", WG_PROBE, rc, ms, 2.0 * C * C * Kt * T * 16 * N / ms / 1e9);
    }
    return 0;
}
