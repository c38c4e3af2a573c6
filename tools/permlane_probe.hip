// what v_permlane16_swap does to two registers: prints the source (register, lane row) of every row of both results
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    unsigned a = 100 + lane / 16, b = 200 + lane / 16;      // a rows: 100..103, b rows: 200..203
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = (int)r[0];
    out[64 + lane] = (int)r[1];
}
int main() {
    int* d; int h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("r0 rows: %d %d %d %d\nr1 rows: %d %d %d %d\n", h[0], h[16], h[32], h[48], h[64], h[80], h[96], h[112]);
    return 0;
}
