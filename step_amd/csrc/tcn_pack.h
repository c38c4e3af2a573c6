// Geometry and weight mapping of the generic temporal-convolution form shared by the fp32
// (tcn.hip) and bf16 (tcn_bf16.hip) implicit-GEMM kernels:
//   out[n, c, PH*rho + ph, v] = epi( SUM_cg SUM_jj wgen[cg][jj][ph*Cpad + c] *
//                                    pre(in[n, cg, S*rho + jj - padL, v]) )
// pack_mode (step_hip.h): 0 Conv2d forward, 1 ConvTranspose2d forward, 2 Conv2d input
// gradient, 3 ConvTranspose2d input gradient.
#pragma once
#include "common.h"

namespace step {

struct PackGeom {
    int cin_g, cout_g, cpad, ph, ks, padL, stride_g;
};

inline bool pack_geom(int Co, int Ci, int Kt, int stride, int mode, PackGeom* g) {
    if (Kt < 1 || (Kt & 1) == 0) return false;
    const int p = (Kt - 1) / 2;
    g->ph = 1;
    g->stride_g = 1;
    g->ks = (Kt + 1) / 2;
    g->padL = p;                               // Kt-1-p == p for odd Kt
    switch (mode) {
        case 0: g->cin_g = Ci; g->cout_g = Co; g->stride_g = stride; break;
        case 1: if (stride != 1) return false; g->cin_g = Ci; g->cout_g = Co; break;
        case 2:
            g->cin_g = Co; g->cout_g = Ci;
            if (stride == 2) {
                g->ph = 2;
                g->padL = p / 2;
                int slots = 0;
                for (int ph = 0; ph < 2; ++ph) {
                    const int jmin = (ph + p) & 1;
                    const int s = (ph + p + 2 * g->padL - jmin) / 2 + 1;
                    slots = s > slots ? s : slots;
                }
                g->ks = (slots + 1) / 2;
            } else if (stride != 1) return false;
            break;
        case 3: if (stride != 1) return false; g->cin_g = Co; g->cout_g = Ci; break;
        default: return false;
    }
    if (stride != 1 && stride != 2) return false;
    g->cpad = round_up(g->cout_g, 32);
    return true;
}

// wgen[cg][jj][m] of the generic form, read from the module's weight tensor w
__device__ __forceinline__ float pack_value(const float* __restrict__ w, int Co, int Ci, int Kt,
                                            int stride, int mode, const PackGeom& g, int cg, int jj,
                                            int m) {
    const int p = (Kt - 1) / 2;
    const int ph = m / g.cpad;
    const int c = m % g.cpad;                      // GEMM output channel
    float val = 0.f;
    if (c < g.cout_g && cg < g.cin_g) {
        if (mode == 0) {                           // w (Co,Ci,Kt): cg = ci, c = co
            if (jj < Kt) val = w[((size_t)c * Ci + cg) * Kt + jj];
        } else if (mode == 1) {                    // w (Ci,Co,Kt): cg = ci, c = co
            if (jj < Kt) val = w[((size_t)cg * Co + c) * Kt + (Kt - 1 - jj)];
        } else if (mode == 2) {                    // w (Co,Ci,Kt): cg = co, c = ci
            int j;
            if (stride == 1) j = Kt - 1 - jj;
            else j = ph + p + 2 * g.padL - 2 * jj;
            if (j >= 0 && j < Kt) val = w[((size_t)cg * Ci + c) * Kt + j];
        } else {                                   // w (Ci,Co,Kt): cg = co, c = ci
            if (jj < Kt) val = w[((size_t)c * Co + cg) * Kt + jj];
        }
    }
    return val;
}

}  // namespace step
