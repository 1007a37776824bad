// Element loops of the three weight-packing kernels, shared by the per-layer kernels (geometry in kernel arguments) and the
// batched kernel of conv_pack.hip (geometry in device memory): element i of a packed image is computed the same way by both.
#pragma once
#include "igemm_geom.h"

namespace faoctasr {

// LDS-patch / narrow-map image  Wp[phase][chunk][r = t*KC + c][Mpad]
__device__ __forceinline__ void patch_pack_elems(const float* __restrict__ w, float* __restrict__ wp, const PatchGeom& g, long total,
                                                 long first, long stride) {
    for (long i = first; i < total; i += stride) {
        int ph = 0;
        while (ph + 1 < g.nphase && i >= g.pack_off[ph + 1]) ++ph;
        const long li = i - g.pack_off[ph];
        const int T = g.t0[ph + 1] - g.t0[ph], KC = g.kc[ph];
        const int m = (int)(li % g.Mpad);
        const long row = li / g.Mpad;                 // chunk * (KC*T) + t*KC + c
        const int chunk = (int)(row / (KC * T));
        const int r = (int)(row - (long)chunk * KC * T);
        const int t = r / KC, c = chunk * KC + (r - t * KC);
        float v = 0.f;
        if (m < g.M && c < g.C) v = w[(long)m * g.wsm + (long)c * g.wsc + (g.taps[g.t0[ph] + t] >> 16)];
        wp[i] = v;
    }
}

// Winograd U = G g G^T in the order the kernel's weight waves stream it: [mtile][chunk][xi][k][m][j], c = chunk*8 + k + 4j
__device__ __forceinline__ void wino_pack_elems(const float* __restrict__ w, float* __restrict__ up, const WinoGeom& g, long total,
                                                long first, long stride) {
    for (long i = first; i < total; i += stride) {
        long li = i;
        const int j = (int)(li & 1); li >>= 1;
        const int ml = (int)(li & 63); li >>= 6;
        const int k = (int)(li & 3); li >>= 2;
        const int xi = (int)(li & 15); li >>= 4;
        const int ch = (int)(li % g.nchunks);
        const int mt = (int)(li / g.nchunks);
        const int m = mt * WN_MT + (ml ^ (16 * (k & 1))), c = ch * WN_KC + k + 4 * j;      // slot ml holds row ml ^ 16 (k & 1)
        float v = 0.f;
        if (m < g.M && c < g.C) {
            const float* wq = w + (long)m * g.wsm + (long)c * g.wsc;
            float gg[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) gg[a][b] = wq[g.widx[a * 3 + b]];
            const int a = xi >> 2, b = xi & 3;
            // row a of G applied to the columns, then row b of G applied to the result
            float t[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                t[q] = a == 0 ? gg[0][q] : a == 3 ? gg[2][q] : a == 1 ? 0.5f * (gg[0][q] + gg[1][q] + gg[2][q]) : 0.5f * (gg[0][q] - gg[1][q] + gg[2][q]);
            }
            v = b == 0 ? t[0] : b == 3 ? t[2] : b == 1 ? 0.5f * (t[0] + t[1] + t[2]) : 0.5f * (t[0] - t[1] + t[2]);
        }
        up[i] = v;
    }
}

// bf16x3: fp32 W -> two bf16 planes (hi, lo) in the LDS image order  Wp[plane][phase][g16][tap][h][Mpad][8 ch]
__device__ __forceinline__ void split_pack_elems(const float* __restrict__ w, __bf16* __restrict__ wp, const SplitGeom& g, long total,
                                                 long first, long stride) {
    for (long i = first; i < total; i += stride) {
        int ph = 0;
        while (ph + 1 < g.nphase && i >= g.pack_off[ph + 1]) ++ph;
        long li = i - g.pack_off[ph];
        const int T = g.t0[ph + 1] - g.t0[ph];
        const int j = (int)(li & 7); li >>= 3;
        const int m = (int)(li % g.Mpad); li /= g.Mpad;
        const int h = (int)(li & 1); li >>= 1;
        const int t = (int)(li % T);
        const int g16 = (int)(li / T);
        const int c = g16 * 16 + 8 * h + j;
        float v = 0.f;
        if (m < g.M && c < g.C) v = w[(long)m * g.wsm + (long)c * g.wsc + (g.taps[g.t0[ph] + t] >> 16)];
        const __bf16 hi = (__bf16)v;
        wp[i] = hi;
        wp[g.plane_stride + i] = (__bf16)(v - (float)hi);
    }
}

}  // namespace faoctasr
