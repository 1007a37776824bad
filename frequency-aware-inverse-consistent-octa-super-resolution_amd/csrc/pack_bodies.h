// Block-level loops of the three weight-packing kernels, shared by the per-layer kernels (geometry in kernel arguments) and the
// batched kernel of conv_pack.hip (geometry in device memory): both write the same image.
//
// A block owns whole ROWS of an image (a row = the innermost, output-channel-major run of the layout), so the decomposition of the
// row index into (phase, chunk, tap, channel) is block-uniform scalar work and a thread's inner loop is one gather and one store.
// (The first version decomposed every ELEMENT with five runtime divisions: the batched launch took 1.18 ms for ~0.5 GB.)
#pragma once
#include "igemm_geom.h"
#include "split16.h"

namespace faoctasr {

// LDS-patch / narrow-map image  Wp[phase][chunk][r = t*KC + c][Mpad];  row = (phase, chunk, r)
__device__ __forceinline__ long patch_pack_rows(const PatchGeom& g) { return g.pack_off[4] / g.Mpad; }
__device__ __forceinline__ void patch_pack_block(const float* __restrict__ w, float* __restrict__ wp, const PatchGeom& g, long lb, long nb) {
    const int Mpad = g.Mpad;
    const long nrows = g.pack_off[4] / Mpad;
    for (long row = lb; row < nrows; row += nb) {
        int ph = 0;
        while (ph + 1 < g.nphase && row * Mpad >= g.pack_off[ph + 1]) ++ph;
        const long lr = row - g.pack_off[ph] / Mpad;                  // chunk * (KC*T) + t*KC + c
        const int T = g.t0[ph + 1] - g.t0[ph], KC = g.kc[ph];
        const int chunk = (int)(lr / (KC * T));
        const int r = (int)(lr - (long)chunk * KC * T);
        const int t = r / KC, c = chunk * KC + (r - t * KC);
        const bool cok = c < g.C;
        const float* src = w + (long)c * g.wsc + (g.taps[g.t0[ph] + t] >> 16);
        float* dst = wp + row * Mpad;
        for (int m = threadIdx.x; m < Mpad; m += 256) dst[m] = (cok && m < g.M) ? src[(long)m * g.wsm] : 0.f;
    }
}

// Winograd U = G g G^T in the order the kernel's weight waves stream it: [mtile][chunk][xi][k][m][j], c = chunk*8 + k + 4j;
// row = (mtile, chunk) = 8192 floats: a thread loads the 3x3 filter of one (m, c) once and writes its 16 transformed taps
__device__ __forceinline__ long wino_pack_rows(const WinoGeom& g) { return (long)g.mtiles * g.nchunks; }
__device__ __forceinline__ void wino_pack_block(const float* __restrict__ w, float* __restrict__ up, const WinoGeom& g, long lb, long nb) {
    const long nrows = (long)g.mtiles * g.nchunks;
    for (long row = lb; row < nrows; row += nb) {
        const int mt = (int)(row / g.nchunks), ch = (int)(row - (long)mt * g.nchunks);
        float* dst = up + row * (16 * WN_KC * WN_MT);
        for (int e = threadIdx.x; e < 512; e += 256) {                // e = k*128 + ml*2 + j
            const int j = e & 1, ml = (e >> 1) & 63, k = e >> 7;
            const int m = mt * WN_MT + (ml ^ (16 * (k & 1))), c = ch * WN_KC + k + 4 * j;      // slot ml holds row ml ^ 16 (k & 1)
            float u[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) u[q] = 0.f;
            if (m < g.M && c < g.C) {
                const float* wq = w + (long)m * g.wsm + (long)c * g.wsc;
                float gg[3][3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) gg[a][b] = wq[g.widx[a * 3 + b]];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    // row a of G applied to the columns, then row b of G applied to the result
                    float t[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        t[q] = a == 0 ? gg[0][q] : a == 3 ? gg[2][q] : a == 1 ? 0.5f * (gg[0][q] + gg[1][q] + gg[2][q]) : 0.5f * (gg[0][q] - gg[1][q] + gg[2][q]);
                    u[a * 4 + 0] = t[0];
                    u[a * 4 + 1] = 0.5f * (t[0] + t[1] + t[2]);
                    u[a * 4 + 2] = 0.5f * (t[0] - t[1] + t[2]);
                    u[a * 4 + 3] = t[2];
                }
            }
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) dst[xi * 512 + e] = u[xi];
        }
    }
}

// bf16x3 / f16x2: fp32 W -> two 16-bit planes (hi, lo) in the LDS image order  Wp[plane][phase][g16][tap][h][Mpad][8 ch];  row = (phase, g16, tap, h).
// F16: the planes hold w * s, s from the weights' absmax slot at the end of the image (written before this runs: split_absmax_block)
__device__ __forceinline__ long split_pack_rows(const SplitGeom& g) { return g.pack_off[4] / (8L * g.Mpad); }
template <bool F16>
__device__ __forceinline__ void split_pack_block(const float* __restrict__ w, unsigned short* __restrict__ wp, const SplitGeom& g, long lb, long nb) {
    const long rowlen = 8L * g.Mpad;
    const long nrows = g.pack_off[4] / rowlen;
    float sc = 1.f;
    if constexpr (F16) sc = f16x2_scale(split_w_absmax(reinterpret_cast<const unsigned*>(wp) + split_scale_slot(g)));
    for (long row = lb; row < nrows; row += nb) {
        int ph = 0;
        while (ph + 1 < g.nphase && row * rowlen >= g.pack_off[ph + 1]) ++ph;
        long lr = row - g.pack_off[ph] / rowlen;                      // (g16 * T + t) * 2 + h
        const int T = g.t0[ph + 1] - g.t0[ph];
        const int h = (int)(lr & 1); lr >>= 1;
        const int t = (int)(lr % T), g16 = (int)(lr / T);
        const int widx = g.taps[g.t0[ph] + t] >> 16;
        unsigned short* dst = wp + row * rowlen;
        for (int e = threadIdx.x; e < rowlen; e += 256) {             // e = m*8 + j
            const int j = e & 7, m = e >> 3;
            const int c = g16 * 16 + 8 * h + j;
            float v = 0.f;
            if (m < g.M && c < g.C) v = w[(long)m * g.wsm + (long)c * g.wsc + widx] * sc;
            unsigned hi, lo;
            split_pair<F16>(v, 0.f, hi, lo);
            dst[e] = (unsigned short)hi;
            dst[g.plane_stride + e] = (unsigned short)lo;
        }
    }
}

// the largest |w| of a weight tensor, as fp32 bit patterns, into the f16x2 image's slots: block `part` of SPLIT_WPARTS takes every
// SPLIT_WPARTS-th 16-byte piece and stores ITS maximum to slot `part` -- plain stores, nothing to zero, no atomics; readers take the
// maximum of the slots (split_w_absmax).  A layer on this route has at most 256 x 256 x 9 weights (2.4 MB).
__device__ __forceinline__ void split_absmax_block(const float* __restrict__ w, float* __restrict__ wp, const SplitGeom& g, int part, unsigned* red /* LDS, 4 words */) {
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    unsigned mx = 0;
    const bool vec = (reinterpret_cast<unsigned long>(w) & 15) == 0;
    const long n4 = vec ? g.w_elems >> 2 : 0;
    const u32x4v* w4 = reinterpret_cast<const u32x4v*>(w);
    for (long i = (long)part * 256 + threadIdx.x; i < n4; i += 256L * SPLIT_WPARTS) {
        const u32x4v v = w4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned b = v[j] & 0x7fffffffu;
            mx = b > mx ? b : mx;
        }
    }
    for (long i = (n4 << 2) + (long)part * 256 + threadIdx.x; i < g.w_elems; i += 256L * SPLIT_WPARTS) {
        const unsigned b = __builtin_bit_cast(unsigned, w[i]) & 0x7fffffffu;
        mx = b > mx ? b : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)mx, o, 64);
        mx = other > mx ? other : mx;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = red[0];
        for (int i = 1; i < 4; ++i) m = red[i] > m ? red[i] : m;
        reinterpret_cast<unsigned*>(wp)[split_scale_slot(g) + part] = m;
    }
}

}  // namespace faoctasr
