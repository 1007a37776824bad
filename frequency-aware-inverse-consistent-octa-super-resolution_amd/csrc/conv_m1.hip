// Single-output-channel convolution (the generator head 64 -> 1, 3x3: model.py:438) for gfx950.
// With M = 1 an MFMA tile would be 63/64 padding, and the op is HBM-bound anyway (read 64 input planes, write 1):
// these are plain VALU kernels over an LDS-staged input patch.
//   forward : y[n][0][oy][ox] = act( sum_{c,t} w[c][t] * x[n][c][oy - pad + kh][ox - pad + kw] + bias )   (stride 1)
//   wgrad   : dw[c][t]       += sum_{n,oy,ox} dy[n][0][oy][ox] * x[n][c][oy - pad + kh][ox - pad + kw]
// (the input gradient has C = 1 gathered channel and M = 64 outputs and stays on the MFMA path.)
#include "common.h"

namespace faoctasr {

constexpr int M1_TH = 8, M1_TW = 32, M1_KC = 16;

__global__ __launch_bounds__(256) void conv_m1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int N, int C,
                                                          int H, int W, int KH, int KW, int pad, int act, float slope) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* P = reinterpret_cast<float*>(smem);                        // [KC][PH][PW]
    const int PH = M1_TH + KH - 1, PW = M1_TW + KW - 1, PHW = PH * PW, T = KH * KW;
    float* Wl = P + M1_KC * PHW;                                      // [KC][T]
    const int tid = threadIdx.x;
    const int tiles_x = (W + M1_TW - 1) / M1_TW, tiles_y = (H + M1_TH - 1) / M1_TH;
    const int n = blockIdx.x / (tiles_x * tiles_y);
    const int r = blockIdx.x - n * tiles_x * tiles_y;
    const int y0 = (r / tiles_x) * M1_TH, x0 = (r % tiles_x) * M1_TW;
    const int ty = tid >> 5, tx = tid & 31;
    const long hw = (long)H * W;
    const float* xn = x + (long)n * C * hw;
    constexpr unsigned OOB = 0x7fffffffu;
    const float invPHW = 1.0f / (float)PHW, invPW = 1.0f / (float)PW;
    float acc = 0.f;
    for (int c0 = 0; c0 < C; c0 += M1_KC) {
        const long bytes = (long)(C - c0) * hw * 4;
        const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn + (long)c0 * hw), 0,
                                                           (int)(bytes < 0x7ffffff0L ? bytes : 0x7ffffff0L), 0x00020000);
        __syncthreads();
        for (int e = tid; e < M1_KC * PHW; e += 256) {
            const int c = (int)(((float)e + 0.5f) * invPHW);
            const int q = e - c * PHW;
            const int py = (int)(((float)q + 0.5f) * invPW);
            const int iy = y0 - pad + py, ix = x0 - pad + (q - py * PW);
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            P[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, ok ? 4u * (unsigned)(c * (int)hw + iy * W + ix) : OOB, 0, 0));
        }
        for (int e = tid; e < M1_KC * T; e += 256) Wl[e] = (c0 + e / T < C) ? w[(long)c0 * T + e] : 0.f;
        __syncthreads();
        const float* pp = P + ty * PW + tx;
        for (int c = 0; c < M1_KC; ++c)
            for (int kh = 0; kh < KH; ++kh)
                for (int kw = 0; kw < KW; ++kw) acc += Wl[c * T + kh * KW + kw] * pp[c * PHW + kh * PW + kw];
    }
    const int oy = y0 + ty, ox = x0 + tx;
    if (oy < H && ox < W) {
        if (bias) acc += bias[0];
        y[(long)n * hw + (long)oy * W + ox] = act_apply(acc, act, slope);
    }
}

// one thread = one (channel, tap) weight of the current chunk; it walks the pixels of the tile
__global__ __launch_bounds__(256) void conv_m1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw, int N, int C, int H, int W, int KH, int KW, int pad,
                                                            int tiles_per_block, int CCH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int PH = M1_TH + KH - 1, PW = M1_TW + KW - 1, PHW = PH * PW, T = KH * KW;
    float* P = reinterpret_cast<float*>(smem);                        // [CCH][PH][PW]
    float* D = P + CCH * PHW;                                         // [TH*TW]
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * CCH;
    const int cl = tid / T, t = tid - cl * T;                         // this thread's (local channel, tap)
    const bool mine = cl < CCH && c0 + cl < C;
    const int kh = t / KW, kw = t - kh * KW;
    const int tiles_x = (W + M1_TW - 1) / M1_TW, tiles_y = (H + M1_TH - 1) / M1_TH;
    const long ntiles = (long)N * tiles_x * tiles_y;
    const long hw = (long)H * W;
    constexpr unsigned OOB = 0x7fffffffu;
    const float invPHW = 1.0f / (float)PHW, invPW = 1.0f / (float)PW;
    float acc = 0.f;
    long tile = (long)blockIdx.x * tiles_per_block;
    long tend = tile + tiles_per_block;
    tend = tend < ntiles ? tend : ntiles;
    for (; tile < tend; ++tile) {
        const int n = (int)(tile / (tiles_x * tiles_y));
        const int r = (int)(tile - (long)n * tiles_x * tiles_y);
        const int y0 = (r / tiles_x) * M1_TH, x0 = (r % tiles_x) * M1_TW;
        const long bytes = (long)(C - c0) * hw * 4;
        const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + ((long)n * C + c0) * hw), 0,
                                                           (int)(bytes < 0x7ffffff0L ? bytes : 0x7ffffff0L), 0x00020000);
        __syncthreads();
        for (int e = tid; e < CCH * PHW; e += 256) {
            const int c = (int)(((float)e + 0.5f) * invPHW);
            const int q = e - c * PHW;
            const int py = (int)(((float)q + 0.5f) * invPW);
            const int iy = y0 - pad + py, ix = x0 - pad + (q - py * PW);
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            P[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, ok ? 4u * (unsigned)(c * (int)hw + iy * W + ix) : OOB, 0, 0));
        }
        {
            const int py = tid >> 5, px = tid & 31;
            const int oy = y0 + py, ox = x0 + px;
            D[tid] = (oy < H && ox < W) ? dy[(long)n * hw + (long)oy * W + ox] : 0.f;
        }
        __syncthreads();
        if (mine) {
            const float* pp = P + cl * PHW + kh * PW + kw;
#pragma unroll 4
            for (int py = 0; py < M1_TH; ++py)
#pragma unroll 8
                for (int px = 0; px < M1_TW; ++px) acc += D[py * M1_TW + px] * pp[py * PW + px];
        }
    }
    if (mine) atomicAdd(dw + (long)(c0 + cl) * T + t, acc);
}

// ---- 3x3 fast forms (round 2): no LDS, no barrier ---------------------------------------------------------------------------
// The LDS-patch kernels above read two LDS words per FMA (158 / 174 us for the 64 -> 1 head at 256^2, batch 8: 0.85 TB/s on a
// 134 MB read).  For the 3x3 head a thread owns 4 adjacent pixels of a row; per (channel, input row) it loads its aligned float4 and
// the two neighbouring columns (L1 hits), the weights arrive through the scalar cache (uniform index), and the 36 FMAs per channel
// run out of registers.  W % 4 == 0.
typedef float m1f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void m1_row6(const float* __restrict__ row, int xq, int W, bool rowok, float (&v)[6]) {
    // v[0..5] = columns xq-1 .. xq+4 of `row`, zero outside the image
    m1f4 c = {0.f, 0.f, 0.f, 0.f};
    float l = 0.f, r = 0.f;
    if (rowok) {
        c = *reinterpret_cast<const m1f4*>(row + xq);
        if (xq > 0) l = row[xq - 1];
        if (xq + 4 < W) r = row[xq + 4];
    }
    v[0] = l; v[1] = c[0]; v[2] = c[1]; v[3] = c[2]; v[4] = c[3]; v[5] = r;
}

__global__ __launch_bounds__(256) void conv_m1_fwd3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, int N, int C, int H, int W, int act, float slope) {
    const int wq = W >> 2;                                              // 4-pixel groups per row
    const long item = (long)blockIdx.x * 256 + threadIdx.x;             // (n, row, group)
    const long total = (long)N * H * wq;
    if (item >= total) return;
    const int xq = (int)(item % wq) * 4;
    const long t = item / wq;
    const int oy = (int)(t % H), n = (int)(t / H);
    const long hw = (long)H * W;
    const float* xn = x + (long)n * C * hw;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
        const float* wc = w + c * 9;                                    // uniform: scalar loads
        const float* xc = xn + (long)c * hw;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = oy - 1 + kh;
            float v[6];
            m1_row6(xc + (long)iy * W, xq, W, (unsigned)iy < (unsigned)H, v);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float wv = wc[kh * 3 + kw];
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[p] += wv * v[p + kw];
            }
        }
    }
    const float b = bias ? bias[0] : 0.f;
    m1f4 o = {act_apply(acc[0] + b, act, slope), act_apply(acc[1] + b, act, slope), act_apply(acc[2] + b, act, slope), act_apply(acc[3] + b, act, slope)};
    *reinterpret_cast<m1f4*>(y + (long)n * hw + (long)oy * W + xq) = o;
}

// weight gradient: a wave owns (image n, 4 channels, a segment of rows); lane = 4-pixel column group (strided over the row).  The
// three most recent input rows of each channel live in a register ring, so every x element is loaded once; 36 accumulators (4
// channels x 9 taps) per lane are summed over the wave at the end and added atomically (one lane per value).
constexpr int M1W_CH = 4, M1W_ROWS = 32;
__global__ __launch_bounds__(256) void conv_m1_wgrad3_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N,
                                                             int C, int H, int W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cgroups = (C + M1W_CH - 1) / M1W_CH, rsegs = (H + M1W_ROWS - 1) / M1W_ROWS;
    const long item = (long)blockIdx.x * 4 + wave;                      // (n, row segment, channel group)
    if (item >= (long)N * rsegs * cgroups) return;                       // whole waves only; no barrier in this kernel
    const int cg = (int)(item % cgroups);
    const long t = item / cgroups;
    const int rs = (int)(t % rsegs), n = (int)(t / rsegs);
    const int c0 = cg * M1W_CH, r0 = rs * M1W_ROWS, r1 = r0 + M1W_ROWS < H ? r0 + M1W_ROWS : H;
    const long hw = (long)H * W;
    const float* dyn = dy + (long)n * hw;
    float acc[M1W_CH][9];
#pragma unroll
    for (int c = 0; c < M1W_CH; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[c][k] = 0.f;
    for (int xq = lane * 4; xq < W; xq += 256) {
        float ring[M1W_CH][3][6];                                        // rows oy-1, oy, oy+1 of each channel
#pragma unroll
        for (int c = 0; c < M1W_CH; ++c) {
            const bool cok = c0 + c < C;
            const float* xc = x + ((long)n * C + (cok ? c0 + c : 0)) * hw;
            m1_row6(xc + (long)(r0 - 1) * W, xq, W, cok && r0 - 1 >= 0, ring[c][0]);
            m1_row6(xc + (long)r0 * W, xq, W, cok, ring[c][1]);
        }
        for (int oy = r0; oy < r1; ++oy) {
            const m1f4 g = *reinterpret_cast<const m1f4*>(dyn + (long)oy * W + xq);
#pragma unroll
            for (int c = 0; c < M1W_CH; ++c) {
                const bool cok = c0 + c < C;
                const float* xc = x + ((long)n * C + (cok ? c0 + c : 0)) * hw;
                m1_row6(xc + (long)(oy + 1) * W, xq, W, cok && oy + 1 < H, ring[c][2]);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int p = 0; p < 4; ++p) acc[c][kh * 3 + kw] += g[p] * ring[c][kh][p + kw];
#pragma unroll
                for (int q = 0; q < 6; ++q) { ring[c][0][q] = ring[c][1][q]; ring[c][1][q] = ring[c][2][q]; }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < M1W_CH; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float s = wave_sum(acc[c][k]);
            if (lane == c * 9 + k && c0 + c < C) atomicAdd(dw + (long)(c0 + c) * 9 + k, s);
        }
}

// y[N,1,H,W] = act(conv(x[N,C,H,W], w[1,C,KH,KW], stride 1, zero padding `pad` with 2*pad = K-1) + bias)
int launch_conv_m1_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int KH, int KW, int pad,
                       int act, float slope, hipStream_t stream) {
    if (KH == 3 && KW == 3 && pad == 1 && (W & 3) == 0 && N > 0 && H > 0) {
        const long total = (long)N * H * (W >> 2);
        hipLaunchKernelGGL(conv_m1_fwd3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, w, bias, y, N, C, H, W, act, slope);
        return check_launch("conv_m1_fwd3");
    }
    const int PH = M1_TH + KH - 1, PW = M1_TW + KW - 1;
    const size_t lds = ((size_t)M1_KC * PH * PW + (size_t)M1_KC * KH * KW) * 4;
    const long blocks = (long)N * ((W + M1_TW - 1) / M1_TW) * ((H + M1_TH - 1) / M1_TH);
    if (blocks <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(conv_m1_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, x, w, bias, y, N, C, H, W, KH, KW, pad,
                       act, slope);
    return check_launch("conv_m1_fwd");
}

// dw[1,C,KH,KW] (+)= correlation of dy[N,1,H,W] with x[N,C,H,W]
int launch_conv_m1_wgrad(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int KH, int KW, int pad, int accumulate,
                         hipStream_t st) {
    const int T = KH * KW;
    if (!accumulate && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)C * T, st) != hipSuccess) return fail(FAOCTASR_EHIP, "memset dw failed");
    if (KH == 3 && KW == 3 && pad == 1 && (W & 3) == 0 && N > 0 && H > 0) {
        const long items = (long)N * ((H + M1W_ROWS - 1) / M1W_ROWS) * ((C + M1W_CH - 1) / M1W_CH);
        hipLaunchKernelGGL(conv_m1_wgrad3_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, x, dy, dw, N, C, H, W);
        return check_launch("conv_m1_wgrad3");
    }
    const int CCH = 256 / T;                                           // channels per block: one thread per (channel, tap)
    const int PH = M1_TH + KH - 1, PW = M1_TW + KW - 1;
    const size_t lds = ((size_t)CCH * PH * PW + M1_TH * M1_TW) * 4;
    const int gy = (C + CCH - 1) / CCH;
    const long ntiles = (long)N * ((W + M1_TW - 1) / M1_TW) * ((H + M1_TH - 1) / M1_TH);
    if (ntiles <= 0) return FAOCTASR_OK;
    long gx = 1024 / gy;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    const int tpb = (int)((ntiles + gx - 1) / gx);
    gx = (ntiles + tpb - 1) / tpb;
    auto k = conv_m1_wgrad_kernel;
    lds_optin((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)gx, gy), dim3(256), lds, st, x, dy, dw, N, C, H, W, KH, KW, pad, tpb, CCH);
    return check_launch("conv_m1_wgrad");
}

}  // namespace faoctasr
