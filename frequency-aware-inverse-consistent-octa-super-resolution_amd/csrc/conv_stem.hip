// Input gradient and weight gradient of the 4x4 stride-2 pad-1 STEM convolutions with 1..4 input channels (the discriminators'
// first layers on the image / on the three Haar detail bands, model.py; the generators' 1 -> 128 entry) for gfx950.
//
// On the MFMA kernels these shapes are almost all padding: the input gradient has C <= 4 output channels in a 64-row tile (1.7 TF,
// 159 us per call at batch 8) and the weight gradient a reduction of depth C*16 <= 64 (3.5 TF, 76 us) -- 2.5 ms per step for
// 0.3 GFLOP per call.  Both are a few hundred MFLOP over a tensor that is read once, i.e. HBM-bound VALU work:
//   dgrad : dx[n][c][iy][ix] = sum_m sum_{kh,kw} dy[n][m][oy][ox] * w[m][c][kh][kw],   iy = 2 oy - 1 + kh, ix = 2 ox - 1 + kw
//           a thread owns the 2x2 block (2a..2a+1, 2b..2b+1): it needs dy[a-1..a+1][b-1..b+1] and uses each of the 16 taps once
//   wgrad : dw[m][c][kh][kw] += sum_{n,oy,ox} dy[n][m][oy][ox] * x[n][c][2 oy - 1 + kh][2 ox - 1 + kw]
//           a thread owns a dy pixel, a block MG output channels: 16 C patch values are loaded once for MG * 16 C FMAs
#include "common.h"

namespace faoctasr {

constexpr int ST_TH = 8, ST_TW = 32;           // dy pixels per block: 8 rows x 32 columns (one per thread)

// wave-wide sum into lane 63 with six DPP adds (row_shr 1/2/4/8: inclusive scan of each 16-lane row; row_bcast 15 / 31: the row
// totals carried into the next rows).  __shfl_xor compiles to ds_bpermute + a full lgkmcnt wait per step, which made a 128-value
// epilogue cost more than the kernel's main loop.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}

template <int C>
__global__ __launch_bounds__(256) void stem_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N,
                                                         int M, int OH, int OW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Wl = reinterpret_cast<float*>(smem);                       // [M][C][16]
    const int tid = threadIdx.x;
    for (int i = tid; i < M * C * 16; i += 256) Wl[i] = w[i];
    __syncthreads();
    const int tiles_x = (OW + ST_TW - 1) / ST_TW, tiles_y = (OH + ST_TH - 1) / ST_TH;
    const int n = blockIdx.x / (tiles_x * tiles_y);
    const int r = blockIdx.x - n * tiles_x * tiles_y;
    const int a = (r / tiles_x) * ST_TH + (tid >> 5), b = (r % tiles_x) * ST_TW + (tid & 31);
    if (a >= OH || b >= OW) return;
    const long ohw = (long)OH * OW;
    const float* dyn = dy + (long)n * M * ohw;
    // validity of the 3x3 neighbourhood (zero outside the map)
    const bool ru = a > 0, rd = a + 1 < OH, cl = b > 0, cr = b + 1 < OW;
    const long o11 = (long)a * OW + b;
    float acc[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[c][q] = 0.f;
#pragma unroll 4
    for (int m = 0; m < M; ++m) {                                     // four channels' nine loads in flight per thread
        const float* p = dyn + (long)m * ohw + o11;
        const float d11 = p[0];
        const float d10 = cl ? p[-1] : 0.f, d12 = cr ? p[1] : 0.f;
        const float d01 = ru ? p[-OW] : 0.f, d21 = rd ? p[OW] : 0.f;
        const float d00 = (ru && cl) ? p[-OW - 1] : 0.f, d02 = (ru && cr) ? p[-OW + 1] : 0.f;
        const float d20 = (rd && cl) ? p[OW - 1] : 0.f, d22 = (rd && cr) ? p[OW + 1] : 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* k = Wl + (m * C + c) * 16;                   // k[kh*4 + kw], the same address for every lane
            // (2a, 2b): kh in {1,3} <-> oy in {a, a-1}; kw in {1,3} <-> ox in {b, b-1}
            acc[c][0] += d11 * k[5] + d10 * k[7] + d01 * k[13] + d00 * k[15];
            // (2a, 2b+1): kw in {0,2} <-> ox in {b+1, b}
            acc[c][1] += d12 * k[4] + d11 * k[6] + d02 * k[12] + d01 * k[14];
            // (2a+1, 2b): kh in {0,2} <-> oy in {a+1, a}
            acc[c][2] += d21 * k[1] + d20 * k[3] + d11 * k[9] + d10 * k[11];
            // (2a+1, 2b+1)
            acc[c][3] += d22 * k[0] + d21 * k[2] + d12 * k[8] + d11 * k[10];
        }
    }
    const int IW = 2 * OW;
    const long ihw = 4 * ohw;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float* q = dx + ((long)n * C + c) * ihw + (long)(2 * a) * IW + 2 * b;
        *reinterpret_cast<float2*>(q) = make_float2(acc[c][0], acc[c][1]);
        *reinterpret_cast<float2*>(q + IW) = make_float2(acc[c][2], acc[c][3]);
    }
}

// MG output channels per block; the slab of dy tiles [t0, t1) of the block is walked tile by tile
template <int C, int MG>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N,
                                                         int M, int OH, int OW, int tiles_per_block) {
    __shared__ float red[4][MG * C * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (OW + ST_TW - 1) / ST_TW, tiles_y = (OH + ST_TH - 1) / ST_TH;
    const int ntiles = N * tiles_x * tiles_y;
    const int m0 = blockIdx.y * MG;
    const int IH = 2 * OH, IW = 2 * OW;
    const long ohw = (long)OH * OW, ihw = (long)IH * IW;
    float acc[MG][C][16];
#pragma unroll
    for (int j = 0; j < MG; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[j][c][t] = 0.f;
    int t1 = (blockIdx.x + 1) * tiles_per_block;
    t1 = t1 < ntiles ? t1 : ntiles;
    for (int tile = blockIdx.x * tiles_per_block; tile < t1; ++tile) {
        const int n = tile / (tiles_x * tiles_y);
        const int r = tile - n * tiles_x * tiles_y;
        const int oy = (r / tiles_x) * ST_TH + (tid >> 5), ox = (r % tiles_x) * ST_TW + (tid & 31);
        if (oy >= OH || ox >= OW) continue;
        float g[MG];
#pragma unroll
        for (int j = 0; j < MG; ++j) g[j] = m0 + j < M ? dy[((long)n * M + m0 + j) * ohw + (long)oy * OW + ox] : 0.f;
        const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;                 // only the first / last row and column can fall outside
        const bool rt = iy0 >= 0, rb = iy0 + 3 < IH, cl = ix0 >= 0, cr = ix0 + 3 < IW;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* p = x + ((long)n * C + c) * ihw + (long)iy0 * IW + ix0;
            float xp[16];
#pragma unroll
            for (int kh = 0; kh < 4; ++kh) {
                const bool rok = (kh > 0 || rt) && (kh < 3 || rb);
                const float2 mid = rok ? *reinterpret_cast<const float2*>(p + kh * IW + 1) : make_float2(0.f, 0.f);   // ix0+1 = 2 ox: 8-byte aligned
                xp[kh * 4 + 0] = (rok && cl) ? p[kh * IW] : 0.f;
                xp[kh * 4 + 1] = mid.x;
                xp[kh * 4 + 2] = mid.y;
                xp[kh * 4 + 3] = (rok && cr) ? p[kh * IW + 3] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < MG; ++j)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[j][c][t] += g[j] * xp[t];
        }
    }
    // wave reduction (DPP), then the four waves through LDS, then one atomic per value
#pragma unroll
    for (int j = 0; j < MG; ++j)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float v = wave_sum_to_lane63(acc[j][c][t]);
                if (lane == 63) red[wave][(j * C + c) * 16 + t] = v;
            }
    __syncthreads();
    for (int i = tid; i < MG * C * 16; i += 256) {
        const int j = i / (C * 16), ct = i - j * C * 16;
        if (m0 + j < M) atomicAdd(dw + (long)(m0 + j) * C * 16 + ct, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
    }
}

static bool stem_shape(int C, int IH, int IW, int KH, int KW, int stride, int pad, int reflect) {
    return C >= 1 && C <= 4 && KH == 4 && KW == 4 && stride == 2 && pad == 1 && !reflect && (IH & 1) == 0 && (IW & 1) == 0 && IH >= 4 && IW >= 4;
}

bool stem_wgrad_eligible(int C, int IH, int IW, int KH, int KW, int stride, int pad, int reflect) {
    return stem_shape(C, IH, IW, KH, KW, stride, pad, reflect);
}

bool stem_dgrad_eligible(int C, int IH, int IW, int M, int KH, int KW, int stride, int pad) {
    return stem_shape(C, IH, IW, KH, KW, stride, pad, 0) && (long)M * C * 64 <= 48 * 1024;
}

// 1 launched, 0 not this shape, < 0 error
int launch_stem_dgrad(const float* dy, const float* w, float* dx, int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                      hipStream_t s) {
    if (!stem_dgrad_eligible(C, IH, IW, M, KH, KW, stride, pad)) return 0;
    const int OH = IH / 2, OW = IW / 2;
    const long blocks = (long)N * ((OW + ST_TW - 1) / ST_TW) * ((OH + ST_TH - 1) / ST_TH);
    if (blocks <= 0) return 1;
    if (blocks > 0x7fffffffL) return 0;
    const size_t lds = (size_t)M * C * 64;
    auto go = [&](auto k) { hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, s, dy, w, dx, N, M, OH, OW); };
    if (C == 1) go(stem_dgrad_kernel<1>);
    else if (C == 2) go(stem_dgrad_kernel<2>);
    else if (C == 3) go(stem_dgrad_kernel<3>);
    else go(stem_dgrad_kernel<4>);
    const int rc = check_launch("stem_dgrad");
    return rc == FAOCTASR_OK ? 1 : rc;
}

int launch_stem_wgrad(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                      int reflect, int accumulate, hipStream_t s) {
    if (!stem_shape(C, IH, IW, KH, KW, stride, pad, reflect)) return 0;
    const int OH = IH / 2, OW = IW / 2;
    const long ntiles = (long)N * ((OW + ST_TW - 1) / ST_TW) * ((OH + ST_TH - 1) / ST_TH);
    if (ntiles > 0x7fffffffL) return 0;
    if (!accumulate && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)M * C * 16, s) != hipSuccess) return fail(FAOCTASR_EHIP, "memset dw failed");
    if (ntiles <= 0) return 1;
    const int MG = C == 1 ? 8 : C == 2 ? 4 : 2;
    const int gy = (M + MG - 1) / MG;
    // ~1024 blocks: enough to fill the chip, few enough that the epilogue (MG*C*16 atomics per block) stays small
    long slabs = 1024 / gy;
    slabs = slabs < 1 ? 1 : slabs;
    slabs = slabs > ntiles ? ntiles : slabs;
    const int tpb = (int)((ntiles + slabs - 1) / slabs);
    const dim3 grid((unsigned)((ntiles + tpb - 1) / tpb), (unsigned)gy);
    auto go = [&](auto k) { hipLaunchKernelGGL(k, grid, dim3(256), 0, s, x, dy, dw, N, M, OH, OW, tpb); };
    if (C == 1) go(stem_wgrad_kernel<1, 8>);
    else if (C == 2) go(stem_wgrad_kernel<2, 4>);
    else if (C == 3) go(stem_wgrad_kernel<3, 2>);
    else go(stem_wgrad_kernel<4, 2>);
    const int rc = check_launch("stem_wgrad");
    return rc == FAOCTASR_OK ? 1 : rc;
}

}  // namespace faoctasr
