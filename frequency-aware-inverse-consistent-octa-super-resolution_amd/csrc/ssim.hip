// Fused SSIM (ssim.py:17-37) for gfx950: the 11x11 sigma-1.5 window is the outer product of the
// 1-D taps (ssim.py:11-13), so all five moment maps (mu1, mu2, E[a^2], E[b^2], E[ab]) are produced
// by one row pass + one column pass through LDS, followed by the pointwise map and a block
// reduction; zero padding of 5 as F.conv2d(padding=window_size//2).  HBM traffic is the
// algorithmic minimum (read a, b once per tile + halo; write one float per image).
// Backward recomputes the moments on a halo, forms the five partial-derivative maps, and applies
// the same separable filter to them (the window is symmetric, so the adjoint of the zero-padded
// correlation is itself).
#include "common.h"

namespace faoctasr {

struct Taps { float g[11]; };

constexpr int R = 5;                 // window radius
constexpr float C1 = 0.01f * 0.01f;  // ssim.py:29-30
constexpr float C2 = 0.03f * 0.03f;

__device__ __forceinline__ float ssim_point(float m1, float m2, float e11, float e22, float e12) {
    const float s11 = e11 - m1 * m1, s22 = e22 - m2 * m2, s12 = e12 - m1 * m2;
    return ((2.f * m1 * m2 + C1) * (2.f * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s11 + s22 + C2));
}

// forward: tile 22 x 64 outputs, one (n,c) plane per blockIdx.z.  The kernel is VALU-bound, not HBM-bound (2 x 11 taps x 5
// moments = ~135 FMA per pixel against 8 bytes), so it is organised around the packed fp32 pipe: the two images are staged
// interleaved as (a, b) pairs and the moments travel as the pairs (mu_a, mu_b), (E a^2, E b^2) plus E ab -- three
// v_pk_fma_f32 / v_fma_f32 per tap instead of five -- and both filter passes are register-blocked along the filter direction
// (4 row-pass outputs from 14 staged pairs, 3 column-pass outputs from 13 row-filtered values per thread; 512 threads), which also cuts the
// LDS reads per pixel from ~90 to ~30.  Tap order per output is unchanged from the plain form.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int SF_TY = 22, SF_TX = 64, SF_PY = SF_TY + 2 * R, SF_PX = 74;      // 32 staged rows of 74 pairs

__global__ __launch_bounds__(512) void ssim_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ sums, int C, int H, int W, const Taps tp) {
    __shared__ __attribute__((aligned(16))) f2 ab[SF_PY * SF_PX];                  // (a, b)
    __shared__ __attribute__((aligned(16))) f2 hz01[SF_PY * SF_TX], hz23[SF_PY * SF_TX];   // (mu_a, mu_b), (E a^2, E b^2) row-filtered
    __shared__ float hz4[SF_PY * SF_TX];                                          // E ab row-filtered
    __shared__ float red[8];
    const int plane = blockIdx.z;
    const int y0 = blockIdx.y * SF_TY, x0 = blockIdx.x * SF_TX;
    const float* ap = a + (long)plane * H * W;
    const float* bp = b + (long)plane * H * W;
    {   // 4 rows per pass, 128 threads per row (74 used); 60 KiB of LDS allow two blocks per CU, hence 512 threads each
        const int c = threadIdx.x & 127, rbase = threadIdx.x >> 7;
        const int xx = x0 + c - R;
        const bool cin = c < SF_PX && (unsigned)xx < (unsigned)W;
#pragma unroll
        for (int r = rbase; r < SF_PY; r += 4) {
            const int yy = y0 + r - R;
            const bool in = cin && (unsigned)yy < (unsigned)H;
            if (c < SF_PX) ab[r * SF_PX + c] = in ? f2{ap[(long)yy * W + xx], bp[(long)yy * W + xx]} : f2{0.f, 0.f};
        }
    }
    __syncthreads();
    {   // row pass: thread = (row r, 4 output columns from c0)
        const int r = threadIdx.x >> 4, c0 = (threadIdx.x & 15) * 4;
        f2 v[14];
#pragma unroll
        for (int q = 0; q < 14; ++q) v[q] = ab[r * SF_PX + c0 + q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
            float s4 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const f2 x = v[j + k];
                const float g = tp.g[k];
                const f2 gx = g * x;
                s01 += gx;
                s23 += gx * x;
                s4 += gx[0] * x[1];
            }
            hz01[r * SF_TX + c0 + j] = s01;
            hz23[r * SF_TX + c0 + j] = s23;
            hz4[r * SF_TX + c0 + j] = s4;
        }
    }
    __syncthreads();
    float local = 0.f;
    {   // column pass: thread = (column c, 3 output rows from r0)
        const int c = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * 3;
        f2 m01[3], m23[3];
        float m4[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { m01[j] = f2{0.f, 0.f}; m23[j] = f2{0.f, 0.f}; m4[j] = 0.f; }
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            int rr = r0 + i;
            rr = rr < SF_PY ? rr : SF_PY - 1;                          // rows past the tile feed only masked outputs
            const f2 v01 = hz01[rr * SF_TX + c], v23 = hz23[rr * SF_TX + c];
            const float v4 = hz4[rr * SF_TX + c];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (i - j >= 0 && i - j < 11) {
                    const float g = tp.g[i - j];
                    m01[j] += g * v01;
                    m23[j] += g * v23;
                    m4[j] += g * v4;
                }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (r0 + j < SF_TY && y0 + r0 + j < H && x0 + c < W) local += ssim_point(m01[j][0], m01[j][1], m23[j][0], m23[j][1], m4[j]);
    }
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sums + plane / C, ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])));
}

// backward: tile 16 x 32 outputs
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ gout, int gN, float gscale, float* __restrict__ da,
                                                       float* __restrict__ db, int C, int H, int W, const Taps tp) {
    constexpr int TY = 16, TX = 32;
    constexpr int QY = TY + 2 * R, QX = TX + 2 * R;       // region where the ssim map's partials are needed
    constexpr int PY = QY + 2 * R, PX = QX + 2 * R;       // input region
    __shared__ float as[PY * PX], bs[PY * PX];
    __shared__ float hz[5][PY * QX];                      // row-filtered moments; later reused for row-filtered partials
    __shared__ float pm[5][QY * QX];                      // partial-derivative maps f_m1, f_m2, f_e11, f_e22, f_e12
    const int plane = blockIdx.z;
    const int y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
    const float* ap = a + (long)plane * H * W;
    const float* bp = b + (long)plane * H * W;
    for (int i = threadIdx.x; i < PY * PX; i += 256) {
        const int r = i / PX, c = i - r * PX;
        const int yy = y0 + r - 2 * R, xx = x0 + c - 2 * R;
        const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        as[i] = in ? ap[(long)yy * W + xx] : 0.f;
        bs[i] = in ? bp[(long)yy * W + xx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PY * QX; i += 256) {
        const int r = i / QX, c = i - r * QX;
        float sa = 0.f, sb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float av = as[r * PX + c + k], bv = bs[r * PX + c + k], g = tp.g[k];
            sa += g * av; sb += g * bv; saa += g * av * av; sbb += g * bv * bv; sab += g * av * bv;
        }
        hz[0][i] = sa; hz[1][i] = sb; hz[2][i] = saa; hz[3][i] = sbb; hz[4][i] = sab;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < QY * QX; i += 256) {
        const int r = i / QX, c = i - r * QX;
        const int yy = y0 + r - R, xx = x0 + c - R;
        float f[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float g = tp.g[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] += g * hz[q][(r + k) * QX + c];
            }
            const float m1 = m[0], m2 = m[1];
            const float s11 = m[2] - m1 * m1, s22 = m[3] - m2 * m2, s12 = m[4] - m1 * m2;
            const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * s12 + C2, B1 = m1 * m1 + m2 * m2 + C1, B2 = s11 + s22 + C2;
            const float inv = 1.f / (B1 * B2);
            const float S = A1 * A2 * inv;
            f[0] = (2.f * m2 * (A2 - A1)) * inv - S * (2.f * m1 / B1 - 2.f * m1 / B2);
            f[1] = (2.f * m1 * (A2 - A1)) * inv - S * (2.f * m2 / B1 - 2.f * m2 / B2);
            f[2] = -S / B2;
            f[3] = -S / B2;
            f[4] = 2.f * A1 * inv;
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) pm[q][i] = f[q];
    }
    __syncthreads();
    // row pass over the partial maps: rows QY, output columns TX (reuses hz storage, row stride TX)
    for (int i = threadIdx.x; i < QY * TX; i += 256) {
        const int r = i / TX, c = i - r * TX;
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = tp.g[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) s[q] += g * pm[q][r * QX + c + k];
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) hz[q][i] = s[q];
    }
    __syncthreads();
    const float gv = gout[gN > 1 ? plane / C : 0] * gscale;
    for (int i = threadIdx.x; i < TY * TX; i += 256) {
        const int r = i / TX, c = i - r * TX;
        const int yy = y0 + r, xx = x0 + c;
        if (yy < H && xx < W) {
            float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float g = tp.g[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) s[q] += g * hz[q][(r + k) * TX + c];
            }
            const float av = as[(r + 2 * R) * PX + c + 2 * R], bv = bs[(r + 2 * R) * PX + c + 2 * R];
            const long off = (long)plane * H * W + (long)yy * W + xx;
            if (da) da[off] = gv * (s[0] + 2.f * av * s[2] + bv * s[4]);
            if (db) db[off] = gv * (s[1] + 2.f * bv * s[3] + av * s[4]);
        }
    }
}

static Taps make_taps() {
    // ssim.py:7-9: gauss = Tensor([exp(-(x - 5)^2 / (2 * 1.5^2))]) (fp32) / gauss.sum()
    Taps t;
    float sum = 0.f;
    for (int k = 0; k < 11; ++k) {
        t.g[k] = (float)exp(-(double)((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5));
        sum += t.g[k];
    }
    for (int k = 0; k < 11; ++k) t.g[k] /= sum;
    return t;
}

}  // namespace faoctasr

using namespace faoctasr;

extern "C" {

int faoctasr_ssim_fwd(const float* a, const float* b, float* sums, int N, int C, int H, int W, faoctasr_stream_t stream) {
    if (!a || !b || !sums) return fail(FAOCTASR_EINVAL, "ssim_fwd: null pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(FAOCTASR_EINVAL, "ssim_fwd: bad shape");
    if ((long)N * C > 65535) return fail(FAOCTASR_EUNSUPPORTED, "ssim_fwd: more than 65535 planes");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, sizeof(float) * N, st) != hipSuccess) return fail(FAOCTASR_EHIP, "ssim_fwd: memset failed");
    dim3 grid((W + SF_TX - 1) / SF_TX, (H + SF_TY - 1) / SF_TY, N * C);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(512), 0, st, a, b, sums, C, H, W, make_taps());
    return check_launch("ssim_fwd");
}

int faoctasr_ssim_bwd(const float* a, const float* b, const float* g, int gN, float gscale, float* da, float* db, int N, int C, int H,
                      int W, faoctasr_stream_t stream) {
    if (!a || !b || !g) return fail(FAOCTASR_EINVAL, "ssim_bwd: null pointer");
    if (gN != 1 && gN != N) return fail(FAOCTASR_EINVAL, "ssim_bwd: gN must be 1 or N");
    if ((long)N * C > 65535) return fail(FAOCTASR_EUNSUPPORTED, "ssim_bwd: more than 65535 planes");
    if (!da && !db) return FAOCTASR_OK;
    dim3 grid((W + 31) / 32, (H + 15) / 16, N * C);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, g, gN, gscale, da, db, C, H, W, make_taps());
    return check_launch("ssim_bwd");
}

}  // extern "C"
