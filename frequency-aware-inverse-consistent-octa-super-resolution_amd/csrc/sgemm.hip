// Batched row-major SGEMM on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), used for the
// circulant form of the reference's FFT Gaussian split (utils.py:93-117): the shifted mask is
// separable, so ifft2(mask * fft2(x)) == C_H x C_W^T with two n x n real circulants.
//   C_b[M,N] = A_b[M,K] * B_b[K,N],  b in [0,batch), arbitrary leading dimensions and batch strides
//   (stride 0 = operand shared by every batch entry).
// Block 256 threads, tile 64x64, K chunk 16; waves 2x2, one 32x32 accumulator each.
#include "common.h"

namespace faoctasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void sgemm_batched_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                            float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                                            long sA, long sB, long sC) {
    constexpr int KC = 16, LDA = KC + 1, NT = 64;
    __shared__ float A_s[64 * LDA];
    __shared__ float B_s[KC * NT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Ab = A + (long)blockIdx.z * sA;
    const float* Bb = B + (long)blockIdx.z * sB;
    float* Cb = C + (long)blockIdx.z * sC;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int ml = e >> 4, kk = e & 15;
            const int m = m0 + ml, k = k0 + kk;
            A_s[ml * LDA + kk] = (m < M && k < K) ? Ab[(long)m * lda + k] : 0.f;
            const int kb = e >> 6, nb = e & 63;
            const int kg = k0 + kb, n = n0 + nb;
            B_s[kb * NT + nb] = (kg < K && n < N) ? Bb[(long)kg * ldb + n] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk0 = 0; kk0 < KC; kk0 += 2) {
            const float af = A_s[(wm * 32 + l31) * LDA + kk0 + lh];
            const float bf = B_s[(kk0 + lh) * NT + wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
    }
    const int n = n0 + wn * 32 + l31;
    if (n < N) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int m = m0 + wm * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            if (m < M) Cb[(long)m * ldc + n] = acc[rr];
        }
    }
}


// The n x n circulant of the reference's Gaussian low-pass (utils.py:71-80 mask, 107-117 filter): the centred taps
// g[k] = exp(-(k - int(n/2))^2 / (2 r^2)) meet spectrum bin k at g[(k + n/2) mod n] (fftshift, mask, ifftshift), so the
// filter is multiplication by C[a][b] = c[(a - b) mod n], c = real(ifft(t)), t[k] = g[(k + n/2) mod n].  Built once per
// (n, radius) in double precision on the device; the caller owns and caches the matrix (the ABI never allocates).
__global__ __launch_bounds__(256) void circulant_lowpass_kernel(float* __restrict__ out, int n, double inv2r2) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)n * n) return;
    const int a = (int)(e / n), b = (int)(e % n);
    const int j = ((a - b) % n + n) % n;
    const int c0 = n / 2;
    double s = 0.0;
    for (int k = 0; k < n; ++k) {
        const int kk = (k + c0) % n;
        const double d = (double)(kk - c0);
        const double t = exp(-d * d * inv2r2);
        const long ph = ((long)k * j) % n;                    // exact phase reduction: cos(2 pi k j / n)
        s += t * cospi(2.0 * (double)ph / (double)n);
    }
    out[e] = (float)(s / (double)n);
}

}  // namespace faoctasr

using namespace faoctasr;

extern "C" int faoctasr_sgemm_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                      long strideA, long strideB, long strideC, int batch, faoctasr_stream_t stream) {
    if (!A || !B || !C) return fail(FAOCTASR_EINVAL, "sgemm_batched: null pointer");
    if (M < 0 || N < 0 || K < 0 || batch < 0 || lda < K || ldb < N || ldc < N) return fail(FAOCTASR_EINVAL, "sgemm_batched: bad shape");
    if (M == 0 || N == 0 || batch == 0) return FAOCTASR_OK;
    if (batch > 65535) return fail(FAOCTASR_EUNSUPPORTED, "sgemm_batched: batch %d > 65535", batch);
    dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
    hipLaunchKernelGGL(sgemm_batched_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, lda, ldb, ldc, strideA, strideB,
                       strideC);
    return check_launch("sgemm_batched");
}

extern "C" int faoctasr_circulant_lowpass(float* out, int n, float radius, faoctasr_stream_t stream) {
    if (!out) return fail(FAOCTASR_EINVAL, "circulant_lowpass: null pointer");
    if (n < 1 || n > 8192 || !(radius > 0.f)) return fail(FAOCTASR_EINVAL, "circulant_lowpass: n %d, radius %g", n, (double)radius);
    const long total = (long)n * n;
    hipLaunchKernelGGL(circulant_lowpass_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                       0.5 / ((double)radius * (double)radius));
    return check_launch("circulant_lowpass");
}
