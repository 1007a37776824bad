// Evaluation metrics of the reference's test loop on the device (utils.py:182-242 `eval` / `eval_6m`):
//   skimage.metrics.peak_signal_noise_ratio(y, gt, data_range=2)     10 log10(R^2 / MSE)
//   skimage.metrics.structural_similarity(y, gt)                     7x7 uniform window, K1 .01, K2 .03, sample covariance,
//                                                                    mean over the map cropped by 3 (float images: data range 2)
//   skimage.metrics.mean_squared_error(y, gt)
//   skimage.metrics.normalized_mutual_information(y, gt)             (H(y) + H(gt)) / H(y, gt) on the joint 100 x 100 histogram
//                                                                    over [min, max] of each image (numpy.histogram2d semantics)
// The reference copies every output to the host and scores it with skimage; here the super-resolved image never leaves the GPU:
// four small kernels per batch of image pairs, results as doubles [N][4] = {psnr, ssim, mse, nmi}.  All arithmetic is fp64; histogram bin edges follow numpy.linspace in fp64 so that bin membership is numpy's.
// BatchNorm folding for the inference forward (conv weights scaled by gamma / sqrt(var + eps)) lives here too.
#include "common.h"

namespace faoctasr {

constexpr int EV_P = 64;                  // partial min/max blocks per image
constexpr int EV_WIN = 7, EV_PAD = 3;

__device__ __forceinline__ double block_sum_256d(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// partial min / max of both images: mm[n][p][4] = {ymin, ymax, gmin, gmax}
__global__ __launch_bounds__(256) void eval_minmax_kernel(const float* __restrict__ y, const float* __restrict__ g, float* __restrict__ mm, long HW) {
    __shared__ float red[4][4];
    const int n = blockIdx.y, p = blockIdx.x;
    const long per = (HW + EV_P - 1) / EV_P;
    long e0 = (long)p * per, e1 = e0 + per;
    e1 = e1 < HW ? e1 : HW;
    const float* yp = y + (long)n * HW;
    const float* gp = g + (long)n * HW;
    float v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const float a = yp[e], b = gp[e];
        v[0] = fminf(v[0], a); v[1] = fmaxf(v[1], a); v[2] = fminf(v[2], b); v[3] = fmaxf(v[3], b);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v[0] = fminf(v[0], __shfl_xor(v[0], o, 64)); v[1] = fmaxf(v[1], __shfl_xor(v[1], o, 64));
        v[2] = fminf(v[2], __shfl_xor(v[2], o, 64)); v[3] = fmaxf(v[3], __shfl_xor(v[3], o, 64));
    }
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < 4; ++q) red[threadIdx.x >> 6][q] = v[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = mm + ((long)n * EV_P + p) * 4;
        o[0] = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
        o[1] = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
        o[2] = fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2]));
        o[3] = fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3]));
    }
}

// squared error over the image and the 7x7 uniform-window SSIM map over its valid region: sums[n][2] += {sum (y-g)^2, sum S}
__global__ __launch_bounds__(256) void eval_sqerr_ssim_kernel(const float* __restrict__ y, const float* __restrict__ g, double* __restrict__ sums,
                                                              int H, int W, float data_range) {
    constexpr int T = 32, PW = T + 2 * EV_PAD;
    __shared__ float ys[PW * PW], gs[PW * PW];
    __shared__ double red[4];
    const int n = blockIdx.z, y0 = blockIdx.y * T, x0 = blockIdx.x * T;
    const float* yp = y + (long)n * H * W;
    const float* gp = g + (long)n * H * W;
    for (int i = threadIdx.x; i < PW * PW; i += 256) {
        const int r = i / PW, c = i - r * PW;
        const int yy = y0 + r - EV_PAD, xx = x0 + c - EV_PAD;
        const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        ys[i] = in ? yp[(long)yy * W + xx] : 0.f;
        gs[i] = in ? gp[(long)yy * W + xx] : 0.f;
    }
    __syncthreads();
    // fp64 throughout (the kernel is nowhere near a bottleneck): in fp32 the variances of flat regions are rounding noise that is
    // not small against C2 = (0.03 R)^2 (two constant images: 2e-4 relative on the result)
    const double C1 = (0.01 * (double)data_range) * (0.01 * (double)data_range), C2 = (0.03 * (double)data_range) * (0.03 * (double)data_range);
    const double inv = 1.0 / (double)(EV_WIN * EV_WIN), cov_norm = (double)(EV_WIN * EV_WIN) / (double)(EV_WIN * EV_WIN - 1);
    double se = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < T * T; i += 256) {
        const int r = i / T, c = i - r * T;
        const int yy = y0 + r, xx = x0 + c;
        if (yy >= H || xx >= W) continue;
        const double d = (double)ys[(r + EV_PAD) * PW + c + EV_PAD] - (double)gs[(r + EV_PAD) * PW + c + EV_PAD];
        se += d * d;
        if (yy < EV_PAD || yy >= H - EV_PAD || xx < EV_PAD || xx >= W - EV_PAD) continue;      // skimage crops the map by (win-1)/2
        double sa = 0.0, sb = 0.0, saa = 0.0, sbb = 0.0, sab = 0.0;
#pragma unroll
        for (int u = 0; u < EV_WIN; ++u)
#pragma unroll
            for (int v = 0; v < EV_WIN; ++v) {
                const double a = (double)ys[(r + u) * PW + c + v], b = (double)gs[(r + u) * PW + c + v];
                sa += a; sb += b; saa += a * a; sbb += b * b; sab += a * b;
            }
        const double ux = sa * inv, uy = sb * inv;
        const double vx = cov_norm * (saa * inv - ux * ux), vy = cov_norm * (sbb * inv - uy * uy), vxy = cov_norm * (sab * inv - ux * uy);
        ss += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
    se = block_sum_256d(se, red);
    ss = block_sum_256d(ss, red);
    if (threadIdx.x == 0) {
        atomicAdd(sums + 2 * n, se);
        atomicAdd(sums + 2 * n + 1, ss);
    }
}

struct Range { double lo, hi; };
__device__ __forceinline__ Range hist_range(float mn, float mx) {
    Range r{(double)mn, (double)mx};
    if (r.lo == r.hi) { r.lo -= 0.5; r.hi += 0.5; }                       // numpy: a degenerate range is widened by +-0.5
    return r;
}
// numpy.histogramdd: edges = linspace(lo, hi, bins + 1); bin = searchsorted(edges, v, 'right') - 1, v == hi goes to the last bin
__device__ __forceinline__ int hist_bin(float vf, const Range& r, int bins) {
    const double v = (double)vf, step = (r.hi - r.lo) / (double)bins;
    int k = (int)floor((v - r.lo) / step);
    k = k < 0 ? 0 : (k > bins - 1 ? bins - 1 : k);
    auto edge = [&](int j) { return j == bins ? r.hi : r.lo + (double)j * step; };
    while (k > 0 && v < edge(k)) --k;
    while (k < bins - 1 && v >= edge(k + 1)) ++k;
    return k;
}

__global__ __launch_bounds__(256) void eval_hist2d_kernel(const float* __restrict__ y, const float* __restrict__ g, const float* __restrict__ mm,
                                                          unsigned* __restrict__ hist, long HW, int bins) {
    __shared__ Range ry, rg;
    const int n = blockIdx.y, p = blockIdx.x;
    if (threadIdx.x == 0) {
        float v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
        for (int q = 0; q < EV_P; ++q) {
            const float* o = mm + ((long)n * EV_P + q) * 4;
            v[0] = fminf(v[0], o[0]); v[1] = fmaxf(v[1], o[1]); v[2] = fminf(v[2], o[2]); v[3] = fmaxf(v[3], o[3]);
        }
        ry = hist_range(v[0], v[1]);
        rg = hist_range(v[2], v[3]);
    }
    __syncthreads();
    const long per = (HW + EV_P - 1) / EV_P;
    long e0 = (long)p * per, e1 = e0 + per;
    e1 = e1 < HW ? e1 : HW;
    unsigned* h = hist + (long)n * bins * bins;
    for (long e = e0 + threadIdx.x; e < e1; e += 256)
        atomicAdd(h + hist_bin(y[(long)n * HW + e], ry, bins) * bins + hist_bin(g[(long)n * HW + e], rg, bins), 1u);
}

// one block per image: entropies of the joint histogram and its marginals, and the four results
__global__ __launch_bounds__(256) void eval_finish_kernel(const unsigned* __restrict__ hist, const double* __restrict__ sums, double* __restrict__ out,
                                                          int H, int W, int bins, float data_range) {
    extern __shared__ double marg[];                                      // [2][bins]
    __shared__ double red[4];
    const int n = blockIdx.x;
    const unsigned* h = hist + (long)n * bins * bins;
    const double total = (double)H * (double)W;
    for (int i = threadIdx.x; i < 2 * bins; i += 256) marg[i] = 0.0;
    __syncthreads();
    double hj = 0.0;
    for (int i = threadIdx.x; i < bins * bins; i += 256) {
        const unsigned c = h[i];
        if (c) {
            const double p = (double)c / total;
            hj -= p * log(p);
            atomicAdd(&marg[i / bins], (double)c);
            atomicAdd(&marg[bins + i % bins], (double)c);
        }
    }
    hj = block_sum_256d(hj, red);
    __syncthreads();
    double ha = 0.0, hb = 0.0;
    for (int i = threadIdx.x; i < bins; i += 256) {
        const double a = marg[i] / total, b = marg[bins + i] / total;
        if (a > 0.0) ha -= a * log(a);
        if (b > 0.0) hb -= b * log(b);
    }
    ha = block_sum_256d(ha, red);
    hb = block_sum_256d(hb, red);
    if (threadIdx.x == 0) {
        const double mse = sums[2 * n] / total;
        const double nvalid = (double)(H - 2 * EV_PAD) * (double)(W - 2 * EV_PAD);
        double* o = out + 4 * n;
        o[0] = mse > 0.0 ? 10.0 * log10((double)data_range * (double)data_range / mse) : INFINITY;
        o[1] = sums[2 * n + 1] / nvalid;
        o[2] = mse;
        o[3] = hj > 0.0 ? (ha + hb) / hj : 1.0;                              // two constant images: skimage returns 1
    }
}

// conv weights with an eval-mode BatchNorm folded in: w'[m][k] = w[m][k] * s[m], b'[m] = (bias[m] - mean[m]) * s[m] + beta[m],
// s = gamma / sqrt(var + eps).  transposed != 0: w is [K0][M][K1] (ConvTranspose2d: [C][M][kh*kw]) and m is the middle index.
__global__ void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var, float eps,
                               float* __restrict__ wf, float* __restrict__ bf, int M, long K0, long K1) {
    const long total = K0 * M * K1;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int m = (int)((i / K1) % M);
        wf[i] = w[i] * ((gamma ? gamma[m] : 1.f) / sqrtf(var[m] + eps));
    }
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += stride) {
        const float s = (gamma ? gamma[m] : 1.f) / sqrtf(var[m] + eps);
        bf[m] = ((bias ? bias[m] : 0.f) - mean[m]) * s + (beta ? beta[m] : 0.f);
    }
}

}  // namespace faoctasr

using namespace faoctasr;

extern "C" {

long faoctasr_eval_workspace_bytes(int N, int bins) {
    if (N <= 0 || bins <= 0) return 0;
    return (long)N * EV_P * 4 * sizeof(float) + (long)N * 2 * sizeof(double) + (long)N * bins * bins * sizeof(unsigned) + 64;
}

// y, gt: [N][H][W] fp32 device images (single channel); out: [N][4] doubles {psnr, ssim, mse, nmi}; workspace:
// faoctasr_eval_workspace_bytes(N, bins) bytes, 8-byte aligned.  bins = 100 and data_range = 2 reproduce utils.py:209-212.
int faoctasr_eval_metrics(const float* y, const float* gt, double* out, void* workspace, int N, int H, int W, float data_range, int bins,
                          faoctasr_stream_t stream) {
    if (!y || !gt || !out || !workspace) return fail(FAOCTASR_EINVAL, "eval_metrics: null pointer");
    if (N <= 0 || H < EV_WIN || W < EV_WIN || bins < 1 || bins > 1024 || !(data_range > 0.f))
        return fail(FAOCTASR_EINVAL, "eval_metrics: bad shape (N %d, %d x %d, bins %d)", N, H, W, bins);
    if (((size_t)workspace & 7) != 0) return fail(FAOCTASR_EINVAL, "eval_metrics: workspace must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    double* sums = reinterpret_cast<double*>(workspace);
    float* mm = reinterpret_cast<float*>(sums + 2L * N);
    unsigned* hist = reinterpret_cast<unsigned*>(mm + (long)N * EV_P * 4);
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * N, st) != hipSuccess || hipMemsetAsync(hist, 0, sizeof(unsigned) * (size_t)N * bins * bins, st) != hipSuccess)
        return fail(FAOCTASR_EHIP, "eval_metrics: memset failed");
    hipLaunchKernelGGL(eval_minmax_kernel, dim3(EV_P, N), dim3(256), 0, st, y, gt, mm, HW);
    hipLaunchKernelGGL(eval_sqerr_ssim_kernel, dim3((W + 31) / 32, (H + 31) / 32, N), dim3(256), 0, st, y, gt, sums, H, W, data_range);
    hipLaunchKernelGGL(eval_hist2d_kernel, dim3(EV_P, N), dim3(256), 0, st, y, gt, mm, hist, HW, bins);
    hipLaunchKernelGGL(eval_finish_kernel, dim3(N), dim3(256), 2 * bins * sizeof(double), st, hist, sums, out, H, W, bins, data_range);
    return check_launch("eval_metrics");
}

// Eval-mode BatchNorm folded into the preceding convolution (inference forward, utils.py:186 `model.eval()`).
// w: [M][K] (Conv2d, transposed = 0, K = C*kh*kw) or [C][M][kh*kw] (ConvTranspose2d, transposed = 1, K = kh*kw and K0 = C).
int faoctasr_bn_fold(const float* w, const float* bias, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, float* w_folded, float* bias_folded, int M, long K, int transposed, long K0,
                     faoctasr_stream_t stream) {
    if (!w || !running_mean || !running_var || !w_folded || !bias_folded) return fail(FAOCTASR_EINVAL, "bn_fold: null pointer");
    if (M <= 0 || K <= 0 || (transposed && K0 <= 0)) return fail(FAOCTASR_EINVAL, "bn_fold: bad shape");
    const long k0 = transposed ? K0 : 1;
    const long total = k0 * M * K;
    long blocks = (total + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, bias, gamma, beta, running_mean, running_var,
                       eps, w_folded, bias_folded, M, k0, K);
    return check_launch("bn_fold");
}

}  // extern "C"
