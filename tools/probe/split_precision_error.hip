// How exact is a split-precision contraction on the bf16 / f16 matrix cores, next to the exact-f32 MFMA?  
// One 32 x 32 output tile per wave, C = A[32][K] * B[K][32], operands shaped like the model's (weights N(0, 0.02), activations
// post-BatchNorm / ReLU, gradients ~1e-4 with a heavy tail); each arithmetic is compared with an fp64 host reference.
//   f32 fma    : sequential fmaf chain on the host (what a CPU reference does per output)
//   f32 mfma   : v_mfma_f32_32x32x2_f32 (the shipped "f32" kernels)
//   bf16x3     : x = hi + lo, 3 products (shipped opt-in path)
//   bf16x6     : x = hi + mid + lo (exact: 3 x 8 bits), the 6 products of order <= 2^-16; dropped: mid*lo, lo*mid, lo*lo (2^-24)
//   bf16x6s    : same, small terms accumulated first
//   f16x2      : x*s = hi + lo in f16 (s = power of two from the tensor's max), 3 products, result / (sa sb)
// build + run: hipcc -O3 --offload-arch=gfx950 tools/probe/split_precision_error.hip -o /tmp/spe && /tmp/spe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A: [32][K] row-major, B: [K][32] row-major (K multiple of 16), C: [32][32]
template <int KIND>
__global__ __launch_bounds__(64) void gemm_tile(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K, float sa, float sb) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    A += (size_t)blockIdx.x * 32 * K; B += (size_t)blockIdx.x * 32 * K; C += (size_t)blockIdx.x * 1024;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (KIND == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + lh], B[(k + lh) * 32 + l31], acc, 0, 0, 0);
    } else if constexpr (KIND == 4) {
        for (int k0 = 0; k0 < K; k0 += 16) {
            f16x8 ah, al, bh, bl;
            for (int j = 0; j < 8; ++j) {
                const float a = A[l31 * K + k0 + 8 * lh + j] * sa, b = B[(k0 + 8 * lh + j) * 32 + l31] * sb;
                ah[j] = (_Float16)a; al[j] = (_Float16)(a - (float)ah[j]);
                bh[j] = (_Float16)b; bl[j] = (_Float16)(b - (float)bh[j]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
        const float inv = 1.0f / (sa * sb);
        for (int r = 0; r < 16; ++r) acc[r] *= inv;
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            bf16x8 a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                float av = A[l31 * K + k0 + 8 * lh + j], bv = B[(k0 + 8 * lh + j) * 32 + l31];
                for (int p = 0; p < 3; ++p) {
                    a[p][j] = (__bf16)av; av -= (float)a[p][j];
                    b[p][j] = (__bf16)bv; bv -= (float)b[p][j];
                }
            }
            if constexpr (KIND == 1) {                                   // bf16x3
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
            } else if constexpr (KIND == 2) {                            // bf16x6, big terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
            } else {                                                     // bf16x6, small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
            }
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
}

// bf16x6 with the second-order terms kept in their own accumulator (added once at the end): the small sums do not lose their low
// bits against the large running sum
__global__ __launch_bounds__(64) void gemm_tile_two_acc(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    A += (size_t)blockIdx.x * 32 * K; B += (size_t)blockIdx.x * 32 * K; C += (size_t)blockIdx.x * 1024;
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        for (int j = 0; j < 8; ++j) {
            float av = A[l31 * K + k0 + 8 * lh + j], bv = B[(k0 + 8 * lh + j) * 32 + l31];
            for (int p = 0; p < 3; ++p) {
                a[p][j] = (__bf16)av; av -= (float)a[p][j];
                b[p][j] = (__bf16)bv; bv -= (float)b[p][j];
            }
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc2, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r] + acc2[r];
}

static float pow2_scale(const std::vector<float>& v, float target) {      // power of two s with max|v| * s <= target
    float mx = 0.f;
    for (float x : v) mx = fmaxf(mx, fabsf(x));
    int e;
    frexpf(target / (mx > 0 ? mx : 1.f), &e);
    return ldexpf(1.0f, e - 1);
}

int main() {
    struct Case { const char* name; int K; int a_kind, b_kind; } cases[] = {
        {"fwd 64->64 3x3 (K=576) W x act", 576, 0, 1}, {"fwd 256->256 3x3 (K=2304) W x act", 2304, 0, 1}, {"fwd 128->64 7x7 (K=6272) W x act", 6272, 0, 1},
        {"dgrad (K=2304) W x grad", 2304, 0, 2}, {"wgrad 32 rows (K=8192 px) grad x act", 8192, 2, 1}, {"wgrad (K=65536 px) grad x act", 65536, 2, 1}};
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    auto gen = [&](int kind) -> float {
        if (kind == 0) return 0.02f * nd(rng);                           // weights_init_normal
        if (kind == 1) { const float v = nd(rng) * 1.3f + 0.2f; return v > 0.f ? v : 0.f; }      // BatchNorm + ReLU
        const float v = nd(rng); return 1e-4f * v * expf(1.5f * nd(rng));                        // gradients: heavy tailed
    };
    printf("%-44s %11s %11s %11s %11s %11s %11s %11s\n", "case (rel. L2 error vs fp64)", "f32 fma", "f32 mfma", "bf16x3", "bf16x6", "bf16x6s", "bf16x6 2acc", "f16x2");
    for (const Case& cs : cases) {
        const int K = cs.K;
        const int NT = K >= 32768 ? 8 : (K >= 8192 ? 32 : 64);         // tiles per case
        std::vector<float> A((size_t)NT * 32 * K), B((size_t)NT * K * 32), C((size_t)NT * 1024);
        for (auto& v : A) v = gen(cs.a_kind);
        for (auto& v : B) v = gen(cs.b_kind);
        std::vector<double> ref((size_t)NT * 1024);
        std::vector<float> fma((size_t)NT * 1024);
        for (int t = 0; t < NT; ++t)
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0;
                    float f = 0.f;
                    for (int k = 0; k < K; ++k) {
                        const float a = A[((size_t)t * 32 + i) * K + k], b = B[((size_t)t * K + k) * 32 + j];
                        s += (double)a * b;
                        f = fmaf(a, b, f);
                    }
                    ref[(size_t)t * 1024 + i * 32 + j] = s;
                    fma[(size_t)t * 1024 + i * 32 + j] = f;
                }
        auto err = [&](const std::vector<float>& c) {
            double num = 0, den = 0;
            for (size_t i = 0; i < c.size(); ++i) { const double d = c[i] - ref[i]; num += d * d; den += ref[i] * ref[i]; }
            return sqrt(num / den);
        };
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        const float sa = pow2_scale(A, 16384.f), sb = pow2_scale(B, 16384.f);
        double e[6];
        auto run = [&](auto kern, int idx, float s0, float s1) {
            hipLaunchKernelGGL(kern, dim3(NT), dim3(64), 0, 0, dA, dB, dC, K, s0, s1);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            e[idx] = err(C);
        };
        run(gemm_tile<0>, 0, 1.f, 1.f);
        run(gemm_tile<1>, 1, 1.f, 1.f);
        run(gemm_tile<2>, 2, 1.f, 1.f);
        run(gemm_tile<3>, 3, 1.f, 1.f);
        hipLaunchKernelGGL(gemm_tile_two_acc, dim3(NT), dim3(64), 0, 0, dA, dB, dC, K);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        e[4] = err(C);
        run(gemm_tile<4>, 5, sa, sb);
        printf("%-44s %11.3e %11.3e %11.3e %11.3e %11.3e %11.3e %11.3e\n", cs.name, err(fma), e[0], e[1], e[2], e[3], e[4], e[5]);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
