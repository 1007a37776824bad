// Does VALU work of a co-resident wave hide behind MFMAs on gfx950?  One 512-thread block per CU: waves 0-3 (one per SIMD) run
// a chain-free stream of MFMAs, waves 4-7 a stream of independent v_fma_f32.  Times: MFMA alone, VALU alone, both.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: f32 16x16x4, 1: f32 32x32x2, 2: bf16 32x32x16
__global__ __launch_bounds__(512) void probe(float* out, int n_mfma, int n_valu, int mode) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    __syncthreads();
    if (wave < 4) {
        if (mode & 1) {
            if constexpr (KIND == 0) {
                f32x4 acc[8];
                for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float a = threadIdx.x * 1e-3f, b = 1.0f;
                float v[4] = {a, b, a + 1.f, b + 2.f};
                if (mode & 8) {
                    for (int it = 0; it < n_mfma; it += 8)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);      // 4 fillers in the same wave
                            __builtin_amdgcn_sched_barrier(0);
                        }
                } else {
                for (int it = 0; it < n_mfma; it += 8)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                }
                for (int i = 0; i < 8; ++i) r += acc[i][0];
                r += v[0] + v[1] + v[2] + v[3];
            } else if constexpr (KIND == 1) {
                f32x16 acc[4];
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
                const float a = threadIdx.x * 1e-3f, b = 1.0f;
                for (int it = 0; it < n_mfma; it += 4)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                for (int i = 0; i < 4; ++i) r += acc[i][0];
            } else {
                f32x16 acc[4];
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
                bf16x8 a, b;
                for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f); b[j] = (__bf16)1.0f; }
                float v[4] = {1.f, 2.f, 3.f, 4.f};
                if (mode & 8) {
                    for (int it = 0; it < n_mfma; it += 4)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                } else {
                for (int it = 0; it < n_mfma; it += 4)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                }
                for (int i = 0; i < 4; ++i) r += acc[i][0];
                r += v[0] + v[1] + v[2] + v[3];
            }
        }
    } else if (mode & 4) {                                               // co-resident LDS read stream
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4* p = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
        for (int it = 0; it < n_valu / 2; it += 8)
#pragma unroll
            for (int i = 0; i < 8; ++i) s4 += p[i * 64 + ((it >> 3) & 7) * 0];
        r = s4[0] + s4[1];
    } else if (mode & 2) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
        const float m = 1.0001f, c = 0.5f;
        for (int it = 0; it < n_valu; it += 16)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
        for (int i = 0; i < 16; ++i) r += v[i];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND>
static void run(const char* name, int cyc_per_mfma) {
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    const int n_mfma = 1 << 16;
    const int n_valu = n_mfma * cyc_per_mfma / 4;        // the same nominal issue time (4 cycles per wave64 VALU op)
    float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float t9 = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, out, n_mfma, n_valu, 9);
        hipEventRecord(e);
        hipEventSynchronize(e);
        hipEventElapsedTime(&t9, s, e);
    }
    for (int mode = 1; mode <= 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(s);
            hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, out, n_mfma, n_valu, mode);
            hipEventRecord(e);
            hipEventSynchronize(e);
            hipEventElapsedTime(&t[mode], s, e);
        }
    }
    printf("%-28s MFMA alone %7.3f ms | VALU alone %7.3f ms | both %7.3f ms  -> %s\n", name, t[1], t[2], t[3],
           t[3] > 0.85f * (t[1] + t[2]) ? "ADD (no overlap)" : (t[3] < 1.15f * (t[1] > t[2] ? t[1] : t[2]) ? "OVERLAP" : "partial"));
    printf("%-28s                       | LDS reads alone %6.3f ms | MFMA + LDS reads %7.3f ms  -> %s\n", "", t[4], t[5],
           t[5] > 0.85f * (t[1] + t[4]) ? "ADD (no overlap)" : (t[5] < 1.15f * (t[1] > t[4] ? t[1] : t[4]) ? "OVERLAP" : "partial"));
    printf("%-28s                       | MFMA with 4 v_fma_f32 fillers per MFMA in the SAME wave %7.3f ms (x%.2f of MFMA alone)\n", "", t9, t9 / t[1]);
    hipFree(out);
}

int main() {
    run<0>("v_mfma_f32_16x16x4_f32", 32);
    run<1>("v_mfma_f32_32x32x2_f32", 64);
    run<2>("v_mfma_f32_32x32x16_bf16", 32);
    return 0;
}
