// Shared helpers for the gfx950 kernels (error text, launch checks, activation math).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "faoctasr.h"

namespace faoctasr {

char* err_buf();
int fail(int code, const char* fmt, ...);
// which kernel family the last convolution-type call of this thread was dispatched to (diagnostics: faoctasr_last_route)
enum Route { ROUTE_NONE = 0, ROUTE_GATHER_FLAT = 1, ROUTE_PATCH = 2, ROUTE_WINOGRAD = 3, ROUTE_BF16X3 = 4, ROUTE_M1_FWD = 5, ROUTE_NARROW = 6,
             ROUTE_WGRAD_FLAT = 11, ROUTE_WGRAD_PATCH = 12, ROUTE_WGRAD_S1 = 13, ROUTE_M1_WGRAD = 14, ROUTE_WGRAD_X3 = 15, ROUTE_STEM_DGRAD = 7, ROUTE_STEM_WGRAD = 16 };
void set_route(int r);
extern thread_local int g_no_split_k;
// the absmax slot faoctasr_out_absmax left for this thread's next producer call (BatchNorm forward / backward, cat2_act forward):
// taken (and cleared) by that call
extern thread_local unsigned* g_out_absmax;
unsigned* take_out_absmax();

// Opt a kernel in to more than 64 KiB of dynamic LDS.  hipFuncSetAttribute is issued once per (kernel, size step), not
// per launch: the only process-wide state of the library is this grow-only record of attributes already set
// (one process drives one GPU).
void lds_optin(const void* kernel, size_t lds_bytes);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FAOCTASR_EHIP, "%s: %s", what, hipGetErrorString(e));
    return FAOCTASR_OK;
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case FAOCTASR_ACT_RELU: return v > 0.f ? v : 0.f;
        case FAOCTASR_ACT_LRELU: return v > 0.f ? v : v * slope;
        case FAOCTASR_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
    switch (act) {
        case FAOCTASR_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case FAOCTASR_ACT_LRELU: return y > 0.f ? 1.f : slope;
        case FAOCTASR_ACT_TANH: return 1.f - y * y;
        default: return 1.f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for 256-thread blocks; result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace faoctasr
