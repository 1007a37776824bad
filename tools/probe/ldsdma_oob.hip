// Probe: does an out-of-range lane of `buffer_load_dword ... lds` (LDS-DMA through a buffer descriptor) write 0 to LDS,
// or leave the LDS word untouched?  (decides whether zero padding can be done by the descriptor's range check)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float* g, int nrec_bytes, float* out) {
    __shared__ float buf[256];
    const int lane = threadIdx.x;
    buf[lane] = -7.0f; buf[lane + 64] = -7.0f;
    __syncthreads();
    auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, nrec_bytes, 0x00020000);
    // lanes 0..31 in range, odd lanes >= 32 out of range
    unsigned off = (lane < 32 || (lane & 1) == 0) ? 4u * lane : 0x7fffffffu;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)buf, 4, off, 0, 0, 0);
    __syncthreads();
    out[lane] = buf[lane];
    out[lane + 64] = buf[lane + 64];
}
int main() {
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
    float *d, *o; hipMalloc(&d, 256); hipMalloc(&o, 512);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, 256, o);
    float r[128]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%g ", r[i]);
    printf("\n| untouched tail: %g %g\n", r[64], r[127]);
    return 0;
}
