// Split-precision operands for the 16-bit matrix cores (igemm_bf16x3.hip, wgrad_x3.hip): one fp32 value as hi + lo.
//
//   bf16x3 (F16 = false): hi = bf16(x), lo = bf16(x - hi); 16 significant bits, any fp32 exponent, no scale.
//                         hi*hi + hi*lo + lo*hi drops lo*lo ~ 2^-16: 4.4e-6 relative L2 per layer.
//   f16x2  (F16 = true) : hi = f16(x s), lo = f16(x s - hi); 22 significant bits, and fp16's 5-bit exponent makes a scale necessary:
//                         s is the power of two that puts the tensor's largest magnitude in [2^14, 2^15) -- the largest value is
//                         below fp16's 65504, an element 2^-17 of the maximum or larger keeps a normal hi (11 bits), and lo reaches
//                         down to fp16's subnormal quantum 2^-24, i.e. 2^-38 of the maximum.  The dropped lo*lo is ~2^-22.
//                         Measured against fp64 the contraction is in the exact-f32 MFMA's error class: in a bare 32 x 32 tile
//                         (tools/probe/split_precision_error.hip, profiles/r04_split_precision_error.log) 2.7e-7 vs the f32 chain's
//                         3.2e-7 at K = 576, 5.4e-7 vs 6.5e-7 at K = 2304 -- the f32 chain rounds its accumulator K times, this one
//                         3 K / 16 times; against the shipped f32 KERNELS, which split their reduction over chunks and blocks,
//                         0.65x .. 1.7x of their error per layer and operand, mean 0.97x over 94 measurements
//                         (tests/test_gpu_ops.py::test_conv2d_f16x2, profiles/r04_f16x2_layer_errors.txt); bf16x3 is ~15x above both.
// The largest magnitude travels as its fp32 BIT PATTERN ("absmax slot", faoctasr_absmax_bits): non-negative floats order like unsigned
// integers, so the reduction is an atomicMax and nobody divides.  A slot is FAOCTASR_ABSMAX_SLOT_WORDS = 128 words (512 B) of which
// 8 are used, one per 64-byte line: float atomics execute at the memory side, one address at a time (~8 ns each) -- the 2048 blocks
// of a producer all finishing together cost 65 us on ONE word (measured: BatchNorm kernels 34 -> 100 us) and ~1 us spread over 8
// lines after a block-level reduction.  Readers take the maximum of the 8 words (absmax_read).
#pragma once
#include <hip/hip_runtime.h>

namespace faoctasr {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2s __attribute__((ext_vector_type(2)));

constexpr int ABSMAX_PARTS = 8, ABSMAX_STRIDE = 16;               // include/faoctasr.h: FAOCTASR_ABSMAX_SLOT_WORDS = PARTS * STRIDE
__device__ __forceinline__ unsigned absmax_read(const unsigned* __restrict__ slot) {
    unsigned m = slot[0];
#pragma unroll
    for (int i = 1; i < ABSMAX_PARTS; ++i) {
        const unsigned v = slot[i * ABSMAX_STRIDE];
        m = v > m ? v : m;
    }
    return m;
}
// a block's maximum (fp32 bits, all 256 threads call; red: 4 words of LDS) into the part chosen by its block id
__device__ __forceinline__ void absmax_publish_block(unsigned mx, unsigned* __restrict__ slot, unsigned* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)mx, o, 64);
        mx = other > mx ? other : mx;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = red[0];
        for (unsigned i = 1; i < (blockDim.x >> 6); ++i) m = red[i] > m ? red[i] : m;
        if (m) atomicMax(slot + ((blockIdx.x + blockIdx.y) & (ABSMAX_PARTS - 1)) * ABSMAX_STRIDE, m);
    }
}

// s = 2^(141 - e) for a maximum 1.m x 2^(e - 127): max * s = 1.m x 2^14.  Tensors whose maximum is zero, subnormal-small (< 2^-111),
// infinite or NaN are left unscaled.
__host__ __device__ __forceinline__ float f16x2_scale(unsigned absmax_bits) {
    const unsigned e = (absmax_bits >> 23) & 0xffu;
    const unsigned sb = (e < 16u || e == 255u) ? 0x3f800000u : (268u - e) << 23;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, sb);
#else
    float f; __builtin_memcpy(&f, &sb, 4); return f;
#endif
}
__host__ __device__ __forceinline__ float f16x2_inv_scale(unsigned absmax_bits) {
    const unsigned e = (absmax_bits >> 23) & 0xffu;
    const unsigned sb = (e < 16u || e == 255u) ? 0x3f800000u : (e - 14u) << 23;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, sb);
#else
    float f; __builtin_memcpy(&f, &sb, 4); return f;
#endif
}

// (a, b) already scaled -> packed 16-bit hi pair and lo pair (a in the low half): hi = rne(v), lo = rne(v - hi)
template <bool F16>
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    if constexpr (F16) {
        const f16x2s h = __builtin_convertvector(f2{a, b}, f16x2s);
        hi = __builtin_bit_cast(unsigned, h);
        const f16x2s l = __builtin_convertvector(f2{a, b} - __builtin_convertvector(h, f2), f16x2s);
        lo = __builtin_bit_cast(unsigned, l);
    } else {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 h = {(__bf16)a, (__bf16)b};
        hi = __builtin_bit_cast(unsigned, h);
        const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
        const bf2 l = {(__bf16)ra, (__bf16)rb};
        lo = __builtin_bit_cast(unsigned, l);
    }
}

// (a, b) and the tensor's scale -> hi = {f16(a s), f16(b s)}, lo = {f16(a s - hi.a), f16(b s - hi.b)}.  F16: four v_fma_mix*_f16 -- the
// mixed-precision FMA multiplies by the scale, widens its f16 operand, subtracts and rounds to f16 in ONE instruction per value
// (2 per element against 3 + register-pair moves for the convert / widen / subtract / convert sequence; bit-identical to it:
// a s is exact, a s - hi is exact in fp32, and each result is rounded once).  bf16: no scale (s is ignored).
template <bool F16>
__device__ __forceinline__ void split_pair_scaled(float a, float b, float s, unsigned& hi, unsigned& lo) {
    if constexpr (F16) {
        unsigned h, l;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "v"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "v"(s));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(s), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(s), "v"(h));
        hi = h;
        lo = l;
    } else {
        split_pair<false>(a, b, hi, lo);
    }
}

// the fp32 value of the low / high 16-bit half of a packed pair
template <bool F16>
__device__ __forceinline__ float half_lo_f32(unsigned d) {
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2s, d)[0];
    else return __builtin_bit_cast(float, d << 16);
}
template <bool F16>
__device__ __forceinline__ float half_hi_f32(unsigned d) {
    if constexpr (F16) return (float)__builtin_bit_cast(f16x2s, d)[1];
    else return __builtin_bit_cast(float, d & 0xffff0000u);
}
template <bool F16>
__device__ __forceinline__ unsigned cvt_pair(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    if constexpr (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, f16x2s));
    else {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf2));
    }
}

// one 32 x 32 x 16 matrix product on 8-element 16-bit fragments held as raw 128-bit registers
template <bool F16, class V8>
__device__ __forceinline__ void mfma16(const V8& a, const V8& b, float __attribute__((ext_vector_type(16)))& acc) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

}  // namespace faoctasr
