// Implicit-GEMM convolution for NARROW maps (output phase width < 24: the discriminators' 16x16 ... 2x2 layers) on the exact-f32
// MFMA of gfx950: conv forward, conv input gradient, transposed-conv forward / input gradient through the same tap-list
// geometry as igemm_patch.hip, reading the SAME packed weight image.
//
// These layers are plain GEMMs with a gathered B operand -- Y[M][pixels] = Wp[K][M]^T . im2col(X)[K][pixels] with M = 256..512,
// K = C * taps = 2048..8192, pixels = batch * OH * OW = 72..2048 -- and ran on the flat kernel of igemm.hip (K chunks of 16, two
// barriers and 24 scalar loads / stores per 16 MFMAs and wave, per-element index arithmetic on the weights: 0.20 of the MFMA
// peak, 8.3 ms per train step).  Here:
//   * K advances in the pack's own chunks of KR = kc * taps = 64 rows, 128 rows (m) x 64 pixels per block: 64 MFMAs per wave between
//     barriers;
//   * A = the packed weights Wp[chunk][r = tap * kc + c][Mpad]: a slab row is 512 contiguous bytes, 8 global_load_dwordx4 + 8
//     ds_write_b128 per thread and chunk, no index arithmetic;
//   * B is gathered with buffer loads whose 16 per-thread offsets (tap displacement + channel plane, or out of range = zero
//     padding) are computed ONCE -- a thread keeps its pixel for the whole K loop -- and the chunk's channel advance is the scalar
//     offset of the instruction: no vector arithmetic per element;
//   * both operands are register-prefetched one chunk ahead (LDS single buffered, 48 KiB: three blocks per CU);
//   * split-K over chunks when the tile grid cannot fill the chip (fp32 atomics into a zeroed output, as before).
#include <type_traits>

#include "common.h"
#include "igemm_geom.h"

namespace faoctasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NM_MT = 128, NM_NT = 64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int I, int N, class F>
__device__ __forceinline__ void nm_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        nm_static_for<I + 1, N>(f);
    }
}

// NM_KR = rows of a packed chunk (kc * taps): 64 or 32
template <int NM_KR>
__global__ __launch_bounds__(256) void igemm_nm_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                                                       float* __restrict__ y, const PatchGeom g, const int ksplit) {
    __shared__ __attribute__((aligned(16))) float A_s[NM_KR * NM_MT];    // [r][m]
    __shared__ __attribute__((aligned(16))) float B_s[NM_KR * NM_NT];    // [r][pixel]
    __shared__ int taps_s[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the tap table through LDS: indexing the by-value kernel argument with a per-thread index compiles to one dependent global load
    // per tap (16 serialised round trips before the first chunk: several microseconds of a block that lives for ~20)
    if (tid < 64) taps_s[tid] = g.taps[tid];
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    const int ph = blockIdx.z / ksplit, ks = blockIdx.z - ph * ksplit;
    const int GH = g.gh[ph], GW = g.gw[ph];
    const long npix = (long)g.N * GH * GW;
    const long j0 = (long)blockIdx.x * NM_NT;
    if (j0 >= npix) return;
    const int m0 = blockIdx.y * NM_MT;
    const int t0 = g.t0[ph], T = g.t0[ph + 1] - t0, kc = g.kc[ph];       // kc * T == NM_KR (checked by the launcher)
    (void)T;
    const int nchunks = g.C / kc;                                        // C % kc == 0 (checked)
    const int cps = (nchunks + ksplit - 1) / ksplit;
    const int ch0 = ks * cps;
    int ch1 = ch0 + cps;
    ch1 = ch1 < nchunks ? ch1 : nchunks;
    if (ch0 >= ch1) return;
    const long chw = (long)g.IH * g.IW;

    // ---- B gather: this thread's pixel and its 16 rows r = rg + 4 i of every chunk ----
    const int jj = tid & 63, rg = tid >> 6;
    const long j = j0 + jj;
    constexpr int NB = NM_KR / 4, NA = NM_KR / 8;                        // B dwords / A float4 staged per thread and chunk
    unsigned boff[NB];
    {
        constexpr unsigned OOB = 0x80000000u;
        int n = 0, a = 0, b = 0;
        const bool jv = j < npix;
        if (jv) {
            n = (int)(j / ((long)GH * GW));
            const int r = (int)(j - (long)n * GH * GW);
            a = r / GW;
            b = r - a * GW;
        }
        const long img = (long)n * g.C * chw;                            // element offset of the image; < 2^29 checked by the launcher
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r = rg + 4 * i, t = r / kc, cl = r - t * kc;
            const int tp = taps_s[t0 + t];
            int iy = a * g.SI + (tp & 0xff) + g.oy0[ph], ix = b * g.SI + ((tp >> 8) & 0xff) + g.ox0[ph];
            if (g.reflect) {
                iy = iy < 0 ? -iy : iy; iy = iy >= g.IH ? 2 * g.IH - 2 - iy : iy;
                ix = ix < 0 ? -ix : ix; ix = ix >= g.IW ? 2 * g.IW - 2 - ix : ix;
            }
            const bool ok = jv && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
            boff[i] = ok ? 4u * (unsigned)(img + (long)cl * chw + (long)iy * g.IW + ix) : OOB;
        }
    }
    const long x_bytes = (long)g.N * g.C * chw * 4;
    const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)(x_bytes < 0x7ffffff0L ? x_bytes : 0x7ffffff0L), 0x00020000);
    // ---- A slab: 64 rows x 128 floats = 2048 float4; thread piece i = tid + 256 i: row = piece >> 5, 16-byte column = piece & 31 ----
    const float* wslab = wp + g.pack_off[ph] + (long)m0;
    const long slab_stride = (long)NM_KR * g.Mpad;                       // floats per chunk

    f32x4 ra[NA];
    float rb[NB];
    auto load_chunk = [&](int ch) {
        const float* ws = wslab + (long)ch * slab_stride;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int piece = tid + 256 * i;
            ra[i] = *reinterpret_cast<const f32x4*>(ws + (long)(piece >> 5) * g.Mpad + (piece & 31) * 4);
        }
        const int soff = (int)(4L * ch * kc * chw);                      // the chunk's first channel plane (scalar)
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrd, boff[i], soff, 0));
    };

    // 4 waves = 2 (rows) x 2 (pixels): wave tile 64 rows x 32 pixels
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;

    // fragment addresses: A_s[k][m] and B_s[k][pixel] with k = kk + lh; one ds_read2st64_b32 (two rows 2 k-steps apart ... see rd)
    const unsigned lds_a = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)A_s + 4u * (unsigned)(wm * 64 + l31 + lh * NM_MT);
    const unsigned lds_b = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)B_s + 4u * (unsigned)(wn * 32 + l31 + lh * NM_NT);
    // a "quad" = 4 k-steps (8 rows).  ds_read2st64_b32 reads two dwords 64-dword units apart: A rows are 128 floats (2 units), B rows
    // 64 floats (1 unit), so one instruction fetches the operand of k-steps (j, j+1).  6 LDS instructions per 8 MFMAs, issued one
    // quad ahead with a counted wait (the compiler's own schedule waited lgkmcnt(0) before every MFMA pair).
    constexpr int NQ = NM_KR / 8;
    f32x2 A0[2][2], A1[2][2], Bq0[2], Bq1[2];                            // [row block mi][k-step pair], [k-step pair]
    auto rd = [&](auto qc, f32x2 (&A)[2][2], f32x2 (&B)[2]) {
        constexpr int q = decltype(qc)::value;
        const unsigned la0 = lds_a, la1 = lds_a + 128u, lb = lds_b;       // (asm operands cannot name the enclosing function's variables)
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(A[0][0]) : "v"(la0), "n"(2 * (8 * q)), "n"(2 * (8 * q + 2)));
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(A[1][0]) : "v"(la1), "n"(2 * (8 * q)), "n"(2 * (8 * q + 2)));
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(B[0]) : "v"(lb), "n"(8 * q), "n"(8 * q + 2));
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(A[0][1]) : "v"(la0), "n"(2 * (8 * q + 4)), "n"(2 * (8 * q + 6)));
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(A[1][1]) : "v"(la1), "n"(2 * (8 * q + 4)), "n"(2 * (8 * q + 6)));
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(B[1]) : "v"(lb), "n"(8 * q + 4), "n"(8 * q + 6));
    };
    auto wait_set = [&](f32x2 (&A)[2][2], f32x2 (&B)[2], auto yc) {
        constexpr int younger = decltype(yc)::value;
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]), "+v"(A[1][1]), "+v"(B[0]), "+v"(B[1]) : "n"(younger));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto quad = [&](auto qc, f32x2 (&A)[2][2], f32x2 (&B)[2], f32x2 (&An)[2][2], f32x2 (&Bn)[2]) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q + 1 < NQ) {
            rd(std::integral_constant<int, q + 1>{}, An, Bn);
            wait_set(A, B, std::integral_constant<int, 6>{});
        } else {
            wait_set(A, B, std::integral_constant<int, 0>{});
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[0][st >> 1][st & 1], B[st >> 1][st & 1], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[1][st >> 1][st & 1], B[st >> 1][st & 1], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    load_chunk(ch0);
    for (int ch = ch0; ch < ch1; ++ch) {
        if (ch != ch0) __syncthreads();                                  // the previous chunk's fragment reads are done
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int piece = tid + 256 * i;
            *reinterpret_cast<f32x4*>(A_s + (piece >> 5) * NM_MT + (piece & 31) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) B_s[(rg + 4 * i) * NM_NT + jj] = rb[i];
        __syncthreads();
        if (ch + 1 < ch1) load_chunk(ch + 1);                            // in flight under the MFMAs below
        rd(std::integral_constant<int, 0>{}, A0, Bq0);
        nm_static_for<0, NQ / 2>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            quad(std::integral_constant<int, 2 * h>{}, A0, Bq0, A1, Bq1);
            quad(std::integral_constant<int, 2 * h + 1>{}, A1, Bq1, A0, Bq0);
        });
    }

    // ---- epilogue: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
    const long jo = j0 + wn * 32 + l31;
    if (jo >= npix) return;
    const int no = (int)(jo / ((long)GH * GW));
    const int r0 = (int)(jo - (long)no * GH * GW);
    const int ao = r0 / GW, bo = r0 - ao * GW;
    const long ohw = (long)g.OH * g.OW;
    const long obase = (long)no * g.M * ohw + (long)(ao * g.SO + g.py[ph]) * g.OW + (bo * g.SO + g.px[ph]);
    // The bias / split-K cases are decided once and a block whose 128 rows all exist runs straight-line code: with the branches inside
    // the per-register loop every element compiled to its own region -- `global_load_dword (bias); s_waitcnt vmcnt(0); store`, 32 serial
    // memory round trips per thread (round 4's ISA)
    const bool use_bias = bias && ks == 0;
    if (m0 + NM_MT <= g.M) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int mb = m0 + wm * 64 + mi * 32 + 4 * lh;
            float bv[16];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) bv[rr] = use_bias ? bias[mb + (rr & 3) + 8 * (rr >> 2)] : 0.f;
            if (ksplit > 1) {
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) atomicAdd(y + obase + (long)(mb + (rr & 3) + 8 * (rr >> 2)) * ohw, acc[mi][rr] + bv[rr]);
            } else {
#pragma unroll
                for (int rr = 0; rr < 16; ++rr)
                    y[obase + (long)(mb + (rr & 3) + 8 * (rr >> 2)) * ohw] = act_apply(acc[mi][rr] + bv[rr], g.act, g.slope);
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int m = m0 + wm * 64 + mi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            if (m < g.M) {
                float v = acc[mi][rr];
                if (use_bias) v += bias[m];
                if (ksplit > 1) atomicAdd(y + obase + (long)m * ohw, v);
                else y[obase + (long)m * ohw] = act_apply(v, g.act, g.slope);
            }
        }
}

// returns 1 when launched, 0 when the shape is left to the flat kernel, <0 on error.  `wp` must hold the packed image of
// launch_pack (igemm_patch.hip) for this geometry.
int launch_narrow(const float* x, const float* wp, const float* bias, float* y, PatchGeom& g, int act, float slope, hipStream_t s) {
    g.act = act; g.slope = slope;
    if (g.M < 64 || (long)g.N * g.C * g.IH * g.IW >= (1L << 29)) return 0;
    long maxpix = 0;
    int minchunks = 1 << 30;
    int KR = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const int T = g.t0[p + 1] - g.t0[p];
        if (T <= 0 || g.C % g.kc[p] != 0) return 0;
        if (p == 0) KR = g.kc[p] * T;
        if (g.kc[p] * T != KR) return 0;
        const long np = (long)g.N * g.gh[p] * g.gw[p];
        maxpix = np > maxpix ? np : maxpix;
        const int nc = g.C / g.kc[p];
        minchunks = nc < minchunks ? nc : minchunks;
    }
    if (KR != 64 && KR != 32) return 0;
    if (maxpix == 0) return 0;
    const long gx = (maxpix + NM_NT - 1) / NM_NT, gy = (g.M + NM_MT - 1) / NM_MT;
    const long blocks = gx * gy * g.nphase;
    // split K until ~2 blocks per CU exist, each still reducing >= 4 chunks; every split adds one atomic pass over the output
    // (1.3 TB/s chip-wide) and one block prologue
    int ksplit = 1;
    if (act == FAOCTASR_ACT_NONE && blocks < 384) {
        ksplit = (int)((512 + blocks - 1) / blocks);
        if (ksplit > minchunks / 4) ksplit = minchunks / 4;
        if (ksplit < 1) ksplit = 1;
    }
    if (g_no_split_k) ksplit = 1;
    if (ksplit > 1 && hipMemsetAsync(y, 0, sizeof(float) * (size_t)g.N * g.M * g.OH * g.OW, s) != hipSuccess)
        return fail(FAOCTASR_EHIP, "memset y failed");
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)(g.nphase * ksplit));
    if (KR == 64) hipLaunchKernelGGL(igemm_nm_kernel<64>, grid, dim3(256), 0, s, x, wp, bias, y, g, ksplit);
    else hipLaunchKernelGGL(igemm_nm_kernel<32>, grid, dim3(256), 0, s, x, wp, bias, y, g, ksplit);
    const int rc = check_launch("igemm_nm");
    return rc == FAOCTASR_OK ? 1 : rc;
}

}  // namespace faoctasr
