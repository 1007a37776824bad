// LDS-patch implicit-GEMM convolution for gfx950 (the main conv path; igemm.hip keeps the flat
// im2col kernel for small spatial sizes and as the no-pack fallback).
//
// For a tile of TH x 32 output pixels of one image and a chunk of KC input channels, the input
// patch ((TH-1)*SI + span) x (31*SI + span) x KC is staged ONCE in LDS (zero / reflect padding
// resolved while staging) and every tap's B fragment for v_mfma_f32_32x32x2_f32 is then a plain
// ds_read_b32 at a constant offset from the lane's base: no im2col index arithmetic in the K loop
// and ~T times fewer global loads than the flat kernel.  The A operand (weights) comes from a
// pre-packed image  Wp[phase][chunk][r = t*KC + c][Mpad]  whose (chunk, m-tile) slab is copied
// global -> LDS with global_load_lds_dwordx4 (LDS-DMA, no VGPRs), double buffered against the MFMAs.
// One barrier per K chunk.
//
// GEMM per phase (see igemm.hip for the phase decomposition of transposed convolutions):
//   Y[m][(n,a,b)] = sum_{t,c} Wp[(t,c)][m] * X[n][c][a*SI + oy_t][b*SI + ox_t]
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_geom.h"
#include "pack_bodies.h"

namespace faoctasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int reflect_idx_p(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// --- hand-counted LDS fragment reads ---------------------------------------------------------------------------
// hipcc waits lgkmcnt(0) for LDS reads in this loop shape, which serialises a wave's fragment reads with its own MFMAs.
// The reads are therefore issued from inline asm (invisible to the compiler's counter bookkeeping) and retired with a
// COUNTED s_waitcnt that names the destination registers, so the consumer MFMAs cannot be scheduled above it
// (cdna_hip_programming.md 5.7, form (ii)).  Only these reads are in flight inside the k-step loop.
__device__ __forceinline__ void ds_read_f32(float& dst, unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr)); }
__device__ __forceinline__ void ds_read_f32_o128(float& dst, unsigned addr) {
    asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(dst) : "v"(addr));
}
template <int I, int N, class F>
__device__ __forceinline__ void pk_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pk_static_for<I + 1, N>(f);
    }
}
// fragment reads whose offsets are immediates of the instruction (dense-grid specialisation: no address arithmetic at all)
template <int MI, int NI, int AOFF, int BOFF, int BROW>
__device__ __forceinline__ void read_frags_imm(float (&a)[MI], float (&b)[NI], unsigned abase, unsigned bbase) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[0]) : "v"(abase), "n"(AOFF));
    if constexpr (MI == 2) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[1]) : "v"(abase), "n"(AOFF + 128));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[0]) : "v"(bbase), "n"(BOFF));
    if constexpr (NI == 2) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[1]) : "v"(bbase), "n"(BOFF + BROW));
}
template <int MI, int NI>
__device__ __forceinline__ void read_frags(float (&a)[MI], float (&b)[NI], unsigned aaddr, unsigned baddr, unsigned brow_bytes) {
    ds_read_f32(a[0], aaddr);
    if constexpr (MI == 2) ds_read_f32_o128(a[1], aaddr);
    ds_read_f32(b[0], baddr);
    if constexpr (NI == 2) ds_read_f32(b[1], baddr + brow_bytes);
}
// wait until at most `MI+NI` (the other fragment set) of this wave's LDS reads are outstanding
template <int MI, int NI>
__device__ __forceinline__ void wait_frags_keep_next(float (&a)[MI], float (&b)[NI]) {
    if constexpr (MI == 2 && NI == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
    else if constexpr (MI == 1 && NI == 2) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]));
    else if constexpr (MI == 2 && NI == 1) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]));
    else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[0]), "+v"(b[0]));
}
template <int MI, int NI>
__device__ __forceinline__ void wait_frags_all(float (&a)[MI], float (&b)[NI]) {
    if constexpr (MI == 2 && NI == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
    else if constexpr (MI == 1 && NI == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]));
    else if constexpr (MI == 2 && NI == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]));
}

// pack kernel: one thread per packed element (pack_bodies.h)
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, const PatchGeom g) {
    patch_pack_block(w, wp, g, blockIdx.x, gridDim.x);
}

template <int WM, int WN, int MI, int NI, int SI, int DENSE = 0, int KCS = 2>
__global__ __launch_bounds__(256) void igemm_patch_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          const PatchGeom g, const int ksplit) {
    constexpr int MT = WM * MI * 32, TH = WN * NI, NPV = PATCH_MAX_PER_THREAD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ph = blockIdx.z / ksplit, ks = blockIdx.z - ph * ksplit;
    const int GH = g.gh[ph], GW = g.gw[ph];
    const int tiles_x = (GW + 31) >> 5, tiles_y = (GH + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y;
    const int bt = blockIdx.x;
    if (bt >= g.N * tiles) return;
    const int n = bt / tiles;
    const int rt = bt - n * tiles;
    const int ty = rt / tiles_x, tx = rt - ty * tiles_x;
    const int t0 = g.t0[ph], T = g.t0[ph + 1] - t0, KC = g.kc[ph];
    const int PH = (TH - 1) * SI + g.span_y[ph] + 1, PW = 31 * SI + g.span_x[ph] + 1;
    const int PHW = PH * PW;
    const int a_floats = (KC * T * MT + 255) & ~255;              // 1 KiB granules (one LDS-DMA wave-instruction)
    float* const A_lds = reinterpret_cast<float*>(smem);           // 2 buffers
    float* const P_lds = A_lds + 2 * a_floats;                     // 2 buffers of KC*PHW floats
    const int m0 = blockIdx.y * MT;
    const int IH = g.IH, IW = g.IW;
    const long chw = (long)IH * IW;
    const int nchunks = (g.C + KC - 1) / KC;
    // split-K: this block reduces chunks [ch0, ch1) and adds its partial tile atomically (output zeroed by the launcher)
    const int cps = (nchunks + ksplit - 1) / ksplit;
    const int ch0 = ks * cps;
    const int ch1 = (ch0 + cps) < nchunks ? (ch0 + cps) : nchunks;
    if (ch0 >= ch1) return;
    const int y_base = ty * TH * SI + g.oy0[ph], x_base = tx * 32 * SI + g.ox0[ph];
    const float* xin = x + (long)n * g.C * chw;
    const float* wslab = wp + g.pack_off[ph] + m0;                 // + (chunk*KC*T + r)*Mpad
    const int npatch = KC * PHW;
    const float invPW = 1.0f / (float)PW, invPHW = 1.0f / (float)PHW;


    // per-thread patch element geometry is chunk-invariant: decode once into a byte offset inside the chunk's input
    // planes.  Loads go through a buffer descriptor whose num_records ends at the last real channel, so an offset past it
    // returns 0: that implements zero padding (offset OOB), the channel tail of the last chunk and masked tile pixels
    // without selects, and the offsets need one VGPR each instead of a 64-bit address pair.
    constexpr unsigned OOB = 0x7fffffffu;
    unsigned poff[NPV];
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
        const int e = tid + 256 * i;
        unsigned off = OOB;
        if (e < npatch) {
            const int c = (int)(((float)e + 0.5f) * invPHW);
            const int r = e - c * PHW;
            const int py = (int)(((float)r + 0.5f) * invPW);
            const int px = r - py * PW;
            int iy = y_base + py, ix = x_base + px;
            if (g.reflect) {
                // positions beyond the reflected range belong to masked-out tile pixels only
                iy = reflect_idx_p(iy, IH);
                ix = reflect_idx_p(ix, IW);
            }
            if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) off = 4u * (unsigned)(c * (int)chw + iy * IW + ix);
        }
        poff[i] = off;
    }
    float pv[NPV];

    auto issue_A = [&](int chunk, int buf) {
        // slab rows r in [0, KC*T): MT floats each at row stride Mpad -> linear LDS image [r][MT]
        const float* src = wslab + (long)chunk * KC * T * g.Mpad;
        const int pieces = KC * T * (MT / 4);                      // 16-byte pieces
        float* dst = A_lds + buf * a_floats;
        for (int j = wave; j * 64 < pieces; j += 4) {
            int p = j * 64 + lane;
            p = p < pieces ? p : pieces - 1;                        // tail lanes re-read the last piece into the pad
            const int row = p / (MT / 4), q = p - row * (MT / 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)row * g.Mpad + q * 4),
                                             (__attribute__((address_space(3))) void*)(dst + j * 256), 16, 0, 0);
        }
    };
    auto load_patch = [&](int chunk) {
        const int c0 = chunk * KC;
        const long bytes = (long)(g.C - c0) * chw * 4;
        const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin + (long)c0 * chw), 0,
                                                           (int)(bytes < 0x7ffffff0L ? bytes : 0x7ffffff0L), 0x00020000);
#pragma unroll
        for (int i = 0; i < NPV; ++i) pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, poff[i], 0, 0));
    };
    auto store_patch = [&](int buf, int chunk) {
        (void)chunk;
        float* dst = P_lds + buf * npatch;
#pragma unroll
        for (int i = 0; i < NPV; ++i) {
            const int e = tid + 256 * i;
            if (e < npatch) dst[e] = pv[i];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int a_lane = lh * MT + wm * (MI * 32) + l31;
    const int b_lane = lh * PHW + (wn * NI) * SI * PW + l31 * SI;

    issue_A(ch0, 0);
    load_patch(ch0);
    store_patch(0, ch0);
    __syncthreads();

    const int nsteps = T * (KC >> 1);                  // MFMA k-steps per chunk: (tap, channel pair)
    // lane t of tapv holds tap t's offset inside the patch; a k-step fetches it with v_readlane (no memory access)
    int tapv = 0;
    if (lane < T) {
        const int tp = g.taps[t0 + lane];
        tapv = (tp & 0xff) * PW + ((tp >> 8) & 0xff);
    }
    for (int ch = ch0; ch < ch1; ++ch) {
        const int cur = (ch - ch0) & 1;
        if (ch + 1 < ch1) {
            issue_A(ch + 1, cur ^ 1);
            load_patch(ch + 1);
        }
        // software-pipelined fragment reads: the ds_reads of step s+1 are in flight under the MFMAs of step s.
        // A rows are ordered (tap, channel) so the A address just advances by 2 rows per step; the B address advances by
        // 2 channel planes and is re-based at each new tap.  The read issued for the step after the last one runs past the
        // chunk (still inside the LDS allocation: the guard KiB) and is never used.
        const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
        unsigned Aa = lds0 + 4u * (unsigned)(cur * a_floats + a_lane);
        const unsigned Pc = lds0 + 4u * (unsigned)(2 * a_floats + cur * npatch + b_lane);
        unsigned Pa = Pc + 4u * (unsigned)__builtin_amdgcn_readlane(tapv, 0);
        const unsigned brow = 4u * (unsigned)(SI * PW), a_step = 8u * MT, p_step = 8u * (unsigned)PHW;
        float a0[MI], b0[NI], a1[MI], b1[NI];          // two named fragment sets (static indexing, no copies)
        if constexpr (DENSE != 0) {
            // Dense KW x KW tap grid with a compile-time chunk of KCS channels (the 7x7 layers, the 4x4 stride-2 layers, the 2x2
            // phases of the 4x4 transposed convolution): step s = (tap, channel pair) has a compile-time patch offset and weight
            // row, so every fragment read carries its address as an immediate and the k-loop is reads + MFMAs only.  (Vector
            // instructions are not hidden behind f32 MFMAs on this chip: DESIGN.md 4.1a.)  DENSE > 0: taps in ascending (ky, kx)
            // order; DENSE < 0: descending (input gradient of a stride-1 convolution, phases of a transposed one).
            constexpr int KW = DENSE > 0 ? DENSE : -DENSE, CP = KCS / 2, NS = KW * KW * CP;
            constexpr int PWc = 31 * SI + KW, PHc = (TH - 1) * SI + KW, BROW = 4 * SI * PWc;
            const unsigned Ab = lds0 + 4u * (unsigned)(cur * a_floats + a_lane);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            auto boff = [](int st) constexpr {
                const int tt = st / CP, cq = st % CP;
                const int tap = DENSE > 0 ? (tt / KW) * PWc + tt % KW : (KW - 1 - tt / KW) * PWc + (KW - 1 - tt % KW);
                return 4 * (tap + 2 * cq * PHc * PWc);
            };
            read_frags_imm<MI, NI, 0, boff(0), BROW>(a0, b0, Ab, Pc);
            pk_static_for<0, (NS + 1) / 2>([&](auto ic) {
                constexpr int st = 2 * decltype(ic)::value;
                if constexpr (st + 1 < NS) {
                    read_frags_imm<MI, NI, (st + 1) * 8 * MT, boff(st + 1), BROW>(a1, b1, Ab, Pc);
                    wait_frags_keep_next<MI, NI>(a0, b0);
                } else {
                    wait_frags_all<MI, NI>(a0, b0);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mi], b0[ni], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (st + 1 < NS) {
                    if constexpr (st + 2 < NS) {
                        read_frags_imm<MI, NI, (st + 2) * 8 * MT, boff(st + 2), BROW>(a0, b0, Ab, Pc);
                        wait_frags_keep_next<MI, NI>(a1, b1);
                    } else {
                        wait_frags_all<MI, NI>(a1, b1);
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mi], b1[ni], acc[mi][ni], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if (ch + 1 < ch1) store_patch(cur ^ 1, ch + 1);
            __syncthreads();
            continue;
        }
        int t = 0, cp = 0;
        auto advance = [&]() {                           // branch-free: keeps the loop body straight-line
            cp += 2;
            const bool wrap = cp >= KC;
            cp = wrap ? 0 : cp;
            t += wrap ? 1 : 0;
            Aa += a_step;
            const unsigned nt = Pc + 4u * (unsigned)__builtin_amdgcn_readlane(tapv, t & 63);
            Pa = wrap ? nt : Pa + p_step;
        };
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): nothing but our reads is counted from here on
        read_frags<MI, NI>(a0, b0, Aa, Pa, brow);
        for (int pr = nsteps >> 1; pr > 0; --pr) {
            advance();
            read_frags<MI, NI>(a1, b1, Aa, Pa, brow);
            wait_frags_keep_next<MI, NI>(a0, b0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mi], b0[ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            advance();
            read_frags<MI, NI>(a0, b0, Aa, Pa, brow);
            wait_frags_keep_next<MI, NI>(a1, b1);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mi], b1[ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_frags_all<MI, NI>(a0, b0);                    // retire the run-ahead read (used only when nsteps is odd)
        if (nsteps & 1) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mi], b0[ni], acc[mi][ni], 0, 0, 0);
        }
        if (ch + 1 < ch1) store_patch(cur ^ 1, ch + 1);
        __syncthreads();      // LDS-DMA of the next A slab has landed (vmcnt(0)), next patch visible, this chunk's reads done
    }

    // epilogue: bias and activation on the accumulator registers (activation switch hoisted out of the store loops)
    const int mrow0 = m0 + wm * (MI * 32) + 4 * lh;
    if (bias && ks == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = mrow0 + mi * 32 + (rr & 3) + 8 * (rr >> 2);
                const float bv = m < g.M ? bias[m] : 0.f;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni][rr] += bv;
            }
    }
    if (g.act == FAOCTASR_ACT_RELU) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) acc[mi][ni][rr] = fmaxf(acc[mi][ni][rr], 0.f);
    } else if (g.act == FAOCTASR_ACT_LRELU) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const float v = acc[mi][ni][rr];
                    acc[mi][ni][rr] = v > 0.f ? v : v * g.slope;
                }
    } else if (g.act == FAOCTASR_ACT_TANH) {
        for (int mi = 0; mi < MI; ++mi)
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) acc[mi][ni][rr] = tanhf(acc[mi][ni][rr]);
    }
    const long ohw = (long)g.OH * g.OW;
    const int bo = tx * 32 + l31;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int ao = ty * TH + wn * NI + ni;
        if (ao >= GH || bo >= GW) continue;
        float* yo = y + (long)n * g.M * ohw + (long)(ao * g.SO + g.py[ph]) * g.OW + (bo * g.SO + g.px[ph]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = mrow0 + mi * 32 + (rr & 3) + 8 * (rr >> 2);
                if (m < g.M) {
                    if (ksplit > 1) atomicAdd(yo + (long)m * ohw, acc[mi][ni][rr]);
                    else yo[(long)m * ohw] = acc[mi][ni][rr];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// LDS bytes of a block for tile (MT, TH) at chunk size kc
static size_t lds_need(int kc, int T, int MT, int TH, int SI, int span_y, int span_x) {
    const int PH = (TH - 1) * SI + span_y + 1, PW = 31 * SI + span_x + 1;
    const size_t a_floats = ((size_t)kc * T * MT + 255) & ~(size_t)255;
    return (2 * a_floats + 2 * (size_t)kc * PH * PW) * 4 + 1024;      // + guard for the pipelined run-past read
}
constexpr size_t PATCH_LDS_BUDGET = 79 * 1024;     // two blocks per CU (160 KiB)

// K chunk = kc channels x all taps of the phase.  kc depends only on the layer (never on the batch), so a packed
// weight image stays valid across calls: budget for the widest tile the dispatcher may pick for this M.
static int kc_for(int T, int C, int M, int SI, int span_y, int span_x) {
    const int MT = M > 64 ? 128 : 64, TH = M > 64 ? 4 : 8;
    const int ceven = (C + 1) & ~1;
    int best = 2;
    long best_pad = -1;
    for (int kc = 2; kc <= 16 && kc <= ceven; kc += 2) {
        const int PH = (TH - 1) * SI + span_y + 1, PW = 31 * SI + span_x + 1;
        if (kc > 2 && !(kc * T <= 112 && lds_need(kc, T, MT, TH, SI, span_y, span_x) <= PATCH_LDS_BUDGET &&
                        (long)kc * PH * PW <= 256L * PATCH_MAX_PER_THREAD))
            continue;
        const long padded = (long)((C + kc - 1) / kc) * kc;      // channels actually multiplied (zero padded tail)
        // fewer wasted channels first; among equals the larger chunk (fewer barriers), but not below a 32-deep K chunk
        if (best_pad < 0 || padded < best_pad || (padded == best_pad && kc > best) || (kc * T >= 32 && best * T < 32 && padded <= best_pad + best_pad / 8)) {
            best = kc;
            best_pad = padded;
        }
    }
    const int kc = best;
    return kc;
}

int patch_geom_from(const IgemmGeom& f, PatchGeom& g) {
    g = PatchGeom{};
    g.N = f.N; g.C = f.C; g.IH = f.IH; g.IW = f.IW; g.M = f.M; g.OH = f.OH; g.OW = f.OW; g.SI = f.SI; g.SO = f.SO;
    g.nphase = f.nphase; g.reflect = f.reflect; g.act = f.act; g.slope = f.slope; g.wsm = f.wsm; g.wsc = f.wsc;
    for (int p = 0; p < 4; ++p) { g.py[p] = f.ph_py[p]; g.px[p] = f.ph_px[p]; g.gh[p] = f.ph_gh[p]; g.gw[p] = f.ph_gw[p]; }
    for (int p = 0; p < 5; ++p) g.t0[p] = f.ph_t0[p];
    long off = 0;
    g.Mpad = (g.M + 127) / 128 * 128;
    for (int p = 0; p < g.nphase; ++p) {
        const int T = g.t0[p + 1] - g.t0[p];
        int oy0 = 1 << 30, ox0 = 1 << 30, oy1 = -(1 << 30), ox1 = -(1 << 30);
        for (int t = g.t0[p]; t < g.t0[p + 1]; ++t) {
            const int oy = (f.taps[t] & 0xff) - 64, ox = ((f.taps[t] >> 8) & 0xff) - 64;
            oy0 = oy < oy0 ? oy : oy0; ox0 = ox < ox0 ? ox : ox0;
            oy1 = oy > oy1 ? oy : oy1; ox1 = ox > ox1 ? ox : ox1;
        }
        if (T == 0) { oy0 = ox0 = oy1 = ox1 = 0; }
        g.oy0[p] = oy0; g.ox0[p] = ox0; g.span_y[p] = oy1 - oy0; g.span_x[p] = ox1 - ox0;
        g.kc[p] = kc_for(T > 0 ? T : 1, g.C, g.M, g.SI, g.span_y[p], g.span_x[p]);
        g.pack_off[p] = off;
        const int nchunks = (g.C + g.kc[p] - 1) / g.kc[p];
        off += (long)nchunks * g.kc[p] * T * g.Mpad;
        for (int t = g.t0[p]; t < g.t0[p + 1]; ++t) {
            const int oy = (f.taps[t] & 0xff) - 64, ox = ((f.taps[t] >> 8) & 0xff) - 64, wi = f.taps[t] >> 16;
            g.taps[t] = (oy - oy0) | ((ox - ox0) << 8) | (wi << 16);
        }
    }
    for (int p = g.nphase; p < 5; ++p) g.pack_off[p] = off;
    return FAOCTASR_OK;
}

long patch_pack_floats(const PatchGeom& g) { return g.pack_off[4] + 256; }   // +1 KiB: LDS-DMA tail lanes stay in bounds

int launch_pack(const float* w, float* wp, const PatchGeom& g, hipStream_t s) {
    const long total = g.pack_off[4];
    if (total <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)pack_job_blocks(total / g.Mpad)), dim3(256), 0, s, w, wp, g);
    return check_launch("conv_pack");
}

struct PatchCfg { int WM, WN, MI, NI; };

template <int WM, int WN, int MI, int NI>
static size_t patch_lds_bytes(const PatchGeom& g, int SI) {
    constexpr int MT = WM * MI * 32, TH = WN * NI;
    size_t best = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const size_t b = lds_need(g.kc[p], g.t0[p + 1] - g.t0[p], MT, TH, SI, g.span_y[p], g.span_x[p]);
        if (b > best) best = b;
    }
    return best;
}

template <int WM, int WN, int MI, int NI>
static bool patch_fits(const PatchGeom& g, int SI) {
    constexpr int TH = WN * NI;
    for (int p = 0; p < g.nphase; ++p) {
        const int PH = (TH - 1) * SI + g.span_y[p] + 1, PW = 31 * SI + g.span_x[p] + 1;
        if ((long)g.kc[p] * PH * PW > 256L * PATCH_MAX_PER_THREAD) return false;
    }
    return patch_lds_bytes<WM, WN, MI, NI>(g, SI) <= PATCH_LDS_BUDGET;
}

template <int WM, int WN, int MI, int NI>
static long patch_blocks(const PatchGeom& g) {
    constexpr int MT = WM * MI * 32, TH = WN * NI;
    long mx = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const long t = (long)g.N * ((g.gw[p] + 31) / 32) * ((g.gh[p] + TH - 1) / TH);
        if (t > mx) mx = t;
    }
    return mx * ((g.M + MT - 1) / MT) * g.nphase;
}

template <int WM, int WN, int MI, int NI>
static int launch_cfg(const float* x, const float* wp, const float* bias, float* y, const PatchGeom& g, int ksplit, hipStream_t s) {
    constexpr int MT = WM * MI * 32, TH = WN * NI;
    long mx = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const long t = (long)g.N * ((g.gw[p] + 31) / 32) * ((g.gh[p] + TH - 1) / TH);
        if (t > mx) mx = t;
    }
    if (mx == 0) return FAOCTASR_OK;
    if (ksplit > 1) {
        hipError_t e = hipMemsetAsync(y, 0, sizeof(float) * (size_t)g.N * g.M * g.OH * g.OW, s);
        if (e != hipSuccess) return fail(FAOCTASR_EHIP, "memset y: %s", hipGetErrorString(e));
    }
    dim3 grid((unsigned)mx, (g.M + MT - 1) / MT, g.nphase * ksplit);
    const size_t lds = patch_lds_bytes<WM, WN, MI, NI>(g, g.SI);
    // dense tap grids with a known chunk size take the specialisations with compile-time taps
    int dkw = 0, dord = 0;                                              // grid width, +1 ascending / -1 descending
    {
        const int T0 = g.t0[1] - g.t0[0];
        int kw = 0;
        while (kw * kw < T0) ++kw;
        bool ok = kw * kw == T0 && kw >= 2;
        bool asc = true, desc = true;
        for (int p = 0; ok && p < g.nphase; ++p) {
            ok = g.t0[p + 1] - g.t0[p] == T0 && g.span_x[p] == kw - 1 && g.span_y[p] == kw - 1 && g.kc[p] == g.kc[0];
            for (int t = 0; ok && t < T0; ++t) {
                const int ty = g.taps[g.t0[p] + t] & 0xff, tx = (g.taps[g.t0[p] + t] >> 8) & 0xff;
                asc = asc && ty == t / kw && tx == t % kw;
                desc = desc && ty == kw - 1 - t / kw && tx == kw - 1 - t % kw;
            }
        }
        if (ok && (asc || desc)) { dkw = kw; dord = asc ? 1 : -1; }
    }
    auto launch = [&](auto k) {
        lds_optin((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, x, wp, bias, y, g, ksplit);
    };
    bool done = false;
    if constexpr (WM == 1 && WN == 4 && MI == 2 && NI == 2) {          // config B
        if (g.SI == 1 && dkw == 7 && g.kc[0] == 2) {
            if (dord > 0) launch(igemm_patch_kernel<WM, WN, MI, NI, 1, 7, 2>);
            else launch(igemm_patch_kernel<WM, WN, MI, NI, 1, -7, 2>);
            done = true;
        } else if (g.SI == 1 && dkw == 2 && g.kc[0] == 16) {            // the four 2x2 phases of ConvTranspose2d(4, stride 2)
            if (dord > 0) launch(igemm_patch_kernel<WM, WN, MI, NI, 1, 2, 16>);
            else launch(igemm_patch_kernel<WM, WN, MI, NI, 1, -2, 16>);
            done = true;
        }
    }
    if constexpr (WM == 2 && WN == 2 && MI == 2 && NI == 2) {          // config A
        if (g.SI == 2 && dkw == 4 && dord > 0 && g.kc[0] == 2) {        // Conv2d(4, stride 2): discriminator stages, convT input gradient
            launch(igemm_patch_kernel<WM, WN, MI, NI, 2, 4, 2>);
            done = true;
        } else if (g.SI == 2 && dkw == 3 && dord > 0 && g.kc[0] == 4) { // Conv2d(3, stride 2): the generator's down-sampling stages
            launch(igemm_patch_kernel<WM, WN, MI, NI, 2, 3, 4>);
            done = true;
        } else if (g.SI == 1 && dkw == 2 && g.kc[0] == 8) {             // 2x2 phases: input gradient of Conv2d(4, stride 2) with > 64 channels
            if (dord > 0) launch(igemm_patch_kernel<WM, WN, MI, NI, 1, 2, 8>);
            else launch(igemm_patch_kernel<WM, WN, MI, NI, 1, -2, 8>);
            done = true;
        }
    }
    if (done) {
    } else if (g.SI == 1) {
        auto k = igemm_patch_kernel<WM, WN, MI, NI, 1>;
        lds_optin((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, x, wp, bias, y, g, ksplit);
    } else {
        auto k = igemm_patch_kernel<WM, WN, MI, NI, 2>;
        lds_optin((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, x, wp, bias, y, g, ksplit);
    }
    return check_launch("igemm_patch");
}

// split K when the tile grid alone cannot fill the chip or lands badly on it.  Residency is 2 blocks per CU (LDS budget) =
// 512 slots, so a launch costs ceil(blocks * ks / 512) / ks rounds of full-K work: 128 blocks want ks = 4 (one full round of
// quarter-K blocks), and 680 blocks (the 7x7 input gradient on its 134x134 padded grid) want ks = 3 (3.98 rounds of thirds =
// 1.33 instead of 2).  Each split adds one atomic pass over the output, priced as ~30 K-steps per split relative to the K depth.
static int pick_ksplit(const PatchGeom& g, long blocks, int act) {
    if (g_no_split_k) return 1;        // FAOCTASR_CONV_NO_SPLIT_K: no atomics, bit-reproducible output
    if (act != FAOCTASR_ACT_NONE || blocks <= 0) return 1;
    int minchunks = 1 << 30, kdepth = 1 << 30;
    for (int p = 0; p < g.nphase; ++p) {
        const int nc = (g.C + g.kc[p] - 1) / g.kc[p];
        minchunks = nc < minchunks ? nc : minchunks;
        const int kd = g.C * (g.t0[p + 1] - g.t0[p]);
        kdepth = kd < kdepth ? kd : kdepth;
    }
    int best = 1;
    double best_cost = (double)((blocks + 511) / 512);
    const int ks_max = minchunks / 2 < 8 ? minchunks / 2 : 8;
    for (int ks = 2; ks <= ks_max; ++ks) {
        const double rounds = (double)((blocks * ks + 511) / 512) / ks;
        const double cost = rounds * (1.0 + 30.0 * ks / kdepth);
        if (cost < best_cost * 0.97) { best = ks; best_cost = cost; }
    }
    return best;
}

// returns 1 when the patch kernel was launched, 0 when the shape is left to the flat kernel, <0 on error
int launch_patch(const float* x, const float* wp, const float* bias, float* y, PatchGeom& g, int act, float slope, hipStream_t s) {
    g.act = act; g.slope = slope;
    if (g.SI != 1 && g.SI != 2) return 0;
    for (int p = 0; p < g.nphase; ++p)
        if (g.gw[p] < 24 || g.t0[p + 1] - g.t0[p] == 0) return 0;       // narrow maps: the flat kernel wastes fewer lanes
    // config A: 128 x (4x32); B: 64 x (8x32); C: 64 x (4x32)
    int rc;
    if (g.M > 64 && patch_fits<2, 2, 2, 2>(g, g.SI)) {
        rc = launch_cfg<2, 2, 2, 2>(x, wp, bias, y, g, pick_ksplit(g, patch_blocks<2, 2, 2, 2>(g), act), s);
    } else if (patch_fits<1, 4, 2, 2>(g, g.SI) && patch_blocks<1, 4, 2, 2>(g) >= 256) {
        rc = launch_cfg<1, 4, 2, 2>(x, wp, bias, y, g, pick_ksplit(g, patch_blocks<1, 4, 2, 2>(g), act), s);
    } else if (patch_fits<2, 2, 1, 2>(g, g.SI)) {
        rc = launch_cfg<2, 2, 1, 2>(x, wp, bias, y, g, pick_ksplit(g, patch_blocks<2, 2, 1, 2>(g), act), s);
    } else {
        return 0;
    }
    return rc == FAOCTASR_OK ? 1 : rc;
}

}  // namespace faoctasr
