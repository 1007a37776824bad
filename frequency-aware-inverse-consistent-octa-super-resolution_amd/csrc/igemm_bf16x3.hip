// Split-precision ("bf16x3") LDS-patch implicit GEMM for gfx950: fp32-parity convolutions on the bf16 matrix cores.
//
// Every fp32 operand is split into hi = bf16(x) and lo = bf16(x - hi); the product is accumulated in fp32 as
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (the dropped a_lo*b_lo term is ~2^-16 relative)
// with three v_mfma_f32_32x32x16_bf16 per tile step.  The bf16 MFMA runs 16x the f32 MFMA rate, so the contraction is
// ~5x faster than igemm_patch.hip at fp32-level accuracy: on the reference train step the losses move by <= 1.2e-4
// relative and the gradient norms by <= 2.3e-4 (DESIGN.md "bf16x3"), inside the 1e-3 parity bar; plain bf16 operands
// move them by 1.6e-2 / 12 % and are NOT offered.
//
// Structure (same tap-list geometry and phases as igemm_patch.hip):
//   * K is walked in groups of 16 input channels (the MFMA's K) x a tap group; the MFMA k index is the channel, so the
//     LDS patch is channel-innermost: P[plane][h = c/8][py][px][8 ch] bf16 -- a B fragment is one ds_read_b128.
//   * patch staging: each thread loads the 8 channels of a (pixel, h) item with coalesced dword buffer loads (OOB -> 0 for
//     padding / channel tail), splits to hi/lo with v_cvt_pk_bf16_f32 and writes two 16-byte vectors.
//   * weights are pre-packed as Wp[plane][phase][g16][tap][h][Mpad][8 ch] bf16; a (tap-group, 64-row) slab is a run of
//     1 KiB rows that LDS-DMA (global_load_lds_dwordx4) copies verbatim, double buffered against the MFMAs.
//   * block = 4 waves side by side along the pixel rows (64 output channels x TH x 32 pixels), fragment reads software
//     pipelined with hand-counted waits.
#include "common.h"
#include "igemm_geom.h"
#include "pack_bodies.h"
#include "split16.h"
#include <cstdlib>
#include <cstdint>
#include <type_traits>

#ifndef SP_NCW8
#define SP_NCW8 0                 // 1: large stride-1 layers on the 8-consumer-wave form (sp_launch_wide_block)
#endif
#ifndef SP_MIN_W
#define SP_MIN_W 24               // narrower output maps are left to the fp32 narrow-map kernels (a 32-pixel tile row would be mostly empty)
#endif
#ifndef SP_CDMA
#define SP_CDMA 0                 // (experiment) 1: the CONSUMER waves issue the weight LDS-DMA of the next slab, one row per tap step
#endif
#ifndef SP_SPLIT_BELOW
#define SP_SPLIT_BELOW 128        // grids with fewer blocks split the channel groups over the grid's z (atomics into a zeroed output).
                                  // Half a chip of blocks stays unsplit: the 256 -> 256 @32^2 layer alone is slower so (56 vs 46 us), the
                                  // step is faster (118.4 -> 121.8 img/s, two runs each): no memset, no 16 MB of fp32 atomics per call,
                                  // 16-byte epilogue stores, and the two generator chains' kernels share the chip
#endif
#ifndef SP_WIDE
#define SP_WIDE 1                 // 0: one dword store per accumulator register (round 1/2)
#endif
#ifndef SP_PROD_FIRST
#define SP_PROD_FIRST 0           // 1: waves 0-3 produce (the older wave of a SIMD wins issue arbitration), 4-7 consume
#endif
#ifndef SP_PRIO
#define SP_PRIO 0                 // s_setprio of the producer waves
#endif
#ifndef SP_STAGGER
#define SP_STAGGER 0              // (experiment) start phase step of the persistent blocks, in units of 64 cycles (s_sleep), x (block & 7)
#endif
#ifndef SP_TRACE
#define SP_TRACE 0                // diagnostics (tools/variants.py + tools/sp_trace.py): block 0 stamps s_memtime of its phases, slabs 16 .. 47
#endif
#if SP_TRACE
__device__ unsigned faoctasr_sp_trace_buf[4096];
extern "C" int faoctasr_sp_trace_read(unsigned* host_out, int n) {
    if (!host_out || n < 0 || n > 4096) return -1;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(faoctasr_sp_trace_buf), sizeof(unsigned) * (size_t)n) == hipSuccess ? 0 : -3;
}
#ifndef SP_TRACE_Q0
#define SP_TRACE_Q0 16            // first slab recorded (32 slabs fit); 0 for layers whose blocks run a single tile
#endif
#define SPTRACE(cond, base, q, k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (cond) && (q) >= SP_TRACE_Q0 && (q) < SP_TRACE_Q0 + 32) faoctasr_sp_trace_buf[(base) + ((q) - SP_TRACE_Q0) * 4 + (k)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define SPTRACE0(cond, slot) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (cond)) faoctasr_sp_trace_buf[(slot)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define SPTRACE(cond, base, q, k) do { } while (0)
#define SPTRACE0(cond, slot) do { } while (0)
#endif

namespace faoctasr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

#ifndef SP_EPI_FAST
#define SP_EPI_FAST 1             // straight-line wide epilogue for tiles with every output in range (0: the predicated form for every tile)
#endif
#ifndef SP_ABLATE
#define SP_ABLATE 0               // diagnostics (tools/variants.py): 1 no weight DMA, 2 no patch loads, 4 no split / patch stores, 8 no MFMA, 16 no output stores
#endif

constexpr int SP_MT = 64;         // output channels per block
static thread_local bool g_conv_residual_live = false;               // (host) the call being dispatched carries a fused residual
constexpr int SP_NPI = 6;         // (pixel, h) items staged per thread

__device__ __forceinline__ int reflect_idx_s(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

__device__ __forceinline__ void ds_read_v8(bf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for_sp(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_sp<I + 1, N>(f);
    }
}

// every LDS read retired; naming the registers keeps the consumer MFMAs below the wait
template <int MI, int NI>
__device__ __forceinline__ void wait_all(bf16x8 (&a)[MI][2], bf16x8 (&b)[NI][2]) {
    if constexpr (MI == 2 && NI == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
    else if constexpr (MI == 2 && NI == 1)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[0][0]), "+v"(b[0][1]));
    else if constexpr (MI == 1 && NI == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(b[0][0]), "+v"(b[0][1]));
}

// wait until only the OTHER fragment set's (MI + NI) * 2 reads are still outstanding; naming the registers keeps the consumer
// MFMAs below the wait
template <int MI, int NI>
__device__ __forceinline__ void wait_keep_next(bf16x8 (&a)[MI][2], bf16x8 (&b)[NI][2]) {
    if constexpr (MI == 2 && NI == 2)
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
    else if constexpr (MI == 2 && NI == 1)
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[0][0]), "+v"(b[0][1]));
    else if constexpr (MI == 1 && NI == 2)
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
    else
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(b[0][0]), "+v"(b[0][1]));
}

// ---------------------------------------------------------------------------------------------------------------
// weight packing: fp32 W (arbitrary m / c strides, tap list) -> two bf16 planes in the LDS image order
// ---------------------------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, const SplitGeom g) {
    split_pack_block<F16>(w, wp, g, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void split_absmax_kernel(const float* __restrict__ w, float* __restrict__ wp, const SplitGeom g) {
    __shared__ unsigned red[4];
    split_absmax_block(w, wp, g, (int)blockIdx.x, red);
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised 512-thread block: waves 0-3 are CONSUMERS (each 64 rows x NI pixel rows of 32: MI = 2 MFMA tiles high),
// waves 4-7 are PRODUCERS (patch loads, hi/lo split, LDS stores, weight LDS-DMA for the next slab).  The LDS footprint
// allows only one block per CU; without specialisation every wave is in the same phase at the same time and staging,
// fragment reads and MFMAs simply add up (measured by ablation).  With it each SIMD hosts one consumer and one producer wave
// whose pipes overlap.  The two roles run SEPARATE loops over the identical (tile, channel group, tap group) sequence and meet
// at one s_barrier per slab, so the register allocation is max(producer, consumer), not their sum.  Blocks are persistent over
// pixel tiles and the pipeline does not drain at tile boundaries.
// QUAD: the producers stage the patch in 4-pixel pieces (round 3): a thread's item is (channel half, patch row, 4 consecutive pixels
// starting at a 16-byte aligned image column) = 8 x buffer_load_dwordx4, one per channel, instead of 32 dword loads; the <= 3 + 3
// columns of a patch row left and right of the aligned run stay single-pixel items.  Host-side conditions: sp_quad_ok().
// NCW: consumer waves (4, or 8 = two per SIMD: a 16-row pixel tile per block at the 4-wave form's registers per wave -- the weight slab,
// 46 % of the block's memory traffic, then feeds twice the MFMAs; three waves per SIMD leave 168 registers each).
// F16: f16x2 operands (split16.h): the producers stage x * s(x_slot), the packed image holds w * s(w_slot), the epilogue divides
// the two scales out of the accumulators; everything else is the bf16x3 kernel.
template <int NI, int SI, bool QUAD, int NCW = 4, bool F16 = false>
__global__ __launch_bounds__(64 * (NCW + 4)) void igemm_bf16x3_kernel(const float* __restrict__ x, const __bf16* __restrict__ wp,
                                                           const float* __restrict__ bias, float* __restrict__ y, const SplitGeom g,
                                                           const int ksplit, const int wide, const unsigned* __restrict__ x_slot,
                                                           const unsigned* __restrict__ w_slot, const float* __restrict__ res) {
    constexpr int MI = 2, TH = NCW * NI, NT = 256, NPI = SP_NPI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SPTRACE0(tid == 0, 8);                                              // kernel entry
    const int l31 = lane & 31, lh = lane >> 5;
    const bool producer = (SP_PROD_FIRST && NCW == 4) ? wave < 4 : wave >= NCW;
    const int wn = NCW == 4 ? wave & 3 : (wave >= NCW ? wave - NCW : wave), stid = tid & 255;      // consumer index 0 .. NCW-1 / producer index 0 .. 3
    const int ph = blockIdx.z / ksplit, ks = blockIdx.z - ph * ksplit;
    const int GH = g.gh[ph], GW = g.gw[ph];
    const int tiles_x = (GW + 31) >> 5, tiles_y = (GH + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y;
    const long total_tiles = (long)g.N * tiles;
    if ((long)blockIdx.x >= total_tiles) return;
#if SP_STAGGER
    // The persistent blocks do identical work per slab and would run in step: every CU asking the fabric for its next patch in the
    // same microsecond, then none for the rest of the slab.  Start them at eight different phases of a slab instead.
    for (int i = (int)((blockIdx.x + blockIdx.y + blockIdx.z) & 7); i > 0; --i) __builtin_amdgcn_s_sleep(SP_STAGGER);
#endif
    const int t0 = g.t0[ph], T = g.t0[ph + 1] - t0, TG = g.tg[ph];
    const int ntg = (T + TG - 1) / TG;
    const int PH = (TH - 1) * SI + g.span_y[ph] + 1, PW = 31 * SI + g.span_x[ph] + 1, PHW = PH * PW;
    const int ngroups = (g.C + 15) >> 4;
    const int gps = (ngroups + ksplit - 1) / ksplit;
    const int g0 = ks * gps;
    const int g1 = (g0 + gps) < ngroups ? (g0 + gps) : ngroups;
    if (g0 >= g1) return;
    const int m0 = blockIdx.y * SP_MT;
    const int IH = g.IH, IW = g.IW;
    const long chw = (long)IH * IW;
    const unsigned a_bytes = (unsigned)TG * 2048u;                      // per plane per buffer: TG taps x 2 halves x 64 rows x 16 B
    const unsigned p_bytes = (unsigned)(2 * PHW) * 16u;                 // per plane per buffer
    const unsigned A_base = 0, P_base = 4u * a_bytes, S_base = P_base + 4u * p_bytes;       // S: 4 x 4 KiB of output staging
    auto tile_coords = [&](long tl, int& n, int& ty, int& tx) {       // total_tiles < 2^31 (sp_launch): 32-bit divisions
        n = (int)((unsigned)tl / (unsigned)tiles);
        const int rt = (int)((unsigned)tl - (unsigned)n * (unsigned)tiles);
        ty = (int)((unsigned)rt / (unsigned)tiles_x);
        tx = rt - ty * tiles_x;
    };

    if (producer) {
        // ================================================ PRODUCER ================================================
        if constexpr (SP_PRIO != 0) __builtin_amdgcn_s_setprio(SP_PRIO);
        constexpr unsigned OOB = 0x80000000u;
        const unsigned cstep = 4u * (unsigned)chw;
        const float* xin = x;
        float sx = 1.f;                                                  // f16x2: the activation tensor's scale (wave-uniform); read below, behind
                                                                         // the first patch's loads: it is not needed before the first split
        // ---- single-pixel items (all of the patch when !QUAD)
        constexpr int NSI = QUAD ? 2 : NPI;
        unsigned poff[NSI];
        float pv[NSI][8];
        // ---- 4-pixel items
        constexpr int NQI = QUAD ? 2 : 1;
        unsigned qoff[NQI];
        f32x4s qv[NQI][8];
        // Patch columns qs .. qs + 4 nq - 1 are 4-pixel pieces; qs is the first column whose image column is a multiple of 4 (tile
        // origins are), or -- zero padding -- the piece before it, so that the pieces cover the whole row and the columns they
        // have beyond it are simply not stored (a piece beyond the image row reads zeros: its offset is the out-of-range
        // sentinel).  A producer wave's memory instructions issue at 160-300 cycles apiece beside its SIMD's MFMA stream
        // (s_memtime trace), so what counts is their NUMBER: 3x3, 16 channels: 8 per wave instead of 48.
        const int qx0 = (-g.ox0[ph]) & 3;
        const bool cover = QUAD && !g.reflect;
        const int qs = cover && qx0 ? qx0 - 4 : qx0;
        const int nq = QUAD ? (cover ? (PW - qs + 3) >> 2 : (PW - qx0) >> 2) : 0;
        const int ns = cover ? 0 : PW - 4 * nq;                          // single columns per patch row: qx0 on the left, the rest on the right
        const int nqi = 2 * PH * nq, nitems = 2 * PH * ns;
        // item -> (h, patch row, patch column): fixed for the kernel; only the image offsets change with the tile
        int s_pos[NSI], q_pos[NQI];                                      // py | (px + 4) << 8 | h << 16, or -1
        int s_slot[NSI], q_slot[NQI];                                    // LDS slot (16-byte units) of the item's first pixel (a covering piece: may lie before its row)
        {
            const float inv_ns = 1.0f / (float)(ns > 0 ? ns : 1), inv_nq = 1.0f / (float)(nq > 0 ? nq : 1);
#pragma unroll
            for (int i = 0; i < NSI; ++i) {
                const int it = stid + NT * i;
                s_pos[i] = -1;
                s_slot[i] = 0;
                if (it < nitems) {
                    const int h = it >= PH * ns ? 1 : 0;
                    const int r = it - h * PH * ns;
                    const int py = (int)(((float)r + 0.5f) * inv_ns);
                    const int sx = r - py * ns;
                    const int px = sx < qx0 ? sx : sx + 4 * nq;
                    s_pos[i] = py | ((px + 4) << 8) | (h << 16);
                    s_slot[i] = h * PHW + py * PW + px;
                }
            }
#pragma unroll
            for (int i = 0; i < NQI; ++i) {
                const int it = stid + NT * i;
                q_pos[i] = -1;
                q_slot[i] = 0;
                if (QUAD && it < nqi) {
                    const int h = it >= PH * nq ? 1 : 0;
                    const int r = it - h * PH * nq;
                    const int py = (int)(((float)r + 0.5f) * inv_nq);
                    const int px = qs + 4 * (r - py * nq);
                    q_pos[i] = py | ((px + 4) << 8) | (h << 16);
                    q_slot[i] = h * PHW + py * PW + px;
                }
            }
        }
        if constexpr ((SP_ABLATE & 2) != 0) {
            for (int i = 0; i < NSI; ++i)
                for (int j = 0; j < 8; ++j) pv[i][j] = (float)(tid + i + j);
            for (int i = 0; i < NQI; ++i)
                for (int j = 0; j < 8; ++j) qv[i][j] = f32x4s{(float)tid, (float)i, (float)j, 1.f};
        }
        auto compute_poff = [&](long tl) {
            int n, ty, tx;
            tile_coords(tl, n, ty, tx);
            const int y_base = ty * TH * SI + g.oy0[ph], x_base = tx * 32 * SI + g.ox0[ph];
            xin = x + (long)n * g.C * chw;
            auto off_of = [&](int pos) -> unsigned {
                if (pos < 0) return OOB;
                int iy = y_base + (pos & 0xff), ix = x_base + ((pos >> 8) & 0xff) - 4;
                if (g.reflect) {
                    iy = reflect_idx_s(iy, IH);
                    ix = reflect_idx_s(ix, IW);                          // 4-pixel items: inside the row by sp_quad_ok, so this leaves them alone
                }
                if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) return 4u * (unsigned)(8 * (pos >> 16) * (int)chw + iy * IW + ix);
                return OOB;
            };
#pragma unroll
            for (int i = 0; i < NSI; ++i) poff[i] = off_of(s_pos[i]);
#pragma unroll
            for (int i = 0; i < NQI; ++i) qoff[i] = off_of(q_pos[i]);
        };
        auto load_patch = [&](int grp) {
            if constexpr ((SP_ABLATE & 2) != 0) return;
            const long bytes = (long)(g.C - grp * 16) * chw * 4;
            const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin + (long)grp * 16 * chw), 0,
                                                               (int)(bytes < 0x7ffffff0L ? bytes : 0x7ffffff0L), 0x00020000);
            if constexpr (QUAD) {
#pragma unroll
                for (int i = 0; i < NQI; ++i)
                    if (NT * i < nqi) {                                  // block-uniform: no instruction for an item row nobody has
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            qv[i][j] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(srd, qoff[i] + j * cstep, 0, 0));
                    }
            }
#pragma unroll
            for (int i = 0; i < NSI; ++i)
                if (NT * i < nitems) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        pv[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, poff[i] + j * cstep, 0, 0));
                }
        };
        auto store_patch = [&](int buf, auto&& hook) {                   // hook(): one weight DMA, if any is left (see the loop below)
            if constexpr ((SP_ABLATE & 4) != 0) return;
            char* hi_p = smem + P_base + (buf * 2 + 0) * p_bytes;
            char* lo_p = smem + P_base + (buf * 2 + 1) * p_bytes;
            if constexpr (QUAD) {
#pragma unroll
                for (int i = 0; i < NQI; ++i) {
                    if (NT * i < nqi) {                                  // block-uniform (hook() issues a DMA: every lane must be there);
                        const bool qok = q_pos[i] >= 0;                  // lanes without an item compute on zeros and store nothing
                        const int px0 = ((q_pos[i] >> 8) & 0xff) - 4;
                        // hi = bf16(x) of two CHANNELS of a pixel is one v_cvt_pk_bf16_f32 (the LDS order); x - float(hi) of two PIXELS
                        // of a channel is one v_pk_add_f32 on the register pair the dwordx4 load left them in: 2.5 instructions per
                        // element and no moves (pairing channels for the subtraction cost two v_mov per pair)
                        // one PIXEL PAIR at a time (32 temporaries instead of 64: the 8-consumer-wave form has 168 registers per wave)
#pragma unroll
                        for (int kp = 0; kp < 2; ++kp) {
                            unsigned hd[2][4], ld[2][4];                 // [pixel of the pair][channel pair]
                            if constexpr (F16) {
                                // f16x2: a channel pair of a pixel is four v_fma_mix*_f16 (scale, convert, widen, subtract, convert: split16.h)
#pragma unroll
                                for (int k = 0; k < 2; ++k) {
#pragma unroll
                                    for (int c2 = 0; c2 < 4; ++c2) {
                                        split_pair_scaled<true>(qv[i][2 * c2][2 * kp + k], qv[i][2 * c2 + 1][2 * kp + k], sx, hd[k][c2], ld[k][c2]);
                                        if (c2 & 1) hook();
                                    }
                                }
                            } else {
                            f32x2s lo[8];                                // [channel]
#pragma unroll
                            for (int k = 0; k < 2; ++k)
#pragma unroll
                                for (int c2 = 0; c2 < 4; ++c2)
                                    hd[k][c2] = cvt_pair<F16>(qv[i][2 * c2][2 * kp + k], qv[i][2 * c2 + 1][2 * kp + k]);
                            hook();
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                const unsigned d0 = hd[0][c >> 1], d1 = hd[1][c >> 1];
                                const f32x2s hf = {(c & 1) ? half_hi_f32<F16>(d0) : half_lo_f32<F16>(d0), (c & 1) ? half_hi_f32<F16>(d1) : half_lo_f32<F16>(d1)};
                                lo[c] = f32x2s{qv[i][c][2 * kp], qv[i][c][2 * kp + 1]} - hf;
                                if ((c & 3) == 3) hook();
                            }
#pragma unroll
                            for (int k = 0; k < 2; ++k)
#pragma unroll
                                for (int c2 = 0; c2 < 4; ++c2)
                                    ld[k][c2] = cvt_pair<F16>(lo[2 * c2][k], lo[2 * c2 + 1][k]);
                            }
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                if (qok && (unsigned)(px0 + 2 * kp + k) < (unsigned)PW) {       // not: a covering piece's columns beyond the patch row
                                    *reinterpret_cast<u32x4s*>(hi_p + (q_slot[i] + 2 * kp + k) * 16) = u32x4s{hd[k][0], hd[k][1], hd[k][2], hd[k][3]};
                                    *reinterpret_cast<u32x4s*>(lo_p + (q_slot[i] + 2 * kp + k) * 16) = u32x4s{ld[k][0], ld[k][1], ld[k][2], ld[k][3]};
                                }
                                hook();
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NSI; ++i) {
                if (s_pos[i] >= 0) {
                    u32x4s hv, lv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned h, l;
                        split_pair_scaled<F16>(pv[i][2 * j], pv[i][2 * j + 1], sx, h, l);
                        hv[j] = h;
                        lv[j] = l;
                    }
                    *reinterpret_cast<u32x4s*>(hi_p + s_slot[i] * 16) = hv;
                    *reinterpret_cast<u32x4s*>(lo_p + s_slot[i] * 16) = lv;
                }
            }
        };
        // A slab (group grp, tap group tgi) of packed weights -> LDS buffer abuf: per plane ntaps*2 rows of 1 KiB that LDS-DMA copies
        // verbatim.  Everything but the lane's 16 bytes inside the row is wave-uniform and kept on the scalar unit (with `wn` in a
        // vector register the compiler ran this loop under an exec mask with ~28 vector instructions per DMA).
        const int wnu = __builtin_amdgcn_readfirstlane(wn);
        const char* const wbase = reinterpret_cast<const char*>(wp + g.pack_off[ph] + (long)m0 * 8);
        const long w_row = (long)g.Mpad * 16, w_plane = g.plane_stride * 2;      // bytes between rows / planes
        const unsigned lane16 = (unsigned)lane * 16u;
        // One DMA at a time: the rows of the slab being fetched are handed out by next_row(); the split of the patch (vector ALU work)
        // calls it between its steps.  Memory instructions of this CU are accepted at ~25 B/clk whoever issues them (s_memtime:
        // 9 DMA + 8 loads of 1 KiB per wave took 2700 cycles with the MFMA waves parked at a barrier, 3400 beside them): issued in
        // a bunch a wave just sits in front of a full queue; with ~12 vector instructions between two of them the queue has
        // drained when the next one arrives.
        // (Round 4, measured and not kept: the row cursor advanced by additions instead of `plane * w_plane + rr * w_row` per DMA -- the ISA
        // goes from ~22 scalar instructions with three 64-bit multiplies per DMA to ~12 with none, and nothing changes: 64 -> 64 @256^2
        // 122.4-132.7 vs 127.2-129.9 us, the step 60.3-61.8 vs 60.6 ms.  The producers' phase is not bound by its scalar work.)
        const char* d_src = wbase;                                       // row cursor of the slab being fetched
        char* d_dst = smem;
        int d_r = 0, d_rows = 0, d_rows2 = 0;
        auto begin_A = [&](int grp, int tgi, int abuf) {
            const int tb = tgi * TG;
            int nt = T - tb;
            nt = nt < TG ? nt : TG;
            d_rows = nt * 2;                                            // (tap, h) rows per plane
            d_rows2 = ((SP_ABLATE & 1) || SP_CDMA) ? 0 : 2 * d_rows;
            d_r = wnu;
            d_src = wbase + ((long)grp * T + tb) * 2 * w_row;
            d_dst = smem + A_base + (abuf * 2) * a_bytes;
        };
        auto next_row = [&]() {
            if (d_r < d_rows2) {
                const int plane = d_r >= d_rows ? 1 : 0;
                const int rr = d_r - plane * d_rows;                    // tl*2 + h
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(d_src + plane * w_plane + rr * w_row + lane16),
                                                 (__attribute__((address_space(3))) void*)(d_dst + plane * a_bytes + rr * 1024), 16, 0, 0);
                d_r += 4;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto rest_of_A = [&]() {
            while (d_r < d_rows2) next_row();
        };

        // Patch pipeline, two patches deep (round 3).  Patch p = (tile, channel group) in the order the consumers walk them.  While the
        // consumers reduce patch p from LDS, patch p + 1 waits in registers and is split + stored at the LAST slab of p, and the loads
        // of patch p + 2 are issued right behind that store -- a whole slab before their use (round 1 issued them at the start of the
        // slab that stores them).  The weight DMA of the next slab is issued BEFORE the patch loads, so that `vmcnt(number of patch
        // loads)` -- loads return in order -- says "the DMA has landed" at the barrier without waiting for the patch.
        long tile_l = blockIdx.x;                                        // load cursor
        int grp_l = g0;
        bool valid_l = true;
        auto advance_l = [&]() {
            if (++grp_l >= g1) {
                grp_l = g0;
                tile_l += gridDim.x;
                valid_l = tile_l < total_tiles;
                if (valid_l) compute_poff(tile_l);
            }
        };
        int npl_full = 0;                                                // memory instructions of one load_patch (block-uniform)
        if constexpr ((SP_ABLATE & 2) == 0) {
            for (int i = 0; i < NQI; ++i) npl_full += (QUAD && NT * i < nqi) ? 8 : 0;
            for (int i = 0; i < NSI; ++i) npl_full += NT * i < nitems ? 8 : 0;
        }
        auto barrier_keep = [&](int n) {                                 // LDS stores visible, everything but the last n memory loads landed
            switch (n) {
                case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 16: asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 24: asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 32: asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 40: asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                case 48: asm volatile("s_waitcnt vmcnt(48) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            }
        };
        auto no_hook = [&]() {};

        SPTRACE0(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 12);   // producer set-up done
        compute_poff(tile_l);
        load_patch(grp_l);                                               // patch 0
        SPTRACE0(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 13);   // first patch's loads issued
        advance_l();
        begin_A(g0, 0, 0);
        if constexpr (SP_CDMA != 0) d_rows2 = (SP_ABLATE & 1) ? 0 : 2 * d_rows;      // slab 0 is the producers' in every variant
        rest_of_A();
        SPTRACE0(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 14);   // first weight slab's DMAs issued
        if constexpr (F16) sx = f16x2_scale(absmax_read(x_slot));
        store_patch(0, no_hook);
        int in_flight = 0;
        if (valid_l) {
            load_patch(grp_l);                                           // patch 1
            advance_l();
        }
        SPTRACE0(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 9);    // first patch split + stored, second patch's loads issued
        barrier_keep(0);
        SPTRACE0(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 10);   // ... and the first weight slab landed
        long tile = blockIdx.x;
        int slab = 0, pcount = 0;
        while (true) {
            const long next_tile = tile + gridDim.x;
            const bool has_next = next_tile < total_tiles;
            for (int grp = g0; grp < g1; ++grp, ++pcount) {
                const int pbuf = pcount & 1;
                const bool last_grp = grp + 1 >= g1;
                const bool more_patch = !last_grp || has_next;
                for (int tgi = 0; tgi < ntg; ++tgi, ++slab) {
                    const int abuf = slab & 1;
                    SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 3072, slab, 0);
                    in_flight = 0;
                    const bool swap_patch = tgi == ntg - 1 && more_patch;
                    d_rows2 = 0;
                    if (tgi + 1 < ntg) begin_A(grp, tgi + 1, abuf ^ 1);
                    else if (!last_grp) begin_A(grp + 1, 0, abuf ^ 1);
                    else if (has_next) begin_A(g0, 0, abuf ^ 1);
                    if (swap_patch) store_patch(pbuf ^ 1, next_row);     // patch p + 1 (loaded a slab or a channel group ago), DMA in between
                    SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 1024, slab, 0);
                    rest_of_A();
                    SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 1024, slab, 1);
                    if (swap_patch && valid_l) {
                        load_patch(grp_l);                               // patch p + 2
                        advance_l();
                        in_flight = npl_full;
                    }
                    SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 1024, slab, 2);
                    barrier_keep(in_flight);                             // LDS-DMA landed, patch visible, consumers done
                    SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 0 : 64 * NCW), 1024, slab, 3);
                }
            }
            if (!has_next) break;
            tile = next_tile;
        }
        return;
    }

    // ==================================================== CONSUMER ====================================================
    int tapv = 0;                                                       // lane t: tap t's offset in 16-byte pixel slots
    if (lane < T) {
        const int tp = g.taps[t0 + lane];
        tapv = (tp & 0xff) * PW + ((tp >> 8) & 0xff);
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const unsigned a_lane = (unsigned)(lh * 64 + l31) * 16u;                                     // + tl*2048, + mi*512
    const unsigned b_lane = (unsigned)((lh * PH + (wn * NI) * SI) * PW + l31 * SI) * 16u;         // + tap*16, + ni*SI*PW*16
    const unsigned b_row = (unsigned)(SI * PW) * 16u;

    // (SP_CDMA) weight rows of the NEXT slab, fetched by the consumers: wave wn takes rows wn, wn + 4, ...; one per tap step
    const int wnu_c = __builtin_amdgcn_readfirstlane(wn);
    const char* const wbase_c = reinterpret_cast<const char*>(wp + g.pack_off[ph] + (long)m0 * 8);
    const long w_row_c = (long)g.Mpad * 16, w_plane_c = g.plane_stride * 2;
    const unsigned lane16_c = (unsigned)lane * 16u;
    const char* c_src = wbase_c;
    char* c_dst = smem;
    int c_r = 0, c_rows = 0, c_rows2 = 0;
    auto c_begin = [&](int grp, int tgi, int abuf) {
        const int tb2 = tgi * TG;
        int nt2 = T - tb2;
        nt2 = nt2 < TG ? nt2 : TG;
        c_rows = nt2 * 2;
        c_rows2 = 2 * c_rows;
        c_r = wnu_c;
        c_src = wbase_c + ((long)grp * T + tb2) * 2 * w_row_c;
        c_dst = smem + A_base + (abuf * 2) * a_bytes;
    };
    auto c_next_row = [&]() {
        if (c_r < c_rows2) {
            const int plane = c_r >= c_rows ? 1 : 0;
            const int rr = c_r - plane * c_rows;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(c_src + plane * w_plane_c + rr * w_row_c + lane16_c),
                                             (__attribute__((address_space(3))) void*)(c_dst + plane * a_bytes + rr * 1024), 16, 0, 0);
            c_r += 4;
        }
    };

    float inv = 1.f;                                                    // f16x2: the operands were x * s_x and w * s_w
    if constexpr (F16) inv = f16x2_inv_scale(absmax_read(x_slot)) * f16x2_inv_scale(split_w_absmax(w_slot));
    long tile = blockIdx.x;
    asm volatile("s_barrier" ::: "memory");                             // first patch + first weight slab staged by the producers
    int slab = 0, pcount = 0;
    while (true) {
        const long next_tile = tile + gridDim.x;
        const bool has_next = next_tile < total_tiles;
        for (int grp = g0; grp < g1; ++grp, ++pcount) {
            const int pbuf = pcount & 1;
            for (int tgi = 0; tgi < ntg; ++tgi, ++slab) {
                const int abuf = slab & 1;
                const int tb = tgi * TG;
                int nt = T - tb;
                nt = nt < TG ? nt : TG;
                const unsigned Ah = lds0 + A_base + (abuf * 2 + 0) * a_bytes + a_lane, Al = Ah + a_bytes;
                const unsigned Ph = lds0 + P_base + (pbuf * 2 + 0) * p_bytes + b_lane, Pl = Ph + p_bytes;
                bf16x8 a0[MI][2], b0[NI][2], a1[MI][2], b1[NI][2];
                // (Round 4, measured and not kept: the reads issued in the order of their first use -- the lo*hi term's set, then the hi*lo
                // term's -- with one counted wait per term instead of one lgkmcnt(0) per step, so that the youngest read has 8 MFMAs of cover
                // instead of 4: correct, and within the noise on every layer (64 -> 64 @256^2 122.0 -> 120.9 us, 256 -> 256 @32^2 56.6 -> 58.4, step
                // 123.0-125.5 -> 122.8-125.8 img/s): the 20 % this loop runs over its MFMAs is not exposed LDS latency.)
                // One tap step = 3 MI NI MFMAs on the CURRENT fragment set with the 2 (MI + NI) ds_read_b128 of the NEXT tap's set
                // issued one per MFMA gap (round 3; the round-1 loop issued the eight reads in a bunch before the twelve MFMAs and
                // the matrix pipe idled for ~170 of every 560 cycles: s_memtime trace).  Terms outermost, so that
                // consecutive MFMAs go to different accumulators.
                auto rd_all = [&](bf16x8 (&a)[MI][2], bf16x8 (&b)[NI][2], int tl) {
                    const unsigned ao = (unsigned)tl * 2048u;
                    const unsigned bo = 16u * (unsigned)__builtin_amdgcn_readlane(tapv, (tb + tl) & 63);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        ds_read_v8(a[mi][0], Ah + ao + mi * 512u);
                        ds_read_v8(a[mi][1], Al + ao + mi * 512u);
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        ds_read_v8(b[ni][0], Ph + bo + ni * b_row);
                        ds_read_v8(b[ni][1], Pl + bo + ni * b_row);
                    }
                };
                auto step = [&](bf16x8 (&a)[MI][2], bf16x8 (&b)[NI][2], bf16x8 (&an)[MI][2], bf16x8 (&bn)[NI][2], int tl_next) {
                    const unsigned ao = (unsigned)tl_next * 2048u;       // past the slab on its last step: read, never used
                    const unsigned bo = 16u * (unsigned)__builtin_amdgcn_readlane(tapv, (tb + tl_next) & 63);
                    wait_all<MI, NI>(a, b);
                    static_for_sp<0, 3 * MI * NI>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int term = j / (MI * NI), t = j % (MI * NI), mi = t / NI, ni = t % NI;
                        if constexpr ((SP_ABLATE & 8) != 0) {
                            if constexpr (j == 0) acc[0][0][0] += (float)a[0][0][0] + (float)a[MI - 1][1][0] + (float)b[0][0][0] + (float)b[NI - 1][1][0];
                        } else {
                            // term 0: lo*hi, 1: hi*lo, 2: hi*hi
                            mfma16<F16>(a[mi][term == 0 ? 1 : 0], b[ni][term == 1 ? 1 : 0], acc[mi][ni]);
                        }
                        if constexpr (j < 2 * MI) ds_read_v8(an[j / 2][j % 2], (j % 2 ? Al : Ah) + ao + (j / 2) * 512u);
                        else if constexpr (j < 2 * (MI + NI)) ds_read_v8(bn[(j - 2 * MI) / 2][(j - 2 * MI) % 2], ((j - 2 * MI) % 2 ? Pl : Ph) + bo + ((j - 2 * MI) / 2) * b_row);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                };
                SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 256 : 0), 2048, slab, 0);
                if constexpr (SP_CDMA != 0) {
                    c_rows2 = 0;
                    const bool last_grp_c = grp + 1 >= g1;
                    if (tgi + 1 < ntg) c_begin(grp, tgi + 1, abuf ^ 1);
                    else if (!last_grp_c) c_begin(grp + 1, 0, abuf ^ 1);
                    else if (has_next) c_begin(g0, 0, abuf ^ 1);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                rd_all(a0, b0, 0);
                for (int tl = 0; tl < nt; tl += 2) {
                    if constexpr (SP_CDMA != 0) c_next_row();
                    step(a0, b0, a1, b1, tl + 1);
                    if (tl + 1 >= nt) break;
                    if constexpr (SP_CDMA != 0) c_next_row();
                    step(a1, b1, a0, b0, tl + 2);
                }
                if constexpr (SP_CDMA != 0)
                    while (c_r < c_rows2) c_next_row();
                SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 256 : 0), 2048, slab, 1);
                if constexpr (SP_CDMA != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the rows this wave fetched have landed
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // retire run-ahead reads before LDS is rewritten
                SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 256 : 0), 2048, slab, 2);
                asm volatile("s_barrier" ::: "memory");                  // no vmcnt wait: the tile's output stores drain behind the next tile's MFMAs
                SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 256 : 0), 2048, slab, 3);
            }
        }
        // ---- epilogue of this tile; the producers are already staging the next tile
        {
            SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 256 : 0), 3584, slab, 0);
            int cn, cty, ctx;
            tile_coords(tile, cn, cty, ctx);
            const int mrow0 = m0 + 4 * lh;
            if constexpr (F16) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) acc[mi][ni][rr] *= inv;
            }
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
            SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 256 : 0), 3584, slab, 1);
            // one image's M x OH x OW outputs are < 2^30 (sp_launch): a uniform image pointer + 32-bit lane offsets, one
            // multiply-add per element (round 1: a 64-bit address and a bounds branch per element, ~5700 cycles per tile)
            const unsigned ohw = (unsigned)(g.OH * g.OW);
            const int bo = ctx * 32 + l31;
            float* const yimg = y + (long)cn * g.M * (long)ohw;
            const float* const rimg = res ? res + (long)cn * g.M * (long)ohw : nullptr;
            const bool full_m = m0 + SP_MT <= g.M;
            if (wide) {
                // (Round 4, measured and not kept: swapping the MFMA operands -- (patch, weights) instead of (weights, patch), free because both
                // fragments have the same register layout -- makes an accumulator register quad 4 consecutive PIXELS of one channel, i.e. 16
                // contiguous output bytes, and the epilogue 16 global_store_dwordx4 per lane with no LDS transpose and no staging buffer
                // (216 -> 188 registers).  Correct (all split-kernel tests), and slower: 64 -> 64 @256^2 123.3 -> 129.0 us, 64 -> 128 @128^2
                // 67.6 -> 70.1 -- such a store instruction touches 32 lines with 32 bytes each instead of 8 whole lines, and the CU's memory
                // path charges per line; the ablations of the same round say where the time is instead: without ANY global load the layer
                // still takes 97 us of 133, i.e. the consumers' own loop + epilogue + barriers are 1.7x the 56 us of MFMA work.)
                // 32 x 32 accumulator tile -> the wave's 4 KiB of staging (a lane holds ONE pixel of 16 rows) -> a lane reads 4
                // pixels of a row back: 4 dwordx4 stores per tile, each 8 rows x 128 B, instead of 16 dword stores of 2 x 128 B.
                // The stores of a wave drain at its memory-instruction rate (s_memtime trace: 64 of them = the 8100-cycle tile
                // boundary); LDS accesses of one wave execute in order, the region is the wave's own: no barrier.
                float* const S = reinterpret_cast<float*>(smem + S_base) + wn * 1024;
                const int prow = lane >> 3, pq = lane & 7;
                const int ox = ctx * 32 + 4 * pq;
                // A tile with every output in range (wave-uniform test; all but the map's last row / column of tiles): straight-line code,
                // so that a block's 4 read-backs -- and its 4 residual loads -- are in flight together.  The predicated form below
                // compiles to one exec-masked region per store, each with its own LDS round trip and, with a residual, its own
                // global-load wait (ISA of round 4: 16 serial `ds_read_b128; s_waitcnt lgkmcnt(0); global_store` per tile, 3500 of the
                // epilogue's 4400 cycles in the s_memtime trace; this form 2300.  Measured and not kept: all four blocks' LDS round trips
                // issued back to back and then the 16 stores -- 3200-4400 cycles, a `global_store_dwordx4` issues in ~200 when the four
                // consumer waves store together, and the read-backs no longer hide under the previous block's stores).
                const bool interior = SP_EPI_FAST && (SP_ABLATE & 16) == 0 && NCW == 4 && full_m && ctx * 32 + 32 <= GW && cty * TH + wn * NI + NI <= GH;
                if (interior) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int ao = cty * TH + wn * NI + ni;
                        const unsigned o0 = (unsigned)(m0 + prow) * ohw + (unsigned)((ao + g.py[ph]) * g.OW + ox + g.px[ph]);
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {
                            f32x4s r[4], v[4];
                            if (res) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const f32x4s*>(rimg + (o0 + (unsigned)(mi * 32 + 8 * i) * ohw));
                            }
#pragma unroll
                            for (int rr = 0; rr < 16; ++rr) S[((rr & 3) + 8 * (rr >> 2) + 4 * lh) * 32 + l31] = acc[mi][ni][rr];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4s*>(S + (prow + 8 * i) * 32 + 4 * pq);
                            if (res) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] += r[i];
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4s*>(yimg + (o0 + (unsigned)(mi * 32 + 8 * i) * ohw)) = v[i];
                        }
                    }
                } else
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int ao = cty * TH + wn * NI + ni;
                    const bool ok = ao < GH && ox < GW;
                    const unsigned o0 = (unsigned)(m0 + prow) * ohw + (unsigned)((ao + g.py[ph]) * g.OW + ox + g.px[ph]);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) S[((rr & 3) + 8 * (rr >> 2) + 4 * lh) * 32 + l31] = acc[mi][ni][rr];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            f32x4s v = *reinterpret_cast<const f32x4s*>(S + (prow + 8 * i) * 32 + 4 * pq);
                            const int dm = mi * 32 + 8 * i;
                            if (ok && (full_m || m0 + prow + dm < g.M) && ((SP_ABLATE & 16) == 0 || v[0] == 123.456f)) {
                                if (res) v += *reinterpret_cast<const f32x4s*>(rimg + (o0 + (unsigned)dm * ohw));      // fused "+ residual" (faoctasr_conv_set_residual)
                                *reinterpret_cast<f32x4s*>(yimg + (o0 + (unsigned)dm * ohw)) = v;
                            }
                        }
                    }
                }
            } else
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int ao = cty * TH + wn * NI + ni;
                if (ao < GH && bo < GW) {
                    const unsigned o0 = (unsigned)mrow0 * ohw + (unsigned)((ao * g.SO + g.py[ph]) * g.OW + (bo * g.SO + g.px[ph]));
                    if (SP_EPI_FAST && (SP_ABLATE & 16) == 0 && full_m) {
                        // the (split-K, residual) case decided once, then straight-line code: the general form below compiles to one region per
                        // register with those branches inside and an `s_waitcnt vmcnt(0)` in front of EVERY store (it waits for the load the
                        // region may have issued -- and with it for every earlier store's acknowledgement: 64 serial round trips per tile.
                        // Transposed 128 -> 64 4x4 stride 2 @128^2: 191 -> 141-146 us; the step 63.0-64.3 -> 60.8-61.2 ms)
                        auto emit = [&](auto atomic_c, auto res_c) {
                            constexpr bool AT = decltype(atomic_c)::value, RS = decltype(res_c)::value;
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) {
                                float r[16];
                                if constexpr (RS) {
#pragma unroll
                                    for (int rr = 0; rr < 16; ++rr) r[rr] = rimg[o0 + (unsigned)(mi * 32 + (rr & 3) + 8 * (rr >> 2)) * ohw];
                                }
#pragma unroll
                                for (int rr = 0; rr < 16; ++rr) {
                                    float* const p = yimg + (o0 + (unsigned)(mi * 32 + (rr & 3) + 8 * (rr >> 2)) * ohw);
                                    float v = acc[mi][ni][rr];
                                    if constexpr (RS) v += r[rr];
                                    if constexpr (AT) atomicAdd(p, v);
                                    else *p = v;
                                }
                            }
                        };
                        const bool use_res = res && ks == 0;
                        if (ksplit > 1) {
                            if (use_res) emit(std::true_type{}, std::true_type{});
                            else emit(std::true_type{}, std::false_type{});
                        } else {
                            if (use_res) emit(std::false_type{}, std::true_type{});
                            else emit(std::false_type{}, std::false_type{});
                        }
                    } else
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) {
                            const int dm = mi * 32 + (rr & 3) + 8 * (rr >> 2);
                            if ((full_m || mrow0 + dm < g.M) && ((SP_ABLATE & 16) == 0 || acc[mi][ni][rr] == 123.456f)) {
                                float* const p = yimg + (o0 + (unsigned)dm * ohw);
                                const float rv = (res && ks == 0) ? rimg[o0 + (unsigned)dm * ohw] : 0.f;
                                if (ksplit > 1) atomicAdd(p, acc[mi][ni][rr] + rv);
                                else *p = acc[mi][ni][rr] + rv;
                            }
                        }
                    }
                }
            }
            SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 256 : 0), 3584, slab, 2);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            SPTRACE(tid == ((SP_PROD_FIRST && NCW == 4) ? 256 : 0), 3584, slab, 3);
        }
        if (!has_next) break;
        tile = next_tile;
    }
    SPTRACE0(tid == ((SP_PROD_FIRST && NCW == 4) ? 256 : 0), 11);          // consumer wave 0: last epilogue issued
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t SP_LDS_MAX = 156 * 1024;

static size_t sp_lds(const SplitGeom& g, int NI, int SI, bool staging = true, int ncw = 4) {      // NI: rows per tile / 4
    const int TH = 4 * NI;
    size_t best = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const int PH = (TH - 1) * SI + g.span_y[p] + 1, PW = 31 * SI + g.span_x[p] + 1;
        const size_t b = 4 * (size_t)g.tg[p] * 2048 + 4 * (size_t)(2 * PH * PW) * 16 + (staging ? 4096 * (size_t)ncw : 0) + 2048;      // weights, patch, output staging
        best = b > best ? b : best;
    }
    return best;
}

static bool sp_fits(const SplitGeom& g, int NI, int SI) {
    const int TH = 4 * NI;
    for (int p = 0; p < g.nphase; ++p) {
        const int PH = (TH - 1) * SI + g.span_y[p] + 1, PW = 31 * SI + g.span_x[p] + 1;
        if (2 * PH * PW > 256 * SP_NPI) return false;
    }
    return sp_lds(g, NI, SI, false) <= SP_LDS_MAX;                     // the output staging is optional (sp_launch)
}

// 1 when the layer can run on the split kernel (decided by the layer shape only)
int split_geom_from(const IgemmGeom& f, SplitGeom& g, int f16 = 0) {
    g = SplitGeom{};
    g.f16 = f16 ? 1 : 0;
    {
        const long a = (long)f.M * f.wsm, b = (long)f.C * f.wsc;
        g.w_elems = a > b ? a : b;
    }
    g.N = f.N; g.C = f.C; g.IH = f.IH; g.IW = f.IW; g.M = f.M; g.OH = f.OH; g.OW = f.OW; g.SI = f.SI; g.SO = f.SO;
    g.nphase = f.nphase; g.reflect = f.reflect; g.act = f.act; g.slope = f.slope; g.wsm = f.wsm; g.wsc = f.wsc;
    if (g.C < 16 || (g.SI != 1 && g.SI != 2)) return 0;
    for (int p = 0; p < 4; ++p) { g.py[p] = f.ph_py[p]; g.px[p] = f.ph_px[p]; g.gh[p] = f.ph_gh[p]; g.gw[p] = f.ph_gw[p]; }
    for (int p = 0; p < 5; ++p) g.t0[p] = f.ph_t0[p];
    g.Mpad = (g.M + 63) / 64 * 64;
    long off = 0;
    const int ngroups = (g.C + 15) / 16;
    for (int p = 0; p < g.nphase; ++p) {
        const int T = g.t0[p + 1] - g.t0[p];
        if (T == 0 || g.gw[p] < SP_MIN_W) return 0;
        int oy0 = 1 << 30, ox0 = 1 << 30, oy1 = -(1 << 30), ox1 = -(1 << 30);
        for (int t = g.t0[p]; t < g.t0[p + 1]; ++t) {
            const int oy = (f.taps[t] & 0xff) - 64, ox = ((f.taps[t] >> 8) & 0xff) - 64;
            oy0 = oy < oy0 ? oy : oy0; ox0 = ox < ox0 ? ox : ox0;
            oy1 = oy > oy1 ? oy : oy1; ox1 = ox > ox1 ? ox : ox1;
        }
        g.oy0[p] = oy0; g.ox0[p] = ox0; g.span_y[p] = oy1 - oy0; g.span_x[p] = ox1 - ox0;
        // tap group: all taps up to 9, otherwise the divisor-friendly group closest to 8 (7 for 7x7, 8 for 4x4)
        int tg = T;
        if (T > 9) {
            tg = 8;
            for (int cand = 9; cand >= 5; --cand)
                if (T % cand == 0) { tg = cand; break; }
        }
        g.tg[p] = tg;
        g.pack_off[p] = off;
        off += (long)ngroups * T * 16 * g.Mpad;
        for (int t = g.t0[p]; t < g.t0[p + 1]; ++t) {
            const int oy = (f.taps[t] & 0xff) - 64, ox = ((f.taps[t] >> 8) & 0xff) - 64, wi = f.taps[t] >> 16;
            g.taps[t] = (oy - oy0) | ((ox - ox0) << 8) | (wi << 16);
        }
    }
    for (int p = g.nphase; p < 5; ++p) g.pack_off[p] = off;
    g.plane_stride = (off + 7) & ~7L;
    if (!sp_fits(g, 2, g.SI) && !sp_fits(g, 1, g.SI)) return 0;
    return 1;
}

// 4-pixel patch items (kernel template QUAD): the image rows and the activation pointer are 16-byte aligned, so a piece that starts
// at a multiple-of-4 column lies wholly inside its row or wholly outside it; with reflection padding every piece of every tile must
// lie inside (a mirrored piece would need its pixels one by one); the item counts fit two rounds of the 256 producer threads.
#ifndef SP_QUAD
#define SP_QUAD 1                 // 0: single-pixel items only (round 1/2)
#endif
static bool sp_quad_ok(const SplitGeom& g, int NI, const float* x) {
    if (!SP_QUAD || (g.IW & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return false;
    const int TH = 4 * NI;
    for (int p = 0; p < g.nphase; ++p) {
        const int PH = (TH - 1) * g.SI + g.span_y[p] + 1, PW = 31 * g.SI + g.span_x[p] + 1;
        const int qx0 = (-g.ox0[p]) & 3;
        const bool cover = !g.reflect;
        const int qs = cover && qx0 ? qx0 - 4 : qx0;
        const int nq = cover ? (PW - qs + 3) >> 2 : (PW - qx0) >> 2, ns = cover ? 0 : PW - 4 * nq;
        if (nq < 1 || 2 * PH * nq > 512 || 2 * PH * ns > 512 || PH > 255 || PW > 240) return false;
        if (g.reflect) {
            const int tiles_x = (g.gw[p] + 31) / 32;
            if (g.ox0[p] + qx0 < 0 || (tiles_x - 1) * 32 * g.SI + g.ox0[p] + qx0 + 4 * nq > g.IW) return false;
        }
    }
    return true;
}

long split_pack_floats(const SplitGeom& g) { return g.plane_stride + 512; }   // 2 planes x 2 B = 4 B per element, + tail pad

int launch_split_pack(const float* w, float* wp, const SplitGeom& g, hipStream_t s) {
    const long total = g.pack_off[4];
    if (total <= 0) return FAOCTASR_OK;
    if (g.f16) {
        hipLaunchKernelGGL(split_absmax_kernel, dim3(SPLIT_WPARTS), dim3(256), 0, s, w, wp, g);
        hipLaunchKernelGGL(split_pack_kernel<true>, dim3((unsigned)pack_job_blocks(total / (8L * g.Mpad))), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp), g);
    } else {
        hipLaunchKernelGGL(split_pack_kernel<false>, dim3((unsigned)pack_job_blocks(total / (8L * g.Mpad))), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp), g);
    }
    return check_launch("split_pack");
}

template <int NI>
static int sp_launch(const float* x, const float* wp, const float* bias, float* y, const SplitGeom& g, hipStream_t s, const unsigned* x_slot,
                     const float* res) {
    constexpr int TH = 4 * NI;
    long mx = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const long t = (long)g.N * ((g.gw[p] + 31) / 32) * ((g.gh[p] + TH - 1) / TH);
        mx = t > mx ? t : mx;
    }
    if (mx == 0) return FAOCTASR_OK;
    if (mx >= (1L << 31) || (long)g.M * g.OH * g.OW >= (1L << 30)) return fail(FAOCTASR_EINVAL, "igemm_bf16x3: tensor too large");
    const int gy = (g.M + SP_MT - 1) / SP_MT;
    const long blocks = mx * gy * g.nphase;
    int ksplit = 1;
    const int ngroups = (g.C + 15) / 16;
    if (g.act == FAOCTASR_ACT_NONE && blocks < SP_SPLIT_BELOW) {        // one block per CU: fill the chip
        ksplit = (int)(256 / blocks);
        if (ksplit > ngroups / 2) ksplit = ngroups / 2;
        ksplit = ksplit < 1 ? 1 : ksplit;
    }
    if (g_no_split_k) ksplit = 1;
    if (ksplit > 1 && hipMemsetAsync(y, 0, sizeof(float) * (size_t)g.N * g.M * g.OH * g.OW, s) != hipSuccess)
        return fail(FAOCTASR_EHIP, "memset y failed");
    const bool staging = sp_lds(g, NI, g.SI, true) <= SP_LDS_MAX;
    const size_t lds = sp_lds(g, NI, g.SI, staging);
    // persistent blocks: one block per CU walks tiles bx, bx + gridDim.x, ... (the pipeline continues across tile boundaries)
    long nbx = 256 / ((long)gy * g.nphase * ksplit);
    nbx = nbx < 1 ? 1 : nbx;
    if (nbx > mx) nbx = mx;
    dim3 grid((unsigned)nbx, gy, g.nphase * ksplit);
    const bool quad = sp_quad_ok(g, NI, x);
    // 16-byte output pieces: unit output stride, rows and the tensor 16-byte aligned, whole pieces inside the grid, no atomics
    bool wide_ok = SP_WIDE && staging && g.SO == 1 && g.nphase == 1 && (g.OW & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && ksplit == 1;
    for (int p = 0; p < g.nphase; ++p) wide_ok = wide_ok && (g.gw[p] & 3) == 0 && (g.px[p] & 3) == 0;
    const int wide = wide_ok ? 1 : 0;
    const unsigned* w_slot = reinterpret_cast<const unsigned*>(wp) + split_scale_slot(g);
    auto go = [&](auto k) {
        lds_optin((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(512), lds, s, x, reinterpret_cast<const __bf16*>(wp), bias, y, g, ksplit, wide, x_slot, w_slot, res);
    };
    if (g.f16) {
        if (g.SI == 1) {
            if (quad) go(igemm_bf16x3_kernel<NI, 1, true, 4, true>);
            else go(igemm_bf16x3_kernel<NI, 1, false, 4, true>);
        } else {
            if (quad) go(igemm_bf16x3_kernel<NI, 2, true, 4, true>);
            else go(igemm_bf16x3_kernel<NI, 2, false, 4, true>);
        }
    } else if (g.SI == 1) {
        if (quad) go(igemm_bf16x3_kernel<NI, 1, true>);
        else go(igemm_bf16x3_kernel<NI, 1, false>);
    } else {
        if (quad) go(igemm_bf16x3_kernel<NI, 2, true>);
        else go(igemm_bf16x3_kernel<NI, 2, false>);
    }
    return check_launch(g.f16 ? "igemm_f16x2" : "igemm_bf16x3");
}

// Eight consumer waves (two per SIMD) + four producers: a 16-row x 32-pixel tile per block (512 pixels per 64 output channels).  Taken
// for stride-1 single-phase layers whose 512-pixel tiles still give every CU at least two tiles, when the 4-pixel patch items and the
// 16-byte epilogue apply; the tap group shrinks to <= 5 taps so that the two weight buffers leave room for the larger patch.
// (Round 4: tried again with the f16x2 operands and the straight-line epilogue -- 64 -> 64 @256^2 148.7 us against 128.7-134.9 for the 4 + 4 form,
// 64 -> 128 @128^2 73 against 66.5, the step 61.2-61.4 against 59.6-59.8 ms on that box: still off.)
static int sp_launch_wide_block(const float* x, const float* wp, const float* bias, float* y, SplitGeom g, hipStream_t s) {
    if (!SP_NCW8 || g.f16 || g_conv_residual_live || g.SI != 1 || g.nphase != 1) return 0;
    constexpr int TH = 16;
    const long tiles = (long)g.N * ((g.gw[0] + 31) / 32) * ((g.gh[0] + TH - 1) / TH);
    const int gy = (g.M + SP_MT - 1) / SP_MT;
    if (tiles * gy < 512 || tiles >= (1L << 31) || (long)g.M * g.OH * g.OW >= (1L << 30)) return 0;
    const int T = g.t0[1] - g.t0[0];
    int tg = g.tg[0];
    if (tg > 5) tg = T % 5 == 0 ? 5 : (T % 4 == 0 ? 4 : (T % 3 == 0 ? 3 : 5));
    g.tg[0] = tg;
    const size_t lds = sp_lds(g, 4, 1, true, 8);
    if (lds > SP_LDS_MAX || !sp_quad_ok(g, 4, x)) return 0;
    const bool wide_ok = SP_WIDE && g.SO == 1 && (g.OW & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (g.gw[0] & 3) == 0 && (g.px[0] & 3) == 0;
    if (!wide_ok) return 0;
    long nbx = 256 / gy;
    nbx = nbx < 1 ? 1 : nbx;
    if (nbx > tiles) nbx = tiles;
    auto k = igemm_bf16x3_kernel<2, 1, true, 8>;
    lds_optin((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nbx, gy, 1), dim3(768), lds, s, x, reinterpret_cast<const __bf16*>(wp), bias, y, g, 1, 1,
                       (const unsigned*)nullptr, (const unsigned*)nullptr, (const float*)nullptr);
    const int rc = check_launch("igemm_bf16x3 (8 consumer waves)");
    return rc == FAOCTASR_OK ? 1 : rc;
}

int launch_split(const float* x, const float* wp, const float* bias, float* y, SplitGeom& g, int act, float slope, hipStream_t s,
                 const unsigned* x_slot, const float* res) {
    g.act = act; g.slope = slope;
    {
        const int rc = sp_launch_wide_block(x, wp, bias, y, g, s);
        if (rc != 0) return rc < 0 ? rc : FAOCTASR_OK;
    }
    // (round 4, measured and not kept: 4-row tiles for the grids that leave half the chip idle -- the 256 -> 256 @32^2 trunk at batch 8: the layer
    // alone 45.0 -> 42.3 us, the step 126.3 / 121.8 -> 122.6 / 117.8 img/s, two runs each on one box: like split-K before it, filling the
    // chip with one chain's kernel takes the room the other chain's kernels were running in)
    if (sp_fits(g, 2, g.SI)) return sp_launch<2>(x, wp, bias, y, g, s, x_slot, res);
    return sp_launch<1>(x, wp, bias, y, g, s, x_slot, res);
}

int split_try(const IgemmGeom& f, const float* x, const float* w, const float* bias, float* y, int act, float slope, float* wpack,
              int wpack_state, hipStream_t s, PackJob* sink, int f16, const unsigned* x_slot, const float* res) {
    SplitGeom g;
    if (!split_geom_from(f, g, f16)) return 0;
    if (f16 && !sink && !x_slot) return fail(FAOCTASR_EINVAL, "precision 3 (f16x2) needs the gathered tensor's absmax slot: faoctasr_conv_set_scales");
    if (sink) {
        sink->type = PACK_SPLIT; sink->w = w; sink->wp = wpack; sink->g.split = g; sink->total = g.pack_off[4];
        return 1;
    }
    if (wpack_state == 1) {
        const int rc = launch_split_pack(w, wpack, g, s);
        if (rc) return rc;
    }
    g_conv_residual_live = res != nullptr;
    const int rc = launch_split(x, wpack, bias, y, g, act, slope, s, x_slot, res);
    g_conv_residual_live = false;
    return rc == FAOCTASR_OK ? 1 : rc;
}

long split_pack_floats_for(const IgemmGeom& f) {
    SplitGeom g;
    if (!split_geom_from(f, g)) return 0;
    return split_pack_floats(g);
}

}  // namespace faoctasr
