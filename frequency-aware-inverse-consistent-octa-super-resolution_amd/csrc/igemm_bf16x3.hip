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
#include <cstdlib>

namespace faoctasr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef SP_ABLATE
#define SP_ABLATE 0               // diagnostics (tools/variants.py): 1 no weight DMA, 2 no patch loads, 4 no split / patch stores, 8 no MFMA, 16 no output stores
#endif

constexpr int SP_MT = 64;         // output channels per block
constexpr int SP_NPI = 6;         // (pixel, h) items staged per thread

__device__ __forceinline__ int reflect_idx_s(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

__device__ __forceinline__ void ds_read_v8(bf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); }

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
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, const SplitGeom g) {
    split_pack_block(w, wp, g, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised 512-thread block: waves 0-3 are CONSUMERS (each 64 rows x NI pixel rows of 32: MI = 2 MFMA tiles high),
// waves 4-7 are PRODUCERS (patch loads, hi/lo split, LDS stores, weight LDS-DMA for the next slab).  The LDS footprint
// allows only one block per CU; without specialisation every wave is in the same phase at the same time and staging,
// fragment reads and MFMAs simply add up (measured by ablation).  With it each SIMD hosts one consumer and one producer wave
// whose pipes overlap.  The two roles run SEPARATE loops over the identical (tile, channel group, tap group) sequence and meet
// at one s_barrier per slab, so the register allocation is max(producer, consumer), not their sum.  Blocks are persistent over
// pixel tiles and the pipeline does not drain at tile boundaries.
template <int NI, int SI>
__global__ __launch_bounds__(512) void igemm_bf16x3_kernel(const float* __restrict__ x, const __bf16* __restrict__ wp,
                                                           const float* __restrict__ bias, float* __restrict__ y, const SplitGeom g,
                                                           const int ksplit) {
    constexpr int MI = 2, TH = 4 * NI, NT = 256, NPI = SP_NPI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const bool producer = wave >= 4;
    const int wn = wave & 3, stid = tid & 255;
    const int ph = blockIdx.z / ksplit, ks = blockIdx.z - ph * ksplit;
    const int GH = g.gh[ph], GW = g.gw[ph];
    const int tiles_x = (GW + 31) >> 5, tiles_y = (GH + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y;
    const long total_tiles = (long)g.N * tiles;
    if ((long)blockIdx.x >= total_tiles) return;
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
    const unsigned A_base = 0, P_base = 4u * a_bytes;
    auto tile_coords = [&](long tl, int& n, int& ty, int& tx) {
        n = (int)(tl / tiles);
        const int rt = (int)(tl - (long)n * tiles);
        ty = rt / tiles_x;
        tx = rt - ty * tiles_x;
    };

    if (producer) {
        // ================================================ PRODUCER ================================================
        constexpr unsigned OOB = 0x80000000u;
        unsigned poff[NPI];
        const int nitems = 2 * PHW;
        const float invPW = 1.0f / (float)PW;
        const unsigned cstep = 4u * (unsigned)chw;
        const float* xin = x;
        float pv[NPI][8];
        if constexpr ((SP_ABLATE & 2) != 0)
            for (int i = 0; i < NPI; ++i)
                for (int j = 0; j < 8; ++j) pv[i][j] = (float)(tid + i + j);
        auto compute_poff = [&](long tl) {
            int n, ty, tx;
            tile_coords(tl, n, ty, tx);
            const int y_base = ty * TH * SI + g.oy0[ph], x_base = tx * 32 * SI + g.ox0[ph];
            xin = x + (long)n * g.C * chw;
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                const int it = stid + NT * i;                           // item = h*PHW + pixel
                unsigned off = OOB;
                if (it < nitems) {
                    const int h = it >= PHW ? 1 : 0;
                    const int p = it - h * PHW;
                    const int py = (int)(((float)p + 0.5f) * invPW);
                    const int px = p - py * PW;
                    int iy = y_base + py, ix = x_base + px;
                    if (g.reflect) {
                        iy = reflect_idx_s(iy, IH);
                        ix = reflect_idx_s(ix, IW);
                    }
                    if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) off = 4u * (unsigned)(8 * h * (int)chw + iy * IW + ix);
                }
                poff[i] = off;
            }
        };
        auto load_patch = [&](int grp) {
            if constexpr ((SP_ABLATE & 2) != 0) return;
            const long bytes = (long)(g.C - grp * 16) * chw * 4;
            const auto srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin + (long)grp * 16 * chw), 0,
                                                               (int)(bytes < 0x7ffffff0L ? bytes : 0x7ffffff0L), 0x00020000);
#pragma unroll
            for (int i = 0; i < NPI; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    pv[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, poff[i] + j * cstep, 0, 0));
        };
        auto store_patch = [&](int buf) {
            if constexpr ((SP_ABLATE & 4) != 0) return;
            char* hi_p = smem + P_base + (buf * 2 + 0) * p_bytes;
            char* lo_p = smem + P_base + (buf * 2 + 1) * p_bytes;
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                const int it = stid + NT * i;
                if (it < nitems) {
                    bf16x8 hv, lv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = pv[i][j];
                        const __bf16 hb = (__bf16)v;
                        hv[j] = hb;
                        lv[j] = (__bf16)(v - (float)hb);
                    }
                    *reinterpret_cast<bf16x8*>(hi_p + it * 16) = hv;
                    *reinterpret_cast<bf16x8*>(lo_p + it * 16) = lv;
                }
            }
        };
        // A slab (group grp, tap group tgi) -> LDS buffer abuf: per plane ntaps*2 rows of 1 KiB
        auto issue_A = [&](int grp, int tgi, int abuf) {
            if constexpr ((SP_ABLATE & 1) != 0) return;
            const int tb = tgi * TG;
            int nt = T - tb;
            nt = nt < TG ? nt : TG;
            const int rows = nt * 2;                                    // (tap, h)
            for (int r = wn; r < 2 * rows; r += 4) {
                const int plane = r >= rows ? 1 : 0;
                const int rr = r - plane * rows;                        // tl*2 + h
                const __bf16* src = wp + plane * g.plane_stride + g.pack_off[ph] +
                                    ((((long)grp * T + tb) * 2 + rr) * g.Mpad + m0) * 8 + lane * 8;
                char* dst = smem + A_base + (abuf * 2 + plane) * a_bytes + rr * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        };

        long tile = blockIdx.x;
        compute_poff(tile);
        load_patch(g0);
        issue_A(g0, 0, 0);
        store_patch(0);
        __syncthreads();
        int slab = 0, pcount = 0;
        while (true) {
            const long next_tile = tile + gridDim.x;
            const bool has_next = next_tile < total_tiles;
            for (int grp = g0; grp < g1; ++grp, ++pcount) {
                const int pbuf = pcount & 1;
                const bool last_grp = grp + 1 >= g1;
                // prefetch the next patch (next channel group, or the first group of the NEXT tile) into registers
                if (!last_grp) {
                    load_patch(grp + 1);
                } else if (has_next) {
                    compute_poff(next_tile);
                    load_patch(g0);
                }
                const bool more_patch = !last_grp || has_next;
                for (int tgi = 0; tgi < ntg; ++tgi, ++slab) {
                    const int abuf = slab & 1;
                    if (tgi + 1 < ntg) issue_A(grp, tgi + 1, abuf ^ 1);
                    else if (!last_grp) issue_A(grp + 1, 0, abuf ^ 1);
                    else if (has_next) issue_A(g0, 0, abuf ^ 1);
                    if (tgi == ntg - 1 && more_patch) store_patch(pbuf ^ 1);
                    __syncthreads();                                     // LDS-DMA landed (vmcnt(0)), patch visible, consumers done
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

    long tile = blockIdx.x;
    __syncthreads();                                                    // first patch + first weight slab staged by the producers
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
                auto rd = [&](bf16x8 (&a)[MI][2], bf16x8 (&b)[NI][2], int tl) {
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
                auto mm = [&](bf16x8 (&a)[MI][2], bf16x8 (&b)[NI][2]) {
                    if constexpr ((SP_ABLATE & 8) != 0) {
                        acc[0][0][0] += (float)a[0][0][0] + (float)a[0][1][0] + (float)a[MI - 1][0][0] + (float)a[MI - 1][1][0] + (float)b[0][0][0] +
                                        (float)b[0][1][0] + (float)b[NI - 1][0][0] + (float)b[NI - 1][1][0];
                        return;
                    }
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], acc[mi][ni], 0, 0, 0);   // lo*hi
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], acc[mi][ni], 0, 0, 0);   // hi*lo
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], acc[mi][ni], 0, 0, 0);   // hi*hi
                        }
                };
                __builtin_amdgcn_s_waitcnt(0xC07F);
                rd(a0, b0, 0);
                for (int tl = 0; tl < nt; tl += 2) {
                    rd(a1, b1, tl + 1);                                 // runs past the slab on the last odd step: never used
                    wait_keep_next<MI, NI>(a0, b0);
                    mm(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tl + 1 >= nt) break;
                    rd(a0, b0, tl + 2);
                    wait_keep_next<MI, NI>(a1, b1);
                    mm(a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // retire run-ahead reads before LDS is rewritten
                __syncthreads();
            }
        }
        // ---- epilogue of this tile; the producers are already staging the next tile
        {
            int cn, cty, ctx;
            tile_coords(tile, cn, cty, ctx);
            const int mrow0 = m0 + 4 * lh;
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
            const int bo = ctx * 32 + l31;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int ao = cty * TH + wn * NI + ni;
                if (ao < GH && bo < GW) {
                    float* yo = y + (long)cn * g.M * ohw + (long)(ao * g.SO + g.py[ph]) * g.OW + (bo * g.SO + g.px[ph]);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) {
                            const int m = mrow0 + mi * 32 + (rr & 3) + 8 * (rr >> 2);
                            if (m < g.M && ((SP_ABLATE & 16) == 0 || acc[mi][ni][rr] == 123.456f)) {
                                if (ksplit > 1) atomicAdd(yo + (long)m * ohw, acc[mi][ni][rr]);
                                else yo[(long)m * ohw] = acc[mi][ni][rr];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }
        if (!has_next) break;
        tile = next_tile;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t SP_LDS_MAX = 156 * 1024;

static size_t sp_lds(const SplitGeom& g, int NI, int SI) {
    const int TH = 4 * NI;
    size_t best = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const int PH = (TH - 1) * SI + g.span_y[p] + 1, PW = 31 * SI + g.span_x[p] + 1;
        const size_t b = 4 * (size_t)g.tg[p] * 2048 + 4 * (size_t)(2 * PH * PW) * 16 + 2048;
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
    return sp_lds(g, NI, SI) <= SP_LDS_MAX;
}

// 1 when the layer can run on the split kernel (decided by the layer shape only)
int split_geom_from(const IgemmGeom& f, SplitGeom& g) {
    g = SplitGeom{};
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
        if (T == 0 || g.gw[p] < 24) return 0;
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

long split_pack_floats(const SplitGeom& g) { return g.plane_stride + 512; }   // 2 planes x 2 B = 4 B per element, + tail pad

int launch_split_pack(const float* w, float* wp, const SplitGeom& g, hipStream_t s) {
    const long total = g.pack_off[4];
    if (total <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)pack_job_blocks(total / (8L * g.Mpad))), dim3(256), 0, s, w, reinterpret_cast<__bf16*>(wp), g);
    return check_launch("split_pack");
}

template <int NI>
static int sp_launch(const float* x, const float* wp, const float* bias, float* y, const SplitGeom& g, hipStream_t s) {
    constexpr int TH = 4 * NI;
    long mx = 0;
    for (int p = 0; p < g.nphase; ++p) {
        const long t = (long)g.N * ((g.gw[p] + 31) / 32) * ((g.gh[p] + TH - 1) / TH);
        mx = t > mx ? t : mx;
    }
    if (mx == 0) return FAOCTASR_OK;
    const int gy = (g.M + SP_MT - 1) / SP_MT;
    const long blocks = mx * gy * g.nphase;
    int ksplit = 1;
    const int ngroups = (g.C + 15) / 16;
    if (g.act == FAOCTASR_ACT_NONE && blocks < 256) {                   // one block per CU: fill the chip
        ksplit = (int)(256 / blocks);
        if (ksplit > ngroups / 2) ksplit = ngroups / 2;
        ksplit = ksplit < 1 ? 1 : ksplit;
    }
    if (g_no_split_k) ksplit = 1;
    if (ksplit > 1 && hipMemsetAsync(y, 0, sizeof(float) * (size_t)g.N * g.M * g.OH * g.OW, s) != hipSuccess)
        return fail(FAOCTASR_EHIP, "memset y failed");
    const size_t lds = sp_lds(g, NI, g.SI);
    // persistent blocks: one block per CU walks tiles bx, bx + gridDim.x, ... (the pipeline continues across tile boundaries)
    long nbx = 256 / ((long)gy * g.nphase * ksplit);
    nbx = nbx < 1 ? 1 : nbx;
    if (nbx > mx) nbx = mx;
    dim3 grid((unsigned)nbx, gy, g.nphase * ksplit);
    if (g.SI == 1) {
        auto k = igemm_bf16x3_kernel<NI, 1>;
        lds_optin((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(512), lds, s, x, reinterpret_cast<const __bf16*>(wp), bias, y, g, ksplit);
    } else {
        auto k = igemm_bf16x3_kernel<NI, 2>;
        lds_optin((const void*)k, lds);
        hipLaunchKernelGGL(k, grid, dim3(512), lds, s, x, reinterpret_cast<const __bf16*>(wp), bias, y, g, ksplit);
    }
    return check_launch("igemm_bf16x3");
}

int launch_split(const float* x, const float* wp, const float* bias, float* y, SplitGeom& g, int act, float slope, hipStream_t s) {
    g.act = act; g.slope = slope;
    if (sp_fits(g, 2, g.SI)) return sp_launch<2>(x, wp, bias, y, g, s);
    return sp_launch<1>(x, wp, bias, y, g, s);
}

int split_try(const IgemmGeom& f, const float* x, const float* w, const float* bias, float* y, int act, float slope, float* wpack,
              int wpack_state, hipStream_t s, PackJob* sink) {
    SplitGeom g;
    if (!split_geom_from(f, g)) return 0;
    if (sink) {
        sink->type = PACK_SPLIT; sink->w = w; sink->wp = wpack; sink->g.split = g; sink->total = g.pack_off[4];
        return 1;
    }
    if (wpack_state == 1) {
        const int rc = launch_split_pack(w, wpack, g, s);
        if (rc) return rc;
    }
    const int rc = launch_split(x, wpack, bias, y, g, act, slope, s);
    return rc == FAOCTASR_OK ? 1 : rc;
}

long split_pack_floats_for(const IgemmGeom& f) {
    SplitGeom g;
    if (!split_geom_from(f, g)) return 0;
    return split_pack_floats(g);
}

}  // namespace faoctasr
