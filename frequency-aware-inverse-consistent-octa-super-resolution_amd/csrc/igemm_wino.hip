// Winograd F(2x2, 3x3) convolution for gfx950, fp32 throughout (f32 MFMA v_mfma_f32_16x16x4_f32).
//
// Serves every stride-1 3x3 gather of the path -- forward of the 3x3 convolutions (model.py:242-258, 403-421, 483-506) and
// their input gradients (a 3x3 correlation with the transposed, rotated filter) -- which is 52 % of the convolution time of
// the benchmark step.  Y = A^T [ (G g G^T) .* (B^T d B) ] A turns 36 multiply-adds per 2x2 output tile and channel pair into
// 16, i.e. 2.25x less MFMA work; all arithmetic stays fp32 (the transforms are exact +/- and *0.5, rounding grows by a small
// constant: per-layer error ~1e-6 relative, against the 1e-3 parity bar).
//
// Block = 64 output channels x 32 Winograd tiles (2 tile rows x 16 tile columns = 4 x 32 output pixels), 512 threads:
//   waves 4-7 PRODUCERS: waves 4-5 transform the input -- alternately, one wave a whole 8-channel chunk: lane = (two adjacent
//             tiles, channel pair), 48 buffer loads (zero padding = offset past the descriptor's range), B^T d B as packed
//             float2 arithmetic, 16 ds_write_b128 into V[xi][k][tile][j]; waves 6-7 copy the chunk's transformed-weight slab
//             U[xi][k][m] (32 KiB contiguous in the packed image) through registers;
//   waves 0-3 CONSUMERS: wave w owns output channels 16w..16w+15 x all 32 tiles x all 16 Winograd positions xi:
//             32 accumulator tiles of 16x16 (128 registers).  Because one lane holds all 16 xi of its (channel, tile)
//             elements, the output transform A^T M A, bias and activation run in registers and results go straight to HBM.
// Roles run separate loops over the same (tile, chunk) sequence and meet at one barrier per chunk; blocks are persistent
// over pixel tiles so the pipeline does not drain between tiles.
//
// LDS operand order: within a chunk the 8 channels are split as c = k + 4 j (k = 0..3 is the MFMA's K index = lane >> 4,
// j = 0, 1 the two MFMAs of the chunk); element (xi, k, row, j) sits at ((xi * 4 + k) * ROWS + (row ^ 16 (k & 1))) * 2 + j, so
// one ds_read_b64 per lane fetches the operands of both MFMAs.  ds_read_b64 is serviced in two groups of 32 lanes over 64
// banks: lanes 0-15 (k even) and 16-31 (k odd) of a group read 128 contiguous bytes each, and the XOR puts the odd-k rows'
// 16-row blocks in the other half of the 256-byte bank line -- without it every fragment read was a 2-way conflict
// (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE).
#include <type_traits>

#include "common.h"
#include "igemm_geom.h"
#include "pack_bodies.h"

#ifndef WINO_TRACE
#define WINO_TRACE 0       // diagnostics (tools/variants.py + tools/wino_trace.py): block 0 stamps s_memtime of its phases
#endif
#if WINO_TRACE
// The stamps go to a buffer of their own inside the code object (never through an operand pointer: round 1 wrote them behind
// `bias`, which is NULL or M floats long in every model layer) and are read back with faoctasr_wino_trace_read.
__device__ unsigned faoctasr_wino_trace_buf[4096];
extern "C" int faoctasr_wino_trace_read(unsigned* host_out, int n) {
    if (!host_out || n < 0 || n > 4096) return -1;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(faoctasr_wino_trace_buf), sizeof(unsigned) * (size_t)n) == hipSuccess ? 0 : -3;
}
#define WTRACE(cond, base, q, k, v) do { if (blockIdx.x == 0 && blockIdx.y == 0 && (cond) && (q) >= 16 && (q) < 48) faoctasr_wino_trace_buf[(base) + ((q) - 16) * 4 + (k)] = (v); } while (0)
#define WNOW() ((unsigned)__builtin_amdgcn_s_memtime())
#else
#define WTRACE(cond, base, q, k, v) do { } while (0)
#define WNOW() 0u
#endif
#ifndef WINO_ABLATE
#define WINO_ABLATE 0      // diagnostics, compile time (tools/variants.py): 1 no MFMA, 2 no fragment reads, 4 no patch loads, 8 no V stores, 16 no U DMA, 32 no output stores, 64 patch loads from one 4 KiB window (cache hits: wrong results, same instructions)
#endif
#ifndef WINO_UNT
#define WINO_UNT 0         // experiment: the weight slabs with non-temporal loads
#endif
#ifndef WINO_PROD_FIRST
#define WINO_PROD_FIRST 1  // 1: waves 0-3 produce, 4-7 consume (see `producer` in the kernel); 0: the other way round (round 2)
#endif
#ifndef WINO_ROT
#define WINO_ROT 0         // (experiment, no gain measured: 242-246 vs 229 us on c8) blocks walk the channel chunks in rotated order (see `rot` in the kernel); 0: all in the same order
#endif

namespace faoctasr {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

constexpr int WN_U_FLOATS = 16 * WN_KC * WN_MT;      // 8192 floats = 32 KiB per chunk
constexpr int WN_V_FLOATS = 16 * WN_KC * 32;         // 4096 floats = 16 KiB per chunk
constexpr unsigned WN_SENT = 0x40000000u;            // offset sentinel: beyond any admitted per-image extent

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// U = G g G^T, written in the order the kernel's weight waves want: [mtile][chunk][xi][k][m][j], c = chunk*8 + k + 4j (pack_bodies.h)
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ up, const WinoGeom g) {
    wino_pack_block(w, up, g, blockIdx.x, gridDim.x);
}

// ksplit > 1 (grid z): the channel chunks are divided over `ksplit` blocks per (tile range, channel tile); each applies the (linear)
// output transform to its partial sums and adds them into a zeroed y with fp32 atomics (no activation then; bias from split 0).
// For grids that cannot fill the chip by tiles alone: the 32 x 32 maps at batch 1-2 (256 channels = 32 chunks).
__global__ __launch_bounds__(512) void igemm_wino_kernel(const float* __restrict__ x, const float* __restrict__ up,
                                                         const float* __restrict__ bias, float* __restrict__ y, const WinoGeom g,
                                                         const int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const U_lds = reinterpret_cast<float*>(smem);               // 3 x WN_U_FLOATS (weights are fetched two slabs ahead)
    float* const V_lds = U_lds + 3 * WN_U_FLOATS;                      // 2 x WN_V_FLOATS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Which waves produce: the SIMD's issue arbiter serves its OLDEST wave first, and an MFMA stream always has its next instruction
    // queued -- with the consumers in waves 0-3 the co-resident producer wave made progress only while its consumer sat at the
    // barrier (s_memtime trace, DESIGN.md 4.1a: the transform took ~3000 cycles beside the MFMA loop whatever its instruction
    // count, ~400 alone; s_setprio did not change that).  Producers first: their few instructions issue when ready and the MFMAs
    // fill every other slot.
    const bool producer = WINO_PROD_FIRST ? wave < 4 : wave >= 4;
    const int T0 = WINO_PROD_FIRST ? 0 : 256, C0 = WINO_PROD_FIRST ? 256 : 0;      // first thread of the producers / consumers (trace stamps)
    (void)T0; (void)C0;
    const int wn = wave & 3;
    const int tiles_x = (g.OW + 31) >> 5, tiles_y = (g.OH + 3) >> 2;
    const int tiles = tiles_x * tiles_y;
    const long total_tiles = (long)g.N * tiles;
    if ((long)blockIdx.x >= total_tiles) return;
    const int mt = blockIdx.y;
    const int cps = (g.nchunks + ksplit - 1) / ksplit;                 // chunks per split
    const int ch_base = (int)blockIdx.z * cps;
    const int nchunks = g.nchunks - ch_base < cps ? g.nchunks - ch_base : cps;      // this block's chunks: ch_base .. ch_base + nchunks - 1
    if (nchunks <= 0) return;
    const long chw = (long)g.IH * g.IW;
    // The persistent blocks run in step, and with one chunk order they would all stream the SAME 32 KiB slab of transformed weights
    // at the same moment -- 32 CUs of an XCD asking its L2 for the same lines.  The reduction over channels does not care about
    // order, so block b starts at chunk rot(b): consecutive blocks of one XCD (ids equal mod 8) get different slabs.
    const int rot = WINO_ROT ? (int)((blockIdx.x >> 3) % (unsigned)nchunks) : 0;
    auto rot_chunk = [&](int ch) { const int c = ch + rot; return ch_base + (c >= nchunks ? c - nchunks : c); };
    auto tile_coords = [&](long tl, int& n, int& ty, int& tx) {
        n = (int)(tl / tiles);
        const int rt = (int)(tl - (long)n * tiles);
        ty = rt / tiles_x;
        tx = rt - ty * tiles_x;
    };

    if (producer) {
        // ================================================ PRODUCER ================================================
        // Slab q = (tile q / nchunks, chunk q % nchunks) of this block's tile sequence; the consumers work on slab q between
        // barrier(q-1) and barrier(q).  Two producer roles, so that each wave's memory counter tracks one kind of traffic:
        //   waves 4-5 (TRANSFORM), alternating slabs: in step q wave (q+1)&1 turns the patch of slab q+1 (registers, loaded TWO
        //       steps ago) into V[(q+1)&1] and then issues the loads of slab q+3 into the same registers.  With only plain loads
        //       outstanding the compiler counts vmcnt exactly (it must wait vmcnt(0) before an LDS store once an LDS-DMA is in
        //       flight in the same wave, which made the prefetch one step deep and left ~3000 cycles of HBM latency exposed per
        //       slab: s_memtime trace).
        //   waves 6-7 (WEIGHTS): the transformed-weight slab of slab q+2 into U[(q+2)%3] through registers (see below).
        __builtin_amdgcn_s_setprio(3);                                   // staging ahead of the consumers' MFMA stream in the issue arbiter
        const int stid = tid & 255;
        const int nslab_u = nchunks;
        const long my_tiles = (total_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
        const long q_total = my_tiles * nchunks;
        if (stid >= 128) {
            // ---------------------------------------------- weights ----------------------------------------------
            // Register staging (16 x global_load_dwordx4 -> 16 x ds_write_b128 per wave and slab), not LDS-DMA: back-to-back
            // global_load_lds_dwordx4 of ONE wave issue at ~177 cycles each (s_memtime trace: 2850 cycles for a wave's 16
            // rows, more than a consumer slab), plain loads pipeline.  Slab q+2 is stored during step q from registers loaded
            // during step q-1 (two slabs in flight per wave were tried: slower, the L2 -> CU path is the limit); the loop is peeled like the transform loop so that the compiler's vmcnt counts stay exact.
            const int w2 = (stid >> 6) & 1;                              // rows w2, w2+2, ... of the 32-row (1 KiB each) slab
            const float* const usrc = up + (long)mt * g.nchunks * WN_U_FLOATS + w2 * 256 + lane * 4;
            float* const udst = U_lds + w2 * 256 + lane * 4;
            f32x4w wr[16];
            auto load_u = [&](int ch) {
                if constexpr ((WINO_ABLATE & 16) != 0) return;
                const float* src = usrc + (long)rot_chunk(ch) * WN_U_FLOATS;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
#if WINO_UNT
                    wr[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4w*>(src + i * 512));
#else
                    wr[i] = *reinterpret_cast<const f32x4w*>(src + i * 512);
#endif
                }
            };
            auto store_u = [&](int buf) {
                if constexpr ((WINO_ABLATE & 16) != 0) return;
                float* dst = udst + buf * WN_U_FLOATS;
#pragma unroll
                for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4w*>(dst + i * 512) = wr[i];
            };
            int ch_u = 0, ubuf = 0;
            auto next_load_u = [&]() {
                if (++ch_u == nslab_u) ch_u = 0;
                load_u(ch_u);
            };
            auto next_store_u = [&]() {
                ubuf = ubuf == 2 ? 0 : ubuf + 1;
                store_u(ubuf);
            };
            auto barrier_u = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
            load_u(0);
            store_u(0);                                                  // slab 0 -> U[0]
            if (q_total > 1) {
                next_load_u();
                next_store_u();                                          // slab 1 -> U[1]
            }
            if (q_total > 2) next_load_u();                              // slab 2 in flight
            barrier_u();                                                 // barrier(-1)
            long q = 0;
            for (; q + 3 < q_total; ++q) {
                WTRACE(tid == T0 + 128, 3072, q, 0, WNOW());
                next_store_u();                                          // slab q+2
                WTRACE(tid == T0 + 128, 3072, q, 1, WNOW());
                next_load_u();                                           // slab q+3
                WTRACE(tid == T0 + 128, 3072, q, 2, WNOW());
                barrier_u();
                WTRACE(tid == T0 + 128, 3072, q, 3, WNOW());
            }
            for (; q < q_total; ++q) {
                if (q + 2 < q_total) next_store_u();
                if (q + 3 < q_total) next_load_u();
                barrier_u();
            }
            return;
        }
        // ------------------------------------------------ transform ------------------------------------------------
        // Waves 4 and 5 take the slabs alternately (wave w: slabs w, w+2, ...); one wave transforms a WHOLE slab: lane = (pair of
        // horizontally adjacent tiles, channel k) with the channels k and k+4 of the chunk packed as float2 operands.  Per slab and
        // wave: 48 dword loads of the shared 4 x 6 patch (0 address instructions: the per-tile offsets sit in 24 registers, the
        // chunk's channel advance in the buffer descriptors), 56 v_pk_add_f32 (B^T d per column once for both tiles: 24; (.) B per
        // tile: 32), 16 ds_write_b128 (V[xi][k][tile A, tile B][j] is 16 contiguous bytes) = 120 vector instructions, against
        // 2 waves x 145 of round 2's (tile, channel pair) threads (32 loads + 48 address adds + 24 moves + 32 packed adds + 8 stores
        // each).  On this chip a co-resident wave's vector instructions do not hide behind an f32 MFMA stream, they add to it
        // (DESIGN.md 4.1a), so the producers' instruction count per slab is what sets the slab period.
        // Schedule of wave w: in step q with (q + 1) & 1 == w it turns its registers (slab q+1, loaded two steps ago) into
        // V[(q+1) & 1] and refills them with slab q+3; in the other steps it only meets the barrier.
        const int tw = __builtin_amdgcn_readfirstlane(stid >> 6);        // 0 / 1; wave-uniform by construction, and the compiler has to know (descriptors in SGPRs)
        const int tp = lane & 15, kk = lane >> 4;                        // tile pair, channel k (and k + 4)
        const int tr = tp >> 3, tq = tp & 7;                             // tiles (tr, 2 tq) and (tr, 2 tq + 1) of the block's 2 x 16
        f32x4w* const vdst0 = reinterpret_cast<f32x4w*>(V_lds) + ((kk * 32 + ((tr * 16 + 2 * tq) ^ (16 * (kk & 1)))) >> 1);   // + xi*64, + buf*1024
        unsigned voff[4][6];                                             // byte offset of patch element (r, c) of channel k inside the image
        const float* ximg = x;
        f32x2 d[4][6];                                                   // [row][column] x (channel k, channel k+4)
        auto set_tile = [&](long tl) {
            int n, ty, tx;
            tile_coords(tl, n, ty, tx);
            const int iy0 = 4 * ty + 2 * tr + g.oy0, ix0 = 32 * tx + 4 * tq + g.ox0;
            unsigned ro[4], co[6];
#pragma unroll
            for (int r = 0; r < 4; ++r) ro[r] = (unsigned)(iy0 + r) < (unsigned)g.IH ? 4u * (unsigned)((iy0 + r) * g.IW) : WN_SENT;
#pragma unroll
            for (int c = 0; c < 6; ++c) co[c] = (unsigned)(ix0 + c) < (unsigned)g.IW ? 4u * (unsigned)(ix0 + c) : WN_SENT;
            const unsigned koff = 4u * (unsigned)(kk * (int)chw);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    voff[r][c] = ((ro[r] | co[c]) & WN_SENT) ? WN_SENT : ro[r] + co[c] + koff;
                    if constexpr ((WINO_ABLATE & 64) != 0) voff[r][c] = (voff[r][c] & 0xffcu) | ((unsigned)kk << 12);
                }
            ximg = x + (long)n * g.C * chw;
        };
        auto load_d = [&](int ch) {
            if constexpr ((WINO_ABLATE & 4) != 0) return;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // the descriptor starts at channel ch*8 + 4j and ends with the image: a lane whose channel k + 4j + 8 ch >= C reads 0
                const int c0 = (WINO_ABLATE & 64) ? 4 * j : rot_chunk(ch) * WN_KC + 4 * j;
                const int left = g.C - c0 > 0 ? g.C - c0 : 0;
                const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(((WINO_ABLATE & 64) ? x : ximg) + (long)c0 * chw), 0, (int)((long)left * chw * 4), 0x00020000);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c) d[r][c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, voff[r][c], 0, 0));
            }
        };
        auto store_v = [&](int buf) {
            if constexpr ((WINO_ABLATE & 8) != 0) return;
            f32x4w* vd = vdst0 + buf * (WN_V_FLOATS / 4);
            f32x2 t[4][6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {                                // B^T d, once per patch column (shared by the two tiles)
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                // (.) B per tile; one 16-byte store per Winograd position
                const f32x2 a0 = t[r][0] - t[r][2], b0 = t[r][2] - t[r][4];
                const f32x2 a1 = t[r][1] + t[r][2], b1 = t[r][3] + t[r][4];
                const f32x2 a2 = t[r][2] - t[r][1], b2 = t[r][4] - t[r][3];
                const f32x2 a3 = t[r][1] - t[r][3], b3 = t[r][3] - t[r][5];
                vd[(r * 4 + 0) * 64] = f32x4w{a0[0], a0[1], b0[0], b0[1]};
                vd[(r * 4 + 1) * 64] = f32x4w{a1[0], a1[1], b1[0], b1[1]};
                vd[(r * 4 + 2) * 64] = f32x4w{a2[0], a2[1], b2[0], b2[1]};
                vd[(r * 4 + 3) * 64] = f32x4w{a3[0], a3[1], b3[0], b3[1]};
            }
        };
        long tl_load = blockIdx.x;                                       // (tile, chunk) of this wave's load cursor: slabs tw, tw + 2, ...
        int ch_load = tw;
        auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        auto next_load = [&]() {                                         // two slabs on (nchunks >= 2)
            ch_load += 2;
            if (ch_load >= nchunks) {
                ch_load -= nchunks;
                tl_load += gridDim.x;
                set_tile(tl_load);
            }
            load_d(ch_load);
        };
        // prologue: wave 0 puts slab 0 into V[0] and fetches slab 2; wave 1 fetches slab 1 (stored in step 0)
        set_tile(tl_load);
        if (tw < q_total) load_d(ch_load);
        if (tw == 0) {
            store_v(0);
            if (2 < q_total) next_load();
        }
        barrier();                                                       // barrier(-1)
        for (long q = 0; q < q_total; ++q) {
            WTRACE(tid == T0, 1024, q, 0, WNOW());
            if (((int)(q + 1) & 1) == tw && q + 1 < q_total) {           // wave-uniform
                store_v(tw);                                             // slab q+1
                WTRACE(tid == T0, 1024, q, 1, WNOW());
                if (q + 3 < q_total) next_load();                        // slab q+3
            }
            WTRACE(tid == T0, 1024, q, 2, WNOW());
            barrier();
            WTRACE(tid == T0, 1024, q, 3, WNOW());
        }
        return;
    }

    // ==================================================== CONSUMER ====================================================
    f32x4w acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[xi][nt][r] = 0.f;
    const int l15 = lane & 15, lk = lane >> 4;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const int sw = 16 * (lk & 1);
    const unsigned ua0 = lds0 + (unsigned)((lk * 64 + ((16 * wn) ^ sw) + l15) * 8);                      // + xi*2048 + buf*32768
    const unsigned vb0 = lds0 + 3u * WN_U_FLOATS * 4u + (unsigned)((lk * 32 + sw + l15) * 8);             // tiles 0-15:  + xi*1024 + buf*16384
    const unsigned vc0 = lds0 + 3u * WN_U_FLOATS * 4u + (unsigned)((lk * 32 + (16 ^ sw) + l15) * 8);      // tiles 16-31
    const long ohw = (long)g.OH * g.OW;
    const bool pair_ok = (g.OW & 1) == 0;
    const int m_base = mt * WN_MT + 16 * wn + 4 * lk;                   // this lane's 4 output channels: the same for every tile
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (bias && blockIdx.z == 0 && m_base + r < g.M) ? bias[m_base + r] : 0.f;
    const bool split = ksplit > 1;

    // Finished tiles are not stored at once: the 16 (float2) results of a lane wait in registers and leave two per slab
    // during the next tile's reduction.  All blocks of the persistent grid run in step, so storing at the tile boundary would
    // hit HBM in bursts with every MFMA pipe idle behind a full store queue (measured: 19 % of the kernel).
    f32x2 pend[16];                                                     // item j = (nt = j >> 3, r = (j >> 1) & 3, a2 = j & 1)
    float* pbase = y;                                                   // &y[n][m_base][4 ty][ox] of the pending tile
    unsigned pmask = 0;                                                  // bit j: item j is inside the tensor
    bool px1 = false;                                                    // second pixel of the pair inside the row (odd OW)
    auto st = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int nt = j >> 3, r = (j >> 1) & 3, a2 = j & 1;
        if ((pmask & (1u << j)) && (WINO_ABLATE & 32) == 0) {
            float* p = pbase + r * ohw + (long)(2 * nt + a2) * g.OW;
            if (split) {
                atomicAdd(p, pend[j][0]);
                if (px1) atomicAdd(p + 1, pend[j][1]);
            } else if (pair_ok) {
                *reinterpret_cast<f32x2*>(p) = pend[j];
            } else {
                p[0] = pend[j][0];
                if (px1) p[1] = pend[j][1];
            }
        }
    };
    auto flush_from = [&](int k) {                                       // items k .. 15
        switch (k) {
            case 0: st(std::integral_constant<int, 0>{}); [[fallthrough]];
            case 1: st(std::integral_constant<int, 1>{}); [[fallthrough]];
            case 2: st(std::integral_constant<int, 2>{}); [[fallthrough]];
            case 3: st(std::integral_constant<int, 3>{}); [[fallthrough]];
            case 4: st(std::integral_constant<int, 4>{}); [[fallthrough]];
            case 5: st(std::integral_constant<int, 5>{}); [[fallthrough]];
            case 6: st(std::integral_constant<int, 6>{}); [[fallthrough]];
            case 7: st(std::integral_constant<int, 7>{}); [[fallthrough]];
            case 8: st(std::integral_constant<int, 8>{}); [[fallthrough]];
            case 9: st(std::integral_constant<int, 9>{}); [[fallthrough]];
            case 10: st(std::integral_constant<int, 10>{}); [[fallthrough]];
            case 11: st(std::integral_constant<int, 11>{}); [[fallthrough]];
            case 12: st(std::integral_constant<int, 12>{}); [[fallthrough]];
            case 13: st(std::integral_constant<int, 13>{}); [[fallthrough]];
            case 14: st(std::integral_constant<int, 14>{}); [[fallthrough]];
            case 15: st(std::integral_constant<int, 15>{}); [[fallthrough]];
            default: break;
        }
    };
    auto st_pair = [&](int ch) {                                         // items 2 ch, 2 ch + 1 (ch < 8)
        switch (ch) {
            case 0: st(std::integral_constant<int, 0>{}); st(std::integral_constant<int, 1>{}); break;
            case 1: st(std::integral_constant<int, 2>{}); st(std::integral_constant<int, 3>{}); break;
            case 2: st(std::integral_constant<int, 4>{}); st(std::integral_constant<int, 5>{}); break;
            case 3: st(std::integral_constant<int, 6>{}); st(std::integral_constant<int, 7>{}); break;
            case 4: st(std::integral_constant<int, 8>{}); st(std::integral_constant<int, 9>{}); break;
            case 5: st(std::integral_constant<int, 10>{}); st(std::integral_constant<int, 11>{}); break;
            case 6: st(std::integral_constant<int, 12>{}); st(std::integral_constant<int, 13>{}); break;
            case 7: st(std::integral_constant<int, 14>{}); st(std::integral_constant<int, 15>{}); break;
            default: break;
        }
    };

    long tl = blockIdx.x;
    asm volatile("s_barrier" ::: "memory");
    int slab = 0, ub = 0;
    while (true) {
        const long next_tile = tl + gridDim.x;
        const bool has_next = next_tile < total_tiles;
        for (int ch = 0; ch < nchunks; ++ch, ++slab) {
            const unsigned ua = ua0 + (unsigned)ub * (WN_U_FLOATS * 4u), vb = vb0 + (unsigned)(slab & 1) * (WN_V_FLOATS * 4u),
                           vc = vc0 + (unsigned)(slab & 1) * (WN_V_FLOATS * 4u);
            ub = ub == 2 ? 0 : ub + 1;
            WTRACE(tid == C0, 2048, slab, 0, WNOW());
            if (pmask) st_pair(ch);
            // fragment reads run two xi steps (8 MFMAs = 256 cycles) ahead of their use
            f32x2 a[3], b0[3], b1[3];
            if constexpr ((WINO_ABLATE & 2) != 0) {
                for (int i = 0; i < 3; ++i) a[i] = b0[i] = b1[i] = f32x2{(float)ua, (float)vb};
            } else {
            asm volatile("ds_read_b64 %0, %1" : "=v"(a[0]) : "v"(ua));
            asm volatile("ds_read_b64 %0, %1" : "=v"(b0[0]) : "v"(vb));
            asm volatile("ds_read_b64 %0, %1" : "=v"(b1[0]) : "v"(vc));
            asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(a[1]) : "v"(ua));
            asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(b0[1]) : "v"(vb));
            asm volatile("ds_read_b64 %0, %1 offset:1024" : "=v"(b1[1]) : "v"(vc));
            }
            static_for<0, 16>([&](auto ic) {
                constexpr int xi = decltype(ic)::value;
                constexpr int cur = xi % 3, nx2 = (xi + 2) % 3;
                if constexpr ((WINO_ABLATE & 2) != 0) {
                } else if constexpr (xi + 2 < 16) {
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[nx2]) : "v"(ua), "n"((xi + 2) * 2048));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b0[nx2]) : "v"(vb), "n"((xi + 2) * 1024));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b1[nx2]) : "v"(vc), "n"((xi + 2) * 1024));
                    asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[cur]), "+v"(b0[cur]), "+v"(b1[cur]));
                } else if constexpr ((WINO_ABLATE & 2) != 0) {
                } else if constexpr (xi + 1 < 16) {
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[cur]), "+v"(b0[cur]), "+v"(b1[cur]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur]), "+v"(b0[cur]), "+v"(b1[cur]));
                }
                if constexpr ((WINO_ABLATE & 1) == 0) {
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b0[cur][0], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b1[cur][0], acc[xi][1], 0, 0, 0);
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][1], b0[cur][1], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][1], b1[cur][1], acc[xi][1], 0, 0, 0);
                } else {
                    acc[xi][0][0] += a[cur][0] + b0[cur][0] + b1[cur][1];
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            WTRACE(tid == C0, 2048, slab, 1, WNOW());
            asm volatile("s_barrier" ::: "memory");                      // all LDS reads of this slab retired (lgkmcnt(0) above)
            WTRACE(tid == C0, 2048, slab, 2, WNOW());
        }
        // ---- output transform of this tile into the pending registers (the producers are already staging the next tile)
        {
            if (pmask && nchunks < 8) flush_from(2 * nchunks);           // short reductions: what the slabs did not get to
            int n, ty, tx;
            tile_coords(tl, n, ty, tx);
            const int ox = 32 * tx + 2 * l15;
            pbase = y + ((long)n * g.M + m_base) * ohw + (long)(4 * ty) * g.OW + ox;
            px1 = ox + 1 < g.OW;
            pmask = 0;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s0[4], s1[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        s0[b] = acc[0 + b][nt][r] + acc[4 + b][nt][r] + acc[8 + b][nt][r];
                        s1[b] = acc[4 + b][nt][r] - acc[8 + b][nt][r] - acc[12 + b][nt][r];
                    }
                    pend[nt * 8 + r * 2 + 0] = f32x2{s0[0] + s0[1] + s0[2] + bv[r], s0[1] - s0[2] - s0[3] + bv[r]};
                    pend[nt * 8 + r * 2 + 1] = f32x2{s1[0] + s1[1] + s1[2] + bv[r], s1[1] - s1[2] - s1[3] + bv[r]};
                    if (m_base + r < g.M && ox < g.OW) {
                        if (4 * ty + 2 * nt < g.OH) pmask |= 1u << (nt * 8 + r * 2);
                        if (4 * ty + 2 * nt + 1 < g.OH) pmask |= 1u << (nt * 8 + r * 2 + 1);
                    }
                }
            }
            // one activation dispatch per tile (inlining the switch -- with tanhf -- at each of the 64 values made the epilogue a
            // cold 7700-cycle instruction-cache walk: s_memtime trace)
            if (g.act == FAOCTASR_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 16; ++j) pend[j] = f32x2{fmaxf(pend[j][0], 0.f), fmaxf(pend[j][1], 0.f)};
            } else if (g.act == FAOCTASR_ACT_LRELU) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    pend[j] = f32x2{pend[j][0] > 0.f ? pend[j][0] : pend[j][0] * g.slope, pend[j][1] > 0.f ? pend[j][1] : pend[j][1] * g.slope};
            } else if (g.act == FAOCTASR_ACT_TANH) {
#pragma unroll
                for (int j = 0; j < 16; ++j) pend[j] = f32x2{tanhf(pend[j][0]), tanhf(pend[j][1])};
            }
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[xi][nt][r] = 0.f;
        }
        if (!has_next) break;
        tl = next_tile;
    }
    flush_from(0);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// 1 when the gather is a dense stride-1 3x3 correlation this kernel serves
static int wino_geom_from(const IgemmGeom& f, WinoGeom& g) {
    if (f.nphase != 1 || f.SI != 1 || f.SO != 1 || f.reflect || f.ph_t0[1] - f.ph_t0[0] != 9) return 0;
    if (f.C < 16 || f.M < 16 || f.ph_gw[0] < 24 || f.ph_gh[0] < 2) return 0;
    if ((long)f.C * f.IH * f.IW * 4 >= (long)WN_SENT) return 0;
    g = WinoGeom{};
    int oy0 = 1 << 30, ox0 = 1 << 30;
    for (int t = 0; t < 9; ++t) {
        const int oy = (f.taps[t] & 0xff) - 64, ox = ((f.taps[t] >> 8) & 0xff) - 64;
        oy0 = oy < oy0 ? oy : oy0;
        ox0 = ox < ox0 ? ox : ox0;
    }
    int seen = 0;
    for (int t = 0; t < 9; ++t) {
        const int i = (f.taps[t] & 0xff) - 64 - oy0, j = ((f.taps[t] >> 8) & 0xff) - 64 - ox0;
        if (i < 0 || i > 2 || j < 0 || j > 2) return 0;
        g.widx[i * 3 + j] = f.taps[t] >> 16;
        seen |= 1 << (i * 3 + j);
    }
    if (seen != 0x1ff) return 0;
    g.N = f.N; g.C = f.C; g.IH = f.IH; g.IW = f.IW; g.M = f.M; g.OH = f.OH; g.OW = f.OW;
    g.oy0 = oy0; g.ox0 = ox0; g.wsm = f.wsm; g.wsc = f.wsc;
    g.nchunks = (g.C + WN_KC - 1) / WN_KC;
    g.mtiles = (g.M + WN_MT - 1) / WN_MT;
    return 1;
}

static long wino_tiles(const WinoGeom& g) { return (long)g.N * ((g.OW + 31) / 32) * ((g.OH + 3) / 4); }

// the persistent grid wants at least half a chip of blocks; smaller problems stay on the split-K patch kernel
static bool wino_worth(const WinoGeom& g) { return wino_tiles(g) * g.mtiles >= 128; }

long wino_pack_floats_for(const IgemmGeom& f) {
    WinoGeom g;
    if (!wino_geom_from(f, g)) return 0;              // sizing is asked on a nominal grid: shape eligibility only
    return (long)g.mtiles * g.nchunks * WN_U_FLOATS + 256;
}

// 1 = enough tiles without splitting; otherwise the number of channel splits that brings the grid to >= 128 blocks (each split keeps
// >= 2 chunks and divides the chunk count), or 0 = leave the shape to the other kernels.
// WHETHER this kernel takes a shape must not depend on FAOCTASR_CONV_NO_SPLIT_K: the packed-weight image is recorded (pack plans)
// without that flag; under it the same shapes run unsplit here (one block owns an output element's whole reduction).
// ... nor on the fused activation: faoctasr_conv_pack_job records the image with FAOCTASR_ACT_NONE, so a call that fuses an
// activation (which the atomics of a split cannot apply) runs the same shapes unsplit here too instead of reading the Winograd
// image as a patch image on another route.
static int wino_ksplit(const WinoGeom& g, int act) {
    if (wino_worth(g)) return 1;
    const long blocks = wino_tiles(g) * g.mtiles;
    for (int ks = 2; ks <= 16; ks *= 2)
        if (g.nchunks % ks == 0 && g.nchunks / ks >= 2 && blocks * ks >= 128) return (g_no_split_k || act != FAOCTASR_ACT_NONE) ? 1 : ks;
    return 0;
}

int wino_try(const IgemmGeom& f, const float* x, const float* w, const float* bias, float* y, int act, float slope, float* wpack,
             int wpack_state, hipStream_t s, PackJob* sink) {
    WinoGeom g;
    if (!wino_geom_from(f, g)) return 0;
    const int ksplit = wino_ksplit(g, act);
    if (ksplit == 0) return 0;
    g.act = act; g.slope = slope;
    if (sink) {
        sink->type = PACK_WINO; sink->w = w; sink->wp = wpack; sink->g.wino = g;
        sink->total = (long)g.mtiles * g.nchunks * WN_U_FLOATS;
        return 1;
    }
    if (wpack_state == 1) {
        hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)pack_job_blocks((long)g.mtiles * g.nchunks)), dim3(256), 0, s, w, wpack, g);
        const int rc = check_launch("wino_pack");
        if (rc) return rc;
    }
    const long tiles = wino_tiles(g);
#ifndef WINO_CUS
#define WINO_CUS 256       // blocks of the persistent grid (experiment: fewer than the chip's 256 CUs, tools/variants.py)
#endif
    long nbx = WINO_CUS / g.mtiles;
    nbx = nbx < 1 ? 1 : nbx;
    nbx = nbx > tiles ? tiles : nbx;
    const size_t lds = (3 * (size_t)WN_U_FLOATS + 2 * (size_t)WN_V_FLOATS) * 4;
    auto k = igemm_wino_kernel;
    lds_optin((const void*)k, lds);
    if (ksplit > 1 && hipMemsetAsync(y, 0, sizeof(float) * (size_t)g.N * g.M * g.OH * g.OW, s) != hipSuccess) return fail(FAOCTASR_EHIP, "memset y failed");
    hipLaunchKernelGGL(k, dim3((unsigned)nbx, g.mtiles, (unsigned)ksplit), dim3(512), lds, s, x, wpack, bias, y, g, ksplit);
    const int rc = check_launch("igemm_wino");
    return rc == FAOCTASR_OK ? 1 : rc;
}

}  // namespace faoctasr
