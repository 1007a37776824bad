// Winograd F(2x2, 3x3), fp32, 64 output channels x 64 Winograd tiles per block: the large-grid form of igemm_wino.hip (round 3).
//
// Why a second kernel.  s_memtime traces of igemm_wino_kernel (profiles/r03_wino_phase_traces_*.log) showed its slab
// period set by two things its block shape fixes: (1) the 32 KiB slab of transformed filter per 8 channels arrives at ~12 B/clk per CU
// while 64 channels x 32 tiles consume 16 B/clk of it at full MFMA rate; (2) a producer wave that shares a SIMD with an MFMA stream
// gets about one instruction issued per MFMA, so wave specialisation cannot make the producers cheaper than their instruction count.
// Both ratios halve with 64 tiles per block, which needs 256 accumulator registers per wave -- so this kernel runs FOUR waves per block,
// one per SIMD (one block per CU by its 128 KiB of LDS), each with the 512-register budget of a lone wave: accumulators in AGPRs.
// There is no wave specialisation: a step of every wave is
//     producer phase : its quarter of the next slab's transformed filter (8 x ds_write_b128 from registers loaded one step ago), its
//                      quarter of the next slab's input transform (36 dwords loaded one step ago -> 28 v_pk_add_f32 -> 8 ds_write_b128),
//                      then the loads of the slab after that into the same registers (8 x buffer_load_dwordx4 + 36 x buffer_load_dword);
//     consumer phase : 128 MFMAs (16 Winograd positions x 4 tile groups x 2 k-steps) on the current slab, five ds_read_b64 per
//                      position issued two positions ahead, four deferred output stores of the previous tile;
//     one barrier.
// The producer phase runs at full issue rate (nothing to share the SIMD with) and is ~90 instructions against 4096 cycles of MFMA.
//
// Wave w: output channels 16 w .. 16 w + 15 of the block's 64, all 64 tiles, all 16 positions.  Input transform: tile half w & 1
// (tile rows 2 (w & 1), +1 of the block's 4 x 16 tiles), Winograd rows 2 (w >> 1), +1; lane = (pair of horizontally adjacent tiles,
// channel k and k + 4), as in igemm_wino.hip.  The packed filter image is igemm_wino.hip's ([m-tile][chunk][xi][k][m][j]), so one pack
// plan entry serves both kernels.  LDS: U 2 x 32 KiB, V 2 x 32 KiB ([xi][k][tile 64][j], tile ^ 16 (k & 1) as in the 32-tile kernel).
#include <type_traits>

#include "common.h"
#include "igemm_geom.h"

#ifndef W64_ABL
#define W64_ABL 0          // diagnostics (tools/variants.py): 1 no filter DMA, 2 no patch loads, 4 no transform / V stores, 8 no output stores, 16 no MFMA
#endif

namespace faoctasr {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

constexpr int W6_U_FLOATS = 16 * WN_KC * WN_MT;      // 8192 floats = 32 KiB per slab
constexpr int W6_V_FLOATS = 16 * WN_KC * 64;         // 8192 floats = 32 KiB per slab
constexpr unsigned W6_SENT = 0x40000000u;

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

}  // namespace

__global__ __launch_bounds__(256) void igemm_wino64_kernel(const float* __restrict__ x, const float* __restrict__ up,
                                                           const float* __restrict__ bias, float* __restrict__ y, const WinoGeom g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const U_lds = reinterpret_cast<float*>(smem);               // 2 x W6_U_FLOATS
    float* const V_lds = U_lds + 2 * W6_U_FLOATS;                      // 2 x W6_V_FLOATS
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave 0..3 (uniform, and the compiler has to know)
    const int tiles_x = (g.OW + 31) >> 5, tiles_y = (g.OH + 7) >> 3;
    const int tiles = tiles_x * tiles_y;
    const long total_tiles = (long)g.N * tiles;
    if ((long)blockIdx.x >= total_tiles) return;
    const int mt = blockIdx.y;
    const int nchunks = g.nchunks;
    const long chw = (long)g.IH * g.IW;
    const long my_tiles = (total_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int q_total = (int)(my_tiles * nchunks);
    auto tile_coords = [&](long tl, int& n, int& ty, int& tx) __attribute__((always_inline)) {
        n = (int)(tl / tiles);
        const int rt = (int)(tl - (long)n * tiles);
        ty = rt / tiles_x;
        tx = rt - ty * tiles_x;
    };

    // ================================================ producer state ================================================
    // ---- transformed filter: rows wv, wv + 4, ... of the 32-row (1 KiB each) slab
    // through registers (8 x buffer_load_dwordx4 one step ahead, 8 x ds_write_b128): back-to-back LDS-DMA instructions of one wave issue
    // at ~177 cycles each (1400 cycles of a step's producer phase, measured), plain loads at ~20
    const __amdgpu_buffer_rsrc_t usrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(up + (long)mt * nchunks * W6_U_FLOATS), 0,
                                                                          (int)((long)nchunks * W6_U_FLOATS * 4), 0x00020000);
    const unsigned uvoff = 4u * (unsigned)(wv * 256 + lane * 4);
    float* const udst = U_lds + wv * 256 + lane * 4;
    f32x4w wr[8];
    // ---- input transform
    const int hh = wv & 1, PP = wv >> 1;                               // tile half, Winograd row pair (rows 2 PP, 2 PP + 1)
    const int tp = lane & 15, kk = lane >> 4;
    const int tr = 2 * hh + (tp >> 3), tq = tp & 7;                    // tiles (tr, 2 tq) and (tr, 2 tq + 1) of the block's 4 x 16
    f32x4w* const vdst0 = reinterpret_cast<f32x4w*>(V_lds) + ((kk * 64 + ((tr * 16 + 2 * tq) ^ (16 * (kk & 1)))) >> 1) + (2 * PP) * 4 * 128;
    // (per position xi: 4 k x 64 tiles x 8 B = 2 KiB = 128 16-byte units; xi = 4 p + column, p = 2 PP + local phase)
    unsigned voff[3][6];                                               // patch rows PP, PP + 1, PP + 2
    const float* ximg = x;
    f32x2 d[3][6];
    int t_load = 0, ch_load = 0;                                       // load cursor: index into this block's tile sequence, chunk
    int t_set = -1;
    auto set_tile = [&](long tl) __attribute__((always_inline)) {
        int n, ty, tx;
        tile_coords(tl, n, ty, tx);
        // register slot s holds patch row rs[s]:  PP = 0: rows (0, 2, 1);  PP = 1: rows (2, 1, 3)  -- chosen so that both row pairs are
        // the SAME expressions of the slots (store_slab), with static register indices (a run-time row index sent d[] to scratch)
        const int iy0 = 8 * ty + 2 * tr + g.oy0, ix0 = 32 * tx + 4 * tq + g.ox0;
        const int rs[3] = {2 * PP, 2 - PP, 1 + 2 * PP};
        unsigned ro[3], co[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) ro[r] = (unsigned)(iy0 + rs[r]) < (unsigned)g.IH ? 4u * (unsigned)((iy0 + rs[r]) * g.IW) : W6_SENT;
#pragma unroll
        for (int c = 0; c < 6; ++c) co[c] = (unsigned)(ix0 + c) < (unsigned)g.IW ? 4u * (unsigned)(ix0 + c) : W6_SENT;
        const unsigned koff = 4u * (unsigned)(kk * (int)chw);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) voff[r][c] = ((ro[r] | co[c]) & W6_SENT) ? W6_SENT : ro[r] + co[c] + koff;
        ximg = x + (long)n * g.C * chw;
    };
    auto load_slab = [&]() __attribute__((always_inline)) {                                           // the cursor's slab: this wave's patch rows
        const int t = t_load < my_tiles ? t_load : 0;                  // slabs past the end re-read the first tile (nobody uses them)
        if (t != t_set) {
            t_set = t;
            set_tile(blockIdx.x + (long)t * gridDim.x);
        }
        if constexpr ((W64_ABL & 1) == 0) {
            const unsigned sbase = (unsigned)ch_load * (W6_U_FLOATS * 4u);
#pragma unroll
            for (int i = 0; i < 8; ++i) wr[i] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(usrd, uvoff, sbase + i * 4096u, 0));
        }
        if constexpr ((W64_ABL & 2) == 0)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // the descriptor starts at channel ch*8 + 4j and ends with the image: a lane whose channel k + 4j + 8 ch >= C reads 0
            const int c0 = ch_load * WN_KC + 4 * j;
            const int left = g.C - c0 > 0 ? g.C - c0 : 0;
            const __amdgpu_buffer_rsrc_t srd =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg + (long)c0 * chw), 0, (int)((long)left * chw * 4), 0x00020000);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) d[r][c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, voff[r][c], 0, 0));
        }
        if (++ch_load == nchunks) {
            ch_load = 0;
            ++t_load;
        }
    };
    auto store_slab = [&](int buf) __attribute__((always_inline)) {                                   // registers -> U[buf], V[buf]
        if constexpr ((W64_ABL & 1) == 0) {
            float* ud = udst + buf * W6_U_FLOATS;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4w*>(ud + i * 1024) = wr[i];
        }
        if constexpr ((W64_ABL & 4) != 0) return;
        f32x4w* vd = vdst0 + buf * (W6_V_FLOATS / 4);
        // Winograd rows p = 2 PP (local 0) and 2 PP + 1 (local 1).  B^T d: p 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3.  With the
        // slots of set_tile:  local 0 = slot0 - slot1 for both pairs;  local 1 = slot1 + sgn * slot2, sgn = +1 (PP = 0: d2 + d1), -1 (PP = 1:
        // d1 - d3)
        const float sg = PP == 0 ? 1.f : -1.f;
        const f32x2 sgn = {sg, sg};
#pragma unroll
        for (int lp = 0; lp < 2; ++lp) {
            f32x2 t[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) t[c] = lp == 0 ? d[0][c] - d[1][c] : d[1][c] + sgn * d[2][c];
            const f32x2 a0 = t[0] - t[2], b0 = t[2] - t[4];
            const f32x2 a1 = t[1] + t[2], b1 = t[3] + t[4];
            const f32x2 a2 = t[2] - t[1], b2 = t[4] - t[3];
            const f32x2 a3 = t[1] - t[3], b3 = t[3] - t[5];
            f32x4w* v = vd + lp * 4 * 128;
            v[0 * 128] = f32x4w{a0[0], a0[1], b0[0], b0[1]};
            v[1 * 128] = f32x4w{a1[0], a1[1], b1[0], b1[1]};
            v[2 * 128] = f32x4w{a2[0], a2[1], b2[0], b2[1]};
            v[3 * 128] = f32x4w{a3[0], a3[1], b3[0], b3[1]};
        }
    };

    // ================================================ consumer state ================================================
    // A tile's accumulators are born in its first slab (MFMA with C = 0) and die in its output transform: carrying zeroed
    // accumulators around the tile loop made the compiler permute all 256 AGPRs at every tile boundary through VGPR temporaries
    f32x4w acc[16][4];
    const int l15 = lane & 15, lk = lane >> 4;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const int sw = 16 * (lk & 1);
    const unsigned ua0 = lds0 + (unsigned)((lk * 64 + ((16 * wv) ^ sw) + l15) * 8);                      // + xi*2048 + buf*32768
    const unsigned va0 = lds0 + 2u * W6_U_FLOATS * 4u + (unsigned)((lk * 64 + sw + l15) * 8);             // tile groups 0 (+256: 2)
    const unsigned vb0 = lds0 + 2u * W6_U_FLOATS * 4u + (unsigned)((lk * 64 + (16 ^ sw) + l15) * 8);      // tile groups 1 (+256: 3)
    const long ohw = (long)g.OH * g.OW;
    const int OW = g.OW;
    const bool pair_ok = (g.OW & 1) == 0;
    const int m_base = mt * WN_MT + 16 * wv + 4 * lk;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (bias && m_base + r < g.M) ? bias[m_base + r] : 0.f;

    // Output: dwordx2 stores straight from the accumulator layout (16 lanes x 8 B per channel row) drain at ~250 cycles per wave
    // instruction here -- 128 of them per tile and wave, and since every wave also waits for its own loads (vmcnt counts stores as
    // well) they cannot trickle out in the background as in the 32-tile kernel.  The tile's results are transposed through LDS instead
    // (the V buffer the last slab has just released; 8 KiB per wave and half tile) and leave as 16 dwordx4 stores per wave and tile,
    // each 8 full 128-byte rows.
    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ================================================ prologue ================================================
    load_slab();                                                        // slab 0
    store_slab(0);
    load_slab();                                                        // slab 1 (stored in step 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();

    long tl = blockIdx.x;
    int q = 0;
    while (true) {
        const long next_tile = tl + gridDim.x;
        const bool has_next = next_tile < total_tiles;
        auto chunk = [&](auto firstc, int ch) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(firstc)::value;

            // ---- producer phase: slab q + 1 into the other buffers, slab q + 2 into the registers
            store_slab((q + 1) & 1);
            load_slab();
            // ---- consumer phase: slab q
            const unsigned ua = ua0 + (unsigned)(q & 1) * (W6_U_FLOATS * 4u);
            const unsigned va = va0 + (unsigned)(q & 1) * (W6_V_FLOATS * 4u), vb = vb0 + (unsigned)(q & 1) * (W6_V_FLOATS * 4u);
            f32x2 a[2], b0[2], b1[2], b2[2], b3[2];                      // fragments one position (8 MFMAs = 256 cycles) ahead
            asm volatile("ds_read_b64 %0, %1" : "=v"(a[0]) : "v"(ua));
            asm volatile("ds_read_b64 %0, %1" : "=v"(b0[0]) : "v"(va));
            asm volatile("ds_read_b64 %0, %1" : "=v"(b1[0]) : "v"(vb));
            asm volatile("ds_read_b64 %0, %1 offset:256" : "=v"(b2[0]) : "v"(va));
            asm volatile("ds_read_b64 %0, %1 offset:256" : "=v"(b3[0]) : "v"(vb));
            sfor<0, 16>([&](auto ic) __attribute__((always_inline)) {
                constexpr int xi = decltype(ic)::value;
                constexpr int cur = xi & 1, nx = (xi + 1) & 1;
                if constexpr (xi + 1 < 16) {
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[nx]) : "v"(ua), "n"((xi + 1) * 2048));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b0[nx]) : "v"(va), "n"((xi + 1) * 2048));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b1[nx]) : "v"(vb), "n"((xi + 1) * 2048));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b2[nx]) : "v"(va), "n"((xi + 1) * 2048 + 256));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b3[nx]) : "v"(vb), "n"((xi + 1) * 2048 + 256));
                    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a[cur]), "+v"(b0[cur]), "+v"(b1[cur]), "+v"(b2[cur]), "+v"(b3[cur]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur]), "+v"(b0[cur]), "+v"(b1[cur]), "+v"(b2[cur]), "+v"(b3[cur]));
                }
                if constexpr ((W64_ABL & 16) != 0) {
                    acc[xi][0][0] += a[cur][0] + b0[cur][0] + b1[cur][1] + b2[cur][0] + b3[cur][1];
                } else {
                if constexpr (FIRST) {
                    const f32x4w z = {0.f, 0.f, 0.f, 0.f};
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b0[cur][0], z, 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b1[cur][0], z, 0, 0, 0);
                    acc[xi][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b2[cur][0], z, 0, 0, 0);
                    acc[xi][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b3[cur][0], z, 0, 0, 0);
                } else {
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b0[cur][0], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b1[cur][0], acc[xi][1], 0, 0, 0);
                acc[xi][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b2[cur][0], acc[xi][2], 0, 0, 0);
                acc[xi][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][0], b3[cur][0], acc[xi][3], 0, 0, 0);
                }
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][1], b0[cur][1], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][1], b1[cur][1], acc[xi][1], 0, 0, 0);
                acc[xi][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][1], b2[cur][1], acc[xi][2], 0, 0, 0);
                acc[xi][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][1], b3[cur][1], acc[xi][3], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // LDS reads of this slab retired (lgkmcnt(0) above); the DMA has landed
                    ++q;
        };
        chunk(std::true_type{}, 0);
        for (int ch = 1; ch < nchunks; ++ch) chunk(std::false_type{}, ch);
        // ---- output transform of this tile: two halves (tile rows 0-1, 2-3 = output rows 0-3, 4-7) through the wave's 8 KiB of staging
        {
            int n, ty, tx;
            tile_coords(tl, n, ty, tx);
            float* const stage = V_lds + ((q + 1) & 1) * W6_V_FLOATS + wv * 2048;  // q = the NEXT step: V[(q - 1) & 1] was consumed by the step that has just ended (V[q & 1] already holds the next slab)
            // write side: [channel 16][row 4][32 px]; lane (l15, lk) owns channels 4 lk + r, pixels 2 l15, 2 l15 + 1
            float* const wbase = stage + (4 * lk) * 128 + 2 * l15;
            // read side: 16-byte piece (lane & 7) of row (lane >> 3) of each group of 8 (channel, row) rows
            const int rrow = lane >> 3, rpiece = lane & 7;
            const int m0w = mt * WN_MT + 16 * wv;
            const int ox = 32 * tx + 4 * rpiece;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ntl = 0; ntl < 2; ++ntl) {
                    const int nt = 2 * h + ntl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float s0[4], s1[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            s0[b] = acc[0 + b][nt][r] + acc[4 + b][nt][r] + acc[8 + b][nt][r];
                            s1[b] = acc[4 + b][nt][r] - acc[8 + b][nt][r] - acc[12 + b][nt][r];
                        }
                        f32x2 o[2];
                        o[0] = f32x2{s0[0] + s0[1] + s0[2] + bv[r], s0[1] - s0[2] - s0[3] + bv[r]};
                        o[1] = f32x2{s1[0] + s1[1] + s1[2] + bv[r], s1[1] - s1[2] - s1[3] + bv[r]};
#pragma unroll
                        for (int a2 = 0; a2 < 2; ++a2) {
                            if (g.act == FAOCTASR_ACT_RELU) o[a2] = f32x2{fmaxf(o[a2][0], 0.f), fmaxf(o[a2][1], 0.f)};
                            else if (g.act == FAOCTASR_ACT_LRELU)
                                o[a2] = f32x2{o[a2][0] > 0.f ? o[a2][0] : o[a2][0] * g.slope, o[a2][1] > 0.f ? o[a2][1] : o[a2][1] * g.slope};
                            else if (g.act == FAOCTASR_ACT_TANH) o[a2] = f32x2{tanhf(o[a2][0]), tanhf(o[a2][1])};
                            *reinterpret_cast<f32x2*>(wbase + r * 128 + (2 * ntl + a2) * 32) = o[a2];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rl = 8 * i + rrow, ch = rl >> 2, row = rl & 3;      // (channel, row) of this lane's piece
                    const f32x4w v = *reinterpret_cast<const f32x4w*>(stage + rl * 32 + 4 * rpiece);
                    const int oy = 8 * ty + 4 * h + row;
                    if ((W64_ABL & 8) == 0 && m0w + ch < g.M && oy < g.OH && ox < g.OW)
                        *reinterpret_cast<f32x4w*>(y + (((long)n * g.M + m0w + ch) * g.OH + oy) * g.OW + ox) = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            barrier();                                                   // the next step's producers write into this V buffer
        }
        if (!has_next) break;
        tl = next_tile;
    }
}

// 1 launched, 0 not taken (the 32-tile kernel or the others go on), < 0 error.  `wpack` holds igemm_wino.hip's packed image.
int wino64_launch(const WinoGeom& g, const float* x, const float* wpack, const float* bias, float* y, hipStream_t s) {
    const long tiles = (long)g.N * ((g.OW + 31) / 32) * ((g.OH + 7) / 8);
    if (tiles * g.mtiles < 256 || g.nchunks < 2 || (g.OW & 3)) return 0;  // the persistent grid wants the whole chip; 16-byte output pieces
    if (tiles * g.nchunks >= (1L << 30)) return 0;
    long nbx = 256 / g.mtiles;
    nbx = nbx < 1 ? 1 : nbx;
    nbx = nbx > tiles ? tiles : nbx;
    const size_t lds = (2 * (size_t)W6_U_FLOATS + 2 * (size_t)W6_V_FLOATS) * 4;
    auto k = igemm_wino64_kernel;
    lds_optin((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)nbx, g.mtiles), dim3(256), lds, s, x, wpack, bias, y, g);
    const int rc = check_launch("igemm_wino64");
    return rc == FAOCTASR_OK ? 1 : rc;
}

}  // namespace faoctasr
