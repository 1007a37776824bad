// Split-precision ("bf16x3") weight gradient of the stride-1 3x3 "same" convolutions on the bf16 matrix cores of gfx950 --
// the weight-gradient counterpart of igemm_bf16x3.hip, selected by TrainStep(precision="bf16x3") (BASELINE configs 3 / 5).
//
//   dW[m][c][t] = sum over pixels p=(n,y,x) of  dY[n][m][y][x] * X[n][c][y + kh_t - 1][x + kw_t - 1]
//
// Both operands are fp32 activations; each is split into hi = bf16(v), lo = bf16(v - hi) while it is staged, and a product
// is accumulated in fp32 as hi*hi + hi*lo + lo*hi (three v_mfma_f32_32x32x16_bf16, the dropped lo*lo term is ~2^-16 relative).
// The MFMA's K dimension is 16 consecutive pixels of a row.  A lane needs 8 consecutive pixels of its row (A = dY) and of
// its column (B = X shifted by the tap):
//   * dY is kept [m][pixel] (as in global memory): the A fragment is one aligned ds_read_b128;
//   * a tap shift would misalign that read for X, so X is kept TRANSPOSED, pixel-major and channel-minor ([pixel][32 channels],
//     64 bytes per pixel), and the B fragment is two ds_read_b64_tr_b16: the hardware transpose read takes a free row (= pixel)
//     address from each lane, so a tap is a constant byte offset, and delivers to each lane its channel's 4 pixels.  Four
//     consecutive pixels x 32 channels tile the 64 LDS banks exactly: the reads are conflict-free without swizzle, and every
//     fragment address is one per-lane base plus an immediate.
// Block = 256 threads, tile 64 rows (m) x 64 channels x 9 taps, pixel tile 2 rows x 32; wave (mh, cb) owns 32 rows x one
// 32-channel block x 9 taps (144 accumulator registers).  Staging: 4x4 (channel x pixel) register blocks are converted with
// v_cvt_pk_bf16_f32 across two CHANNELS at a time, so a pixel's four channels are already packed for one ds_write_b64.
// The epilogue turns the accumulators ([m][channel] per tap) into dW's [m][c][t] order through LDS so that the atomics
// into the gradient arena are contiguous runs.  LDS is single buffered (59 KiB, two blocks per CU alternate their phases).
#include <type_traits>

#include "common.h"
#include "split16.h"
#include "igemm_geom.h"

namespace faoctasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WgX3Geom {
    int N, C, H, W, M;
    long wsm, wsc;           // dW element strides of row m / channel c (9 taps contiguous)
    int tiles_x, tiles_y, tiles_per_block;
    int gx, gy, slices;      // 64-channel slabs, 64-row blocks, pixel ranges
    long dw_elems;           // extent of dW (two-pass reduction: a slice's partial dW is ws[slice * dw_elems ...])
};

constexpr int X3_DLD = 72;                         // dY row stride in LDS (bf16): 144 B rows, conflict-free ds_read_b128
constexpr int X3_PROWS = 4, X3_PCOLS = 40;         // patch: 2 + 2 rows, ten 4-pixel chunks [x0-4, x0+36)
constexpr int X3_PPIX = X3_PROWS * X3_PCOLS;       // 160 pixels per channel block
constexpr unsigned X3_CB_BYTES = X3_PPIX * 64;     // one 32-channel block of one plane
constexpr unsigned X3_XPLANE = 2 * X3_CB_BYTES;    // hi plane, then lo plane
constexpr unsigned X3_DPLANE = 64 * X3_DLD * 2;    // dY plane bytes
constexpr unsigned X3_LDS = 2 * X3_DPLANE + 2 * X3_XPLANE;      // 18432 + 40960 = 59392

template <int I, int N, class F>
__device__ __forceinline__ void x3_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        x3_static_for<I + 1, N>(f);
    }
}

template <unsigned IMM>
__device__ __forceinline__ void x3_read_tr(bf16x4& d, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM)); }
template <unsigned IMM>
__device__ __forceinline__ void x3_read_128(bf16x8& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM)); }
// every LDS read retired; naming the registers keeps the MFMAs that use them below the wait
__device__ __forceinline__ void x3_wait_set(bf16x8& ah, bf16x8& al, bf16x4 (&bh)[3][2], bf16x4 (&bl)[3][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah), "+v"(al), "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bh[1][0]), "+v"(bh[1][1]), "+v"(bh[2][0]), "+v"(bh[2][1]),
                 "+v"(bl[0][0]), "+v"(bl[0][1]), "+v"(bl[1][0]), "+v"(bl[1][1]), "+v"(bl[2][0]), "+v"(bl[2][1]));
}

// Wave-specialised (round 3): a 512-thread block, one per CU.  Waves 0-3 contract (wave (mh, cb): 32 rows x 32 channels x 9 taps, as
// before), waves 4-7 stage: the LDS image is double buffered, tile i + 1 is split and stored while tile i is contracted, and the
// loads of tile i + 2 are issued right behind that store -- a whole tile before their use.  Rounds 2's form (256 threads, two blocks
// per CU alternating) issued a tile's loads and waited for them at once: the ~2.4 us until a burst of loads from every CU has arrived
// (igemm_bf16x3.hip) were paid per tile and covered only by the other block's MFMAs.
// F16: f16x2 operands (split16.h): x and dY are staged as x * s(x_slot), dY * s(dy_slot); the scales are divided out before the atomics.
template <bool F16>
__global__ __launch_bounds__(512) void wgrad_x3_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                       const WgX3Geom g, const unsigned* __restrict__ x_slot, const unsigned* __restrict__ dy_slot,
                                                       float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int stid = tid & 255;
    const int l31 = lane & 31, lh = lane >> 5;
    const int mh = wave & 1, cb = (wave >> 1) & 1;                      // the consumer wave's 32 rows and 32-channel block

    const int per_slice = g.gx * g.gy;
    const int bid = blockIdx.x;
    const int nfull = (g.slices / 8) * 8 * per_slice;                    // ids below this: the blocks of one slice share an XCD
    int slice, inner;
    if (bid < nfull) {
        const int xcd = bid & 7, k = bid >> 3;
        inner = k % per_slice;
        slice = (k / per_slice) * 8 + xcd;
    } else {
        const int r = bid - nfull;
        inner = r % per_slice;
        slice = (g.slices / 8) * 8 + r / per_slice;
    }
    const int c0 = (inner % g.gx) * 64, m0 = (inner / g.gx) * 64;
    const long hw = (long)g.H * g.W;
    const long ntiles = (long)g.N * g.tiles_y * g.tiles_x;
    const long tile0 = (long)slice * g.tiles_per_block;
    long tile1 = tile0 + g.tiles_per_block;
    tile1 = tile1 < ntiles ? tile1 : ntiles;
    if (tile0 >= tile1) return;
    const int ntl = (int)(tile1 - tile0);

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;

    if (producer) {
        // ================================================ PRODUCER ================================================
        float sx = 1.f, sd = 1.f;
        if constexpr (F16) { sx = f16x2_scale(absmax_read(x_slot)); sd = f16x2_scale(absmax_read(dy_slot)); }
        // ---- staging maps (tile-invariant; addresses of LDS buffer 0) ----
        // dY item i (2 per thread): row m = it >> 3, chunk = it & 7 -> tile row chunk >> 2, pixels 8 (chunk & 3) .. +7
        unsigned dg_off[2], dl_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int it = stid + 256 * i, m = it >> 3, ch = it & 7;
            dg_off[i] = 4u * (unsigned)(m * (int)hw + (ch >> 2) * g.W + (ch & 3) * 8);
            dl_off[i] = lds0 + 2u * (unsigned)(m * X3_DLD + (ch >> 2) * 32 + (ch & 3) * 8);
        }
        // X item i (3 per thread, 640 used): a 4-channel x 4-pixel register block = channel quad q8 of block cbs, patch row, chunk ck.
        // Lane bits: q8 fastest, then the chunk.  A pixel is only 64 bytes, so the 16 lanes of a ds_write_b64 group cannot all hit
        // different banks: with the 8 quads on consecutive lanes a group is 8 quads x 2 chunks (256 bytes apart) = 2-way, against 8-way
        // with the chunks on consecutive lanes; a wave's global loads still cover 8 rows x 128 contiguous bytes per instruction.
        unsigned xg_off[3], xl_off[3];
        int x_row[3], x_ck[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int it = stid + 256 * i;
            const int q8 = it & 7, cklo = (it >> 3) & 7, rest = it >> 6;     // rest 0..7: (cbs, row) with chunks 0..7; rest 8..9: chunks 8, 9 of all rows
            int cbs, row, ck;
            if (rest < 8) { cbs = rest & 1; row = rest >> 1; ck = cklo; }
            else { cbs = rest & 1; row = cklo >> 1; ck = 8 + (cklo & 1); }
            const bool use = rest < 10;
            x_row[i] = use ? row : -1;
            x_ck[i] = ck;
            xg_off[i] = 4u * (unsigned)((cbs * 8 + q8) * 4 * (int)hw + row * g.W + 4 * ck);
            xl_off[i] = lds0 + 2 * X3_DPLANE + (unsigned)cbs * X3_CB_BYTES + (unsigned)(row * X3_PCOLS + 4 * ck) * 64u + (unsigned)q8 * 8u;
        }
        int tn, ty, tx;                                                  // load cursor
        {
            const long per_img = (long)g.tiles_y * g.tiles_x;
            tn = (int)(tile0 / per_img);
            const int r = (int)(tile0 - (long)tn * per_img);
            ty = r / g.tiles_x;
            tx = r - ty * g.tiles_x;
        }
        f32x4 dv[2][2], xv[3][4];
        auto load_tile = [&]() {                                         // the cursor's tile -> registers; cursor + 1
            const int y0 = ty * 2, x0 = tx * 32;
            const float* dsrc = dy + ((long)tn * g.M + m0) * hw + (long)y0 * g.W + x0;
            const float* xsrc = x + ((long)tn * g.C + c0) * hw + (long)(y0 - 1) * g.W + (x0 - 4);
            if (++tx == g.tiles_x) { tx = 0; if (++ty == g.tiles_y) { ty = 0; ++tn; } }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* p = reinterpret_cast<const char*>(dsrc) + dg_off[i];
                dv[i][0] = *reinterpret_cast<const f32x4*>(p);
                dv[i][1] = *reinterpret_cast<const f32x4*>(p + 16);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int iy = y0 - 1 + x_row[i], gx0 = x0 - 4 + 4 * x_ck[i];
                const bool ok = x_row[i] >= 0 && (unsigned)iy < (unsigned)g.H && (unsigned)gx0 < (unsigned)g.W;      // zero padding: row / chunk outside
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    xv[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok) xv[i][c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xsrc + (long)c * hw) + xg_off[i]);
                }
            }
        };
        auto store_tile = [&](int buf) {                                 // registers -> (hi, lo) bf16 -> LDS buffer buf
            const unsigned bo = (unsigned)buf * X3_LDS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                split_pair_scaled<F16>(dv[i][0][0], dv[i][0][1], sd, h0, l0);
                split_pair_scaled<F16>(dv[i][0][2], dv[i][0][3], sd, h1, l1);
                split_pair_scaled<F16>(dv[i][1][0], dv[i][1][1], sd, h2, l2);
                split_pair_scaled<F16>(dv[i][1][2], dv[i][1][3], sd, h3, l3);
                const u32x4 hi = {h0, h1, h2, h3}, lo = {l0, l1, l2, l3};
                const unsigned da = dl_off[i] + bo;
                asm volatile("ds_write_b128 %0, %1" ::"v"(da), "v"(hi) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(da), "v"(lo), "n"(X3_DPLANE) : "memory");
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (x_row[i] < 0) continue;
                const unsigned xa = xl_off[i] + bo;
#pragma unroll
                for (int px = 0; px < 4; ++px) {                         // one pixel's four channels -> 8 bytes per plane
                    unsigned h0, h1, l0, l1;
                    split_pair_scaled<F16>(xv[i][0][px], xv[i][1][px], sx, h0, l0);
                    split_pair_scaled<F16>(xv[i][2][px], xv[i][3][px], sx, h1, l1);
                    const u32x2 hi = {h0, h1}, lo = {l0, l1};
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(xa), "v"(hi), "n"(64 * px) : "memory");
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(xa), "v"(lo), "n"(64 * px + X3_XPLANE) : "memory");
                }
            }
        };
        auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        load_tile();                                                     // tile 0
        store_tile(0);
        if (ntl > 1) load_tile();                                        // tile 1
        barrier();
        for (int i = 0; i < ntl; ++i) {                                  // the consumers contract tile i out of buffer i & 1
            if (i + 1 < ntl) store_tile((i + 1) & 1);                    // loaded a whole tile ago
            if (i + 2 < ntl) load_tile();
            barrier();
        }
        return;
    }

    // ==================================================== CONSUMER ====================================================
    // ---- fragment addresses (LDS buffer 0) ----
    // A: row mh*32 + l31, pixels (row kb >> 1, 16 (kb & 1) + 8 lh + 0..7)
    const unsigned a_addr0 = lds0 + 2u * (unsigned)((mh * 32 + l31) * X3_DLD + 8 * lh);
    // B (transpose read): 16-lane group: channel half (lane >> 4) & 1, lane-in-group = 4 q + p supplies pixel row q, channels 4 p..+3
    const int gi = lane & 15, q4 = gi >> 2, p4 = gi & 3, chh = (lane >> 4) & 1;
    const unsigned b_addr0 = lds0 + 2 * X3_DPLANE + (unsigned)cb * X3_CB_BYTES + (unsigned)(8 * lh + q4) * 64u + (unsigned)(chh * 4 + p4) * 8u;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    asm volatile("s_barrier" ::: "memory");                             // tile 0 staged
    for (int it = 0; it < ntl; ++it) {
        const unsigned a_addr = a_addr0 + (unsigned)(it & 1) * X3_LDS, b_addr = b_addr0 + (unsigned)(it & 1) * X3_LDS;
        // ---- contraction: 4 k-blocks of 16 pixels x 9 taps x 3 split products ----
        // Twelve sub-steps (k-block kb, kernel row kh), each 9 MFMAs on fragment set s & 1 with the 12 transpose reads (+ the two dY
        // reads of a new k-block) of sub-step s + 1 issued two per MFMA gap into the other set (round 3; before, a sub-step's reads were
        // issued in a bunch and waited for with lgkmcnt(0) in front of its MFMAs: the LDS latency was exposed twelve times per tile).
        {
            bf16x8 a_hi[2], a_lo[2];
            bf16x4 bh[2][3][2], bl[2][3][2];
            auto rd = [&](auto sc, auto jc) {                            // read j (0 .. 13) of sub-step s
                constexpr int s = decltype(sc)::value, j = decltype(jc)::value;
                constexpr int kb = s / 3, kh = s % 3, set = s & 1;
                if constexpr (j < 12) {
                    constexpr int kw = j >> 2, which = j & 3;
                    // pixel of k index j: row (kb >> 1) + kh, column 16 (kb & 1) + j + kw + 3 (patch column 0 = image column x0 - 4, pad 1)
                    constexpr unsigned imm = (unsigned)(((kb >> 1) + kh) * X3_PCOLS + 16 * (kb & 1) + kw + 3) * 64u;
                    if constexpr (which == 0) x3_read_tr<imm>(bh[set][kw][0], b_addr);
                    else if constexpr (which == 1) x3_read_tr<imm + 4 * 64>(bh[set][kw][1], b_addr);
                    else if constexpr (which == 2) x3_read_tr<imm + X3_XPLANE>(bl[set][kw][0], b_addr);
                    else x3_read_tr<imm + 4 * 64 + X3_XPLANE>(bl[set][kw][1], b_addr);
                } else if constexpr (kh == 0) {                          // a new k-block: its dY fragments
                    constexpr unsigned a_imm = 2u * (unsigned)((kb >> 1) * 32 + 16 * (kb & 1));
                    if constexpr (j == 12) x3_read_128<a_imm>(a_hi[kb & 1], a_addr);
                    else x3_read_128<a_imm + X3_DPLANE>(a_lo[kb & 1], a_addr);
                }
            };
            x3_static_for<0, 14>([&](auto jc) { rd(std::integral_constant<int, 0>{}, jc); });
            x3_static_for<0, 12>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int kb = s / 3, kh = s % 3, set = s & 1, ab = kb & 1;
                x3_wait_set(a_hi[ab], a_lo[ab], bh[set], bl[set]);
                __builtin_amdgcn_sched_barrier(0);
                x3_static_for<0, 9>([&](auto mc) {
                    constexpr int m = decltype(mc)::value, kw = m / 3, term = m % 3;
                    const bf16x8 b_hi = __builtin_shufflevector(bh[set][kw][0], bh[set][kw][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 b_lo = __builtin_shufflevector(bl[set][kw][0], bl[set][kw][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    f32x16& d = acc[kh * 3 + kw];
                    if constexpr (term == 0) mfma16<F16>(a_lo[ab], b_hi, d);
                    else if constexpr (term == 1) mfma16<F16>(a_hi[ab], b_lo, d);
                    else mfma16<F16>(a_hi[ab], b_hi, d);
                    if constexpr (s + 1 < 12 && m < 7) {
                        rd(std::integral_constant<int, s + 1>{}, std::integral_constant<int, 2 * m>{});
                        rd(std::integral_constant<int, s + 1>{}, std::integral_constant<int, 2 * m + 1>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every read of this buffer retired; the next tile is staged
    }

    // ---- epilogue: accumulators [m][channel] per tap -> dW order [m][c][t] through LDS, contiguous atomics ----
    // per wave and pass: 8 rows x 32 channels x 9 taps = 2304 floats (9216 B); the four waves use disjoint regions
    float* stage = reinterpret_cast<float*>(smem) + wave * 2304;
    const long col_base = (long)(c0 + cb * 32) * g.wsc;
    float inv = 1.f;
    if constexpr (F16) inv = f16x2_inv_scale(absmax_read(x_slot)) * f16x2_inv_scale(absmax_read(dy_slot));
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                        // rows 8 j .. 8 j + 7 of the wave's 32 = registers 4 j .. 4 j + 3 of both lane halves
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[((r + 4 * lh) * 32 + l31) * 9 + t] = F16 ? acc[t][4 * j + r] * inv : acc[t][4 * j + r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // two-pass (ws): this slice's partial dW, plain stores (wgrad_reduce_kernel); else atomics into dW.  The choice is made once and
        // the read-backs go twelve at a time: with `if (ws)` inside the loop every element compiled to its own
        // `ds_read_b32; s_waitcnt lgkmcnt(0); store` region, 144 serial LDS round trips per block
        auto flush = [&](auto two_pass) {
            float* const out = decltype(two_pass)::value ? ws + (long)slice * g.dw_elems : dw;
#pragma unroll
            for (int i0 = 0; i0 < 36; i0 += 12) {
                float v[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = stage[lane + 64 * (i0 + i)];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const int e = lane + 64 * (i0 + i);                  // element of [8 rows][288 = 32 channels x 9 taps]
                    const int row = e / 288, col = e - row * 288;
                    const long o = (long)(m0 + mh * 32 + 8 * j + row) * g.wsm + col_base + col;
                    if constexpr (decltype(two_pass)::value) out[o] = v[i];
                    else atomicAdd(out + o, v[i]);
                }
            }
        };
        if (ws) flush(std::true_type{});
        else flush(std::false_type{});
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One kernel ROW per block: the wide filters (7x7, pad 3, reflection or zero padding: model.py:451,473).  49 taps would need
// 49 accumulator tiles per wave; a block takes the KW taps of one kernel row kh instead (KW accumulators, 112 registers at
// KW = 7), so its input patch is just the 2 rows y0 + kh - pad, y0 + 1 + kh - pad of the 2-row pixel tile -- every input row is
// still staged once per (tile, kh) pair that uses it, and dY is re-read by the KH blocks of a tile (L2).  Everything else is the
// 3x3 kernel above: dY [m][pixel] for ds_read_b128, X transposed [pixel][32 channels] for ds_read_b64_tr_b16, hi/lo planes.
// Reflection: a mirrored ROW is a redirected source row; a border chunk (image columns -4..-1 or W..W+3) is the neighbouring
// chunk with its four pixels permuted (out[q] = in[4-q] left, in[2-q] right: pad <= 3, so the one pixel this cannot supply is
// never read), exactly as wgrad_s1.hip does.
// Stride 2 (round 3: the 4x4 and 3x3 stride-2 convolutions and, with x / dy swapped, the stride-2 transposed ones): the patch row is
// 72 columns (output x at tap kw reads patch column 2 x + kw - pad + 4 <= 62 + KW + 3 - pad), the two patch rows are input rows
// 2 (y0 + r) + kh - pad, and a lane's transpose-read address advances two pixels per k index -- everything else is unchanged.
constexpr int X3R_PROWS = 2;
template <int S> struct X3R {
    static constexpr int PCOLS = S == 1 ? X3_PCOLS : 72;             // patch columns: 4-pixel chunks [S x0 - 4, ...)
    static constexpr int NCK = PCOLS / 4;
    static constexpr unsigned CB_BYTES = X3R_PROWS * PCOLS * 64;     // one 32-channel block of one plane
    static constexpr unsigned XPLANE = 2 * CB_BYTES;
    static constexpr unsigned LDS = 2 * X3_DPLANE + 2 * XPLANE;      // S = 1: 18432 + 20480 = 38912; S = 2: 18432 + 36864 = 55296
    static constexpr int NXI = (2 * 8 * X3R_PROWS * NCK + 255) / 256;   // X items per staging thread: 2 / 3
};

struct WgX3RowGeom {
    int N, C, H, W, OH, OW, M, KH, pad, reflect;     // x: C x H x W, dy: M x OH x OW
    long wsm, wsc;           // dW element strides of row m / channel c (KH*KW taps contiguous)
    int tiles_x, tiles_y, tiles_per_block;
    int gx, gy, slices;      // 64-channel slabs, 64-row blocks, pixel ranges (x KH kernel rows)
    long dw_elems;
};

template <int KW, int S, int PAD, bool F16>
__global__ __launch_bounds__(512) void wgrad_x3_row_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                           const WgX3RowGeom g, const unsigned* __restrict__ x_slot, const unsigned* __restrict__ dy_slot,
                                                           float* __restrict__ ws) {
    // wave-specialised like wgrad_x3_kernel (round 3): waves 0-3 contract, waves 4-7 stage into a double-buffered LDS image, a tile ahead
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int stid = tid & 255;
    const int l31 = lane & 31, lh = lane >> 5;
    const int mh = wave & 1, cb = (wave >> 1) & 1;

    const int per_slice = g.gx * g.gy * g.KH;
    const int bid = blockIdx.x;
    const int nfull = (g.slices / 8) * 8 * per_slice;
    int slice, inner;
    if (bid < nfull) {
        const int xcd = bid & 7, k = bid >> 3;
        inner = k % per_slice;
        slice = (k / per_slice) * 8 + xcd;
    } else {
        const int r = bid - nfull;
        inner = r % per_slice;
        slice = (g.slices / 8) * 8 + r / per_slice;
    }
    const int kh = inner % g.KH;
    const int rest = inner / g.KH;
    const int c0 = (rest % g.gx) * 64, m0 = (rest / g.gx) * 64;
    const long hw = (long)g.H * g.W, ohw = (long)g.OH * g.OW;
    typedef X3R<S> R;
    const long ntiles = (long)g.N * g.tiles_y * g.tiles_x;
    const long tile0 = (long)slice * g.tiles_per_block;
    long tile1 = tile0 + g.tiles_per_block;
    tile1 = tile1 < ntiles ? tile1 : ntiles;
    if (tile0 >= tile1) return;
    const int ntl = (int)(tile1 - tile0);

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    const unsigned d_lds0 = lds0, x_lds0 = lds0 + 2 * X3_DPLANE;

    if (producer) {
        float sx = 1.f, sd = 1.f;
        if constexpr (F16) { sx = f16x2_scale(absmax_read(x_slot)); sd = f16x2_scale(absmax_read(dy_slot)); }
        unsigned dg_off[2], dl_off[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int it = stid + 256 * i, m = it >> 3, ch = it & 7;
            dg_off[i] = 4u * (unsigned)(m * (int)ohw + (ch >> 2) * g.OW + (ch & 3) * 8);
            dl_off[i] = d_lds0 + 2u * (unsigned)(m * X3_DLD + (ch >> 2) * 32 + (ch & 3) * 8);
        }
        // X item i (R::NXI per thread): channel quad q8 of block cbs, patch row, chunk ck; lane bits: q8 fastest, then the chunk
        unsigned xc_off[R::NXI], xl_off[R::NXI];
        int x_row[R::NXI], x_ck[R::NXI];
#pragma unroll
        for (int i = 0; i < R::NXI; ++i) {
            const int it = stid + 256 * i;
            const int q8 = it & 7, rem = it >> 3;
            const int ck = rem % R::NCK, rc = rem / R::NCK;              // rc = row * 2 + cbs
            const int cbs = rc & 1, row = (rc >> 1) & 1;
            x_row[i] = rc < 4 ? row : -1;
            x_ck[i] = ck;
            xc_off[i] = 4u * (unsigned)((cbs * 8 + q8) * 4 * (int)hw);      // channel part; row / column are resolved per tile (padding)
            xl_off[i] = x_lds0 + (unsigned)cbs * R::CB_BYTES + (unsigned)(row * R::PCOLS + 4 * ck) * 64u + (unsigned)q8 * 8u;
        }
        int tn, ty, tx;                                                  // load cursor
        {
            const long per_img = (long)g.tiles_y * g.tiles_x;
            tn = (int)(tile0 / per_img);
            const int r = (int)(tile0 - (long)tn * per_img);
            ty = r / g.tiles_x;
            tx = r - ty * g.tiles_x;
        }
        f32x4 dv[2][2], xv[R::NXI][4];
        int flip[R::NXI] = {};
        auto load_tile = [&]() {
            const int y0 = ty * 2, x0 = tx * 32;
            const float* dsrc = dy + ((long)tn * g.M + m0) * ohw + (long)y0 * g.OW + x0;
            const float* ximg = x + ((long)tn * g.C + c0) * hw;
            if (++tx == g.tiles_x) { tx = 0; if (++ty == g.tiles_y) { ty = 0; ++tn; } }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* p = reinterpret_cast<const char*>(dsrc) + dg_off[i];
                dv[i][0] = *reinterpret_cast<const f32x4*>(p);
                dv[i][1] = *reinterpret_cast<const f32x4*>(p + 16);
            }
#pragma unroll
            for (int i = 0; i < R::NXI; ++i) {
                int iy = S * (y0 + x_row[i]) + kh - PAD, gx0 = S * x0 - 4 + 4 * x_ck[i];
                bool ok = x_row[i] >= 0;
                flip[i] = 0;
                if (g.reflect) {
                    iy = iy < 0 ? -iy : (iy >= g.H ? 2 * g.H - 2 - iy : iy);
                    if (gx0 < 0) { gx0 = 0; flip[i] = 1; }
                    else if (gx0 >= g.W) { gx0 = g.W - 4; flip[i] = 2; }
                } else {
                    ok = ok && (unsigned)iy < (unsigned)g.H && (unsigned)gx0 < (unsigned)g.W;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    xv[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok) xv[i][c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ximg + (long)c * hw + (long)iy * g.W + gx0) + xc_off[i]);
                }
            }
        };
        auto store_tile = [&](int buf) {
            const unsigned bo = (unsigned)buf * R::LDS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                split_pair_scaled<F16>(dv[i][0][0], dv[i][0][1], sd, h0, l0);
                split_pair_scaled<F16>(dv[i][0][2], dv[i][0][3], sd, h1, l1);
                split_pair_scaled<F16>(dv[i][1][0], dv[i][1][1], sd, h2, l2);
                split_pair_scaled<F16>(dv[i][1][2], dv[i][1][3], sd, h3, l3);
                const u32x4 hi = {h0, h1, h2, h3}, lo = {l0, l1, l2, l3};
                const unsigned da = dl_off[i] + bo;
                asm volatile("ds_write_b128 %0, %1" ::"v"(da), "v"(hi) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(da), "v"(lo), "n"(X3_DPLANE) : "memory");
            }
#pragma unroll
            for (int i = 0; i < R::NXI; ++i) {
                if (x_row[i] < 0) continue;
                const unsigned xa = xl_off[i] + bo;
#pragma unroll
                for (int c = 0; c < 4; ++c) {                            // mirrored border chunks (reflection only)
                    const f32x4 v = xv[i][c];
                    const f32x4 l = {v[0], v[3], v[2], v[1]}, r = {v[2], v[1], v[0], v[3]};
                    xv[i][c] = flip[i] == 1 ? l : (flip[i] == 2 ? r : v);
                }
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    unsigned h0, h1, l0, l1;
                    split_pair_scaled<F16>(xv[i][0][px], xv[i][1][px], sx, h0, l0);
                    split_pair_scaled<F16>(xv[i][2][px], xv[i][3][px], sx, h1, l1);
                    const u32x2 hi = {h0, h1}, lo = {l0, l1};
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(xa), "v"(hi), "n"(64 * px) : "memory");
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(xa), "v"(lo), "n"(64 * px + R::XPLANE) : "memory");
                }
            }
        };
        auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        load_tile();                                                     // tile 0
        store_tile(0);
        if (ntl > 1) load_tile();                                        // tile 1
        barrier();
        for (int i = 0; i < ntl; ++i) {                                  // the consumers contract tile i out of buffer i & 1
            if (i + 1 < ntl) store_tile((i + 1) & 1);
            if (i + 2 < ntl) load_tile();
            barrier();
        }
        return;
    }

    const unsigned a_addr0 = d_lds0 + 2u * (unsigned)((mh * 32 + l31) * X3_DLD + 8 * lh);
    const int gi = lane & 15, q4 = gi >> 2, p4 = gi & 3, chh = (lane >> 4) & 1;
    const unsigned b_addr0 = x_lds0 + (unsigned)cb * R::CB_BYTES + (unsigned)(S * (8 * lh + q4)) * 64u + (unsigned)(chh * 4 + p4) * 8u;

    f32x16 acc[KW];
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    asm volatile("s_barrier" ::: "memory");                             // tile 0 staged
    for (int it = 0; it < ntl; ++it) {
        const unsigned a_addr = a_addr0 + (unsigned)(it & 1) * R::LDS, b_addr = b_addr0 + (unsigned)(it & 1) * R::LDS;
        // ---- contraction: 4 k-blocks of 16 pixels x KW taps x 3 split products ----
        x3_static_for<0, 4>([&](auto kc) {
            constexpr int kb = decltype(kc)::value;
            constexpr unsigned a_imm = 2u * (unsigned)((kb >> 1) * 32 + 16 * (kb & 1));
            bf16x8 a_hi, a_lo;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a_hi) : "v"(a_addr), "n"(a_imm));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a_lo) : "v"(a_addr), "n"(a_imm + X3_DPLANE));
            bf16x4 bh[KW][2], bl[KW][2];
            x3_static_for<0, KW>([&](auto wc) {
                constexpr int kw = decltype(wc)::value;
                // pixel of k index j: patch row kb >> 1, column S (16 (kb & 1) + j) + kw + 4 - pad
                constexpr unsigned imm = (unsigned)((kb >> 1) * R::PCOLS + S * 16 * (kb & 1) + kw + (4 - PAD)) * 64u;
                const unsigned ba = b_addr;                              // (asm operands cannot name a captured variable / array element)
                bf16x4 h0, h1, l0, l1;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h0) : "v"(ba), "n"(imm));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h1) : "v"(ba), "n"(imm + S * 4 * 64));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(l0) : "v"(ba), "n"(imm + R::XPLANE));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(l1) : "v"(ba), "n"(imm + S * 4 * 64 + R::XPLANE));
                bh[kw][0] = h0; bh[kw][1] = h1; bl[kw][0] = l0; bl[kw][1] = l1;
            });
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_hi), "+v"(a_lo));
            x3_static_for<0, KW>([&](auto wc) {
                constexpr int kw = decltype(wc)::value;
                bf16x4 h0 = bh[kw][0], h1 = bh[kw][1], l0 = bl[kw][0], l1 = bl[kw][1];
                asm volatile("" : "+v"(h0), "+v"(h1), "+v"(l0), "+v"(l1));      // the fragments are ordered behind the wait
                bh[kw][0] = h0; bh[kw][1] = h1; bl[kw][0] = l0; bl[kw][1] = l1;
            });
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const bf16x8 b_hi = __builtin_shufflevector(bh[kw][0], bh[kw][1], 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 b_lo = __builtin_shufflevector(bl[kw][0], bl[kw][1], 0, 1, 2, 3, 4, 5, 6, 7);
                f32x16& d = acc[kw];
                mfma16<F16>(a_lo, b_hi, d);
                mfma16<F16>(a_hi, b_lo, d);
                mfma16<F16>(a_hi, b_hi, d);
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every read of this buffer retired; the next tile is staged
    }

    // ---- epilogue: [m][channel] per tap -> dW[m][c][kh][0..KW) through LDS, atomics in runs of KW floats per channel ----
    float* stage = reinterpret_cast<float*>(smem) + wave * (8 * 32 * KW);
    const long col_base = (long)(c0 + cb * 32) * g.wsc + (long)kh * KW;
    float inv = 1.f;
    if constexpr (F16) inv = f16x2_inv_scale(absmax_read(x_slot)) * f16x2_inv_scale(absmax_read(dy_slot));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < KW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[((r + 4 * lh) * 32 + l31) * KW + t] = F16 ? acc[t][4 * j + r] * inv : acc[t][4 * j + r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto flush = [&](auto two_pass) {                                // as in wgrad_x3_kernel: the (ws) choice once, read-backs in groups
            float* const out = decltype(two_pass)::value ? ws + (long)slice * g.dw_elems : dw;
            constexpr int NE = (8 * 32 * KW + 63) / 64;                  // 8 * 32 * KW is a multiple of 64 (4 KW elements per lane)
            static_assert(8 * 32 * KW % 64 == 0, "whole waves of elements");
            constexpr int NB = KW == 7 ? 7 : KW == 4 ? 8 : 6;             // read-backs in flight together
            static_assert(NE % NB == 0, "whole batches");
#pragma unroll
            for (int i0 = 0; i0 < NE; i0 += NB) {
                float v[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) v[i] = stage[lane + 64 * (i0 + i)];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int e = lane + 64 * (i0 + i);                  // element of [8 rows][32 channels][KW taps]
                    const int row = e / (32 * KW), rem = e - row * (32 * KW), ch = rem / KW, t = rem - ch * KW;
                    const long o = (long)(m0 + mh * 32 + 8 * j + row) * g.wsm + col_base + (long)ch * g.wsc + t;
                    if constexpr (decltype(two_pass)::value) out[o] = v[i];
                    else atomicAdd(out + o, v[i]);
                }
            }
        };
        if (ws) flush(std::true_type{});
        else flush(std::false_type{});
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Second pass of the two-pass weight-gradient reduction: dW[i] += sum over slices of ws[slice][i], slices in a FIXED order, one add
// into dW per element.  The slices' partials were written with plain 6 TB/s stores instead of `slices` fp32 atomics per element
// (1.3 TB/s chip-wide: 38 MB per launch on the 256 -> 256 and 64 -> 64 3x3 layers, 29 us of a 67 / 136 us kernel), and the sum no
// longer depends on the order in which blocks finish: with every weight gradient of a step on ONE stream (TrainStep's side stream)
// dW is bit-reproducible.  The final add stays an atomic because a captured step issues weight gradients of one layer from
// several streams at once (train.TrainStep.capture_side_wgrad).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long elems, int slices) {
    // block = 16 float4 elements x 16 slice groups: thread (e, grp) sums the slices grp, grp + 16, ... of its element (four loads in
    // flight), the 16 group sums meet in LDS and are added in group order.  (One thread per element walking all slices was
    // latency-bound: 23 us per launch for 38 MB, and 36 blocks on the 64 -> 64 layers.)
    __shared__ f32x4 part[16][16];
    const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long i = ((long)blockIdx.x * 16 + e) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < elems) {
        const float* p = ws + i;
        int k = grp;
        for (; k + 48 < slices; k += 64) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p + (long)k * elems), b = *reinterpret_cast<const f32x4*>(p + (long)(k + 16) * elems);
            const f32x4 c = *reinterpret_cast<const f32x4*>(p + (long)(k + 32) * elems), d = *reinterpret_cast<const f32x4*>(p + (long)(k + 48) * elems);
            s += (a + b) + (c + d);
        }
        for (; k < slices; k += 16) s += *reinterpret_cast<const f32x4*>(p + (long)k * elems);
    }
    part[grp][e] = s;
    __syncthreads();
    if (grp == 0 && i < elems) {
        f32x4 t = part[0][e];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += part[g][e];
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(dw + i + j, t[j]);
    }
}

static int launch_wgrad_reduce(const float* ws, float* dw, long elems, int slices, hipStream_t s) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((elems / 4 + 15) / 16)), dim3(256), 0, s, ws, dw, elems, slices);
    return check_launch("wgrad_reduce");
}

// workspace of the two-pass reduction, or nullptr (atomics straight into dW) when the caller's is missing / too small / unaligned
static float* x3_ws(long elems, int slices) {
    float* ws = g_wgrad_ws;
    if (!ws || (elems & 3) || (reinterpret_cast<uintptr_t>(ws) & 15) || g_wgrad_ws_floats < elems * slices) return nullptr;
    return ws;
}

template <int KW, int S, int PAD, bool F16>
static int x3_row_go2(const float* x, const float* dy, float* dw, WgX3RowGeom& g, hipStream_t s, const unsigned* x_slot, const unsigned* dy_slot) {
    const long ntiles = (long)g.N * g.tiles_y * g.tiles_x;
    long slices = 256 / ((long)g.gx * g.gy * g.KH);                      // one 512-thread block per CU
    if (slices < 1) slices = 1;
    if (slices > ntiles / 4) slices = ntiles / 4 > 0 ? ntiles / 4 : 1;
    g.tiles_per_block = (int)((ntiles + slices - 1) / slices);
    g.slices = (int)((ntiles + g.tiles_per_block - 1) / g.tiles_per_block);
    if (g_wgrad_dry_run) return 1;
    g.dw_elems = (long)g.M * g.wsm;
    float* const ws = x3_ws(g.dw_elems, g.slices);
    auto k = wgrad_x3_row_kernel<KW, S, PAD, F16>;
    const size_t lds = 2 * (size_t)X3R<S>::LDS > (size_t)4 * 8 * 32 * KW * 4 ? 2 * (size_t)X3R<S>::LDS : (size_t)4 * 8 * 32 * KW * 4;
    lds_optin((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(g.gx * g.gy * g.KH * g.slices)), dim3(512), lds, s, x, dy, dw, g, x_slot, dy_slot, ws);
    int rc = check_launch("wgrad_x3_row");
    if (rc == FAOCTASR_OK && ws) rc = launch_wgrad_reduce(ws, dw, g.dw_elems, g.slices, s);
    return rc == FAOCTASR_OK ? 1 : rc;
}
template <int KW, int S, int PAD>
static int x3_row_go(const float* x, const float* dy, float* dw, WgX3RowGeom& g, hipStream_t s, int f16, const unsigned* x_slot, const unsigned* dy_slot) {
    return f16 ? x3_row_go2<KW, S, PAD, true>(x, dy, dw, g, s, x_slot, dy_slot) : x3_row_go2<KW, S, PAD, false>(x, dy, dw, g, s, x_slot, dy_slot);
}

// one kernel row per block: 7x7 pad 3 stride 1 ("same", reflection or zero padding); 4x4 and 3x3 pad 1 stride 2 (zero padding, IH = 2 OH)
static int launch_wgrad_x3_row(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int OH, int OW, int KH, int KW,
                               int stride, int pad, int reflect, long wsm, long wsc, hipStream_t s, int f16, const unsigned* x_slot,
                               const unsigned* dy_slot) {
    if (KH != KW || (C & 63) || (M & 63) || (OW & 31) || (OH & 1) || (IW & 3) || wsc != (long)KH * KW) return 0;
    if ((long)64 * IH * IW >= (1L << 29) || (long)64 * OH * OW >= (1L << 29)) return 0;      // 32-bit byte offsets inside a 64-channel slab
    WgX3RowGeom g;
    g.N = N; g.C = C; g.H = IH; g.W = IW; g.OH = OH; g.OW = OW; g.M = M; g.KH = KH; g.pad = pad; g.reflect = reflect; g.wsm = wsm; g.wsc = wsc;
    g.tiles_x = OW / 32; g.tiles_y = OH / 2;
    g.gx = C / 64; g.gy = M / 64;
    if (stride == 1 && KH == 7 && pad == 3 && OH == IH && OW == IW) {
        if (reflect && (IH <= pad || IW < 8)) return 0;
        return x3_row_go<7, 1, 3>(x, dy, dw, g, s, f16, x_slot, dy_slot);
    }
    if (stride == 2 && pad == 1 && !reflect && IH == 2 * OH && IW == 2 * OW) {
        if (KH == 4) return x3_row_go<4, 2, 1>(x, dy, dw, g, s, f16, x_slot, dy_slot);
        if (KH == 3) return x3_row_go<3, 2, 1>(x, dy, dw, g, s, f16, x_slot, dy_slot);
    }
    return 0;
}

// returns 1 when launched, 0 when the shape is left to the fp32 kernels, <0 on error.  dw zeroed / accumulating, as elsewhere.
int launch_wgrad_x3(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int OH, int OW, int KH, int KW,
                    int stride, int pad, int reflect, long wsm, long wsc, hipStream_t s, int f16, const unsigned* x_slot, const unsigned* dy_slot) {
    if (f16 && !g_wgrad_dry_run && (!x_slot || !dy_slot)) return fail(FAOCTASR_EINVAL, "wgrad f16x2: missing absmax slots");
    if ((stride == 1 && KH == 7) || stride == 2)
        return launch_wgrad_x3_row(x, dy, dw, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, reflect, wsm, wsc, s, f16, x_slot, dy_slot);
    if (stride != 1 || KH != 3 || KW != 3 || pad != 1 || reflect || OH != IH || OW != IW) return 0;
    if ((C & 63) || (M & 63) || (OW & 31) || (OH & 1) || wsc != 9) return 0;
    if ((long)64 * IH * IW >= (1L << 29)) return 0;                      // 32-bit byte offsets inside a 64-channel slab
    WgX3Geom g;
    g.N = N; g.C = C; g.H = IH; g.W = IW; g.M = M; g.wsm = wsm; g.wsc = wsc;
    g.tiles_x = OW / 32; g.tiles_y = OH / 2;
    g.gx = C / 64; g.gy = M / 64;
    const long ntiles = (long)N * g.tiles_y * g.tiles_x;
    long slices = 256 / ((long)g.gx * g.gy);                             // one 512-thread block per CU
    if (slices < 1) slices = 1;
    if (slices > ntiles / 4) slices = ntiles / 4 > 0 ? ntiles / 4 : 1;
    g.tiles_per_block = (int)((ntiles + slices - 1) / slices);
    g.slices = (int)((ntiles + g.tiles_per_block - 1) / g.tiles_per_block);
    if (g_wgrad_dry_run) return 1;
    g.dw_elems = (long)M * wsm;
    float* const ws = x3_ws(g.dw_elems, g.slices);
    if (f16) {
        lds_optin((const void*)wgrad_x3_kernel<true>, 2 * X3_LDS);
        hipLaunchKernelGGL(wgrad_x3_kernel<true>, dim3((unsigned)(g.gx * g.gy * g.slices)), dim3(512), 2 * X3_LDS, s, x, dy, dw, g, x_slot, dy_slot, ws);
    } else {
        lds_optin((const void*)wgrad_x3_kernel<false>, 2 * X3_LDS);
        hipLaunchKernelGGL(wgrad_x3_kernel<false>, dim3((unsigned)(g.gx * g.gy * g.slices)), dim3(512), 2 * X3_LDS, s, x, dy, dw, g, x_slot, dy_slot, ws);
    }
    int rc = check_launch("wgrad_x3");
    if (rc == FAOCTASR_OK && ws) rc = launch_wgrad_reduce(ws, dw, g.dw_elems, g.slices, s);
    return rc == FAOCTASR_OK ? 1 : rc;
}

}  // namespace faoctasr
