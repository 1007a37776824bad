// Weight gradient of the stride-1 "same" convolutions (3x3 pad 1, 7x7 reflect-pad 3) on wide maps -- 78 % of the weight-gradient
// time of the train step -- for gfx950, exact fp32 on v_mfma_f32_32x32x2_f32.
//
//   dW[m][(c,t)] = sum over pixels p=(n,y,x) of  dY[n][m][y][x] * X[n][c][y + kh_t - pad][x + kw_t - pad]
//
// GEMM with the pixels as the reduction dimension, like wgrad_patch.hip, restructured after an s_memtime trace of that kernel
// (profiles/r02_wgrad_phase_trace.txt): per pixel tile a wave spent 1500 cycles issuing the next tile's loads and 500 storing
// them to LDS with no MFMA in flight from either co-resident block (16300 cycles per tile pair against 12288 of MFMA), because
// vector-memory and LDS work of one wave does not overlap another wave's f32 MFMA stream on this chip.  Here:
//   * the staging is a three-deep pipeline INSIDE the MFMA loop: while tile t is reduced from LDS buffer `cur`, the registers
//     hold tile t+1 (its LDS stores are issued one per k-step at the start of the loop) and are then refilled with tile t+2 (one
//     16-byte global load per k-step at the end of the loop): a load or a store costs its issue slot, never a burst;
//   * every tile -- image borders included -- takes the same path in 16-byte pieces: dY rows are always whole (M % 64 == 0,
//     OW % 32 == 0, OH even are required), a patch row is ten aligned 4-float chunks [x0-4+4k, +4) kept in LDS as loaded,
//     zero padding = a chunk or a row that is not loaded, reflection = a redirected row / a mirrored chunk;
//   * a wave owns all 64 rows of the block (MI = 2): the dY fragment is one ds_read_b128 per four k-steps and row block, so
//     the column fragments come two k-steps at a time (ds_read2_b32): 2 + 2 NI LDS reads per 8 NI MFMAs (0.33 per MFMA at
//     NI = 3; wgrad_patch.hip: 1.33);
//   * WK = 2: the block's four waves are 2 (pixel rows of the tile) x 2 (column halves) for filters whose column count is a
//     multiple of 192 (64x64x3x3: 576); the two pixel-row partials meet in LDS before the atomics.  WK = 1: 1 x 4 waves.
// Partial sums over pixel ranges go to the gradient arena with fp32 atomics (dw is zeroed / accumulating, as before).
#include <type_traits>

#include "common.h"

namespace faoctasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WgS1Geom {
    int N, C, H, W;          // gathered tensor X (H x W per channel)
    int OH, OW;              // dY map: H x W for stride 1 (pad = (K-1)/2), H/2 x W/2 for stride 2
    int M, KH, KW, pad, reflect;
    long wsm, wsc;           // dW element strides of row m / channel c (taps contiguous)
    int ncols;               // C * KH * KW
    int tiles_x, tiles_y;    // 32-pixel column tiles, 2-row tiles per image
    int tiles_per_block;     // pixel tiles reduced by one block
    int gx, gy, slices;      // column slabs, 64-row blocks, pixel ranges
};

constexpr int S1_LD = 68;                 // dY tile row stride in LDS (floats): 16-byte rows, conflict-free ds_read_b128 column reads
constexpr int S1_NDV = 4;                 // 16-byte dY pieces per thread and tile
// per input stride S: patch row stride in floats (ten / eighteen 4-float chunks: columns S*x0 - 4 ... ) and patch pieces per thread
template <int S> struct S1Cfg { static constexpr int RS = S == 1 ? 40 : 72, NCK = RS / 4, NPV = S == 1 ? 5 : 6; };
constexpr int S1_TRASH = 16;              // bytes at the end of the LDS image that absorb the stores of unused patch pieces

template <int I, int N, class F>
__device__ __forceinline__ void s1_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        s1_static_for<I + 1, N>(f);
    }
}

// 256 threads = 4 waves; block tile 64 rows (m) x CT = (4 / WK) * NI * 32 columns (c,t); pixel tile 2 rows x 32 of the dY map.
// S = input stride: 1 (the "same" 3x3 / 7x7 layers) or 2 (4x4 stride-2 convolutions and, with the operands swapped, the 4x4
// transposed convolution: round 1's wgrad_patch kernel keeps the shapes this one declines).
template <int WK, int NI, int S>
__global__ __launch_bounds__(256, 2) void wgrad_s1_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                          const WgS1Geom g) {
    constexpr int WN = 4 / WK, CT = WN * NI * 32, KS = 32 / WK;          // k-steps (pixel pairs) per tile and wave
    constexpr int NQ = KS / 4;                                           // quads of k-steps = dY fragment reads
    constexpr int NRD = 2 + 2 * NI;                                      // LDS read instructions per quad
    constexpr int S1_RS = S1Cfg<S>::RS, NCK = S1Cfg<S>::NCK, S1_NPV = S1Cfg<S>::NPV;
    static_assert(NRD <= 15, "counted lgkmcnt wait");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wk = WK == 2 ? wave >> 1 : 0, wn = WK == 2 ? wave & 1 : wave;

    // block id -> (slab, row block, pixel slice): the gx*gy blocks of one pixel slice share dY / X tiles, so they are given ids that
    // are equal mod 8 (same XCD under round-robin dispatch: speed only, never correctness)
    const int per_slice = g.gx * g.gy;
    int bid = blockIdx.x;
    const int nfull = (g.slices / 8) * 8 * per_slice;                    // ids below this follow the XCD-grouped order
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
    const int col0 = (inner % g.gx) * CT, m0 = (inner / g.gx) * 64;

    const int T = g.KH * g.KW;
    const int c_lo = col0 / T;
    int c_hi = (col0 + CT - 1) / T;
    c_hi = c_hi < g.C ? c_hi : g.C - 1;
    const int NCH = c_hi - c_lo + 1;
    const int PH = g.KH + S;                                             // patch rows of a 2-row tile
    const int CS = PH * S1_RS + 4;                                       // channel stride (floats): +4 turns 4-way bank conflicts into 2-way
    const int npatch = NCH * CS;
    const long hw = (long)g.H * g.W, ohw = (long)g.OH * g.OW;

    const long ntiles = (long)g.N * g.tiles_y * g.tiles_x;
    const long tile0 = (long)slice * g.tiles_per_block;
    long tile1 = tile0 + g.tiles_per_block;
    tile1 = tile1 < ntiles ? tile1 : ntiles;
    if (tile0 >= tile1) return;

    // LDS image: D[2][64][S1_LD] | P[2][npatch] | trash
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    constexpr unsigned DBUF = 64 * S1_LD * 4;                            // bytes of one dY buffer
    const unsigned pbuf = (unsigned)npatch * 4;
    const unsigned p_lds0 = lds0 + 2 * DBUF;
    const unsigned trash = p_lds0 + 2 * pbuf;

    // ---- tile-invariant per-thread staging state ------------------------------------------------------------------------
    // dY piece j: rows m = (tid >> 4) + 16 j, tile row (tid >> 3) & 1, pixels 4 (tid & 7) .. +3
    const unsigned d_goff = 4u * (unsigned)((tid >> 4) * (int)ohw + ((tid >> 3) & 1) * g.OW + (tid & 7) * 4);   // + j * 16 * ohw floats
    const unsigned d_lds = lds0 + 4u * (unsigned)((tid >> 4) * S1_LD + ((tid >> 3) & 1) * 32 + (tid & 7) * 4);   // + j * 16 * S1_LD floats
    // patch piece i: item = tid + 256 i -> (channel, row, chunk)
    unsigned p_goff[S1_NPV], p_lds[S1_NPV];
    int p_rc[S1_NPV];                                                    // row << 4 | chunk, or -1 = piece unused
    const int items = NCH * PH * NCK;
#pragma unroll
    for (int i = 0; i < S1_NPV; ++i) {
        const int item = tid + 256 * i;
        const int c = item / (PH * NCK), r = item - c * (PH * NCK), row = r / NCK, ck = r - row * NCK;
        const bool use = item < items;
        p_rc[i] = use ? (row << 5) | ck : -1;
        p_goff[i] = 4u * (unsigned)(c * (int)hw + row * g.W + 4 * ck);
        p_lds[i] = use ? p_lds0 + 4u * (unsigned)(c * CS + row * S1_RS + 4 * ck) : trash;
    }

    f32x4 dv[S1_NDV], pv[S1_NPV];
    int pflag = 0;                                                       // reflection: 2 bits per piece (1 mirror left chunk, 2 mirror right chunk)

    struct Tile {                                                        // uniform
        const float* dsrc;                                               // dY at (n, m0, y0, x0)
        const float* xsrc;                                               // X at (n, c_lo, S*y0 - pad, S*x0 - 4): may lie outside the tensor, only in-range pieces are read
        int y0, x0;
        bool interior;
    };
    int tn, ty, tx;                                                      // coordinates of the tile the NEXT load will fetch
    {
        const long per_img = (long)g.tiles_y * g.tiles_x;
        tn = (int)(tile0 / per_img);
        const int r = (int)(tile0 - (long)tn * per_img);
        ty = r / g.tiles_x;
        tx = r - ty * g.tiles_x;
    }
    auto next_tile = [&]() {
        Tile t;
        t.y0 = ty * 2; t.x0 = tx * 32;
        t.dsrc = dy + ((long)tn * g.M + m0) * ohw + (long)t.y0 * g.OW + t.x0;
        t.xsrc = x + ((long)tn * g.C + c_lo) * hw + (long)(S * t.y0 - g.pad) * g.W + (S * t.x0 - 4);
        t.interior = S * t.y0 - g.pad >= 0 && S * t.y0 - g.pad + PH <= g.H && S * t.x0 >= 4 && S * t.x0 - 4 + S1_RS <= g.W;
        if (++tx == g.tiles_x) { tx = 0; if (++ty == g.tiles_y) { ty = 0; ++tn; } }
        return t;
    };

    auto load_piece = [&](const Tile& t, auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < S1_NDV) {
            dv[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(t.dsrc + (long)j * 16 * ohw) + d_goff);
        } else {
            constexpr int i = j - S1_NDV;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int rc = p_rc[i];
            if (t.interior) {
                if (rc >= 0) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(t.xsrc) + p_goff[i]);
                if (g.reflect) pflag &= ~(3 << (2 * i));
            } else if (rc >= 0) {
                const int row = rc >> 5, ck = rc & 31;
                const int iy = S * t.y0 - g.pad + row, gx = S * t.x0 - 4 + 4 * ck;
                if (!g.reflect) {                                        // zero padding: rows / chunks outside the image stay zero
                    if ((unsigned)iy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W)
                        v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(t.xsrc) + p_goff[i]);
                } else {                                                 // ReflectionPad2d: mirrored row; border chunk = mirrored neighbour chunk
                    const int iy2 = iy < 0 ? -iy : (iy >= g.H ? 2 * g.H - 2 - iy : iy);
                    const int gx2 = gx < 0 ? 0 : (gx >= g.W ? g.W - 4 : gx);
                    const int fl = gx < 0 ? 1 : (gx >= g.W ? 2 : 0);
                    pflag = (pflag & ~(3 << (2 * i))) | (fl << (2 * i));
                    v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(t.xsrc) + p_goff[i] +
                                                        4 * ((iy2 - iy) * g.W + (gx2 - gx)));
                }
            }
            pv[i] = v;
        }
    };
    auto store_piece = [&](int buf, auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < S1_NDV) {
            const unsigned da = d_lds + buf * DBUF;
            const f32x4 v = dv[j];
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(da), "v"(v), "n"(j * 16 * S1_LD * 4) : "memory");
        } else {
            constexpr int i = j - S1_NDV;
            f32x4 v = pv[i];
            if (g.reflect) {
                // left border chunk (image columns -4..-1 -> 4,3,2,1) from the chunk [0,4): out[q] = in[4-q], q = 1..3 (q = 0 is
                // never read: pad <= 3); right border chunk (W..W+3 -> W-2,W-3,W-4,..) from [W-4,W): out[q] = in[2-q], q = 0..2
                const int fl = (pflag >> (2 * i)) & 3;
                const f32x4 l = {v[0], v[3], v[2], v[1]}, r = {v[2], v[1], v[0], v[3]};
                v = fl == 1 ? l : (fl == 2 ? r : v);
            }
            const unsigned pl = p_lds[i];
            const unsigned pa = pl + (pl == trash ? 0u : buf * pbuf);
            asm volatile("ds_write_b128 %0, %1" ::"v"(pa), "v"(v) : "memory");
        }
    };
    constexpr int NPIECE = S1_NDV + S1_NPV;

    // ---- fragment addressing -------------------------------------------------------------------------------------------------
    // k-step s of a wave covers the pixel pair (lane half lh): WK = 1: (tile row lh, x = s); WK = 2: (tile row wk, x = 16 lh + s)
    const int rowsel = WK == 2 ? wk : lh, xoff = WK == 2 ? 16 * lh : 0;
    const unsigned a_addr = lds0 + 4u * (unsigned)(l31 * S1_LD + rowsel * 32 + xoff);                  // + 32 rows for mi = 1
    unsigned b_addr[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        int col = col0 + (wn * NI + ni) * 32 + l31;
        col = col < g.ncols ? col : g.ncols - 1;                         // clamped columns are never written back
        const int c = col / T, t = col - c * T;
        const int kh = t / g.KW, kw = t - kh * g.KW;
        b_addr[ni] = p_lds0 + 4u * (unsigned)((c - c_lo) * CS + (rowsel * S + kh) * S1_RS + kw + (4 - g.pad) + S * xoff);
    }

    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // ---- prologue: tile0 into buffer 0, tile0+1 into the registers -------------------------------------------------------------
    {
        const Tile t = next_tile();
        s1_static_for<0, NPIECE>([&](auto jc) { load_piece(t, jc); });
        s1_static_for<0, NPIECE>([&](auto jc) { store_piece(0, jc); });
    }
    if (tile0 + 1 < tile1) {
        const Tile t = next_tile();
        s1_static_for<0, NPIECE>([&](auto jc) { load_piece(t, jc); });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    for (long tile = tile0; tile < tile1; ++tile) {
        const int cur = (int)(tile - tile0) & 1;
        const bool have_next = tile + 1 < tile1;                         // the registers hold tile + 1
        const bool have_nn = tile + 2 < tile1;
        Tile nn;
        nn.dsrc = dy; nn.xsrc = x; nn.y0 = 0; nn.x0 = 0; nn.interior = true;
        if (have_nn) nn = next_tile();
        const unsigned aa = a_addr + cur * DBUF;
        unsigned ba[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) ba[ni] = b_addr[ni] + cur * pbuf;

        f32x4 A0[2], A1[2];                                              // two named fragment sets (static indexing)
        f32x2 B0[2][NI], B1[2][NI];                                     // [k-step pair][column tile]
        auto rd = [&](auto qc, f32x4 (&A)[2], f32x2 (&B)[2][NI]) {
            constexpr int q = decltype(qc)::value;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[0]) : "v"(aa), "n"(16 * q));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[1]) : "v"(aa), "n"(16 * q + 32 * S1_LD * 4));
            // two k-steps of a column per instruction (ds_read2_b32: same LDS cycles as two reads, one issue slot)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const unsigned b_ = ba[ni];                          // (asm operands cannot name a captured array element)
                    f32x2 v;
                    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(b_), "n"(S * (4 * q + 2 * kp)), "n"(S * (4 * q + 2 * kp + 1)));
                    B[kp][ni] = v;
                }
        };
        // wait until at most `younger` LDS operations are outstanding; names the set so that its MFMAs stay behind the wait
        auto wait_set = [&](f32x4 (&A)[2], f32x2 (&B)[2][NI], auto yc) {
            constexpr int younger = decltype(yc)::value;
            if constexpr (NI == 3)
                asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(A[0]), "+v"(A[1]), "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[0][2]), "+v"(B[1][0]), "+v"(B[1][1]),
                             "+v"(B[1][2]) : "n"(younger));
            else if constexpr (NI == 2)
                asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(A[0]), "+v"(A[1]), "+v"(B[0][0]), "+v"(B[0][1]), "+v"(B[1][0]), "+v"(B[1][1]) : "n"(younger));
            else
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(A[0]), "+v"(A[1]), "+v"(B[0][0]), "+v"(B[1][0]) : "n"(younger));
            __builtin_amdgcn_sched_barrier(0);
        };
        // staging slot of k-step s: stores of tile+1 in steps 0..NPIECE-1, loads of tile+2 in the last NPIECE steps
        // (STEADY: both hold -- no branches between the MFMAs of all but a block's last two tiles)
        auto slot = [&](auto sc, auto steadyc) {
            constexpr int s = decltype(sc)::value;
            constexpr bool STEADY = decltype(steadyc)::value;
            if constexpr (s < NPIECE) {
                if (STEADY || have_next) store_piece(cur ^ 1, sc);
            }
            if constexpr (s >= KS - NPIECE) {
                if (STEADY || have_nn) load_piece(nn, std::integral_constant<int, s - (KS - NPIECE)>{});
            }
        };
        auto quad = [&](auto qc, auto steadyc, f32x4 (&A)[2], f32x2 (&B)[2][NI], f32x4 (&An)[2], f32x2 (&Bn)[2][NI]) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q + 1 < NQ) {
                rd(std::integral_constant<int, q + 1>{}, An, Bn);
                wait_set(A, B, std::integral_constant<int, NRD>{});
            } else {
                wait_set(A, B, std::integral_constant<int, 0>{});
            }
            s1_static_for<0, 4>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                slot(std::integral_constant<int, 4 * q + ks>{}, steadyc);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[mi][ks], B[ks >> 1][ni][ks & 1], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        rd(std::integral_constant<int, 0>{}, A0, B0);
        auto body = [&](auto steadyc) {
            s1_static_for<0, NQ / 2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                quad(std::integral_constant<int, 2 * h>{}, steadyc, A0, B0, A1, B1);
                quad(std::integral_constant<int, 2 * h + 1>{}, steadyc, A1, B1, A0, B0);
            });
        };
        body(std::false_type{});      // (a second, branch-free copy of the loop for the steady state was tried: the two copies spill 440 B/lane)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------------------
    if constexpr (WK == 2) {
        // the two pixel-row partials of a column half meet in LDS (the tile buffers are free after the last barrier)
        float* red = reinterpret_cast<float*>(smem) + wn * (2 * NI * 16 * 64);
        if (wk == 1) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((mi * NI + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
        }
        __syncthreads();
        if (wk == 1) return;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] += red[((mi * NI + ni) * 16 + r) * 64 + lane];
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = col0 + (wn * NI + ni) * 32 + l31;
        if (col >= g.ncols) continue;
        const int c = col / T, t = col - c * T;
        float* dst = dw + (long)c * g.wsc + t;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = m0 + mi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                atomicAdd(dst + (long)m * g.wsm, acc[mi][ni][rr]);
            }
    }
}

template <int WK, int NI, int S>
static bool s1_fits(const WgS1Geom& g, size_t* lds_out) {
    constexpr int CT = (4 / WK) * NI * 32;
    const int T = g.KH * g.KW;
    int nch = (CT + T - 2) / T + 1;                                      // channels a slab of CT columns can touch
    nch = nch < g.C ? nch : g.C;
    const int PH = g.KH + S, CS = PH * S1Cfg<S>::RS + 4;
    if ((long)nch * PH * S1Cfg<S>::NCK > 256 * S1Cfg<S>::NPV) return false;       // patch pieces per thread
    size_t lds = 2 * (size_t)64 * S1_LD * 4 + 2 * (size_t)nch * CS * 4 + S1_TRASH;
    const size_t red = WK == 2 ? (size_t)2 * 2 * NI * 16 * 64 * 4 : 0;   // epilogue exchange reuses the tile buffers
    lds = lds > red ? lds : red;
    if (lds > 80 * 1024) return false;                                   // two blocks per CU
    *lds_out = lds;
    return true;
}

#ifndef S1_SLOTS
#define S1_SLOTS 512          // tools/variants.py experiments: pixel slices per output tile = S1_SLOTS / (output tiles)
#endif

template <int WK, int NI, int S>
static int s1_launch(const float* x, const float* dy, float* dw, WgS1Geom g, size_t lds, hipStream_t s) {
    constexpr int CT = (4 / WK) * NI * 32;
    g.gx = (g.ncols + CT - 1) / CT;
    g.gy = g.M / 64;
    const long ntiles = (long)g.N * g.tiles_y * g.tiles_x;
    // one residency round: 256 CUs x 2 blocks; split the pixel tiles so that gx*gy*slices just fits
    long slices = S1_SLOTS / ((long)g.gx * g.gy);
    if (slices < 1) slices = 1;
    if (slices > ntiles / 4) slices = ntiles / 4 > 0 ? ntiles / 4 : 1;
    g.tiles_per_block = (int)((ntiles + slices - 1) / slices);
    g.slices = (int)((ntiles + g.tiles_per_block - 1) / g.tiles_per_block);
    auto k = wgrad_s1_kernel<WK, NI, S>;
    lds_optin((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(g.gx * g.gy * g.slices)), dim3(256), lds, s, x, dy, dw, g);
    return check_launch("wgrad_s1");
}

// returns 1 when launched, 0 when the shape is left to the other weight-gradient kernels, <0 on error.  dw must already be
// zeroed / hold the running gradient (accumulation is by atomics).
int launch_wgrad_s1(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int OH, int OW, int KH, int KW,
                    int stride, int pad, int reflect, long wsm, long wsc, hipStream_t s) {
    if (KH != KW || (M & 63) || (OW & 31) || (OH & 1) || C * KH * KW < 128) return 0;
    if ((long)C * IH * IW >= (1L << 29)) return 0;                       // 32-bit byte offsets inside one image
    WgS1Geom g;
    g.N = N; g.C = C; g.H = IH; g.W = IW; g.OH = OH; g.OW = OW; g.M = M; g.KH = KH; g.KW = KW; g.pad = pad; g.reflect = reflect;
    g.wsm = wsm; g.wsc = wsc; g.ncols = C * KH * KW;
    g.tiles_x = OW / 32; g.tiles_y = OH / 2;
    g.tiles_per_block = 1; g.gx = g.gy = g.slices = 1;
    if (stride == 2) {
        // 4x4 stride-2 pad-1 layers (discriminator convolutions, and the generator's 4x4 transposed convolution with x / dy swapped):
        // patch column of output x at tap kw = 2 x + kw - pad + 4 <= 62 + KW + 3 - pad < 72; zero padding only
        if (reflect || pad != 1 || KH != 4 || IH != 2 * OH || IW != 2 * OW || (IW & 3)) return 0;
        size_t l = 0;
        if (!s1_fits<2, 2, 2>(g, &l)) return 0;
        const int rc = s1_launch<2, 2, 2>(x, dy, dw, g, l, s);
        return rc == FAOCTASR_OK ? 1 : rc;
    }
    if (stride != 1 || OH != IH || OW != IW || 2 * pad != KH - 1) return 0;                            // "same" convolutions only
    if (pad < 1 || pad > 3 || KW + 31 + (4 - pad) > S1Cfg<1>::RS || C * KH * KW < 192) return 0;
    // slab width: least padded MFMA work; ties go to the wider register tile per wave (fewer LDS reads per MFMA)
    size_t l23 = 0, l12 = 0, l13 = 0;
    const bool f23 = s1_fits<2, 3, 1>(g, &l23), f12 = s1_fits<1, 2, 1>(g, &l12), f13 = s1_fits<1, 3, 1>(g, &l13);
    auto padded = [&](int ct) { return (long)((g.ncols + ct - 1) / ct) * ct; };
    long best = -1;
    int pick = 0;
#ifdef S1_FORCE
    (void)best; (void)padded;
    pick = S1_FORCE;                                                     // experiments (tools/variants.py): 23, 12 or 13
    if ((pick == 23 && !f23) || (pick == 12 && !f12) || (pick == 13 && !f13)) return 0;
#else
    if (f13) { best = padded(384); pick = 13; }
    if (f23 && (best < 0 || padded(192) <= best)) { best = padded(192); pick = 23; }
    if (f12 && (best < 0 || padded(256) < best)) { best = padded(256); pick = 12; }
#endif
    int rc;
    if (pick == 23) rc = s1_launch<2, 3, 1>(x, dy, dw, g, l23, s);
    else if (pick == 12) rc = s1_launch<1, 2, 1>(x, dy, dw, g, l12, s);
    else if (pick == 13) rc = s1_launch<1, 3, 1>(x, dy, dw, g, l13, s);
    else return 0;
    return rc == FAOCTASR_OK ? 1 : rc;
}

}  // namespace faoctasr
