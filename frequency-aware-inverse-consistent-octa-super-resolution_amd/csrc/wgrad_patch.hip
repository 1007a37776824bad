// LDS-patch weight gradient for gfx950 (f32 MFMA 32x32x2), the wide-map counterpart of igemm_wgrad_kernel.
//
//   dW[m][(c,t)] = sum over pixels p=(n,y,x) of  dY[n][m][y][x] * X[n][c][y*S + kh_t - pad][x*S + kw_t - pad]
//
// GEMM with the PIXELS as the reduction dimension.  A block owns 64 output channels (rows m) x CT = WN*NI*32
// flattened (c,t) columns and walks a range of pixel tiles (TH=2 rows x 32 pixels of one image).  Per tile it stages
//   * the dY tile [64][64 pixels] (A operand: A[i=m][k=pixel]) and
//   * the input patch of the channels its columns touch, [(c)][(TH-1)*S+KH][31*S+KW] (zero / reflect padding
//     resolved while staging),
// once in LDS; the B fragment of column j=(c,t) for pixel p is then patch[c][row(p)*S+kh][col(p)*S+kw], i.e. a
// per-lane constant base plus a wave-uniform pixel offset: no im2col arithmetic in the MFMA loop, and dY / X are read
// about once per column slab instead of once per 64 columns.  Fragment reads are software pipelined with hand-counted
// waits (see igemm_patch.hip).  Partial sums go to the gradient arena with fp32 atomics (split over pixel ranges).
#include <type_traits>

#include "common.h"
#include "igemm_geom.h"

#ifndef WG_TRACE
#define WG_TRACE 0        // diagnostics (tools/variants.py + tools/wgrad_trace.py): block (0,0,0) stamps s_memtime of its tile phases
#endif
#if WG_TRACE
// stamps live in a buffer of their own (never in an operand: round 1 wrote them into the head of dw), read by faoctasr_wgrad_trace_read
__device__ unsigned faoctasr_wgrad_trace_buf[128];
extern "C" int faoctasr_wgrad_trace_read(unsigned* host_out, int n) {
    if (!host_out || n < 0 || n > 128) return -1;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(faoctasr_wgrad_trace_buf), sizeof(unsigned) * (size_t)n) == hipSuccess ? 0 : -3;
}
#define WGT(k, v) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0 && tile - tile0 >= 4 && tile - tile0 < 36) faoctasr_wgrad_trace_buf[(tile - tile0 - 4) * 4 + (k)] = (v); } while (0)
#define WGNOW() ((unsigned)__builtin_amdgcn_s_memtime())
#else
#define WGT(k, v) do { } while (0)
#define WGNOW() 0u
#endif

namespace faoctasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradGeom {
    int N, C, IH, IW;        // gathered tensor X
    int M, OH, OW;           // dY
    int S, pad, KH, KW, reflect;
    long wsm, wsc;           // dW element strides for row m / gathered channel c (tap index contiguous)
    int ncols;               // C * KH * KW
    int tiles_per_img, tiles_x;
    int tiles_per_block;     // pixel tiles reduced by one block
};

__device__ __forceinline__ int reflect_idx_w(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

template <int I, int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wg_static_for<I + 1, N>(f);
    }
}

constexpr int WG_TH = 2, WG_PIX = WG_TH * 32, WG_LDY = WG_PIX + 1, WG_MT = 64;
constexpr int WG_NPV = 20;                    // patch elements staged per thread

// 256 threads = 4 waves as 2 (rows m) x 2 (columns): a wave owns 32 rows x NI*32 columns, so every SIMD of the CU hosts
// exactly one wave of each resident block (3-wave blocks left two SIMDs doubly loaded and ran ~2x slower).
template <int NI, int S>
__global__ __launch_bounds__(256, 2) void wgrad_patch_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dw, const WgradGeom g) {
    constexpr int NT = 256, CT = 2 * NI * 32, MI = 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int T = g.KH * g.KW;
    const int col0 = blockIdx.x * CT;
    const int m0 = blockIdx.y * WG_MT;
    const int c_lo = col0 / T;
    int c_hi = (col0 + CT - 1) / T;
    c_hi = c_hi < g.C ? c_hi : g.C - 1;
    const int NCH = c_hi - c_lo + 1;
    const int PH = (WG_TH - 1) * S + g.KH, PW = 31 * S + g.KW, PHW = PH * PW;
    const int npatch = NCH * PHW;
    const int ndy = WG_MT * WG_PIX;
    constexpr int NDY = (WG_MT * WG_PIX + NT - 1) / NT;               // dY elements staged per thread
    float* const D_lds = reinterpret_cast<float*>(smem);              // 2 x [64][65]
    float* const P_lds = D_lds + 2 * WG_MT * WG_LDY;                  // 2 x [NCH][PH][PW]
    const long ohw = (long)g.OH * g.OW, ihw = (long)g.IH * g.IW;

    const long ntiles = (long)g.N * g.tiles_per_img;
    const long tile0 = (long)blockIdx.z * g.tiles_per_block;
    long tile1 = tile0 + g.tiles_per_block;
    tile1 = tile1 < ntiles ? tile1 : ntiles;
    if (tile0 >= tile1) return;

    // Staging loads go through buffer descriptors (SRD in SGPRs, one 32-bit offset VGPR per load): an offset past
    // num_records returns 0, which implements image-border zero padding, the M / C tails and masked tile pixels without
    // selects.  (LDS-DMA was tried for these ragged tiles: 4-byte pieces are instruction-bound, ~1.4x slower.)
    constexpr unsigned OOB = 0x7fffffffu;
    // patch element e = tid + NT*i -> (c, py, px), decoded ONCE (float-reciprocal division is too expensive per tile):
    // packed c<<16 | py<<8 | px, or -1 past the patch
    int pdec[WG_NPV];
    {
        const float invPHW = 1.0f / (float)PHW, invPW = 1.0f / (float)PW;
#pragma unroll
        for (int i = 0; i < WG_NPV; ++i) {
            const int e = tid + NT * i;
            int d = -1;
            if (e < npatch) {
                const int c = (int)(((float)e + 0.5f) * invPHW);
                const int r = e - c * PHW;
                const int py = (int)(((float)r + 0.5f) * invPW);
                d = (c << 16) | (py << 8) | (r - py * PW);
            }
            pdec[i] = d;
        }
    }
    float dv[NDY], pv[WG_NPV];
    // Interior tiles (no border, no tail: almost all of a 256x256 map) need no per-element address arithmetic: the element's byte
    // offset relative to the tile origin is tile-invariant, so it is computed once (prel / drel0) and the tile origin moves into
    // the buffer descriptors' base addresses.  Vector instructions are not hidden behind f32 MFMAs on this chip, so
    // the ~500 address instructions per tile and thread of the general path were ~20 % of this kernel's time.
    // One tile column spanning the whole map (32-wide maps): every tile has the same left / right padding columns, so those are
    // tile-invariant too and are folded into prel as permanently out-of-range elements.
    const bool one_col = g.tiles_x == 1 && g.OW == 32 && !g.reflect;
    constexpr bool USE_PREL = S != 1;                                   // stride 1 takes the wide path below instead (registers)
    unsigned prel[USE_PREL ? WG_NPV : 1];
#pragma unroll
    for (int i = 0; i < (USE_PREL ? WG_NPV : 1); ++i) {
        const int d = pdec[i];
        const int ixs = (d & 0xff) - g.pad;                              // input column when the tile starts at x0 = 0
        const bool colok = !one_col || (unsigned)ixs < (unsigned)g.IW;
        prel[i] = (d >= 0 && colok) ? 4u * (unsigned)((d >> 16) * (int)ihw + ((d >> 8) & 0xff) * g.IW + (d & 0xff)) : OOB;
    }
    const unsigned drel0 = 4u * (unsigned)((tid >> 6) * (int)ohw + ((tid & 63) >> 5) * g.OW + (tid & 31));   // + i * 4 rows of m

    // Wide fast path (stride 1): the texture path sustains only one wave-instruction per ~50 cycles here (s_memtime trace: 2000
    // cycles to issue a tile's 36 dword loads, a third of the MFMA time), so interior tiles fetch 16-byte pieces instead -- dY
    // rows as 4 float4 per thread, the patch as aligned 4-column chunks [x0 - 4 + 4 ck, +4) of each (channel, row), at most 5 per
    // thread -- and scatter them into the same LDS layout.  9 loads per thread and tile instead of 36.
    constexpr int WCH = WG_NPV / 4;                                      // chunks per thread
    const bool wide_ok = S == 1 && (g.IW & 3) == 0 && (ihw & 3) == 0 && (g.OW & 3) == 0 && (ohw & 3) == 0 && g.pad <= 4 && PW - g.pad <= 36 &&
                         NCH * PH * 10 <= NT * WCH;
    int w_lds[WCH], w_rel[WCH], w_ck[WCH];                               // LDS float offset of the chunk's first column, global float offset, chunk index
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int item = tid + NT * i;
        const int c = item / (PH * 10), r = item - c * (PH * 10), row = r / 10, ck = r - row * 10;
        const bool use = wide_ok && item < NCH * PH * 10;
        w_ck[i] = use ? ck : -1;
        w_lds[i] = c * PHW + row * PW + 4 * ck - (4 - g.pad);
        w_rel[i] = c * (int)ihw + row * g.IW + 4 * ck;
    }
    bool loaded_wide = false;

    auto tile_coords = [&](long tile, int& n, int& y0, int& x0) {
        n = (int)(tile / g.tiles_per_img);
        const int r = (int)(tile - (long)n * g.tiles_per_img);
        const int ty = r / g.tiles_x;
        y0 = ty * WG_TH;
        x0 = (r - ty * g.tiles_x) * 32;
    };
    auto load_tile = [&](long tile) {
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
        const float* dyp = dy + ((long)n * g.M + m0) * ohw;
        const long dy_bytes = (long)(g.M - m0) * ohw * 4;
        {
            const int iy0f = y0 * S - g.pad, ix0f = x0 * S - g.pad;
            const bool xin = one_col || (ix0f >= 0 && ix0f + PW <= g.IW && x0 + 32 <= g.OW);
            const bool interior = iy0f >= 0 && iy0f + PH <= g.IH && xin && y0 + WG_TH <= g.OH && m0 + WG_MT <= g.M;
            loaded_wide = false;
            const bool xwide = g.reflect ? (ix0f >= 0 && ix0f + PW <= g.IW) : true;      // zero padding = whole chunks outside the row
            if (wide_ok && iy0f >= 0 && iy0f + PH <= g.IH && xwide && y0 + WG_TH <= g.OH && x0 + 32 <= g.OW && m0 + WG_MT <= g.M) {
                loaded_wide = true;
                const float* dbase = dyp + (long)y0 * g.OW + x0;
#pragma unroll
                for (int i = 0; i < NDY / 4; ++i) {
                    const int e4 = tid + NT * i;
                    const float4 v = *reinterpret_cast<const float4*>(dbase + (long)(e4 >> 4) * ohw + ((e4 & 15) >> 3) * g.OW + (e4 & 7) * 4);
                    dv[4 * i + 0] = v.x; dv[4 * i + 1] = v.y; dv[4 * i + 2] = v.z; dv[4 * i + 3] = v.w;
                }
                const float* xbase = x + ((long)n * g.C + c_lo) * ihw + (long)iy0f * g.IW + (x0 - 4);
#pragma unroll
                for (int i = 0; i < WCH; ++i) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (w_ck[i] >= 0 && (unsigned)(x0 - 4 + 4 * w_ck[i]) < (unsigned)g.IW) v = *reinterpret_cast<const float4*>(xbase + w_rel[i]);
                    pv[4 * i + 0] = v.x; pv[4 * i + 1] = v.y; pv[4 * i + 2] = v.z; pv[4 * i + 3] = v.w;
                }
                return;
            }
            if (USE_PREL && interior) {
                const long dshift = (long)y0 * g.OW + x0;
                const long db = dy_bytes - dshift * 4;
                const auto dsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dyp + dshift), 0, (int)(db < 0x7ffffff0L ? db : 0x7ffffff0L), 0x00020000);
#pragma unroll
                for (int i = 0; i < NDY; ++i)
                    dv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dsrd, drel0, (int)(i * 16 * ohw), 0));
                const long xshift = (long)iy0f * g.IW + ix0f;
                const long xb = (long)(g.C - c_lo) * ihw * 4 - xshift * 4;
                const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + ((long)n * g.C + c_lo) * ihw + xshift), 0,
                                                                    (int)(xb < 0x7ffffff0L ? xb : 0x7ffffff0L), 0x00020000);
#pragma unroll
                for (int i = 0; i < WG_NPV; ++i) pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrd, prel[USE_PREL ? i : 0], 0, 0));
                return;
            }
        }
        const auto dsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dyp), 0, (int)(dy_bytes < 0x7ffffff0L ? dy_bytes : 0x7ffffff0L), 0x00020000);
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int e = tid + NT * i;
            const int m = e >> 6, p = e & 63;
            const int yy = y0 + (p >> 5), xx = x0 + (p & 31);
            const bool ok = e < ndy && yy < g.OH && xx < g.OW;
            const unsigned off = ok ? 4u * (unsigned)(m * (int)ohw + yy * g.OW + xx) : OOB;
            dv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dsrd, off, 0, 0));
        }
        const float* xp = x + ((long)n * g.C + c_lo) * ihw;
        const long x_bytes = (long)(g.C - c_lo) * ihw * 4;
        const auto xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xp), 0, (int)(x_bytes < 0x7ffffff0L ? x_bytes : 0x7ffffff0L), 0x00020000);
        const int iy0 = y0 * S - g.pad, ix0 = x0 * S - g.pad;
#pragma unroll
        for (int i = 0; i < WG_NPV; ++i) {
            const int d = pdec[i];
            int iy = iy0 + ((d >> 8) & 0xff), ix = ix0 + (d & 0xff);
            // branch-free reflection (selected by the uniform flag): i -> |i|, then 2n-2-i past the end
            const int ry = iy < 0 ? -iy : iy, rx = ix < 0 ? -ix : ix;
            const int ry2 = ry >= g.IH ? 2 * g.IH - 2 - ry : ry, rx2 = rx >= g.IW ? 2 * g.IW - 2 - rx : rx;
            iy = g.reflect ? ry2 : iy;
            ix = g.reflect ? rx2 : ix;
            const bool ok = d >= 0 && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
            const unsigned off = ok ? 4u * (unsigned)((d >> 16) * (int)ihw + iy * g.IW + ix) : OOB;
            pv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrd, off, 0, 0));
        }
    };
    auto store_tile = [&](int buf) {
        float* dd = D_lds + buf * WG_MT * WG_LDY;
        if (loaded_wide) {
#pragma unroll
            for (int i = 0; i < NDY / 4; ++i) {
                const int e4 = tid + NT * i;
                // rows are 65 floats apart (conflict-free column reads), i.e. 4-byte aligned only: volatile keeps the four stores
                // from being merged into one misaligned ds_write_b128
                volatile float* row = dd + (e4 >> 4) * WG_LDY + ((e4 & 15) >> 3) * 32 + (e4 & 7) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) row[j] = dv[4 * i + j];
            }
            float* pd = P_lds + buf * npatch;
#pragma unroll
            for (int i = 0; i < WCH; ++i) {
                if (w_ck[i] < 0) continue;
                const int px0 = 4 * w_ck[i] - (4 - g.pad);               // patch column of the chunk's first float
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((unsigned)(px0 + j) < (unsigned)PW) pd[w_lds[i] + j] = pv[4 * i + j];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int e = tid + NT * i;
            if (e < ndy) dd[(e >> 6) * WG_LDY + (e & 63)] = dv[i];
        }
        float* pd = P_lds + buf * npatch;
#pragma unroll
        for (int i = 0; i < WG_NPV; ++i) {
            const int e = tid + NT * i;
            if (e < npatch) pd[e] = pv[i];
        }
    };

    // this lane's NI columns: base offset of (c,t) inside the patch
    int bbase[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        int col = col0 + (wn * NI + ni) * 32 + l31;
        col = col < g.ncols ? col : g.ncols - 1;                       // clamped columns are never written back
        const int c = col / T, t = col - c * T;
        const int kh = t / g.KW, kw = t - kh * g.KW;
        bbase[ni] = (c - c_lo) * PHW + kh * PW + kw + lh * S;          // lh selects the second pixel of the pair
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    load_tile(tile0);
    store_tile(0);
    __syncthreads();

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    for (long tile = tile0; tile < tile1; ++tile) {
        const int cur = (int)(tile - tile0) & 1;
        WGT(0, WGNOW());
        if (tile + 1 < tile1) load_tile(tile + 1);
        WGT(1, WGNOW());
        // A fragment: D[m = l31 (+32)][pixel 2s + lh]; B fragment: P[bbase + pixel offset]
        // Addresses inside the tile are immediates of the LDS reads: pixel pair s of a row sits 8 S s bytes on, the second pixel
        // row S * PW floats further -- no address arithmetic between the MFMAs
        // (vector instructions are not hidden behind f32 MFMAs on this chip: DESIGN.md 4.1a).
        const unsigned Aa = lds0 + 4u * (unsigned)(cur * WG_MT * WG_LDY + (wm * 32 + l31) * WG_LDY + lh);
        unsigned Ba[NI], Bb[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            Ba[ni] = lds0 + 4u * (unsigned)(2 * WG_MT * WG_LDY + cur * npatch + bbase[ni]);
            Bb[ni] = Ba[ni] + 4u * (unsigned)(S * PW);
        }
        float a0[MI], b0[NI], a1[MI], b1[NI];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        auto rd = [&](auto sc, float (&a)[MI], float (&b)[NI]) {
            constexpr int st = decltype(sc)::value;
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[0]) : "v"(Aa), "n"(8 * st));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const unsigned ba = st < 16 ? Ba[ni] : Bb[ni];
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b[ni]) : "v"(ba), "n"(8 * S * (st & 15)));
            }
        };
        auto wait_set = [&](float (&a)[MI], float (&b)[NI], auto morec) {
            constexpr bool more = decltype(morec)::value;                // a younger set of 1 + NI reads is in flight
            if constexpr (!more) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(b[0]));
            else if constexpr (NI == 4) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
            else if constexpr (NI == 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
            else if constexpr (NI == 2) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]));
            else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[0]), "+v"(b[0]));
        };
        rd(std::integral_constant<int, 0>{}, a0, b0);
        wg_static_for<0, WG_PIX / 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            rd(std::integral_constant<int, 2 * i + 1>{}, a1, b1);
            wait_set(a0, b0, std::true_type{});
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mi], b0[ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (2 * i + 2 < WG_PIX / 2) {
                rd(std::integral_constant<int, 2 * i + 2>{}, a0, b0);
                wait_set(a1, b1, std::true_type{});
            } else {
                wait_set(a1, b1, std::false_type{});
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mi], b1[ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WGT(2, WGNOW());
        if (tile + 1 < tile1) store_tile(cur ^ 1);
        __syncthreads();
        WGT(3, WGNOW());
    }

    // epilogue: row m from the register index, column from the lane
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = col0 + (wn * NI + ni) * 32 + l31;
        if (col >= g.ncols) continue;
        const int c = col / T, t = col - c * T;
        float* dst = dw + (long)c * g.wsc + t;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = m0 + (wm + mi) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                if (m < g.M) atomicAdd(dst + (long)m * g.wsm, acc[mi][ni][rr]);
            }
    }
}

template <int NI, int S>
static bool wg_try(const float* x, const float* dy, float* dw, WgradGeom& g, hipStream_t s) {
    constexpr int CT = 2 * NI * 32, NT = 256;
    const int T = g.KH * g.KW;
    int nch = CT / T + 2;
    nch = nch < g.C ? nch : g.C;
    const int PH = (WG_TH - 1) * g.S + g.KH, PW = 31 * g.S + g.KW;
    const long npatch = (long)nch * PH * PW;
    if (npatch > (long)NT * WG_NPV) return false;
    const size_t lds = (2 * (size_t)WG_MT * WG_LDY + 2 * (size_t)npatch) * 4 + 1024;
    if (lds > 79 * 1024) return false;                                  // two blocks per CU
    const int gx = (g.ncols + CT - 1) / CT, gy = (g.M + WG_MT - 1) / WG_MT;
    const long ntiles = (long)g.N * g.tiles_per_img;
    // one residency round: 256 CUs x 2 blocks; split the pixel tiles so that gx*gy*slices just fits
    long slices = 512 / ((long)gx * gy);
    if (slices < 1) slices = 1;
    if (slices > ntiles / 4) slices = ntiles / 4 > 0 ? ntiles / 4 : 1;
    g.tiles_per_block = (int)((ntiles + slices - 1) / slices);
    slices = (ntiles + g.tiles_per_block - 1) / g.tiles_per_block;
    auto k = wgrad_patch_kernel<NI, S>;
    lds_optin((const void*)k, lds);
    hipLaunchKernelGGL(k, dim3(gx, gy, (unsigned)slices), dim3(NT), lds, s, x, dy, dw, g);
    return true;
}

// returns 1 when launched, 0 when the shape is left to the flat kernel, <0 on error.  dw must already be zeroed / hold
// the running gradient (accumulation is by atomics).
int launch_wgrad_patch(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int OH, int OW, int KH, int KW,
                       int stride, int pad, int reflect, long wsm, long wsc, hipStream_t s) {
    if (OW < 24 || (stride != 1 && stride != 2) || KH > 15 || KW > 15 || C * KH * KW < 64) return 0;
    WgradGeom g;
    g.N = N; g.C = C; g.IH = IH; g.IW = IW; g.M = M; g.OH = OH; g.OW = OW; g.S = stride; g.pad = pad; g.KH = KH; g.KW = KW;
    g.reflect = reflect; g.wsm = wsm; g.wsc = wsc; g.ncols = C * KH * KW;
    g.tiles_x = (OW + 31) / 32;
    g.tiles_per_img = g.tiles_x * ((OH + WG_TH - 1) / WG_TH);
    g.tiles_per_block = 1;
    // column slab = 2 * NI * 32: minimise padded columns, ties to the wider slab (dY / X are re-read once per slab)
    const int n = g.ncols;
    // cost model: padded MFMA work, plus the per-tile staging of the dY tile amortised over NI column tiles
    int order[4] = {4, 3, 2, 1};
    double best_cost = -1;
    int best = 0;
    for (int i = 0; i < 4; ++i) {
        const int ct = 64 * order[i];
        const double cost = (double)((n + ct - 1) / ct) * ct * (1.0 + 1.5 / order[i]);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = order[i]; }
    }
    bool ok = false;
    for (int ni = best; ni >= 1 && !ok; --ni) {
        if (stride == 1) {
            if (ni == 4) ok = wg_try<4, 1>(x, dy, dw, g, s);
            else if (ni == 3) ok = wg_try<3, 1>(x, dy, dw, g, s);
            else if (ni == 2) ok = wg_try<2, 1>(x, dy, dw, g, s);
            else ok = wg_try<1, 1>(x, dy, dw, g, s);
        } else {
            if (ni == 4) ok = wg_try<4, 2>(x, dy, dw, g, s);
            else if (ni == 3) ok = wg_try<3, 2>(x, dy, dw, g, s);
            else if (ni == 2) ok = wg_try<2, 2>(x, dy, dw, g, s);
            else ok = wg_try<1, 2>(x, dy, dw, g, s);
        }
    }
    if (!ok) return 0;
    const int rc = check_launch("wgrad_patch");
    return rc == FAOCTASR_OK ? 1 : rc;
}

}  // namespace faoctasr
