// Fused SSIM (ssim.py:17-37) for gfx950: the 11x11 sigma-1.5 window is the outer product of the
// 1-D taps (ssim.py:11-13), so all five moment maps (mu1, mu2, E[a^2], E[b^2], E[ab]) are produced
// by one row pass + one column pass through LDS, followed by the pointwise map and a block
// reduction; zero padding of 5 as F.conv2d(padding=window_size//2).  HBM traffic is the
// algorithmic minimum (read a, b once per tile + halo; write one float per image).
// Backward recomputes the moments on a halo, forms the five partial-derivative maps, and applies
// the same separable filter to them (the window is symmetric, so the adjoint of the zero-padded
// correlation is itself).
#include <type_traits>

#include "common.h"

namespace faoctasr {

struct Taps { float g[11]; };

constexpr int R = 5;                 // window radius
constexpr float C1 = 0.01f * 0.01f;  // ssim.py:29-30
constexpr float C2 = 0.03f * 0.03f;

__device__ __forceinline__ float ssim_point(float m1, float m2, float e11, float e22, float e12) {
    const float s11 = e11 - m1 * m1, s22 = e22 - m2 * m2, s12 = e12 - m1 * m2;
    return ((2.f * m1 * m2 + C1) * (2.f * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s11 + s22 + C2));
}

// forward: tile 22 x 64 outputs, one (n,c) plane per blockIdx.z.  The kernel is VALU-bound, not HBM-bound (2 x 11 taps x 5
// moments = ~135 FMA per pixel against 8 bytes), so it is organised around the packed fp32 pipe: the two images are staged
// interleaved as (a, b) pairs and the moments travel as the pairs (mu_a, mu_b), (E a^2, E b^2) plus E ab -- three
// v_pk_fma_f32 / v_fma_f32 per tap instead of five -- and both filter passes are register-blocked along the filter direction
// (4 row-pass outputs from 14 staged pairs, 3 column-pass outputs from 13 row-filtered values per thread; 512 threads), which also cuts the
// LDS reads per pixel from ~90 to ~30.  Tap order per output is unchanged from the plain form.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int SF_TY = 22, SF_TX = 64, SF_PY = SF_TY + 2 * R, SF_PX = 74;      // 32 staged rows of 74 pairs

__global__ __launch_bounds__(512) void ssim_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ sums, int C, int H, int W, const Taps tp) {
    __shared__ __attribute__((aligned(16))) f2 ab[SF_PY * SF_PX];                  // (a, b)
    __shared__ __attribute__((aligned(16))) f2 hz01[SF_PY * SF_TX], hz23[SF_PY * SF_TX];   // (mu_a, mu_b), (E a^2, E b^2) row-filtered
    __shared__ float hz4[SF_PY * SF_TX];                                          // E ab row-filtered
    __shared__ float red[8];
    const int plane = blockIdx.z;
    const int y0 = blockIdx.y * SF_TY, x0 = blockIdx.x * SF_TX;
    const float* ap = a + (long)plane * H * W;
    const float* bp = b + (long)plane * H * W;
    {   // 4 rows per pass, 128 threads per row (74 used); 60 KiB of LDS allow two blocks per CU, hence 512 threads each
        const int c = threadIdx.x & 127, rbase = threadIdx.x >> 7;
        const int xx = x0 + c - R;
        const bool cin = c < SF_PX && (unsigned)xx < (unsigned)W;
#pragma unroll
        for (int r = rbase; r < SF_PY; r += 4) {
            const int yy = y0 + r - R;
            const bool in = cin && (unsigned)yy < (unsigned)H;
            if (c < SF_PX) ab[r * SF_PX + c] = in ? f2{ap[(long)yy * W + xx], bp[(long)yy * W + xx]} : f2{0.f, 0.f};
        }
    }
    __syncthreads();
    {   // row pass: thread = (row r, 4 output columns from c0)
        const int r = threadIdx.x >> 4, c0 = (threadIdx.x & 15) * 4;
        f2 v[14];
#pragma unroll
        for (int q = 0; q < 14; ++q) v[q] = ab[r * SF_PX + c0 + q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
            float s4 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const f2 x = v[j + k];
                const float g = tp.g[k];
                const f2 gx = g * x;
                s01 += gx;
                s23 += gx * x;
                s4 += gx[0] * x[1];
            }
            hz01[r * SF_TX + c0 + j] = s01;
            hz23[r * SF_TX + c0 + j] = s23;
            hz4[r * SF_TX + c0 + j] = s4;
        }
    }
    __syncthreads();
    float local = 0.f;
    {   // column pass: thread = (column c, 3 output rows from r0)
        const int c = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * 3;
        f2 m01[3], m23[3];
        float m4[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { m01[j] = f2{0.f, 0.f}; m23[j] = f2{0.f, 0.f}; m4[j] = 0.f; }
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            int rr = r0 + i;
            rr = rr < SF_PY ? rr : SF_PY - 1;                          // rows past the tile feed only masked outputs
            const f2 v01 = hz01[rr * SF_TX + c], v23 = hz23[rr * SF_TX + c];
            const float v4 = hz4[rr * SF_TX + c];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (i - j >= 0 && i - j < 11) {
                    const float g = tp.g[i - j];
                    m01[j] += g * v01;
                    m23[j] += g * v23;
                    m4[j] += g * v4;
                }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (r0 + j < SF_TY && y0 + r0 + j < H && x0 + c < W) local += ssim_point(m01[j][0], m01[j][1], m23[j][0], m23[j][1], m4[j]);
    }
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sums + plane / C, ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])));
}

// backward: tile 16 x 32 outputs
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ gout, int gN, float gscale, float* __restrict__ da,
                                                       float* __restrict__ db, int C, int H, int W, const Taps tp) {
    constexpr int TY = 16, TX = 32;
    constexpr int QY = TY + 2 * R, QX = TX + 2 * R;       // region where the ssim map's partials are needed
    constexpr int PY = QY + 2 * R, PX = QX + 2 * R;       // input region
    __shared__ float as[PY * PX], bs[PY * PX];
    __shared__ float hz[5][PY * QX];                      // row-filtered moments; later reused for row-filtered partials
    __shared__ float pm[5][QY * QX];                      // partial-derivative maps f_m1, f_m2, f_e11, f_e22, f_e12
    const int plane = blockIdx.z;
    const int y0 = blockIdx.y * TY, x0 = blockIdx.x * TX;
    const float* ap = a + (long)plane * H * W;
    const float* bp = b + (long)plane * H * W;
    for (int i = threadIdx.x; i < PY * PX; i += 256) {
        const int r = i / PX, c = i - r * PX;
        const int yy = y0 + r - 2 * R, xx = x0 + c - 2 * R;
        const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        as[i] = in ? ap[(long)yy * W + xx] : 0.f;
        bs[i] = in ? bp[(long)yy * W + xx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PY * QX; i += 256) {
        const int r = i / QX, c = i - r * QX;
        float sa = 0.f, sb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float av = as[r * PX + c + k], bv = bs[r * PX + c + k], g = tp.g[k];
            sa += g * av; sb += g * bv; saa += g * av * av; sbb += g * bv * bv; sab += g * av * bv;
        }
        hz[0][i] = sa; hz[1][i] = sb; hz[2][i] = saa; hz[3][i] = sbb; hz[4][i] = sab;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < QY * QX; i += 256) {
        const int r = i / QX, c = i - r * QX;
        const int yy = y0 + r - R, xx = x0 + c - R;
        float f[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float g = tp.g[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] += g * hz[q][(r + k) * QX + c];
            }
            const float m1 = m[0], m2 = m[1];
            const float s11 = m[2] - m1 * m1, s22 = m[3] - m2 * m2, s12 = m[4] - m1 * m2;
            const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * s12 + C2, B1 = m1 * m1 + m2 * m2 + C1, B2 = s11 + s22 + C2;
            const float inv = 1.f / (B1 * B2);
            const float S = A1 * A2 * inv;
            f[0] = (2.f * m2 * (A2 - A1)) * inv - S * (2.f * m1 / B1 - 2.f * m1 / B2);
            f[1] = (2.f * m1 * (A2 - A1)) * inv - S * (2.f * m2 / B1 - 2.f * m2 / B2);
            f[2] = -S / B2;
            f[3] = -S / B2;
            f[4] = 2.f * A1 * inv;
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) pm[q][i] = f[q];
    }
    __syncthreads();
    // row pass over the partial maps: rows QY, output columns TX (reuses hz storage, row stride TX)
    for (int i = threadIdx.x; i < QY * TX; i += 256) {
        const int r = i / TX, c = i - r * TX;
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = tp.g[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) s[q] += g * pm[q][r * QX + c + k];
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) hz[q][i] = s[q];
    }
    __syncthreads();
    const float gv = gout[gN > 1 ? plane / C : 0] * gscale;
    for (int i = threadIdx.x; i < TY * TX; i += 256) {
        const int r = i / TX, c = i - r * TX;
        const int yy = y0 + r, xx = x0 + c;
        if (yy < H && xx < W) {
            float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float g = tp.g[k];
#pragma unroll
                for (int q = 0; q < 5; ++q) s[q] += g * hz[q][(r + k) * TX + c];
            }
            const float av = as[(r + 2 * R) * PX + c + 2 * R], bv = bs[(r + 2 * R) * PX + c + 2 * R];
            const long off = (long)plane * H * W + (long)yy * W + xx;
            if (da) da[off] = gv * (s[0] + 2.f * av * s[2] + bv * s[4]);
            if (db) db[off] = gv * (s[1] + 2.f * bv * s[3] + av * s[4]);
        }
    }
}


// ---- sliding-window forms (W even): no block-level barrier, no tile halo re-reads beyond 10 rows per row segment --------------
// One WAVE owns a strip of 128 output columns (lane = 2 adjacent columns) and walks down a segment of rows.  Per input row:
// the lane's (a, b) pairs go to a wave-private LDS row (plus 5 halo columns on each side, loaded by lanes 0..9), every lane reads
// the 12 pairs its two columns need (6 ds_read_b128) and forms the five row-filtered moments; the column pass runs over an
// 11-row register ring (row loop unrolled 11x, so ring slots are static registers).  LDS traffic 14 B in / 96 B out per lane and
// row instead of ~135 B per pixel, one global read of each input, and the only synchronisation is the wave's own lgkmcnt.
constexpr int SL_COLS = 128, SL_PAIRS = SL_COLS + 2 * R + 2;          // pairs per LDS row (index i <-> column x0 - 5 + i)

template <int I, int N, class F>
__device__ __forceinline__ void ss_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ss_static_for<I + 1, N>(f);
    }
}

// planar scalars: v_pk_fma_f32 has the FLOP rate of two v_fma_f32 on this chip, and the packed form cost ~85 v_mov per row to
// keep its operands in aligned register pairs (ISA count of the first version)
// FOUR filtered maps, not five: SSIM needs sigma_a^2 + sigma_b^2 only as a sum, so E a^2 and E b^2 travel as one map
// E (a^2 + b^2) (20 % fewer filter FMAs in the forward and -- its two partial derivatives being equal -- in the backward)
struct Mom { float ma, mb, mss, mab; };                                  // mu_a, mu_b, E (a^2 + b^2), E ab

// row-filtered moments of the lane's two columns from the 12 staged pairs
__device__ __forceinline__ void row_moments(const f2 (&v)[12], const Taps& tp, Mom (&out)[2]) {
    float ss[12], ab[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) { ss[q] = __builtin_fmaf(v[q][1], v[q][1], v[q][0] * v[q][0]); ab[q] = v[q][0] * v[q][1]; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        Mom s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float g = tp.g[k];
            s.ma += g * v[j + k][0]; s.mb += g * v[j + k][1];
            s.mss += g * ss[j + k]; s.mab += g * ab[j + k];
        }
        out[j] = s;
    }
}

__device__ __forceinline__ Mom mom_zero() { return Mom{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void mom_axpy(Mom& acc, float g, const Mom& h) {
    acc.ma += g * h.ma; acc.mb += g * h.mb; acc.mss += g * h.mss; acc.mab += g * h.mab;
}
// the same value as ssim_point(m1, m2, e11, e22, e12) with e11 + e22 given as one number
__device__ __forceinline__ float ssim_point_sum(float m1, float m2, float ess, float e12) {
    const float s12 = e12 - m1 * m2, mm = m1 * m1 + m2 * m2;
    return ((2.f * m1 * m2 + C1) * (2.f * s12 + C2)) / ((mm + C1) * ((ess - mm) + C2));
}

// stage one input row of the strip in the wave's LDS row and fetch the lane's 12 pairs.  `cur` = this lane's two columns of
// the row as loaded from global memory ((a0,a1),(b0,b1)), `halo` = the pair of the lane's halo column (lanes 0..9).
__device__ __forceinline__ void exchange_row(f2* rowbuf, int lane, const f2& a2, const f2& b2, const f2& halo, f2 (&v)[12]) {
    rowbuf[R + 2 * lane] = f2{a2[0], b2[0]};
    rowbuf[R + 2 * lane + 1] = f2{a2[1], b2[1]};
    if (lane < 2 * R) rowbuf[lane < R ? lane : SL_COLS + lane] = halo;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // other lanes' stores before this lane's loads: the wave runs
    __builtin_amdgcn_wave_barrier();                                     // in lockstep and its LDS operations retire in order, so this
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");               // only pins the compiler's ordering (no instruction)
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v* src = reinterpret_cast<const f4v*>(rowbuf + 2 * lane);     // 16-byte aligned: pair index 2 * lane
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const f4v t = src[q];
        v[2 * q] = f2{t[0], t[1]};
        v[2 * q + 1] = f2{t[2], t[3]};
    }
}

// work item -> (plane, strip, row segment)
struct SlideItem { int plane, x0, r0, r1; bool valid; };
__device__ __forceinline__ SlideItem slide_item(long item, int planes, int nstrips, int nseg, int seg_rows, int H) {
    SlideItem it;
    const int seg = (int)(item % nseg);
    const long t = item / nseg;
    const int strip = (int)(t % nstrips);
    it.plane = (int)(t / nstrips);
    it.valid = it.plane < planes;
    it.x0 = strip * SL_COLS;
    it.r0 = seg * seg_rows;
    it.r1 = it.r0 + seg_rows < H ? it.r0 + seg_rows : H;
    return it;
}

__global__ __launch_bounds__(256) void ssim_fwd_slide_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ sums,
                                                             int planes, int C, int H, int W, int nstrips, int nseg, int seg_rows, const Taps tp) {
    __shared__ __attribute__((aligned(16))) f2 rows_lds[4][SL_PAIRS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const SlideItem it = slide_item((long)blockIdx.x * 4 + wave, planes, nstrips, nseg, seg_rows, H);
    if (!it.valid || it.r0 >= it.r1) return;                             // whole waves only: no barrier anywhere in this kernel
    f2* rowbuf = rows_lds[wave];
    const float* ap = a + (long)it.plane * H * W;
    const float* bp = b + (long)it.plane * H * W;
    const int xc = it.x0 + 2 * lane;                                     // the lane's first column
    const bool cin = xc < W;                                             // W even: both columns in or out
    const int xh = lane < R ? it.x0 - R + lane : it.x0 + SL_COLS + (lane - R);      // halo column of lanes 0..9
    const bool hin = lane < 2 * R && (unsigned)xh < (unsigned)W;
    auto load_row = [&](int y, f2& a2, f2& b2, f2& halo) {
        a2 = f2{0.f, 0.f}; b2 = f2{0.f, 0.f}; halo = f2{0.f, 0.f};
        if ((unsigned)y < (unsigned)H) {                                 // uniform; rows outside the image are zero padding
            if (cin) {
                a2 = *reinterpret_cast<const f2*>(ap + (long)y * W + xc);
                b2 = *reinterpret_cast<const f2*>(bp + (long)y * W + xc);
            }
            if (hin) halo = f2{ap[(long)y * W + xh], bp[(long)y * W + xh]};
        }
    };
    // ring of 12 slots (11 live rows + the one being written) so that the 12x unrolled row loop also keeps a static 3-deep
    // register prefetch of input rows: at ~760 cycles of arithmetic per row and two waves per SIMD, one row of lookahead left
    // the HBM latency exposed (226 us per 512 planes; 3 rows: see DESIGN.md 4.3)
    constexpr int RING = 12, PF = 3;
    Mom ring[RING][2];
#pragma unroll
    for (int s = 0; s < RING; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) ring[s][j] = mom_zero();
    float local = 0.f;
    const int y_first = it.r0 - R, y_last = it.r1 - 1 + R;              // input rows that feed the segment's outputs
    f2 na2[PF], nb2[PF], nh[PF];                                         // rows yin .. yin + PF - 1, slot = row index mod PF
#pragma unroll
    for (int p = 0; p < PF; ++p) load_row(y_first + p <= y_last ? y_first + p : -1, na2[p], nb2[p], nh[p]);
    for (int ybase = y_first; ybase <= y_last; ybase += RING) {
        ss_static_for<0, RING>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int yin = ybase + j;
            if (yin <= y_last) {                                         // uniform
                const f2 a2 = na2[j % PF], b2 = nb2[j % PF], hh = nh[j % PF];
                load_row(yin + PF <= y_last ? yin + PF : -1, na2[j % PF], nb2[j % PF], nh[j % PF]);
                f2 v[12];
                exchange_row(rowbuf, lane, a2, b2, hh, v);
                row_moments(v, tp, ring[j]);
                const int yo = yin - R;
                if (yo >= it.r0) {                                       // rows yo-5 .. yo+5 = yin-10 .. yin sit in slots j+2 .. j+12 (mod 12)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        Mom m = mom_zero();
#pragma unroll
                        for (int k = 0; k < 11; ++k) mom_axpy(m, tp.g[k], ring[(j + 2 + k) % RING][c]);
                        if (xc + c < W) local += ssim_point_sum(m.ma, m.mb, m.mss, m.mab);
                    }
                }
            }
        });
    }
    local = wave_sum(local);
    if (lane == 0) atomicAdd(sums + it.plane / C, local);
}


// Backward, sliding form.  Two separable filters in sequence per wave: (1) moments -> the partial-derivative maps
// f = d ssim / d (mu_a, mu_b, E a^2 = E b^2 [one map: the two are equal], E ab) at the rows that have left the first ring; (2) the
// same window applied to f (the window is symmetric, so the adjoint of the zero-padded correlation is itself) through a second
// exchange + ring; then da = g (F0 + 2 a F2 + b F4), db = g (F1 + 2 b F2 + a F4).  A wave computes f on its 128 lane columns and gradients on the
// inner 116 (strips overlap by 12 columns, row segments by 20 rows); one wave per 64-thread block (two 11-row rings = 220 VGPRs).
constexpr int SB_OUT = 116, SB_LEFT = 6;                                 // output columns x0+6 .. x0+121 of the 128 lane columns
struct F5 { float f0, f1, f2_, f4; };                                    // (four maps; the name is historical)

__device__ __forceinline__ F5 ssim_partials(const Mom& m) {
    const float m1 = m.ma, m2 = m.mb;
    const float mm = m1 * m1 + m2 * m2, s12 = m.mab - m1 * m2;
    const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * s12 + C2, B1 = mm + C1, B2 = (m.mss - mm) + C2;
    const float rB1 = 1.f / B1, rB2 = 1.f / B2;                          // two divisions instead of seven
    const float inv = rB1 * rB2;
    const float S = A1 * A2 * inv;
    F5 f;
    f.f0 = (2.f * m2 * (A2 - A1)) * inv - S * (2.f * m1 * rB1 - 2.f * m1 * rB2);
    f.f1 = (2.f * m1 * (A2 - A1)) * inv - S * (2.f * m2 * rB1 - 2.f * m2 * rB2);
    f.f2_ = -S * rB2;
    f.f4 = 2.f * A1 * inv;
    return f;
}
__device__ __forceinline__ F5 f5_zero() { return F5{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void f5_axpy(F5& acc, float g, const F5& h) {
    acc.f0 += g * h.f0; acc.f1 += g * h.f1; acc.f2_ += g * h.f2_; acc.f4 += g * h.f4;
}

// (two waves per SIMD: with four maps the two 11-row rings + the exchange leave the kernel 8 registers over 256; the compiler is asked
// to fit, which it does without scratch)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void ssim_bwd_slide_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gout,
                                                            int gN, float gscale, float* __restrict__ da, float* __restrict__ db, int planes, int C,
                                                            int H, int W, int nstrips, int nseg, int seg_rows, const Taps tp) {
    __shared__ __attribute__((aligned(16))) f2 rowbuf[SL_PAIRS];
    __shared__ __attribute__((aligned(16))) f2 fbuf[(SL_COLS + 12) * 2];    // per column (f0,f1),(f2,f4): 16 bytes; index i <-> column x0 - 6 + i
    const int lane = threadIdx.x;
    SlideItem it;
    {
        const long item = blockIdx.x;
        const int seg = (int)(item % nseg);
        const long t = item / nseg;
        const int strip = (int)(t % nstrips);
        it.plane = (int)(t / nstrips);
        it.valid = it.plane < planes;
        it.x0 = strip * SB_OUT - SB_LEFT;
        it.r0 = seg * seg_rows;
        it.r1 = it.r0 + seg_rows < H ? it.r0 + seg_rows : H;
    }
    if (!it.valid || it.r0 >= it.r1) return;
    const float* ap = a + (long)it.plane * H * W;
    const float* bp = b + (long)it.plane * H * W;
    const int xc = it.x0 + 2 * lane;                                     // even (x0 even): the lane's two columns are both in or both out
    const bool cin = (unsigned)xc < (unsigned)W;
    const int xh = lane < R ? it.x0 - R + lane : it.x0 + SL_COLS + (lane - R);
    const bool hin = lane < 2 * R && (unsigned)xh < (unsigned)W;
    const bool cout = cin && 2 * lane >= SB_LEFT && 2 * lane < SB_LEFT + SB_OUT;      // this lane writes gradients
    auto load_row = [&](int y, f2& a2, f2& b2, f2& halo) {
        a2 = f2{0.f, 0.f}; b2 = f2{0.f, 0.f}; halo = f2{0.f, 0.f};
        if ((unsigned)y < (unsigned)H) {
            if (cin) {
                a2 = *reinterpret_cast<const f2*>(ap + (long)y * W + xc);
                b2 = *reinterpret_cast<const f2*>(bp + (long)y * W + xc);
            }
            if (hin) halo = f2{ap[(long)y * W + xh], bp[(long)y * W + xh]};
        }
    };
    const float gv = gout[gN > 1 ? it.plane / C : 0] * gscale;
    Mom ring1[11][2];
    F5 ring2[11][2];
#pragma unroll
    for (int s = 0; s < 11; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) { ring1[s][j] = mom_zero(); ring2[s][j] = f5_zero(); }
    // input row yin enters ring 1; the map row yo1 = yin - 5 leaves it as f and enters ring 2 (row-filtered); the gradient row
    // yo2 = yin - 10 leaves ring 2.  Rows outside the image are zero in both stages.
    const int y_first = it.r0 - 2 * R, y_last = it.r1 - 1 + 2 * R;
    f2 na2, nb2, nh;
    load_row(y_first, na2, nb2, nh);
    for (int ybase = y_first; ybase <= y_last; ybase += 11) {
        ss_static_for<0, 11>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int yin = ybase + j;
            if (yin <= y_last) {
                const f2 a2 = na2, b2 = nb2, hh = nh;
                if (yin + 1 <= y_last) load_row(yin + 1, na2, nb2, nh);
                // the inputs of the gradient row this iteration will finish (L2-resident: read 10 rows ago), issued before the arithmetic
                const int yo2 = yin - 2 * R;
                f2 av = {0.f, 0.f}, bv = {0.f, 0.f};
                if (yo2 >= it.r0 && cout) {
                    av = *reinterpret_cast<const f2*>(ap + (long)yo2 * W + xc);
                    bv = *reinterpret_cast<const f2*>(bp + (long)yo2 * W + xc);
                }
                f2 v[12];
                exchange_row(rowbuf, lane, a2, b2, hh, v);
                row_moments(v, tp, ring1[j]);
                // stage 1 column pass -> partials of map row yo1 on the lane's two columns
                const int yo1 = yin - R;
                F5 f[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    Mom m = mom_zero();
#pragma unroll
                    for (int k = 0; k < 11; ++k) mom_axpy(m, tp.g[k], ring1[(j + 1 + k) % 11][c]);
                    f[c] = ssim_partials(m);
                    if (!cin || (unsigned)yo1 >= (unsigned)H) f[c] = f5_zero();
                }
                // exchange f across the lanes: column index i = 6 + 2 lane + c
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    f2* dst = fbuf + (SB_LEFT + 2 * lane + c) * 2;
                    dst[0] = f2{f[c].f0, f[c].f1}; dst[1] = f2{f[c].f2_, f[c].f4};
                }
                if (lane < 6) {                                         // columns x0-6 .. x0-1 and x0+128 .. x0+133 never hold map values
                    f2* z0 = fbuf + lane * 2;
                    f2* z1 = fbuf + (SB_LEFT + SL_COLS + lane) * 2;
                    z0[0] = z0[1] = f2{0.f, 0.f};
                    z1[0] = z1[1] = f2{0.f, 0.f};
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // the lane's output columns need f columns (2 lane + c) - 5 .. + 5 -> indices 2 lane + 1 .. 2 lane + 12: each column is one
                // 16-byte read, consumed at once by both outputs (no 12-column register array)
                {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    const f4v* src = reinterpret_cast<const f4v*>(fbuf + (2 * lane) * 2);     // 32 * lane bytes
                    F5 acc0 = f5_zero(), acc1 = f5_zero();
#pragma unroll
                    for (int q = 0; q < 12; ++q) {
                        const f4v r = src[q + 1];
                        const F5 wq = F5{r[0], r[1], r[2], r[3]};
                        if (q < 11) f5_axpy(acc0, tp.g[q], wq);
                        if (q >= 1) f5_axpy(acc1, tp.g[q - 1], wq);
                    }
                    ring2[j][0] = acc0;
                    ring2[j][1] = acc1;
                }
                // stage 2 column pass -> gradient row yo2
                if (yo2 >= it.r0 && cout) {                              // (yo2 < r1 by construction of y_last)
                    f2 oa, ob;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        F5 acc = f5_zero();
#pragma unroll
                        for (int k = 0; k < 11; ++k) f5_axpy(acc, tp.g[k], ring2[(j + 1 + k) % 11][c]);
                        oa[c] = gv * (acc.f0 + 2.f * av[c] * acc.f2_ + bv[c] * acc.f4);
                        ob[c] = gv * (acc.f1 + 2.f * bv[c] * acc.f2_ + av[c] * acc.f4);
                    }
                    const long off = (long)it.plane * H * W + (long)yo2 * W + xc;
                    if (da) *reinterpret_cast<f2*>(da + off) = oa;
                    if (db) *reinterpret_cast<f2*>(db + off) = ob;
                }
            }
        });
    }
}

static Taps make_taps() {
    // ssim.py:7-9: gauss = Tensor([exp(-(x - 5)^2 / (2 * 1.5^2))]) (fp32) / gauss.sum()
    Taps t;
    float sum = 0.f;
    for (int k = 0; k < 11; ++k) {
        t.g[k] = (float)exp(-(double)((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5));
        sum += t.g[k];
    }
    for (int k = 0; k < 11; ++k) t.g[k] /= sum;
    return t;
}

}  // namespace faoctasr

using namespace faoctasr;

extern "C" {

int faoctasr_ssim_fwd(const float* a, const float* b, float* sums, int N, int C, int H, int W, faoctasr_stream_t stream) {
    if (!a || !b || !sums) return fail(FAOCTASR_EINVAL, "ssim_fwd: null pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(FAOCTASR_EINVAL, "ssim_fwd: bad shape");
    if ((long)N * C > 65535) return fail(FAOCTASR_EUNSUPPORTED, "ssim_fwd: more than 65535 planes");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, sizeof(float) * N, st) != hipSuccess) return fail(FAOCTASR_EHIP, "ssim_fwd: memset failed");
    if ((W & 1) == 0) {
        // sliding-window form: one wave per (plane, 128-column strip, row segment); segments sized so that ~2048+ waves exist
        const int planes = N * C, nstrips = (W + SL_COLS - 1) / SL_COLS;
        int seg_rows = H;
        while ((long)planes * nstrips * ((H + seg_rows - 1) / seg_rows) < 2048 && seg_rows > 32) seg_rows = (seg_rows + 1) / 2;
        const int nseg = (H + seg_rows - 1) / seg_rows;
        const long items = (long)planes * nstrips * nseg;
        hipLaunchKernelGGL(ssim_fwd_slide_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, a, b, sums, planes, C, H, W, nstrips, nseg,
                           seg_rows, make_taps());
        return check_launch("ssim_fwd");
    }
    dim3 grid((W + SF_TX - 1) / SF_TX, (H + SF_TY - 1) / SF_TY, N * C);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(512), 0, st, a, b, sums, C, H, W, make_taps());
    return check_launch("ssim_fwd");
}

int faoctasr_ssim_bwd(const float* a, const float* b, const float* g, int gN, float gscale, float* da, float* db, int N, int C, int H,
                      int W, faoctasr_stream_t stream) {
    if (!a || !b || !g) return fail(FAOCTASR_EINVAL, "ssim_bwd: null pointer");
    if (gN != 1 && gN != N) return fail(FAOCTASR_EINVAL, "ssim_bwd: gN must be 1 or N");
    if ((long)N * C > 65535) return fail(FAOCTASR_EUNSUPPORTED, "ssim_bwd: more than 65535 planes");
    if (!da && !db) return FAOCTASR_OK;
    if ((W & 1) == 0) {
        const int planes = N * C, nstrips = (W + SB_OUT - 1) / SB_OUT;
        int seg_rows = H;
        while ((long)planes * nstrips * ((H + seg_rows - 1) / seg_rows) < 2048 && seg_rows > 64) seg_rows = (seg_rows + 1) / 2;
        const int nseg = (H + seg_rows - 1) / seg_rows;
        const long items = (long)planes * nstrips * nseg;
        hipLaunchKernelGGL(ssim_bwd_slide_kernel, dim3((unsigned)items), dim3(64), 0, (hipStream_t)stream, a, b, g, gN, gscale, da, db, planes, C, H, W,
                           nstrips, nseg, seg_rows, make_taps());
        return check_launch("ssim_bwd");
    }
    dim3 grid((W + 31) / 32, (H + 15) / 16, N * C);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, g, gN, gscale, da, db, C, H, W, make_taps());
    return check_launch("ssim_bwd");
}

}  // extern "C"
