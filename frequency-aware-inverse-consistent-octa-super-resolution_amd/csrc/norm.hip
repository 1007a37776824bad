// BatchNorm2d (training statistics) / InstanceNorm2d with fused activation + residual, forward and backward, for gfx950.
// HBM-bound: the job is bytes in flight and launches, not arithmetic.
//
// Rows r in [0,R): BatchNorm R = C (each row gathers NI images x HW); InstanceNorm R = N*C with NI = 1.  Element e of row r
// lives at ((e / HW) * R + r) * HW + e % HW.  Statistics are accumulated around a per-row shift (the row's first element) to
// avoid the E[x^2]-E[x]^2 cancellation; partials are combined in a fixed order (deterministic).
//
// Three forms, chosen per call:
//   * streaming (HW % 4 == 0): two kernels per direction (statistics / reduce, then apply / dx).  A block walks a contiguous
//     range of one row image by image -- the only division is a scalar one per block -- with four independent 16-byte loads
//     per operand in flight per thread (round 1 issued one load per thread and iteration behind a 64-bit division: 3 TB/s);
//   * small rows (NI*HW <= 16384 forward, <= 8192 backward): ONE kernel per direction, one block per row holding the whole row
//     in registers -- x (and dy, y) are read once and a launch disappears (150 of the 243 BatchNorm calls of a train step);
//   * generic scalar kernels for odd sizes (HW % 4 != 0).
// Backward of "BN + ReLU/LeakyReLU" without a residual takes the activation mask from x (y = act(x*g + b) is recomputed with
// the forward's own expression), so the saved output y is not read (y == NULL): 4 of the 16 bytes per element of that call.
#include "common.h"
#include "split16.h"

namespace faoctasr {

constexpr int BN_MAX_SPLIT = 64;

struct RowStats {
    float mean, invstd, var;
};

// ---- helpers ---------------------------------------------------------------------------------------------------------------
// the contiguous pieces [lo, hi) of image n that the element range [e0, e1) of row r covers; f(base, lo, hi) with base the
// tensor offset of (n, r, 0)
template <class F>
__device__ __forceinline__ void for_row_pieces(int R, int HW, int r, long e0, long e1, F&& f) {
    const int n0 = (int)(e0 / HW), n1 = (int)((e1 - 1) / HW);        // uniform: scalar division, once per block
    for (int n = n0; n <= n1; ++n) {
        const long pb = (long)n * HW;
        const long lo = (e0 > pb ? e0 : pb) - pb;
        const long hi = (e1 < pb + HW ? e1 : pb + HW) - pb;
        f(((long)n * R + r) * HW, lo, hi);
    }
}

// f16x2 (split16.h): the largest |output| of a kernel, folded into the caller's absmax slot (faoctasr_out_absmax) -- one atomicMax
// per block on the fp32 bit pattern; fmaxf drops NaNs, so a NaN never becomes the scale
__device__ __forceinline__ float amax4(float m, const float4& v) { return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ void amax_publish(float m, unsigned* __restrict__ slot) {
    __shared__ unsigned amax_red[4];
    absmax_publish_block(__builtin_bit_cast(unsigned, m), slot, amax_red);      // m >= 0: its bits order like the value
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float4 act_mask4(const float4& g, const float4& yv, int act, float slope) {
    return make_float4(g.x * act_grad_from_out(yv.x, act, slope), g.y * act_grad_from_out(yv.y, act, slope),
                       g.z * act_grad_from_out(yv.z, act, slope), g.w * act_grad_from_out(yv.w, act, slope));
}
// the forward's pre-activation, with the forward's own expression (x * gsc + bsh): only its sign is used
__device__ __forceinline__ float4 pre4(const float4& xv, float gsc, float bsh) {
    return make_float4(xv.x * gsc + bsh, xv.y * gsc + bsh, xv.z * gsc + bsh, xv.w * gsc + bsh);
}

__device__ __forceinline__ RowStats finish_stats(float a, float q, float shift, long L, float eps) {
    const float invL = 1.0f / (float)L;
    const float dm = a * invL;
    RowStats s;
    s.mean = shift + dm;
    float var = q * invL - dm * dm;
    s.var = var > 0.f ? var : 0.f;
    s.invstd = 1.0f / sqrtf(s.var + eps);
    return s;
}

__device__ __forceinline__ void publish_stats(const RowStats& s, int r, long L, float* save_mean, float* save_invstd, float* rmean, float* rvar,
                                              float momentum) {
    save_mean[r] = s.mean;
    save_invstd[r] = s.invstd;
    if (rmean) {
        const float unb = L > 1 ? s.var * ((float)L / (float)(L - 1)) : s.var;
        rmean[r] = (1.f - momentum) * rmean[r] + momentum * s.mean;
        rvar[r] = (1.f - momentum) * rvar[r] + momentum * unb;
    }
}

// ---- streaming kernels (HW % 4 == 0) ------------------------------------------------------------------------------------------
// Independent 16-byte loads per operand and thread.  2, not 4: alone the kernels run as fast either way (more waves fit), but at 2 every
// streaming kernel stays within 48 VGPRs -- what is left on a SIMD beside two 232-VGPR Winograd waves -- so that BatchNorm passes of one
// generator chain co-reside with the other chain's convolutions (step 109.4 -> 108.8 ms)
#ifndef BN_SMALL_FWD
#define BN_SMALL_FWD 16384      // rows up to this length run as ONE kernel per direction with the row in registers (66 / 122 / 234 VGPRs)
#define BN_SMALL_BWD 8192
#endif
#ifndef BN_NU
#define BN_NU 2
#endif
constexpr int NU = BN_NU;

__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ x, float* __restrict__ ws, int R, int HW, long L, int S,
                                                         long per) {
    __shared__ float red[4];
    const int r = blockIdx.y, s = blockIdx.x;
    long e0 = (long)s * per, e1 = e0 + per;
    e1 = e1 < L ? e1 : L;
    const float shift = x[(long)r * HW];
    float a = 0.f, q = 0.f;
    for_row_pieces(R, HW, r, e0, e1, [&](long base, long lo, long hi) {
        for (long i = lo + 4L * threadIdx.x; i < hi; i += 1024L * NU) {
            float4 v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u)
                v[u] = i + 1024L * u < hi ? ld4(x + base + i + 1024L * u) : make_float4(shift, shift, shift, shift);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                float d;
                d = v[u].x - shift; a += d; q += d * d;
                d = v[u].y - shift; a += d; q += d * d;
                d = v[u].z - shift; a += d; q += d * d;
                d = v[u].w - shift; a += d; q += d * d;
            }
        }
    });
    a = block_sum_256(a, red);
    q = block_sum_256(q, red);
    if (threadIdx.x == 0) {
        ws[((long)r * S + s) * 2 + 0] = a;
        ws[((long)r * S + s) * 2 + 1] = q;
    }
}

__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ res,
                                                         float* __restrict__ y, float* __restrict__ save_mean,
                                                         float* __restrict__ save_invstd, float* __restrict__ rmean,
                                                         float* __restrict__ rvar, const float* __restrict__ ws, int R, int Cg, int HW, long L,
                                                         int S, float eps, float momentum, int act, float slope, long per,
                                                         unsigned* __restrict__ amax) {
    const int r = blockIdx.y;
    float a = 0.f, q = 0.f, mx = 0.f;
    for (int s = 0; s < S; ++s) {   // fixed order: deterministic
        a += ws[((long)r * S + s) * 2 + 0];
        q += ws[((long)r * S + s) * 2 + 1];
    }
    const RowStats st = finish_stats(a, q, x[(long)r * HW], L, eps);
    if (blockIdx.x == 0 && threadIdx.x == 0) publish_stats(st, r, L, save_mean, save_invstd, rmean, rvar, momentum);
    const int cg = r % Cg;
    const float gsc = (gamma ? gamma[cg] : 1.f) * st.invstd;
    const float bsh = (beta ? beta[cg] : 0.f) - st.mean * gsc;
    long e0 = (long)blockIdx.x * per, e1 = e0 + per;
    e1 = e1 < L ? e1 : L;
    for_row_pieces(R, HW, r, e0, e1, [&](long base, long lo, long hi) {
        for (long i = lo + 4L * threadIdx.x; i < hi; i += 1024L * NU) {
            float4 v[NU], rr[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const long o = base + i + 1024L * u;
                const bool ok = i + 1024L * u < hi;
                v[u] = ok ? ld4(x + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                rr[u] = (ok && res) ? ld4(res + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (i + 1024L * u >= hi) continue;
                float4 o = pre4(v[u], gsc, bsh);
                o.x += rr[u].x; o.y += rr[u].y; o.z += rr[u].z; o.w += rr[u].w;
                o.x = act_apply(o.x, act, slope); o.y = act_apply(o.y, act, slope);
                o.z = act_apply(o.z, act, slope); o.w = act_apply(o.w, act, slope);
                st4(y + base + i + 1024L * u, o);
                mx = amax4(mx, o);
            }
        }
    });
    if (amax) amax_publish(mx, amax);
}

// backward reduce: s1 = sum dy', s2 = sum dy' * xhat, dy' = dy * act'(y);  y == NULL: mask from the recomputed pre-activation
// HAS_Y: the activation derivative is taken from the saved output (tanh, or an activation after a residual add); otherwise from x.
// Both backward kernels are held to 64 VGPRs (8 waves / SIMD): next to two 224-VGPR weight-gradient waves per SIMD -- the side
// stream's wgrad_s1 blocks -- one such wave still fits, so the streaming passes run under the matrix work instead of after it.
template <bool HAS_Y, int NU>
__global__ __launch_bounds__(256, 8) void norm_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ y, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                              const float* __restrict__ save_invstd, float* __restrict__ ws, int R, int Cg,
                                                              int HW, long L, int S, int act, float slope, long per) {
    __shared__ float red[4];
    const int r = blockIdx.y, s = blockIdx.x;
    long e0 = (long)s * per, e1 = e0 + per;
    e1 = e1 < L ? e1 : L;
    const float mean = save_mean[r], invstd = save_invstd[r];
    const int cg = r % Cg;
    const float gsc = (gamma ? gamma[cg] : 1.f) * invstd;
    const float bsh = (beta ? beta[cg] : 0.f) - mean * gsc;
    const bool masked = act != FAOCTASR_ACT_NONE;
    float s1 = 0.f, s2 = 0.f;
    for_row_pieces(R, HW, r, e0, e1, [&](long base, long lo, long hi) {
        for (long i = lo + 4L * threadIdx.x; i < hi; i += 1024L * NU) {
            float4 xv[NU], g[NU], yv[HAS_Y ? NU : 1];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const long o = base + i + 1024L * u;
                const bool ok = i + 1024L * u < hi;
                xv[u] = ok ? ld4(x + o) : make_float4(mean, mean, mean, mean);
                g[u] = ok ? ld4(dy + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (HAS_Y) yv[u] = ok ? ld4(y + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (masked) g[u] = act_mask4(g[u], HAS_Y ? yv[HAS_Y ? u : 0] : pre4(xv[u], gsc, bsh), act, slope);
                s1 += g[u].x + g[u].y + g[u].z + g[u].w;
                s2 += g[u].x * ((xv[u].x - mean) * invstd) + g[u].y * ((xv[u].y - mean) * invstd) + g[u].z * ((xv[u].z - mean) * invstd) +
                      g[u].w * ((xv[u].w - mean) * invstd);
            }
        }
    });
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        ws[((long)r * S + s) * 2 + 0] = s1;
        ws[((long)r * S + s) * 2 + 1] = s2;
    }
}

template <bool HAS_Y, int NU>
__global__ __launch_bounds__(256, 8) void norm_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ y, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, float* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dres,
                                                          const float* __restrict__ ws, int R, int Cg, int HW, long L, int S, int act,
                                                          float slope, long per, int accumulate_affine, unsigned* __restrict__ amax) {
    const int r = blockIdx.y;
    float s1 = 0.f, s2 = 0.f, mx = 0.f;
    for (int s = 0; s < S; ++s) {
        s1 += ws[((long)r * S + s) * 2 + 0];
        s2 += ws[((long)r * S + s) * 2 + 1];
    }
    const int cg = r % Cg;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (accumulate_affine) {
            if (dgamma) atomicAdd(dgamma + cg, s2);
            if (dbeta) atomicAdd(dbeta + cg, s1);
        } else {
            if (dgamma) dgamma[cg] = s2;
            if (dbeta) dbeta[cg] = s1;
        }
    }
    const float mean = save_mean[r], invstd = save_invstd[r];
    const float gm = gamma ? gamma[cg] : 1.f;
    const float gi = gm * invstd;
    const float gsc = gm * invstd, bsh = (beta ? beta[cg] : 0.f) - mean * gsc;
    const float m1 = s1 / (float)L, m2 = s2 / (float)L;
    const bool masked = act != FAOCTASR_ACT_NONE;
    long e0 = (long)blockIdx.x * per, e1 = e0 + per;
    e1 = e1 < L ? e1 : L;
    for_row_pieces(R, HW, r, e0, e1, [&](long base, long lo, long hi) {
        for (long i = lo + 4L * threadIdx.x; i < hi; i += 1024L * NU) {
            float4 xv[NU], g[NU], yv[HAS_Y ? NU : 1];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const long o = base + i + 1024L * u;
                const bool ok = i + 1024L * u < hi;
                xv[u] = ok ? ld4(x + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                g[u] = ok ? ld4(dy + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (HAS_Y) yv[u] = ok ? ld4(y + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (i + 1024L * u >= hi) continue;
                const long o = base + i + 1024L * u;
                if (masked) g[u] = act_mask4(g[u], HAS_Y ? yv[HAS_Y ? u : 0] : pre4(xv[u], gsc, bsh), act, slope);
                if (dres) st4(dres + o, g[u]);
                float4 d;
                d.x = gi * (g[u].x - m1 - (xv[u].x - mean) * invstd * m2);
                d.y = gi * (g[u].y - m1 - (xv[u].y - mean) * invstd * m2);
                d.z = gi * (g[u].z - m1 - (xv[u].z - mean) * invstd * m2);
                d.w = gi * (g[u].w - m1 - (xv[u].w - mean) * invstd * m2);
                st4(dx + o, d);
                mx = amax4(mx, d);
            }
        }
    });
    if (amax) amax_publish(mx, amax);
}

// ---- small rows: the whole row of a block in registers, one kernel per direction -----------------------------------------------
// K = 16-byte chunks per thread; chunk k of a thread = chunk (tid + 256 k) of the row
template <int K>
__device__ __forceinline__ void row_chunk_offsets(int R, int HW, int r, long L, long (&off)[K], bool (&ok)[K]) {
    const float invHW = 1.0f / (float)HW;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = 4 * ((int)threadIdx.x + 256 * k);
        ok[k] = e < L;
        int n = (int)(((float)e + 0.5f) * invHW);                    // exact for e < 2^22 and HW % 4 == 0 (e + 0.5 is never near a multiple of HW)
        n = ok[k] ? n : 0;
        off[k] = ((long)n * R + r) * HW + (ok[k] ? e - n * HW : 0);
    }
}

template <int K>
__global__ __launch_bounds__(256) void norm_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ res, float* __restrict__ y,
                                                             float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                             float* __restrict__ rmean, float* __restrict__ rvar, int R, int Cg, int HW, long L,
                                                             float eps, float momentum, int act, float slope, unsigned* __restrict__ amax) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    long off[K];
    bool ok[K];
    row_chunk_offsets<K>(R, HW, r, L, off, ok);
    const float shift = x[(long)r * HW];
    // Every chunk is LOADED (a chunk beyond the row reads the row's first 16 bytes: row_chunk_offsets) and the out-of-row ones are
    // replaced afterwards: with `ok[k] ? ld4(..) : ..` each load compiled to its own exec-masked region ending in `s_waitcnt vmcnt(0)`
    // -- K serial memory round trips (round 4's ISA; 256 x 32^2 rows, batch 8: 7.5-7.7 -> 6.2-7.3 us per launch).  The streaming kernels' second chunk has the same
    // predicated form; loading it unconditionally from a clamped address was measured too and is not kept: they already move 5-6 TB/s and
    // the 64 x 256^2 backward pass got slower (128 -> 136 us)
    float4 v[K], rr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = ld4(x + off[k]);
    if (res) {
#pragma unroll
        for (int k = 0; k < K; ++k) rr[k] = ld4(res + off[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (!ok[k]) v[k] = make_float4(shift, shift, shift, shift);
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float d;
        d = v[k].x - shift; a += d; q += d * d;
        d = v[k].y - shift; a += d; q += d * d;
        d = v[k].z - shift; a += d; q += d * d;
        d = v[k].w - shift; a += d; q += d * d;
    }
    a = block_sum_256(a, red);
    q = block_sum_256(q, red);
    const RowStats st = finish_stats(a, q, shift, L, eps);
    if (threadIdx.x == 0) publish_stats(st, r, L, save_mean, save_invstd, rmean, rvar, momentum);
    const int cg = r % Cg;
    const float gsc = (gamma ? gamma[cg] : 1.f) * st.invstd;
    const float bsh = (beta ? beta[cg] : 0.f) - st.mean * gsc;
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!ok[k]) continue;
        float4 o = pre4(v[k], gsc, bsh);
        if (res) { o.x += rr[k].x; o.y += rr[k].y; o.z += rr[k].z; o.w += rr[k].w; }
        o.x = act_apply(o.x, act, slope); o.y = act_apply(o.y, act, slope);
        o.z = act_apply(o.z, act, slope); o.w = act_apply(o.w, act, slope);
        st4(y + off[k], o);
        mx = amax4(mx, o);
    }
    if (amax) amax_publish(mx, amax);
}

template <int K>
__global__ __launch_bounds__(256) void norm_bwd_small_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                             float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ dres, int R, int Cg, int HW, long L, int act, float slope,
                                                             int accumulate_affine, unsigned* __restrict__ amax) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    long off[K];
    bool ok[K];
    row_chunk_offsets<K>(R, HW, r, L, off, ok);
    const float mean = save_mean[r], invstd = save_invstd[r];
    const int cg = r % Cg;
    const float gm = gamma ? gamma[cg] : 1.f;
    const float gsc = gm * invstd, bsh = (beta ? beta[cg] : 0.f) - mean * gsc;
    const bool masked = act != FAOCTASR_ACT_NONE;
    float4 xv[K], g[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        xv[k] = ok[k] ? ld4(x + off[k]) : make_float4(mean, mean, mean, mean);
        g[k] = ok[k] ? ld4(dy + off[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (masked) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float4 yv = y ? (ok[k] ? ld4(y + off[k]) : make_float4(0.f, 0.f, 0.f, 0.f)) : pre4(xv[k], gsc, bsh);
            g[k] = act_mask4(g[k], yv, act, slope);
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        s1 += g[k].x + g[k].y + g[k].z + g[k].w;
        s2 += g[k].x * ((xv[k].x - mean) * invstd) + g[k].y * ((xv[k].y - mean) * invstd) + g[k].z * ((xv[k].z - mean) * invstd) +
              g[k].w * ((xv[k].w - mean) * invstd);
    }
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        if (accumulate_affine) {
            if (dgamma) atomicAdd(dgamma + cg, s2);
            if (dbeta) atomicAdd(dbeta + cg, s1);
        } else {
            if (dgamma) dgamma[cg] = s2;
            if (dbeta) dbeta[cg] = s1;
        }
    }
    const float gi = gm * invstd;
    const float m1 = s1 / (float)L, m2 = s2 / (float)L;
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!ok[k]) continue;
        if (dres) st4(dres + off[k], g[k]);
        float4 d;
        d.x = gi * (g[k].x - m1 - (xv[k].x - mean) * invstd * m2);
        d.y = gi * (g[k].y - m1 - (xv[k].y - mean) * invstd * m2);
        d.z = gi * (g[k].z - m1 - (xv[k].z - mean) * invstd * m2);
        d.w = gi * (g[k].w - m1 - (xv[k].w - mean) * invstd * m2);
        st4(dx + off[k], d);
        mx = amax4(mx, d);
    }
    if (amax) amax_publish(mx, amax);
}

// ---- generic scalar kernels (HW % 4 != 0) ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void norm_stats_generic_kernel(const float* __restrict__ x, float* __restrict__ ws, int R, int HW, long L, int S,
                                                                 long per) {
    __shared__ float red[4];
    const int r = blockIdx.y, s = blockIdx.x;
    long e0 = (long)s * per, e1 = e0 + per;
    e1 = e1 < L ? e1 : L;
    const float shift = x[(long)r * HW];
    float a = 0.f, q = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const long n = e / HW;
        const float d = x[((long)n * R + r) * HW + (e - n * HW)] - shift;
        a += d; q += d * d;
    }
    a = block_sum_256(a, red);
    q = block_sum_256(q, red);
    if (threadIdx.x == 0) {
        ws[((long)r * S + s) * 2 + 0] = a;
        ws[((long)r * S + s) * 2 + 1] = q;
    }
}

__global__ __launch_bounds__(256) void norm_apply_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ res,
                                                                 float* __restrict__ y, float* __restrict__ save_mean,
                                                                 float* __restrict__ save_invstd, float* __restrict__ rmean,
                                                                 float* __restrict__ rvar, const float* __restrict__ ws, int R, int Cg, int HW,
                                                                 long L, int S, float eps, float momentum, int act, float slope, long per) {
    const int r = blockIdx.y;
    float a = 0.f, q = 0.f;
    for (int s = 0; s < S; ++s) {
        a += ws[((long)r * S + s) * 2 + 0];
        q += ws[((long)r * S + s) * 2 + 1];
    }
    const RowStats st = finish_stats(a, q, x[(long)r * HW], L, eps);
    if (blockIdx.x == 0 && threadIdx.x == 0) publish_stats(st, r, L, save_mean, save_invstd, rmean, rvar, momentum);
    const int cg = r % Cg;
    const float gsc = (gamma ? gamma[cg] : 1.f) * st.invstd;
    const float bsh = (beta ? beta[cg] : 0.f) - st.mean * gsc;
    long e0 = (long)blockIdx.x * per, e1 = e0 + per;
    e1 = e1 < L ? e1 : L;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const long n = e / HW;
        const long off = ((long)n * R + r) * HW + (e - n * HW);
        float v = x[off] * gsc + bsh;
        if (res) v += res[off];
        y[off] = act_apply(v, act, slope);
    }
}

__global__ __launch_bounds__(256) void norm_bwd_reduce_generic_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                      const float* __restrict__ y, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                                      const float* __restrict__ save_invstd, float* __restrict__ ws, int R,
                                                                      int Cg, int HW, long L, int S, int act, float slope, long per) {
    __shared__ float red[4];
    const int r = blockIdx.y, s = blockIdx.x;
    long e0 = (long)s * per, e1 = e0 + per;
    e1 = e1 < L ? e1 : L;
    const float mean = save_mean[r], invstd = save_invstd[r];
    const int cg = r % Cg;
    const float gsc = (gamma ? gamma[cg] : 1.f) * invstd;
    const float bsh = (beta ? beta[cg] : 0.f) - mean * gsc;
    float s1 = 0.f, s2 = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const long n = e / HW;
        const long off = ((long)n * R + r) * HW + (e - n * HW);
        float g = dy[off];
        const float xv = x[off];
        if (act != FAOCTASR_ACT_NONE) g *= act_grad_from_out(y ? y[off] : xv * gsc + bsh, act, slope);
        s1 += g;
        s2 += g * ((xv - mean) * invstd);
    }
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        ws[((long)r * S + s) * 2 + 0] = s1;
        ws[((long)r * S + s) * 2 + 1] = s2;
    }
}

__global__ __launch_bounds__(256) void norm_bwd_dx_generic_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  const float* __restrict__ y, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                                  const float* __restrict__ save_invstd, float* __restrict__ dx,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dres,
                                                                  const float* __restrict__ ws, int R, int Cg, int HW, long L, int S, int act,
                                                                  float slope, long per, int accumulate_affine) {
    const int r = blockIdx.y;
    float s1 = 0.f, s2 = 0.f;
    for (int s = 0; s < S; ++s) {
        s1 += ws[((long)r * S + s) * 2 + 0];
        s2 += ws[((long)r * S + s) * 2 + 1];
    }
    const int cg = r % Cg;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (accumulate_affine) {
            if (dgamma) atomicAdd(dgamma + cg, s2);
            if (dbeta) atomicAdd(dbeta + cg, s1);
        } else {
            if (dgamma) dgamma[cg] = s2;
            if (dbeta) dbeta[cg] = s1;
        }
    }
    const float mean = save_mean[r], invstd = save_invstd[r];
    const float gm = gamma ? gamma[cg] : 1.f;
    const float gi = gm * invstd;
    const float gsc = gm * invstd, bsh = (beta ? beta[cg] : 0.f) - mean * gsc;
    const float m1 = s1 / (float)L, m2 = s2 / (float)L;
    long e0 = (long)blockIdx.x * per, e1 = e0 + per;
    e1 = e1 < L ? e1 : L;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const long n = e / HW;
        const long off = ((long)n * R + r) * HW + (e - n * HW);
        float g = dy[off];
        const float xv = x[off];
        if (act != FAOCTASR_ACT_NONE) g *= act_grad_from_out(y ? y[off] : xv * gsc + bsh, act, slope);
        if (dres) dres[off] = g;
        dx[off] = gi * (g - m1 - (xv - mean) * invstd * m2);
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
static void norm_split(long L, int R, int& S, long& per) {
    // ~2048 blocks (8 per CU) so that every CU keeps >= 32 KiB of loads in flight; each block >= 4096 elements
    long want = (2048 + R - 1) / R;
    long maxs = (L + 4095) / 4096;
    long s = want < maxs ? want : maxs;
    if (s < 1) s = 1;
    if (s > BN_MAX_SPLIT) s = BN_MAX_SPLIT;
    per = (L + s - 1) / s;
    per = (per + 3) & ~3L;
    S = (int)((L + per - 1) / per);
}

static int norm_fwd(const float* x, const float* gamma, const float* beta, const float* res, float* y, float* save_mean,
                    float* save_invstd, float* rmean, float* rvar, int NI, int R, int Cg, int HW, float eps, float momentum,
                    int act, float slope, float* ws, hipStream_t st, unsigned* amax) {
    if (!x || !y || !save_mean || !save_invstd || !ws) return fail(FAOCTASR_EINVAL, "norm_fwd: null pointer");
    if (NI <= 0 || R <= 0 || HW <= 0) return fail(FAOCTASR_EINVAL, "norm_fwd: bad shape");
    const long L = (long)NI * HW;
    if ((HW & 3) == 0 && L <= BN_SMALL_FWD) {
        auto go = [&](auto k) {
            hipLaunchKernelGGL(k, dim3(R), dim3(256), 0, st, x, gamma, beta, res, y, save_mean, save_invstd, rmean, rvar, R, Cg, HW, L, eps,
                               momentum, act, slope, amax);
        };
        if (L <= 4096) go(norm_fwd_small_kernel<4>);
        else if (L <= 8192) go(norm_fwd_small_kernel<8>);
        else go(norm_fwd_small_kernel<16>);
        return check_launch("norm_fwd_small");
    }
    int S; long per;
    norm_split(L, R, S, per);
    if ((HW & 3) == 0) {
        hipLaunchKernelGGL(norm_stats_kernel, dim3(S, R), dim3(256), 0, st, x, ws, R, HW, L, S, per);
        hipLaunchKernelGGL(norm_apply_kernel, dim3(S, R), dim3(256), 0, st, x, gamma, beta, res, y, save_mean, save_invstd, rmean, rvar, ws, R,
                           Cg, HW, L, S, eps, momentum, act, slope, per, amax);
    } else {
        if (amax) return fail(FAOCTASR_EUNSUPPORTED, "norm_fwd: faoctasr_out_absmax needs a map size that is a multiple of 4");
        hipLaunchKernelGGL(norm_stats_generic_kernel, dim3(S, R), dim3(256), 0, st, x, ws, R, HW, L, S, per);
        hipLaunchKernelGGL(norm_apply_generic_kernel, dim3(S, R), dim3(256), 0, st, x, gamma, beta, res, y, save_mean, save_invstd, rmean, rvar,
                           ws, R, Cg, HW, L, S, eps, momentum, act, slope, per);
    }
    return check_launch("norm_fwd");
}

static int norm_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* beta, const float* save_mean,
                    const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* dres, int NI, int R, int Cg, int HW,
                    int act, float slope, int accumulate_affine, float* ws, hipStream_t st, unsigned* amax) {
    if (!x || !dy || !dx || !save_mean || !save_invstd || !ws) return fail(FAOCTASR_EINVAL, "norm_bwd: null pointer");
    if (amax && (HW & 3)) return fail(FAOCTASR_EUNSUPPORTED, "norm_bwd: faoctasr_out_absmax needs a map size that is a multiple of 4");
    if (act == FAOCTASR_ACT_TANH && !y) return fail(FAOCTASR_EINVAL, "norm_bwd: the tanh derivative needs the forward output");
    const long L = (long)NI * HW;
    const int accumulate = (Cg != R) || accumulate_affine;
    if (accumulate && !accumulate_affine) {
        if (dgamma) (void)hipMemsetAsync(dgamma, 0, sizeof(float) * Cg, st);
        if (dbeta) (void)hipMemsetAsync(dbeta, 0, sizeof(float) * Cg, st);
    }
    if ((HW & 3) == 0 && L <= BN_SMALL_BWD) {
        auto go = [&](auto k) {
            hipLaunchKernelGGL(k, dim3(R), dim3(256), 0, st, x, dy, y, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, dres, R, Cg, HW, L,
                               act, slope, accumulate, amax);
        };
        if (L <= 4096) go(norm_bwd_small_kernel<4>);
        else go(norm_bwd_small_kernel<8>);
        return check_launch("norm_bwd_small");
    }
    int S; long per;
    norm_split(L, R, S, per);
    if ((HW & 3) == 0) {
        auto go = [&](auto kr, auto kd) {
            hipLaunchKernelGGL(kr, dim3(S, R), dim3(256), 0, st, x, dy, y, gamma, beta, save_mean, save_invstd, ws, R, Cg, HW, L, S, act, slope, per);
            hipLaunchKernelGGL(kd, dim3(S, R), dim3(256), 0, st, x, dy, y, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, dres, ws, R, Cg,
                               HW, L, S, act, slope, per, accumulate, amax);
        };
        if (y && act != FAOCTASR_ACT_NONE) go(norm_bwd_reduce_kernel<true, 2>, norm_bwd_dx_kernel<true, 2>);
        else go(norm_bwd_reduce_kernel<false, BN_NU>, norm_bwd_dx_kernel<false, BN_NU>);
    } else {
        hipLaunchKernelGGL(norm_bwd_reduce_generic_kernel, dim3(S, R), dim3(256), 0, st, x, dy, y, gamma, beta, save_mean, save_invstd, ws, R,
                           Cg, HW, L, S, act, slope, per);
        hipLaunchKernelGGL(norm_bwd_dx_generic_kernel, dim3(S, R), dim3(256), 0, st, x, dy, y, gamma, beta, save_mean, save_invstd, dx, dgamma,
                           dbeta, dres, ws, R, Cg, HW, L, S, act, slope, per, accumulate);
    }
    return check_launch("norm_bwd");
}

}  // namespace faoctasr

using namespace faoctasr;

extern "C" {

long faoctasr_bn_workspace_floats(int C) { return (long)C * BN_MAX_SPLIT * 2; }

int faoctasr_batchnorm_train_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                                 float* save_mean, float* save_invstd, float* running_mean, float* running_var, int N, int C,
                                 int HW, float eps, float momentum, int act, float slope, float* workspace, faoctasr_stream_t stream) {
    return norm_fwd(x, gamma, beta, residual, y, save_mean, save_invstd, running_mean, running_var, N, C, C, HW, eps, momentum, act,
                    slope, workspace, (hipStream_t)stream, take_out_absmax());
}

int faoctasr_batchnorm_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* dres,
                                 int N, int C, int HW, int act, float slope, int accumulate_affine, float* workspace,
                                 faoctasr_stream_t stream) {
    return norm_bwd(x, dy, y, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, dres, N, C, C, HW, act, slope, accumulate_affine,
                    workspace, (hipStream_t)stream, take_out_absmax());
}

// InstanceNorm2d = the same kernels over R = N*C rows of one image each (workspace: faoctasr_bn_workspace_floats(N*C))
int faoctasr_instancenorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
                              int N, int C, int HW, float eps, int act, float slope, float* workspace, faoctasr_stream_t stream) {
    return norm_fwd(x, gamma, beta, nullptr, y, save_mean, save_invstd, nullptr, nullptr, 1, N * C, C, HW, eps, 0.f, act, slope,
                    workspace, (hipStream_t)stream, nullptr);
}

int faoctasr_instancenorm_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta, int N, int C, int HW,
                              int act, float slope, float* workspace, faoctasr_stream_t stream) {
    return norm_bwd(x, dy, y, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, nullptr, 1, N * C, C, HW, act, slope, 0, workspace,
                    (hipStream_t)stream, nullptr);
}

}  // extern "C"
