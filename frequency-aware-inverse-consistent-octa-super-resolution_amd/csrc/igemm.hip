// Implicit-GEMM convolution family for gfx950 on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One "tap-list" gather kernel serves nn.Conv2d forward (any stride, zero or reflect padding),
// conv input-gradient, nn.ConvTranspose2d forward and its input-gradient; one split-K kernel
// serves every weight gradient.  GEMM view (per phase of the output grid):
//     Y[m][j] = sum_{k=(c,t)} Wt[m][c][t] * X[n(j)][c][a(j)*SI + oy_t][b(j)*SI + ox_t]
// m = output channel, j = flattened (n,a,b) over the phase's sub-grid, t = tap.  A strided
// transposed convolution is split into SO*SO output-parity phases, each a dense stride-1
// gather over its own subset of taps (the phases partition the KHxKW taps), so no MFMA work
// is spent on structural zeros.
//
// Reference call sites: model.py:102-122 (4x4 s2/s1 convs + bias), 242-258,275-286 (stems),
// 412-414,438 (3x3 @64ch), 431/469 (ConvTranspose2d), 450-451,472-473 (ReflectionPad2d(3)+7x7),
// 458 (3x3 s2), 494,499 (3x3 @256ch); backward = aten::convolution_backward under
// loss.backward() (train.py:238,255,267).
#include <cstring>

#include "common.h"
#include "igemm_geom.h"

namespace faoctasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// ------------------------------------------------------------------------------------------
// gather ("forward-like") kernel.  Block 256 threads = 4 waves (2 along M x 2 along N),
// tile MT x 128 pixels, K chunk 16, register-prefetched staging.
// ------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void igemm_gather_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           const IgemmGeom g, const int ksplit) {
    constexpr int NT = 128, KC = 16, LDA = KC + 1;
    constexpr int MI = MT / 64;          // 32-row MFMA tiles per wave along M
    constexpr int NA = MT * KC / 256;    // A elements staged per thread
    __shared__ float A_s[MT * LDA];
    __shared__ float B_s[KC * NT];
    __shared__ int taps_s[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ph = blockIdx.z / ksplit, ks = blockIdx.z - ph * ksplit;
    const int GH = g.ph_gh[ph], GW = g.ph_gw[ph];
    const int t0 = g.ph_t0[ph], T = g.ph_t0[ph + 1] - t0;
    const long npix = (long)g.N * GH * GW;
    const long j0 = (long)blockIdx.x * NT;
    if (j0 >= npix) return;
    const int m0 = blockIdx.y * MT;
    const int Kfull = g.C * T;
    // split-K: this block reduces chunks [kbeg, K) of the phase's K range and adds its partial tile atomically
    const int nchunks_all = (Kfull + KC - 1) / KC;
    const int cps = (nchunks_all + ksplit - 1) / ksplit;
    const int kbeg = ks * cps * KC;
    int K = (ks + 1) * cps * KC;
    K = K < Kfull ? K : Kfull;
    if (ksplit > 1 && kbeg >= K) return;
    const float invT = T > 0 ? 1.0f / (float)T : 0.f;
    const int IH = g.IH, IW = g.IW;
    const long chw = (long)IH * IW;

    if (tid < 64) taps_s[tid] = g.taps[tid];

    // this thread's pixel for B staging
    const int jj = tid & (NT - 1), rg = tid >> 7;
    const long j = j0 + jj;
    const bool jv = j < npix;
    int n = 0, a = 0, b = 0;
    if (jv) {
        n = (int)(j / ((long)GH * GW));
        int r = (int)(j - (long)n * GH * GW);
        a = r / GW;
        b = r - a * GW;
    }
    const float* xin = x + (long)n * g.C * chw;
    const int iy0 = a * g.SI, ix0 = b * g.SI;

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    float ra[NA], rb[8];
    __syncthreads();   // taps_s visible

    // When the tap count divides the chunk (the 4x4 layers: T = 16 forward, 4 per phase in the input gradient) a thread's
    // element of every chunk is the same (tap, channel-within-chunk): decode it ONCE -- the (c, t) split, the tap lookup, padding
    // test and in-plane offset cost ~30 vector instructions per element and chunk otherwise, and vector instructions are not
    // hidden behind f32 MFMAs on this chip.  Per chunk what is left is base + channel * stride.
    const bool fixed_taps = T > 0 && (KC % T) == 0 && (kbeg % T) == 0;
    const int akk = tid & 15;                                           // A element: (row tid>>4 + 16 i, k offset akk)
    int a_cl = 0;
    long a_off = 0;
    int b_cl[8];
    long b_off[8];                                                      // in-plane offset of B element i, -1 = padding
    if (fixed_taps) {
        a_cl = akk / T;
        a_off = taps_s[t0 + akk % T] >> 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kk = rg + 2 * i;
            b_cl[i] = kk / T;
            const int tp = taps_s[t0 + kk % T];
            int iy = iy0 + (tp & 0xff) - 64, ix = ix0 + ((tp >> 8) & 0xff) - 64;
            if (g.reflect) {
                iy = reflect_idx(iy, IH);
                ix = reflect_idx(ix, IW);
            }
            b_off[i] = (jv && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) ? (long)iy * IW + ix : -1;
        }
    }

    auto load_chunk = [&](int k0) {
        if (fixed_taps) {
            const int cb = k0 / T;                                      // uniform: first channel of the chunk
            const int ca = cb + a_cl;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int m = m0 + (tid >> 4) + 16 * i;
                ra[i] = (m < g.M && ca < g.C && k0 + akk < K) ? w[(long)m * g.wsm + (long)ca * g.wsc + a_off] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cb + b_cl[i];
                rb[i] = (b_off[i] >= 0 && c < g.C && k0 + rg + 2 * i < K) ? xin[(long)c * chw + b_off[i]] : 0.f;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i;
            const int ml = e >> 4, kk = e & 15;
            const int k = k0 + kk, m = m0 + ml;
            float v = 0.f;
            if (m < g.M && k < K) {
                const int c = (int)(((float)k + 0.5f) * invT);
                const int t = k - c * T;
                v = w[(long)m * g.wsm + (long)c * g.wsc + (taps_s[t0 + t] >> 16)];
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = k0 + rg + 2 * i;
            float v = 0.f;
            if (jv && k < K) {
                const int c = (int)(((float)k + 0.5f) * invT);
                const int t = k - c * T;
                const int tp = taps_s[t0 + t];
                int iy = iy0 + (tp & 0xff) - 64, ix = ix0 + ((tp >> 8) & 0xff) - 64;
                if (g.reflect) {
                    iy = reflect_idx(iy, IH);
                    ix = reflect_idx(ix, IW);
                }
                if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) v = xin[(long)c * chw + (long)iy * IW + ix];
            }
            rb[i] = v;
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    load_chunk(kbeg);
    for (int k0 = kbeg; k0 < K || k0 == kbeg; k0 += KC) {
        __syncthreads();   // previous chunk's MFMA reads are done
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i;
            A_s[(e >> 4) * LDA + (e & 15)] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) B_s[(rg + 2 * i) * NT + jj] = rb[i];
        __syncthreads();
        if (k0 + KC < K) load_chunk(k0 + KC);   // in flight under the MFMAs below
#pragma unroll
        for (int kk0 = 0; kk0 < KC; kk0 += 2) {
            float af[MI], bf[2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = A_s[(wm * (MT / 2) + mi * 32 + l31) * LDA + kk0 + lh];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bf[ni] = B_s[(kk0 + lh) * NT + wn * 64 + ni * 32 + l31];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        if (K <= kbeg) break;
    }

    // epilogue: C/D layout of the 32x32 MFMA: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int py = g.ph_py[ph], px = g.ph_px[ph];
    const long ohw = (long)g.OH * g.OW;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const long jo = j0 + wn * 64 + ni * 32 + l31;
        if (jo >= npix) continue;
        const int no = (int)(jo / ((long)GH * GW));
        const int r = (int)(jo - (long)no * GH * GW);
        const int ao = r / GW, bo = r - ao * GW;
        const long obase = (long)no * g.M * ohw + (long)(ao * g.SO + py) * g.OW + (bo * g.SO + px);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int m = m0 + wm * (MT / 2) + mi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                if (m < g.M) {
                    float v = acc[mi][ni][rr];
                    if (bias && ks == 0) v += bias[m];
                    if (ksplit > 1) atomicAdd(y + obase + (long)m * ohw, v);
                    else y[obase + (long)m * ohw] = act_apply(v, g.act, g.slope);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// weight-gradient kernel (split-K over pixels, fp32 atomics into a zeroed dW).
//   dW[m][c][t] = sum_j dY[n(j)][m][a][b] * X[n(j)][c][a*s+oy_t][b*s+ox_t]
// tile 64 (m) x 64 (columns (c,t)), pixel chunk 32.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dw, const IgemmGeom g, int chunks_per_slice) {
    constexpr int MT = 64, CT = 64, PJ = 32, LD = PJ + 1;
    __shared__ float A_s[MT * LD];
    __shared__ float B_s[CT * LD];
    __shared__ int taps_s[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = g.ph_t0[1];
    const int K = g.C * T;                      // number of weight columns per output channel
    const float invT = 1.0f / (float)T;
    const int m0 = blockIdx.y * MT, col0 = blockIdx.x * CT;
    const int OH = g.OH, OW = g.OW, IH = g.IH, IW = g.IW;
    const long npix = (long)g.N * OH * OW;
    const long nchunks = (npix + PJ - 1) / PJ;
    const long ch_begin = (long)blockIdx.z * chunks_per_slice;
    long ch_end = ch_begin + chunks_per_slice;
    if (ch_end > nchunks) ch_end = nchunks;
    if (ch_begin >= ch_end) return;
    if (tid < 64) taps_s[tid] = g.taps[tid];
    __syncthreads();

    const int pl = tid & 31, rg = tid >> 5;     // pixel within chunk, row group (8 groups)
    // per-thread fixed columns: cl = rg + 8*i
    int cc[8], coy[8], cox[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int col = col0 + rg + 8 * i;
        if (col < K) {
            const int c = (int)(((float)col + 0.5f) * invT);
            const int t = col - c * T;
            const int tp = taps_s[t];
            cc[i] = c;
            coy[i] = (tp & 0xff) - 64;
            cox[i] = ((tp >> 8) & 0xff) - 64;
        } else {
            cc[i] = -1; coy[i] = 0; cox[i] = 0;
        }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const long ihw = (long)IH * IW, ohw = (long)OH * OW;
    float ra[8], rb[8];

    auto load_chunk = [&](long ch) {
        const long p = ch * PJ + pl;
        const bool pv = p < npix;
        int n = 0, a = 0, b = 0;
        if (pv) {
            n = (int)(p / ohw);
            const int r = (int)(p - (long)n * ohw);
            a = r / OW;
            b = r - a * OW;
        }
        const float* dyp = dy + (long)n * g.M * ohw + (long)a * OW + b;
        const float* xp = x + (long)n * g.C * ihw;
        const int iy0 = a * g.SI, ix0 = b * g.SI;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + rg + 8 * i;
            ra[i] = (pv && m < g.M) ? dyp[(long)m * ohw] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = 0.f;
            if (pv && cc[i] >= 0) {
                int iy = iy0 + coy[i], ix = ix0 + cox[i];
                if (g.reflect) {
                    iy = reflect_idx(iy, IH);
                    ix = reflect_idx(ix, IW);
                }
                if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) v = xp[(long)cc[i] * ihw + (long)iy * IW + ix];
            }
            rb[i] = v;
        }
    };

    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    load_chunk(ch_begin);
    for (long ch = ch_begin; ch < ch_end; ++ch) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            A_s[(rg + 8 * i) * LD + pl] = ra[i];
            B_s[(rg + 8 * i) * LD + pl] = rb[i];
        }
        __syncthreads();
        if (ch + 1 < ch_end) load_chunk(ch + 1);
        // (Round 4, measured and not kept: all 32 fragment reads of the chunk first, then its 16 MFMAs behind counted waits, here and in
        // igemm_gather_kernel -- the compiled loop is `ds_read2_b32 x2; s_waitcnt lgkmcnt(0); v_mfma x2` eight times.  Slower: 256 -> 512
        // 4x4 stride 2 @32^2 weight gradient 137 -> 155 us, 512 -> 512 @16^2 76-84 -> 88-89; these blocks are 4 waves with several blocks
        // per CU, and the other waves already cover a wave's LDS round trip.)
#pragma unroll
        for (int kk0 = 0; kk0 < PJ; kk0 += 2) {
            const float af = A_s[(wm * 32 + l31) * LD + kk0 + lh];
            const float bf = B_s[(wn * 32 + l31) * LD + kk0 + lh];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
    }
    // epilogue: row (m) from the register index, column (c,t) from the lane
    const int col = col0 + wn * 32 + l31;
    if (col < K) {
        const int c = (int)(((float)col + 0.5f) * invT);
        const int t = col - c * T;
        float* dst = dw + (long)c * g.wsc + (taps_s[t] >> 16);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int m = m0 + wm * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            if (m < g.M) atomicAdd(dst + (long)m * g.wsm, acc[rr]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side: geometry builders and launchers
// ------------------------------------------------------------------------------------------
static int pack_tap(int oy, int ox, int widx) { return (oy + 64) | ((ox + 64) << 8) | (widx << 16); }

// forward-mode gather: iy = a*stride + (kh - pad)
static int geom_fwd(IgemmGeom& g, int N, int C, int IH, int IW, int M, int OH, int OW, int KH, int KW, int stride, int pad,
                    int reflect, long wsm, long wsc) {
    if (KH * KW > 64 || KH > 16 || KW > 16) return fail(FAOCTASR_EUNSUPPORTED, "kernel %dx%d too large", KH, KW);
    if (pad > 60) return fail(FAOCTASR_EUNSUPPORTED, "pad %d too large", pad);
    if (reflect && (pad >= IH || pad >= IW)) return fail(FAOCTASR_EINVAL, "reflect pad %d >= input size", pad);
    g = IgemmGeom{};
    g.N = N; g.C = C; g.IH = IH; g.IW = IW; g.M = M; g.OH = OH; g.OW = OW;
    g.SI = stride; g.SO = 1; g.nphase = 1; g.reflect = reflect; g.wsm = wsm; g.wsc = wsc;
    g.ph_py[0] = 0; g.ph_px[0] = 0; g.ph_gh[0] = OH; g.ph_gw[0] = OW; g.ph_t0[0] = 0; g.ph_t0[1] = KH * KW;
    for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KW; ++kw) g.taps[kh * KW + kw] = pack_tap(kh - pad, kw - pad, kh * KW + kw);
    return FAOCTASR_OK;
}

// transposed-mode gather: out (oy,ox) reads src ((oy+pad-kh)/stride, ...) when divisible
static int geom_transposed(IgemmGeom& g, int N, int C, int IH, int IW, int M, int OH, int OW, int KH, int KW, int stride,
                           int pad, long wsm, long wsc) {
    if (KH * KW > 64) return fail(FAOCTASR_EUNSUPPORTED, "kernel %dx%d too large", KH, KW);
    if (stride < 1 || stride > 2) return fail(FAOCTASR_EUNSUPPORTED, "stride %d unsupported for transposed gather", stride);
    g = IgemmGeom{};
    g.N = N; g.C = C; g.IH = IH; g.IW = IW; g.M = M; g.OH = OH; g.OW = OW;
    g.SI = 1; g.SO = stride; g.nphase = stride * stride; g.reflect = 0; g.wsm = wsm; g.wsc = wsc;
    int nt = 0;
    for (int py = 0; py < stride; ++py)
        for (int px = 0; px < stride; ++px) {
            const int ph = py * stride + px;
            g.ph_py[ph] = py; g.ph_px[ph] = px;
            g.ph_gh[ph] = OH > py ? (OH - py + stride - 1) / stride : 0;
            g.ph_gw[ph] = OW > px ? (OW - px + stride - 1) / stride : 0;
            g.ph_t0[ph] = nt;
            for (int kh = 0; kh < KH; ++kh) {
                int ny = py + pad - kh;
                if (((ny % stride) + stride) % stride) continue;
                for (int kw = 0; kw < KW; ++kw) {
                    int nx = px + pad - kw;
                    if (((nx % stride) + stride) % stride) continue;
                    // exact division (ny, nx are multiples of stride, possibly negative)
                    g.taps[nt++] = pack_tap(ny / stride, nx / stride, kh * KW + kw);
                }
            }
            g.ph_t0[ph + 1] = nt;
        }
    return FAOCTASR_OK;
}

static int launch_gather(const float* x, const float* w, const float* bias, float* y, IgemmGeom& g, int act, float slope,
                         hipStream_t s) {
    g.act = act; g.slope = slope;
    long maxpix = 0;
    for (int p = 0; p < g.nphase; ++p) {
        long np = (long)g.N * g.ph_gh[p] * g.ph_gw[p];
        if (np > maxpix) maxpix = np;
    }
    if (maxpix == 0 || g.M == 0) return FAOCTASR_OK;
    const long gx = (maxpix + 127) / 128;
    if (gx > 0x7fffffffL) return fail(FAOCTASR_EUNSUPPORTED, "grid too large");
    const long blocks128 = gx * ((g.M + 127) / 128) * g.nphase;
    if (g.M >= 128 && blocks128 >= 512) {
        dim3 grid((unsigned)gx, (g.M + 127) / 128, g.nphase);
        hipLaunchKernelGGL(igemm_gather_kernel<128>, grid, dim3(256), 0, s, x, w, bias, y, g, 1);
    } else {
        // few pixels and a long reduction (the deep discriminator layers: 8x8 maps, K = 8192): split K over
        // blockIdx.z and accumulate with fp32 atomics into a zeroed output (no fused activation in that mode)
        const long blocks64 = gx * ((g.M + 63) / 64) * g.nphase;
        int maxT = 0;
        for (int p = 0; p < g.nphase; ++p) maxT = g.ph_t0[p + 1] - g.ph_t0[p] > maxT ? g.ph_t0[p + 1] - g.ph_t0[p] : maxT;
        const int nchunks = (g.C * maxT + 15) / 16;
        int ksplit = 1;
        if (act == FAOCTASR_ACT_NONE && blocks64 < 384 && nchunks >= 8) {
            ksplit = (int)((768 + blocks64 - 1) / blocks64);
            if (ksplit > nchunks / 4) ksplit = nchunks / 4;
            if (ksplit < 1) ksplit = 1;
        }
        if (g_no_split_k) ksplit = 1;
        if (ksplit > 1) {
            hipError_t e = hipMemsetAsync(y, 0, sizeof(float) * (size_t)g.N * g.M * g.OH * g.OW, s);
            if (e != hipSuccess) return fail(FAOCTASR_EHIP, "memset y: %s", hipGetErrorString(e));
        }
        dim3 grid((unsigned)gx, (g.M + 63) / 64, g.nphase * ksplit);
        hipLaunchKernelGGL(igemm_gather_kernel<64>, grid, dim3(256), 0, s, x, w, bias, y, g, ksplit);
    }
    return check_launch("igemm_gather");
}

static int launch_wgrad(const float* x, const float* dy, float* dw, IgemmGeom& g, long dw_elems, int accumulate, hipStream_t s) {
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * dw_elems, s);
        if (e != hipSuccess) return fail(FAOCTASR_EHIP, "memset dw: %s", hipGetErrorString(e));
    }
    const int K = g.C * g.ph_t0[1];
    const long npix = (long)g.N * g.OH * g.OW;
    if (npix == 0 || K == 0 || g.M == 0) return FAOCTASR_OK;
    const long nchunks = (npix + 31) / 32;
    const int gx = (K + 63) / 64, gy = (g.M + 63) / 64;
    long slices = 2048 / ((long)gx * gy);
    if (slices < 1) slices = 1;
    if (slices > nchunks) slices = nchunks;
    if (slices > 65535) slices = 65535;
    const int cps = (int)((nchunks + slices - 1) / slices);
    slices = (nchunks + cps - 1) / cps;
    dim3 grid(gx, gy, (unsigned)slices);
    hipLaunchKernelGGL(igemm_wgrad_kernel, grid, dim3(256), 0, s, x, dy, dw, g, cps);
    return check_launch("igemm_wgrad");
}

static bool bad_ptr(const void* a, const void* b, const void* c) { return !a || !b || !c; }

// the absmax slots faoctasr_conv_set_scales left for this thread's next convolution-type call: taken (and cleared) by that call
thread_local const unsigned* g_scale_a = nullptr;
thread_local const unsigned* g_scale_b = nullptr;
thread_local bool g_wgrad_dry_run = false;
thread_local float* g_wgrad_ws = nullptr;
thread_local long g_wgrad_ws_floats = 0;
// a weight-gradient entry takes the workspace for the duration of the call and clears it on every way out
struct WgradWsScope {
    ~WgradWsScope() { g_wgrad_ws = nullptr; g_wgrad_ws_floats = 0; }
};
static const unsigned* take_scale_a() {
    const unsigned* a = g_scale_a;
    g_scale_a = g_scale_b = nullptr;
    return a;
}
// the tensor faoctasr_conv_set_residual left for this thread's next gather call
thread_local const float* g_conv_residual = nullptr;
static const float* take_residual() {
    const float* r = g_conv_residual;
    g_conv_residual = nullptr;
    return r;
}

// wpack_state: 0 = no pack buffer (flat kernel), 1 = pack the weights into wpack now, 2 = wpack already holds them
// precision: 0 = f32 MFMA, Winograd F(2x2,3x3) for the dense stride-1 3x3 gathers and direct implicit GEMM elsewhere;
//            1 = f32 MFMA, direct implicit GEMM only; 2 = bf16x3 split operands on the bf16 MFMA where the layer shape allows it
// sink: record the weight-packing job this call would launch (wpack_state 1) and launch nothing (faoctasr_conv_pack_job);
// sink->blocks stays 0 when the call's route uses no packed image.
static int run_gather(const float* x, const float* w, const float* bias, float* y, IgemmGeom& g, int act, float slope, float* wpack,
                      int wpack_state, int precision, hipStream_t s, PackJob* sink = nullptr, const unsigned* slot_a = nullptr,
                      const float* res = nullptr) {
    // res (faoctasr_conv_set_residual): y = gather(...) + res.  The split kernels add it in their epilogue; every other route runs
    // as usual and is followed by one in-place y += res pass.
    if (res && !sink) {
        if (wpack && wpack_state && (precision & 0xff) >= 2) {
            const int rc = split_try(g, x, w, bias, y, act, slope, wpack, wpack_state, s, nullptr, (precision & 0xff) == 3, slot_a, res);
            if (rc != 0) { set_route(ROUTE_BF16X3); return rc < 0 ? rc : FAOCTASR_OK; }
        }
        const int rc = run_gather(x, w, bias, y, g, act, slope, wpack, wpack_state, precision, s, nullptr, slot_a, nullptr);
        if (rc) return rc;
        return faoctasr_axpby(y, res, y, (long)g.N * g.M * g.OH * g.OW, 1.f, 1.f, (faoctasr_stream_t)s);
    }
    // FAOCTASR_CONV_NO_SPLIT_K: a reproducible result (one block owns the whole reduction of an output element) for this call
    struct Scope { int prev; Scope(int v) : prev(g_no_split_k) { g_no_split_k = v; } ~Scope() { g_no_split_k = prev; } } scope((precision & FAOCTASR_CONV_NO_SPLIT_K) ? 1 : 0);
    precision &= 0xff;
    if (precision < 0 || precision > 3) return fail(FAOCTASR_EINVAL, "unknown conv precision %d (0 = f32, 1 = f32 direct, 2 = bf16x3, 3 = f16x2)", precision);
    if (wpack && wpack_state && precision >= 2) {
        const int rc = split_try(g, x, w, bias, y, act, slope, wpack, wpack_state, s, sink, precision == 3, slot_a);
        if (rc != 0 && sink) { sink->blocks = pack_job_blocks(sink->total / (8L * sink->g.split.Mpad)); return FAOCTASR_OK; }
        if (rc != 0) { set_route(ROUTE_BF16X3); return rc < 0 ? rc : FAOCTASR_OK; }
    }
    if (wpack && wpack_state && precision == 0) {
        const int rc = wino_try(g, x, w, bias, y, act, slope, wpack, wpack_state, s, sink);
        if (rc != 0 && sink) { sink->blocks = pack_job_blocks((long)sink->g.wino.mtiles * sink->g.wino.nchunks); return FAOCTASR_OK; }
        if (rc != 0) { set_route(ROUTE_WINOGRAD); return rc < 0 ? rc : FAOCTASR_OK; }
    }
    if (wpack && wpack_state) {
        PatchGeom pg;
        int rc = patch_geom_from(g, pg);
        if (rc) return rc;
        if (sink) {
            sink->type = PACK_PATCH; sink->w = w; sink->wp = wpack; sink->g.patch = pg; sink->total = pg.pack_off[4];
            sink->blocks = pack_job_blocks(sink->total / pg.Mpad);
            return FAOCTASR_OK;
        }
        if (wpack_state == 1) {
            rc = launch_pack(w, wpack, pg, s);
            if (rc) return rc;
        }
        rc = launch_patch(x, wpack, bias, y, pg, act, slope, s);
        if (rc != 0) { set_route(ROUTE_PATCH); return rc < 0 ? rc : FAOCTASR_OK; }
        rc = launch_narrow(x, wpack, bias, y, pg, act, slope, s);
        if (rc != 0) { set_route(ROUTE_NARROW); return rc < 0 ? rc : FAOCTASR_OK; }
    }
    set_route(ROUTE_GATHER_FLAT);
    return launch_gather(x, w, bias, y, g, act, slope, s);
}

static long wpack_floats(IgemmGeom& g, int precision) {
    PatchGeom pg;
    if (patch_geom_from(g, pg)) return 0;
    long n = patch_pack_floats(pg);
    if (precision >= 2) {
        const long m = split_pack_floats_for(g);
        n = m > n ? m : n;
    }
    if (precision == 0) {
        const long m = wino_pack_floats_for(g);
        n = m > n ? m : n;
    }
    return n;
}

}  // namespace faoctasr

using namespace faoctasr;

extern "C" {

// kind: 0 conv2d fwd, 1 conv2d dgrad, 2 conv_transpose2d fwd, 3 conv_transpose2d dgrad.  (C, M) as in the matching call.
long faoctasr_conv_wpack_floats(int kind, int C, int M, int KH, int KW, int stride, int pad, int precision) {
    IgemmGeom g;
    const long kk = (long)KH * KW;
    // spatial sizes do not enter the packed layout; use a nominal 64x64 grid
    int rc;
    switch (kind) {
        case 0: rc = geom_fwd(g, 1, C, 64, 64, M, 64, 64, KH, KW, stride, pad, 0, (long)C * kk, kk); break;
        case 1: rc = geom_transposed(g, 1, M, 64, 64, C, 64, 64, KH, KW, stride, pad, kk, (long)C * kk); break;
        case 2: rc = geom_transposed(g, 1, C, 64, 64, M, 64, 64, KH, KW, stride, pad, kk, (long)M * kk); break;
        case 3: rc = geom_fwd(g, 1, M, 64, 64, C, 64, 64, KH, KW, stride, pad, 0, (long)M * kk, kk); break;
        default: return fail(FAOCTASR_EINVAL, "conv_wpack_floats: unknown kind %d", kind);
    }
    if (rc) return rc;
    return wpack_floats(g, precision & 0xff);
}

// The packing job a wpack_state == 1 call of the matching entry point would launch, written to a host slot of
// FAOCTASR_PACK_JOB_BYTES; (N, C, IH, IW, M, ...) exactly as that call receives them.  Returns the job's blocks (0: the call's
// route has no packed image, the slot is then unused), which the caller accumulates into the next job's block_base.
long faoctasr_conv_pack_job(void* job_host, long block_base, int kind, const float* w, float* wpack, int N, int C, int IH, int IW, int M,
                            int KH, int KW, int stride, int pad, int reflect, int out_pad, int precision) {
    if (!job_host || !w || !wpack) return fail(FAOCTASR_EINVAL, "conv_pack_job: null pointer");
    if (N <= 0 || C <= 0 || M <= 0 || stride <= 0 || pad < 0 || block_base < 0) return fail(FAOCTASR_EINVAL, "conv_pack_job: bad shape");
    IgemmGeom g;
    const long kk = (long)KH * KW;
    int rc, OH, OW;
    switch (kind) {
        case 0:
            OH = (IH + 2 * pad - KH) / stride + 1; OW = (IW + 2 * pad - KW) / stride + 1;
            if (M == 1 && stride == 1 && !reflect && 2 * pad == KH - 1 && 2 * pad == KW - 1 && KH <= 7 && IW >= 32) return 0;   // VALU head
            rc = geom_fwd(g, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, reflect, (long)C * kk, kk);
            break;
        case 1:
            OH = (IH + 2 * pad - KH) / stride + 1; OW = (IW + 2 * pad - KW) / stride + 1;
            if (stem_dgrad_eligible(C, IH, IW, M, KH, KW, stride, pad)) return 0;                    // VALU stem: reads the weights in place
            rc = geom_transposed(g, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, kk, (long)C * kk);
            break;
        case 2:
            OH = (IH - 1) * stride - 2 * pad + KH + out_pad; OW = (IW - 1) * stride - 2 * pad + KW + out_pad;
            rc = geom_transposed(g, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, kk, (long)M * kk);
            break;
        case 3:
            OH = (IH - 1) * stride - 2 * pad + KH + out_pad; OW = (IW - 1) * stride - 2 * pad + KW + out_pad;
            rc = geom_fwd(g, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, 0, (long)M * kk, kk);
            break;
        default: return fail(FAOCTASR_EINVAL, "conv_pack_job: unknown kind %d", kind);
    }
    if (OH <= 0 || OW <= 0) return fail(FAOCTASR_EINVAL, "conv_pack_job: bad shape");
    if (rc) return rc;
    PackJob job{};
    rc = run_gather(nullptr, w, nullptr, nullptr, g, FAOCTASR_ACT_NONE, 0.f, wpack, 1, precision, nullptr, &job);
    if (rc) return rc;
    if (job.blocks <= 0 || job.total <= 0) return 0;
    job.block0 = block_base;
    memset(job_host, 0, PACK_JOB_BYTES);
    memcpy(job_host, &job, sizeof(job));
    return job.blocks;
}

int faoctasr_conv_set_scales(const unsigned* slot_a, const unsigned* slot_b) {
    g_scale_a = slot_a;
    g_scale_b = slot_b;
    return FAOCTASR_OK;
}

// Which absmax slots the precision-3 form of a convolution-type call reads (bit 0: slot a, bit 1: slot b; 0: the call runs on a kernel
// that needs none -- narrow map, stem, head).  kind 0..3 as faoctasr_conv_pack_job, 4 = conv2d_wgrad, 5 = conv_transpose2d_wgrad;
// the remaining arguments exactly as the call receives them.  The answer comes from the dispatch code itself (the packing-job and
// dry-run paths of the launchers), so a caller that asks never computes a maximum for nothing and never misses one.
int faoctasr_conv_needs_scales(int kind, int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad, int reflect, int out_pad) {
    if (N <= 0 || C <= 0 || M <= 0 || stride <= 0 || pad < 0 || KH <= 0 || KW <= 0) return fail(FAOCTASR_EINVAL, "conv_needs_scales: bad shape");
    IgemmGeom g;
    const long kk = (long)KH * KW;
    int rc, OH, OW;
    if (kind == 0 || kind == 1 || kind == 4) { OH = (IH + 2 * pad - KH) / stride + 1; OW = (IW + 2 * pad - KW) / stride + 1; }
    else { OH = (IH - 1) * stride - 2 * pad + KH + out_pad; OW = (IW - 1) * stride - 2 * pad + KW + out_pad; }
    if (OH <= 0 || OW <= 0) return fail(FAOCTASR_EINVAL, "conv_needs_scales: bad shape");
    const bool head = M == 1 && stride == 1 && !reflect && 2 * pad == KH - 1 && 2 * pad == KW - 1 && KH <= 7 && IW >= 32;
    if (kind >= 4) {
        struct Dry { Dry() { g_wgrad_dry_run = true; } ~Dry() { g_wgrad_dry_run = false; } } dry;
        if (kind == 4) {
            if (head || stem_wgrad_eligible(C, IH, IW, KH, KW, stride, pad, reflect) || OW < 24) return 0;
            rc = launch_wgrad_x3(nullptr, nullptr, nullptr, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, reflect, (long)C * kk, kk, nullptr, 1, nullptr, nullptr);
        } else if (kind == 5) {
            if (IW < 24) return 0;
            rc = launch_wgrad_x3(nullptr, nullptr, nullptr, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, 0, (long)M * kk, kk, nullptr, 1, nullptr, nullptr);
        } else {
            return fail(FAOCTASR_EINVAL, "conv_needs_scales: unknown kind %d", kind);
        }
        return rc < 0 ? rc : (rc ? 3 : 0);
    }
    switch (kind) {
        case 0:
            if (head) return 0;
            rc = geom_fwd(g, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, reflect, (long)C * kk, kk);
            break;
        case 1:
            if (stem_dgrad_eligible(C, IH, IW, M, KH, KW, stride, pad)) return 0;
            rc = geom_transposed(g, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, kk, (long)C * kk);
            break;
        case 2: rc = geom_transposed(g, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, kk, (long)M * kk); break;
        default: rc = geom_fwd(g, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, 0, (long)M * kk, kk); break;
    }
    if (rc) return rc;
    PackJob job{};
    float dummy;
    rc = split_try(g, nullptr, &dummy, nullptr, nullptr, FAOCTASR_ACT_NONE, 0.f, &dummy, 1, nullptr, &job, 1, nullptr);
    return rc < 0 ? rc : (rc ? 1 : 0);
}

int faoctasr_conv_set_residual(const float* residual) {
    g_conv_residual = residual;
    return FAOCTASR_OK;
}

int faoctasr_conv_set_workspace(float* ws, long nfloats) {
    g_wgrad_ws = ws;
    g_wgrad_ws_floats = ws ? nfloats : 0;
    return FAOCTASR_OK;
}

long faoctasr_conv_wgrad_workspace_floats(int C, int M, int KH, int KW, int stride) {
    // slices x the extent of dW; slices = the pixel ranges of a one-block-per-CU grid: 256 / ((C / 64) (M / 64)) for the stride-1 3x3
    // kernel, a further / KH for the kernels that give every kernel row its own block (wgrad_x3.hip)
    if (C <= 0 || M <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return fail(FAOCTASR_EINVAL, "conv_wgrad_workspace_floats: bad shape");
    if ((C & 63) || (M & 63)) return 0;                   // not a shape the split weight-gradient kernels take
    long blocks = (long)(C / 64) * (M / 64);
    if (!(stride == 1 && KH == 3)) blocks *= KH;
    long slices = 256 / blocks;
    slices = slices < 1 ? 1 : slices;
    return slices * (long)C * M * KH * KW;
}

int faoctasr_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int IH, int IW, int M,
                        int KH, int KW, int stride, int pad, int reflect, int act, float slope, float* wpack, int wpack_state,
                        int precision, faoctasr_stream_t stream) {
    const unsigned* const slot_a = take_scale_a();
    const float* const res = take_residual();
    if (bad_ptr(x, w, y)) return fail(FAOCTASR_EINVAL, "conv2d_fwd: null pointer");
    if (N < 0 || C <= 0 || M <= 0 || stride <= 0 || pad < 0) return fail(FAOCTASR_EINVAL, "conv2d_fwd: bad shape");
    const int OH = (IH + 2 * pad - KH) / stride + 1, OW = (IW + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return fail(FAOCTASR_EINVAL, "conv2d_fwd: kernel (%d x %d) larger than padded input (%d x %d)", KH, KW, IH + 2 * pad, IW + 2 * pad);
    if (M == 1 && stride == 1 && !reflect && 2 * pad == KH - 1 && 2 * pad == KW - 1 && KH <= 7 && IW >= 32)
    {
        set_route(ROUTE_M1_FWD);
        const int rc1 = launch_conv_m1_fwd(x, w, bias, y, N, C, IH, IW, KH, KW, pad, act, slope, (hipStream_t)stream);
        if (rc1 || !res) return rc1;
        return faoctasr_axpby(y, res, y, (long)N * M * OH * OW, 1.f, 1.f, stream);
    }
    IgemmGeom g;
    int rc = geom_fwd(g, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, reflect, (long)C * KH * KW, (long)KH * KW);
    if (rc) return rc;
    return run_gather(x, w, bias, y, g, act, slope, wpack, wpack_state, precision, (hipStream_t)stream, nullptr, slot_a, res);
}

int faoctasr_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int C, int IH, int IW, int M, int KH, int KW,
                          int stride, int pad, float* wpack, int wpack_state, int precision, faoctasr_stream_t stream) {
    const unsigned* const slot_a = take_scale_a();
    const float* const res = take_residual();
    if (bad_ptr(dy, w, dx)) return fail(FAOCTASR_EINVAL, "conv2d_dgrad: null pointer");
    const int OH = (IH + 2 * pad - KH) / stride + 1, OW = (IW + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return fail(FAOCTASR_EINVAL, "conv2d_dgrad: bad shape");
    {
        const int rc = launch_stem_dgrad(dy, w, dx, N, C, IH, IW, M, KH, KW, stride, pad, (hipStream_t)stream);
        if (rc != 0) {
            set_route(ROUTE_STEM_DGRAD);
            if (rc < 0 || !res) return rc < 0 ? rc : FAOCTASR_OK;
            return faoctasr_axpby(dx, res, dx, (long)N * C * IH * IW, 1.f, 1.f, stream);
        }
    }
    IgemmGeom g;
    // source = dy [N,M,OH,OW] (gathered channels = M), output = dx [N,C,IH,IW]; w[m][c][t]
    int rc = geom_transposed(g, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, (long)KH * KW, (long)C * KH * KW);
    if (rc) return rc;
    return run_gather(dy, w, nullptr, dx, g, FAOCTASR_ACT_NONE, 0.f, wpack, wpack_state, precision, (hipStream_t)stream, nullptr, slot_a, res);
}

int faoctasr_conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int KH, int KW,
                          int stride, int pad, int reflect, int accumulate, int precision, faoctasr_stream_t stream) {
    const unsigned *const slot_x = g_scale_a, *const slot_dy = g_scale_b;
    g_scale_a = g_scale_b = nullptr;
    WgradWsScope ws_scope;
    if (bad_ptr(x, dy, dw)) return fail(FAOCTASR_EINVAL, "conv2d_wgrad: null pointer");
    precision &= 0xff;             // the weight gradients have no split-K policy to switch: they always accumulate with atomics
    if (precision < 0 || precision > 3) return fail(FAOCTASR_EINVAL, "unknown conv precision %d (0 = f32, 1 = f32 direct, 2 = bf16x3, 3 = f16x2)", precision);
    if (precision == 3 && (!slot_x || !slot_dy)) return fail(FAOCTASR_EINVAL, "conv2d_wgrad: precision 3 (f16x2) needs both absmax slots: faoctasr_conv_set_scales");
    const int OH = (IH + 2 * pad - KH) / stride + 1, OW = (IW + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return fail(FAOCTASR_EINVAL, "conv2d_wgrad: bad shape");
    if (M == 1 && stride == 1 && !reflect && 2 * pad == KH - 1 && 2 * pad == KW - 1 && KH <= 7 && IW >= 32)
    {
        set_route(ROUTE_M1_WGRAD);
        return launch_conv_m1_wgrad(x, dy, dw, N, C, IH, IW, KH, KW, pad, accumulate, (hipStream_t)stream);
    }
    {
        const int rc = launch_stem_wgrad(x, dy, dw, N, C, IH, IW, M, KH, KW, stride, pad, reflect, accumulate, (hipStream_t)stream);
        if (rc != 0) { set_route(ROUTE_STEM_WGRAD); return rc < 0 ? rc : FAOCTASR_OK; }
    }
    IgemmGeom g;
    int rc = geom_fwd(g, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, reflect, (long)C * KH * KW, (long)KH * KW);
    if (rc) return rc;
    if (OW >= 24) {      // wide maps: LDS-patch weight gradient (atomics into dw, zeroed here unless accumulating)
        if (!accumulate && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)M * C * KH * KW, (hipStream_t)stream) != hipSuccess)
            return fail(FAOCTASR_EHIP, "memset dw failed");
        if (precision >= 2) {      // split-precision operands on the 16-bit MFMA where the layer shape allows it
            rc = launch_wgrad_x3(x, dy, dw, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, reflect, (long)C * KH * KW, (long)KH * KW,
                                 (hipStream_t)stream, precision == 3, slot_x, slot_dy);
            if (rc != 0) { set_route(ROUTE_WGRAD_X3); return rc < 0 ? rc : FAOCTASR_OK; }
        }
        rc = launch_wgrad_s1(x, dy, dw, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, reflect, (long)C * KH * KW, (long)KH * KW,
                             (hipStream_t)stream);
        if (rc != 0) { set_route(ROUTE_WGRAD_S1); return rc < 0 ? rc : FAOCTASR_OK; }
        rc = launch_wgrad_patch(x, dy, dw, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, reflect, (long)C * KH * KW, (long)KH * KW,
                                (hipStream_t)stream);
        if (rc != 0) { set_route(ROUTE_WGRAD_PATCH); return rc < 0 ? rc : FAOCTASR_OK; }
        set_route(ROUTE_WGRAD_FLAT);
        return launch_wgrad(x, dy, dw, g, (long)M * C * KH * KW, 1, (hipStream_t)stream);
    }
    set_route(ROUTE_WGRAD_FLAT);
    return launch_wgrad(x, dy, dw, g, (long)M * C * KH * KW, accumulate, (hipStream_t)stream);
}

int faoctasr_conv_transpose2d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int IH, int IW,
                                  int M, int KH, int KW, int stride, int pad, int out_pad, int act, float slope, float* wpack,
                                  int wpack_state, int precision, faoctasr_stream_t stream) {
    const unsigned* const slot_a = take_scale_a();
    const float* const res = take_residual();
    if (bad_ptr(x, w, y)) return fail(FAOCTASR_EINVAL, "conv_transpose2d_fwd: null pointer");
    const int OH = (IH - 1) * stride - 2 * pad + KH + out_pad, OW = (IW - 1) * stride - 2 * pad + KW + out_pad;
    if (OH <= 0 || OW <= 0) return fail(FAOCTASR_EINVAL, "conv_transpose2d_fwd: bad shape");
    IgemmGeom g;
    // w[c][m][t]: m stride = KK, c stride = M*KK
    int rc = geom_transposed(g, N, C, IH, IW, M, OH, OW, KH, KW, stride, pad, (long)KH * KW, (long)M * KH * KW);
    if (rc) return rc;
    return run_gather(x, w, bias, y, g, act, slope, wpack, wpack_state, precision, (hipStream_t)stream, nullptr, slot_a, res);
}

int faoctasr_conv_transpose2d_dgrad(const float* dy, const float* w, float* dx, int N, int C, int IH, int IW, int M, int KH,
                                    int KW, int stride, int pad, int out_pad, float* wpack, int wpack_state, int precision,
                                    faoctasr_stream_t stream) {
    const unsigned* const slot_a = take_scale_a();
    const float* const res = take_residual();
    if (bad_ptr(dy, w, dx)) return fail(FAOCTASR_EINVAL, "conv_transpose2d_dgrad: null pointer");
    const int OH = (IH - 1) * stride - 2 * pad + KH + out_pad, OW = (IW - 1) * stride - 2 * pad + KW + out_pad;
    IgemmGeom g;
    // dx[n][c][iy][ix] = sum_{m,t} dy[n][m][iy*s-p+kh][..] * w[c][m][t]: forward-mode gather over dy
    int rc = geom_fwd(g, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, 0, (long)M * KH * KW, (long)KH * KW);
    if (rc) return rc;
    return run_gather(dy, w, nullptr, dx, g, FAOCTASR_ACT_NONE, 0.f, wpack, wpack_state, precision, (hipStream_t)stream, nullptr, slot_a, res);
}

int faoctasr_conv_transpose2d_wgrad(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int KH,
                                    int KW, int stride, int pad, int out_pad, int accumulate, int precision, faoctasr_stream_t stream) {
    const unsigned *const slot_x = g_scale_a, *const slot_dy = g_scale_b;
    g_scale_a = g_scale_b = nullptr;
    WgradWsScope ws_scope;
    if (bad_ptr(x, dy, dw)) return fail(FAOCTASR_EINVAL, "conv_transpose2d_wgrad: null pointer");
    precision &= 0xff;
    if (precision == 3 && (!slot_x || !slot_dy)) return fail(FAOCTASR_EINVAL, "conv_transpose2d_wgrad: precision 3 (f16x2) needs both absmax slots: faoctasr_conv_set_scales");
    const int OH = (IH - 1) * stride - 2 * pad + KH + out_pad, OW = (IW - 1) * stride - 2 * pad + KW + out_pad;
    IgemmGeom g;
    // dw[c][m][t] = sum x[n][c][iy][ix] * dy[n][m][iy*s-p+kh][..]: conv-wgrad with source dy (M channels) and "dY" = x
    int rc = geom_fwd(g, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, 0, (long)M * KH * KW, (long)KH * KW);
    if (rc) return rc;
    if (IW >= 24) {
        if (!accumulate && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)M * C * KH * KW, (hipStream_t)stream) != hipSuccess)
            return fail(FAOCTASR_EHIP, "memset dw failed");
        if (precision >= 2) {      // split-precision operands on the 16-bit MFMA (stride-2 row kernel with x / dy swapped)
            rc = launch_wgrad_x3(dy, x, dw, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, 0, (long)M * KH * KW, (long)KH * KW, (hipStream_t)stream,
                                 precision == 3, slot_dy, slot_x);
            if (rc != 0) { set_route(ROUTE_WGRAD_X3); return rc < 0 ? rc : FAOCTASR_OK; }
        }
        rc = launch_wgrad_s1(dy, x, dw, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, 0, (long)M * KH * KW, (long)KH * KW,
                             (hipStream_t)stream);
        if (rc != 0) { set_route(ROUTE_WGRAD_S1); return rc < 0 ? rc : FAOCTASR_OK; }
        rc = launch_wgrad_patch(dy, x, dw, N, M, OH, OW, C, IH, IW, KH, KW, stride, pad, 0, (long)M * KH * KW, (long)KH * KW,
                                (hipStream_t)stream);
        if (rc != 0) { set_route(ROUTE_WGRAD_PATCH); return rc < 0 ? rc : FAOCTASR_OK; }
        set_route(ROUTE_WGRAD_FLAT);
        return launch_wgrad(dy, x, dw, g, (long)M * C * KH * KW, 1, (hipStream_t)stream);
    }
    set_route(ROUTE_WGRAD_FLAT);
    return launch_wgrad(dy, x, dw, g, (long)M * C * KH * KW, accumulate, (hipStream_t)stream);
}

}  // extern "C"
