// HBM-bound kernels of the train step: BatchNorm2d(train)/InstanceNorm2d with fused
// activation + residual, activations, cat, Haar DWT/IDWT, frequency-split mixing, losses,
// discriminator head, AdamW.  All fp32, NCHW, float4-vectorised where the shape allows.
#include <mutex>
#include <unordered_map>
#include "common.h"
#include "split16.h"

namespace faoctasr {

thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int get_route();
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

thread_local int g_route = 0;
thread_local unsigned* g_out_absmax = nullptr;         // faoctasr_out_absmax: the slot the thread's next producer call folds max|output| into
unsigned* take_out_absmax() {
    unsigned* p = g_out_absmax;
    g_out_absmax = nullptr;
    return p;
}
thread_local int g_no_split_k = 0;         // set by run_gather for the duration of a call made with FAOCTASR_CONV_NO_SPLIT_K
void set_route(int r) { g_route = r; }
int get_route() { return g_route; }

void lds_optin(const void* kernel, size_t lds_bytes) {
    if (lds_bytes <= 64 * 1024) return;
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> granted;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[kernel];
    if (have >= lds_bytes) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    have = lds_bytes;
}

static inline int grid_for(long n, int per_block, int cap = 4096) {
    long b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// inference-mode BatchNorm: per-channel affine from the running statistics
__global__ void bn_eval_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ rmean, const float* __restrict__ rvar, float* __restrict__ y, int C, int HW,
                                   long total, float eps, int act, float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)((i / HW) % C);
        const float sc = (gamma ? gamma[c] : 1.f) / sqrtf(rvar[c] + eps);
        y[i] = act_apply((x[i] - rmean[c]) * sc + (beta ? beta[c] : 0.f), act, slope);
    }
}

__global__ void bn_eval_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ gamma,
                                   const float* __restrict__ rvar, float* __restrict__ dx, int C, int HW, long total, float eps, int act,
                                   float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)((i / HW) % C);
        float g = dy[i];
        if (act != FAOCTASR_ACT_NONE) g *= act_grad_from_out(y[i], act, slope);
        dx[i] = g * (gamma ? gamma[c] : 1.f) / sqrtf(rvar[c] + eps);
    }
}

// ------------------------------------------------------------------------------------------
// pointwise
// ------------------------------------------------------------------------------------------
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act, float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = act_apply(v.x, act, slope); v.y = act_apply(v.y, act, slope);
        v.z = act_apply(v.z, act, slope); v.w = act_apply(v.w, act, slope);
        reinterpret_cast<float4*>(y)[i] = v;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = act_apply(x[i], act, slope);
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long n, int act,
                               float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        g.x *= act_grad_from_out(v.x, act, slope); g.y *= act_grad_from_out(v.y, act, slope);
        g.z *= act_grad_from_out(v.z, act, slope); g.w *= act_grad_from_out(v.w, act, slope);
        reinterpret_cast<float4*>(dx)[i] = g;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dx[i] = dy[i] * act_grad_from_out(y[i], act, slope);
}

// y[n, 0:Ca] = act(a[n]), y[n, Ca:Ca+Cb] = act(b[n]); rows of length HW (per-image blocks contiguous)
__global__ __launch_bounds__(256) void cat2_act_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int N, long la,
                                    long lb, int act, float slope, unsigned* __restrict__ amax) {
    const long tot = (long)N * (la + lb);
    const long stride = (long)gridDim.x * blockDim.x;
    float mx = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const long n = i / (la + lb);
        const long r = i - n * (la + lb);
        const float v = r < la ? a[n * la + r] : b[n * lb + (r - la)];
        const float o = act_apply(v, act, slope);
        y[i] = o;
        mx = fmaxf(mx, fabsf(o));
    }
    if (amax) {                                       // f16x2: max|y| into the caller's absmax slot (faoctasr_out_absmax)
        __shared__ unsigned red[4];
        absmax_publish_block(__builtin_bit_cast(unsigned, mx), amax, red);
    }
}

__global__ void cat2_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ da,
                                    float* __restrict__ db, int N, long la, long lb, int act, float slope) {
    const long tot = (long)N * (la + lb);
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const long n = i / (la + lb);
        const long r = i - n * (la + lb);
        float g = dy[i];
        if (act != FAOCTASR_ACT_NONE) g *= act_grad_from_out(y[i], act, slope);
        if (r < la) {
            if (da) da[n * la + r] = g;
        } else if (db) {
            db[n * lb + (r - la)] = g;
        }
    }
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n, float alpha,
                             float beta) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = alpha * a[i] + beta * b[i];
}

__global__ void fill_kernel(float* __restrict__ p, long n, float v) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// per-channel sum over (N, HW): grid (S splits, C), block 256; partials combined with one atomic per block
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ dy, float* __restrict__ db, int N, int C, int HW,
                                                          long per) {
    __shared__ float red[4];
    const int c = blockIdx.y;
    const long L = (long)N * HW;
    long e0 = (long)blockIdx.x * per, e1 = e0 + per;
    e1 = e1 < L ? e1 : L;
    float s = 0.f;
    if ((HW & 3) == 0) {
        for (long e = e0 + 4L * threadIdx.x; e < e1; e += 1024) {
            const long n = e / HW;
            const float4 v = *reinterpret_cast<const float4*>(dy + ((long)n * C + c) * HW + (e - n * HW));
            s += v.x + v.y + v.z + v.w;
        }
    } else {
        for (long e = e0 + threadIdx.x; e < e1; e += 256) {
            const long n = e / HW;
            s += dy[((long)n * C + c) * HW + (e - n * HW)];
        }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(db + c, s);
}

// fold of the gradient of ReflectionPad2d(p): each padded position maps back to one source pixel
__global__ void reflect_pad_bwd_kernel(const float* __restrict__ dxp, float* __restrict__ dx, long NC, int H, int W, int p) {
    const long tot = NC * H * W;
    const int HP = H + 2 * p, WP = W + 2 * p;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const long nc = i / ((long)H * W);
        const int r = (int)(i - nc * H * W);
        const int yy = r / W, xx = r - yy * W;
        // source pixel (yy,xx) appears at padded rows {yy+p} plus mirrored rows when 1<=yy<=p or H-1-p<=yy<=H-2
        int ry[3], rx[3], ny = 0, nx = 0;
        ry[ny++] = yy + p;
        if (yy >= 1 && yy <= p) ry[ny++] = p - yy;
        if (yy <= H - 2 && yy >= H - 1 - p) ry[ny++] = 2 * (H - 1) - yy + p;
        rx[nx++] = xx + p;
        if (xx >= 1 && xx <= p) rx[nx++] = p - xx;
        if (xx <= W - 2 && xx >= W - 1 - p) rx[nx++] = 2 * (W - 1) - xx + p;
        float s = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) s += dxp[(nc * HP + ry[a]) * WP + rx[b]];
        dx[i] = s;
    }
}

// ------------------------------------------------------------------------------------------
// Haar DWT / IDWT.  One thread = one 2x2 input block.
// ------------------------------------------------------------------------------------------
__global__ void haar_fwd_kernel(const float* __restrict__ x, float* __restrict__ ll, float* __restrict__ hi, long NC, int H, int W) {
    const int h = H >> 1, w = W >> 1;
    const long tot = NC * h * w;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const long nc = i / ((long)h * w);
        const int r = (int)(i - nc * h * w);
        const int yy = r / w, xx = r - yy * w;
        const float* p = x + (nc * H + 2 * yy) * W + 2 * xx;
        const float2 t = *reinterpret_cast<const float2*>(p);
        const float2 u = *reinterpret_cast<const float2*>(p + W);
        const float a = t.x, b = t.y, c = u.x, d = u.y;
        if (ll) ll[i] = (a + b + c + d) * 0.5f;
        if (hi) {
            float* q = hi + (nc * 3) * h * w + r;
            q[0] = (a + b - c - d) * 0.5f;
            q[(long)h * w] = (a - b + c - d) * 0.5f;
            q[2L * h * w] = (a - b - c + d) * 0.5f;
        }
    }
}

// synthesis: x 2x2 block from (ll, lh, hl, hh); NULL inputs are zeros.  Serves AFB2D.backward and SFB2D.forward.
__global__ void haar_inv_kernel(const float* __restrict__ ll, const float* __restrict__ hi, float* __restrict__ x, long NC, int H, int W) {
    const int h = H >> 1, w = W >> 1;
    const long tot = NC * h * w;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const long nc = i / ((long)h * w);
        const int r = (int)(i - nc * h * w);
        const int yy = r / w, xx = r - yy * w;
        const float l = ll ? ll[i] : 0.f;
        float lh = 0.f, hl = 0.f, hh = 0.f;
        if (hi) {
            const float* q = hi + (nc * 3) * h * w + r;
            lh = q[0]; hl = q[(long)h * w]; hh = q[2L * h * w];
        }
        float* p = x + (nc * H + 2 * yy) * W + 2 * xx;
        *reinterpret_cast<float2*>(p) = make_float2((l + lh + hl + hh) * 0.5f, (l + lh - hl - hh) * 0.5f);
        *reinterpret_cast<float2*>(p + W) = make_float2((l - lh + hl - hh) * 0.5f, (l - lh - hl + hh) * 0.5f);
    }
}

// discriminator front ends on single-channel images: mode 0 -> y[N,1,h,w] = LL; mode 1 -> y[N,3,h,w] = (LH,HL,HH)*0.5+0.5
__global__ void haar_dfront_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int mode) {
    const int h = H >> 1, w = W >> 1;
    const long tot = (long)N * h * w;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const long n = i / ((long)h * w);
        const int r = (int)(i - n * h * w);
        const int yy = r / w, xx = r - yy * w;
        const float* p = x + (n * H + 2 * yy) * W + 2 * xx;
        const float2 t = *reinterpret_cast<const float2*>(p);
        const float2 u = *reinterpret_cast<const float2*>(p + W);
        const float a = t.x, b = t.y, c = u.x, d = u.y;
        if (mode == 0) {
            y[i] = (a + b + c + d) * 0.5f;
        } else {
            float* q = y + n * 3 * h * w + r;
            q[0] = (a + b - c - d) * 0.25f + 0.5f;
            q[(long)h * w] = (a - b + c - d) * 0.25f + 0.5f;
            q[2L * h * w] = (a - b - c + d) * 0.25f + 0.5f;
        }
    }
}

__global__ void haar_dfront_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int mode) {
    const int h = H >> 1, w = W >> 1;
    const long tot = (long)N * h * w;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += stride) {
        const long n = i / ((long)h * w);
        const int r = (int)(i - n * h * w);
        const int yy = r / w, xx = r - yy * w;
        float l = 0.f, lh = 0.f, hl = 0.f, hh = 0.f;
        if (mode == 0) {
            l = dy[i];
        } else {
            const float* q = dy + n * 3 * h * w + r;
            lh = q[0] * 0.5f; hl = q[(long)h * w] * 0.5f; hh = q[2L * h * w] * 0.5f;
        }
        float* p = dx + (n * H + 2 * yy) * W + 2 * xx;
        *reinterpret_cast<float2*>(p) = make_float2((l + lh + hl + hh) * 0.5f, (l + lh - hl - hh) * 0.5f);
        *reinterpret_cast<float2*>(p + W) = make_float2((l - lh + hl - hh) * 0.5f, (l - lh - hl + hh) * 0.5f);
    }
}

// ------------------------------------------------------------------------------------------
// frequency split mixing
// ------------------------------------------------------------------------------------------
__global__ void freq_mix_fwd_kernel(const float* __restrict__ x, const float* __restrict__ lo_hp, const float* __restrict__ lo_lp,
                                    float* __restrict__ hf, float* __restrict__ lf, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float xv = x[i];
        hf[i] = (fabsf(xv - lo_hp[i]) + xv) * 0.5f;
        lf[i] = -fabsf(lo_lp[i]);
    }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ void freq_mix_bwd_kernel(const float* __restrict__ x, const float* __restrict__ lo_hp, const float* __restrict__ lo_lp,
                                    const float* __restrict__ g_hf, const float* __restrict__ g_lf, float* __restrict__ s_hp,
                                    float* __restrict__ s_lp, float* __restrict__ dx_direct, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gh = g_hf ? g_hf[i] : 0.f, gl = g_lf ? g_lf[i] : 0.f;
        const float sh = 0.5f * gh * sgn(x[i] - lo_hp[i]);
        s_hp[i] = sh;
        s_lp[i] = -gl * sgn(lo_lp[i]);
        dx_direct[i] = 0.5f * gh + sh;
    }
}

// ------------------------------------------------------------------------------------------
// losses, discriminator head, AdamW
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float loss_term(float a, float b, int kind) {
    if (kind == 0) { const float d = a - b; return d * d; }
    if (kind == 1) return fabsf(a - b);
    // BCEWithLogits(input=a, target=b) = max(a,0) - a*b + log(1+exp(-|a|))
    return fmaxf(a, 0.f) - a * b + log1pf(expf(-fabsf(a)));
}

__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ part, long n, int kind) {
    __shared__ float red[4];
    const long stride = (long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += loss_term(a[i], b[i], kind);
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ part, int np, float* __restrict__ out, float scale) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) s += part[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}

__global__ void loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                float* __restrict__ d, long n, int kind, float scale, int wrt) {
    const float gs = g[0] * scale;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float av = a[i], bv = b[i];
        float v;
        if (kind == 0) v = 2.f * (av - bv) * (wrt == 0 ? 1.f : -1.f);
        else if (kind == 1) v = sgn(av - bv) * (wrt == 0 ? 1.f : -1.f);
        else v = wrt == 0 ? (1.f / (1.f + expf(-av)) - bv) : -av;
        d[i] = gs * v;
    }
}

__global__ __launch_bounds__(256) void mean_mix_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ out, int La, int Lb, float wa, float wb) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    float sa = 0.f, sb = 0.f;
    for (int i = threadIdx.x; i < La; i += 256) sa += a[(long)n * La + i];
    for (int i = threadIdx.x; i < Lb; i += 256) sb += b[(long)n * Lb + i];
    sa = block_sum_256(sa, red);
    sb = block_sum_256(sb, red);
    if (threadIdx.x == 0) out[n] = wa * (sa / (float)La) + wb * (sb / (float)Lb);
}

__global__ void mean_mix_bwd_kernel(const float* __restrict__ g, float* __restrict__ da, float* __restrict__ db, int La, int Lb,
                                    float wa, float wb) {
    const int n = blockIdx.x;
    const float gv = g[n];
    if (da) for (int i = threadIdx.x; i < La; i += blockDim.x) da[(long)n * La + i] = gv * wa / (float)La;
    if (db) for (int i = threadIdx.x; i < Lb; i += blockDim.x) db[(long)n * Lb + i] = gv * wb / (float)Lb;
}

// torch.optim.AdamW single-tensor arithmetic order: decay, moments, bias corrections, update
__device__ __forceinline__ void adamw_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                           float lr, float b1, float b2, float eps, float wd, float step_size, float inv_sqrt_bc2, float gscale) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gp[k] * gscale;
            float pk = pp[k] * (1.f - lr * wd);
            mp[k] = b1 * mp[k] + (1.f - b1) * gg;
            vp[k] = b2 * vp[k] + (1.f - b2) * gg * gg;
            const float denom = sqrtf(vp[k]) * inv_sqrt_bc2 + eps;
            pp[k] = pk - step_size * (mp[k] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gg = g[i] * gscale;
        float pk = p[i] * (1.f - lr * wd);
        m[i] = b1 * m[i] + (1.f - b1) * gg;
        v[i] = b2 * v[i] + (1.f - b2) * gg * gg;
        const float denom = sqrtf(v[i]) * inv_sqrt_bc2 + eps;
        p[i] = pk - step_size * (m[i] / denom);
    }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             float lr, float b1, float b2, float eps, float wd, float step_size, float inv_sqrt_bc2, float gscale) {
    adamw_body(p, g, m, v, n, lr, b1, b2, eps, wd, step_size, inv_sqrt_bc2, gscale);
}

// the same update with its scalars read from device memory, so that a captured hipGraph replays with the step's own learning
// rate and bias corrections: hyper = {lr, beta1, beta2, eps, weight_decay, lr / bc1, 1 / sqrt(bc2), grad_scale}
__global__ void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                 const float* __restrict__ hyper) {
    adamw_body(p, g, m, v, n, hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7]);
}

// ---- input pipeline (train.py:129-140): ToTensor -> RandomCrop -> [Resize bicubic] -> Normalize, fused --------------------
// torch's upsample_bicubic2d (align_corners = False, A = -0.75, border indices clamped), the arithmetic torchvision's tensor
// Resize(BICUBIC) runs: src = (dst + 0.5) * in/out - 0.5, taps at floor(src) - 1 .. + 2, rows interpolated first.
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
    constexpr float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ void prep_crop_resize_kernel(const unsigned char* __restrict__ img, const int* __restrict__ tops, const int* __restrict__ lefts,
                                        float* __restrict__ out, int N, int H, int W, int crop, int osz, float mean, float inv_std) {
    const long total = (long)N * osz * osz;
    const float scale = (float)crop / (float)osz;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % osz), oy = (int)((i / osz) % osz), n = (int)(i / ((long)osz * osz));
        const unsigned char* src = img + (long)n * H * W + (long)tops[n] * W + lefts[n];
        float v;
        if (osz == crop) {
            v = (float)src[(long)oy * W + ox] / 255.f;                 // true division: ToTensor's .div(255)
        } else {
            const float sy = ((float)oy + 0.5f) * scale - 0.5f, sx = ((float)ox + 0.5f) * scale - 0.5f;
            const float fy = floorf(sy), fx = floorf(sx);
            float cy[4], cx[4];
            cubic_coeffs(sy - fy, cy);
            cubic_coeffs(sx - fx, cx);
            const int iy = (int)fy, ix = (int)fx;
            v = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                int yy = iy - 1 + a;
                yy = yy < 0 ? 0 : (yy > crop - 1 ? crop - 1 : yy);
                float row = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int xx = ix - 1 + b;
                    xx = xx < 0 ? 0 : (xx > crop - 1 ? crop - 1 : xx);
                    row += cx[b] * ((float)src[(long)yy * W + xx] / 255.f);
                }
                v += cy[a] * row;
            }
        }
        out[i] = (v - mean) * inv_std;
    }
}

// largest |x| of a tensor as its fp32 bit pattern (non-negative floats order like unsigned integers): the "absmax slot" of the
// f16x2 contraction (split16.h).  16-byte loads, one atomicMax per block into one of the slot's 8 lines; NaN / infinity sort above
// every finite value and leave the tensor unscaled (f16x2_scale).
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ slot) {
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    unsigned mx = 0;
    const long n4 = (reinterpret_cast<unsigned long>(x) & 15) ? 0 : n >> 2;      // an operand that starts off a 16-byte boundary: scalar loads
    const u32x4v* x4 = reinterpret_cast<const u32x4v*>(x);
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    // four independent 16-byte loads in flight per thread (one load per iteration ran at 1.2 TB/s: latency-bound)
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const u32x4v v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned a = v0[j] & 0x7fffffffu, b = v1[j] & 0x7fffffffu, c = v2[j] & 0x7fffffffu, d = v3[j] & 0x7fffffffu;
            const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
            const unsigned m4 = ab > cd ? ab : cd;
            mx = m4 > mx ? m4 : mx;
        }
    }
    for (; i < n4; i += stride) {
        const u32x4v v = x4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned b = v[j] & 0x7fffffffu;
            mx = b > mx ? b : mx;
        }
    }
    for (long k = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        const unsigned b = __builtin_bit_cast(unsigned, x[k]) & 0x7fffffffu;
        mx = b > mx ? b : mx;
    }
    __shared__ unsigned red[4];
    absmax_publish_block(mx, slot, red);
}

}  // namespace faoctasr

using namespace faoctasr;

extern "C" {

int faoctasr_version(void) { return 400; }      // round 4: f16x2 (precision 3), absmax slots, two-pass weight-gradient reduction, fused residual
const char* faoctasr_last_error(void) { return err_buf(); }
int faoctasr_last_route(void) { return faoctasr::get_route(); }

int faoctasr_batchnorm_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                float* y, int N, int C, int HW, float eps, int act, float slope, faoctasr_stream_t stream) {
    if (!x || !y || !running_mean || !running_var) return fail(FAOCTASR_EINVAL, "batchnorm_eval_fwd: null pointer");
    const long total = (long)N * C * HW;
    if (total <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(bn_eval_fwd_kernel, dim3(grid_for(total, 1024)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, running_mean,
                       running_var, y, C, HW, total, eps, act, slope);
    return check_launch("bn_eval_fwd");
}

int faoctasr_batchnorm_eval_bwd(const float* dy, const float* y, const float* gamma, const float* running_var, float* dx, int N, int C, int HW,
                                float eps, int act, float slope, faoctasr_stream_t stream) {
    if (!dy || !dx || !running_var) return fail(FAOCTASR_EINVAL, "batchnorm_eval_bwd: null pointer");
    if (act != FAOCTASR_ACT_NONE && !y) return fail(FAOCTASR_EINVAL, "batchnorm_eval_bwd: activation mask needs the forward output");
    const long total = (long)N * C * HW;
    if (total <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(bn_eval_bwd_kernel, dim3(grid_for(total, 1024)), dim3(256), 0, (hipStream_t)stream, dy, y, gamma, running_var, dx, C,
                       HW, total, eps, act, slope);
    return check_launch("bn_eval_bwd");
}

int faoctasr_act_fwd(const float* x, float* y, long n, int act, float slope, faoctasr_stream_t stream) {
    if (!x || !y) return fail(FAOCTASR_EINVAL, "act_fwd: null pointer");
    if (n <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, y, n, act, slope);
    return check_launch("act_fwd");
}

int faoctasr_act_bwd(const float* dy, const float* y, float* dx, long n, int act, float slope, faoctasr_stream_t stream) {
    if (!dy || !y || !dx) return fail(FAOCTASR_EINVAL, "act_bwd: null pointer");
    if (n <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, act, slope);
    return check_launch("act_bwd");
}

int faoctasr_cat2_act_fwd(const float* a, const float* b, float* y, int N, int Ca, int Cb, int HW, int act, float slope,
                          faoctasr_stream_t stream) {
    if (!a || !b || !y) return fail(FAOCTASR_EINVAL, "cat2_act_fwd: null pointer");
    const long tot = (long)N * (Ca + Cb) * HW;
    if (tot <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(cat2_act_fwd_kernel, dim3(grid_for(tot, 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, N, (long)Ca * HW,
                       (long)Cb * HW, act, slope, faoctasr::take_out_absmax());
    return check_launch("cat2_act_fwd");
}

int faoctasr_cat2_act_bwd(const float* dy, const float* y, float* da, float* db, int N, int Ca, int Cb, int HW, int act, float slope,
                          faoctasr_stream_t stream) {
    if (!dy) return fail(FAOCTASR_EINVAL, "cat2_act_bwd: null pointer");
    if (act != FAOCTASR_ACT_NONE && !y) return fail(FAOCTASR_EINVAL, "cat2_act_bwd: mask needs y");
    const long tot = (long)N * (Ca + Cb) * HW;
    if (tot <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(cat2_act_bwd_kernel, dim3(grid_for(tot, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, da, db, N,
                       (long)Ca * HW, (long)Cb * HW, act, slope);
    return check_launch("cat2_act_bwd");
}

int faoctasr_axpby(const float* a, const float* b, float* y, long n, float alpha, float beta, faoctasr_stream_t stream) {
    if (!a || !b || !y) return fail(FAOCTASR_EINVAL, "axpby: null pointer");
    if (n <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, alpha, beta);
    return check_launch("axpby");
}

int faoctasr_fill(float* p, long n, float value, faoctasr_stream_t stream) {
    if (!p) return fail(FAOCTASR_EINVAL, "fill: null pointer");
    if (n <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, n, value);
    return check_launch("fill");
}

int faoctasr_channel_sum(const float* dy, float* db, int N, int C, int HW, int accumulate, faoctasr_stream_t stream) {
    if (!dy || !db) return fail(FAOCTASR_EINVAL, "channel_sum: null pointer");
    if (C <= 0) return FAOCTASR_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate && hipMemsetAsync(db, 0, sizeof(float) * C, st) != hipSuccess) return fail(FAOCTASR_EHIP, "channel_sum: memset failed");
    const long L = (long)N * HW;
    long S = (1024 + C - 1) / C;                      // ~1024 blocks in total
    const long maxs = (L + 4095) / 4096;
    S = S < maxs ? S : maxs;
    S = S < 1 ? 1 : S;
    long per = (L + S - 1) / S;
    per = (per + 3) & ~3L;
    S = (L + per - 1) / per;
    hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)S, C), dim3(256), 0, st, dy, db, N, C, HW, per);
    return check_launch("channel_sum");
}

int faoctasr_reflect_pad_bwd(const float* dxp, float* dx, int NC, int H, int W, int p, faoctasr_stream_t stream) {
    if (!dxp || !dx) return fail(FAOCTASR_EINVAL, "reflect_pad_bwd: null pointer");
    if (p >= H || p >= W) return fail(FAOCTASR_EINVAL, "reflect_pad_bwd: pad >= size");
    const long tot = (long)NC * H * W;
    if (tot <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3(grid_for(tot, 256)), dim3(256), 0, (hipStream_t)stream, dxp, dx, (long)NC, H, W, p);
    return check_launch("reflect_pad_bwd");
}

int faoctasr_haar_dwt2d_fwd(const float* x, float* ll, float* hi, long NC, int H, int W, faoctasr_stream_t stream) {
    if (!x) return fail(FAOCTASR_EINVAL, "haar_dwt2d_fwd: null pointer");
    if ((H & 1) || (W & 1)) return fail(FAOCTASR_EUNSUPPORTED, "haar_dwt2d: odd size %dx%d (reference path uses even sizes only)", H, W);
    const long tot = NC * (H / 2) * (W / 2);
    if (tot <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(haar_fwd_kernel, dim3(grid_for(tot, 256)), dim3(256), 0, (hipStream_t)stream, x, ll, hi, NC, H, W);
    return check_launch("haar_fwd");
}

int faoctasr_haar_dwt2d_bwd(const float* dll, const float* dhi, float* dx, long NC, int H, int W, faoctasr_stream_t stream) {
    if (!dx) return fail(FAOCTASR_EINVAL, "haar_dwt2d_bwd: null pointer");
    if ((H & 1) || (W & 1)) return fail(FAOCTASR_EUNSUPPORTED, "haar_dwt2d: odd size %dx%d", H, W);
    const long tot = NC * (H / 2) * (W / 2);
    if (tot <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(haar_inv_kernel, dim3(grid_for(tot, 256)), dim3(256), 0, (hipStream_t)stream, dll, dhi, dx, NC, H, W);
    return check_launch("haar_inv");
}

int faoctasr_haar_dfront_fwd(const float* x, float* y, int N, int H, int W, int mode, faoctasr_stream_t stream) {
    if (!x || !y) return fail(FAOCTASR_EINVAL, "haar_dfront_fwd: null pointer");
    if ((H & 1) || (W & 1)) return fail(FAOCTASR_EUNSUPPORTED, "haar_dfront: odd size");
    const long tot = (long)N * (H / 2) * (W / 2);
    if (tot <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(haar_dfront_fwd_kernel, dim3(grid_for(tot, 256)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, mode);
    return check_launch("haar_dfront_fwd");
}

int faoctasr_haar_dfront_bwd(const float* dy, float* dx, int N, int H, int W, int mode, faoctasr_stream_t stream) {
    if (!dy || !dx) return fail(FAOCTASR_EINVAL, "haar_dfront_bwd: null pointer");
    const long tot = (long)N * (H / 2) * (W / 2);
    if (tot <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(haar_dfront_bwd_kernel, dim3(grid_for(tot, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, H, W, mode);
    return check_launch("haar_dfront_bwd");
}

int faoctasr_freq_mix_fwd(const float* x, const float* low_hp, const float* low_lp, float* hf, float* lf, long n,
                          faoctasr_stream_t stream) {
    if (!x || !low_hp || !low_lp || !hf || !lf) return fail(FAOCTASR_EINVAL, "freq_mix_fwd: null pointer");
    if (n <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(freq_mix_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, low_hp, low_lp, hf, lf, n);
    return check_launch("freq_mix_fwd");
}

int faoctasr_freq_mix_bwd(const float* x, const float* low_hp, const float* low_lp, const float* g_hf, const float* g_lf, float* s_hp,
                          float* s_lp, float* dx_direct, long n, faoctasr_stream_t stream) {
    if (!x || !low_hp || !low_lp || !s_hp || !s_lp || !dx_direct) return fail(FAOCTASR_EINVAL, "freq_mix_bwd: null pointer");
    if (n <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(freq_mix_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, low_hp, low_lp, g_hf, g_lf,
                       s_hp, s_lp, dx_direct, n);
    return check_launch("freq_mix_bwd");
}

long faoctasr_loss_workspace_floats(void) { return 1024; }

int faoctasr_loss_fwd(const float* a, const float* b, float* out, long n, int kind, float scale, float* workspace,
                      faoctasr_stream_t stream) {
    if (!a || !b || !out || !workspace) return fail(FAOCTASR_EINVAL, "loss_fwd: null pointer");
    if (kind < 0 || kind > 2) return fail(FAOCTASR_EINVAL, "loss_fwd: unknown kind %d", kind);
    const int nb = n > 0 ? grid_for(n, 2048, 1024) : 0;
    if (nb) hipLaunchKernelGGL(loss_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, workspace, n, kind);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, nb, out, scale);
    return check_launch("loss_fwd");
}

int faoctasr_loss_bwd(const float* a, const float* b, const float* g, float* d, long n, int kind, float scale, int wrt,
                      faoctasr_stream_t stream) {
    if (!a || !b || !g || !d) return fail(FAOCTASR_EINVAL, "loss_bwd: null pointer");
    if (kind < 0 || kind > 2) return fail(FAOCTASR_EINVAL, "loss_bwd: unknown kind %d", kind);
    if (n <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, g, d, n, kind, scale, wrt);
    return check_launch("loss_bwd");
}

int faoctasr_mean_mix_fwd(const float* a, const float* b, float* out, int N, int La, int Lb, float wa, float wb,
                          faoctasr_stream_t stream) {
    if (!a || !b || !out) return fail(FAOCTASR_EINVAL, "mean_mix_fwd: null pointer");
    if (N <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(mean_mix_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, a, b, out, La, Lb, wa, wb);
    return check_launch("mean_mix_fwd");
}

int faoctasr_mean_mix_bwd(const float* g, float* da, float* db, int N, int La, int Lb, float wa, float wb, faoctasr_stream_t stream) {
    if (!g) return fail(FAOCTASR_EINVAL, "mean_mix_bwd: null pointer");
    if (N <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(mean_mix_bwd_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, g, da, db, La, Lb, wa, wb);
    return check_launch("mean_mix_bwd");
}

int faoctasr_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, faoctasr_stream_t stream) {
    if (!p || !g || !m || !v) return fail(FAOCTASR_EINVAL, "adamw_step: null pointer");
    if (step < 1) return fail(FAOCTASR_EINVAL, "adamw_step: step must be >= 1");
    if (n <= 0) return FAOCTASR_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                       eps, weight_decay, step_size, inv_sqrt_bc2, grad_scale);
    return check_launch("adamw");
}

int faoctasr_prep_crop_resize(const unsigned char* img, const int* tops, const int* lefts, float* out, int N, int H, int W, int crop,
                              int out_size, float mean, float std, faoctasr_stream_t stream) {
    if (!img || !tops || !lefts || !out) return fail(FAOCTASR_EINVAL, "prep_crop_resize: null pointer");
    if (N < 0 || crop <= 0 || out_size <= 0 || crop > H || crop > W || std == 0.f) return fail(FAOCTASR_EINVAL, "prep_crop_resize: bad shape");
    const long total = (long)N * out_size * out_size;
    if (total == 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(prep_crop_resize_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, img, tops, lefts, out, N, H,
                       W, crop, out_size, mean, 1.f / std);
    return check_launch("prep_crop_resize");
}

int faoctasr_out_absmax(unsigned* slot) {
    faoctasr::g_out_absmax = slot;
    return FAOCTASR_OK;
}

int faoctasr_absmax_bits(const float* x, long n, unsigned* slot, faoctasr_stream_t stream) {
    if (!x || !slot) return fail(FAOCTASR_EINVAL, "absmax_bits: null pointer");
    if (n <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(absmax_bits_kernel, dim3(grid_for(n, 8192, 2048)), dim3(256), 0, (hipStream_t)stream, x, n, slot);
    return check_launch("absmax_bits");
}

int faoctasr_adamw_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, faoctasr_stream_t stream) {
    if (!p || !g || !m || !v || !hyper) return fail(FAOCTASR_EINVAL, "adamw_step_dev: null pointer");
    if (n <= 0) return FAOCTASR_OK;
    hipLaunchKernelGGL(adamw_dev_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper);
    return check_launch("adamw_dev");
}

}  // extern "C"
