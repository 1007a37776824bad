// Geometry descriptors shared by the flat (igemm.hip) and LDS-patch (igemm_patch.hip) implicit-GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace faoctasr {

constexpr int PATCH_MAX_PER_THREAD = 20;   // patch elements staged per thread (256 threads)

// flat im2col kernel: taps packed (oy+64) | (ox+64)<<8 | widx<<16
struct IgemmGeom {
    int N, C, IH, IW;      // gather source tensor
    int M, OH, OW;         // output tensor
    int SI, SO;            // input step / output step per sub-grid step
    int nphase;
    int reflect;
    int act;
    float slope;
    long wsm, wsc;         // weight strides (elements) for output channel m / gathered channel c
    int ph_py[4], ph_px[4], ph_gh[4], ph_gw[4], ph_t0[5];
    int taps[64];
};

// LDS-patch kernel: taps packed relative to the patch origin, (oy-oy0) | (ox-ox0)<<8 | widx<<16
struct PatchGeom {
    int N, C, IH, IW, M, OH, OW, SI, SO, nphase, reflect, act;
    float slope;
    int Mpad;
    long wsm, wsc;
    int py[4], px[4], gh[4], gw[4], t0[5], kc[4], oy0[4], ox0[4], span_y[4], span_x[4];
    long pack_off[5];
    int taps[64];
};

// Winograd F(2x2,3x3) kernel (igemm_wino.hip)
constexpr int WN_MT = 64;                 // output channels per block
constexpr int WN_KC = 8;                  // channels per chunk
struct WinoGeom {
    int N, C, IH, IW, M, OH, OW, act;
    float slope;
    int oy0, ox0;                  // input offset of filter tap (0, 0) relative to the output pixel
    int widx[9];                   // weight index of filter tap (i, j)
    long wsm, wsc;
    int nchunks, mtiles;
};

// bf16x3 split-precision gather kernel (igemm_bf16x3.hip)
struct SplitGeom {
    int N, C, IH, IW, M, OH, OW, SI, SO, nphase, reflect, act;
    float slope;
    int Mpad;
    long wsm, wsc;
    int py[4], px[4], gh[4], gw[4], t0[5], oy0[4], ox0[4], span_y[4], span_x[4];
    int tg[4];                     // taps per tap group
    long pack_off[5];              // bf16 element offset of each phase inside a plane
    long plane_stride;             // bf16 elements between the hi and the lo plane
    int f16;                       // 1: f16x2 planes of w * s (split16.h); the weights' absmax slot is the image's last word
    long w_elems;                  // extent of the weight tensor (its absmax is taken over all of it)
    int taps[64];                  // (oy-oy0) | (ox-ox0)<<8 | widx<<16
};
// float index of the weights' absmax slots inside a split-precision packed image: SPLIT_WPARTS partial maxima (one per block of the
// absmax pass, plain stores: nothing to zero, no atomics), whose maximum every reader takes (split_w_absmax)
constexpr int SPLIT_WPARTS = 8;
inline __host__ __device__ long split_scale_slot(const SplitGeom& g) { return g.plane_stride + 504; }
__device__ __forceinline__ unsigned split_w_absmax(const unsigned* slots) {
    unsigned m = slots[0];
#pragma unroll
    for (int i = 1; i < SPLIT_WPARTS; ++i) m = slots[i] > m ? slots[i] : m;
    return m;
}

// One weight-packing job of a batched launch (conv_pack.hip; include/faoctasr.h FAOCTASR_PACK_JOB_BYTES).  A job is what a
// wpack_state == 1 call would have launched by itself: the same geometry, the same element order.
enum PackType { PACK_PATCH = 0, PACK_WINO = 1, PACK_SPLIT = 2 };
struct PackJob {
    long block0;                   // first block of this job in the batched grid
    long total;                    // packed elements (0: nothing to pack)
    const float* w;
    float* wp;
    int type, blocks;
    union {
        PatchGeom patch;
        WinoGeom wino;
        SplitGeom split;
    } g;
};
constexpr int PACK_JOB_BYTES = 1024;
static_assert(sizeof(PackJob) <= PACK_JOB_BYTES, "PackJob outgrew its slot");
inline int pack_job_blocks(long rows) { return (int)(rows > 1024 ? 1024 : rows); }      // a block owns whole rows of the image (pack_bodies.h)

int patch_geom_from(const IgemmGeom& f, PatchGeom& g);
long patch_pack_floats(const PatchGeom& g);
int launch_pack(const float* w, float* wp, const PatchGeom& g, hipStream_t s);
int launch_patch(const float* x, const float* wp, const float* bias, float* y, PatchGeom& g, int act, float slope, hipStream_t s);
// narrow maps (phase width < 24) on the same packed image (igemm_nm.hip): 1 launched, 0 not eligible, <0 error
int launch_narrow(const float* x, const float* wp, const float* bias, float* y, PatchGeom& g, int act, float slope, hipStream_t s);
int launch_wgrad_patch(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int OH, int OW, int KH, int KW,
                       int stride, int pad, int reflect, long wsm, long wsc, hipStream_t s);
// bf16x3 split-precision weight gradient of the stride-1 3x3 layers (wgrad_x3.hip); same contract
// f16: 0 = bf16x3, 1 = f16x2 with the absmax slots of x and dy (split16.h)
int launch_wgrad_x3(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int OH, int OW, int KH, int KW,
                    int stride, int pad, int reflect, long wsm, long wsc, hipStream_t s, int f16 = 0, const unsigned* x_slot = nullptr,
                    const unsigned* dy_slot = nullptr);
// stride-1 "same" convolutions on wide maps (wgrad_s1.hip); same contract
int launch_wgrad_s1(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int OH, int OW, int KH, int KW,
                    int stride, int pad, int reflect, long wsm, long wsc, hipStream_t s);

// bf16x3 split-precision gather kernel (igemm_bf16x3.hip)
// `sink` (wpack_state == 1 only): record the packing job there instead of launching anything -- 1 recorded, 0 not eligible
// f16: 0 = bf16x3, 1 = f16x2 (x_slot: absmax slot of the gathered activation tensor, split16.h)
int split_try(const IgemmGeom& f, const float* x, const float* w, const float* bias, float* y, int act, float slope, float* wpack,
              int wpack_state, hipStream_t s, PackJob* sink = nullptr, int f16 = 0, const unsigned* x_slot = nullptr,
              const float* res = nullptr);          // 1 launched, 0 not eligible, <0 error; res: y += res fused into the epilogue
long split_pack_floats_for(const IgemmGeom& f);       // 0 when not eligible
// the activation operands' absmax slots handed over by faoctasr_conv_set_scales for this thread's next convolution-type call
extern thread_local const unsigned* g_scale_a;
extern thread_local const unsigned* g_scale_b;
// the workspace faoctasr_conv_set_workspace left for this thread's next weight-gradient call (two-pass reduction, wgrad_x3.hip)
extern thread_local float* g_wgrad_ws;
extern thread_local long g_wgrad_ws_floats;

// Winograd F(2x2,3x3) fp32 kernel for dense stride-1 3x3 gathers (igemm_wino.hip)
int wino_try(const IgemmGeom& f, const float* x, const float* w, const float* bias, float* y, int act, float slope, float* wpack,
             int wpack_state, hipStream_t s, PackJob* sink = nullptr);           // 1 launched, 0 not eligible, <0 error
long wino_pack_floats_for(const IgemmGeom& f);        // 0 when not eligible

// single-output-channel 'same' stride-1 convolution on the VALU (conv_m1.hip)
int launch_conv_m1_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int KH, int KW, int pad,
                       int act, float slope, hipStream_t stream);
int launch_conv_m1_wgrad(const float* x, const float* dy, float* dw, int N, int C, int H, int W, int KH, int KW, int pad, int accumulate,
                         hipStream_t st);

// 4x4 stride-2 stems with 1..4 input channels on the VALU (conv_stem.hip): 1 launched, 0 not this shape, <0 error
bool stem_dgrad_eligible(int C, int IH, int IW, int M, int KH, int KW, int stride, int pad);
bool stem_wgrad_eligible(int C, int IH, int IW, int KH, int KW, int stride, int pad, int reflect);
// set by faoctasr_conv_needs_scales: the split weight-gradient launchers then answer "would launch" (1 / 0) without launching
extern thread_local bool g_wgrad_dry_run;
int launch_stem_dgrad(const float* dy, const float* w, float* dx, int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                      hipStream_t s);
int launch_stem_wgrad(const float* x, const float* dy, float* dw, int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                      int reflect, int accumulate, hipStream_t s);

}  // namespace faoctasr
