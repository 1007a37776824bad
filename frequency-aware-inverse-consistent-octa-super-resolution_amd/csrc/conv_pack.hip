// Batched weight packing: every packed-weight image of a train step in ONE launch.
//
// A step packs ~240 weight images (forward + input-gradient image of every convolution, once per optimizer update).  As
// separate launches they cost ~10 us each -- 2.6 ms per step at batch 8, almost all of it launch latency of tiny dependent
// kernels.  faoctasr_conv_pack_job records, per layer, exactly the job the per-layer path would launch (same geometry, same
// element loop: pack_bodies.h); the caller keeps the job table in device memory and faoctasr_conv_pack_run packs all of them
// with one grid: block -> job by binary search over the jobs' first-block prefix, then the job's own grid-stride loop.
#include <cstring>

#include "common.h"
#include "igemm_geom.h"
#include "pack_bodies.h"

namespace faoctasr {

__global__ __launch_bounds__(256) void conv_pack_batch_kernel(const char* __restrict__ jobs, int njobs) {
    const long b = blockIdx.x;
    int lo = 0, hi = njobs - 1;                    // last job with block0 <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (reinterpret_cast<const PackJob*>(jobs + (size_t)mid * PACK_JOB_BYTES)->block0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const PackJob& J = *reinterpret_cast<const PackJob*>(jobs + (size_t)lo * PACK_JOB_BYTES);
    const long lb = b - J.block0;
    if (lb >= J.blocks) return;                    // cannot happen for a table built by faoctasr_conv_pack_job; keeps a bad table harmless
    if (J.type == PACK_PATCH) patch_pack_block(J.w, J.wp, J.g.patch, lb, J.blocks);
    else if (J.type == PACK_WINO) wino_pack_block(J.w, J.wp, J.g.wino, lb, J.blocks);
    else if (J.type == PACK_SPLIT) {
        if (J.g.split.f16) split_pack_block<true>(J.w, reinterpret_cast<unsigned short*>(J.wp), J.g.split, lb, J.blocks);
        else split_pack_block<false>(J.w, reinterpret_cast<unsigned short*>(J.wp), J.g.split, lb, J.blocks);
    }
}

// f16x2 images: the weights' absmax slots, SPLIT_WPARTS blocks per job, before the images are packed (pack_bodies.h, split16.h)
__global__ __launch_bounds__(256) void conv_pack_absmax_kernel(const char* __restrict__ jobs, int njobs) {
    __shared__ unsigned red[4];
    const PackJob& J = *reinterpret_cast<const PackJob*>(jobs + (size_t)(blockIdx.x / SPLIT_WPARTS) * PACK_JOB_BYTES);
    if (J.type == PACK_SPLIT && J.g.split.f16) split_absmax_block(J.w, J.wp, J.g.split, (int)(blockIdx.x % SPLIT_WPARTS), red);
}

}  // namespace faoctasr

using namespace faoctasr;

extern "C" {

int faoctasr_conv_pack_scales(const void* jobs_dev, int njobs, faoctasr_stream_t stream) {
    if (njobs == 0) return FAOCTASR_OK;
    if (!jobs_dev || njobs < 0) return fail(FAOCTASR_EINVAL, "conv_pack_scales: bad job table");
    hipLaunchKernelGGL(conv_pack_absmax_kernel, dim3((unsigned)njobs * SPLIT_WPARTS), dim3(256), 0, (hipStream_t)stream, (const char*)jobs_dev, njobs);
    return check_launch("conv_pack_absmax");
}

int faoctasr_conv_pack_run(const void* jobs_dev, int njobs, long nblocks, faoctasr_stream_t stream) {
    if (njobs == 0 || nblocks == 0) return FAOCTASR_OK;
    if (!jobs_dev || njobs < 0 || nblocks < 0 || nblocks > 0x7fffffffL) return fail(FAOCTASR_EINVAL, "conv_pack_run: bad job table");
    hipLaunchKernelGGL(conv_pack_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const char*)jobs_dev, njobs);
    return check_launch("conv_pack_batch");
}

}  // extern "C"
