// Per-step weight preparation in ONE launch: every data gradient of a step needs its convolution's weights re-packed
// ([Cout][tap][Cin] -> [Cin][selected taps][Cout], per stride-parity class; bf16 arm: cast on the way).  Issued per layer
// that was 72-94 launches of 8-20 us per step on the step's critical stream (0.6 ms at configs[2], 1.45 ms at configs[1]:
// profiles/r3a_*_trace_summary.txt).  Weights only change in the optimiser's update, so all of a model's classes are
// re-packed at the START of a step by one launch over a job table that lives in device memory (caller-owned, built once:
// the weights sit in a fixed arena, the destinations are persistent).  Replaces nothing in the reference -- cuDNN's
// BackwardData reads the forward filter layout; this is the price of one-launch-per-class data gradients without
// zero insertion.
#include "common.h"

namespace {

typedef __bf16 bf16_t;

// dst[ci][t][co] = src[co][tapsel[t]][ci] through a 64 x 65 LDS tile; one block = one 64 x 64 tile of one tap of one job: a wave
// reads one 256-byte row of the source per load (16 in flight per thread) and writes one 128 / 256-byte run of the destination per
// store.  (32 x 32 tiles -- 128-byte reads, 64-byte bf16 writes, four times the blocks -- took 190 us for the 512 px ResNet-50 localizer's ~300 MB of weights, this 145.)
constexpr int RT = 64;
template <typename T>
__device__ __forceinline__ void repack_tile(const loans_repack_job& j, int local, float (*tile)[RT + 1]) {
    // the tap varies fastest over the blocks: blocks that run together read the taps of the SAME weight rows -- runs that lie 4 Cin
    // bytes apart, one DRAM page -- instead of the same tap of rows that lie nine taps apart
    const int rem = local / j.ntaps;
    const int t = local - rem * j.ntaps;
    const int co0 = (rem % j.tiles_co) * RT, ci0 = (rem / j.tiles_co) * RT;
    const int tx = threadIdx.x & (RT - 1), ty = threadIdx.x / RT;
    const int st = j.tapsel[t];
    const float* src = static_cast<const float*>(j.src);
    T* dst = static_cast<T*>(j.dst);
#pragma unroll 4
    for (int k = ty; k < RT; k += 256 / RT) {
        const int co = co0 + k, ci = ci0 + tx;
        tile[k][tx] = (co < j.Cout && ci < j.Cin) ? src[((int64_t)co * j.src_taps + st) * j.Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = ty; k < RT; k += 256 / RT) {
        const int ci = ci0 + k, co = co0 + tx;
        if (ci < j.Cin && co < j.Cout) dst[((int64_t)ci * j.ntaps + t) * j.Cout + co] = (T)tile[tx][k];
    }
}

__global__ __launch_bounds__(256) void repack_batch_kernel(const loans_repack_job* jobs, int njobs) {
    __shared__ float tile[RT][RT + 1];
    // the job of this block: the last one whose first tile is <= blockIdx.x (uniform: scalar loads)
    int lo = 0, hi = njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_tile <= b) lo = mid; else hi = mid - 1;
    }
    const loans_repack_job& j = jobs[lo];
    if (j.dst_bf16) repack_tile<bf16_t>(j, b - j.first_tile, tile);
    else repack_tile<float>(j, b - j.first_tile, tile);
}

}  // namespace

extern "C" int loans_repack_dgrad_batch(const loans_repack_job* jobs_dev, int32_t njobs, int32_t total_tiles, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_tiles <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(repack_batch_kernel, dim3(total_tiles), dim3(256), 0, as_stream(stream), jobs_dev, njobs);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}
