// GPU side of the input contract (reference common/datasets/image_dataset.py:16-28,98): the LANCZOS resize that
// `resize_image` delegates to Pillow, and the `image / 255` float conversion of `get_example`.
//
// Pillow's 8-bit resampler (libImaging/Resample.c, ImagingResampleHorizontal_8bpc / Vertical_8bpc) is integer
// arithmetic: per output coordinate a window [xmin, xmin + n) of the input and n coefficients in 22-bit fixed point
// (PRECISION_BITS = 32 - 8 - 2); acc = 2^21 + sum(pixel * k); out = clip8(acc >> 22); horizontal pass first, each pass
// rounds to uint8.  The coefficient tables come from the host (loans_amd/common/datasets/resample.py, the same double
// arithmetic as precompute_coeffs + normalize_coeffs_8bpc), so the kernels are bit-exact against Pillow.
// Bound: HBM/L2 streaming; one thread per output pixel (3 channels), windows are at most a few dozen taps.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// src [B][H][inW][3] -> dst [B][H][outW][3]
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* src, uint8_t* dst, const int32_t* bounds,
                                                         const int32_t* kk, int ks, int64_t rows, int inW, int outW) {
    const int64_t total = rows * outW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % outW);
        const int64_t row = i / outW;
        const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
        const int32_t* k = kk + (int64_t)xx * ks;
        const uint8_t* p = src + (row * inW + xmin) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < n; ++x) {
            const int w = k[x];
            s0 += (int)p[3 * x] * w;
            s1 += (int)p[3 * x + 1] * w;
            s2 += (int)p[3 * x + 2] * w;
        }
        uint8_t* o = dst + i * 3;
        o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
    }
}

// src [B][inH][W][3] -> F32 ? dst_f [B][3][outH][W] = u8 / 255 : dst_u [B][outH][W][3]
template <bool F32>
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* src, uint8_t* dst_u, float* dst_f,
                                                         const int32_t* bounds, const int32_t* kk, int ks, int B,
                                                         int inH, int outH, int W) {
    const int64_t total = (int64_t)B * outH * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int64_t t = i / W;
        const int yy = (int)(t % outH);
        const int64_t b = t / outH;
        const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
        const int32_t* k = kk + (int64_t)yy * ks;
        const uint8_t* p = src + ((b * inH + ymin) * W + x) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < n; ++y) {
            const int w = k[y];
            const uint8_t* q = p + (int64_t)y * W * 3;
            s0 += (int)q[0] * w;
            s1 += (int)q[1] * w;
            s2 += (int)q[2] * w;
        }
        if (F32) {      // image / 255 in float32 (IEEE division, what numpy does for a float32 array)
            const int64_t plane = (int64_t)outH * W;
            float* o = dst_f + b * 3 * plane + (int64_t)yy * W + x;
            o[0] = (float)clip8(s0) / 255.f;
            o[plane] = (float)clip8(s1) / 255.f;
            o[2 * plane] = (float)clip8(s2) / 255.f;
        } else {
            uint8_t* o = dst_u + i * 3;
            o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
        }
    }
}

__global__ __launch_bounds__(256) void u8hwc3_to_f32chw_kernel(const uint8_t* src, float* dst, int B, int64_t HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        const uint8_t* s = src + i * 3;
        float* o = dst + b * 3 * HW + p;
        o[0] = (float)s[0] / 255.f;
        o[HW] = (float)s[1] / 255.f;
        o[2 * HW] = (float)s[2] / 255.f;
    }
}

// ---- a batch of frames of DIFFERENT sizes in one launch pair (the reference's naive augmentation branch, common/datasets/
// image_dataset.py:86-90, crops every frame to a size of its own before `resize_image`).  One job per frame: where it lies in
// the packed source / intermediate buffers and where ITS coefficient tables lie in `tables` (int32 words: bounds [out][2],
// then coefficients [out][ks], per axis).  blockIdx.y = the frame, blockIdx.x strides over its output pixels; frame j writes
// slot j of dst, so the batch comes out in input order without a gather.
__global__ __launch_bounds__(256) void resample_h_ragged_kernel(const uint8_t* src, uint8_t* tmp, const loans_resample_job* jobs,
                                                                const int32_t* tables, int outW) {
    const loans_resample_job j = jobs[blockIdx.y];
    const int32_t* bounds = tables + j.hb_off;
    const int32_t* kk = tables + j.hk_off;
    const uint8_t* s = src + j.src_off;
    uint8_t* d = tmp + j.tmp_off;
    const int total = j.inH * outW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int xx = i % outW, row = i / outW;
        const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
        const int32_t* k = kk + (int64_t)xx * j.hks;
        // j.flip: the frame is the horizontal mirror of what lies in memory (random_flip of the naive branch,
        // image_dataset.py:40-44,90): pixel x of the frame is pixel inW - 1 - x of the buffer
        const uint8_t* p = s + ((int64_t)row * j.inW + (j.flip ? j.inW - 1 - xmin : xmin)) * 3;
        const int step = j.flip ? -3 : 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < n; ++x) {
            const int w = k[x];
            s0 += (int)p[step * x] * w;
            s1 += (int)p[step * x + 1] * w;
            s2 += (int)p[step * x + 2] * w;
        }
        uint8_t* o = d + (int64_t)i * 3;
        o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
    }
}

__global__ __launch_bounds__(256) void resample_v_ragged_kernel(const uint8_t* tmp, float* dst, const loans_resample_job* jobs,
                                                                const int32_t* tables, int outH, int outW) {
    const loans_resample_job j = jobs[blockIdx.y];
    const int32_t* bounds = tables + j.vb_off;
    const int32_t* kk = tables + j.vk_off;
    const uint8_t* s = tmp + j.tmp_off;
    const int64_t plane = (int64_t)outH * outW;
    float* out = dst + (int64_t)blockIdx.y * 3 * plane;
    const int total = outH * outW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int x = i % outW, yy = i / outW;
        const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
        const int32_t* k = kk + (int64_t)yy * j.vks;
        const uint8_t* p = s + ((int64_t)ymin * outW + x) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < n; ++y) {
            const int w = k[y];
            const uint8_t* q = p + (int64_t)y * outW * 3;
            s0 += (int)q[0] * w;
            s1 += (int)q[1] * w;
            s2 += (int)q[2] * w;
        }
        float* o = out + i;
        o[0] = (float)clip8(s0) / 255.f;
        o[plane] = (float)clip8(s1) / 255.f;
        o[2 * plane] = (float)clip8(s2) / 255.f;
    }
}

bool table_ok(const int32_t* b, const int32_t* k, int ks) { return b && k && ks > 0; }

}  // namespace

extern "C" int loans_resize_lanczos_u8_f32(const uint8_t* src, uint8_t* tmp, float* dst, int32_t B, int32_t inH,
                                           int32_t inW, int32_t outH, int32_t outW, const int32_t* hbounds,
                                           const int32_t* hk, int32_t hks, const int32_t* vbounds, const int32_t* vk,
                                           int32_t vks, void* stream) {
    if (!src || !tmp || !dst || B <= 0 || inH <= 0 || inW <= 0 || outH <= 0 || outW <= 0) return LOANS_EINVAL;
    if (!table_ok(hbounds, hk, hks) || !table_ok(vbounds, vk, vks)) return LOANS_EINVAL;
    if ((int64_t)B * inH * (int64_t)(inW > outW ? inW : outW) * 3 >= ((int64_t)1 << 40)) return LOANS_ERANGE;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((int64_t)B * inH * outW, 256)), dim3(256), 0, st, src, tmp, hbounds,
                       hk, hks, (int64_t)B * inH, inW, outW);
    LOANS_LAUNCH_CHECK();
    hipLaunchKernelGGL(resample_v_kernel<true>, dim3(grid_for((int64_t)B * outH * outW, 256)), dim3(256), 0, st, tmp,
                       (uint8_t*)nullptr, dst, vbounds, vk, vks, B, inH, outH, outW);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_resize_lanczos_u8(const uint8_t* src, uint8_t* tmp, uint8_t* dst, int32_t B, int32_t inH, int32_t inW,
                                       int32_t outH, int32_t outW, const int32_t* hbounds, const int32_t* hk,
                                       int32_t hks, const int32_t* vbounds, const int32_t* vk, int32_t vks, void* stream) {
    if (!src || !tmp || !dst || B <= 0 || inH <= 0 || inW <= 0 || outH <= 0 || outW <= 0) return LOANS_EINVAL;
    if (!table_ok(hbounds, hk, hks) || !table_ok(vbounds, vk, vks)) return LOANS_EINVAL;
    if ((int64_t)B * inH * (int64_t)(inW > outW ? inW : outW) * 3 >= ((int64_t)1 << 40)) return LOANS_ERANGE;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((int64_t)B * inH * outW, 256)), dim3(256), 0, st, src, tmp, hbounds,
                       hk, hks, (int64_t)B * inH, inW, outW);
    LOANS_LAUNCH_CHECK();
    hipLaunchKernelGGL(resample_v_kernel<false>, dim3(grid_for((int64_t)B * outH * outW, 256)), dim3(256), 0, st, tmp, dst,
                       (float*)nullptr, vbounds, vk, vks, B, inH, outH, outW);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_u8hwc3_to_f32chw(const uint8_t* src, float* dst, int32_t B, int32_t H, int32_t W, void* stream) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(u8hwc3_to_f32chw_kernel, dim3(grid_for((int64_t)B * H * W, 256)), dim3(256), 0, as_stream(stream), src,
                       dst, B, (int64_t)H * W);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_resize_ragged_u8_f32(const uint8_t* src, uint8_t* tmp, float* dst, const loans_resample_job* jobs,
                                          int32_t njobs, const int32_t* tables, int32_t max_inH, int32_t outH, int32_t outW,
                                          void* stream) {
    if (!src || !tmp || !dst || !jobs || !tables || njobs <= 0 || njobs > 65535 || max_inH <= 0 || outH <= 0 || outW <= 0)
        return LOANS_EINVAL;
    if ((int64_t)max_inH * outW >= ((int64_t)1 << 30)) return LOANS_ERANGE;
    hipStream_t st = as_stream(stream);
    const int bh = (int)(((int64_t)max_inH * outW + 255) / 256), bv = (outH * outW + 255) / 256;
    hipLaunchKernelGGL(resample_h_ragged_kernel, dim3(bh < 64 ? bh : 64, njobs), dim3(256), 0, st, src, tmp, jobs, tables, outW);
    LOANS_LAUNCH_CHECK();
    hipLaunchKernelGGL(resample_v_ragged_kernel, dim3(bv < 64 ? bv : 64, njobs), dim3(256), 0, st, tmp, dst, jobs, tables, outH, outW);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}
