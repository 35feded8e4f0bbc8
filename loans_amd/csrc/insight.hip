// VisualBackprop (reference insights/visual_backprop.py:16-53, used by SheepLocalizer.predict(return_visual_backprop=True),
// sheep/sheep_localizer.py:105-108) and the grayscale rois of SheepLocalizer(transform_rois_to_grayscale=True) (:65-68).
// Single-channel maps and per-pixel channel means: HBM-bound streaming kernels, far off the training path.
#include "common.h"

namespace {

// out[row] = (1 / cdiv) * sum_c act(x[row][c]), act = identity or relu(x * scale[c] + shift[c]) (the pool's input inside the
// fused stem: relu(bn1(conv1)) is never materialised).  One wave per row, 4 channels per lane and trip.
template <typename T>
__global__ __launch_bounds__(256) void channel_mean_kernel(const T* x, const float* scale, const float* shift, float* out,
                                                           int64_t rows, int C, float inv) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* p = x + row * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 v = io4<T>::ld(p + c);
        if (scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
            v = v * sc + sh;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        s += (v.x + v.y) + (v.z + v.w);
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s * inv;
}

// F.deconvolution_2d(feature, ones(1, 1, kh, kw), stride, pad, outsize = (H, W)) * avg   (visual_backprop.py:31-40):
// out[b][y][x] = avg[b][y][x] * sum over the feature cells (i, j) whose kh x kw footprint at stride s covers (y, x)
__global__ __launch_bounds__(256) void vbp_scale_kernel(const float* feat, const float* avg, float* out, int B, int fh, int fw,
                                                        int H, int W, int kh, int kw, int sy, int sx, int ph, int pw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
    // rows i with 0 <= y + ph - sy * i < kh
    const int ylo = max(0, (y + ph - kh + sy) / sy), yhi = min(fh - 1, (y + ph) / sy);
    const int xlo = max(0, (x + pw - kw + sx) / sx), xhi = min(fw - 1, (x + pw) / sx);
    float s = 0.f;
    for (int fy = ylo; fy <= yhi; ++fy)
        for (int fx = xlo; fx <= xhi; ++fx) s += feat[((int64_t)b * fh + fy) * fw + fx];
    out[i] = s * avg[i];
}

// per image: x = (x - min) / (max - min)   (visual_backprop.py:48-52)
__global__ __launch_bounds__(256) void minmax_normalize_kernel(float* x, int n) {
    __shared__ float smin[4], smax[4];
    float* p = x + (int64_t)blockIdx.x * n;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) { lo = fminf(lo, p[i]); hi = fmaxf(hi, p[i]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    lo = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    hi = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    const float inv = 1.0f / (hi - lo);
    for (int i = threadIdx.x; i < n; i += 256) p[i] = (p[i] - lo) * inv;
}

// rois NHWC4 (channels 0, 1, 2 as the reference names them b, g, r after split_axis) -> 0.299 r + 0.587 g + 0.114 b
__global__ __launch_bounds__(256) void gray_fwd_kernel(const float* rois, float* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(rois + i * 4);
    out[i] = 0.299f * v.z + 0.587f * v.y + 0.114f * v.x;
}
__global__ __launch_bounds__(256) void gray_bwd_kernel(const float* g, float* grois, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = g[i];
    *reinterpret_cast<f32x4*>(grois + i * 4) = f32x4{0.114f * v, 0.587f * v, 0.299f * v, 0.f};
}

template <typename T>
int channel_mean_impl(const void* x, const float* scale, const float* shift, float* out, int64_t rows, int32_t C, int32_t cdiv,
                      void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || (C & 3) || cdiv <= 0 || (scale == nullptr) != (shift == nullptr)) return LOANS_EINVAL;
    if ((rows + 3) / 4 >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(channel_mean_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream),
                       static_cast<const T*>(x), scale, shift, out, rows, C, 1.0f / (float)cdiv);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

}  // namespace

extern "C" int loans_channel_mean_f32(const float* x, const float* scale, const float* shift, float* out, int64_t rows,
                                      int32_t C, int32_t cdiv, void* stream) {
    return channel_mean_impl<float>(x, scale, shift, out, rows, C, cdiv, stream);
}
extern "C" int loans_channel_mean_bf16(const void* x, const float* scale, const float* shift, float* out, int64_t rows,
                                       int32_t C, int32_t cdiv, void* stream) {
    return channel_mean_impl<__bf16>(x, scale, shift, out, rows, C, cdiv, stream);
}

extern "C" int loans_vbp_scale_f32(const float* feat, const float* avg, float* out, int32_t B, int32_t fh, int32_t fw, int32_t H,
                                   int32_t W, int32_t kh, int32_t kw, int32_t sy, int32_t sx, int32_t ph, int32_t pw, void* stream) {
    if (!feat || !avg || !out || B <= 0 || fh <= 0 || fw <= 0 || H <= 0 || W <= 0) return LOANS_EINVAL;
    if (kh <= 0 || kw <= 0 || sy <= 0 || sx <= 0 || ph < 0 || pw < 0) return LOANS_EINVAL;
    const int64_t n = (int64_t)B * H * W;
    if ((n + 255) / 256 >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(vbp_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), feat, avg, out, B, fh,
                       fw, H, W, kh, kw, sy, sx, ph, pw);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_minmax_normalize_f32(float* x, int32_t B, int32_t n, void* stream) {
    if (!x || B <= 0 || n <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(minmax_normalize_kernel, dim3(B), dim3(256), 0, as_stream(stream), x, n);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_gray_fwd_f32(const float* rois_nhwc4, float* out, int64_t npix, void* stream) {
    if (!rois_nhwc4 || !out || npix <= 0) return LOANS_EINVAL;
    if ((npix + 255) / 256 >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(gray_fwd_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, as_stream(stream), rois_nhwc4, out, npix);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}
extern "C" int loans_gray_bwd_f32(const float* g, float* grois_nhwc4, int64_t npix, void* stream) {
    if (!g || !grois_nhwc4 || npix <= 0) return LOANS_EINVAL;
    if ((npix + 255) / 256 >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(gray_bwd_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, as_stream(stream), g, grois_nhwc4, npix);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}
