// Small kernels around the conv stacks: preprocessing, GAP, Linear heads, rotation-dropout
// multiply, spatial transformer (grid + bilinear sampler), losses and the fused Adam-AMSGrad.
// None of them is GEMM-shaped; they are coalesced streaming / wave-shuffle reduction kernels.
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 ldnt4(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ void stnt4(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }

// block-wide sum for 256-thread blocks; result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 4 floats */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- preprocessing ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_kernel(const float* img, float* out, int B, int HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        const float* src = img + b * 3 * HW + p;
        // f32 multiply, truncate toward zero to uint8 (numpy astype(uint8)), back to f32
        const float r = (float)(((int)(src[0] * 255.f)) & 255);
        const float g = (float)(((int)(src[HW] * 255.f)) & 255);
        const float bl = (float)(((int)(src[2 * (int64_t)HW] * 255.f)) & 255);
        f32x4 v = {bl - 103.063f, g - 115.903f, r - 123.152f, 0.f};
        st4(out + i * 4, v);
    }
}

// packed 3-channel rows inside a zero border: one thread per (b, yp, xp) of the padded buffer
template <typename TO>
__global__ __launch_bounds__(256) void prep_dense_kernel(const float* img, TO* out, int B, int H, int W, int pad, int Hp, int Wp) {
    const int64_t total = (int64_t)B * Hp * Wp;
    const int64_t HW = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp);
        const int64_t t = i / Wp;
        const int yp = (int)(t % Hp);
        const int64_t b = t / Hp;
        const int y = yp - pad, x = xp - pad;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const float* src = img + b * 3 * HW + (int64_t)y * W + x;
            const float r = (float)(((int)(src[0] * 255.f)) & 255);
            const float g = (float)(((int)(src[HW] * 255.f)) & 255);
            const float bl = (float)(((int)(src[2 * HW] * 255.f)) & 255);
            v0 = bl - 103.063f; v1 = g - 115.903f; v2 = r - 123.152f;
        }
        TO* o = out + i * 3;
        o[0] = (TO)v0; o[1] = (TO)v1; o[2] = (TO)v2;         // bf16: round to nearest even, as the bf16 arm's staging does
    }
}

__global__ __launch_bounds__(256) void nchw3_to_nhwc4_kernel(const float* in, float* out, int B, int HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        const float* src = in + b * 3 * HW + p;
        f32x4 v = {src[0], src[HW], src[2 * (int64_t)HW], 0.f};
        st4(out + i * 4, v);
    }
}

// ---- global average pooling ----------------------------------------------------------------
// block = 64 channel groups x 4 pixel phases of one image: thread (tx, part) sums pixels part, part + 4, ... with four loads in
// flight, the phases fold through LDS.  (One thread per (image, channel group) walking all HW pixels one dependent load after the
// other took 76 us for the ResNet-50 localizer's 64 x 16 x 16 x 2048 map -- 67 MB, 11 us of HBM time -- on the forward's critical path.)
template <typename T>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const T* x, float* y, int B, int HW, int C4) {
    __shared__ f32x4 red[4][64];
    const int tx = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int chunks = (C4 + 63) >> 6;
    const int b = blockIdx.x / chunks, c = (blockIdx.x - b * chunks) * 64 + tx;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 s0 = z, s1 = z, s2 = z, s3 = z;
    if (c < C4) {
        const T* base = x + ((int64_t)b * HW * C4 + c) * 4;
        const int64_t step = (int64_t)C4 * 4;
        int p = part;
        for (; p + 12 < HW; p += 16) {
            s0 += io4<T>::ld(base + p * step); s1 += io4<T>::ld(base + (p + 4) * step);
            s2 += io4<T>::ld(base + (p + 8) * step); s3 += io4<T>::ld(base + (p + 12) * step);
        }
        for (; p < HW; p += 4) s0 += io4<T>::ld(base + p * step);
    }
    red[part][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && c < C4)
        st4(y + ((int64_t)b * C4 + c) * 4, ((red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx])) * (1.f / (float)HW));
}

template <typename T>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* gy, T* gx, int B, int HW, int C4) {
    const int64_t total = (int64_t)B * HW * C4;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i % C4, b = i / ((int64_t)HW * C4);
        io4<T>::st(gx + i * 4, ld4(gy + (b * C4 + c) * 4) * inv);
    }
}

// ---- Linear --------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_chainer(float x) { return tanhf(x * 0.5f) * 0.5f + 0.5f; }

// one block per sample; N small (<= 8)
template <typename T>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const T* x, const float* W, const float* b, float* y,
                                                         int K, int N, int act_in, int act_out) {
    __shared__ float sh[4];
    const int s = blockIdx.x;
    const T* xr = x + (int64_t)s * K;
    for (int n = 0; n < N; ++n) {
        const float* wr = W + (int64_t)n * K;
        float acc = 0.f;
        auto fma4 = [&](f32x4 xv, f32x4 wv) {
            if (act_in) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
            acc += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        };
        // one block per sample walks K = 41 472 (the assessor's head) in 40 trips: eight trips' loads go out together, or every
        // trip waits for its own two (77 us at B = 256 for 42 MB: a latency chain, not a stream)
        constexpr int LU = 8;
        int k = threadIdx.x * 4;
        for (; k + (LU - 1) * 1024 < K; k += LU * 1024) {
            f32x4 xv[LU], wv[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) { xv[u] = io4<T>::ld(xr + k + u * 1024); wv[u] = ld4(wr + k + u * 1024); }
#pragma unroll
            for (int u = 0; u < LU; ++u) fma4(xv[u], wv[u]);
        }
        for (; k < K; k += 1024) fma4(io4<T>::ld(xr + k), ld4(wr + k));
        acc = block_sum(acc, sh);
        if (threadIdx.x == 0) {
            float v = acc + (b ? b[n] : 0.f);
            if (act_out) v = sigmoidf_chainer(v);
            y[(int64_t)s * N + n] = v;
        }
    }
}

__device__ __forceinline__ float gz_of(const float* y, const float* gy, int64_t i, int act_out) {
    const float g = gy[i];
    if (!act_out) return g;
    const float yy = y[i];
    return g * yy * (1.f - yy);
}

// gx[s][k] = sum_n gz[s][n] W[n][k]  (* (x > 0) when act_in)
template <typename T>
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const T* x, const float* W, const float* y,
                                                           const float* gy, T* gx, int B, int K4, int N,
                                                           int act_in, int act_out) {
    // grid = (column blocks, samples): no 64-bit division per element, the sample's gz values are uniform per block
    const int64_t s = blockIdx.y;
    for (int kq = blockIdx.x * blockDim.x + threadIdx.x; kq < K4; kq += gridDim.x * blockDim.x) {
        const int64_t i = s * K4 + kq;
        f32x4 xv = {1.f, 1.f, 1.f, 1.f};
        if (act_in) xv = io4<T>::ld(x + i * 4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int n = 0; n < N; ++n) acc += ld4(W + ((int64_t)n * K4 + kq) * 4) * gz_of(y, gy, s * N + n, act_out);
        if (act_in) {
            acc.x = xv.x > 0.f ? acc.x : 0.f; acc.y = xv.y > 0.f ? acc.y : 0.f;
            acc.z = xv.z > 0.f ? acc.z : 0.f; acc.w = xv.w > 0.f ? acc.w : 0.f;
        }
        io4<T>::st(gx + i * 4, acc);
    }
}

// gW[n][k] += sum_{s in slab} gz[s][n] act(x[s][k]);  grid = (k-blocks, sample slabs)
template <int NMAX, typename T>
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const T* x, const float* y, const float* gy, float* gW,
                                                           int B, int K4, int N, int act_in, int act_out, int slab) {
    const int kq = blockIdx.x * blockDim.x + threadIdx.x;
    if (kq >= K4) return;
    const int s0 = blockIdx.y * slab;
    const int s1 = min(s0 + slab, B);
    f32x4 acc[NMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = s0; s < s1; ++s) {
        f32x4 xv = io4<T>::ld(x + ((int64_t)s * K4 + kq) * 4);
        if (act_in) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
            if (n < N) acc[n] += xv * gz_of(y, gy, (int64_t)s * N + n, act_out);
    }
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
        if (n < N) {
            float* dst = gW + ((int64_t)n * K4 + kq) * 4;
            atomic_add_f32(dst + 0, acc[n].x); atomic_add_f32(dst + 1, acc[n].y);
            atomic_add_f32(dst + 2, acc[n].z); atomic_add_f32(dst + 3, acc[n].w);
        }
}

__global__ __launch_bounds__(256) void linear_bwd_b_kernel(const float* y, const float* gy, float* gb, int B, int N,
                                                           int act_out) {
    __shared__ float sh[4];
    for (int n = 0; n < N; ++n) {
        float acc = 0.f;
        for (int s = threadIdx.x; s < B; s += 256) acc += gz_of(y, gy, (int64_t)s * N + n, act_out);
        acc = block_sum(acc, sh);
        if (threadIdx.x == 0) gb[n] += acc;
    }
}

// ---- elementwise ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mul_kernel(const float* x, const float* m, float* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = x[i] * m[i];
}
__global__ __launch_bounds__(256) void axpby_kernel(float a, const float* x, float b, float* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (b == 0.f) ? a * x[i] : a * x[i] + b * y[i];
}

// ---- spatial transformer -------------------------------------------------------------------
__device__ __forceinline__ float lin_coord(int i, int n) {
    // numpy.linspace(-1, 1, n, dtype=float32): float64 arithmetic, last point exactly 1
    if (n == 1) return -1.f;
    if (i == n - 1) return 1.f;
    return (float)(-1.0 + (double)i * (2.0 / (double)(n - 1)));
}

__global__ __launch_bounds__(256) void st_grid_fwd_kernel(const float* theta, float* grid, int B, int th, int tw) {
    const int64_t total = (int64_t)B * th * tw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / (th * tw);
        const int p = (int)(i - b * th * tw);
        const int yy = p / tw, xx = p - yy * tw;
        const float xs = lin_coord(xx, tw), ys = lin_coord(yy, th);
        const float* t = theta + b * 6;
        grid[(b * 2 + 0) * th * tw + p] = t[0] * xs + t[1] * ys + t[2];
        grid[(b * 2 + 1) * th * tw + p] = t[3] * xs + t[4] * ys + t[5];
    }
}

// one block per sample: gtheta[b][r][c] = sum_p ggrid[b][r][p] * coords[c][p]
__global__ __launch_bounds__(256) void st_grid_bwd_kernel(const float* ggrid, float* gtheta, int th, int tw) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const int P = th * tw;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = threadIdx.x; p < P; p += 256) {
        const int yy = p / tw, xx = p - yy * tw;
        const float xs = lin_coord(xx, tw), ys = lin_coord(yy, th);
        const float gu = ggrid[((int64_t)b * 2 + 0) * P + p], gv = ggrid[((int64_t)b * 2 + 1) * P + p];
        acc[0] += gu * xs; acc[1] += gu * ys; acc[2] += gu;
        acc[3] += gv * xs; acc[4] += gv * ys; acc[5] += gv;
    }
    for (int k = 0; k < 6; ++k) {
        const float s = block_sum(acc[k], sh);
        if (threadIdx.x == 0) gtheta[b * 6 + k] = s;
    }
}

struct Bilin {
    int u0, v0;              // padded-image indices of the top-left neighbour
    double wu0, wu1, wv0, wv1;
    float u, v;              // unclipped padded coordinates
};

__device__ __forceinline__ Bilin bilin_setup(float gx, float gy, int H, int W) {
    Bilin s;
    // chainer spatial_transformer_sampler: (u + 1) * (W - 1) / 2 + 1 in float32
    s.u = __fadd_rn(__fdiv_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)(W - 1)), 2.f), 1.f);
    s.v = __fadd_rn(__fdiv_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)(H - 1)), 2.f), 1.f);
    const float uc = fminf(fmaxf(s.u, 0.f), (float)(W + 1));
    const float vc = fminf(fmaxf(s.v, 0.f), (float)(H + 1));
    s.u0 = min(max((int)floorf(uc), 0), W);
    s.v0 = min(max((int)floorf(vc), 0), H);
    // int32 - float32 promotes to float64 in NumPy
    s.wu0 = (double)uc - (double)s.u0;
    s.wu1 = (double)(s.u0 + 1) - (double)uc;
    s.wv0 = (double)vc - (double)s.v0;
    s.wv1 = (double)(s.v0 + 1) - (double)vc;
    return s;
}

__device__ __forceinline__ float pad_fetch(const float* plane, int vp, int up, int H, int W) {
    const int y = vp - 1, x = up - 1;     // one-pixel zero border
    return ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? plane[(int64_t)y * W + x] : 0.f;
}

__global__ __launch_bounds__(256) void st_sampler_fwd_kernel(const float* img, const float* grid, float* rois, int B,
                                                             int H, int W, int P) {
    const int64_t total = (int64_t)B * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / P;
        const int p = (int)(i - b * P);
        const Bilin s = bilin_setup(grid[(b * 2 + 0) * P + p], grid[(b * 2 + 1) * P + p], H, W);
        const float w1 = (float)(s.wu1 * s.wv1), w2 = (float)(s.wu0 * s.wv1);
        const float w3 = (float)(s.wu1 * s.wv0), w4 = (float)(s.wu0 * s.wv0);
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* plane = img + (b * 3 + c) * (int64_t)H * W;
            float y = __fmul_rn(w1, pad_fetch(plane, s.v0, s.u0, H, W));
            y = __fadd_rn(y, __fmul_rn(w2, pad_fetch(plane, s.v0, s.u0 + 1, H, W)));
            y = __fadd_rn(y, __fmul_rn(w3, pad_fetch(plane, s.v0 + 1, s.u0, H, W)));
            y = __fadd_rn(y, __fmul_rn(w4, pad_fetch(plane, s.v0 + 1, s.u0 + 1, H, W)));
            o[c] = y;
        }
        f32x4 v = {o[0], o[1], o[2], 0.f};
        st4(rois + i * 4, v);
    }
}

__global__ __launch_bounds__(256) void st_sampler_bwd_kernel(const float* img, const float* grid, const float* grois,
                                                             float* ggrid, int accumulate, int B, int H, int W, int P) {
    const int64_t total = (int64_t)B * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / P;
        const int p = (int)(i - b * P);
        const Bilin s = bilin_setup(grid[(b * 2 + 0) * P + p], grid[(b * 2 + 1) * P + p], H, W);
        const float wu0 = (float)s.wu0, wu1 = (float)s.wu1, wv0 = (float)s.wv0, wv1 = (float)s.wv1;
        const f32x4 g = ld4(grois + i * 4);
        float gu = 0.f, gv = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* plane = img + (b * 3 + c) * (int64_t)H * W;
            const float x1 = pad_fetch(plane, s.v0, s.u0, H, W), x2 = pad_fetch(plane, s.v0, s.u0 + 1, H, W);
            const float x3 = pad_fetch(plane, s.v0 + 1, s.u0, H, W), x4 = pad_fetch(plane, s.v0 + 1, s.u0 + 1, H, W);
            const float du = -wv1 * x1 + wv1 * x2 - wv0 * x3 + wv0 * x4;
            const float dv = -wu1 * x1 - wu0 * x2 + wu1 * x3 + wu0 * x4;
            gu += du * g[c];
            gv += dv * g[c];
        }
        const bool um = s.u > 0.f && s.u < (float)(W + 1);
        const bool vm = s.v > 0.f && s.v < (float)(H + 1);
        gu = um ? gu / 2.f * (float)(W - 1) : 0.f;
        gv = vm ? gv / 2.f * (float)(H - 1) : 0.f;
        float* du_p = ggrid + (b * 2 + 0) * P + p;
        float* dv_p = ggrid + (b * 2 + 1) * P + p;
        if (accumulate) { *du_p += gu; *dv_p += gv; } else { *du_p = gu; *dv_p = gv; }
    }
}

// ---- losses --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* y, const float* t, float tconst, float* loss, int n) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = y[i] - (t ? t[i] : tconst);
        acc += d * d;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = acc / (float)n;
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* y, const float* t, float tconst, const float* gloss,
                                                      float* gy, int n) {
    const float coeff = gloss[0] * 2.f / (float)n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        gy[i] = coeff * (y[i] - (t ? t[i] : tconst));
}

// corner points of sample b: TL = (0,0), TR = (0,tw-1), BL = (th-1,0)
__global__ __launch_bounds__(256) void grid_loss_fwd_kernel(const float* grid, float* loss, int kind, float imgH,
                                                            float imgW, float oob_scale, int B, int th, int tw) {
    __shared__ float sh[4];
    const int P = th * tw;
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* gx = grid + (int64_t)b * 2 * P;
        const float* gy = gx + P;
        const float tlx = gx[0], tly = gy[0], trx = gx[tw - 1], bly = gy[(th - 1) * tw];
        if (kind == 0) {
            // (g + 1) / 2 * size, common/utils.py:145-148
            const float tly_s = (tly + 1.f) / 2.f * imgH, bly_s = (bly + 1.f) / 2.f * imgH;
            const float tlx_s = (tlx + 1.f) / 2.f * imgW, trx_s = (trx + 1.f) / 2.f * imgW;
            acc += fmaxf(tly_s - bly_s, 0.f) + fmaxf(tlx_s - trx_s, 0.f);
        } else {
            const float v[4] = {tlx, tly, trx, bly};
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += fabsf(fminf(v[k] + 1.f, 0.f)) + fmaxf(v[k] - 1.f, 0.f);
        }
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = kind == 0 ? acc / (float)B : acc * oob_scale;
}

__global__ __launch_bounds__(256) void grid_loss_bwd_kernel(const float* grid, const float* gloss, float* ggrid,
                                                            int kind, float imgH, float imgW, float oob_scale, int B,
                                                            int th, int tw) {
    const int P = th * tw;
    const float gl = gloss[0];
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        const float* gx = grid + (int64_t)b * 2 * P;
        const float* gy = gx + P;
        float* dx = ggrid + (int64_t)b * 2 * P;
        float* dy = dx + P;
        const float tlx = gx[0], tly = gy[0], trx = gx[tw - 1], bly = gy[(th - 1) * tw];
        if (kind == 0) {
            const float tly_s = (tly + 1.f) / 2.f * imgH, bly_s = (bly + 1.f) / 2.f * imgH;
            const float tlx_s = (tlx + 1.f) / 2.f * imgW, trx_s = (trx + 1.f) / 2.f * imgW;
            // F.maximum(distance, zeros), common/utils.py:169,175: Chainer's Maximum gives the gradient to its first argument
            // where x1 >= x2 -- an exact tie (distance == 0) passes it
            const float m1 = (tly_s - bly_s) >= 0.f ? gl / (float)B * (imgH / 2.f) : 0.f;
            const float m2 = (tlx_s - trx_s) >= 0.f ? gl / (float)B * (imgW / 2.f) : 0.f;
            dy[0] += m1;
            dy[(th - 1) * tw] -= m1;
            dx[0] += m2;
            dx[tw - 1] -= m2;
        } else {
            const float s = gl * oob_scale;
            // common/utils.py:312-313: |min(v + 1, 0)| has gradient sign(min(v + 1, 0)) -- zero at the tie v == -1 --,
            // max(v - 1, 0) passes the gradient where v - 1 >= 0 (Chainer's Maximum: x1 >= x2), tie included
            auto d = [&](float v) { return ((v + 1.f) < 0.f ? -s : 0.f) + ((v - 1.f) >= 0.f ? s : 0.f); };
            dx[0] += d(tlx);
            dy[0] += d(tly);
            dx[tw - 1] += d(trx);
            dy[(th - 1) * tw] += d(bly);
        }
    }
}

// ---- Adam / AMSGrad (Chainer placement of eps and bias correction) ----------------------------
struct AdamHyper { float lr_t, omb1, omb2, eps, eta, wd, gscale; };

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float& vh, const AdamHyper& h) {
    g *= h.gscale;
    m += h.omb1 * (g - m);
    v += h.omb2 * (g * g - v);
    vh = fmaxf(vh, v);
    p -= h.eta * (h.lr_t * m / (sqrtf(vh) + h.eps) + h.wd * p);
}

// vh == nullptr: plain Adam (chainer.optimizers.Adam's default amsgrad=False) -- the denominator is sqrt(v) itself
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, float* vh, int64_t n,
                                                   AdamHyper h, const float* lr_dev) {
    if (lr_dev) h.lr_t = *lr_dev;       // step-dependent rate read from device memory: the launch can live in a hipGraph
    const bool ams = vh != nullptr;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        // g, m, v, vhat are read once and m, v, vhat written once per step (900 MB at the ResNet-50 localizer: nothing of it stays in
        // the Infinity Cache anyway): non-temporal.  p stays cacheable -- the bf16 cast and the re-packs of the next step read it.
        // Alone, 78 M / 25.6 M parameters: 0.519 -> 0.502 ms, 0.198 -> 0.179 (profiles/r5_adam_nt.txt)
        f32x4 pp = ld4(p + i * 4), gg = ldnt4(g + i * 4), mm = ldnt4(m + i * 4), vv = ldnt4(v + i * 4);
        f32x4 hh = ams ? ldnt4(vh + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = pp[e], b = mm[e], c = vv[e], d = hh[e];
            adam1(a, gg[e], b, c, d, h);            // v >= 0: with d = 0 the max is v
            pp[e] = a; mm[e] = b; vv[e] = c; hh[e] = d;
        }
        st4(p + i * 4, pp); stnt4(m + i * 4, mm); stnt4(v + i * 4, vv);
        if (ams) stnt4(vh + i * 4, hh);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        float d = ams ? vh[i] : 0.f;
        adam1(p[i], g[i], m[i], v[i], d, h);
        if (ams) vh[i] = d;
    }
}

}  // namespace

extern "C" int loans_prep_images_f32(const float* images_nchw, float* out_nhwc4, int32_t B, int32_t H, int32_t W, void* stream) {
    if (!images_nchw || !out_nhwc4 || B <= 0 || H <= 0 || W <= 0) return LOANS_EINVAL;
    if ((int64_t)H * W >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(prep_kernel, dim3(grid_for((int64_t)B * H * W, 256)), dim3(256), 0, as_stream(stream), images_nchw, out_nhwc4, B, H * W);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_prep_images_dense_f32(const float* images_nchw, float* out_padded, int32_t B, int32_t H, int32_t W,
                                           int32_t pad, int32_t Hp, int32_t Wp, void* stream) {
    if (!images_nchw || !out_padded || B <= 0 || H <= 0 || W <= 0 || pad < 0) return LOANS_EINVAL;
    if (Hp < H + pad || Wp < W + pad) return LOANS_EINVAL;
    if ((int64_t)B * Hp * Wp * 3 >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(prep_dense_kernel<float>, dim3(grid_for((int64_t)B * Hp * Wp, 256)), dim3(256), 0, as_stream(stream),
                       images_nchw, out_padded, B, H, W, pad, Hp, Wp);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_prep_images_dense_bf16(const float* images_nchw, void* out_padded, int32_t B, int32_t H, int32_t W,
                                            int32_t pad, int32_t Hp, int32_t Wp, void* stream) {
    if (!images_nchw || !out_padded || B <= 0 || H <= 0 || W <= 0 || pad < 0) return LOANS_EINVAL;
    if (Hp < H + pad || Wp < W + pad) return LOANS_EINVAL;
    if ((int64_t)B * Hp * Wp * 3 >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(prep_dense_kernel<__bf16>, dim3(grid_for((int64_t)B * Hp * Wp, 256)), dim3(256), 0, as_stream(stream),
                       images_nchw, static_cast<__bf16*>(out_padded), B, H, W, pad, Hp, Wp);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_nchw3_to_nhwc4_f32(const float* in, float* out, int32_t B, int32_t H, int32_t W, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0) return LOANS_EINVAL;
    if ((int64_t)H * W >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(grid_for((int64_t)B * H * W, 256)), dim3(256), 0, as_stream(stream), in, out, B, H * W);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_gap_fwd_f32(const float* x, float* y, int32_t B, int32_t HW, int32_t C, void* stream) {
    if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || (C & 3)) return LOANS_EINVAL;
    hipLaunchKernelGGL(gap_fwd_kernel<float>, dim3(B * ((C / 4 + 63) / 64)), dim3(256), 0, as_stream(stream), x, y, B, HW, C / 4);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_gap_bwd_f32(const float* gy, float* gx, int32_t B, int32_t HW, int32_t C, void* stream) {
    if (!gy || !gx || B <= 0 || HW <= 0 || C <= 0 || (C & 3)) return LOANS_EINVAL;
    hipLaunchKernelGGL(gap_bwd_kernel<float>, dim3(grid_for((int64_t)B * HW * (C / 4), 256)), dim3(256), 0, as_stream(stream), gy, gx, B, HW, C / 4);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_gap_fwd_bf16_f32(const void* x, float* y, int32_t B, int32_t HW, int32_t C, void* stream) {
    if (!x || !y || B <= 0 || HW <= 0 || C <= 0 || (C & 3)) return LOANS_EINVAL;
    hipLaunchKernelGGL(gap_fwd_kernel<__bf16>, dim3(B * ((C / 4 + 63) / 64)), dim3(256), 0, as_stream(stream),
                       static_cast<const __bf16*>(x), y, B, HW, C / 4);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_gap_bwd_f32_bf16(const float* gy, void* gx, int32_t B, int32_t HW, int32_t C, void* stream) {
    if (!gy || !gx || B <= 0 || HW <= 0 || C <= 0 || (C & 3)) return LOANS_EINVAL;
    hipLaunchKernelGGL(gap_bwd_kernel<__bf16>, dim3(grid_for((int64_t)B * HW * (C / 4), 256)), dim3(256), 0, as_stream(stream),
                       gy, static_cast<__bf16*>(gx), B, HW, C / 4);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <typename T>
static int linear_fwd_impl(const T* x, const float* W, const float* b, float* y, int32_t B, int32_t K,
                           int32_t N, int32_t act_in, int32_t act_out, void* stream) {
    if (!x || !W || !y || B <= 0 || K <= 0 || (K & 3) || N <= 0 || N > 8) return LOANS_EINVAL;
    hipLaunchKernelGGL(linear_fwd_kernel<T>, dim3(B), dim3(256), 0, as_stream(stream), x, W, b, y, K, N, act_in, act_out);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <typename T>
static int linear_bwd_impl(const T* x, const float* W, const float* y, const float* gy, T* gx,
                           float* gW, float* gb, int32_t B, int32_t K, int32_t N, int32_t act_in,
                           int32_t act_out, void* stream) {
    if (!x || !W || !gy || B <= 0 || K <= 0 || (K & 3) || N <= 0 || N > 8) return LOANS_EINVAL;
    if (act_out && !y) return LOANS_EINVAL;
    hipStream_t st = as_stream(stream);
    const int K4 = K / 4;
    if (gx) {
        hipLaunchKernelGGL(linear_bwd_x_kernel<T>, dim3(grid_for(K4, 256, 64), B), dim3(256), 0, st, x, W, y, gy, gx, B, K4, N, act_in, act_out);
        LOANS_LAUNCH_CHECK();
    }
    if (gW) {
        const int slab = 16;
        dim3 grid((K4 + 255) / 256, (B + slab - 1) / slab);
        if (N == 1) hipLaunchKernelGGL((linear_bwd_w_kernel<1, T>), grid, dim3(256), 0, st, x, y, gy, gW, B, K4, N, act_in, act_out, slab);
        else hipLaunchKernelGGL((linear_bwd_w_kernel<8, T>), grid, dim3(256), 0, st, x, y, gy, gW, B, K4, N, act_in, act_out, slab);
        LOANS_LAUNCH_CHECK();
    }
    if (gb) {
        hipLaunchKernelGGL(linear_bwd_b_kernel, dim3(1), dim3(256), 0, st, y, gy, gb, B, N, act_out);
        LOANS_LAUNCH_CHECK();
    }
    return LOANS_OK;
}

extern "C" int loans_linear_fwd_f32(const float* x, const float* W, const float* b, float* y, int32_t B, int32_t K,
                                    int32_t N, int32_t act_in, int32_t act_out, void* stream) {
    return linear_fwd_impl<float>(x, W, b, y, B, K, N, act_in, act_out, stream);
}

extern "C" int loans_linear_bwd_f32(const float* x, const float* W, const float* y, const float* gy, float* gx,
                                    float* gW, float* gb, int32_t B, int32_t K, int32_t N, int32_t act_in,
                                    int32_t act_out, void* stream) {
    return linear_bwd_impl<float>(x, W, y, gy, gx, gW, gb, B, K, N, act_in, act_out, stream);
}

extern "C" int loans_linear_fwd_bf16(const void* x, const float* W, const float* b, float* y, int32_t B, int32_t K,
                                     int32_t N, int32_t act_in, int32_t act_out, void* stream) {
    return linear_fwd_impl<__bf16>(static_cast<const __bf16*>(x), W, b, y, B, K, N, act_in, act_out, stream);
}

extern "C" int loans_linear_bwd_bf16(const void* x, const float* W, const float* y, const float* gy, void* gx,
                                     float* gW, float* gb, int32_t B, int32_t K, int32_t N, int32_t act_in,
                                     int32_t act_out, void* stream) {
    return linear_bwd_impl<__bf16>(static_cast<const __bf16*>(x), W, y, gy, static_cast<__bf16*>(gx), gW, gb, B, K, N,
                                   act_in, act_out, stream);
}

extern "C" int loans_mul_f32(const float* x, const float* mask, float* y, int64_t n, void* stream) {
    if (!x || !mask || !y || n <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), x, mask, y, n);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_axpby_f32(float a, const float* x, float b, float* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), a, x, b, y, n);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_st_grid_fwd_f32(const float* theta, float* grid, int32_t B, int32_t th, int32_t tw, void* stream) {
    if (!theta || !grid || B <= 0 || th <= 0 || tw <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(st_grid_fwd_kernel, dim3(grid_for((int64_t)B * th * tw, 256)), dim3(256), 0, as_stream(stream), theta, grid, B, th, tw);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_st_grid_bwd_f32(const float* ggrid, float* gtheta, int32_t B, int32_t th, int32_t tw, void* stream) {
    if (!ggrid || !gtheta || B <= 0 || th <= 0 || tw <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(st_grid_bwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), ggrid, gtheta, th, tw);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_st_sampler_fwd_f32(const float* images_nchw, const float* grid, float* rois_nhwc4, int32_t B,
                                        int32_t H, int32_t W, int32_t th, int32_t tw, void* stream) {
    if (!images_nchw || !grid || !rois_nhwc4 || B <= 0 || H <= 1 || W <= 1 || th <= 0 || tw <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(st_sampler_fwd_kernel, dim3(grid_for((int64_t)B * th * tw, 256)), dim3(256), 0, as_stream(stream),
                       images_nchw, grid, rois_nhwc4, B, H, W, th * tw);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_st_sampler_bwd_grid_f32(const float* images_nchw, const float* grid, const float* grois_nhwc4,
                                             float* ggrid, int32_t accumulate, int32_t B, int32_t H, int32_t W,
                                             int32_t th, int32_t tw, void* stream) {
    if (!images_nchw || !grid || !grois_nhwc4 || !ggrid || B <= 0 || H <= 1 || W <= 1 || th <= 0 || tw <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(st_sampler_bwd_kernel, dim3(grid_for((int64_t)B * th * tw, 256)), dim3(256), 0, as_stream(stream),
                       images_nchw, grid, grois_nhwc4, ggrid, accumulate, B, H, W, th * tw);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_mse_fwd_f32(const float* y, const float* target, float tconst, float* loss, int32_t n, void* stream) {
    if (!y || !loss || n <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), y, target, tconst, loss, n);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_mse_bwd_f32(const float* y, const float* target, float tconst, const float* gloss, float* gy,
                                 int32_t n, void* stream) {
    if (!y || !gloss || !gy || n <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), y, target, tconst, gloss, gy, n);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_grid_loss_fwd_f32(const float* grid, float* loss, int32_t kind, float imgH, float imgW,
                                       float oob_scale, int32_t B, int32_t th, int32_t tw, void* stream) {
    if (!grid || !loss || (kind != 0 && kind != 1) || B <= 0 || th <= 0 || tw <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(grid_loss_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), grid, loss, kind, imgH, imgW, oob_scale, B, th, tw);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_grid_loss_bwd_f32(const float* grid, const float* gloss, float* ggrid, int32_t kind, float imgH,
                                       float imgW, float oob_scale, int32_t B, int32_t th, int32_t tw, void* stream) {
    if (!grid || !gloss || !ggrid || (kind != 0 && kind != 1) || B <= 0 || th <= 0 || tw <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(grid_loss_bwd_kernel, dim3(grid_for(B, 256)), dim3(256), 0, as_stream(stream), grid, gloss, ggrid, kind, imgH, imgW, oob_scale, B, th, tw);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_adam_amsgrad_f32(float* p, const float* g, float* m, float* v, float* vhat, int64_t n, double lr_t,
                                      double beta1, double beta2, double eps, double eta, double weight_decay_rate,
                                      double grad_scale, void* stream) {
    if (!p || !g || !m || !v || !vhat || n <= 0) return LOANS_EINVAL;
    AdamHyper h;
    h.lr_t = (float)lr_t; h.omb1 = (float)(1.0 - beta1); h.omb2 = (float)(1.0 - beta2);
    h.eps = (float)eps; h.eta = (float)eta; h.wd = (float)weight_decay_rate; h.gscale = (float)grad_scale;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, vhat, n, h,
                       (const float*)nullptr);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_adam_amsgrad_devlr_f32(float* p, const float* g, float* m, float* v, float* vhat, int64_t n,
                                            const float* lr_t_dev, double beta1, double beta2, double eps, double eta,
                                            double weight_decay_rate, double grad_scale, void* stream) {
    if (!p || !g || !m || !v || !vhat || !lr_t_dev || n <= 0) return LOANS_EINVAL;
    AdamHyper h;
    h.lr_t = 0.f; h.omb1 = (float)(1.0 - beta1); h.omb2 = (float)(1.0 - beta2);
    h.eps = (float)eps; h.eta = (float)eta; h.wd = (float)weight_decay_rate; h.gscale = (float)grad_scale;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, vhat, n, h,
                       lr_t_dev);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr_t, const float* lr_t_dev,
                              double beta1, double beta2, double eps, double eta, double weight_decay_rate, double grad_scale,
                              void* stream) {
    if (!p || !g || !m || !v || n <= 0) return LOANS_EINVAL;
    AdamHyper h;
    h.lr_t = (float)lr_t; h.omb1 = (float)(1.0 - beta1); h.omb2 = (float)(1.0 - beta2);
    h.eps = (float)eps; h.eta = (float)eta; h.wd = (float)weight_decay_rate; h.gscale = (float)grad_scale;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, (float*)nullptr, n,
                       h, lr_t_dev);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" const char* loans_hip_version(void) { return "loans_hip 0.1 (gfx950, fp32 MFMA 32x32x2)"; }
