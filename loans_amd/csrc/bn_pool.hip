// Batch-normalisation, ReLU, residual-add and max-pool kernels (NHWC, C % 4 == 0; activations fp32 or bf16 storage,
// arithmetic and per-channel coefficients always fp32).
// All of them are HBM-bound streaming passes: 16-byte accesses, grid-stride over float4
// elements, per-channel coefficients re-read from L1/L2.  Reductions go per-thread (fp32)
// -> LDS across the block -> one fp64 atomic per channel per block.
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}
__device__ __forceinline__ f32x4 maskpos4(f32x4 g, f32x4 m) {
    g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f;
    g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
    return g;
}

// 32 lanes per channel, one replica each, folded with shuffles; EVERY load of the kernel is issued before the first wait.  (One thread
// per channel walking the replicas was compiled to eight rounds of eight loads with a full wait each, the four parameter loads waited
// for one by one behind them: 5.8 us per launch, 53 of them on the forward's critical path of a ResNet-50 step -- now 3.x us,
// profiles/r5_bn_coefficient_kernels.txt.  The fp64 sums are folded as a tree now: the same value to the last bit or two of a double.)
static_assert(LOANS_STATS_REPLICAS == 32, "bn_finalize_kernel folds one replica per lane of a 32-lane group");
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* stats, int C, double inv_count, double adjust, float eps,
                                                          float decay, const float* gamma, const float* beta, float* rmean, float* rvar,
                                                          int eps_in_rv, float* mean, float* rstd, float* scale, float* shift) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), r = threadIdx.x & 31;
    const bool ok = c < C;
    const int cc = ok ? c : 0;
    double s1 = stats[(size_t)r * 2 * C + cc], s2 = stats[(size_t)r * 2 * C + C + cc];
    const float g = gamma[cc], bt = beta[cc], rm = rmean[cc], rv = rvar[cc];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 32); s2 += __shfl_xor(s2, o, 32); }
    if (!ok || r != 0) return;
    const double mu = s1 * inv_count;
    double var = s2 * inv_count - mu * mu;
    if (var < 0.0) var = 0.0;
    const double vpe = var + (double)eps;
    const float rs = (float)(1.0 / sqrt(vpe));
    const float muf = (float)mu;
    mean[c] = muf;
    rstd[c] = rs;
    const float sc = g * rs;
    scale[c] = sc;
    shift[c] = bt - muf * sc;
    rmean[c] = decay * rm + (1.f - decay) * muf;
    rvar[c] = decay * rv + (1.f - decay) * (float)(adjust * (eps_in_rv ? vpe : var));
}

__global__ void bn_eval_coeffs_kernel(int C, float eps, const float* gamma, const float* beta, const float* rmean,
                                      const float* rvar, float* mean, float* rstd, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rs = 1.f / sqrtf(rvar[c] + eps);
    mean[c] = rmean[c];
    rstd[c] = rs;
    const float sc = gamma[c] * rs;
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

// signbits (optional): one byte per four channels, bit e = (y[4i + e] > 0) -- the ReLU mask the backward passes of this BN
// need, at 1/16 (fp32) or 1/8 (bf16) of the bytes of y
__device__ __forceinline__ uint8_t posbits4(f32x4 v) {
    return (uint8_t)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
}
__device__ __forceinline__ f32x4 maskbits4(f32x4 g, uint8_t m) {
    g.x = (m & 1) ? g.x : 0.f; g.y = (m & 2) ? g.y : 0.f;
    g.z = (m & 4) ? g.z : 0.f; g.w = (m & 8) ? g.w : 0.f;
    return g;
}

template <int MODE, typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* x, const float* scale, const float* shift,
                                                       const T* x2, const float* scale2, const float* shift2,
                                                       T* y, int64_t n4, int C4, int relu, uint8_t* signbits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        f32x4 v = io4<T>::ld(x + i * 4);
        const f32x4 s = ld4(scale + c), t = ld4(shift + c);
        v = v * s + t;
        if (MODE == 1) v += io4<T>::ld(x2 + i * 4);
        if (MODE == 2) v += io4<T>::ld(x2 + i * 4) * ld4(scale2 + c) + ld4(shift2 + c);
        if (relu) v = relu4(v);
        io4<T>::st(y + i * 4, v);
        if (signbits) signbits[i] = posbits4(v);
    }
}

// One block row = one output image row (blockIdx.y walks b * OH + oh, wave-uniform: scalar divisions only); threads
// cover its OW x C4 four-channel elements with 32-bit index math (64-bit div / mod per element cost more than the loads).
template <typename TO>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const TO* x, const float* scale, const float* shift,
                                                              TO* y, uint8_t* idx, int B, int H, int W, int C4,
                                                              int OH, int OW) {
    const int rowlen = OW * C4;
    for (int row = blockIdx.y; row < B * OH; row += gridDim.y) {
        const int b = row / OH, oh = row - b * OH;
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rowlen; e += gridDim.x * blockDim.x) {
            const int ow = e / C4, c4 = e - ow * C4;
            const f32x4 s = ld4(scale + c4 * 4), t = ld4(shift + c4 * 4);
            f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int ih = oh * 2 + r;
                if (ih >= H) continue;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int iw = ow * 2 + q;
                    if (iw >= W) continue;
                    f32x4 v = relu4(io4<TO>::ld(x + (((int64_t)b * H + ih) * W + iw) * C4 * 4 + c4 * 4) * s + t);
                    const int k = r * 3 + q;
                    if (v.x > best.x) { best.x = v.x; a0 = k; }
                    if (v.y > best.y) { best.y = v.y; a1 = k; }
                    if (v.z > best.z) { best.z = v.z; a2 = k; }
                    if (v.w > best.w) { best.w = v.w; a3 = k; }
                }
            }
            const int64_t o = ((int64_t)row * OW + ow) * C4 + c4;
            io4<TO>::st(y + o * 4, best);
            *reinterpret_cast<uchar4*>(idx + o * 4) = make_uchar4(a0, a1, a2, a3);
        }
    }
}

template <typename TG>
__global__ __launch_bounds__(256) void maxpool_relu_bwd_kernel(const TG* gy, const uint8_t* idx, const TG* x,
                                                               const float* scale, const float* shift, TG* gx,
                                                               int B, int H, int W, int C4, int OH, int OW) {
    const int rowlen = W * C4;
    for (int row = blockIdx.y; row < B * H; row += gridDim.y) {     // one input image row per block row (see the forward)
        const int b = row / H, ih = row - b * H;
        const int oh_lo = ih >= 2 ? (ih - 1) >> 1 : 0, oh_hi = min(ih >> 1, OH - 1);
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rowlen; e += gridDim.x * blockDim.x) {
            const int iw = e / C4, c4 = e - iw * C4;
            f32x4 g = {0.f, 0.f, 0.f, 0.f};
            const int ow_lo = iw >= 2 ? (iw - 1) >> 1 : 0, ow_hi = min(iw >> 1, OW - 1);
            for (int oh = oh_lo; oh <= oh_hi; ++oh)
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    const int k = (ih - 2 * oh) * 3 + (iw - 2 * ow);
                    const int64_t o = (((int64_t)b * OH + oh) * OW + ow) * C4 + c4;
                    const uchar4 a = *reinterpret_cast<const uchar4*>(idx + o * 4);
                    const f32x4 gv = io4<TG>::ld(gy + o * 4);
                    if (a.x == k) g.x += gv.x;
                    if (a.y == k) g.y += gv.y;
                    if (a.z == k) g.z += gv.z;
                    if (a.w == k) g.w += gv.w;
                }
            const int64_t i = (int64_t)row * rowlen + e;
            const f32x4 pre = io4<TG>::ld(x + i * 4) * ld4(scale + c4 * 4) + ld4(shift + c4 * 4);
            io4<TG>::st(gx + i * 4, maskpos4(g, pre));
        }
    }
}

// Per-channel reductions over rows.  Thread (cg = tid % C4, rl = tid / C4) walks rows rl, rl+RL, ...
// of the block's slab; NS = number of sums per channel.
// MASK: 0 = g = gy; 1 = g = gy * (mask > 0); 2 = g = gy * (x*scale+shift > 0): the ReLU behind THIS BN, its output
// recomputed from the x being read anyway instead of fetched (the inner BNs of a residual unit: one tensor less per pass);
// 3 = `mask` points at the sign bits bn_apply wrote (one byte per four channels) instead of at the activation tensor
template <bool DUAL, int MASK, typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* gy, const T* mask, const T* x,
                                                            const float* mean, const float* rstd, const T* x2,
                                                            const float* mean2, const float* rstd2, double* sums,
                                                            int64_t rows, int C4, int C4T, int rows_per_block,
                                                            const float* scale, const float* shift) {
    constexpr int NS = DUAL ? 3 : 2;
    __shared__ f32x4 red[NS][256];
    // channels are processed in slabs of C4 (<= 256) float4 groups; blockIdx.y picks the slab, C4T = all groups
    const int tid = threadIdx.x;
    const int cl = tid % C4, rl = tid / C4, RL = 256 / C4;
    const int cg = blockIdx.y * C4 + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const f32x4 mu = ld4(mean + cg * 4), rs = ld4(rstd + cg * 4);
    f32x4 mu2 = mu, rs2 = rs;
    if (DUAL) { mu2 = ld4(mean2 + cg * 4); rs2 = ld4(rstd2 + cg * 4); }
    f32x4 sc = mu, sh = mu;
    if (MASK == 2) { sc = ld4(scale + cg * 4); sh = ld4(shift + cg * 4); }
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = sg, sgx2 = sg;
    if (rl < RL) {
        for (int64_t r = r0 + rl; r < r1; r += RL) {
            const int64_t o = (r * C4T + cg) * 4;
            f32x4 g = io4<T>::ld(gy + o);
            const f32x4 xv = io4<T>::ld(x + o);
            if (MASK == 1) g = maskpos4(g, io4<T>::ld(mask + o));
            if (MASK == 2) g = maskpos4(g, xv * sc + sh);
            if (MASK == 3) g = maskbits4(g, reinterpret_cast<const uint8_t*>(mask)[o >> 2]);
            sg += g;
            sgx += g * ((xv - mu) * rs);
            if (DUAL) sgx2 += g * ((io4<T>::ld(x2 + o) - mu2) * rs2);
        }
    }
    red[0][tid] = sg;
    red[1][tid] = sgx;
    if (DUAL) red[2][tid] = sgx2;
    __syncthreads();
    if (tid < C4) {
        for (int k = 1; k < RL; ++k) {
            sg += red[0][tid + k * C4];
            sgx += red[1][tid + k * C4];
            if (DUAL) sgx2 += red[2][tid + k * C4];
        }
        const int C = C4T * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomic_add_f64(sums + cg * 4 + e, (double)sg[e]);
            atomic_add_f64(sums + C + cg * 4 + e, (double)sgx[e]);
            if (DUAL) {
                atomic_add_f64(sums + 2 * C + cg * 4 + e, (double)sg[e]);
                atomic_add_f64(sums + 3 * C + cg * 4 + e, (double)sgx2[e]);
            }
        }
    }
}

__global__ void bn_bwd_coeffs_kernel(const double* sums, int C, double inv_count, const float* gamma, const float* mean,
                                     const float* rstd, float* ggamma, float* gbeta, float* k1, float* k2, float* k3) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double db = sums[c], dg = sums[C + c];
    ggamma[c] += (float)dg;
    gbeta[c] += (float)db;
    const double a = (double)gamma[c] * (double)rstd[c];
    k1[c] = (float)a;
    k2[c] = (float)(-a * (double)rstd[c] * dg * inv_count);
    k3[c] = (float)(a * ((double)mean[c] * (double)rstd[c] * dg - db) * inv_count);
}

// the same from the sums a data gradient's epilogue took (LOANS_F_BNSUMS): `reps` (<= 32) replicas of
// [sum g m | sum g m (y - mean)]; 32 lanes per channel fold the replicas with shuffles (a serial loop over 32 replicas cost 18 us
// per call on 2048 channels)
__global__ __launch_bounds__(256) void bn_bwd_coeffs_rep_kernel(const double* sums, int reps, int rep_stride, int centred, int C,
                                                                double inv_count, const float* gamma, const float* mean,
                                                                const float* rstd, float* ggamma, float* gbeta,
                                                                float* k1, float* k2, float* k3) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), r = threadIdx.x & 31;
    const int cc = c < C ? c : 0;
    double db = 0.0, dc = 0.0;
    if (c < C && r < reps) {
        db = sums[(size_t)r * rep_stride + c];
        dc = sums[(size_t)r * rep_stride + C + c];
    }
    // (the parameter loads go out with the replica loads, not one by one behind the fold)
    const float rs = rstd[cc], gm = gamma[cc], mu = mean[cc], gg0 = ggamma[cc], gb0 = gbeta[cc];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { db += __shfl_xor(db, o, 32); dc += __shfl_xor(dc, o, 32); }
    if (c >= C || r != 0) return;
    const double dg = centred ? dc * (double)rs : dc;  // sum g m xhat (centred: the second sum is sum g m (y - mean))
    ggamma[c] = gg0 + (float)dg;
    gbeta[c] = gb0 + (float)db;
    const double a = (double)gm * (double)rs;
    k1[c] = (float)a;
    k2[c] = (float)(-a * (double)rs * dg * inv_count);
    k3[c] = (float)(a * ((double)mu * (double)rs * dg - db) * inv_count);
}

template <bool DUAL, int MASK, typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* gy, const T* mask, const T* x,
                                                           const float* k1, const float* k2, const float* k3, T* gx,
                                                           const T* x2, const float* k1b, const float* k2b,
                                                           const float* k3b, T* gx2, int64_t n4, int C4,
                                                           const float* scale, const float* shift) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        f32x4 g = io4<T>::ld(gy + i * 4);
        const f32x4 xv = io4<T>::ld(x + i * 4);
        if (MASK == 1) g = maskpos4(g, io4<T>::ld(mask + i * 4));
        if (MASK == 2) g = maskpos4(g, xv * ld4(scale + c) + ld4(shift + c));
        if (MASK == 3) g = maskbits4(g, reinterpret_cast<const uint8_t*>(mask)[i]);
        io4<T>::st(gx + i * 4, ld4(k1 + c) * g + ld4(k2 + c) * xv + ld4(k3 + c));
        if (DUAL) io4<T>::st(gx2 + i * 4, ld4(k1b + c) * g + ld4(k2b + c) * io4<T>::ld(x2 + i * 4) + ld4(k3b + c));
    }
}

// ---- the stem's tail, fused: max-pool backward + ReLU mask + BN backward without the dense pre-BN gradient ----
// g[pixel] = (sum over the windows whose argmax is this pixel of gy) * (x*scale+shift > 0) is at most one quarter dense,
// and materialising it costs a write and two reads of the largest activation of the network (B x 112 x 112 x 64 at 224^2).
// Pass 1 takes the BN sums straight from the pooled gradient: every pooled element gathers the one x it came from.
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const __bf16* p) { return (float)*p; }

template <typename T>
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_kernel(const T* gy, const uint8_t* idx, const T* x,
                                                                 const float* scale, const float* shift,
                                                                 const float* mean, const float* rstd, double* sums,
                                                                 int rows, int H, int W, int OH, int OW,
                                                                 int C4, int C4T, int rows_per_block, int replicas) {
    __shared__ f32x4 red[2][256];
    const int tid = threadIdx.x;
    const int cl = tid % C4, rl = tid / C4, RL = 256 / C4;
    const int cg = blockIdx.y * C4 + cl;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, rows);
    const f32x4 sc = ld4(scale + cg * 4), sh = ld4(shift + cg * 4), mu = ld4(mean + cg * 4), rs = ld4(rstd + cg * 4);
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = sg;
    if (rl < RL) {
        // PF pooled elements per trip: their (gradient, argmax) loads go out together, then the PF x 4 gathers those argmaxes
        // address, then the arithmetic -- two memory latencies per PF elements instead of per element
        constexpr int PF = 4;
        auto gather = [&](int r, uchar4 a, float* xv) {          // r = (b * OH + oh) * OW + ow
            const int ow = r % OW, t = r / OW, oh = t % OH, b = t / OH;
            const int64_t base = (((int64_t)b * H + oh * 2) * W + ow * 2) * C4T * 4 + cg * 4;
            const int ks[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kr = ks[e] / 3, kq = ks[e] - kr * 3;
                xv[e] = ldf(x + base + ((int64_t)kr * W + kq) * C4T * 4 + e);
            }
        };
        auto fold = [&](f32x4 g, const float* xv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = (xv[e] * sc[e] + sh[e]) > 0.f ? g[e] : 0.f;
                sg[e] += ge;
                sgx[e] += ge * ((xv[e] - mu[e]) * rs[e]);
            }
        };
        int r = r0 + rl;
        for (; r + (PF - 1) * RL < r1; r += PF * RL) {
            f32x4 g[PF];
            uchar4 a[PF];
            float xv[PF][4];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int64_t o = ((int64_t)(r + u * RL) * C4T + cg) * 4;
                g[u] = io4<T>::ld(gy + o);
                a[u] = *reinterpret_cast<const uchar4*>(idx + o);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) gather(r + u * RL, a[u], xv[u]);
#pragma unroll
            for (int u = 0; u < PF; ++u) fold(g[u], xv[u]);
        }
        for (; r < r1; r += RL) {
            const int64_t o = ((int64_t)r * C4T + cg) * 4;
            const f32x4 g = io4<T>::ld(gy + o);
            const uchar4 a = *reinterpret_cast<const uchar4*>(idx + o);
            float xv[4];
            gather(r, a, xv);
            fold(g, xv);
        }
    }
    red[0][tid] = sg;
    red[1][tid] = sgx;
    __syncthreads();
    if (tid < C4) {
        for (int k = 1; k < RL; ++k) {
            sg += red[0][tid + k * C4];
            sgx += red[1][tid + k * C4];
        }
        const int C = C4T * 4;
        sums += (size_t)(blockIdx.x % replicas) * 2 * C;        // [replicas][2][C], see loans_bn_bwd_reduce_rep_*
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomic_add_f64(sums + cg * 4 + e, (double)sg[e]);
            atomic_add_f64(sums + C + cg * 4 + e, (double)sgx[e]);
        }
    }
}

// Pass 2: gx = k1 * g + k2 * x + k3 with g rebuilt on the fly.  One thread owns a 2 x 2 patch of input pixels (4 channels):
// the patch at (2a, 2b) is covered by the four windows (a-1 | a, b-1 | b) only, so each window's (idx, gy) is loaded once
// per four outputs (a thread per pixel loads nine); contributions are added in the window order of maxpool_relu_bwd.
template <typename T>
__global__ __launch_bounds__(256) void pool_bn_bwd_apply_kernel(const T* gy, const uint8_t* idx, const T* x,
                                                                const float* scale, const float* shift,
                                                                const float* k1, const float* k2, const float* k3, T* gx,
                                                                float* gxsum, int B, int H, int W, int C4, int OH, int OW) {
    const int PH = (H + 1) >> 1, PW = (W + 1) >> 1;
    const int rowlen = PW * C4;
    // gxsum (optional, C4 divides 256): per-channel sum of gx = the gradient of the bias of the convolution in front of
    // this BN (conv1 has one, sheep/resnet.py:43); a thread's channel group is fixed (every stride is a multiple of C4)
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    for (int row = blockIdx.y; row < B * PH; row += gridDim.y) {
        const int b = row / PH, a = row - b * PH;
        for (int el = blockIdx.x * blockDim.x + threadIdx.x; el < rowlen; el += gridDim.x * blockDim.x) {
            const int pb = el / C4, c4 = el - pb * C4;
            f32x4 g[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) g[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int wi = 0; wi < 2; ++wi) {
                const int oh = a - 1 + wi;
                if (oh < 0 || oh >= OH) continue;
#pragma unroll
                for (int wj = 0; wj < 2; ++wj) {
                    const int ow = pb - 1 + wj;
                    if (ow < 0 || ow >= OW) continue;
                    const int64_t o = ((((int64_t)b * OH + oh) * OW + ow) * C4 + c4) * 4;
                    const uchar4 av = *reinterpret_cast<const uchar4*>(idx + o);
                    const f32x4 gv = io4<T>::ld(gy + o);
                    // window (a-1+wi, pb-1+wj) starts at pixel (2a-2+2wi, 2pb-2+2wj): patch pixel (i, j) is its
                    // position (i + 2 - 2wi, j + 2 - 2wj) where that is < 3
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int kr = i + 2 - 2 * wi;
                        if (kr > 2) continue;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int kq = j + 2 - 2 * wj;
                            if (kq > 2) continue;
                            const int k = kr * 3 + kq;
                            if (av.x == k) g[i][j].x += gv.x;
                            if (av.y == k) g[i][j].y += gv.y;
                            if (av.z == k) g[i][j].z += gv.z;
                            if (av.w == k) g[i][j].w += gv.w;
                        }
                    }
                }
            }
            const f32x4 sc = ld4(scale + c4 * 4), sh = ld4(shift + c4 * 4);
            const f32x4 c1 = ld4(k1 + c4 * 4), c2 = ld4(k2 + c4 * 4), c3 = ld4(k3 + c4 * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ih = 2 * a + i;
                if (ih >= H) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int iw = 2 * pb + j;
                    if (iw >= W) continue;
                    const int64_t p = ((((int64_t)b * H + ih) * W + iw) * C4 + c4) * 4;
                    const f32x4 xv = io4<T>::ld(x + p);
                    const f32x4 v = c1 * maskpos4(g[i][j], xv * sc + sh) + c2 * xv + c3;
                    io4<T>::st(gx + p, v);
                    bsum += v;
                }
            }
        }
    }
    if (gxsum) {
        __shared__ f32x4 red[256];
        const int tid = threadIdx.x;
        red[tid] = bsum;
        __syncthreads();
        if (tid < C4) {
            for (int k = tid + C4; k < 256; k += C4) bsum += red[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) atomic_add_f32(gxsum + tid * 4 + e, bsum[e]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, float* out, int64_t rows, int C4, int C4T, int rows_per_block) {
    __shared__ f32x4 red[256];
    const int tid = threadIdx.x;
    const int cl = tid % C4, rl = tid / C4, RL = 256 / C4;
    const int cg = blockIdx.y * C4 + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (rl < RL)
        for (int64_t r = r0 + rl; r < r1; r += RL) s += io4<T>::ld(x + (r * C4T + cg) * 4);
    red[tid] = s;
    __syncthreads();
    if (tid < C4) {
        for (int k = 1; k < RL; ++k) s += red[tid + k * C4];
#pragma unroll
        for (int e = 0; e < 4; ++e) atomic_add_f32(out + cg * 4 + e, s[e]);
    }
}

// Epilogue of a split-K convolution over the finished sums, in place: v = out (+ bias); MASK: v *= (ref > 0);
// ADDEND (masked by ref > 0 with ADDEND_MASK): v += addend; STATS: per-channel sum / sum of squares of v (fp64 atomics).
// Same thread map as the reductions above.
__global__ __launch_bounds__(256) void igemm_finalize_kernel(float* out, const float* bias, double* stats, const float* ref,
                                                             const float* addend, int flags, int64_t rows, int C4, int C4T,
                                                             int rows_per_block) {
    __shared__ f32x4 red[2][256];
    const int tid = threadIdx.x;
    const int cl = tid % C4, rl = tid / C4, RL = 256 / C4;
    const int cg = blockIdx.y * C4 + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const bool f_bias = flags & LOANS_F_BIAS, f_stats = flags & LOANS_F_STATS, f_mask = flags & LOANS_F_MASK;
    const bool f_add = flags & LOANS_F_ADDEND, f_addmask = flags & LOANS_F_ADDEND_MASK;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (f_bias) bv = ld4(bias + cg * 4);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (rl < RL) {
        for (int64_t r = r0 + rl; r < r1; r += RL) {
            const int64_t o = (r * C4T + cg) * 4;
            f32x4 v = ld4(out + o) + bv;
            if (f_mask || f_addmask) {
                const f32x4 m = ld4(ref + o);
                if (f_mask) v = maskpos4(v, m);
                if (f_add) v += f_addmask ? maskpos4(ld4(addend + o), m) : ld4(addend + o);
            } else if (f_add) {
                v += ld4(addend + o);
            }
            st4(out + o, v);
            s1 += v;
            s2 += v * v;
        }
    }
    if (!f_stats) return;
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    if (tid < C4) {
        for (int k = 1; k < RL; ++k) {
            s1 += red[0][tid + k * C4];
            s2 += red[1][tid + k * C4];
        }
        const int C = C4T * 4;
        double* st = stats + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomic_add_f64(st + cg * 4 + e, (double)s1[e]);
            atomic_add_f64(st + C + cg * 4 + e, (double)s2[e]);
        }
    }
}


// ---- the streaming passes on 16-byte units (round 3) ---------------------------------------------------------------------------
// The kernels above walk float4 (fp32) / 8-byte (bf16) elements with a 64-bit modulo per element for the channel and re-read
// their per-channel coefficients in every iteration: measured alone on a res2-sized bf16 tensor they reach 3.1 - 3.6 TB/s
// where a plain copy does 4.7 and a plain read 6.5 (tools/bn_bench2.py) -- 60 % of what the same bytes cost a streaming
// kernel.  Here a thread handles one 16-BYTE unit per tensor and iteration (4 fp32 or 8 bf16 channels) and, because the unit
// count per row U = C / V divides the block size and the grid stride is a multiple of it, ALWAYS THE SAME channels: the
// coefficients are loaded once into registers, the loop is loads, a few FMAs and stores, two units in flight.  Offered where
// U <= 256 divides 256 (every BN of both localizers); other channel counts keep the kernels above.
template <typename T> struct unit16;
template <> struct unit16<float> {
    static constexpr int V = 4, NV = 1;
    typedef f32x4 raw_t;
    template <bool NT = false> static __device__ __forceinline__ raw_t ldr(const float* p) {
        if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        return *reinterpret_cast<const f32x4*>(p);
    }
    static __device__ __forceinline__ void cvt(raw_t r, f32x4* v) { v[0] = r; }
    static __device__ __forceinline__ void ld(const float* p, f32x4* v) { v[0] = *reinterpret_cast<const f32x4*>(p); }
    template <bool NT = false> static __device__ __forceinline__ void st(float* p, const f32x4* v) {
        if (NT) __builtin_nontemporal_store(v[0], reinterpret_cast<f32x4*>(p));
        else *reinterpret_cast<f32x4*>(p) = v[0];
    }
};
template <> struct unit16<__bf16> {
    static constexpr int V = 8, NV = 2;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef bf16x8 raw_t;
    template <bool NT = false> static __device__ __forceinline__ raw_t ldr(const __bf16* p) {
        if (NT) return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
        return *reinterpret_cast<const bf16x8*>(p);
    }
    static __device__ __forceinline__ void cvt(raw_t r, f32x4* v) {
        v[0] = __builtin_convertvector(__builtin_shufflevector(r, r, 0, 1, 2, 3), f32x4);
        v[1] = __builtin_convertvector(__builtin_shufflevector(r, r, 4, 5, 6, 7), f32x4);
    }
    static __device__ __forceinline__ void ld(const __bf16* p, f32x4* v) { cvt(ldr(p), v); }
    template <bool NT = false> static __device__ __forceinline__ void st(__bf16* p, const f32x4* v) {
        const loans_bf16x4 a = __builtin_convertvector(v[0], loans_bf16x4), b = __builtin_convertvector(v[1], loans_bf16x4);
        const bf16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
        if (NT) __builtin_nontemporal_store(r, reinterpret_cast<bf16x8*>(p));
        else *reinterpret_cast<bf16x8*>(p) = r;
    }
};

// The units one block streams (round 3, tools/stream_patterns.hip): its own CONTIGUOUS slab, a multiple of 256 units long so
// that a thread's channel unit stays tid % U, walked 256 units at a time with SLAB_UNR steps in flight.  Two reads + one write
// of 512 MiB bf16 tensors: 5.3-5.8 TB/s like this against 4.3-4.8 for a grid-stride walk of persistent blocks (the form of
// the first u16 kernels: every block touching every part of the tensor) -- what an HBM page sees is a dense burst, not a
// trickle from 4096 blocks.  NT: non-temporal loads and stores (+5 % on tensors far beyond the 256 MB Infinity Cache).
constexpr int SLAB_UNR = 4;
struct Slab { int64_t i, end; };
__device__ __forceinline__ Slab slab_of(int64_t nunits) {
    const int64_t per = ((nunits + gridDim.x - 1) / gridDim.x + 255) & ~(int64_t)255;
    const int64_t lo = (int64_t)blockIdx.x * per;
    return Slab{lo + threadIdx.x, lo + per < nunits ? lo + per : nunits};
}
static inline int slab_grid(int64_t nunits) { return grid_for(nunits, 256 * SLAB_UNR, 8192); }
// LOANS_BN_NT: 0 never, 1 always, unset: tensors of at least 96 MB (they cannot stay in the Infinity Cache with their partners)
static inline bool slab_nt(int64_t tensor_bytes) {
    static const int mode = [] { const char* e = getenv("LOANS_BN_NT"); return e && *e ? atoi(e) : -1; }();
    constexpr long mb = 96;
    return mode < 0 ? tensor_bytes >= ((int64_t)mb << 20) : mode != 0;
}

// one byte of sign bits per four channels: a unit owns NV consecutive bytes
template <int NV> __device__ __forceinline__ void st_bits(uint8_t* p, const f32x4* v) {
    if (NV == 1) p[0] = posbits4(v[0]);
    else *reinterpret_cast<uint16_t*>(p) = (uint16_t)(posbits4(v[0]) | (posbits4(v[1]) << 8));
}
template <int NV> __device__ __forceinline__ void ld_bits(const uint8_t* p, uint8_t* b) {
    if (NV == 1) b[0] = p[0];
    else { const uint16_t w = *reinterpret_cast<const uint16_t*>(p); b[0] = (uint8_t)(w & 255); b[1] = (uint8_t)(w >> 8); }
}

template <int MODE, bool BITS, bool NT, typename T>
__global__ __launch_bounds__(256) void bn_apply_u16_kernel(const T* x, const float* scale, const float* shift, const T* x2,
                                                           const float* scale2, const float* shift2, T* y, int64_t nunits,
                                                           int U, int relu, uint8_t* signbits) {
    constexpr int V = unit16<T>::V, NV = unit16<T>::NV;
    typedef typename unit16<T>::raw_t raw_t;
    const int cu = threadIdx.x % U;             // this thread's channel unit, the same in every iteration
    f32x4 s[NV], t[NV], s2[NV], t2[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        s[q] = ld4(scale + cu * V + 4 * q); t[q] = ld4(shift + cu * V + 4 * q);
        if (MODE == 2) { s2[q] = ld4(scale2 + cu * V + 4 * q); t2[q] = ld4(shift2 + cu * V + 4 * q); }
    }
    auto one = [&](int64_t i, raw_t xr, raw_t rr) {
        f32x4 v[NV], xv[NV], rv[NV];
        unit16<T>::cvt(xr, xv);
        if (MODE != 0) unit16<T>::cvt(rr, rv);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            v[q] = xv[q] * s[q] + t[q];
            if (MODE == 1) v[q] += rv[q];
            if (MODE == 2) v[q] += rv[q] * s2[q] + t2[q];
            if (relu) v[q] = relu4(v[q]);
        }
        unit16<T>::template st<NT>(y + i * V, v);
        if (BITS) st_bits<NV>(signbits + i * NV, v);
    };
    Slab sl = slab_of(nunits);
    int64_t i = sl.i;
    for (; i + (SLAB_UNR - 1) * 256 < sl.end; i += SLAB_UNR * 256) {
        raw_t a[SLAB_UNR], r[SLAB_UNR];
#pragma unroll
        for (int u = 0; u < SLAB_UNR; ++u) {
            a[u] = unit16<T>::template ldr<NT>(x + (i + u * 256) * V);
            if (MODE != 0) r[u] = unit16<T>::template ldr<NT>(x2 + (i + u * 256) * V);
        }
#pragma unroll
        for (int u = 0; u < SLAB_UNR; ++u) one(i + u * 256, a[u], r[u]);
    }
    for (; i < sl.end; i += 256) {
        raw_t a = unit16<T>::template ldr<NT>(x + i * V), r = a;
        if (MODE != 0) r = unit16<T>::template ldr<NT>(x2 + i * V);
        one(i, a, r);
    }
}

// gx = k1 g + k2 x + k3 (and gx2 for the second BN of a dual); MASK as in bn_bwd_apply_kernel
template <bool DUAL, int MASK, bool NT, typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_u16_kernel(const T* gy, const T* mask, const T* x, const float* k1,
                                                               const float* k2, const float* k3, T* gx, const T* x2,
                                                               const float* k1b, const float* k2b, const float* k3b, T* gx2,
                                                               int64_t nunits, int U, const float* scale, const float* shift) {
    constexpr int V = unit16<T>::V, NV = unit16<T>::NV;
    typedef typename unit16<T>::raw_t raw_t;
    const int cu = threadIdx.x % U;
    f32x4 a1[NV], a2[NV], a3[NV], b1[NV], b2[NV], b3[NV], sc[NV], sh[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int c = cu * V + 4 * q;
        a1[q] = ld4(k1 + c); a2[q] = ld4(k2 + c); a3[q] = ld4(k3 + c);
        if (DUAL) { b1[q] = ld4(k1b + c); b2[q] = ld4(k2b + c); b3[q] = ld4(k3b + c); }
        if (MASK == 2) { sc[q] = ld4(scale + c); sh[q] = ld4(shift + c); }
    }
    const uint8_t* bits = reinterpret_cast<const uint8_t*>(mask);
    auto one = [&](int64_t i, raw_t gr, raw_t xr, raw_t mr, const uint8_t* mb, raw_t wr) {
        f32x4 o[NV], g[NV], xv[NV], mv[NV], xw[NV];
        unit16<T>::cvt(gr, g);
        unit16<T>::cvt(xr, xv);
        if (MASK == 1) unit16<T>::cvt(mr, mv);
        if (DUAL) unit16<T>::cvt(wr, xw);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if (MASK == 1) g[q] = maskpos4(g[q], mv[q]);
            if (MASK == 2) g[q] = maskpos4(g[q], xv[q] * sc[q] + sh[q]);
            if (MASK == 3) g[q] = maskbits4(g[q], mb[q]);
            o[q] = a1[q] * g[q] + a2[q] * xv[q] + a3[q];
        }
        unit16<T>::template st<NT>(gx + i * V, o);
        if (DUAL) {
#pragma unroll
            for (int q = 0; q < NV; ++q) o[q] = b1[q] * g[q] + b2[q] * xw[q] + b3[q];
            unit16<T>::template st<NT>(gx2 + i * V, o);
        }
    };
    Slab sl = slab_of(nunits);
    int64_t i = sl.i;
    for (; i + (SLAB_UNR - 1) * 256 < sl.end; i += SLAB_UNR * 256) {
        raw_t g[SLAB_UNR], xv[SLAB_UNR], mv[SLAB_UNR], xw[SLAB_UNR];
        uint8_t mb[SLAB_UNR][NV];
#pragma unroll
        for (int u = 0; u < SLAB_UNR; ++u) {
            const int64_t j = i + u * 256;
            g[u] = unit16<T>::template ldr<NT>(gy + j * V);
            xv[u] = unit16<T>::template ldr<NT>(x + j * V);
            if (MASK == 1) mv[u] = unit16<T>::template ldr<NT>(mask + j * V);
            if (MASK == 3) ld_bits<NV>(bits + j * NV, mb[u]);
            if (DUAL) xw[u] = unit16<T>::template ldr<NT>(x2 + j * V);
        }
#pragma unroll
        for (int u = 0; u < SLAB_UNR; ++u) one(i + u * 256, g[u], xv[u], mv[u], mb[u], xw[u]);
    }
    for (; i < sl.end; i += 256) {
        raw_t g = unit16<T>::template ldr<NT>(gy + i * V), xv = unit16<T>::template ldr<NT>(x + i * V), mv = g, xw = g;
        uint8_t mb[NV];
        if (MASK == 1) mv = unit16<T>::template ldr<NT>(mask + i * V);
        if (MASK == 3) ld_bits<NV>(bits + i * NV, mb);
        if (DUAL) xw = unit16<T>::template ldr<NT>(x2 + i * V);
        one(i, g, xv, mv, mb, xw);
    }
}

// the two (dual: three) per-channel sums of a BN backward over 16-byte units: thread (cu = tid % U, rl = tid / U) walks the rows
// rl, rl + RL, ... of its block's slab (no index arithmetic beyond an add), the block folds through LDS, one fp64 atomic per
// channel and sum
template <bool DUAL, int MASK, typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_u16_kernel(const T* gy, const T* mask, const T* x, const float* mean,
                                                                const float* rstd, const T* x2, const float* mean2,
                                                                const float* rstd2, double* sums, int64_t rows, int U,
                                                                int rows_per_block, const float* scale, const float* shift,
                                                                int replicas, int UB) {
    constexpr int V = unit16<T>::V, NV = unit16<T>::NV, NS = DUAL ? 3 : 2;
    __shared__ f32x4 red[NS * NV][256];
    // wide rows are cut into slabs of UB units (blockIdx.y): a block's final atomics are 2 x its channels, and with all of a
    // 2048-channel row in one block they outnumbered the loads 1 : 16 (fp64 atomics retire ~20 x slower than fp32 ones)
    const int tid = threadIdx.x, cl = tid % UB, cu = blockIdx.y * UB + cl, rl = tid / UB, RL = 256 / UB;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    f32x4 mu[NV], rs[NV], mu2[NV], rs2[NV], sc[NV], sh[NV], sg[NV], sgx[NV], sgx2[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int c = cu * V + 4 * q;
        mu[q] = ld4(mean + c); rs[q] = ld4(rstd + c);
        if (DUAL) { mu2[q] = ld4(mean2 + c); rs2[q] = ld4(rstd2 + c); }
        if (MASK == 2) { sc[q] = ld4(scale + c); sh[q] = ld4(shift + c); }
        sg[q] = sgx[q] = sgx2[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const uint8_t* bits = reinterpret_cast<const uint8_t*>(mask);
    // UF rows per iteration, their loads issued before any arithmetic: a thread that waits for one 16-byte load pair at a
    // time leaves the CU with ~16 KB in flight (wide-channel tensors run one block per CU), a quarter of what HBM needs
    constexpr int UF = DUAL ? 2 : 4;
    auto fold = [&](f32x4* g, const f32x4* xv, const f32x4* mv, const uint8_t* mb, const f32x4* xw) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if (MASK == 1) g[q] = maskpos4(g[q], mv[q]);
            if (MASK == 2) g[q] = maskpos4(g[q], xv[q] * sc[q] + sh[q]);
            if (MASK == 3) g[q] = maskbits4(g[q], mb[q]);
            sg[q] += g[q];
            sgx[q] += g[q] * ((xv[q] - mu[q]) * rs[q]);
            if (DUAL) sgx2[q] += g[q] * ((xw[q] - mu2[q]) * rs2[q]);
        }
    };
    int64_t r = r0 + rl;
    for (; r + (int64_t)(UF - 1) * RL < r1; r += (int64_t)UF * RL) {
        f32x4 g[UF][NV], xv[UF][NV], mv[UF][NV], xw[UF][NV];
        uint8_t mb[UF][NV];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int64_t i = (r + (int64_t)u * RL) * U + cu;
            unit16<T>::ld(gy + i * V, g[u]);
            unit16<T>::ld(x + i * V, xv[u]);
            if (MASK == 1) unit16<T>::ld(mask + i * V, mv[u]);
            if (MASK == 3) ld_bits<NV>(bits + i * NV, mb[u]);
            if (DUAL) unit16<T>::ld(x2 + i * V, xw[u]);
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) fold(g[u], xv[u], mv[u], mb[u], xw[u]);
    }
    for (; r < r1; r += RL) {
        const int64_t i = r * U + cu;
        f32x4 g[NV], xv[NV], mv[NV], xw[NV];
        uint8_t mb[NV];
        unit16<T>::ld(gy + i * V, g);
        unit16<T>::ld(x + i * V, xv);
        if (MASK == 1) unit16<T>::ld(mask + i * V, mv);
        if (MASK == 3) ld_bits<NV>(bits + i * NV, mb);
        if (DUAL) unit16<T>::ld(x2 + i * V, xw);
        fold(g, xv, mv, mb, xw);
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        red[q][tid] = sg[q];
        red[NV + q][tid] = sgx[q];
        if (DUAL) red[2 * NV + q][tid] = sgx2[q];
    }
    __syncthreads();
    if (tid < UB) {
        const int C = U * V;
        // `replicas` accumulators [replica][2 | 4][C], picked by block: thousands of blocks adding to the same 2 C addresses
        // cost more than the pass itself (measured: +0.1 ms per 1024 blocks on 128 addresses)
        sums += (size_t)(blockIdx.x % replicas) * (DUAL ? 4 : 2) * C;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            f32x4 a = sg[q], b = sgx[q], c2 = sgx2[q];
            for (int k = 1; k < RL; ++k) {
                a += red[q][tid + k * UB];
                b += red[NV + q][tid + k * UB];
                if (DUAL) c2 += red[2 * NV + q][tid + k * UB];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = cu * V + 4 * q + e;
                atomic_add_f64(sums + ch, (double)a[e]);
                atomic_add_f64(sums + C + ch, (double)b[e]);
                if (DUAL) {
                    atomic_add_f64(sums + 2 * C + ch, (double)a[e]);
                    atomic_add_f64(sums + 3 * C + ch, (double)c2[e]);
                }
            }
        }
    }
}


// ---- the stem's pool on 16-byte units (round 3; see the unit16 kernels above) -----------------------------------------------------
// One thread per (pooled pixel, 16-byte channel unit); the thread's channels never change (the unit count per row divides the
// block size), so the BN coefficients live in registers: 0.127 -> 0.109 ms on a quarter of configs[2]'s conv1 output.  The same
// treatment of the fused BACKWARD (one 16-byte unit per 2 x 2 patch / per pooled pixel with whole-unit window loads) was built
// and measured SLOWER than the kernels above (apply 0.39 -> 0.83 ms, reduce 0.16 -> 0.28 ms at 100-126 VGPRs): removed.
template <int NV> __device__ __forceinline__ void ld_idx(const uint8_t* p, uint8_t* k) {
    if (NV == 1) { const uchar4 a = *reinterpret_cast<const uchar4*>(p); k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; }
    else {
        const uint2 a = *reinterpret_cast<const uint2*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) { k[e] = (uint8_t)(a.x >> (8 * e)); k[4 + e] = (uint8_t)(a.y >> (8 * e)); }
    }
}
template <int NV> __device__ __forceinline__ void st_idx(uint8_t* p, const int* k) {
    if (NV == 1) *reinterpret_cast<uchar4*>(p) = make_uchar4(k[0], k[1], k[2], k[3]);
    else {
        uint2 a;
        a.x = (unsigned)k[0] | ((unsigned)k[1] << 8) | ((unsigned)k[2] << 16) | ((unsigned)k[3] << 24);
        a.y = (unsigned)k[4] | ((unsigned)k[5] << 8) | ((unsigned)k[6] << 16) | ((unsigned)k[7] << 24);
        *reinterpret_cast<uint2*>(p) = a;
    }
}

// SEL: also writes xsel = the RAW x at each pooled element's argmax (the tensor the stem's BN-backward sums then read
// contiguously instead of gathering it out of the four times larger x through idx: loans_bn_relu_maxpool_sel_*)
template <typename T, bool SEL>
__global__ __launch_bounds__(256) void bn_relu_maxpool_u16_kernel(const T* x, const float* scale, const float* shift, T* y,
                                                                  uint8_t* idx, T* xsel, int B, int H, int W, int U, int OH, int OW) {
    constexpr int V = unit16<T>::V, NV = unit16<T>::NV;
    const int cu = threadIdx.x % U, pl = threadIdx.x / U, PL = 256 / U;      // PL output pixels of a row per block pass
    f32x4 s[NV], t[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) { s[q] = ld4(scale + cu * V + 4 * q); t[q] = ld4(shift + cu * V + 4 * q); }
    for (int row = blockIdx.y; row < B * OH; row += gridDim.y) {
        const int b = row / OH, oh = row - b * OH;
        for (int ow = blockIdx.x * PL + pl; ow < OW; ow += gridDim.x * PL) {
            f32x4 best[NV], bestx[NV];
            int arg[V];
#pragma unroll
            for (int q = 0; q < NV; ++q) { best[q] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; bestx[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int e = 0; e < V; ++e) arg[e] = 0;
            // the nine window loads first (clamped addresses; a load behind each bounds test is a memory latency of its own)
            typename unit16<T>::raw_t raw[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ih = min(oh * 2 + r, H - 1), iw = min(ow * 2 + c, W - 1);
                    raw[r][c] = unit16<T>::ldr(x + ((((int64_t)b * H + ih) * W + iw) * U + cu) * V);
                }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (oh * 2 + r >= H) continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (ow * 2 + c >= W) continue;
                    f32x4 v[NV];
                    unit16<T>::cvt(raw[r][c], v);
#pragma unroll
                    for (int q = 0; q < NV; ++q) {
                        const f32x4 raw_v = v[q];
                        v[q] = relu4(v[q] * s[q] + t[q]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (v[q][e] > best[q][e]) {
                                best[q][e] = v[q][e]; arg[4 * q + e] = r * 3 + c;
                                if (SEL) bestx[q][e] = raw_v[e];
                            }
                    }
                }
            }
            const int64_t o = (((int64_t)row * OW + ow) * U + cu) * V;
            unit16<T>::st(y + o, best);
            st_idx<NV>(idx + o, arg);
            if (SEL) unit16<T>::st(xsel + o, bestx);
        }
    }
}


// pool_bn_bwd_apply_kernel with the thread's four channels fixed (C4 divides the block size): coefficients in registers, no
// division per element -- the 2 x 2-patch scheme itself unchanged (the 16-byte-unit form of it ran slower, see above)
template <typename T>
__global__ __launch_bounds__(256) void pool_bn_bwd_apply_v4_kernel(const T* gy, const uint8_t* idx, const T* x, const float* scale,
                                                                   const float* shift, const float* k1, const float* k2,
                                                                   const float* k3, T* gx, float* gxsum, int B, int H, int W,
                                                                   int C4, int OH, int OW, int gx_reps) {
    const int PH = (H + 1) >> 1, PW = (W + 1) >> 1;
    const int tid = threadIdx.x, c4 = tid % C4, pl = tid / C4, PL = 256 / C4;
    const f32x4 sc = ld4(scale + c4 * 4), sh = ld4(shift + c4 * 4);
    const f32x4 c1 = ld4(k1 + c4 * 4), c2 = ld4(k2 + c4 * 4), c3 = ld4(k3 + c4 * 4);
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    for (int row = blockIdx.y; row < B * PH; row += gridDim.y) {
        const int b = row / PH, a = row - b * PH;
        for (int pb = blockIdx.x * PL + pl; pb < PW; pb += gridDim.x * PL) {
            // all twelve loads of the patch are issued before any arithmetic (clamped addresses, validity kept aside): with a
            // load behind every bounds test the thread paid two memory latencies per patch, one after the other
            f32x4 gv[2][2], xv[2][2];
            uchar4 av[2][2];
            bool wok[2][2], xok[2][2];
            int64_t xp[2][2];
#pragma unroll
            for (int wi = 0; wi < 2; ++wi)
#pragma unroll
                for (int wj = 0; wj < 2; ++wj) {
                    const int oh = a - 1 + wi, ow = pb - 1 + wj;
                    wok[wi][wj] = (unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW;
                    const int ohc = min(max(oh, 0), OH - 1), owc = min(max(ow, 0), OW - 1);
                    const int64_t o = ((((int64_t)b * OH + ohc) * OW + owc) * C4 + c4) * 4;
                    av[wi][wj] = *reinterpret_cast<const uchar4*>(idx + o);
                    gv[wi][wj] = io4<T>::ld(gy + o);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ih = 2 * a + i, iw = 2 * pb + j;
                    xok[i][j] = ih < H && iw < W;
                    xp[i][j] = ((((int64_t)b * H + min(ih, H - 1)) * W + min(iw, W - 1)) * C4 + c4) * 4;
                    xv[i][j] = io4<T>::ld(x + xp[i][j]);
                }
            f32x4 g[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) g[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int wi = 0; wi < 2; ++wi) {
#pragma unroll
                for (int wj = 0; wj < 2; ++wj) {
                    const f32x4 gw = wok[wi][wj] ? gv[wi][wj] : f32x4{0.f, 0.f, 0.f, 0.f};
                    const uchar4 aw = av[wi][wj];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int kr = i + 2 - 2 * wi;
                        if (kr > 2) continue;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int kq = j + 2 - 2 * wj;
                            if (kq > 2) continue;
                            const int k = kr * 3 + kq;
                            if (aw.x == k) g[i][j].x += gw.x;
                            if (aw.y == k) g[i][j].y += gw.y;
                            if (aw.z == k) g[i][j].z += gw.z;
                            if (aw.w == k) g[i][j].w += gw.w;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!xok[i][j]) continue;
                    const f32x4 v = c1 * maskpos4(g[i][j], xv[i][j] * sc + sh) + c2 * xv[i][j] + c3;
                    io4<T>::st(gx + xp[i][j], v);
                    bsum += v;
                }
        }
    }
    if (gxsum) {
        __shared__ f32x4 red[256];
        red[tid] = bsum;
        __syncthreads();
        if (tid < C4) {
            for (int k = tid + C4; k < 256; k += C4) bsum += red[k];
            // gx_reps replicas [rep][C]: the blocks' closing atomics on ONE set of C addresses made the pass slower the more
            // blocks it had (0.33 ms at 1024 blocks, 0.50 at 4096 on a quarter of configs[2]'s conv1 output)
            float* dst = gxsum + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) % gx_reps) * C4 * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomic_add_f32(dst + tid * 4 + e, bsum[e]);
        }
    }
}

// dst[c] += sum over r of src[r][c]
// (32 lanes per channel, lane r takes replicas r, r + 32, ...: one round of loads in flight instead of `reps` dependent ones)
__global__ __launch_bounds__(256) void fold_replicas_kernel(const float* src, float* dst, int reps, int C) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), r0 = threadIdx.x & 31;
    const int cc = c < C ? c : 0;
    const float d0 = dst[cc];
    float a = 0.f;
    if (c < C)
        for (int r = r0; r < reps; r += 32) a += src[(size_t)r * C + c];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 32);
    if (c < C && r0 == 0) dst[c] = d0 + a;
}

// a channel count the 16-byte-unit kernels tile: U = C / V units per row, U <= 256 and 256 % U == 0
template <typename T> static inline int units_per_row(int C) {
    constexpr int V = unit16<T>::V;
    if (C % V) return 0;
    const int U = C / V;
    return (U <= 256 && 256 % U == 0) ? U : 0;
}

// channel slabs: C4 = float4 groups per block (<= 256, divides 256), *slabs = number of slabs
int reduce_geometry(int64_t rows, int C, int* rows_per_block, int* c4_block, int* slabs) {
    const int C4T = C / 4;
    const int C4 = C4T > 256 ? 256 : C4T;
    *c4_block = C4;
    *slabs = C4T / C4;
    const int RL = 256 / C4;
    int64_t rpb = ((rows + 1023) / 1024 + RL - 1) / RL * RL;   // <= 1024 blocks
    if (rpb < 8 * RL) rpb = 8 * RL;
    *rows_per_block = (int)rpb;
    return (int)((rows + rpb - 1) / rpb);
}

// C/4 float4 groups must tile 256-thread blocks: a divisor of 256, or a multiple of 256 (slabs)
bool reduce_channels_ok(int C) {
    if (C < 4 || (C & 3)) return false;
    const int c4 = C / 4;
    return c4 <= 256 ? (256 % c4 == 0) : (c4 % 256 == 0);
}

}  // namespace

extern "C" int loans_bn_finalize_f32(const double* stats, int32_t C, int64_t count, float eps, float decay,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     int32_t eps_in_running_var, float* mean, float* rstd, float* scale, float* shift,
                                     void* stream) {
    if (!stats || !gamma || !beta || !running_mean || !running_var || !mean || !rstd || !scale || !shift) return LOANS_EINVAL;
    if (C <= 0 || count <= 0) return LOANS_EINVAL;
    const double adjust = (double)count / (count > 1 ? (double)(count - 1) : 1.0);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, as_stream(stream), stats, C,
                       1.0 / (double)count, adjust, eps, decay, gamma, beta, running_mean, running_var,
                       eps_in_running_var, mean, rstd, scale, shift);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_bn_eval_coeffs_f32(int32_t C, float eps, const float* gamma, const float* beta,
                                        const float* running_mean, const float* running_var, float* mean, float* rstd,
                                        float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !running_mean || !running_var || !mean || !rstd || !scale || !shift || C <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), C, eps, gamma,
                       beta, running_mean, running_var, mean, rstd, scale, shift);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <typename T>
static int bn_apply_impl(const T* x, const float* scale, const float* shift, const T* x2,
                         const float* scale2, const float* shift2, T* y, int64_t rows, int32_t C,
                         int32_t mode, int32_t relu, void* stream, uint8_t* signbits = nullptr) {
    if (!x || !scale || !shift || !y || rows <= 0 || C <= 0 || (C & 3)) return LOANS_EINVAL;
    if (mode < 0 || mode > 2 || (mode >= 1 && !x2) || (mode == 2 && (!scale2 || !shift2))) return LOANS_EINVAL;
    hipStream_t st = as_stream(stream);
    if (const int U = units_per_row<T>(C)) {
        const int64_t nunits = rows * U;
        const int g16 = slab_grid(nunits);
        const bool nt = slab_nt(nunits * 16);
#define LAUNCH_A16_(M, B_, N_) hipLaunchKernelGGL((bn_apply_u16_kernel<M, B_, N_, T>), dim3(g16), dim3(256), 0, st, x, scale, shift, x2, scale2, shift2, y, nunits, U, relu, signbits)
#define LAUNCH_A16(M, B_) do { if (nt) LAUNCH_A16_(M, B_, true); else LAUNCH_A16_(M, B_, false); } while (0)
        if (signbits) { if (mode == 0) LAUNCH_A16(0, true); else if (mode == 1) LAUNCH_A16(1, true); else LAUNCH_A16(2, true); }
        else { if (mode == 0) LAUNCH_A16(0, false); else if (mode == 1) LAUNCH_A16(1, false); else LAUNCH_A16(2, false); }
#undef LAUNCH_A16
#undef LAUNCH_A16_
        LOANS_LAUNCH_CHECK();
        return LOANS_OK;
    }
    const int64_t n4 = rows * (C / 4);
    const int grid = grid_for(n4, 256);
    if (mode == 0) hipLaunchKernelGGL((bn_apply_kernel<0, T>), dim3(grid), dim3(256), 0, st, x, scale, shift, x2, scale2, shift2, y, n4, C / 4, relu, signbits);
    if (mode == 1) hipLaunchKernelGGL((bn_apply_kernel<1, T>), dim3(grid), dim3(256), 0, st, x, scale, shift, x2, scale2, shift2, y, n4, C / 4, relu, signbits);
    if (mode == 2) hipLaunchKernelGGL((bn_apply_kernel<2, T>), dim3(grid), dim3(256), 0, st, x, scale, shift, x2, scale2, shift2, y, n4, C / 4, relu, signbits);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_bn_apply_f32(const float* x, const float* scale, const float* shift, const float* x2,
                                  const float* scale2, const float* shift2, float* y, int64_t rows, int32_t C,
                                  int32_t mode, int32_t relu, void* stream) {
    return bn_apply_impl<float>(x, scale, shift, x2, scale2, shift2, y, rows, C, mode, relu, stream);
}

extern "C" int loans_bn_apply_bits_f32(const float* x, const float* scale, const float* shift, const float* x2,
                                       const float* scale2, const float* shift2, float* y, uint8_t* signbits, int64_t rows,
                                       int32_t C, int32_t mode, int32_t relu, void* stream) {
    if (!signbits) return LOANS_EINVAL;
    return bn_apply_impl<float>(x, scale, shift, x2, scale2, shift2, y, rows, C, mode, relu, stream, signbits);
}

extern "C" int loans_bn_apply_bits_bf16(const void* x, const float* scale, const float* shift, const void* x2,
                                        const float* scale2, const float* shift2, void* y, uint8_t* signbits, int64_t rows,
                                        int32_t C, int32_t mode, int32_t relu, void* stream) {
    if (!signbits) return LOANS_EINVAL;
    return bn_apply_impl<__bf16>(static_cast<const __bf16*>(x), scale, shift, static_cast<const __bf16*>(x2), scale2, shift2,
                                 static_cast<__bf16*>(y), rows, C, mode, relu, stream, signbits);
}

extern "C" int loans_bn_apply_bf16(const void* x, const float* scale, const float* shift, const void* x2,
                                   const float* scale2, const float* shift2, void* y, int64_t rows, int32_t C,
                                   int32_t mode, int32_t relu, void* stream) {
    return bn_apply_impl<__bf16>(static_cast<const __bf16*>(x), scale, shift, static_cast<const __bf16*>(x2), scale2, shift2,
                                 static_cast<__bf16*>(y), rows, C, mode, relu, stream);
}

template <typename TO>
static int bn_relu_maxpool_impl(const TO* x, const float* scale, const float* shift, TO* y, uint8_t* idx,
                                int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream, TO* xsel = nullptr,
                                bool want_sel = false) {
    if (!x || !scale || !shift || !y || !idx || B <= 0 || H < 3 || W < 3 || C <= 0 || (C & 3)) return LOANS_EINVAL;
    if (want_sel && !xsel) return LOANS_EINVAL;
    if (OH != (H - 2) / 2 + 1 || OW != (W - 2) / 2 + 1) return LOANS_EINVAL;   // cover_all out size, k=3 s=2 p=0
    if ((int64_t)B * H >= ((int64_t)1 << 31) || (int64_t)W * (C / 4) >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    if (const int U = units_per_row<TO>(C)) {
        const int PL = 256 / U;
        const dim3 g16((OW + PL - 1) / PL, (unsigned)min((int64_t)B * OH, (int64_t)65535));
        if (want_sel)
            hipLaunchKernelGGL((bn_relu_maxpool_u16_kernel<TO, true>), g16, dim3(256), 0, as_stream(stream), x, scale, shift, y, idx,
                               xsel, B, H, W, U, OH, OW);
        else
            hipLaunchKernelGGL((bn_relu_maxpool_u16_kernel<TO, false>), g16, dim3(256), 0, as_stream(stream), x, scale, shift, y, idx,
                               xsel, B, H, W, U, OH, OW);
        LOANS_LAUNCH_CHECK();
        return LOANS_OK;
    }
    if (want_sel) return LOANS_EINVAL;          // (channel counts the 16-byte-unit kernel does not tile: the caller keeps the gather)
    const dim3 grid((OW * (C / 4) + 255) / 256, (unsigned)min((int64_t)B * OH, (int64_t)65535));
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<TO>, grid, dim3(256), 0, as_stream(stream), x, scale,
                       shift, y, idx, B, H, W, C / 4, OH, OW);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_bn_relu_maxpool_f32(const float* x, const float* scale, const float* shift, float* y, uint8_t* idx,
                                         int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream) {
    return bn_relu_maxpool_impl<float>(x, scale, shift, y, idx, B, H, W, C, OH, OW, stream);
}

extern "C" int loans_bn_relu_maxpool_bf16(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx,
                                          int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream) {
    return bn_relu_maxpool_impl<__bf16>(static_cast<const __bf16*>(x), scale, shift, static_cast<__bf16*>(y), idx, B, H, W, C, OH, OW, stream);
}

extern "C" int loans_bn_relu_maxpool_sel_f32(const float* x, const float* scale, const float* shift, float* y, uint8_t* idx, float* xsel,
                                             int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream) {
    return bn_relu_maxpool_impl<float>(x, scale, shift, y, idx, B, H, W, C, OH, OW, stream, xsel, true);
}

extern "C" int loans_bn_relu_maxpool_sel_bf16(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx, void* xsel,
                                              int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream) {
    return bn_relu_maxpool_impl<__bf16>(static_cast<const __bf16*>(x), scale, shift, static_cast<__bf16*>(y), idx, B, H, W, C, OH, OW,
                                        stream, static_cast<__bf16*>(xsel), true);
}

template <typename TG>
static int maxpool_relu_bwd_impl(const TG* gy, const uint8_t* idx, const TG* x, const float* scale,
                                 const float* shift, TG* gx, int32_t B, int32_t H, int32_t W, int32_t C,
                                 int32_t OH, int32_t OW, void* stream) {
    if (!gy || !idx || !x || !scale || !shift || !gx || B <= 0 || H < 3 || W < 3 || C <= 0 || (C & 3)) return LOANS_EINVAL;
    if (OH != (H - 2) / 2 + 1 || OW != (W - 2) / 2 + 1) return LOANS_EINVAL;
    if ((int64_t)B * H >= ((int64_t)1 << 31) || (int64_t)W * (C / 4) >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    const dim3 grid((W * (C / 4) + 255) / 256, (unsigned)min((int64_t)B * H, (int64_t)65535));
    hipLaunchKernelGGL(maxpool_relu_bwd_kernel<TG>, grid, dim3(256), 0, as_stream(stream), gy, idx, x,
                       scale, shift, gx, B, H, W, C / 4, OH, OW);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_maxpool_relu_bwd_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale,
                                          const float* shift, float* gx, int32_t B, int32_t H, int32_t W, int32_t C,
                                          int32_t OH, int32_t OW, void* stream) {
    return maxpool_relu_bwd_impl<float>(gy, idx, x, scale, shift, gx, B, H, W, C, OH, OW, stream);
}

extern "C" int loans_maxpool_relu_bwd_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale,
                                           const float* shift, void* gx, int32_t B, int32_t H, int32_t W, int32_t C,
                                           int32_t OH, int32_t OW, void* stream) {
    return maxpool_relu_bwd_impl<__bf16>(static_cast<const __bf16*>(gy), idx, static_cast<const __bf16*>(x), scale, shift,
                                         static_cast<__bf16*>(gx), B, H, W, C, OH, OW, stream);
}

template <typename T>
static int pool_bn_bwd_reduce_impl(const T* gy, const uint8_t* idx, const T* x, const float* scale, const float* shift,
                                   const float* mean, const float* rstd, double* sums, int32_t B, int32_t H, int32_t W,
                                   int32_t C, int32_t OH, int32_t OW, void* stream, int replicas = 1) {
    if (!gy || !idx || !x || !scale || !shift || !mean || !rstd || !sums || B <= 0 || H < 3 || W < 3) return LOANS_EINVAL;
    if (replicas < 1 || replicas > 32) return LOANS_EINVAL;
    if (!reduce_channels_ok(C)) return LOANS_EINVAL;
    if (OH != (H - 2) / 2 + 1 || OW != (W - 2) / 2 + 1) return LOANS_EINVAL;
    const int64_t rows = (int64_t)B * OH * OW;
    if (rows >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    int rpb, c4b, slabs;
    int grid = reduce_geometry(rows, C, &rpb, &c4b, &slabs);
    if (replicas > 1) {             // replicated accumulators: four times the blocks (the closing atomics no longer collide)
        const int RL = 256 / c4b;
        int64_t r4 = ((rows + 4095) / 4096 + RL - 1) / RL * RL;
        if (r4 < 8 * RL) r4 = 8 * RL;
        rpb = (int)r4;
        grid = (int)((rows + rpb - 1) / rpb);
    }
    hipLaunchKernelGGL(pool_bn_bwd_reduce_kernel<T>, dim3(grid, slabs), dim3(256), 0, as_stream(stream), gy, idx, x, scale,
                       shift, mean, rstd, sums, (int)rows, H, W, OH, OW, c4b, C / 4, rpb, replicas);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <typename T>
static int pool_bn_bwd_apply_impl(const T* gy, const uint8_t* idx, const T* x, const float* scale, const float* shift,
                                  const float* k1, const float* k2, const float* k3, T* gx, float* gxsum, int32_t B,
                                  int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream, int gx_reps = 1) {
    if (!gy || !idx || !x || !scale || !shift || !k1 || !k2 || !k3 || !gx || B <= 0 || H < 3 || W < 3 || C <= 0 || (C & 3))
        return LOANS_EINVAL;
    if (gxsum && (C / 4 > 256 || 256 % (C / 4))) return LOANS_EINVAL;
    if (OH != (H - 2) / 2 + 1 || OW != (W - 2) / 2 + 1) return LOANS_EINVAL;
    const int PH = (H + 1) / 2, PW = (W + 1) / 2;
    if ((int64_t)B * PH >= ((int64_t)1 << 31) || (int64_t)PW * (C / 4) >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    // <= ~2048 blocks (8 per CU; the loops are grid-stride): every block ends with C float atomics on the same two
    // cache lines of gxsum, and 32 k blocks of them cost more than the pass itself
    if (gx_reps < 1 || gx_reps > 64) return LOANS_EINVAL;
    if (gx_reps > 1) {
        if (!gxsum || C / 4 > 256 || 256 % (C / 4)) return LOANS_EINVAL;
        const int PL = 256 / (C / 4), gxb = (PW + PL - 1) / PL;
        const dim3 g4(gxb, (unsigned)min((int64_t)B * PH, (int64_t)max(1, 4096 / gxb)));
        hipLaunchKernelGGL(pool_bn_bwd_apply_v4_kernel<T>, g4, dim3(256), 0, as_stream(stream), gy, idx, x, scale, shift, k1, k2, k3,
                           gx, gxsum, B, H, W, C / 4, OH, OW, gx_reps);
        LOANS_LAUNCH_CHECK();
        return LOANS_OK;
    }
    const int gx_blocks = (PW * (C / 4) + 255) / 256;
    const dim3 grid(gx_blocks, (unsigned)min((int64_t)B * PH, (int64_t)max(1, 2048 / gx_blocks)));
    hipLaunchKernelGGL(pool_bn_bwd_apply_kernel<T>, grid, dim3(256), 0, as_stream(stream), gy, idx, x, scale, shift, k1, k2,
                       k3, gx, gxsum, B, H, W, C / 4, OH, OW);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_pool_bn_bwd_reduce_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale,
                                            const float* shift, const float* mean, const float* rstd, double* sums,
                                            int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream) {
    return pool_bn_bwd_reduce_impl<float>(gy, idx, x, scale, shift, mean, rstd, sums, B, H, W, C, OH, OW, stream);
}

extern "C" int loans_pool_bn_bwd_reduce_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale,
                                             const float* shift, const float* mean, const float* rstd, double* sums,
                                             int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream) {
    return pool_bn_bwd_reduce_impl<__bf16>(static_cast<const __bf16*>(gy), idx, static_cast<const __bf16*>(x), scale, shift,
                                           mean, rstd, sums, B, H, W, C, OH, OW, stream);
}

extern "C" int loans_pool_bn_bwd_apply_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale,
                                           const float* shift, const float* k1, const float* k2, const float* k3, float* gx,
                                           float* gxsum, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                                           void* stream) {
    return pool_bn_bwd_apply_impl<float>(gy, idx, x, scale, shift, k1, k2, k3, gx, gxsum, B, H, W, C, OH, OW, stream);
}

extern "C" int loans_pool_bn_bwd_apply_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale,
                                            const float* shift, const float* k1, const float* k2, const float* k3, void* gx,
                                            float* gxsum, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                                            void* stream) {
    return pool_bn_bwd_apply_impl<__bf16>(static_cast<const __bf16*>(gy), idx, static_cast<const __bf16*>(x), scale, shift,
                                          k1, k2, k3, static_cast<__bf16*>(gx), gxsum, B, H, W, C, OH, OW, stream);
}

template <typename T>
static int bn_bwd_reduce_impl(const T* gy, const T* mask, const T* x, const float* mean,
                              const float* rstd, const T* x2, const float* mean2, const float* rstd2,
                              double* sums, int64_t rows, int32_t C, void* stream,
                              const float* scale = nullptr, const float* shift = nullptr, bool bits = false, int replicas = 1) {
    if (!gy || !x || !mean || !rstd || !sums || rows <= 0) return LOANS_EINVAL;
    if (!reduce_channels_ok(C)) return LOANS_EINVAL;
    if (x2 && (!mean2 || !rstd2)) return LOANS_EINVAL;
    if (scale && (!shift || mask || x2)) return LOANS_EINVAL;
    hipStream_t st = as_stream(stream);
    if (bits && !mask) return LOANS_EINVAL;
    if (replicas < 1 || replicas > 32) return LOANS_EINVAL;
    const int U = units_per_row<T>(C);
    if (replicas > 1 && !U) return LOANS_EINVAL;            // replicated accumulators: the 16-byte-unit kernel only
    if (U) {
        const int UB = U <= 16 ? U : 16, slabs = U / UB;    // >= 256-byte row segments per block
        const int RL = 256 / UB;
        const int max_blocks = (replicas > 1 ? 4096 : 1024) / slabs;  // un-replicated sums: every block adds to the same addresses
        int64_t rpb16 = ((rows + max_blocks - 1) / max_blocks + RL - 1) / RL * RL;
        // >= 32 rows per thread (a block ends with 2, dual: 4, fp64 atomics per channel) -- unless that leaves most of the machine
        // idle: the deep stages' small maps (res5 .. res7 of a 512 px step: 16 384 .. 1 024 rows) ran on 2 .. 32 blocks per channel
        // slab, 40-70 us for 5-20 us of traffic on the main stream's critical path; down to 4 rows per thread until 512 blocks are out
        int64_t rpt = 32;
        while (rpt > 4 && ((rows + rpt * RL - 1) / (rpt * RL)) * slabs < 512) rpt >>= 1;
        if (rpb16 < rpt * RL) rpb16 = rpt * RL;
        const int g16 = (int)((rows + rpb16 - 1) / rpb16);
#define LAUNCH_R16(D, M) \
    hipLaunchKernelGGL((bn_bwd_reduce_u16_kernel<D, M, T>), dim3(g16, slabs), dim3(256), 0, st, gy, mask, x, mean, rstd, x2, mean2, rstd2, sums, rows, U, (int)rpb16, scale, shift, replicas, UB)
        if (x2) { if (bits) LAUNCH_R16(true, 3); else if (mask) LAUNCH_R16(true, 1); else LAUNCH_R16(true, 0); }
        else if (scale) LAUNCH_R16(false, 2);
        else { if (bits) LAUNCH_R16(false, 3); else if (mask) LAUNCH_R16(false, 1); else LAUNCH_R16(false, 0); }
#undef LAUNCH_R16
        LOANS_LAUNCH_CHECK();
        return LOANS_OK;
    }
    int rpb, c4b, slabs;
    const int grid = reduce_geometry(rows, C, &rpb, &c4b, &slabs);
#define LAUNCH_RED(D, M) \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<D, M, T>), dim3(grid, slabs), dim3(256), 0, st, gy, mask, x, mean, rstd, x2, mean2, rstd2, sums, rows, c4b, C / 4, rpb, scale, shift)
    if (bits && !mask) return LOANS_EINVAL;
    if (x2) { if (bits) LAUNCH_RED(true, 3); else if (mask) LAUNCH_RED(true, 1); else LAUNCH_RED(true, 0); }
    else if (scale) LAUNCH_RED(false, 2);
    else { if (bits) LAUNCH_RED(false, 3); else if (mask) LAUNCH_RED(false, 1); else LAUNCH_RED(false, 0); }
#undef LAUNCH_RED
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_bn_bwd_reduce_f32(const float* gy, const float* mask, const float* x, const float* mean,
                                       const float* rstd, const float* x2, const float* mean2, const float* rstd2,
                                       double* sums, int64_t rows, int32_t C, void* stream) {
    return bn_bwd_reduce_impl<float>(gy, mask, x, mean, rstd, x2, mean2, rstd2, sums, rows, C, stream);
}

extern "C" int loans_bn_bwd_reduce_bf16(const void* gy, const void* mask, const void* x, const float* mean,
                                        const float* rstd, const void* x2, const float* mean2, const float* rstd2,
                                        double* sums, int64_t rows, int32_t C, void* stream) {
    return bn_bwd_reduce_impl<__bf16>(static_cast<const __bf16*>(gy), static_cast<const __bf16*>(mask), static_cast<const __bf16*>(x),
                                      mean, rstd, static_cast<const __bf16*>(x2), mean2, rstd2, sums, rows, C, stream);
}

extern "C" int loans_igemm_finalize_f32(float* out, const float* bias, double* stats, const float* ref, const float* addend,
                                        int32_t flags, int64_t rows, int32_t C, void* stream) {
    if (!out || rows <= 0 || !reduce_channels_ok(C)) return LOANS_EINVAL;
    if ((flags & LOANS_F_BIAS) && !bias) return LOANS_EINVAL;
    if ((flags & LOANS_F_STATS) && !stats) return LOANS_EINVAL;
    if ((flags & (LOANS_F_MASK | LOANS_F_ADDEND_MASK)) && !ref) return LOANS_EINVAL;
    if ((flags & LOANS_F_ADDEND_MASK) && !(flags & LOANS_F_ADDEND)) return LOANS_EINVAL;
    if ((flags & LOANS_F_ADDEND) && !addend) return LOANS_EINVAL;
    if (flags & ~(LOANS_F_BIAS | LOANS_F_STATS | LOANS_F_MASK | LOANS_F_ADDEND | LOANS_F_ADDEND_MASK)) return LOANS_EINVAL;
    int rpb, c4b, slabs;
    const int grid = reduce_geometry(rows, C, &rpb, &c4b, &slabs);
    hipLaunchKernelGGL(igemm_finalize_kernel, dim3(grid, slabs), dim3(256), 0, as_stream(stream), out, bias, stats, ref, addend,
                       flags, rows, c4b, C / 4, rpb);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_bn_bwd_coeffs_f32(const double* sums, int32_t C, int64_t count, const float* gamma,
                                       const float* mean, const float* rstd, float* ggamma, float* gbeta, float* k1,
                                       float* k2, float* k3, void* stream) {
    if (!sums || !gamma || !mean || !rstd || !ggamma || !gbeta || !k1 || !k2 || !k3 || C <= 0 || count <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), sums, C,
                       1.0 / (double)count, gamma, mean, rstd, ggamma, gbeta, k1, k2, k3);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_bn_bwd_coeffs_rep_f32(const double* sums, int32_t replicas, int32_t rep_stride, int32_t centred, int32_t C,
                                           int64_t count, const float* gamma, const float* mean, const float* rstd, float* ggamma,
                                           float* gbeta, float* k1, float* k2, float* k3, void* stream) {
    if (!sums || !gamma || !mean || !rstd || !ggamma || !gbeta || !k1 || !k2 || !k3 || C <= 0 || count <= 0 || replicas <= 0 ||
        replicas > 32 || rep_stride < 2 * C)
        return LOANS_EINVAL;
    hipLaunchKernelGGL(bn_bwd_coeffs_rep_kernel, dim3((C + 7) / 8), dim3(256), 0, as_stream(stream), sums, replicas, rep_stride,
                       centred, C, 1.0 / (double)count, gamma, mean, rstd, ggamma, gbeta, k1, k2, k3);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <typename T>
static int bn_bwd_apply_impl(const T* gy, const T* mask, const T* x, const float* k1,
                             const float* k2, const float* k3, T* gx, const T* x2, const float* k1b,
                             const float* k2b, const float* k3b, T* gx2, int64_t rows, int32_t C,
                             void* stream, const float* scale = nullptr, const float* shift = nullptr, bool bits = false) {
    if (!gy || !x || !k1 || !k2 || !k3 || !gx || rows <= 0 || C <= 0 || (C & 3)) return LOANS_EINVAL;
    if (bits && !mask) return LOANS_EINVAL;
    if (x2 && (!k1b || !k2b || !k3b || !gx2)) return LOANS_EINVAL;
    if (scale && (!shift || mask || x2)) return LOANS_EINVAL;
    hipStream_t st = as_stream(stream);
    if (const int U = units_per_row<T>(C)) {
        const int64_t nunits = rows * U;
        const int g16 = slab_grid(nunits);
        const bool nt = slab_nt(nunits * 16);
#define LAUNCH_P16_(D, M, N_) \
    hipLaunchKernelGGL((bn_bwd_apply_u16_kernel<D, M, N_, T>), dim3(g16), dim3(256), 0, st, gy, mask, x, k1, k2, k3, gx, x2, k1b, k2b, k3b, gx2, nunits, U, scale, shift)
#define LAUNCH_P16(D, M) do { if (nt) LAUNCH_P16_(D, M, true); else LAUNCH_P16_(D, M, false); } while (0)
        if (x2) { if (bits) LAUNCH_P16(true, 3); else if (mask) LAUNCH_P16(true, 1); else LAUNCH_P16(true, 0); }
        else if (scale) LAUNCH_P16(false, 2);
        else { if (bits) LAUNCH_P16(false, 3); else if (mask) LAUNCH_P16(false, 1); else LAUNCH_P16(false, 0); }
#undef LAUNCH_P16
#undef LAUNCH_P16_
        LOANS_LAUNCH_CHECK();
        return LOANS_OK;
    }
    const int64_t n4 = rows * (C / 4);
    const int grid = grid_for(n4, 256);
#define LAUNCH_APP(D, M) \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<D, M, T>), dim3(grid), dim3(256), 0, st, gy, mask, x, k1, k2, k3, gx, x2, k1b, k2b, k3b, gx2, n4, C / 4, scale, shift)
    if (x2) { if (bits) LAUNCH_APP(true, 3); else if (mask) LAUNCH_APP(true, 1); else LAUNCH_APP(true, 0); }
    else if (scale) LAUNCH_APP(false, 2);
    else { if (bits) LAUNCH_APP(false, 3); else if (mask) LAUNCH_APP(false, 1); else LAUNCH_APP(false, 0); }
#undef LAUNCH_APP
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_bn_bwd_apply_f32(const float* gy, const float* mask, const float* x, const float* k1,
                                      const float* k2, const float* k3, float* gx, const float* x2, const float* k1b,
                                      const float* k2b, const float* k3b, float* gx2, int64_t rows, int32_t C,
                                      void* stream) {
    return bn_bwd_apply_impl<float>(gy, mask, x, k1, k2, k3, gx, x2, k1b, k2b, k3b, gx2, rows, C, stream);
}

extern "C" int loans_bn_bwd_apply_bf16(const void* gy, const void* mask, const void* x, const float* k1,
                                       const float* k2, const float* k3, void* gx, const void* x2, const float* k1b,
                                       const float* k2b, const float* k3b, void* gx2, int64_t rows, int32_t C,
                                       void* stream) {
    return bn_bwd_apply_impl<__bf16>(static_cast<const __bf16*>(gy), static_cast<const __bf16*>(mask), static_cast<const __bf16*>(x),
                                     k1, k2, k3, static_cast<__bf16*>(gx), static_cast<const __bf16*>(x2), k1b, k2b, k3b,
                                     static_cast<__bf16*>(gx2), rows, C, stream);
}

// g = gy * bit: the mask comes from the sign bits loans_bn_apply_bits_* wrote (one byte per four channels)
extern "C" int loans_bn_bwd_reduce_bits_f32(const float* gy, const uint8_t* signbits, const float* x, const float* mean,
                                            const float* rstd, const float* x2, const float* mean2, const float* rstd2,
                                            double* sums, int64_t rows, int32_t C, void* stream) {
    return bn_bwd_reduce_impl<float>(gy, reinterpret_cast<const float*>(signbits), x, mean, rstd, x2, mean2, rstd2, sums, rows, C,
                                     stream, nullptr, nullptr, true);
}

extern "C" int loans_bn_bwd_reduce_bits_bf16(const void* gy, const uint8_t* signbits, const void* x, const float* mean,
                                             const float* rstd, const void* x2, const float* mean2, const float* rstd2,
                                             double* sums, int64_t rows, int32_t C, void* stream) {
    return bn_bwd_reduce_impl<__bf16>(static_cast<const __bf16*>(gy), reinterpret_cast<const __bf16*>(signbits),
                                      static_cast<const __bf16*>(x), mean, rstd, static_cast<const __bf16*>(x2), mean2, rstd2,
                                      sums, rows, C, stream, nullptr, nullptr, true);
}

extern "C" int loans_bn_bwd_apply_bits_f32(const float* gy, const uint8_t* signbits, const float* x, const float* k1,
                                           const float* k2, const float* k3, float* gx, const float* x2, const float* k1b,
                                           const float* k2b, const float* k3b, float* gx2, int64_t rows, int32_t C,
                                           void* stream) {
    return bn_bwd_apply_impl<float>(gy, reinterpret_cast<const float*>(signbits), x, k1, k2, k3, gx, x2, k1b, k2b, k3b, gx2, rows,
                                    C, stream, nullptr, nullptr, true);
}

extern "C" int loans_bn_bwd_apply_bits_bf16(const void* gy, const uint8_t* signbits, const void* x, const float* k1,
                                            const float* k2, const float* k3, void* gx, const void* x2, const float* k1b,
                                            const float* k2b, const float* k3b, void* gx2, int64_t rows, int32_t C,
                                            void* stream) {
    return bn_bwd_apply_impl<__bf16>(static_cast<const __bf16*>(gy), reinterpret_cast<const __bf16*>(signbits),
                                     static_cast<const __bf16*>(x), k1, k2, k3, static_cast<__bf16*>(gx),
                                     static_cast<const __bf16*>(x2), k1b, k2b, k3b, static_cast<__bf16*>(gx2), rows, C, stream,
                                     nullptr, nullptr, true);
}

// g = gy * (x*scale+shift > 0): the ReLU mask of the BN being differentiated, recomputed from x (no mask tensor)
extern "C" int loans_bn_bwd_reduce_xmask_f32(const float* gy, const float* x, const float* scale, const float* shift,
                                             const float* mean, const float* rstd, double* sums, int64_t rows, int32_t C,
                                             void* stream) {
    if (!scale || !shift) return LOANS_EINVAL;
    return bn_bwd_reduce_impl<float>(gy, nullptr, x, mean, rstd, nullptr, nullptr, nullptr, sums, rows, C, stream, scale, shift);
}

extern "C" int loans_bn_bwd_reduce_xmask_bf16(const void* gy, const void* x, const float* scale, const float* shift,
                                              const float* mean, const float* rstd, double* sums, int64_t rows, int32_t C,
                                              void* stream) {
    if (!scale || !shift) return LOANS_EINVAL;
    return bn_bwd_reduce_impl<__bf16>(static_cast<const __bf16*>(gy), nullptr, static_cast<const __bf16*>(x), mean, rstd, nullptr,
                                      nullptr, nullptr, sums, rows, C, stream, scale, shift);
}

extern "C" int loans_bn_bwd_apply_xmask_f32(const float* gy, const float* x, const float* scale, const float* shift,
                                            const float* k1, const float* k2, const float* k3, float* gx, int64_t rows,
                                            int32_t C, void* stream) {
    if (!scale || !shift) return LOANS_EINVAL;
    return bn_bwd_apply_impl<float>(gy, nullptr, x, k1, k2, k3, gx, nullptr, nullptr, nullptr, nullptr, nullptr, rows, C, stream,
                                    scale, shift);
}

extern "C" int loans_bn_bwd_apply_xmask_bf16(const void* gy, const void* x, const float* scale, const float* shift,
                                             const float* k1, const float* k2, const float* k3, void* gx, int64_t rows,
                                             int32_t C, void* stream) {
    if (!scale || !shift) return LOANS_EINVAL;
    return bn_bwd_apply_impl<__bf16>(static_cast<const __bf16*>(gy), nullptr, static_cast<const __bf16*>(x), k1, k2, k3,
                                     static_cast<__bf16*>(gx), nullptr, nullptr, nullptr, nullptr, nullptr, rows, C, stream,
                                     scale, shift);
}

extern "C" int loans_colsum_f32(const float* x, float* out, int64_t rows, int32_t C, void* stream) {
    if (!x || !out || rows <= 0 || !reduce_channels_ok(C)) return LOANS_EINVAL;
    int rpb, c4b, slabs;
    const int grid = reduce_geometry(rows, C, &rpb, &c4b, &slabs);
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(grid, slabs), dim3(256), 0, as_stream(stream), x, out, rows, c4b, C / 4, rpb);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_colsum_bf16(const void* x, float* out, int64_t rows, int32_t C, void* stream) {
    if (!x || !out || rows <= 0 || !reduce_channels_ok(C)) return LOANS_EINVAL;
    int rpb, c4b, slabs;
    const int grid = reduce_geometry(rows, C, &rpb, &c4b, &slabs);
    hipLaunchKernelGGL(colsum_kernel<__bf16>, dim3(grid, slabs), dim3(256), 0, as_stream(stream), static_cast<const __bf16*>(x), out, rows, c4b, C / 4, rpb);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

// ---- BN-backward reduction into REPLICATED accumulators (round 3) ---------------------------------------------------------------
// One entry per storage type for all mask forms: mask_kind 0 = none, 1 = g * (mask tensor > 0), 2 = g * (x scale + shift > 0) (the
// BN's own ReLU; `scale` / `shift`), 3 = sign bits (one byte per four channels).  `sums`: fp64 [replicas][2][C] (dual, x2 given:
// [replicas][4][C] = [sum g | sum g xhat | sum g | sum g xhat2]), zeroed by the caller, summed by loans_bn_bwd_coeffs_rep_f32
// (centred = 0, rep_stride = 2 C or 4 C).  Needs a channel count the 16-byte-unit kernel tiles (C / V a divisor of 256, V = 4
// fp32 / 8 bf16 channels).
template <typename T>
static int bn_bwd_reduce_rep(const void* gy, const void* mask, int mask_kind, const void* x, const float* mean, const float* rstd,
                             const void* x2, const float* mean2, const float* rstd2, const float* scale, const float* shift,
                             double* sums, int replicas, int64_t rows, int C, void* stream) {
    if (mask_kind < 0 || mask_kind > 3 || replicas < 2) return LOANS_EINVAL;
    if ((mask_kind == 1 || mask_kind == 3) != (mask != nullptr)) return LOANS_EINVAL;
    if ((mask_kind == 2) != (scale != nullptr)) return LOANS_EINVAL;
    return bn_bwd_reduce_impl<T>(static_cast<const T*>(gy), static_cast<const T*>(mask), static_cast<const T*>(x), mean, rstd,
                                 static_cast<const T*>(x2), mean2, rstd2, sums, rows, C, stream, scale, shift, mask_kind == 3,
                                 replicas);
}

extern "C" int loans_bn_bwd_reduce_rep_f32(const float* gy, const void* mask, int32_t mask_kind, const float* x, const float* mean,
                                           const float* rstd, const float* x2, const float* mean2, const float* rstd2,
                                           const float* scale, const float* shift, double* sums, int32_t replicas, int64_t rows,
                                           int32_t C, void* stream) {
    return bn_bwd_reduce_rep<float>(gy, mask, mask_kind, x, mean, rstd, x2, mean2, rstd2, scale, shift, sums, replicas, rows, C, stream);
}

extern "C" int loans_bn_bwd_reduce_rep_bf16(const void* gy, const void* mask, int32_t mask_kind, const void* x, const float* mean,
                                            const float* rstd, const void* x2, const float* mean2, const float* rstd2,
                                            const float* scale, const float* shift, double* sums, int32_t replicas, int64_t rows,
                                            int32_t C, void* stream) {
    return bn_bwd_reduce_rep<__bf16>(gy, mask, mask_kind, x, mean, rstd, x2, mean2, rstd2, scale, shift, sums, replicas, rows, C, stream);
}

// the fused pool backward's apply pass with the bias-gradient sums going into `replicas` accumulators gxsum_rep[replicas][C]
// (zeroed by the caller, folded into the bias gradient by loans_fold_replicas_f32); C / 4 must divide 256
extern "C" int loans_pool_bn_bwd_apply_rep_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale,
                                               const float* shift, const float* k1, const float* k2, const float* k3, float* gx,
                                               float* gxsum_rep, int32_t replicas, int32_t B, int32_t H, int32_t W, int32_t C,
                                               int32_t OH, int32_t OW, void* stream) {
    if (replicas < 2) return LOANS_EINVAL;
    return pool_bn_bwd_apply_impl<float>(gy, idx, x, scale, shift, k1, k2, k3, gx, gxsum_rep, B, H, W, C, OH, OW, stream, replicas);
}

extern "C" int loans_pool_bn_bwd_apply_rep_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale,
                                                const float* shift, const float* k1, const float* k2, const float* k3, void* gx,
                                                float* gxsum_rep, int32_t replicas, int32_t B, int32_t H, int32_t W, int32_t C,
                                                int32_t OH, int32_t OW, void* stream) {
    if (replicas < 2) return LOANS_EINVAL;
    return pool_bn_bwd_apply_impl<__bf16>(static_cast<const __bf16*>(gy), idx, static_cast<const __bf16*>(x), scale, shift, k1, k2, k3,
                                          static_cast<__bf16*>(gx), gxsum_rep, B, H, W, C, OH, OW, stream, replicas);
}

extern "C" int loans_fold_replicas_f32(const float* src, float* dst, int32_t replicas, int32_t C, void* stream) {
    if (!src || !dst || replicas <= 0 || C <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(fold_replicas_kernel, dim3((C + 7) / 8), dim3(256), 0, as_stream(stream), src, dst, replicas, C);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

// the stem tail's reduction into replicated accumulators [replicas][2][C] (see loans_bn_bwd_reduce_rep_*)
extern "C" int loans_pool_bn_bwd_reduce_rep_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale,
                                                const float* shift, const float* mean, const float* rstd, double* sums,
                                                int32_t replicas, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                                                void* stream) {
    if (replicas < 2) return LOANS_EINVAL;
    return pool_bn_bwd_reduce_impl<float>(gy, idx, x, scale, shift, mean, rstd, sums, B, H, W, C, OH, OW, stream, replicas);
}

extern "C" int loans_pool_bn_bwd_reduce_rep_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale,
                                                 const float* shift, const float* mean, const float* rstd, double* sums,
                                                 int32_t replicas, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                                                 void* stream) {
    if (replicas < 2) return LOANS_EINVAL;
    return pool_bn_bwd_reduce_impl<__bf16>(static_cast<const __bf16*>(gy), idx, static_cast<const __bf16*>(x), scale, shift, mean, rstd,
                                           sums, B, H, W, C, OH, OW, stream, replicas);
}
