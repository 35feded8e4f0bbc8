// Gradient w.r.t. the 4-channel (RGB) crops through BOTH convolutions of the assessor's first block that read them
// (common/net.py:15,17,22-25: DownResBlock1's c0 3x3/1 and cs 4x4/2), in ONE launch that reads each incoming gradient once.
//
// N = 3 output channels would waste 29/32 of an MFMA tile if the taps stayed in K.  Here the taps go into N instead:
//     T[p][(r, s, ci)] = sum_co gy[p][co] * w[co][r][s][ci]            a plain GEMM, M = gradient pixels, K = C, N = k*k*3
//     gx[iy][ix][ci]   = sum_{(r, s)} T[((iy + pad - r) / stride, (ix + pad - s) / stride)][(r, s, ci)]      (col2im)
// N = 27 (3x3) fills one 32-wide MFMA tile, N = 48 (4x4) one and a half.  A block owns a 16 x 16 tile of gx: it computes T
// for the gradient pixels that tile needs (18 x 18 for the 3x3/1 conv, 10 x 10 for the 4x4/2 one: the halo is recomputed,
// never exchanged), keeps T in LDS, and its 256 threads gather one output pixel each -- no atomics, gx written once.
//   A operand: straight from global memory -- lane (r, h) of a 32-row block reads channels 8g+4h .. +3 of its pixel as one
//     16-byte load (8 bytes for a bf16 gradient); the 16 loads of a row block are issued back to back (fragment trick of
//     igemm.hip: MFMA step j contracts k = 8g+j with 8g+4+j on both operands);
//   B operand: the forward OHWI weights gathered into registers once per (wave, column block);
//   v_mfma_f32_32x32x2_f32, fp32 accumulate; out-of-image gradient pixels are zero rows of T.
// Bound: HBM (each gradient tensor once: 0.94 GB at B = 256 of 75 x 75 crops); replaces five launches of the VALU
// kernel in smalln.hip (2.8 ms per step) on this path -- that kernel stays for geometries this one does not cover.
#include "common.h"

namespace {

constexpr int TILE = 16;             // output tile edge
constexpr int MAX_UNITS = 32;

struct CropConv {
    const void* gy;                  // [B][gH][gW][C], float or bf16
    const float* w;                  // forward weights OHWI [C][k][k][4]
    int k, stride, pad, gH, gW;
    int ncols, ncb;                  // k*k*3, ceil(ncols / 32)
    int RW, RH, rows_pad;            // region of gradient pixels per tile (upper bound), rows padded to 32
    int ld;                          // floats per T row (odd)
    int t_off;                       // float offset of this problem's T in LDS
};

struct CropArgs {
    CropConv c[2];
    int nconv;
    float* out;
    const float* addend;
    int B, H, W, C, tiles_y, tiles_x;
    int nunits;
    unsigned char unit_conv[MAX_UNITS], unit_rb[MAX_UNITS], unit_cb[MAX_UNITS];
};

__device__ __forceinline__ int floordiv(int a, int b) {       // b > 0
    int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

template <typename TG> struct ld4g;
template <> struct ld4g<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
};
template <> struct ld4g<__bf16> {
    static __device__ __forceinline__ f32x4 ld(const __bf16* p) {
        return __builtin_convertvector(*reinterpret_cast<const loans_bf16x4*>(p), f32x4);
    }
};

template <typename TG>
__global__ __launch_bounds__(256, 2) void crop_dgrad_kernel(const CropArgs a) {
    extern __shared__ __attribute__((aligned(16))) float T[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int blk = blockIdx.x;
    const int tx = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty = blk % a.tiles_y;
    const int b = blk / a.tiles_y;
    const int oy0 = ty * TILE, ox0 = tx * TILE;
    const int G = a.C >> 3;                       // 8-deep k groups

    for (int u = wave; u < a.nunits; u += 4) {
        const CropConv& c = a.c[a.unit_conv[u]];
        const int rb = a.unit_rb[u], cb = a.unit_cb[u];
        const int ry0 = floordiv(oy0 + c.pad - (c.k - 1), c.stride);
        const int rx0 = floordiv(ox0 + c.pad - (c.k - 1), c.stride);
        // ---- B fragments: column n = (tap, ci) of this column block, k = 8g + 4h + j ----
        const int n = cb * 32 + r;
        const bool ncol = n < c.ncols;
        const int tap = ncol ? n / 3 : 0, ci = ncol ? n - tap * 3 : 0;
        const int kk4 = c.k * c.k * 4;
        const float* wp = c.w + tap * 4 + ci + (int64_t)(4 * h) * kk4;
        // ---- A rows: region pixel q of this row block ----
        const int q = rb * 32 + r;
        const int qy = q / c.RW, qx = q - qy * c.RW;
        const int y = ry0 + qy, x = rx0 + qx;
        const bool live = qy < c.RH && (unsigned)y < (unsigned)c.gH && (unsigned)x < (unsigned)c.gW;
        const TG* ap = static_cast<const TG*>(c.gy) + ((int64_t)(b * c.gH + (live ? y : 0)) * c.gW + (live ? x : 0)) * a.C + 4 * h;
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int g0 = 0; g0 < G; g0 += 8) {      // 8 groups (64 channels) per pass: 8 A loads + 32 B loads in flight
            f32x4 av[8], bv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int g = g0 + i;
                const bool gok = g < G;
                av[i] = (live && gok) ? ld4g<TG>::ld(ap + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (ncol && gok) {
                    const float* wg = wp + (int64_t)(8 * g) * kk4;
                    bv[i] = f32x4{wg[0], wg[kk4], wg[2 * kk4], wg[3 * kk4]};
                } else {
                    bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[i].w, acc, 0, 0, 0);
            }
        }
        // ---- D layout of the 32x32 MFMA: acc[v] = D[(v / 4) * 8 + h * 4 + v % 4][lane % 32] ----
        if (ncol) {
            float* tp = T + c.t_off + n;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = rb * 32 + (v >> 2) * 8 + h * 4 + (v & 3);
                tp[row * c.ld] = acc[v];
            }
        }
    }
    __syncthreads();

    // ---- col2im: one output pixel per thread ----
    const int py = tid >> 4, px = tid & 15;
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy >= a.H || ox >= a.W) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int p = 0; p < a.nconv; ++p) {
        const CropConv& c = a.c[p];
        const int ry0 = floordiv(oy0 + c.pad - (c.k - 1), c.stride);
        const int rx0 = floordiv(ox0 + c.pad - (c.k - 1), c.stride);
        const float* tb = T + c.t_off;
        for (int kr = 0; kr < c.k; ++kr) {
            const int ny = oy + c.pad - kr;
            if (ny < 0 || ny % c.stride) continue;
            const int qy = ny / c.stride - ry0;             // in [0, RH): the region was sized for it
            for (int ks = 0; ks < c.k; ++ks) {
                const int nx = ox + c.pad - ks;
                if (nx < 0 || nx % c.stride) continue;
                const int qx = nx / c.stride - rx0;
                const float* t = tb + (qy * c.RW + qx) * c.ld + (kr * c.k + ks) * 3;
                s0 += t[0]; s1 += t[1]; s2 += t[2];
            }
        }
    }
    const int64_t off = ((int64_t)(b * a.H + oy) * a.W + ox) * 4;
    f32x4 o = {s0, s1, s2, 0.f};
    if (a.addend) o += *reinterpret_cast<const f32x4*>(a.addend + off);
    *reinterpret_cast<f32x4*>(a.out + off) = o;
}

int setup_conv(CropConv& c, const void* gy, const float* w, const loans_small_conv* s, int H, int W, int& t_off) {
    if (!gy || !w || !s) return LOANS_EINVAL;
    if (s->k < 1 || s->k > 4 || s->stride < 1 || s->stride > 2 || s->pad < 0 || s->pad >= s->k) return LOANS_EINVAL;
    if (s->outH != (H + 2 * s->pad - s->k) / s->stride + 1 || s->outW != (W + 2 * s->pad - s->k) / s->stride + 1) return LOANS_EINVAL;
    c.gy = gy; c.w = w; c.k = s->k; c.stride = s->stride; c.pad = s->pad; c.gH = s->outH; c.gW = s->outW;
    c.ncols = s->k * s->k * 3;
    c.ncb = (c.ncols + 31) / 32;
    // gradient rows a 16-row output tile can touch: floor((oy0 + 15 + pad) / s) - floor((oy0 + pad - (k - 1)) / s) + 1
    c.RH = c.RW = (TILE - 1 + (s->k - 1)) / s->stride + 2 - (s->stride == 1 ? 1 : 0);
    c.rows_pad = (c.RH * c.RW + 31) / 32 * 32;
    c.ld = c.ncols | 1;                              // odd row stride: conflict-free column walks
    if (c.ld == c.ncols) c.ld += 2;
    c.t_off = t_off;
    t_off += c.rows_pad * c.ld;
    return LOANS_OK;
}

template <typename TG>
int crop_dgrad_impl(const void* gy_a, const float* w_a, const loans_small_conv* ca, const void* gy_b, const float* w_b,
                    const loans_small_conv* cb, float* out, const float* addend, int B, int H, int W, int C, void* stream) {
    if (!out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return LOANS_EINVAL;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)B * H * W * 4 >= lim) return LOANS_ERANGE;
    CropArgs a;
    a.nconv = gy_b ? 2 : 1;
    int t_off = 0;
    if (int rc = setup_conv(a.c[0], gy_a, w_a, ca, H, W, t_off)) return rc;
    if (gy_b) {
        if (int rc = setup_conv(a.c[1], gy_b, w_b, cb, H, W, t_off)) return rc;
    }
    for (int p = 0; p < a.nconv; ++p)
        if ((int64_t)B * a.c[p].gH * a.c[p].gW * C >= lim) return LOANS_ERANGE;
    a.nunits = 0;
    for (int p = 0; p < a.nconv; ++p)
        for (int cbk = 0; cbk < a.c[p].ncb; ++cbk)
            for (int rb = 0; rb < a.c[p].rows_pad / 32; ++rb) {
                if (a.nunits >= MAX_UNITS) return LOANS_ERANGE;
                a.unit_conv[a.nunits] = (unsigned char)p;
                a.unit_rb[a.nunits] = (unsigned char)rb;
                a.unit_cb[a.nunits] = (unsigned char)cbk;
                ++a.nunits;
            }
    a.out = out; a.addend = addend; a.B = B; a.H = H; a.W = W; a.C = C;
    a.tiles_y = (H + TILE - 1) / TILE;
    a.tiles_x = (W + TILE - 1) / TILE;
    if ((int64_t)B * a.tiles_y * a.tiles_x >= lim) return LOANS_ERANGE;
    const size_t lds = (size_t)t_off * 4;
    if (lds > 80 * 1024) return LOANS_ERANGE;
    static loans_device_once lds_limit_set;
    auto kern = crop_dgrad_kernel<TG>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), 80 * 1024)) return rc_;
    hipLaunchKernelGGL(kern, dim3(B * a.tiles_y * a.tiles_x), dim3(256), lds, as_stream(stream), a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

}  // namespace

extern "C" int loans_crop_dgrad_f32(const float* gy_a, const float* w_a, const loans_small_conv* ca, const float* gy_b,
                                    const float* w_b, const loans_small_conv* cb, float* out, const float* addend,
                                    int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    return crop_dgrad_impl<float>(gy_a, w_a, ca, gy_b, w_b, cb, out, addend, B, H, W, C, stream);
}

extern "C" int loans_crop_dgrad_bf16_f32(const void* gy_a, const float* w_a, const loans_small_conv* ca, const void* gy_b,
                                         const float* w_b, const loans_small_conv* cb, float* out, const float* addend,
                                         int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    return crop_dgrad_impl<__bf16>(gy_a, w_a, ca, gy_b, w_b, cb, out, addend, B, H, W, C, stream);
}
