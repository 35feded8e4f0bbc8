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
//   B operand: the forward OHWI weights, re-laid into MFMA fragment order by a 64-thread pre-pass of the same call
//     (crop_pack_w_kernel, 48 blocks, caller-owned 64 KB workspace), then 16 coalesced loads per (wave, column block);
//   v_mfma_f32_32x32x2_f32, fp32 accumulate; out-of-image gradient pixels are zero rows of T.
//   bf16 gradient tensors (the bf16 storage arm, round 6): the contraction runs on v_mfma_f32_32x32x16_bf16 -- lane (r, h) reads
//     channels 16 s + 8 h .. + 7 of its pixel as ONE 16-byte load per step (8 loads per row block instead of 16, 8 MFMAs instead
//     of 64), the weights are rounded to bf16 (RNE) by the packing pre-pass, fp32 accumulation and fp32 gx as before.  On fp32
//     MFMAs the kernel was bound by the matrix pipe, not by HBM (56 GFLOP at B = 256 = 0.36 ms at the 157 TFLOP/s peak).
// Bound: HBM (each gradient tensor once: 0.94 GB at B = 256 of 75 x 75 crops); replaces five launches of the VALU
// kernel in smalln.hip (2.8 ms per step) on this path -- that kernel stays for geometries this one does not cover.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 crop_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned crop_u32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 16;             // output tile edge
constexpr int MAX_JOBS = 4;
constexpr int KG = 16;               // 8-deep k groups held in registers: C = 128 (the assessor's width, common/net.py:71)

struct CropConv {
    const void* gy;                  // [B][gH][gW][C], float or bf16
    const float* w;                  // forward weights OHWI [C][k][k][4]
    int k, stride, pad, gH, gW;
    int ncols, ncb;                  // k*k*3, ceil(ncols / 32)
    int RW, RH, rows_pad;            // region of gradient pixels per tile (upper bound), rows padded to 32
    unsigned gy_bytes;               // bytes of the gradient tensor (< 2^31: buffer descriptor range)
    unsigned rw_recip;               // ceil(2^16 / RW): q / RW = (q * rw_recip) >> 16 for q < 2^10
    int ld;                          // floats per T row (odd)
    int t_off;                       // float offset of this problem's T in LDS
};

struct CropArgs {
    CropConv c[2];
    int nconv;
    float* out;
    const float* addend;
    int B, H, W, C, tiles_y, tiles_x;
    int njobs;
    unsigned char job_conv[MAX_JOBS], job_cb[MAX_JOBS];
    int dbg;            // experiment bits (LOANS_EXPERIMENT builds only; 0 in the product library)
};

#ifdef LOANS_EXPERIMENT
#define CDBG(bit) (a.dbg & (bit))
#else
#define CDBG(bit) false
#endif

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

// B fragments in the layout the MFMA lanes want, made once per call (the weights change every step): wpack[job][g][lane] is
// the f32x4 {w[8g + 4h + j][tap(n)][ci(n)], j = 0..3} of column n = 32 cb + lane % 32, h = lane / 32 (zero beyond the real
// columns / channels), so that a wave's B operand of a job is 16 coalesced 1 KB loads instead of 64 strided gathers.
__global__ __launch_bounds__(64) void crop_pack_w_kernel(const CropArgs a, f32x4* wpack) {
    const int jb = blockIdx.x / KG, g = blockIdx.x - jb * KG, lane = threadIdx.x;
    const CropConv& c = a.c[a.job_conv[jb]];
    const int n = a.job_cb[jb] * 32 + (lane & 31), h = lane >> 5;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < c.ncols && 8 * g < a.C) {
        const int tap = n / 3, ci = n - tap * 3, kk4 = c.k * c.k * 4;
        const float* wg = c.w + (int64_t)(8 * g + 4 * h) * kk4 + tap * 4 + ci;
        v = f32x4{wg[0], wg[kk4], wg[2 * kk4], wg[3 * kk4]};
    }
    wpack[(jb * KG + g) * 64 + lane] = v;
}

// the same for the bf16 contraction: wpack16[job][s][lane] = the eight bf16 {w[16 s + 8 h + j][tap(n)][ci(n)], j = 0..7}
__global__ __launch_bounds__(64) void crop_pack_w16_kernel(const CropArgs a, crop_bf16x8* wpack) {
    constexpr int KS = KG / 2;
    const int jb = blockIdx.x / KS, s = blockIdx.x - jb * KS, lane = threadIdx.x;
    const CropConv& c = a.c[a.job_conv[jb]];
    const int n = a.job_cb[jb] * 32 + (lane & 31), h = lane >> 5;
    crop_bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
    if (n < c.ncols && 16 * s < a.C) {
        const int tap = n / 3, ci = n - tap * 3, kk4 = c.k * c.k * 4;
        const float* wg = c.w + (int64_t)(16 * s + 8 * h) * kk4 + tap * 4 + ci;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)wg[j * kk4];
    }
    wpack[(jb * KS + s) * 64 + lane] = v;
}

template <typename TG> struct lda;           // four consecutive channels of a gradient pixel through a bounds-checked descriptor
template <> struct lda<float> {
    static constexpr int GB = 32;            // bytes per 8-deep k group
    static __device__ __forceinline__ f32x4 ld(__amdgpu_buffer_rsrc_t rs, int off) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
};
template <> struct lda<__bf16> {
    static constexpr int GB = 16;
    static __device__ __forceinline__ f32x4 ld(__amdgpu_buffer_rsrc_t rs, int off) {
        return __builtin_convertvector(__builtin_bit_cast(loans_bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0)), f32x4);
    }
};

template <typename TG>
__global__ __launch_bounds__(256, 2) void crop_dgrad_kernel(const CropArgs a, const f32x4* __restrict__ wpack) {
    extern __shared__ __attribute__((aligned(16))) float T[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int blk = blockIdx.x;
    const int tx = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty = blk % a.tiles_y;
    const int b = blk / a.tiles_y;
    const int oy0 = ty * TILE, ox0 = tx * TILE;
    constexpr int ES = (int)sizeof(TG);
    constexpr bool M16 = std::is_same<TG, __bf16>::value;       // bf16 gradients: bf16 MFMAs, 16 channels per step
    constexpr int NS = M16 ? KG / 2 : KG;                        // contraction steps (registers of an operand set: 32 / 64)
    typedef typename std::conditional<M16, crop_bf16x8, f32x4>::type frag_t;

    // A job = one (convolution, 32-column block): its B fragments are loaded once per wave (16 coalesced loads), its row
    // blocks are shared by the four waves round-robin in two alternating register sets, so that the 16 A loads of the wave's
    // next row block are in flight under the 64 MFMAs of the current one and nothing is copied.
    // (Measured and dropped: one flat unit sequence per wave across the jobs, prefetching over job boundaries -- the control
    // flow costs more than the three exposed load latencies per wave it saves: 0.64 vs 0.50 ms at B = 256.)
    for (int jb = 0; jb < a.njobs && !CDBG(32); ++jb) {
        const CropConv& c = a.c[a.job_conv[jb]];
        const int nrb = c.rows_pad >> 5;
        if (wave >= nrb) continue;
        const int ry0 = floordiv(oy0 + c.pad - (c.k - 1), c.stride);
        const int rx0 = floordiv(ox0 + c.pad - (c.k - 1), c.stride);
        const int n = a.job_cb[jb] * 32 + r;
        const bool ncol = n < c.ncols;
        frag_t bv[NS];
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            if constexpr (M16) bv[g] = reinterpret_cast<const crop_bf16x8*>(wpack)[(jb * NS + g) * 64 + lane];
            else bv[g] = CDBG(2) ? f32x4{0.f, 0.f, 0.f, 0.f} : wpack[(jb * KG + g) * 64 + lane];
        }
        // rows beyond the image / the region get an offset beyond the descriptor's range: the loads return zeros
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(c.gy), 0, (int)c.gy_bytes, 0x00020000);
        auto load_a = [&](int rb, frag_t (&av)[NS]) {
            const int q = rb * 32 + r;
            const int qy = (int)(((unsigned)q * c.rw_recip) >> 16), qx = q - qy * c.RW;       // q / RW, exact for q < 2^10
            const int y = ry0 + qy, x = rx0 + qx;
            const bool live = rb < nrb && qy < c.RH && (unsigned)y < (unsigned)c.gH && (unsigned)x < (unsigned)c.gW && !CDBG(1);
            const int off = live ? (((b * c.gH + y) * c.gW + x) * a.C + (M16 ? 8 : 4) * h) * ES : (int)0x80000000;
#pragma unroll
            for (int g = 0; g < NS; ++g) {
                if constexpr (M16) av[g] = __builtin_bit_cast(crop_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off + g * 32, 0, 0));
                else av[g] = lda<TG>::ld(rs, off + g * lda<TG>::GB);
            }
        };
        auto contract = [&](int rb, frag_t (&av)[NS]) {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (!CDBG(4))
#pragma unroll
            for (int g = 0; g < NS; ++g) {
                if constexpr (M16) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[g], bv[g], acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].x, bv[g].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].y, bv[g].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].z, bv[g].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g].w, bv[g].w, acc, 0, 0, 0);
                }
            }
            // D layout of the 32x32 MFMA: acc[v] = D[(v / 4) * 8 + h * 4 + v % 4][lane % 32]
            if (ncol && !CDBG(16)) {
                float* tp = T + c.t_off + n + (rb * 32 + h * 4) * c.ld;
#pragma unroll
                for (int v = 0; v < 16; ++v) tp[((v >> 2) * 8 + (v & 3)) * c.ld] = acc[v];
            }
        };
        // The prefetch is UNCONDITIONAL (a row block beyond the last one is all out-of-range offsets: zeros, no traffic):
        // with a conditional one the number of loads in flight behind a register set depends on the path taken, and the
        // compiler then has to wait for vmcnt(0) -- prefetch included -- before the first MFMA.
        frag_t a0[NS], a1[NS];
        load_a(wave, a0);
        for (int rb = wave; rb < nrb; rb += 8) {
            load_a(rb + 4, a1);
            contract(rb, a0);
            if (rb + 4 >= nrb) break;
            load_a(rb + 8, a0);
            contract(rb + 4, a1);
        }
    }
    __syncthreads();

    if (CDBG(8)) return;
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
        const int sh = c.stride - 1;                        // stride is 1 or 2
        for (int kr = 0; kr < c.k; ++kr) {
            const int ny = oy + c.pad - kr;
            if (ny < 0 || (ny & sh)) continue;
            const int qy = (ny >> sh) - ry0;                // in [0, RH): the region was sized for it
            for (int ks = 0; ks < c.k; ++ks) {
                const int nx = ox + c.pad - ks;
                if (nx < 0 || (nx & sh)) continue;
                const int qx = (nx >> sh) - rx0;
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

int setup_conv(CropConv& c, const void* gy, const float* w, const loans_small_conv* s, int H, int W, int& t_off, int64_t gy_bytes) {
    if (!gy || !w || !s) return LOANS_EINVAL;
    if (s->k < 1 || s->k > 4 || s->stride < 1 || s->stride > 2 || s->pad < 0 || s->pad >= s->k) return LOANS_EINVAL;
    if (s->outH != (H + 2 * s->pad - s->k) / s->stride + 1 || s->outW != (W + 2 * s->pad - s->k) / s->stride + 1) return LOANS_EINVAL;
    c.gy = gy; c.w = w; c.k = s->k; c.stride = s->stride; c.pad = s->pad; c.gH = s->outH; c.gW = s->outW;
    c.ncols = s->k * s->k * 3;
    c.ncb = (c.ncols + 31) / 32;
    // gradient rows a 16-row output tile can touch: floor((oy0 + 15 + pad) / s) - floor((oy0 + pad - (k - 1)) / s) + 1
    c.RH = c.RW = (TILE - 1 + (s->k - 1)) / s->stride + 2 - (s->stride == 1 ? 1 : 0);
    c.rows_pad = (c.RH * c.RW + 31) / 32 * 32;
    if (c.rows_pad > 1024 || gy_bytes >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    c.gy_bytes = (unsigned)gy_bytes;
    c.rw_recip = (65536u + c.RW - 1) / c.RW;
    c.ld = c.ncols | 1;                              // odd row stride: conflict-free column walks
    if (c.ld == c.ncols) c.ld += 2;
    c.t_off = t_off;
    t_off += c.rows_pad * c.ld;
    return LOANS_OK;
}

template <typename TG>
int crop_dgrad_impl(const void* gy_a, const float* w_a, const loans_small_conv* ca, const void* gy_b, const float* w_b,
                    const loans_small_conv* cb, float* out, const float* addend, float* wpack, int B, int H, int W, int C,
                    void* stream) {
    if (!out || !wpack || B <= 0 || H <= 0 || W <= 0 || C != 8 * KG) return LOANS_EINVAL;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)B * H * W * 4 >= lim) return LOANS_ERANGE;
    CropArgs a;
    a.nconv = gy_b ? 2 : 1;
    int t_off = 0;
    if (!ca || (gy_b && !cb)) return LOANS_EINVAL;
    if (int rc = setup_conv(a.c[0], gy_a, w_a, ca, H, W, t_off, (int64_t)B * ca->outH * ca->outW * C * (int64_t)sizeof(TG))) return rc;
    if (gy_b) {
        if (int rc = setup_conv(a.c[1], gy_b, w_b, cb, H, W, t_off, (int64_t)B * cb->outH * cb->outW * C * (int64_t)sizeof(TG))) return rc;
    }
    a.njobs = 0;
    for (int p = 0; p < a.nconv; ++p)
        for (int cbk = 0; cbk < a.c[p].ncb; ++cbk) {
            if (a.njobs >= MAX_JOBS) return LOANS_ERANGE;
            a.job_conv[a.njobs] = (unsigned char)p;
            a.job_cb[a.njobs] = (unsigned char)cbk;
            ++a.njobs;
        }
    a.dbg = 0;
#ifdef LOANS_EXPERIMENT
    if (const char* e = getenv("LOANS_CROP_DBG")) a.dbg = atoi(e);
#endif
    a.out = out; a.addend = addend; a.B = B; a.H = H; a.W = W; a.C = C;
    a.tiles_y = (H + TILE - 1) / TILE;
    a.tiles_x = (W + TILE - 1) / TILE;
    if ((int64_t)B * a.tiles_y * a.tiles_x >= lim) return LOANS_ERANGE;
    const size_t lds = (size_t)t_off * 4;
    if (lds > 80 * 1024) return LOANS_ERANGE;
    static loans_device_once lds_limit_set;
    auto kern = crop_dgrad_kernel<TG>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), 80 * 1024)) return rc_;
    if (std::is_same<TG, __bf16>::value)
        hipLaunchKernelGGL(crop_pack_w16_kernel, dim3(a.njobs * (KG / 2)), dim3(64), 0, as_stream(stream), a, reinterpret_cast<crop_bf16x8*>(wpack));
    else
        hipLaunchKernelGGL(crop_pack_w_kernel, dim3(a.njobs * KG), dim3(64), 0, as_stream(stream), a, reinterpret_cast<f32x4*>(wpack));
    LOANS_LAUNCH_CHECK();
    hipLaunchKernelGGL(kern, dim3(B * a.tiles_y * a.tiles_x), dim3(256), lds, as_stream(stream), a,
                       reinterpret_cast<const f32x4*>(wpack));
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

}  // namespace

extern "C" int loans_crop_dgrad_f32(const float* gy_a, const float* w_a, const loans_small_conv* ca, const float* gy_b,
                                    const float* w_b, const loans_small_conv* cb, float* out, const float* addend,
                                    float* wpack, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    return crop_dgrad_impl<float>(gy_a, w_a, ca, gy_b, w_b, cb, out, addend, wpack, B, H, W, C, stream);
}

extern "C" int loans_crop_dgrad_bf16_f32(const void* gy_a, const float* w_a, const loans_small_conv* ca, const void* gy_b,
                                         const float* w_b, const loans_small_conv* cb, float* out, const float* addend,
                                         float* wpack, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    return crop_dgrad_impl<__bf16>(gy_a, w_a, ca, gy_b, w_b, cb, out, addend, wpack, B, H, W, C, stream);
}
