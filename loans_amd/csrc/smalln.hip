// Data gradient of a convolution whose INPUT has 4 (physical) channels: the gradient w.r.t. the
// RGB crops that links the assessor to the spatial transformer (common/net.py:15,17 r0.c0 / r0.cs).
// N = 4 output columns would waste 15/16 of a 32x32 MFMA tile, so this is a VALU kernel:
//   block = 32 output pixels x 8 lanes; the 8 lanes of a pixel split the gathered gradient's channels
//   (16-byte loads, the 8 lanes cover one contiguous pixel row: coalesced), the weights of the launch
//   W[co][tap][0..3] sit in LDS (row-padded against bank conflicts), partial sums are combined with
//   three wave shuffles.  Same problem descriptor and epilogue flags (MASK / ADDEND) as loans_igemm_f32;
//   weights are the forward OHWI tensor, `tapsel` maps the launch's taps to forward tap indices.
#include "common.h"

namespace {

struct SmallNArgs {
    const void* gy;         // float or bf16 tensor (template parameter of the kernel)
    const float* w;
    float* out;
    const float* ref;
    const float* addend;
    loans_igemm_desc d;
    int M, src_taps;
    int tapsel[LOANS_MAX_TAPS];
};

template <typename TG>
__global__ __launch_bounds__(256) void dgrad_c4_kernel(const SmallNArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* Wl = reinterpret_cast<f32x4*>(smem);        // [ntaps][C + C/16] float4
    const loans_igemm_desc& d = a.d;
    const int C = d.Cin;                    // channels of the gathered gradient (= forward Cout)
    const int CP = C + (C >> 4);            // one float4 of padding every 16 channels
    const int tid = threadIdx.x;
    for (int i = tid; i < d.ntaps * C; i += 256) {
        const int t = i / C, co = i - t * C;
        Wl[t * CP + co + (co >> 4)] = reinterpret_cast<const f32x4*>(a.w)[(int64_t)co * a.src_taps + a.tapsel[t]];
    }
    __syncthreads();

    const int s = tid & 7, p = tid >> 3;
    const int m = blockIdx.x * 32 + p;
    const bool live = m < a.M;
    const int mm = live ? m : 0;
    const int gHW = d.gridH * d.gridW;
    const int b = mm / gHW;
    const int rem = mm - b * gHW;
    const int y = rem / d.gridW;
    const int x = rem - y * d.gridW;
    const int cpl = C >> 3;                 // channels per lane (multiple of 4)
    const int c0 = s * cpl;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < d.ntaps; ++t) {
        const int iy = y * d.isy + d.dy[t], ix = x * d.isx + d.dx[t];
        const bool ok = live && (unsigned)iy < (unsigned)d.inH && (unsigned)ix < (unsigned)d.inW;
        if (!ok) continue;
        const TG* g = static_cast<const TG*>(a.gy) + (int64_t)((b * d.inH + iy) * d.inW + ix) * C + c0;
        const f32x4* wt = Wl + t * CP + c0 + (c0 >> 4);
        for (int c = 0; c < cpl; c += 4) {
            const f32x4 gv = io4<TG>::ld(g + c);
            const f32x4* wc = wt + c + ((c0 + c) >> 4) - (c0 >> 4);
            acc += wc[0] * gv.x + wc[1] * gv.y + wc[2] * gv.z + wc[3] * gv.w;
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
        acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (!live || s != 0) return;
    const int64_t off = ((int64_t)(b * d.outH + y * d.osy + d.oy0) * d.outW + x * d.osx + d.ox0) * 4;
    if (d.flags & LOANS_F_MASK) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(a.ref + off);
        acc.x = r.x > 0.f ? acc.x : 0.f; acc.y = r.y > 0.f ? acc.y : 0.f;
        acc.z = r.z > 0.f ? acc.z : 0.f; acc.w = r.w > 0.f ? acc.w : 0.f;
    }
    if (d.flags & LOANS_F_ADDEND) acc += *reinterpret_cast<const f32x4*>(a.addend + off);
    *reinterpret_cast<f32x4*>(a.out + off) = acc;
}

}  // namespace

template <typename TG>
static int dgrad_c4_impl(const void* gy, const float* w_ohwi, float* out, const float* ref,
                         const float* addend, const loans_igemm_desc* d, const int32_t* tapsel_host,
                         int32_t src_taps, void* stream) {
    if (!gy || !w_ohwi || !out || !d || !tapsel_host) return LOANS_EINVAL;
    if (d->Cout != 4 || d->Cin <= 0 || (d->Cin & 31) || d->ntaps < 1 || d->ntaps > LOANS_MAX_TAPS || src_taps < 1) return LOANS_EINVAL;
    if (d->B <= 0 || d->inH <= 0 || d->inW <= 0 || d->outH <= 0 || d->outW <= 0 || d->gridH <= 0 || d->gridW <= 0) return LOANS_EINVAL;
    if (d->osy <= 0 || d->osx <= 0 || d->isy <= 0 || d->isx <= 0 || d->oy0 < 0 || d->ox0 < 0) return LOANS_EINVAL;
    if ((d->gridH - 1) * d->osy + d->oy0 >= d->outH || (d->gridW - 1) * d->osx + d->ox0 >= d->outW) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_MASK) && !ref) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_ADDEND) && !addend) return LOANS_EINVAL;
    if (d->flags & ~(LOANS_F_MASK | LOANS_F_ADDEND)) return LOANS_EINVAL;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)d->B * d->inH * d->inW * d->Cin >= lim || (int64_t)d->B * d->outH * d->outW * 4 >= lim) return LOANS_ERANGE;
    const size_t lds = (size_t)d->ntaps * (d->Cin + (d->Cin >> 4)) * 16;
    if (lds > 64 * 1024) return LOANS_ERANGE;
    SmallNArgs a;
    a.gy = gy; a.w = w_ohwi; a.out = out; a.ref = ref; a.addend = addend; a.d = *d;
    a.M = d->B * d->gridH * d->gridW;
    a.src_taps = src_taps;
    for (int i = 0; i < d->ntaps; ++i) {
        if (tapsel_host[i] < 0 || tapsel_host[i] >= src_taps) return LOANS_EINVAL;
        a.tapsel[i] = tapsel_host[i];
    }
    hipLaunchKernelGGL(dgrad_c4_kernel<TG>, dim3((a.M + 31) / 32), dim3(256), lds, as_stream(stream), a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_dgrad_c4_f32(const float* gy, const float* w_ohwi, float* out, const float* ref,
                                  const float* addend, const loans_igemm_desc* d, const int32_t* tapsel_host,
                                  int32_t src_taps, void* stream) {
    return dgrad_c4_impl<float>(gy, w_ohwi, out, ref, addend, d, tapsel_host, src_taps, stream);
}

extern "C" int loans_dgrad_c4_bf16_f32(const void* gy, const float* w_ohwi, float* out, const float* ref,
                                       const float* addend, const loans_igemm_desc* d, const int32_t* tapsel_host,
                                       int32_t src_taps, void* stream) {
    return dgrad_c4_impl<__bf16>(gy, w_ohwi, out, ref, addend, d, tapsel_host, src_taps, stream);
}
