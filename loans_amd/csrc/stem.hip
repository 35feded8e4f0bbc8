// conv1 of the localizer (7x7 / 2, 3 -> 64, bias, BN statistics; sheep/resnet.py:43,72) as a DIRECT convolution on the
// fp32 MFMA, for the dense packed-RGB frame buffer of loans_prep_images_dense_f32 (LOANS_F_DENSE geometry).
//
// As an implicit GEMM (igemm.hip) this layer streams 129 KB of operands through LDS per 128 x 64 tile for 2.75 MFLOP --
// 4.9 TB/s of L2 -> LDS traffic at B = 256, which is what bounds it (58 % of the MFMA peak): every output pixel fetches
// its own 7 x 24 floats although neighbours share 5 of 7 rows and 5 of 7 columns, and every tile re-fetches the weights.
// Here a block owns R consecutive output rows of one image (R * Wo pixels x 64 channels), stages the 2R + 5 input rows
// they read ONCE (contiguous in the padded buffer: one streaming copy) next to the whole weight matrix, k-major, and
// builds the MFMA operands straight from that image: lane (pixel r, k half h) of step (ky, jp) reads
//   A = patch[2 oy + ky][6 ox + 2 jp + h]        (6 = 2 pixels x 3 channels: the stride-2 window start)
//   B = Wt[ky * 22 + 2 jp + h][channel]
// with immediate LDS offsets off one base register per 32-pixel tile.  K per row is 22 (7 pixels x 3 + one zero-weight
// column to make it even) instead of the implicit GEMM's 24: 154 instead of 168 computed per 147 real.
// 4 waves: wave w owns channel half (w & 1) and the 32-pixel tiles (w >> 1), (w >> 1) + 2, ... (TMW of them).
#include "common.h"

namespace {

constexpr int SK = 22;              // K columns per kernel row (21 real + 1 zero weight)
constexpr int SKT = 7 * SK;         // 154
constexpr int WLD = 65;             // LDS row stride of the k-major weights (conflict-free transposing writes)
constexpr size_t STEM_LDS_MAX = 80 * 1024;      // two blocks per CU

template <int TMW>
__global__ __launch_bounds__(256) void stem7_kernel(const float* in, const float* w, float* out, const float* bias,
                                                    double* stats, int Hp, int Wp3, int Ho, int Wo, int R, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* patch = reinterpret_cast<float*>(smem);          // [2R + 5][Wp3]
    const int rows_in = 2 * R + 5;
    float* Wt = patch + rows_in * Wp3;                       // [154][65]
    const int tid = threadIdx.x;
    const int per_img = Ho / R;
    const int b = blockIdx.x / per_img, oy0 = (blockIdx.x - b * per_img) * R;

    {   // Both LDS images are filled in two phases -- every global load of the block first (independent, so their
        // latencies overlap), the LDS writes behind them: a copy loop that waits for each element in turn costs more than
        // the whole MFMA phase of the block.
        // The input rows 2 oy0 .. 2 oy0 + 2R + 4 are ONE contiguous run of the padded buffer, 16-byte aligned (checked by
        // the launcher: Hp and Wp3 even); weights [64][7][24] (dense layout, columns 21..23 = window padding) go to
        // Wt[ky * 22 + j][n]: coalesced float4 reads (24 = 6 x 4: a float4 never crosses a kernel row), transposing writes.
        const f32x4* src4 = reinterpret_cast<const f32x4*>(in + ((size_t)b * Hp + 2 * oy0) * Wp3);
        f32x4* dst4 = reinterpret_cast<f32x4*>(patch);
        const int nfl = rows_in * Wp3, n4 = nfl >> 2;
        constexpr int PW = (64 * 168 / 4 + 255) / 256;          // 11 float4 of weights per thread
        constexpr int PU = 10;                                  // float4 of the image per thread and batch
        f32x4 wv[PW], pv[PU];
        const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            const int i = tid + 256 * u;
            wv[u] = i < 64 * 168 / 4 ? w4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int i = tid + 256 * u;
            pv[u] = i < n4 ? src4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            const int i = (tid + 256 * u) * 4;
            if (i < 64 * 168) {
                const int n = i / 168, rem = i - n * 168;
                const int ky = rem / 24, j = rem - ky * 24;
                float* dstw = Wt + (ky * SK + j) * WLD + n;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < SK) dstw[e * WLD] = wv[u][e];
            }
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int i = tid + 256 * u;
            if (i < n4) dst4[i] = pv[u];
        }
        for (int base = 256 * PU; base < n4; base += 256 * PU) {        // frames wider than 224 px: further batches
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int i = base + tid + 256 * u;
                pv[u] = i < n4 ? src4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int i = base + tid + 256 * u;
                if (i < n4) dst4[i] = pv[u];
            }
        }
        if (tid < (nfl & 3)) patch[n4 * 4 + tid] = in[((size_t)b * Hp + 2 * oy0) * Wp3 + n4 * 4 + tid];
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int nt = wave & 1, m0 = wave >> 1;
    int arow[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int p = (m0 + 2 * i) * 32 + r;
        const int oyl = p / Wo, ox = p - oyl * Wo;
        arow[i] = 2 * oyl * Wp3 + 6 * ox + h;
    }
    const float* bptr = Wt + h * WLD + 32 * nt + r;
    f32x16 acc[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
        for (int jp = 0; jp < SK / 2; ++jp) {
            const float bv = bptr[(ky * SK + 2 * jp) * WLD];
#pragma unroll
            for (int i = 0; i < TMW; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(patch[arow[i] + 2 * jp], bv, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TMW; ++i) arow[i] += Wp3;
    }

    // epilogue: + bias, BN statistics of the result, NHWC store (a lane holds one channel of 16 pixels per tile)
    const int col = 32 * nt + r;
    const float bv = (flags & LOANS_F_BIAS) ? bias[col] : 0.f;
    float* obase = out + ((size_t)b * Ho + oy0) * Wo * 64 + col;
    float s = 0.f, q2 = 0.f;
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int p0 = (m0 + 2 * i) * 32 + 4 * h;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = acc[i][e] + bv;
            s += v;
            q2 += v * v;
            obase[(size_t)(p0 + (e & 3) + 8 * (e >> 2)) * 64] = v;
        }
    }
    if (flags & LOANS_F_STATS) {
        s += __shfl_xor(s, 32, 64);
        q2 += __shfl_xor(q2, 32, 64);
        if (h == 0) {
            double* st = stats + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * 64;
            atomic_add_f64(st + col, (double)s);
            atomic_add_f64(st + 64 + col, (double)q2);
        }
    }
}

template <int TMW>
int launch_stem7(const float* in, const float* w, float* out, const float* bias, double* stats, int B, int Hp, int Wp3,
                 int Ho, int Wo, int R, int flags, size_t lds, hipStream_t st) {
    static loans_device_once lds_limit_set;       // per template instance = per kernel, one bit per device
    auto kern = stem7_kernel<TMW>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), STEM_LDS_MAX)) return rc_;
    hipLaunchKernelGGL(kern, dim3(B * (Ho / R)), dim3(256), lds, st, in, w, out, bias, stats, Hp, Wp3, Ho, Wo, R, flags);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

}  // namespace

// rows per block for a frame geometry, 0 = not covered (the caller falls back to the implicit GEMM)
int loans_stem7_rows(int Ho, int Wo, int Wp3, size_t* lds_bytes) {
    for (int R = 4; R >= 1; R >>= 1) {
        if (Ho % R || (R * Wo) % 64) continue;
        const int mt = R * Wo / 32;
        if (mt > 14) continue;
        const size_t lds = ((size_t)(2 * R + 5) * Wp3 + (size_t)SKT * WLD) * sizeof(float);
        if (lds > STEM_LDS_MAX) continue;
        if (lds_bytes) *lds_bytes = lds;
        return R;
    }
    return 0;
}

// LOANS_TILE_STEM of loans_igemm_f32: `d` must be the dense 7x7 / 2, Cout = 64 forward geometry
int loans_stem7_launch(const float* in, const float* w, float* out, const float* bias, double* stats,
                       const loans_igemm_desc* d, hipStream_t st) {
    if (!(d->flags & LOANS_F_DENSE) || (d->flags & ~(LOANS_F_DENSE | LOANS_F_BIAS | LOANS_F_STATS))) return LOANS_EINVAL;
    if (d->ntaps != 7 || d->Cin != 24 || d->Cout != 64 || d->isy != 2 || d->isx != 6) return LOANS_EINVAL;
    // a block's image starts at row 2 oy0 of frame b: 16-byte aligned when row length and row count are even
    if ((d->inW & 1) || (d->inH & 1) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(w) & 15)) return LOANS_EINVAL;
    for (int t = 0; t < 7; ++t)
        if (d->dy[t] != t || d->dx[t] != 0) return LOANS_EINVAL;
    if (d->osy != 1 || d->osx != 1 || d->oy0 || d->ox0 || d->outH != d->gridH || d->outW != d->gridW) return LOANS_EINVAL;
    if (2 * (d->gridH - 1) + 7 > d->inH || 6 * (d->gridW - 1) + 24 > d->inW) return LOANS_EINVAL;
    if ((int64_t)d->B * d->inH * d->inW >= ((int64_t)1 << 31) || (int64_t)d->B * d->gridH * d->gridW * 64 >= ((int64_t)1 << 31))
        return LOANS_ERANGE;
    size_t lds = 0;
    const int R = loans_stem7_rows(d->gridH, d->gridW, d->inW, &lds);
    if (!R) return LOANS_EINVAL;
    // the block copies 2R + 5 whole input rows: the last block's must exist
    if (2 * (d->gridH - R) + 2 * R + 5 > d->inH) return LOANS_EINVAL;
    const int tmw = R * d->gridW / 64;
#define STEM_CASE(T) case T: return launch_stem7<T>(in, w, out, bias, stats, d->B, d->inH, d->inW, d->gridH, d->gridW, R, d->flags, lds, st)
    switch (tmw) {
        STEM_CASE(1); STEM_CASE(2); STEM_CASE(3); STEM_CASE(4); STEM_CASE(5); STEM_CASE(6); STEM_CASE(7);
        default: return LOANS_EINVAL;
    }
#undef STEM_CASE
}
