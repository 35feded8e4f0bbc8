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

constexpr int STEM_F_NT = 0x40000000;        // internal flag bit: output stores non-temporal (loans_conv_nt; common.h)


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
            float* op = obase + (size_t)(p0 + (e & 3) + 8 * (e >> 2)) * 64;
            if (flags & STEM_F_NT) __builtin_nontemporal_store(v, op);
            else *op = v;
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
    const int nt = loans_conv_nt((size_t)B * Ho * Wo * 64 * 4) ? STEM_F_NT : 0;
    hipLaunchKernelGGL(kern, dim3(B * (Ho / R)), dim3(256), lds, st, in, w, out, bias, stats, Hp, Wp3, Ho, Wo, R, flags | nt);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}


// ---- the same layer on the bf16 MFMA (bf16 compute arm: fp32 frame and weights rounded to bf16 RNE, fp32 accumulate, bias and
// statistics in fp32, bf16 NHWC output -- the arithmetic of igemm_kernel<BF16> with LOANS_F_OUT_BF16) ----------------------------
// As an implicit GEMM this layer is the slowest forward launch of the bf16 joint step (0.62 ms at 128 x 3 x 512^2 for 0.20 ms of
// HBM traffic: every tile gathers its own 7 x 24 window per pixel and converts it while staging).  Here a block is PERSISTENT:
//   * it rounds the whole weight matrix to bf16 ONCE into registers -- 21 (+1 zero) groups of 8 K-values x 64 channels are the
//     B fragments of v_mfma_f32_32x32x16_bf16 as they lie in the dense [64][7][24] layout, 88 VGPRs;
//   * per unit (R output rows of one image) it stages the 2R + 5 input rows once, as bf16 (one contiguous run of the padded
//     frame buffer, converted on the way), and every wave builds its A fragments from that image: lane (pixel r, half h) of
//     step s reads the 8 consecutive bf16 of group G = 2 s + h at  patch[2 oy + G / 3][6 ox + 8 (G % 3)]  (4-byte aligned:
//     two ds_read2_b32).  K = 176 computed per 147 real;
//   * a wave owns whole 32-pixel tiles with all 64 channels (one A fragment feeds two MFMAs: half the LDS reads of a 2 x 2
//     wave grid), transposes each finished tile through its own LDS slab and writes the pixels' 128 contiguous bytes;
//   * units are handed out in contiguous runs, so the 5 input rows two neighbouring units share are re-read by the same CU
//     while they are still in L2; BN statistics are summed per unit in fp32, across units in fp64, one atomic per channel
//     and block at the end.
typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sbf16x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) sbf16x8_a4 { sbf16x8 v; };      // an A fragment in the image: dword-aligned only
constexpr int S16_STEPS = 11;       // 22 groups of 8 K-values (21 real: 7 rows x 24), two per MFMA
constexpr int S16_LDC = 68;         // fp32 transposing slab of a wave: [32 pixels][64 channels + 4]
constexpr size_t STEM16_LDS_MAX = 78 * 1024;     // two blocks per CU

// eight consecutive K-values of an operand as bf16 (fp32 operands: rounded to nearest even here)
__device__ __forceinline__ sbf16x8 s16_load8(const float* p) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
    const sbf16x4 l4 = __builtin_convertvector(lo, sbf16x4), h4 = __builtin_convertvector(hi, sbf16x4);
    return __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ sbf16x8 s16_load8(const __bf16* p) { return *reinterpret_cast<const sbf16x8*>(p); }

// TIN = float: fp32 frame buffer and weights (loans_igemm_bf16_f32); TIN = __bf16: the bf16 frame buffer of
// loans_prep_images_dense_bf16 and bf16 weights (loans_igemm_bf16s) -- the same operands, already rounded
template <typename TIN>
__global__ __launch_bounds__(256, 2) void stem7_bf16_kernel(const TIN* in, const TIN* w, __bf16* out, const float* bias,
                                                            double* stats, int Hp, int Wp3, int Ho, int Wo, int R, int units,
                                                            int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rows_in = 2 * R + 5;
    const int nfl = rows_in * Wp3;                              // even (Wp3 is)
    __bf16* patch0 = reinterpret_cast<__bf16*>(smem);           // [<= 7 elements of lead-in][2R + 5][Wp3]
    float* slab = reinterpret_cast<float*>(smem + (((size_t)(nfl + 8) * 2 + 15) & ~(size_t)15));
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    float* Cs = slab + wave * 32 * S16_LDC;
    const int per_img = Ho / R, npx = R * Wo, ntile = (npx + 31) >> 5;

    // weights -> bf16 B fragments, once per block
    sbf16x8 bf[S16_STEPS][2];
#pragma unroll
    for (int s = 0; s < S16_STEPS; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int G = 2 * s + h;
            sbf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
            bf[s][j] = G < 21 ? s16_load8(w + (size_t)(32 * j + r) * 168 + G * 8) : z;
        }
    float bv[2] = {0.f, 0.f};
    if (flags & LOANS_F_BIAS) { bv[0] = bias[r]; bv[1] = bias[32 + r]; }
    double ssum[2] = {0., 0.}, ssq[2] = {0., 0.};

    // this block's run of units
    const int nb = gridDim.x;
    const int u_begin = (int)((long long)units * blockIdx.x / nb), u_end = (int)((long long)units * (blockIdx.x + 1) / nb);
    for (int u = u_begin; u < u_end; ++u) {
        const int b = u / per_img, oy0 = (u - b * per_img) * R;
        __syncthreads();            // the previous unit's readers are done with the image
        // stage the unit's 2R + 5 input rows (one contiguous run) in 16-byte pieces: the loads of a batch first, the LDS writes
        // behind them.  The run starts 16-byte aligned in an fp32 buffer (Hp, Wp3 even: checked) but only 4-byte aligned in
        // a bf16 one: there the copy starts at the aligned address below it and the image sits `lead` elements into the LDS
        // buffer (the elements before it belong to the previous row of the same buffer: readable)
        const TIN* src = in + ((size_t)b * Hp + 2 * oy0) * Wp3;
        constexpr int EPV = 16 / (int)sizeof(TIN);                  // elements per 16 bytes: 4 (fp32) / 8 (bf16)
        const int lead = (int)((reinterpret_cast<uintptr_t>(src) & 15) / sizeof(TIN));
        {
            const TIN* src_al = src - lead;
            const int nel = nfl + lead, nv = nel / EPV;
            constexpr int PU = 8;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            for (int base = 0; base < nv; base += 256 * PU) {
                u32x4 pv[PU];
#pragma unroll
                for (int q = 0; q < PU; ++q) {
                    const int i = base + tid + 256 * q;
                    pv[q] = i < nv ? *reinterpret_cast<const u32x4*>(src_al + (size_t)EPV * i) : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int q = 0; q < PU; ++q) {
                    const int i = base + tid + 256 * q;
                    if (i < nv) {
                        if constexpr (sizeof(TIN) == 4)
                            *reinterpret_cast<sbf16x4*>(patch0 + 4 * i) = __builtin_convertvector(__builtin_bit_cast(f32x4, pv[q]), sbf16x4);
                        else
                            *reinterpret_cast<u32x4*>(patch0 + 8 * i) = pv[q];
                    }
                }
            }
            if (tid < nel - nv * EPV) patch0[nv * EPV + tid] = (__bf16)src_al[nv * EPV + tid];
        }
        const __bf16* patch = patch0 + lead;
        __syncthreads();

        float us[2] = {0.f, 0.f}, uq[2] = {0.f, 0.f};
        for (int mt = wave; mt < ntile; mt += 4) {
            const int p = mt * 32 + r;
            const int pc = p < npx ? p : 0;                     // ragged last tile: rows beyond the unit read pixel 0
            const int oyl = pc / Wo, ox = pc - oyl * Wo;
            const __bf16* abase = patch + 2 * oyl * Wp3 + 6 * ox;
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
            for (int s = 0; s < S16_STEPS; ++s) {
                // group of this half; the zero-weight group 21 re-reads group 20 (finite pixels x 0)
                const int G0 = 2 * s, G1 = 2 * s + 1 < 21 ? 2 * s + 1 : 20;
                const int o0 = (G0 / 3) * Wp3 + 8 * (G0 % 3), o1 = (G1 / 3) * Wp3 + 8 * (G1 % 3);
                const sbf16x8 av = reinterpret_cast<const sbf16x8_a4*>(abase + (h ? o1 : o0))->v;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bf[s][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bf[s][1], acc[1], 0, 0, 0);
            }
            // + bias, statistics of the fp32 result, transpose through the wave's slab
            // (a whole tile -- every tile of a frame whose R * Wo is a multiple of 32 -- takes the form without the per-row selects:
            // the kernel is issue-bound, 38 % of its wave cycles issue and the MFMA pipe is 25 % busy, profiles/r5_cfg3_step_sq_table.txt)
            const bool whole = mt * 32 + 32 <= npx;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float s1 = 0.f, s2 = 0.f;
                if (whole) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                        const float v = acc[j][e] + bv[j];
                        s1 += v;
                        s2 = __builtin_fmaf(v, v, s2);
                        Cs[row * S16_LDC + 32 * j + r] = v;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                        const float v = acc[j][e] + bv[j];
                        const bool ok = mt * 32 + row < npx;
                        s1 += ok ? v : 0.f;
                        s2 = ok ? __builtin_fmaf(v, v, s2) : s2;        // (the same arithmetic as the whole tile's)
                        Cs[row * S16_LDC + 32 * j + r] = v;
                    }
                }
                us[j] += s1; uq[j] += s2;
            }
            // a wave's slab is private and its LDS operations execute in program order: no block barrier (the waves run
            // different tile counts), only the compiler must not move the reads above the other lanes' writes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __bf16* otile = out + (((size_t)b * Ho + oy0) * Wo + (size_t)mt * 32) * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int px = it * 8 + (lane >> 3), g8 = lane & 7;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + px * S16_LDC + 8 * g8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + px * S16_LDC + 8 * g8 + 4);
                const sbf16x4 l4 = __builtin_convertvector(lo, sbf16x4), h4 = __builtin_convertvector(hi, sbf16x4);
                if (mt * 32 + px < npx) {
                    const sbf16x8 ov = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
                    sbf16x8* op = reinterpret_cast<sbf16x8*>(otile + (size_t)px * 64 + 8 * g8);
                    if (flags & STEM_F_NT) __builtin_nontemporal_store(ov, op);
                    else *op = ov;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the slab is rewritten by the next tile
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { ssum[j] += (double)us[j]; ssq[j] += (double)uq[j]; }
    }
    if (flags & LOANS_F_STATS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double s1 = ssum[j], s2 = ssq[j];
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                double* st = stats + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * 64;
                atomic_add_f64(st + 32 * j + r, s1);
                atomic_add_f64(st + 64 + 32 * j + r, s2);
            }
        }
    }
}


// ---- conv1's weight gradient as a direct kernel (fp32 MFMA): dw[co][ky][6 kx' + c] += sum_pixels gy[p][co] * x[2 oy + ky][6 ox + j]
// As an implicit GEMM (wgrad_kernel) the layer runs at 65-69 TFLOP/s: K = 168 is short, every 32-pixel chunk gathers its own
// 7 x 24 window per pixel, and the result tile is small against the reduction.  Here a block is persistent over units (one
// output row of one image): it stages the 7 input rows (one contiguous run) and the row's gradient pixels [Wo][64] once, and
// the GEMM is  rows = 64 channels, columns = 147 real (ky, j) window positions (5 MFMA tiles), reduction = pixels:
//   A = gy[p + h][channel]           (lane = channel, h = the pixel of the MFMA's k pair)
//   B = patch[ky][6 (ox + h) + j]    (lane = window position)
// The eight waves split the row's pixel pairs, each accumulates the whole 64 x 160 tile in registers over ALL the block's units;
// one reduction through LDS and one atomic add per element and block at the very end.  The next unit is brought in by LDS-DMA
// while this one is contracted (two LDS buffers, one barrier per unit).
constexpr int SWG_CT = 5;           // 32-column tiles: 147 real columns of 160

// LDS: two unit buffers, each the 7 input rows and the row's gradient pixels in whole 1 KiB LDS-DMA pieces
__host__ __device__ inline int swg_x_pieces(int Wp3) { return (7 * Wp3 * 4 + 1023) >> 10; }
__host__ __device__ inline int swg_g_pieces(int Wo) { return (Wo + 3) >> 2; }            // a pixel's 64 channels = 256 bytes

constexpr int SWG_NW = 8;           // waves per block: two per SIMD, one block per CU

__global__ __launch_bounds__(64 * SWG_NW) void stem7_wgrad_kernel(const float* x, const float* gy, float* dw, int Hp, int Wp3, int Ho,
                                                             int Wo, int units, unsigned x_bytes, unsigned gy_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int xp = swg_x_pieces(Wp3), gp = swg_g_pieces(Wo);
    const int buf_floats = (xp + gp) * 256;
    float* const lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, (int)gy_bytes, 0x00020000);

    // per lane: the window position of each column tile (columns >= 147 re-read column 146; their sums are dropped)
    int coff[SWG_CT];
#pragma unroll
    for (int ct = 0; ct < SWG_CT; ++ct) {
        const int c = min(ct * 32 + r, 146);
        const int ky = c / 21, j = c - ky * 21;
        coff[ct] = ky * Wp3 + j + 6 * h;
    }
    f32x16 acc[2][SWG_CT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < SWG_CT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][ct][e] = 0.f;

    // unit u = output row oy of image b: its 7 input rows are one contiguous run of the padded buffer, its gradient pixels
    // another; both go to LDS by LDS-DMA (no staging registers: the accumulators own the register file), a piece past the
    // end of a run brings the bytes behind it (finite pixels / the next row's gradient: never multiplied in, see below) or,
    // past the end of the buffer, zeros
    auto stage = [&](int u, int buf) {
        const int b = u / Ho, oy = u - b * Ho;
        const unsigned xoff = (unsigned)((b * Hp + 2 * oy) * Wp3) * 4u + (unsigned)lane * 16u;
        const unsigned goff = (unsigned)((b * Ho + oy) * Wo) * 256u + (unsigned)lane * 16u;
        float* dst = lds + buf * buf_floats;
        for (int q = wave; q < xp; q += SWG_NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(dst + q * 256), 16, (int)(xoff + (unsigned)q * 1024u), 0, 0, 0);
        dst += xp * 256;
        for (int q = wave; q < gp; q += SWG_NW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, (lds_ptr_t)(dst + q * 256), 16, (int)(goff + (unsigned)q * 1024u), 0, 0, 0);
    };

    const int nb = gridDim.x;
    const int u_begin = (int)((long long)units * blockIdx.x / nb), u_end = (int)((long long)units * (blockIdx.x + 1) / nb);
    const int ksteps = (Wo + 1) >> 1;
    int buf = 0;
    if (u_begin < u_end) stage(u_begin, 0);
    for (int u = u_begin; u < u_end; ++u) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();            // unit u has landed; every wave is done with the other buffer
        if (u + 1 < u_end) stage(u + 1, buf ^ 1);
        const float* patch = lds + buf * buf_floats;
        const float* gyt = patch + xp * 256;                    // [Wo][64]
        // software-pipelined over the wave's pixel pairs: the operands of pair n + 1 are read while pair n is contracted
        float a0, a1, bv[SWG_CT];
        auto fetch = [&](int ks) {
            const int px = 2 * ks + h;
            const float* ap = gyt + px * 64 + r;
            a0 = ap[0]; a1 = ap[32];
            if (px >= Wo) { a0 = 0.f; a1 = 0.f; }              // the missing pixel of an odd row
            const float* bp = patch + 12 * ks;
#pragma unroll
            for (int ct = 0; ct < SWG_CT; ++ct) bv[ct] = bp[coff[ct]];
        };
        if (wave < ksteps) fetch(wave);
        for (int ks = wave; ks < ksteps; ks += SWG_NW) {
            const float c0 = a0, c1 = a1;
            float cb[SWG_CT];
#pragma unroll
            for (int ct = 0; ct < SWG_CT; ++ct) cb[ct] = bv[ct];
            if (ks + SWG_NW < ksteps) fetch(ks + SWG_NW);
#pragma unroll
            for (int ct = 0; ct < SWG_CT; ++ct) {
                acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, cb[ct], acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, cb[ct], acc[1][ct], 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    // the waves' partial tiles -> one sum per element -> dw (+=)
    float* red = lds;                                           // [SWG_NW][32][33]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < SWG_CT; ++ct) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e)
                red[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = acc[i][ct][e];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 1024 / (64 * SWG_NW); ++q) {
                const int idx = tid + 64 * SWG_NW * q, row = idx >> 5, col = idx & 31;
                float v = 0.f;
#pragma unroll
                for (int wv = 0; wv < SWG_NW; ++wv) v += red[(wv * 32 + row) * 33 + col];
                const int c = ct * 32 + col;
                if (c < 147) {
                    const int ky = c / 21, j = c - ky * 21;
                    atomic_add_f32(dw + (size_t)(i * 32 + row) * 168 + ky * 24 + j, v);
                }
            }
        }
}


// ---- conv1's weight gradient on the bf16 MFMA (bf16 storage arm: the bf16 frame buffer of loans_prep_images_dense_bf16, a bf16
// gradient; round 5).  As an implicit GEMM (wgrad16_kernel<64, 64>) this is the LAST launch of the joint step's backward -- it can
// only start when the stem's BN / pool backward has written conv1's gradient, so nothing runs beside it -- and it takes 0.47 ms at
// 128 x 3 x 512^2 for 0.20 ms of HBM traffic: K = 168 is three 64-column tiles that each gather their own 7 x 24 windows per
// 32-pixel chunk, two MFMAs per wave between barriers (a 64 x 256 tile that reads the gradient once is SLOWER: 0.61-0.79 ms).
// The direct form of stem7_wgrad_kernel: a block is persistent over units (one output row of one image), stages the unit's 7 input
// rows and its gradient pixels [Wo][64] once, and contracts rows = 64 channels x columns = window positions over 16 pixels per MFMA.
// Both operands lie pixel-major and are read with ds_read_b64_tr_b16 (the transposing LDS read, as in wgrad16_kernel):
//   A = gy[ox0 + 8 h + i][channel]: rows padded to 144 bytes (the four pixel rows of a read lie 36 banks apart);
//   B = x[2 oy + ky][6 (ox0 + 8 h + i) + j]: consecutive pixels lie 6 elements = 12 bytes apart, which no wide LDS read takes (the
//       first version gathered eight ds_read_u16 per fragment: 640 per unit, LDS-bound at 0.215 ms without a single global load).
//       The 7 input rows are one run of 12-byte CELLS (two frame pixels = the stride of the window), 7 (Wo + 3) of them; staged one
//       cell per 16 bytes, pixel ox's window of a row is the 32 elements from cell ox on -- column n' = 8 a + b' is element b' of
//       cell ox + a = window position j = 6 a + b' (b' = 6, 7 are padding, their sums are dropped) -- and a kernel row is ONE
//       32-column MFMA tile with a pixel stride of 16 bytes: two transposing reads per fragment.  7 x 32 = 224 computed columns for
//       147 real ones: the MFMA pipe has the room (1 800 of a unit's 5 300 HBM cycles).
// Waves 0-3 own kernel rows 0-3, waves 4-7 rows 4-6, all 64 channels each (128 / 96 accumulator registers); the four waves of a group
// split the row's 16-pixel steps and keep their tiles over ALL the block's units.  The next unit travels global -> registers while
// this one is contracted (buffer_load_dwordx3 per cell), registers -> the other LDS buffer behind it: one barrier per unit.  One
// reduction over each group's waves through LDS at the very end; the block's tile goes to its slab of `ws` (plain stores, the
// window-padding columns as zeros; loans_fold_slabs_f32 adds the slabs in a fixed order) or, without a workspace, to dw by atomics.
constexpr int SWB_NW = 8;           // waves per block: two per SIMD, one block per CU
constexpr int SWB_GS = 72;          // LDS stride of a gradient pixel (elements): 64 channels + 8
constexpr int SWB_PC = 4;           // input cells per thread and unit (7 (Wo + 3) <= 2048)
constexpr int SWB_PG = 4;           // 16-byte pieces of the gradient row per thread (Wo <= 256)
typedef unsigned su32x4 __attribute__((ext_vector_type(4)));
typedef unsigned su32x3 __attribute__((ext_vector_type(3)));

__host__ __device__ inline int swb_patch_bytes(int Wo) { return 7 * (Wo + 3) * 16; }

__global__ __launch_bounds__(64 * SWB_NW) void stem7_wgrad_bf16_kernel(const __bf16* x, const __bf16* gy, float* dw, float* ws, int Hp,
                                                                  int Wp3, int Ho, int Wo, int units, unsigned x_bytes,
                                                                  unsigned gy_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) sbf16x4* lds_b64_t;
    const int ncr = Wo + 3, ncell = 7 * ncr, nvg = Wo * 8;      // cells per input row (Wp3 = 6 ncr: checked), per unit; gradient pieces
    const int patch_bytes = swb_patch_bytes(Wo), buf_bytes = patch_bytes + Wo * SWB_GS * 2;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;                   // kernel rows 4 grp .. (3 of them in group 1); the group's step phase
    const int nt = grp ? 3 : 4;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x), 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(gy), 0, (int)gy_bytes, 0x00020000);

    // transposing reads: 16-lane group (h, cg) takes pixels 8 h + 4 t .. + 3 and columns 16 cg .. + 15 of a 32-column tile; lane
    // 4 q + p of the group addresses pixel row q, columns 4 p .. 4 p + 3 (element offsets below; the second read is 4 pixels on)
    const int li = lane & 15, fq = li >> 2, fp = li & 3, cg = (lane >> 4) & 1;
    const int trg = (8 * h + fq) * SWB_GS + 16 * cg + 4 * fp;   // gradient tile: pixel stride SWB_GS
    const int trb = (8 * h + fq) * 8 + 16 * cg + 4 * fp;        // input cells: pixel stride 8 elements

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.f;

    // unit u = output row oy of image b: its 7 input rows are 7 (Wo + 3) consecutive 12-byte cells of the padded buffer (4-byte
    // aligned), its gradient pixels Wo x 128 contiguous bytes
    su32x3 rx[SWB_PC];
    su32x4 rg[SWB_PG];
    auto fetch = [&](int u) {
        const int b = u / Ho, oy = u - b * Ho;
        const unsigned xrun = (unsigned)((b * Hp + 2 * oy) * Wp3) * 2u;
        const unsigned grun = (unsigned)u * (unsigned)Wo * 128u;
#pragma unroll
        for (int q = 0; q < SWB_PC; ++q) {
            const int c = tid + 64 * SWB_NW * q;
            rx[q] = __builtin_amdgcn_raw_buffer_load_b96(rs_x, (int)((xrun + 12u * (unsigned)c) | (c < ncell ? 0u : 0x80000000u)), 0, 0);
        }
#pragma unroll
        for (int q = 0; q < SWB_PG; ++q) {
            const int v = tid + 64 * SWB_NW * q;
            rg[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)((grun + 16u * (unsigned)v) | (v < nvg ? 0u : 0x80000000u)), 0, 0);
        }
    };
    auto stash = [&](int buf) {
        char* base = smem + buf * buf_bytes;
#pragma unroll
        for (int q = 0; q < SWB_PC; ++q) {
            const int c = tid + 64 * SWB_NW * q;
            if (c < ncell) *reinterpret_cast<su32x4*>(base + 16 * c) = su32x4{rx[q].x, rx[q].y, rx[q].z, 0u};
        }
#pragma unroll
        for (int q = 0; q < SWB_PG; ++q) {
            const int v = tid + 64 * SWB_NW * q;
            if (v < nvg) *reinterpret_cast<su32x4*>(base + patch_bytes + (v >> 3) * (SWB_GS * 2) + (v & 7) * 16) = rg[q];
        }
    };

    const int nb = gridDim.x;
    const int u_begin = (int)((long long)units * blockIdx.x / nb), u_end = (int)((long long)units * (blockIdx.x + 1) / nb);
    const int ksteps = Wo >> 4;                                 // Wo % 16 == 0 (checked by the launcher)
    int buf = 0;
    if (u_begin < u_end) fetch(u_begin);
    for (int u = u_begin; u < u_end; ++u) {
        stash(buf);
        __syncthreads();            // unit u is in LDS; every wave is done with unit u - 1 (the other buffer, rewritten at u + 1)
        if (u + 1 < u_end) fetch(u + 1);
        const __bf16* cells = reinterpret_cast<const __bf16*>(smem + buf * buf_bytes) + (4 * grp * ncr) * 8 + trb;
        const __bf16* gyt = reinterpret_cast<const __bf16*>(smem + buf * buf_bytes + patch_bytes) + trg;
        for (int ks = wq; ks < ksteps; ks += 4) {
            sbf16x8 af[2], bf[4];
            const __bf16* ap = gyt + 16 * ks * SWB_GS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const sbf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(ap + 32 * i));
                const sbf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(ap + 32 * i + 4 * SWB_GS));
                af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            const __bf16* bp = cells + 16 * ks * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < nt) {
                    const sbf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(bp + t * ncr * 8));
                    const sbf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(bp + t * ncr * 8 + 32));
                    bf[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < nt) {
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[t], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[t], acc[1][t], 0, 0, 0);
                }
        }
        buf ^= 1;
    }
    // each group's four partial tiles -> one sum per element -> the block's slab / dw (+=)
    float* const slab = ws ? ws + (size_t)blockIdx.x * (64 * 168) : nullptr;
    if (slab)
        for (int idx = tid; idx < 64 * 7 * 3; idx += 64 * SWB_NW) {
            const int co = idx / 21, rem = idx - co * 21;
            slab[co * 168 + (rem / 3) * 24 + 21 + rem % 3] = 0.f;
        }
    float* red = reinterpret_cast<float*>(smem) + grp * (4 * 32 * 33);         // [4 waves][32][33] per group
    const int gt = tid & 255;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __syncthreads();
            if (t < nt) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    red[(wq * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = acc[i][t][e];
            }
            __syncthreads();
            if (t < nt) {
                const int ky = 4 * grp + t;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = gt + 256 * q, row = idx >> 5, col = idx & 31;
                    float v = 0.f;
#pragma unroll
                    for (int wv = 0; wv < 4; ++wv) v += red[(wv * 32 + row) * 33 + col];
                    const int a = col >> 3, bq = col & 7, j = 6 * a + bq;
                    if (bq < 6 && j < 21) {
                        const size_t o = (size_t)(i * 32 + row) * 168 + ky * 24 + j;
                        if (slab) slab[o] = v;
                        else atomic_add_f32(dw + o, v);
                    }
                }
            }
        }
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

// rows per unit of the bf16 direct stem for a frame geometry, 0 = not covered
int loans_stem7_bf16_rows(int Ho, int Wo, int Wp3, size_t* lds_bytes) {
    for (int R = 4; R >= 1; R >>= 1) {
        if (Ho % R) continue;
        const size_t lds = ((((size_t)(2 * R + 5) * Wp3 + 8) * 2 + 15) & ~(size_t)15) + (size_t)4 * 32 * S16_LDC * sizeof(float);
        if (lds > STEM16_LDS_MAX) continue;
        if (lds_bytes) *lds_bytes = lds;
        return R;
    }
    return 0;
}

namespace {
template <typename TIN>
int stem7_bf16_launch(const TIN* in, const TIN* w, void* out, const float* bias, double* stats, const loans_igemm_desc* d,
                      hipStream_t st) {
    if (!(d->flags & LOANS_F_DENSE) || (d->flags & ~(LOANS_F_DENSE | LOANS_F_OUT_BF16 | LOANS_F_BIAS | LOANS_F_STATS))) return LOANS_EINVAL;
    if (d->ntaps != 7 || d->Cin != 24 || d->Cout != 64 || d->isy != 2 || d->isx != 6) return LOANS_EINVAL;
    if ((d->inW & 1) || (d->inH & 1) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return LOANS_EINVAL;
    for (int t = 0; t < 7; ++t)
        if (d->dy[t] != t || d->dx[t] != 0) return LOANS_EINVAL;
    if (d->osy != 1 || d->osx != 1 || d->oy0 || d->ox0 || d->outH != d->gridH || d->outW != d->gridW) return LOANS_EINVAL;
    if (2 * (d->gridH - 1) + 7 > d->inH || 6 * (d->gridW - 1) + 24 > d->inW) return LOANS_EINVAL;
    if ((int64_t)d->B * d->inH * d->inW >= ((int64_t)1 << 31) || (int64_t)d->B * d->gridH * d->gridW * 64 >= ((int64_t)1 << 31))
        return LOANS_ERANGE;
    size_t lds = 0;
    const int R = loans_stem7_bf16_rows(d->gridH, d->gridW, d->inW, &lds);
    if (!R) return LOANS_EINVAL;
    if (2 * (d->gridH - R) + 2 * R + 5 > d->inH) return LOANS_EINVAL;       // the last unit's 2R + 5 input rows exist
    static loans_device_once lds_limit_set[2];
    auto kern = stem7_bf16_kernel<TIN>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set[sizeof(TIN) == 2], reinterpret_cast<const void*>(kern), STEM16_LDS_MAX)) return rc_;
    const int cus = loans_device_cus();
    if (cus <= 0) return LOANS_EINVAL;
    const int units = d->B * (d->gridH / R);
    const int nblk = units < 2 * cus ? units : 2 * cus;
    const int nt = loans_conv_nt((size_t)d->B * d->gridH * d->gridW * 64 * 2) ? STEM_F_NT : 0;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, in, w, reinterpret_cast<__bf16*>(out), bias, stats,
                       d->inH, d->inW, d->gridH, d->gridW, R, units, d->flags | nt);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}
}  // namespace

// LOANS_TILE_STEM of loans_igemm_bf16_f32 (fp32 frame buffer and weights; LOANS_F_OUT_BF16 required: the output is bf16)
int loans_stem7_bf16_launch(const float* in, const float* w, void* out, const float* bias, double* stats,
                            const loans_igemm_desc* d, hipStream_t st) {
    if (!(d->flags & LOANS_F_OUT_BF16)) return LOANS_EINVAL;
    return stem7_bf16_launch<float>(in, w, out, bias, stats, d, st);
}

// LOANS_TILE_STEM of loans_igemm_bf16s (bf16 frame buffer and weights)
int loans_stem7_bf16s_launch(const void* in, const void* w, void* out, const float* bias, double* stats,
                             const loans_igemm_desc* d, hipStream_t st) {
    if (d->flags & LOANS_F_OUT_BF16) return LOANS_EINVAL;       // implied there
    return stem7_bf16_launch<__bf16>(reinterpret_cast<const __bf16*>(in), reinterpret_cast<const __bf16*>(w), out, bias, stats, d, st);
}

// LOANS_TILE_STEM of loans_wgrad_f32: `d` is the dense 7x7 / 2, Cout = 64 forward geometry; dw [64][7][24] (+=; the three
// window-padding columns of every row are left alone)
int loans_stem7_wgrad_launch(const float* x, const float* gy, float* dw, const loans_igemm_desc* d, hipStream_t st) {
    if (d->flags != LOANS_F_DENSE) return LOANS_EINVAL;
    if (d->ntaps != 7 || d->Cin != 24 || d->Cout != 64 || d->isy != 2 || d->isx != 6) return LOANS_EINVAL;
    if ((d->inW & 1) || (d->inH & 1) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(gy) & 15)) return LOANS_EINVAL;
    for (int t = 0; t < 7; ++t)
        if (d->dy[t] != t || d->dx[t] != 0) return LOANS_EINVAL;
    if (d->osy != 1 || d->osx != 1 || d->oy0 || d->ox0 || d->outH != d->gridH || d->outW != d->gridW) return LOANS_EINVAL;
    if (2 * (d->gridH - 1) + 7 > d->inH || 6 * (d->gridW - 1) + 24 > d->inW) return LOANS_EINVAL;
    if ((int64_t)d->B * d->inH * d->inW >= ((int64_t)1 << 31) || (int64_t)d->B * d->gridH * d->gridW * 64 >= ((int64_t)1 << 31))
        return LOANS_ERANGE;
    size_t lds = (size_t)2 * (swg_x_pieces(d->inW) + swg_g_pieces(d->gridW)) * 1024;
    if (lds < (size_t)SWG_NW * 32 * 33 * sizeof(float)) lds = (size_t)SWG_NW * 32 * 33 * sizeof(float);
    if (lds > 156 * 1024) return LOANS_EINVAL;
    const int64_t xb = (int64_t)d->B * d->inH * d->inW * 4, gb = (int64_t)d->B * d->gridH * d->gridW * 256;
    if (xb >= 0xFFFFFFF0ll || gb >= 0xFFFFFFF0ll) return LOANS_ERANGE;
    static loans_device_once lds_limit_set;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(stem7_wgrad_kernel), 156 * 1024)) return rc_;
    const int cus = loans_device_cus();
    if (cus <= 0) return LOANS_EINVAL;
    const int per_cu = 1;                   // eight waves with 160 accumulator registers each fill the register file
    const int units = d->B * d->gridH;
    const int nblk = units < per_cu * cus ? units : per_cu * cus;
    hipLaunchKernelGGL(stem7_wgrad_kernel, dim3(nblk), dim3(64 * SWG_NW), lds, st, x, gy, dw, d->inH, d->inW, d->gridH, d->gridW, units,
                       (unsigned)xb, (unsigned)gb);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
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

// LOANS_TILE_STEM of loans_wgrad_bf16s: `d` is the dense 7x7 / 2, Cout = 64 forward geometry on the bf16 frame buffer.
// loans_stem7_wgrad_bf16_slabs: the blocks (= slabs of 64 x 168 floats) a launch runs, 0 = this geometry is not covered
int loans_stem7_wgrad_bf16_slabs(const loans_igemm_desc* d) {
    if (d->flags != LOANS_F_DENSE) return 0;
    if (d->ntaps != 7 || d->Cin != 24 || d->Cout != 64 || d->isy != 2 || d->isx != 6) return 0;
    if ((d->inH & 1) || (d->gridW & 15) || d->gridW > 64 * SWB_PG) return 0;
    if (d->inW != 6 * (d->gridW + 3)) return 0;                 // rows of whole 12-byte cells, Wo + 3 of them (even frame widths)
    if (7 * (d->gridW + 3) > 64 * SWB_NW * SWB_PC) return 0;
    for (int t = 0; t < 7; ++t)
        if (d->dy[t] != t || d->dx[t] != 0) return 0;
    if (d->osy != 1 || d->osx != 1 || d->oy0 || d->ox0 || d->outH != d->gridH || d->outW != d->gridW) return 0;
    if (2 * (d->gridH - 1) + 7 > d->inH) return 0;
    const size_t lds = (size_t)2 * (swb_patch_bytes(d->gridW) + d->gridW * SWB_GS * 2);
    if (lds > 156 * 1024) return 0;
    const int cus = loans_device_cus();
    if (cus <= 0) return 0;
    const int64_t units = (int64_t)d->B * d->gridH;
    return (int)(units < cus ? units : cus);
}

int loans_stem7_wgrad_bf16_launch(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, float* ws, hipStream_t st) {
    const int nblk = loans_stem7_wgrad_bf16_slabs(d);
    if (nblk <= 0) return LOANS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(gy) & 15) || (reinterpret_cast<uintptr_t>(ws) & 15)) return LOANS_EINVAL;
    const int64_t xb = (int64_t)d->B * d->inH * d->inW * 2, gb = (int64_t)d->B * d->gridH * d->gridW * 128;
    if (xb >= 0x7FFFFFF0ll || gb >= 0x7FFFFFF0ll) return LOANS_ERANGE;       // bit 31 of an offset marks a piece that is not loaded
    size_t lds = (size_t)2 * (swb_patch_bytes(d->gridW) + d->gridW * SWB_GS * 2);
    if (lds < (size_t)SWB_NW * 32 * 33 * sizeof(float)) lds = (size_t)SWB_NW * 32 * 33 * sizeof(float);
    static loans_device_once lds_limit_set;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(stem7_wgrad_bf16_kernel), 156 * 1024)) return rc_;
    hipLaunchKernelGGL(stem7_wgrad_bf16_kernel, dim3(nblk), dim3(64 * SWB_NW), lds, st, reinterpret_cast<const __bf16*>(x),
                       reinterpret_cast<const __bf16*>(gy), dw, ws, d->inH, d->inW, d->gridH, d->gridW, d->B * d->gridH, (unsigned)xb,
                       (unsigned)gb);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}
