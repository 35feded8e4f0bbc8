// LOANS_TILE_PW: the 1 x 1 / 1 convolutions with a SHORT K and a wide output -- ResNet-50's bottleneck expansions of res2 / res3
// (common/net.py:59-66 via sheep/resnet.py's ResNet-50 block: conv3 64 -> 256 at 128 x 128, 128 -> 512 at 64 x 64), whose bytes
// are four fifths output.  The implicit-GEMM tiles of igemm_bf16.hip stage both operands in LDS, contract, and then run a two-pass
// epilogue, one block per CU, in sequence: 0.50 / 0.44 of the HBM bound on those two layers (profiles/r4_r50_bench.json).
//
// This form keeps NO operand in LDS.  A wave owns strips of 32 pixels: the strip's A fragments (32 x K bf16) go global -> VGPR
// once, 16 bytes per lane and K step, which IS the 32x32x16 MFMA operand layout; the weight fragments (N x K x 2 <= 128 KB: L1 / L2
// resident) come global -> VGPR per use from a copy in FRAGMENT ORDER (loans_pw_pack_bf16: one coalesced 1 KiB load per fragment;
// from the [Cout][Cin] layout the same loads touch 32 rows x 32 bytes and the kernel is a third slower, tools/pw_probe.hip), one
// 32 x 32 accumulator tile at a time, the next tile's fragments requested before this tile's MFMAs; a finished tile is rounded,
// transposed through a per-wave LDS slab that holds HALF a pixel row ([32][N / 2 + 8] bf16) and written as runs of N bytes per
// pixel, non-temporal when the output is large; the next strip's A fragments are in flight under the second half's stores.  No
// block barrier in the loop: up to 16 waves per CU drift apart and cover each other's latencies.  BN statistics (LOANS_F_STATS)
// are the fp32 accumulators' sums: per tile and lane over 16 rows in fp32 (a fixed order), from there in DOUBLE -- per block in LDS
// (ds_add_f64), once per block with fp64 atomics into the block's replica -- so that, like the tiled kernels' sums, they do not
// depend on the order in which waves and blocks arrive (fp32 LDS atomics made the BN statistics, and with them the whole forward,
// differ in the last bit from run to run; and ds_add_f32 is the slower instruction on this chip: 184 against 158 us on res2's
// expansion).  The products and their order (K ascending from a zero accumulator) are those of loans_igemm_bf16s' other tiles: the
// outputs are bit-identical to theirs.  Cin = 256 (res4's expansions): pw16_k256_kernel below.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LOANS_F_AFFINE_IN: the A operand is relu(x * scale + shift) rounded to bf16 -- the BatchNormalization + ReLU in front of the
// convolution applied while its INPUT goes global -> VGPR (loans_bn_apply_bf16's arithmetic, bit for bit: the same fp32 expression,
// the same round-to-nearest-even): the activation tensor between bn2 and conv3 of a bottleneck is never written or read
__device__ __forceinline__ u32x4 pw_affine_relu(u32x4 raw, const float* sp, const float* tp) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x8_t v = __builtin_bit_cast(bf16x8_t, raw);
    f32x4 lo = __builtin_convertvector(__builtin_shufflevector(v, v, 0, 1, 2, 3), f32x4);
    f32x4 hi = __builtin_convertvector(__builtin_shufflevector(v, v, 4, 5, 6, 7), f32x4);
    lo = lo * *reinterpret_cast<const f32x4*>(sp) + *reinterpret_cast<const f32x4*>(tp);
    hi = hi * *reinterpret_cast<const f32x4*>(sp + 4) + *reinterpret_cast<const f32x4*>(tp + 4);
    lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
    hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
    const bf16x4_t a = __builtin_convertvector(lo, bf16x4_t), b = __builtin_convertvector(hi, bf16x4_t);
    return __builtin_bit_cast(u32x4, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int K, int NC, bool STATS, int OCC, bool AFF>
__global__ __launch_bounds__(256, OCC) void pw16_kernel(const __bf16* __restrict__ A, const u32x4* __restrict__ W,
                                                                   __bf16* __restrict__ out, double* __restrict__ stats, int M, int N_,
                                                                   int nstrips, int nt_out, const float* __restrict__ aff) {
    const int N = NC ? NC : N_;             // NC != 0: the column count is a compile-time constant (slab rows at immediate offsets)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = K / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int half = N / 2, pitch = half + 8;        // bf16 elements per staged half pixel row
    double* sums = reinterpret_cast<double*>(smem);  // [2][N], only with stats
    __bf16* slab = reinterpret_cast<__bf16*>(smem + (STATS ? (size_t)2 * N * sizeof(double) : 0)) + (size_t)wave * 32 * pitch;
    float* affs = reinterpret_cast<float*>(smem + (STATS ? (size_t)2 * N * sizeof(double) : 0) + (size_t)4 * 32 * pitch * 2);   // [2][K], AFF
    const int wave_id = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    const int upr = half / 8, tiles_half = half / 32;
    if (STATS) {
        for (int i = threadIdx.x; i < 2 * N; i += 256) sums[i] = 0.0;
    }
    if (AFF) {
        for (int i = threadIdx.x; i < 2 * K; i += 256) affs[i] = aff[i];
    }
    if (STATS || AFF) __syncthreads();
    auto affine = [&](int s, u32x4* a) {            // (rows past M stay zero operands)
        const bool live = s * 32 + r < M;
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 v = pw_affine_relu(a[ks], affs + ks * 16 + h * 8, affs + K + ks * 16 + h * 8);
            a[ks] = live ? v : zero;
        }
    };
    auto load_a = [&](int s, u32x4* a) {
        // rows past M (the ragged last strip) are zero operands: their accumulators are zero -- nothing for the sums, never stored
        const bool live = s * 32 + r < M;
        const int row = live ? s * 32 + r : M - 1;
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(A + (size_t)row * K + ks * 16 + h * 8);
            a[ks] = live ? v : zero;
        }
    };
    auto load_w = [&](int nt, u32x4* b) {           // fragment order (loans_pw_pack_bf16): one coalesced 1 KiB load per fragment
        const u32x4* wf = W + (size_t)nt * KS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] = wf[ks * 64];
    };
    u32x4 a[KS], an[KS];
    if (wave_id < nstrips) { load_a(wave_id, a); if (AFF) affine(wave_id, a); }
    for (int s = wave_id; s < nstrips; s += nwaves) {
        const int m0 = s * 32;
        u32x4 b[KS], bn[KS];
        load_w(0, b);
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll 1
            for (int t = 0; t < tiles_half; ++t) {
                const int nt = hf * tiles_half + t;
                load_w(nt + 1 < N / 32 ? nt + 1 : 0, bn);          // (the last request wraps to tile 0: the next strip's first)
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ks]), __builtin_bit_cast(bf16x8_t, b[ks]), acc, 0, 0, 0);
                __bf16* col = slab + 4 * h * pitch + t * 32 + r;         // this lane's column of the slab; its 16 rows at fixed offsets
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 v = {acc[e], acc[e + 1]};
                    const bf16x2_t p = __builtin_convertvector(v, bf16x2_t);
                    col[((e & 3) + 8 * (e >> 2)) * pitch] = p[0];
                    col[((e & 3) + 1 + 8 * (e >> 2)) * pitch] = p[1];
                }
                if (STATS) {
                    f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};             // (v_pk_add_f32 / v_pk_fma_f32: half the VALU instructions)
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const f32x2 v = {acc[e], acc[e + 1]};
                        s2 += v;
                        q2 = v * v + q2;
                    }
                    // v_permlane32_swap: (sum, squares) of the two half-waves -> lanes 0..31 hold the channel's sum over all 32 rows,
                    // lanes 32..63 its sum of squares; ONE ds_add_f64 on 64 distinct addresses (no LDS round trip to wait for)
                    // (as inline assembly: this compiler's __builtin_amdgcn_permlane32_swap returns its first result twice; the
                    // s_nop cover the VALU-write -> swap and swap -> VALU-read wait states the compiler no longer sees)
                    float sm = s2[0] + s2[1], sq = q2[0] + q2[1];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(sm), "+v"(sq));
                    atomic_add_f64(sums + h * N + nt * 32 + r, (double)(sm + sq));
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) b[ks] = bn[ks];
            }
            if (hf == 1 && s + nwaves < nstrips) load_a(s + nwaves, an);     // in flight under the store phase
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int u = lane; u < 32 * upr; u += 64) {
                const int pr = u / upr, cu = u - pr * upr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(slab + pr * pitch + cu * 8);
                if (m0 + pr < M) {
                    u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)(m0 + pr) * N + hf * half + cu * 8);
                    if (nt_out) __builtin_nontemporal_store(v, dst);
                    else *dst = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = an[ks];
        if (AFF && s + nwaves < nstrips) affine(s + nwaves, a);
    }
    if (STATS) {
        __syncthreads();
        double* st = stats + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * N;
        for (int i = threadIdx.x; i < 2 * N; i += 256)
            if (sums[i] != 0.0) atomic_add_f64(st + i, sums[i]);
    }
}

// K = 256 (res4's expansions, 256 -> 1024 at 32 x 32): the strip's A fragments are 64 VGPRs, so the weights of a tile come in two
// halves of eight fragments (the second half requested before the first half's MFMAs, the next tile's first half before the
// second's), the next strip's pixels are not prefetched (at B = 64 there is one strip per resident wave anyway), and the slab holds
// 128 columns.  Everything else as above; 2 blocks per CU.
template <int NC, bool STATS, bool AFF>
__global__ __launch_bounds__(256, 2) void pw16_k256_kernel(const __bf16* __restrict__ A, const u32x4* __restrict__ W, __bf16* __restrict__ out,
                                                           double* __restrict__ stats, int M, int N_, int nstrips, int nt_out,
                                                           const float* __restrict__ aff) {
    const int N = NC ? NC : N_;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 256, KS = 16, PC = 128, pitch = PC + 8, upr = PC / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    double* sums = reinterpret_cast<double*>(smem);
    __bf16* slab = reinterpret_cast<__bf16*>(smem + (STATS ? (size_t)2 * N * sizeof(double) : 0)) + (size_t)wave * 32 * pitch;
    float* affs = reinterpret_cast<float*>(smem + (STATS ? (size_t)2 * N * sizeof(double) : 0) + (size_t)4 * 32 * pitch * 2);   // [2][K], AFF
    const int wave_id = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    const int phases = N / PC;
    if (STATS) {
        for (int i = threadIdx.x; i < 2 * N; i += 256) sums[i] = 0.0;
    }
    if (AFF) {
        for (int i = threadIdx.x; i < 2 * K; i += 256) affs[i] = aff[i];
    }
    if (STATS || AFF) __syncthreads();
    auto load_w = [&](int nt, int part, u32x4* b) {
        const u32x4* wf = W + ((size_t)nt * KS + part * 8) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) b[ks] = wf[ks * 64];
    };
    for (int s = wave_id; s < nstrips; s += nwaves) {
        const int m0 = s * 32;
        u32x4 a[KS], b0[8], b1[8];
        {
            const bool live = m0 + r < M;
            const int row = live ? m0 + r : M - 1;
            const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                u32x4 v = *reinterpret_cast<const u32x4*>(A + (size_t)row * K + ks * 16 + h * 8);
                if (AFF) v = pw_affine_relu(v, affs + ks * 16 + h * 8, affs + K + ks * 16 + h * 8);
                a[ks] = live ? v : zero;
            }
        }
        load_w(0, 0, b0);
#pragma unroll 1
        for (int ph = 0; ph < phases; ++ph) {
#pragma unroll 1
            for (int t = 0; t < PC / 32; ++t) {
                const int nt = ph * (PC / 32) + t;
                load_w(nt, 1, b1);
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ks]), __builtin_bit_cast(bf16x8_t, b0[ks]), acc, 0, 0, 0);
                load_w(nt + 1 < N / 32 ? nt + 1 : 0, 0, b0);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[8 + ks]), __builtin_bit_cast(bf16x8_t, b1[ks]), acc, 0, 0, 0);
                __bf16* col = slab + 4 * h * pitch + t * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 v = {acc[e], acc[e + 1]};
                    const bf16x2_t p = __builtin_convertvector(v, bf16x2_t);
                    col[((e & 3) + 8 * (e >> 2)) * pitch] = p[0];
                    col[((e & 3) + 1 + 8 * (e >> 2)) * pitch] = p[1];
                }
                if (STATS) {
                    f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const f32x2 v = {acc[e], acc[e + 1]};
                        s2 += v;
                        q2 = v * v + q2;
                    }
                    float sm = s2[0] + s2[1], sq = q2[0] + q2[1];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(sm), "+v"(sq));
                    atomic_add_f64(sums + h * N + nt * 32 + r, (double)(sm + sq));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int u = lane; u < 32 * upr; u += 64) {
                const int pr = u / upr, cu = u - pr * upr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(slab + pr * pitch + cu * 8);
                if (m0 + pr < M) {
                    u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)(m0 + pr) * N + ph * PC + cu * 8);
                    if (nt_out) __builtin_nontemporal_store(v, dst);
                    else *dst = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (STATS) {
        __syncthreads();
        double* st = stats + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * N;
        for (int i = threadIdx.x; i < 2 * N; i += 256)
            if (sums[i] != 0.0) atomic_add_f64(st + i, sums[i]);
    }
}

// weights [Cout][Cin] bf16 -> fragment order: packed[((nt * KS + ks) * 64 + lane) * 8 + j] = w[nt * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 + j]
__global__ __launch_bounds__(256) void pw16_pack_kernel(const u32x4* __restrict__ w, u32x4* __restrict__ packed, int N, int K) {
    const int KS = K / 16, total = N / 32 * KS * 64;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int lane = i & 63, f = i >> 6, ks = f % KS, nt = f / KS;
        packed[i] = w[((size_t)(nt * 32 + (lane & 31)) * K + ks * 16 + (lane >> 5) * 8) / 8];
    }
}

extern "C" int loans_pw_pack_bf16(const void* w, void* packed, int32_t Cout, int32_t Cin, void* stream) {
    if (!w || !packed || Cout <= 0 || Cout % 32 != 0 || Cin <= 0 || Cin % 16 != 0) return LOANS_EINVAL;
    const int total = Cout / 32 * (Cin / 16) * 64;
    hipLaunchKernelGGL(pw16_pack_kernel, dim3(grid_for(total, 256, 64)), dim3(256), 0, as_stream(stream), static_cast<const u32x4*>(w),
                       static_cast<u32x4*>(packed), Cout, Cin);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

// every LOANS_TILE_PW layer's weights of a step in ONE launch, straight from the fp32 masters (rounded to nearest even, like the
// bf16 shadow loans_cast_bf16 makes): unit u of the launch = one 16-byte group of job j's packed matrix, j found by bisection over
// first_unit
__global__ __launch_bounds__(256) void pw16_pack_batch_kernel(const loans_pw_pack_job* __restrict__ jobs, int njobs, int total_units) {
    for (int u = blockIdx.x * 256 + threadIdx.x; u < total_units; u += gridDim.x * 256) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first_unit <= u) lo = mid; else hi = mid - 1;
        }
        const loans_pw_pack_job j = jobs[lo];
        const int i = u - j.first_unit, KS = j.Cin / 16;
        const int lane = i & 63, f = i >> 6, ks = f % KS, nt = f / KS;
        const float* src = static_cast<const float*>(j.src) + (size_t)(nt * 32 + (lane & 31)) * j.Cin + ks * 16 + (lane >> 5) * 8;
        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src), hi4 = *reinterpret_cast<const f32x4*>(src + 4);
        const loans_bf16x4 l = __builtin_convertvector(lo4, loans_bf16x4), hh = __builtin_convertvector(hi4, loans_bf16x4);
        bf16x8_t o;
        o[0] = l[0]; o[1] = l[1]; o[2] = l[2]; o[3] = l[3];
        o[4] = hh[0]; o[5] = hh[1]; o[6] = hh[2]; o[7] = hh[3];
        reinterpret_cast<bf16x8_t*>(j.dst)[i] = o;
    }
}

extern "C" int loans_pw_pack_batch_f32(const loans_pw_pack_job* jobs_dev, int32_t njobs, int32_t total_units, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_units <= 0) return LOANS_EINVAL;
    hipLaunchKernelGGL(pw16_pack_batch_kernel, dim3(grid_for(total_units, 256, 1024)), dim3(256), 0, as_stream(stream), jobs_dev, njobs,
                       total_units);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

// what LOANS_TILE_PW covers: a 1 x 1 / 1 forward geometry (grid = input = output pixels), Cin in {64, 128} with Cout a multiple of 64
// up to 512, or Cin = 256 with Cout a multiple of 128 up to 1024; flags STATS or none
int loans_pw16_covers(const loans_igemm_desc* d) {
    if (d->ntaps != 1 || d->dy[0] != 0 || d->dx[0] != 0) return 0;
    if (d->isy != 1 || d->isx != 1 || d->osy != 1 || d->osx != 1 || d->oy0 != 0 || d->ox0 != 0) return 0;
    if (d->gridH != d->inH || d->gridW != d->inW || d->gridH != d->outH || d->gridW != d->outW) return 0;
    if (d->Cin == 256) {
        if (d->Cout % 128 != 0 || d->Cout < 128 || d->Cout > 1024) return 0;
    } else {
        if (d->Cin != 64 && d->Cin != 128) return 0;
        if (d->Cout % 64 != 0 || d->Cout < 64 || d->Cout > 512) return 0;
    }
    if (d->flags & ~(LOANS_F_STATS | LOANS_F_AFFINE_IN)) return 0;
    return 1;
}

template <int K, int NC, bool STATS, int OCC, bool AFF>
static int pw16_launch_n(const void* in, const void* w, void* out, double* stats, int M, int N, int nt_out, const float* aff, hipStream_t st) {
    static loans_device_once lds_limit_set;
    const int nstrips = (M + 31) / 32;
    const size_t lds = (size_t)4 * 32 * (N / 2 + 8) * 2 + (STATS ? (size_t)2 * N * sizeof(double) : 0) + (AFF ? (size_t)2 * K * sizeof(float) : 0);
    const void* kern = reinterpret_cast<const void*>(pw16_kernel<K, NC, STATS, OCC, AFF>);
    if (lds > 64 * 1024) {
        const int rc = loans_raise_lds_limit(lds_limit_set, kern, 80 * 1024);
        if (rc != LOANS_OK) return rc;
    }
    const int cus = loans_device_cus();
    if (cus <= 0) return LOANS_EINVAL;
    int blocks = cus * OCC;
    if (blocks > (nstrips + 3) / 4) blocks = (nstrips + 3) / 4;
    hipLaunchKernelGGL((pw16_kernel<K, NC, STATS, OCC, AFF>), dim3(blocks), dim3(256), lds, st, static_cast<const __bf16*>(in), static_cast<const u32x4*>(w),
                       static_cast<__bf16*>(out), stats, M, N, nstrips, nt_out, aff);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <int NC, bool STATS, bool AFF>
static int pw16_launch_k256(const void* in, const void* w, void* out, double* stats, int M, int N, int nt_out, const float* aff, hipStream_t st) {
    const int nstrips = (M + 31) / 32;
    const size_t lds = (size_t)4 * 32 * (128 + 8) * 2 + (STATS ? (size_t)2 * N * sizeof(double) : 0) + (AFF ? (size_t)2 * 256 * sizeof(float) : 0);   // <= 52 KB
    const int cus = loans_device_cus();
    if (cus <= 0) return LOANS_EINVAL;
    int blocks = cus * 2;
    if (blocks > (nstrips + 3) / 4) blocks = (nstrips + 3) / 4;
    hipLaunchKernelGGL((pw16_k256_kernel<NC, STATS, AFF>), dim3(blocks), dim3(256), lds, st, static_cast<const __bf16*>(in),
                       static_cast<const u32x4*>(w), static_cast<__bf16*>(out), stats, M, N, nstrips, nt_out, aff);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

// the two shapes of the ResNet-50 localizer (N = 4 K) have the column count folded into the code
template <int K, bool STATS, bool AFF>
static int pw16_launch_k(const void* in, const void* w, void* out, double* stats, int M, int N, int nt_out, const float* aff, hipStream_t st) {
    // resident blocks per CU: K = 64: 4 (<= 128 VGPRs; the statistics form with a run-time N needs 141: 3), K = 128: 2
    constexpr int OCC = K <= 64 ? 4 : 2;
    if (N == 4 * K) return pw16_launch_n<K, 4 * K, STATS, OCC, AFF>(in, w, out, stats, M, N, nt_out, aff, st);
    return pw16_launch_n<K, 0, STATS, (K <= 64 && STATS) ? 3 : OCC, AFF>(in, w, out, stats, M, N, nt_out, aff, st);
}

template <bool AFF>
static int pw16_launch_aff(const void* in, const void* w, void* out, double* stp, const loans_igemm_desc* d, int M, int N, int nt_out,
                           const float* aff, hipStream_t st) {
    if (d->Cin == 256) {
        if (N == 1024) return stp ? pw16_launch_k256<1024, true, AFF>(in, w, out, stp, M, N, nt_out, aff, st) : pw16_launch_k256<1024, false, AFF>(in, w, out, stp, M, N, nt_out, aff, st);
        return stp ? pw16_launch_k256<0, true, AFF>(in, w, out, stp, M, N, nt_out, aff, st) : pw16_launch_k256<0, false, AFF>(in, w, out, stp, M, N, nt_out, aff, st);
    }
    if (d->Cin == 64)
        return stp ? pw16_launch_k<64, true, AFF>(in, w, out, stp, M, N, nt_out, aff, st) : pw16_launch_k<64, false, AFF>(in, w, out, stp, M, N, nt_out, aff, st);
    return stp ? pw16_launch_k<128, true, AFF>(in, w, out, stp, M, N, nt_out, aff, st) : pw16_launch_k<128, false, AFF>(in, w, out, stp, M, N, nt_out, aff, st);
}

// aff: float[2][Cin] = scale, shift (LOANS_F_AFFINE_IN), else ignored
int loans_pw16_launch(const void* in, const void* w, void* out, double* stats, const float* aff, const loans_igemm_desc* d, hipStream_t st) {
    if (!loans_pw16_covers(d)) return LOANS_EINVAL;
    const int64_t M64 = (int64_t)d->B * d->gridH * d->gridW;
    if (M64 <= 0 || M64 > 0x7FFFFFFF - 64) return LOANS_ERANGE;
    const int M = (int)M64, N = d->Cout;
    double* stp = (d->flags & LOANS_F_STATS) ? stats : nullptr;
    const int nt_out = loans_conv_nt((size_t)M * N * 2);
    if ((d->flags & LOANS_F_STATS) && !stats) return LOANS_EINVAL;
    if (d->flags & LOANS_F_AFFINE_IN) {
        if (!aff) return LOANS_EINVAL;
        return pw16_launch_aff<true>(in, w, out, stp, d, M, N, nt_out, aff, st);
    }
    return pw16_launch_aff<false>(in, w, out, stp, d, M, N, nt_out, nullptr, st);
}
