// Probe (not part of the product): how fast can a 1x1 convolution of the ResNet-50 localizer's kind -- out[M][N] = A[M][K] W[N][K]^T
// on bf16 tensors, K = 64 .. 256, N = 4 K: the bottleneck EXPANSIONS, whose bytes are four fifths output -- run if the kernel is
// built for that shape alone?  loans_igemm_bf16s (256 x 256 tile: one block per CU, load -> contract -> two-pass epilogue in
// sequence) reaches 0.50 / 0.44 / 0.38 of the HBM bound on res2 / res3 / res4's expansions (profiles/r4_r50_bench.json).
//
// Design probed here: NO operand tiles in LDS.  A wave owns a strip of 32 pixels: its A fragments (32 x K) go global -> VGPR
// once (16 bytes per lane and K step, exactly the MFMA operand layout), the weight fragments come global -> VGPR per use (W is
// L2 / L1 resident: N x K x 2 bytes <= 512 KB), one 32 x 32 accumulator tile at a time -> converted, transposed through a per-wave
// LDS slab ([32 pixels][N + 8] bf16) and written as whole 512-byte pixel rows.  No block barrier anywhere: eight waves per CU drift
// apart and cover each other's latencies.
// build: hipcc -O3 --offload-arch=gfx950 -o pw_probe tools/pw_probe.hip ; run: ./pw_probe [B*H*W] [K] [N]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// v2: weights pre-packed in MFMA fragment order (Wp[(nt * KS + ks) * 64 + lane] = the 16 bytes lane (r, h) needs for column tile
// nt, K step ks: one fully coalesced 1 KiB load per fragment), the next tile's fragments requested before this tile's MFMAs, the
// next strip's A fragments requested before this strip's store phase, the slab holds HALF the columns (8 KB per wave: 16 waves per CU)
#ifdef UNPACKED      // the weights as the library holds them, [N][K]: a fragment load touches 32 rows x 32 bytes
#define WLOAD(nt_, ks_) (*reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(Wp) + (size_t)((nt_) * 32 + r) * K + (ks_) * 16 + h * 8))
#else
#define WLOAD(nt_, ks_) Wp[((size_t)(nt_) * KS + (ks_)) * 64 + lane]
#endif
template <int K>
__global__ __launch_bounds__(256, K <= 64 ? 4 : 2) void pw_kernel(const __bf16* __restrict__ A, const u32x4* __restrict__ Wp, __bf16* __restrict__ out,
                                                     int M, int N, int nstrips) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = K / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int half = N / 2, pitch = half + 8;        // bf16 elements per staged pixel row
    __bf16* slab = reinterpret_cast<__bf16*>(smem) + (size_t)wave * 32 * pitch;
    const int wave_id = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    const int upr = half / 8, tiles_half = half / 32;
    auto load_a = [&](int s, u32x4* a) {
        const int row = s * 32 + r < M ? s * 32 + r : M - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const u32x4*>(A + (size_t)row * K + ks * 16 + h * 8);
    };
    u32x4 a[KS], an[KS];
    if (wave_id < nstrips) load_a(wave_id, a);
    for (int s = wave_id; s < nstrips; s += nwaves) {
        const int m0 = s * 32;
        u32x4 b[KS], bn[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] = WLOAD(0, ks);
        for (int hf = 0; hf < 2; ++hf) {
            for (int t = 0; t < tiles_half; ++t) {
                const int nt = hf * tiles_half + t;
                const int nn = nt + 1 < N / 32 ? nt + 1 : 0;            // (the last prefetch wraps to tile 0: the next strip's first)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) bn[ks] = WLOAD(nn, ks);
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ks]), __builtin_bit_cast(bf16x8_t, b[ks]), acc, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    slab[((e & 3) + 8 * (e >> 2) + 4 * h) * pitch + t * 32 + r] = (__bf16)acc[e];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) b[ks] = bn[ks];
            }
            if (hf == 1 && s + nwaves < nstrips) load_a(s + nwaves, an);     // in flight under the store phase
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int u = lane; u < 32 * upr; u += 64) {
                const int pr = u / upr, cu = u - pr * upr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(slab + pr * pitch + cu * 8);
                if (m0 + pr < M) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + (size_t)(m0 + pr) * N + hf * half + cu * 8));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = an[ks];
    }
}

static uint16_t f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int K>
static float run(const __bf16* A, const u32x4* W, __bf16* out, int M, int N, int reps, void* scrub, size_t scrub_bytes) {
    const int nstrips = (M + 31) / 32;
    const size_t lds = (size_t)4 * 32 * (N / 2 + 8) * 2;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pw_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        if (scrub) CHECK(hipMemsetAsync(scrub, i, scrub_bytes, 0));       // the tensors come from HBM, not from the previous repetition
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(pw_kernel<K>, dim3(1024), dim3(256), lds, 0, A, W, out, M, N, nstrips);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64 * 128 * 128, K = argc > 2 ? atoi(argv[2]) : 64, N = argc > 3 ? atoi(argv[3]) : 4 * K;
    if ((K != 64 && K != 128 && K != 256) || N % 64 || N > 2048) { printf("K in {64,128,256}, N %% 32 == 0, N <= 2048\n"); return 1; }
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = f2bf((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hW) v = f2bf((rand() % 2001 - 1000) / 4000.f);
    __bf16 *A, *W, *out; void* scrub;
    const size_t scrub_bytes = (size_t)512 << 20;
    CHECK(hipMalloc(&A, hA.size() * 2)); CHECK(hipMalloc(&W, hW.size() * 2)); CHECK(hipMalloc(&out, (size_t)M * N * 2));
    CHECK(hipMalloc(&scrub, scrub_bytes));
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    {   // fragment order: (column tile nt, K step ks, lane (r, h)) -> W[nt * 32 + r][ks * 16 + h * 8 .. + 7]
        std::vector<uint16_t> hP(hW.size());
        const int KS = K / 16;
        for (int nt = 0; nt < N / 32; ++nt)
            for (int ks = 0; ks < KS; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j)
                        hP[(((size_t)nt * KS + ks) * 64 + lane) * 8 + j] = hW[(size_t)(nt * 32 + (lane & 31)) * K + ks * 16 + (lane >> 5) * 8 + j];
#ifdef UNPACKED
        CHECK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
#else
        CHECK(hipMemcpy(W, hP.data(), hP.size() * 2, hipMemcpyHostToDevice));
#endif
    }
    CHECK(hipMemset(out, 0xFF, (size_t)M * N * 2));
    const u32x4* Wq = reinterpret_cast<const u32x4*>(W);
    float ms = K == 64 ? run<64>(A, Wq, out, M, N, 12, scrub, scrub_bytes) : K == 128 ? run<128>(A, Wq, out, M, N, 12, scrub, scrub_bytes)
                                                                                     : run<256>(A, Wq, out, M, N, 12, scrub, scrub_bytes);
    // check 96 rows (first, last, random) against a double-precision contraction of the same bf16 operands
    std::vector<uint16_t> row(N);
    double worst = 0;
    for (int t = 0; t < 96; ++t) {
        const int m = t < 32 ? t : (t < 64 ? M - 1 - (t - 32) : (int)(((uint64_t)rand() * 2654435761u) % M));
        CHECK(hipMemcpy(row.data(), out + (size_t)m * N, N * 2, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
            const double err = fabs(bf2f(row[n]) - ref) / (fabs(ref) * 0.0078125 + 1e-3);     // in units of one bf16 spacing (+ abs floor)
            if (err > worst) worst = err;
        }
    }
    const double bytes = ((double)M * K + (double)N * K + (double)M * N) * 2;
    printf("pw_probe M=%d K=%d N=%d: %.4f ms  = %.2f TB/s of algorithmic bytes (%.0f MB), %.0f TFLOP/s; worst error %.2f bf16 spacings %s\n",
           M, K, N, ms, bytes / ms * 1e-9, bytes * 1e-6, 2.0 * M * K * N / ms * 1e-9, worst, worst <= 1.01 ? "(ok)" : "(WRONG)");
    return worst <= 1.01 ? 0 : 2;
}
