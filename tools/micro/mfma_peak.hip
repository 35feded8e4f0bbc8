// Development microbenchmark: fp32 MFMA issue rate with 1 / 2 / 4 waves per SIMD (dependent chain per wave),
// optionally with a workgroup barrier every 16 MFMAs.  hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool BAR>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float x = a + threadIdx.x * 1e-6f, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[s % NACC], 0, 0, 0);
        if (BAR) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.f) out[0] = s;
}

template <int NACC, bool BAR>
void run(const char* name, int bpc, float* d) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, BAR>), dim3(256 * bpc), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = 256.0 * bpc * 4 * iters * 16 * 4096.0;
    printf("%-24s blocks/CU %d : %7.3f ms  %6.1f TFLOP/s\n", name, bpc, ms, flop / ms / 1e9);
}

int main() {
    float* d; hipMalloc(&d, 1024);
    for (int bpc : {1, 2, 4}) {
        run<1, false>("1 acc", bpc, d);
        run<4, false>("4 acc", bpc, d);
        run<1, true>("1 acc + barrier/16", bpc, d);
    }
    return 0;
}
