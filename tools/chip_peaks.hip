// Microbenchmark (not part of the product): the MFMA rate the chip sustains with nothing else in the way -- every wave of a
// full grid issues independent v_mfma_f32_32x32x2_f32 (or v_mfma_f32_32x32x16_bf16) back to back from registers -- and the
// HBM write / copy rates of a streaming kernel.  The ceilings the conv kernels' TFLOP/s and GB/s are read against.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_f32_kernel(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_bf16_kernel(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)((float)threadIdx.x * 1e-3f); b[e] = (__bf16)(1.0f + e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void fill_kernel(f32x4* dst, size_t n4) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = v;
}
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* src, f32x4* dst, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read_kernel(const f32x4* src, float* out, size_t n4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) s += src[i];
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}

extern "C" int peak_mfma(int kind, int waves_per_simd, int iters, float* scratch, void* stream) {
    // grid: 256 CUs x (waves_per_simd) blocks of 4 waves
    hipDeviceProp_t p;
    int dev = 0;
    hipGetDevice(&dev);
    hipGetDeviceProperties(&p, dev);
    dim3 grid(p.multiProcessorCount * waves_per_simd), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(mfma_f32_kernel<4>, grid, blk, 0, st, scratch, iters);
    else if (kind == 1) hipLaunchKernelGGL(mfma_bf16_kernel<4>, grid, blk, 0, st, scratch, iters);
    else return -1;
    return (int)hipGetLastError();
}
extern "C" int peak_mem(int kind, void* a, void* b, size_t bytes, int blocks, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const size_t n4 = bytes / 16;
    if (kind == 0) hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, st, (f32x4*)a, n4);
    else if (kind == 1) hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, st, (const f32x4*)a, (f32x4*)b, n4);
    else if (kind == 2) hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, st, (const f32x4*)a, (float*)b, n4);
    else return -1;
    return (int)hipGetLastError();
}
