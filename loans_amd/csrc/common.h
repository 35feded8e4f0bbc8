// Shared device/host helpers for the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "loans_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 loans_bf16x4 __attribute__((ext_vector_type(4)));

// four consecutive elements of an activation tensor as fp32, whatever its storage type (float: 16 bytes, bf16: 8)
template <typename T> struct io4;
template <> struct io4<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct io4<__bf16> {
    static __device__ __forceinline__ f32x4 ld(const __bf16* p) {
        return __builtin_convertvector(*reinterpret_cast<const loans_bf16x4*>(p), f32x4);
    }
    static __device__ __forceinline__ void st(__bf16* p, f32x4 v) {      // round to nearest even
        *reinterpret_cast<loans_bf16x4*>(p) = __builtin_convertvector(v, loans_bf16x4);
    }
};

#define LOANS_LAUNCH_CHECK()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- per-device launcher state ---------------------------------------------------------------------------------------------
// "device = the calling thread's current HIP device" (include/loans_hip.h): whatever a launcher caches about the device is
// keyed by its ordinal, and the caches are lock-free, so that one process may drive several devices from several threads.
// hipGetDevice is a thread-local read; hipDeviceGetAttribute and hipFuncSetAttribute are not stream operations (neither
// blocks, both are legal while a stream is being captured), so a first call inside a hipGraph capture is safe.
#include <atomic>
constexpr int LOANS_MAX_DEVICES = 64;

// one bit per device: "this kernel's MaxDynamicSharedMemorySize attribute has been raised there" (the attribute is per
// device; a launch asking for > 64 KB of dynamic LDS fails on a device where it was never set)
struct loans_device_once {
    std::atomic<uint64_t> bits{0};
};

static inline int loans_raise_lds_limit(loans_device_once& once, const void* kern, size_t lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LOANS_MAX_DEVICES) return LOANS_EINVAL;
    if ((once.bits.load(std::memory_order_acquire) >> dev) & 1) return LOANS_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     // idempotent: a race sets it twice
    if (e != hipSuccess) return (int)e;
    once.bits.fetch_or(1ull << dev, std::memory_order_release);
    return LOANS_OK;
}

// compute units of the CURRENT device (0 on error), cached per ordinal
static inline int loans_device_cus() {
    static std::atomic<int> table[LOANS_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LOANS_MAX_DEVICES) return 0;
    int cus = table[dev].load(std::memory_order_relaxed);
    if (cus > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    table[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

// 64-lane wave reductions (wavefront = 64 on gfx950)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// hardware fp atomics (no CAS loop): global_atomic_add_f32 / _f64
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// Cache policy of the convolution kernels' output stores (round 3): large outputs are written NON-TEMPORAL (buffer-store aux
// bit 1), so that their lines do not push the weights and halo rows other blocks are still reading out of L2 / the Infinity
// Cache.  Timed alone and back to back a write-heavy layer gains 13-27 % (ResNet-50's 1x1 expansions 0.227 -> 0.181 ms) -- but
// most of that is the NEXT repetition finding its input still cached; inside the step, where every input was written just
// before by another kernel, it is worth 1.5-2 % on the ResNet-50 localizer (31.4 -> 30.9 ms on one box) and nothing
// measurable on the ResNet-18 configurations (tools/pointwise_probe.py for the solo numbers).
// LOANS_CONV_NT_MB: smallest output in MB that is written non-temporal (default 16; -1 = never).
#define LOANS_STORE_B128(v, rs, off, nt)                                              \
    do {                                                                              \
        if (nt) __builtin_amdgcn_raw_buffer_store_b128((v), (rs), (off), 0, 2);       \
        else __builtin_amdgcn_raw_buffer_store_b128((v), (rs), (off), 0, 0);          \
    } while (0)
static inline int loans_conv_nt(size_t out_bytes) {
    static const long mb = [] { const char* e = getenv("LOANS_CONV_NT_MB"); return e && *e ? atol(e) : 16L; }();
    return mb >= 0 && out_bytes >= ((size_t)mb << 20);
}

static inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// stem.hip: the direct 7x7 / 2 RGB stem convolution behind LOANS_TILE_STEM (internal; reached through loans_igemm_f32)
int loans_stem7_rows(int Ho, int Wo, int Wp3, size_t* lds_bytes);
int loans_stem7_launch(const float* in, const float* w, float* out, const float* bias, double* stats,
                       const loans_igemm_desc* d, hipStream_t st);
int loans_stem7_wgrad_launch(const float* x, const float* gy, float* dw, const loans_igemm_desc* d, hipStream_t st);
int loans_stem7_wgrad_bf16_slabs(const loans_igemm_desc* d);
int loans_stem7_wgrad_bf16_launch(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, float* ws, hipStream_t st);
int loans_stem7_bf16_rows(int Ho, int Wo, int Wp3, size_t* lds_bytes);
int loans_stem7_bf16_launch(const float* in, const float* w, void* out, const float* bias, double* stats,
                            const loans_igemm_desc* d, hipStream_t st);
int loans_stem7_bf16s_launch(const void* in, const void* w, void* out, const float* bias, double* stats,
                             const loans_igemm_desc* d, hipStream_t st);

// halo_bf16.hip: stride-1 convolutions on bf16 storage with the input tile staged once per channel chunk (LOANS_TILE_HALO_*;
// internal, reached through loans_igemm_bf16s)
int loans_halo16_covers(const loans_igemm_desc* d, int tile);
int loans_halo16_launch(const void* in, const void* w, void* out, const float* bias, double* stats, const void* ref,
                        const void* addend, const loans_igemm_desc* d, int tile, unsigned in_bytes, unsigned w_bytes,
                        unsigned out_bytes, hipStream_t st);

// pw_bf16.hip: 1 x 1 / 1 convolutions with Cin in {64, 128} on bf16 storage, operands never in LDS (LOANS_TILE_PW; internal,
// reached through loans_igemm_bf16s; `w` in loans_pw_pack_bf16's fragment order)
int loans_pw16_covers(const loans_igemm_desc* d);
int loans_pw16_launch(const void* in, const void* w, void* out, double* stats, const float* aff, const loans_igemm_desc* d, hipStream_t st);

// wgrad_halo_bf16.hip: weight gradient of stride-1 3 x 3 convolutions on bf16 storage with all taps in one block
// (LOANS_TILE_WGHALO_*; internal, reached through loans_wgrad_bf16s)
int loans_wgrad_halo16_covers(const loans_igemm_desc* d, int tile);
int loans_wgrad_halo16_slabs(const loans_igemm_desc* d, int tile, int splits);
int loans_wgrad_halo16_launch(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, int tile, int splits,
                              unsigned x_bytes, unsigned gy_bytes, float* ws, int* slabs, hipStream_t st);
