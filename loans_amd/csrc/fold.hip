// Deterministic fold of partial-sum slabs (round 5): dst[i] += sum over s of ws[s][i].
// The weight-gradient kernels of the bf16 arm used to close with one fp32 atomic per partial sum and block (wgrad_halo16_kernel:
// 512 blocks x 36 864 floats = 18.9 M atomic lanes per launch, 41-47 us whatever the layer, and a summation order that changes
// from run to run).  With a workspace they store their raw tiles as plain rows into slab `split`, and this pass adds the slabs up:
// thread (o, q) of a block sums the slabs q, q + Q, q + 2 Q, ... of ONE 16-byte unit o in that order, the Q partial sums of a
// unit are then added in the order q = 0, 1, ... through LDS by the q = 0 thread.  Q (a power of two <= 32 dividing 256) depends on
// (n, slabs) only, so the result is a function of the slabs' bits alone.
#include "common.h"

namespace {

template <int Q>
__global__ __launch_bounds__(256) void fold_slabs_kernel(const float* __restrict__ ws, float* __restrict__ dst, int64_t n4, int slabs) {
    constexpr int OB = 256 / Q;                 // 16-byte units per block
    __shared__ f32x4 red[Q > 1 ? 256 : 1];
    const int ol = threadIdx.x % OB, q = threadIdx.x / OB;
    const int64_t o = (int64_t)blockIdx.x * OB + ol;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (o < n4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(ws) + o;
        int s = q;
        for (; s + 3 * Q < slabs; s += 4 * Q) {         // four loads in flight, added in slab order
            const f32x4 a = __builtin_nontemporal_load(p + (int64_t)s * n4);
            const f32x4 b = __builtin_nontemporal_load(p + (int64_t)(s + Q) * n4);
            const f32x4 c = __builtin_nontemporal_load(p + (int64_t)(s + 2 * Q) * n4);
            const f32x4 d = __builtin_nontemporal_load(p + (int64_t)(s + 3 * Q) * n4);
            acc += a; acc += b; acc += c; acc += d;
        }
        for (; s < slabs; s += Q) acc += __builtin_nontemporal_load(p + (int64_t)s * n4);
    }
    if (Q > 1) {
        red[threadIdx.x] = acc;
        __syncthreads();
        if (q != 0 || o >= n4) return;
#pragma unroll
        for (int k = 1; k < Q; ++k) acc += red[k * OB + ol];
    } else if (o >= n4) {
        return;
    }
    f32x4* d4 = reinterpret_cast<f32x4*>(dst) + o;
    *d4 = *d4 + acc;
}

template <int Q>
int launch_fold(const float* ws, float* dst, int64_t n4, int slabs, hipStream_t st) {
    constexpr int OB = 256 / Q;
    const int64_t nblk = (n4 + OB - 1) / OB;
    if (nblk >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(fold_slabs_kernel<Q>, dim3((unsigned)nblk), dim3(256), 0, st, ws, dst, n4, slabs);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

}  // namespace

extern "C" int loans_fold_slabs_f32(const float* ws, float* dst, int64_t n, int32_t slabs, void* stream) {
    if (!ws || !dst || n <= 0 || (n & 3) || slabs < 1) return LOANS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(dst)) & 15) return LOANS_EINVAL;
    const int64_t n4 = n / 4;
    // threads across the slabs of one unit: enough of them to have ~256 k threads in flight, at least four slabs each
    int q = 1;
    while (q < 32 && n4 * q < (int64_t)262144 && slabs >= 8 * q) q *= 2;
    hipStream_t st = as_stream(stream);
    switch (q) {
        case 1: return launch_fold<1>(ws, dst, n4, slabs, st);
        case 2: return launch_fold<2>(ws, dst, n4, slabs, st);
        case 4: return launch_fold<4>(ws, dst, n4, slabs, st);
        case 8: return launch_fold<8>(ws, dst, n4, slabs, st);
        case 16: return launch_fold<16>(ws, dst, n4, slabs, st);
        default: return launch_fold<32>(ws, dst, n4, slabs, st);
    }
}
