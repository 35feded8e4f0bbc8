// Microbenchmark (not part of the product): which ACCESS PATTERN streams fastest on this chip for the shapes the BN passes
// have -- R reads + W writes of equally sized bf16 tensors, one fused multiply-add per element in between.
//   pattern 0: persistent blocks, grid-stride over 16-byte units, UNR units in flight (what bn_pool.hip launched in round 3)
//   pattern 1: one short-lived block per 256 x UNR units, its units 4 KiB apart (the shape of torch's vectorized kernels)
//   pattern 2: persistent blocks, each walking its own CONTIGUOUS slab, UNR units in flight
// each with default / non-temporal stores / non-temporal loads and stores.
// build: hipcc -O3 --offload-arch=gfx950 -o stream_patterns tools/stream_patterns.hip ; run: ./stream_patterns [MiB per tensor]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ u32x4 ldu(const u32x4* p) {
    if (NT & 2) return __builtin_nontemporal_load(p);
    return *p;
}
template <int NT> __device__ __forceinline__ void stu(u32x4* p, u32x4 v) {
    if (NT & 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ void unpack(u32x4 v, f32x4& lo, f32x4& hi) {
    lo = f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xFFFF0000u)};
    hi = f32x4{__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xFFFF0000u), __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xFFFF0000u)};
}
__device__ __forceinline__ unsigned pk(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ u32x4 pack(f32x4 lo, f32x4 hi) {
    return u32x4{pk(lo.x, lo.y), pk(lo.z, lo.w), pk(hi.x, hi.y), pk(hi.z, hi.w)};
}

// out = k1 * a + k2 * b + k3 on the units' 8 bf16 values (R = 2), out = k1 * a + k3 (R = 1), out = k3 (R = 0)
template <int R, int NT>
__device__ __forceinline__ void unit(const u32x4* a, const u32x4* b, u32x4* o, int64_t i, float k1, float k2, float k3) {
    f32x4 lo = {k3, k3, k3, k3}, hi = lo;
    if (R >= 1) { f32x4 l, h; unpack(ldu<NT>(a + i), l, h); lo += k1 * l; hi += k1 * h; }
    if (R >= 2) { f32x4 l, h; unpack(ldu<NT>(b + i), l, h); lo += k2 * l; hi += k2 * h; }
    stu<NT>(o + i, pack(lo, hi));
}

template <int R, int NT, int UNR>
__global__ __launch_bounds__(256) void pat0(const u32x4* a, const u32x4* b, u32x4* o, int64_t n, float k1, float k2, float k3) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
        u32x4 va[UNR], vb[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (R >= 1) va[u] = ldu<NT>(a + i + u * stride);
            if (R >= 2) vb[u] = ldu<NT>(b + i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            f32x4 lo = {k3, k3, k3, k3}, hi = lo, l, h;
            if (R >= 1) { unpack(va[u], l, h); lo += k1 * l; hi += k1 * h; }
            if (R >= 2) { unpack(vb[u], l, h); lo += k2 * l; hi += k2 * h; }
            stu<NT>(o + i + u * stride, pack(lo, hi));
        }
    }
    for (; i < n; i += stride) unit<R, NT>(a, b, o, i, k1, k2, k3);
}

template <int R, int NT, int UNR>
__global__ __launch_bounds__(256) void pat1(const u32x4* a, const u32x4* b, u32x4* o, int64_t n, float k1, float k2, float k3) {
    const int64_t base = (int64_t)blockIdx.x * 256 * UNR + threadIdx.x;
    u32x4 va[UNR], vb[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int64_t i = base + u * 256;
        if (i < n) {
            if (R >= 1) va[u] = ldu<NT>(a + i);
            if (R >= 2) vb[u] = ldu<NT>(b + i);
        }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int64_t i = base + u * 256;
        if (i < n) {
            f32x4 lo = {k3, k3, k3, k3}, hi = lo, l, h;
            if (R >= 1) { unpack(va[u], l, h); lo += k1 * l; hi += k1 * h; }
            if (R >= 2) { unpack(vb[u], l, h); lo += k2 * l; hi += k2 * h; }
            stu<NT>(o + i, pack(lo, hi));
        }
    }
}

template <int R, int NT, int UNR>
__global__ __launch_bounds__(256) void pat2(const u32x4* a, const u32x4* b, u32x4* o, int64_t n, float k1, float k2, float k3) {
    const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;      // units per block, a multiple of 256
    const int64_t lo_ = (int64_t)blockIdx.x * per, hi_ = lo_ + per < n ? lo_ + per : n;
    int64_t i = lo_ + threadIdx.x;
    for (; i + (UNR - 1) * 256 < hi_; i += UNR * 256) {
        u32x4 va[UNR], vb[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (R >= 1) va[u] = ldu<NT>(a + i + u * 256);
            if (R >= 2) vb[u] = ldu<NT>(b + i + u * 256);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            f32x4 lo = {k3, k3, k3, k3}, hi = lo, l, h;
            if (R >= 1) { unpack(va[u], l, h); lo += k1 * l; hi += k1 * h; }
            if (R >= 2) { unpack(vb[u], l, h); lo += k2 * l; hi += k2 * h; }
            stu<NT>(o + i + u * 256, pack(lo, hi));
        }
    }
    for (; i < hi_; i += 256) unit<R, NT>(a, b, o, i, k1, k2, k3);
}

// read-only (W = 0): sum into a register, one conditional store
template <int NT, int UNR>
__global__ __launch_bounds__(256) void read0(const u32x4* a, u32x4* o, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    u32x4 s = {0, 0, 0, 0};
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) s ^= ldu<NT>(a + i + u * stride);
    }
    for (; i < n; i += stride) s ^= ldu<NT>(a + i);
    if ((s.x ^ s.y ^ s.z ^ s.w) == 0x12345678u) o[0] = s;
}

static float time_ms(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename F> static float run(F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 9; ++r) {
        CHECK(hipEventRecord(e0, 0));
        launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    CHECK(hipGetLastError());
    return time_ms(t);
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 512;
    const size_t bytes = mib << 20;
    const int64_t n = (int64_t)(bytes / 16);
    u32x4 *a, *b, *o;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&o, bytes));
    CHECK(hipMemset(a, 0x3c, bytes)); CHECK(hipMemset(b, 0x3d, bytes));
    printf("tensor %zu MiB each; TB/s = (R + W) x size / time\n", mib);
    const float k1 = 0.5f, k2 = 0.25f, k3 = 1.f;
#define ROW(label, R, expr) do { const float ms = run([&] { expr; }); printf("  %-58s %.3f ms  %.2f TB/s\n", label, ms, (double)((R) + 1) * bytes / ms * 1e-9); fflush(stdout); } while (0)
    char lab[128];
    {
        float ms = run([&] { CHECK(hipMemsetAsync(o, 0, bytes, 0)); });
        printf("  %-58s %.3f ms  %.2f TB/s\n", "hipMemsetAsync", ms, (double)bytes / ms * 1e-9);
        ms = run([&] { CHECK(hipMemcpyAsync(o, a, bytes, hipMemcpyDeviceToDevice, 0)); });
        printf("  %-58s %.3f ms  %.2f TB/s\n", "hipMemcpyAsync D2D", ms, 2.0 * bytes / ms * 1e-9);
    }
    for (int g : {1024, 2048, 4096}) {
        const float ms = run([&] { hipLaunchKernelGGL((read0<0, 4>), dim3(g), dim3(256), 0, 0, a, o, n); });
        printf("  read only, persistent grid-stride x4, %5d blocks           %.3f ms  %.2f TB/s\n", g, ms, (double)bytes / ms * 1e-9);
    }
#define P0(R, NT, UNR, G) snprintf(lab, sizeof lab, "R%d W1 pat0 grid-stride  unr %d nt %d blocks %6d", R, UNR, NT, G); ROW(lab, R, hipLaunchKernelGGL((pat0<R, NT, UNR>), dim3(G), dim3(256), 0, 0, a, b, o, n, k1, k2, k3))
#define P1(R, NT, UNR) { const int G = (int)((n + 256 * UNR - 1) / (256 * UNR)); snprintf(lab, sizeof lab, "R%d W1 pat1 short blocks unr %d nt %d blocks %6d", R, UNR, NT, G); ROW(lab, R, hipLaunchKernelGGL((pat1<R, NT, UNR>), dim3(G), dim3(256), 0, 0, a, b, o, n, k1, k2, k3)); }
#define P2(R, NT, UNR, G) snprintf(lab, sizeof lab, "R%d W1 pat2 contiguous   unr %d nt %d blocks %6d", R, UNR, NT, G); ROW(lab, R, hipLaunchKernelGGL((pat2<R, NT, UNR>), dim3(G), dim3(256), 0, 0, a, b, o, n, k1, k2, k3))
#define SWEEP(R) \
    P0(R, 0, 2, 1024); P0(R, 0, 2, 2048); P0(R, 0, 2, 4096); P0(R, 0, 2, 8192); P0(R, 0, 4, 1024); P0(R, 0, 4, 2048); P0(R, 0, 4, 4096); \
    P0(R, 1, 2, 4096); P0(R, 3, 2, 4096); P0(R, 1, 4, 2048); P0(R, 3, 4, 2048); \
    P1(R, 0, 1); P1(R, 0, 2); P1(R, 0, 4); P1(R, 0, 8); P1(R, 1, 4); P1(R, 3, 4); P1(R, 1, 8); P1(R, 3, 8); \
    P2(R, 0, 2, 2048); P2(R, 0, 4, 1024); P2(R, 0, 4, 2048); P2(R, 0, 4, 4096); P2(R, 0, 4, 8192); P2(R, 1, 4, 2048); P2(R, 3, 4, 2048); P2(R, 3, 4, 8192)
    SWEEP(0);
    SWEEP(1);
    SWEEP(2);
    return 0;
}
