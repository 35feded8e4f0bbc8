// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
//   out[m][n] = sum_k A[m][k] * Wp[n][k],   k = (tap, channel)
//
// A is never materialised: each 32-wide K chunk of a BM x 32 tile is gathered from the NHWC
// activation (16-byte loads along C), staged through LDS and consumed by 32x32x2 MFMAs.  The
// same kernel is the forward convolution and the data gradient (the latter with the gradient
// as the gathered tensor, re-packed weights and one launch per stride-parity class), see
// include/loans_hip.h for the problem descriptor.
//
// Tiling (4 waves = 256 threads, one wave per SIMD, 2 blocks per CU by LDS):
//   block tile BM x BN, wave tile (BM/WM) x (BN/WN) = TM x TN MFMA tiles of 32x32
//   LDS rows are [row][32 + 4] floats: 16-byte aligned and ds_read_b128 conflict-free
//   fragment trick: lane (r, h) reads k = 8g + 4h .. +3 as ONE b128 for A and for B; MFMA
//   step j contracts k = 8g+j (h = 0) with 8g+4+j (h = 1) on both operands, so four MFMAs per
//   tile consume one b128 per operand.
//   register-staged double buffering: global loads of chunk c+1 are issued before the 64
//   MFMAs of chunk c and written to the other LDS buffer after them (one barrier per chunk).
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;

struct IgemmArgs {
    const float* in;
    const float* w;
    float* out;
    const float* bias;
    double* stats;
    const float* ref;
    const float* addend;
    loans_igemm_desc d;
    int M, Ktot, nchunks, tiles_m, tiles_n;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// XCD-aware, bijective block remap: blocks that share an XCD (id % 8) get a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int id, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RA = BM / 32, RB = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);        // [2][BM][LDK]
    float* Bs = As + 2 * BM * LDK;                     // [2][BN][LDK]
    int* taps = reinterpret_cast<int*>(Bs + 2 * BN * LDK);
    int* opix = taps + LOANS_MAX_TAPS;                 // [BM] output pixel index, -1 = invalid row

    const loans_igemm_desc& d = a.d;
    const int tid = threadIdx.x;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = logical % a.tiles_n;
    const int tm = logical / a.tiles_n;
    const int lu = tid & 7, lrow = tid >> 3;
    const bool relu_in = d.flags & LOANS_F_RELU_IN;

    if (tid < LOANS_MAX_TAPS) {
        const int t = tid < d.ntaps ? tid : 0;
        taps[tid] = (int(d.dy[t]) << 16) | (int(d.dx[t]) & 0xffff);
    }

    int rowbase[RA], iy0[RA], ix0[RA];
    {
        const int gHW = d.gridH * d.gridW;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = tm * BM + lrow + 32 * i;
            int pix = -1;
            if (m < a.M) {
                const int b = m / gHW;
                const int rem = m - b * gHW;
                const int y = rem / d.gridW;
                const int x = rem - y * d.gridW;
                iy0[i] = y * d.isy;
                ix0[i] = x * d.isx;
                rowbase[i] = ((b * d.inH + iy0[i]) * d.inW + ix0[i]) * d.Cin;
                pix = (b * d.outH + y * d.osy + d.oy0) * d.outW + x * d.osx + d.ox0;
            } else {
                iy0[i] = -(1 << 20);
                ix0[i] = -(1 << 20);
                rowbase[i] = 0;
            }
            if (lu == 0) opix[lrow + 32 * i] = pix;
        }
    }
    __syncthreads();

    const int cpt = d.Cin >> 2;   // float4 units per tap
    int tap = lu / cpt, c4 = lu - tap * cpt;
    const int nbase = tn * BN + lrow;

    f32x4 ra[RA], rb[RB];
    auto load_chunk = [&](int c) {
        const bool tv = tap < d.ntaps;
        const int tp = taps[tv ? tap : 0];
        const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
        const int toff = (dy * d.inW + dx) * d.Cin + c4 * 4;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            const bool ok = tv && (unsigned)iy < (unsigned)d.inH && (unsigned)ix < (unsigned)d.inW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = ld4(a.in + (int64_t)(rowbase[i] + toff));
            if (relu_in) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            ra[i] = v;
        }
        const int kidx = (c * 8 + lu) * 4;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = nbase + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < d.Cout && kidx < a.Ktot) v = ld4(a.w + (int64_t)n * a.Ktot + kidx);
            rb[i] = v;
        }
        // advance (tap, c4) by one chunk = 8 units
        c4 += 8;
        while (c4 >= cpt) { c4 -= cpt; ++tap; }
    };
    auto store_chunk = [&](int buf) {
        float* Ab = As + buf * BM * LDK;
        float* Bb = Bs + buf * BN * LDK;
#pragma unroll
        for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(Ab + (lrow + 32 * i) * LDK + lu * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(Bb + (lrow + 32 * i) * LDK + lu * 4) = rb[i];
    };

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int c = 0; c < a.nchunks; ++c) {
        const int buf = c & 1;
        const bool more = (c + 1) < a.nchunks;
        if (more) load_chunk(c + 1);
        const float* Ab = As + buf * BM * LDK + (wm * TM * 32 + r) * LDK + h * 4;
        const float* Bb = Bs + buf * BN * LDK + (wn * TN * 32 + r) * LDK + h * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK + g * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK + g * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    const bool f_bias = d.flags & LOANS_F_BIAS, f_stats = d.flags & LOANS_F_STATS;
    const bool f_mask = d.flags & LOANS_F_MASK, f_add = d.flags & LOANS_F_ADDEND;
    const bool f_addmask = d.flags & LOANS_F_ADDEND_MASK;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = tn * BN + wn * TN * 32 + j * 32 + r;
        const bool cok = col < d.Cout;
        const float bv = (f_bias && cok) ? a.bias[col] : 0.f;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int pix = opix[rl];
                if (pix >= 0 && cok) {
                    float v = acc[i][j][e] + bv;
                    s += v;
                    q += v * v;
                    const int64_t off = (int64_t)pix * d.Cout + col;
                    if (f_mask) v = a.ref[off] > 0.f ? v : 0.f;
                    if (f_add) {
                        const float ad = a.addend[off];
                        v += (!f_addmask || a.ref[off] > 0.f) ? ad : 0.f;
                    }
                    a.out[off] = v;
                }
            }
        }
        if (f_stats) {
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (h == 0 && cok) {
                atomic_add_f64(a.stats + col, (double)s);
                atomic_add_f64(a.stats + d.Cout + col, (double)q);
            }
        }
    }
}

template <int BM, int BN>
constexpr size_t igemm_lds_bytes() {
    return (size_t)(2 * BM * LDK + 2 * BN * LDK) * 4 + LOANS_MAX_TAPS * 4 + BM * 4;
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(IgemmArgs& a, hipStream_t st) {
    static bool attr_set = false;
    constexpr size_t lds = igemm_lds_bytes<BM, BN>();
    auto kern = igemm_kernel<BM, BN, WM, WN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.d.Cout + BN - 1) / BN;
    const int nblk = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

int check_desc(const loans_igemm_desc* d) {
    if (!d) return LOANS_EINVAL;
    if (d->B <= 0 || d->inH <= 0 || d->inW <= 0 || d->Cin <= 0 || (d->Cin & 3)) return LOANS_EINVAL;
    if (d->outH <= 0 || d->outW <= 0 || d->Cout <= 0) return LOANS_EINVAL;
    if (d->gridH <= 0 || d->gridW <= 0 || d->osy <= 0 || d->osx <= 0 || d->isy <= 0 || d->isx <= 0) return LOANS_EINVAL;
    if (d->oy0 < 0 || d->ox0 < 0) return LOANS_EINVAL;
    if ((d->gridH - 1) * d->osy + d->oy0 >= d->outH) return LOANS_EINVAL;
    if ((d->gridW - 1) * d->osx + d->ox0 >= d->outW) return LOANS_EINVAL;
    if (d->ntaps < 1 || d->ntaps > LOANS_MAX_TAPS) return LOANS_EINVAL;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)d->B * d->inH * d->inW * d->Cin >= lim) return LOANS_ERANGE;
    if ((int64_t)d->B * d->outH * d->outW * d->Cout >= lim) return LOANS_ERANGE;
    if ((int64_t)d->B * d->gridH * d->gridW >= lim) return LOANS_ERANGE;
    if ((int64_t)d->ntaps * d->Cin * d->Cout >= lim) return LOANS_ERANGE;
    return LOANS_OK;
}

}  // namespace

extern "C" int loans_igemm_f32(const float* in, const float* w, float* out, const float* bias, double* stats,
                               const float* ref, const float* addend, const loans_igemm_desc* d, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!in || !w || !out) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_BIAS) && !bias) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_STATS) && !stats) return LOANS_EINVAL;
    if ((d->flags & (LOANS_F_MASK | LOANS_F_ADDEND_MASK)) && !ref) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_ADDEND_MASK) && !(d->flags & LOANS_F_ADDEND)) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_ADDEND) && !addend) return LOANS_EINVAL;
    IgemmArgs a;
    a.in = in; a.w = w; a.out = out; a.bias = bias; a.stats = stats; a.ref = ref; a.addend = addend;
    a.d = *d;
    a.M = d->B * d->gridH * d->gridW;
    a.Ktot = d->ntaps * d->Cin;
    a.nchunks = (a.Ktot + BK - 1) / BK;
    hipStream_t st = as_stream(stream);
    int tile = d->tile;
    if (tile == 0) {
        if (d->Cout <= 64) {
            tile = LOANS_TILE_128x64;
        } else {
            // prefer the big tile when it still gives >= 2 blocks per CU of work
            const int64_t big = (int64_t)((a.M + 127) / 128) * ((d->Cout + 127) / 128);
            tile = big >= 1024 ? LOANS_TILE_128x128 : LOANS_TILE_128x64;
            if ((int64_t)((a.M + 127) / 128) * ((d->Cout + 63) / 64) < 512) tile = LOANS_TILE_64x64;
        }
    }
    switch (tile) {
        case LOANS_TILE_128x128: return launch_igemm<128, 128, 2, 2>(a, st);
        case LOANS_TILE_128x64: return launch_igemm<128, 64, 2, 2>(a, st);
        case LOANS_TILE_64x64: return launch_igemm<64, 64, 2, 2>(a, st);
        case LOANS_TILE_256x64: return launch_igemm<256, 64, 4, 1>(a, st);
        default: return LOANS_EINVAL;
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient:  dw[co][t][c] += sum_m gy[opix(m)][co] * x[pix(m,t)][c]
// GEMM rows = co, columns = (t,c), reduction = m (split over blocks, fp32 atomics into dw).
// Both operands are staged k-major ([m][row], row contiguous) straight from NHWC memory and
// read as MFMA fragments with ds_read_b32 (32 consecutive lanes -> 32 consecutive banks).
// ------------------------------------------------------------------------------------------
namespace {

struct WgradArgs {
    const float* x;
    const float* gy;
    float* dw;
    loans_igemm_desc d;
    int M, Ktot, tiles_co, tiles_j, splits, chunks_per_split;
};

template <int BT>   // square BT x BT block tile, 4 waves as 2 x 2
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    constexpr int T = BT / 2 / 32;          // MFMA tiles per wave per dim
    constexpr int UPR = BT / 4;             // float4 units per LDS row
    constexpr int RPP = 256 / UPR;          // rows per loader pass
    constexpr int NP = 32 / RPP;            // passes per 32-row chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ys = reinterpret_cast<float*>(smem);     // [2][32][BT]
    float* Xs = Ys + 2 * 32 * BT;                   // [2][32][BT]

    const loans_igemm_desc& d = a.d;
    const int tid = threadIdx.x;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = a.tiles_co * a.tiles_j;
    const int split = logical / ntile;
    const int tile = logical - split * ntile;
    const int tco = tile % a.tiles_co;
    const int tj = tile / a.tiles_co;
    const int unit = tid % UPR, prow = tid / UPR;

    // this thread's fixed column of the X tile: (tap, c4)
    const int cpt = d.Cin >> 2;
    const int ug = tj * UPR + unit;
    const int xtap = ug / cpt;
    const int xc4 = ug - xtap * cpt;
    const bool xtv = xtap < d.ntaps;
    const int dy = xtv ? (int)d.dy[xtap] : 0;
    const int dx = xtv ? (int)d.dx[xtap] : 0;
    const int toff = (dy * d.inW + dx) * d.Cin + xc4 * 4;
    const int yco = tco * BT + unit * 4;
    const bool yv = yco < d.Cout;        // Cout is a multiple of 4 for every layer on this path

    const int gHW = d.gridH * d.gridW;
    const bool relu_in = d.flags & LOANS_F_RELU_IN;
    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    const int total_chunks = (a.M + 31) / 32;
    if (c_end > total_chunks) c_end = total_chunks;

    f32x4 ry[NP], rx[NP];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int m = c * 32 + prow + RPP * p;
            f32x4 vy = {0.f, 0.f, 0.f, 0.f}, vx = {0.f, 0.f, 0.f, 0.f};
            if (m < a.M) {
                const int b = m / gHW;
                const int rem = m - b * gHW;
                const int y = rem / d.gridW;
                const int x = rem - y * d.gridW;
                if (yv) {
                    const int pix = (b * d.outH + y * d.osy + d.oy0) * d.outW + x * d.osx + d.ox0;
                    vy = ld4(a.gy + (int64_t)pix * d.Cout + yco);
                }
                const int iy = y * d.isy + dy, ix = x * d.isx + dx;
                if (xtv && (unsigned)iy < (unsigned)d.inH && (unsigned)ix < (unsigned)d.inW)
                    vx = ld4(a.x + (int64_t)(((b * d.inH + y * d.isy) * d.inW + x * d.isx) * d.Cin + toff));
                if (relu_in) {
                    vx.x = fmaxf(vx.x, 0.f); vx.y = fmaxf(vx.y, 0.f); vx.z = fmaxf(vx.z, 0.f); vx.w = fmaxf(vx.w, 0.f);
                }
            }
            ry[p] = vy;
            rx[p] = vx;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = prow + RPP * p;
            *reinterpret_cast<f32x4*>(Ys + (buf * 32 + row) * BT + unit * 4) = ry[p];
            *reinterpret_cast<f32x4*>(Xs + (buf * 32 + row) * BT + unit * 4) = rx[p];
        }
    };

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        const bool more = (c + 1) < c_end;
        if (more) load_chunk(c + 1);
        const float* Yb = Ys + buf * 32 * BT + wm * T * 32 + r;
        const float* Xb = Xs + buf * 32 * BT + wn * T * 32 + r;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float af[T], bf[T];
#pragma unroll
            for (int i = 0; i < T; ++i) af[i] = Yb[(2 * s + h) * BT + i * 32];
#pragma unroll
            for (int j = 0; j < T; ++j) bf[j] = Xb[(2 * s + h) * BT + j * 32];
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int jc = tj * BT + wn * T * 32 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = tco * BT + wm * T * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < d.Cout && jc < a.Ktot) atomic_add_f32(a.dw + (int64_t)co * a.Ktot + jc, acc[i][j][e]);
            }
        }
}

template <int BT>
int launch_wgrad(WgradArgs& a, int splits_req, hipStream_t st) {
    static bool attr_set = false;
    constexpr size_t lds = (size_t)4 * 32 * BT * 4;
    auto kern = wgrad_kernel<BT>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    a.tiles_co = (a.d.Cout + BT - 1) / BT;
    a.tiles_j = (a.Ktot + BT - 1) / BT;
    const int total_chunks = (a.M + 31) / 32;
    int splits = splits_req;
    if (splits <= 0) {
        const int ntile = a.tiles_co * a.tiles_j;
        splits = (1024 + ntile - 1) / ntile;            // ~4 blocks per CU in flight
        const int max_splits = (total_chunks + 7) / 8;   // >= 8 chunks (256 rows) per block
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    if (splits > total_chunks) splits = total_chunks;
    a.chunks_per_split = (total_chunks + splits - 1) / splits;
    a.splits = (total_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
    const int nblk = a.tiles_co * a.tiles_j * a.splits;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

struct RepackArgs {
    const float* src;
    float* dst;
    int Cout, Cin, src_taps, ntaps;
    int tapsel[LOANS_MAX_TAPS];
};

// dst[ci][t][co] = src[co][tapsel[t]][ci] through a 32x33 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void repack_dgrad_kernel(const RepackArgs a) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int st = a.tapsel[t];
    for (int k = ty; k < 32; k += 8) {
        const int co = co0 + k, ci = ci0 + tx;
        tile[k][tx] = (co < a.Cout && ci < a.Cin) ? a.src[((int64_t)co * a.src_taps + st) * a.Cin + ci] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int ci = ci0 + k, co = co0 + tx;
        if (ci < a.Cin && co < a.Cout) a.dst[((int64_t)ci * a.ntaps + t) * a.Cout + co] = tile[tx][k];
    }
}

}  // namespace

extern "C" int loans_wgrad_f32(const float* x, const float* gy, float* dw, const loans_igemm_desc* d,
                               int32_t splits, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !gy || !dw || (d->Cout & 3)) return LOANS_EINVAL;
    WgradArgs a;
    a.x = x; a.gy = gy; a.dw = dw; a.d = *d;
    a.M = d->B * d->gridH * d->gridW;
    a.Ktot = d->ntaps * d->Cin;
    hipStream_t st = as_stream(stream);
    const bool small = (d->Cout <= 64) || (a.Ktot <= 64);
    int tile = d->tile;
    if (tile == 0) tile = small ? LOANS_TILE_64x64 : LOANS_TILE_128x128;
    if (tile == LOANS_TILE_64x64) return launch_wgrad<64>(a, splits, st);
    if (tile == LOANS_TILE_128x128) return launch_wgrad<128>(a, splits, st);
    return LOANS_EINVAL;
}

extern "C" int loans_repack_dgrad_f32(const float* src, float* dst, int32_t Cout, int32_t Cin, int32_t src_taps,
                                      const int32_t* tapsel_host, int32_t ntaps, void* stream) {
    if (!src || !dst || !tapsel_host || Cout <= 0 || Cin <= 0 || src_taps <= 0) return LOANS_EINVAL;
    if (ntaps < 1 || ntaps > LOANS_MAX_TAPS) return LOANS_EINVAL;
    RepackArgs a;
    a.src = src; a.dst = dst; a.Cout = Cout; a.Cin = Cin; a.src_taps = src_taps; a.ntaps = ntaps;
    for (int i = 0; i < ntaps; ++i) {
        if (tapsel_host[i] < 0 || tapsel_host[i] >= src_taps) return LOANS_EINVAL;
        a.tapsel[i] = tapsel_host[i];
    }
    dim3 grid((Cout + 31) / 32, (Cin + 31) / 32, ntaps);
    hipLaunchKernelGGL(repack_dgrad_kernel, grid, dim3(256), 0, as_stream(stream), a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}
