// Weight gradient of a stride-1 3 x 3 convolution on bf16 tensors with ALL nine taps in one block
// (3 x 3 only; LOANS_TILE_WGHALO_64 / _128 of loans_wgrad_bf16s; replaces F.convolution_2d's backward-filter of sheep/resnet.py:121-160,
// common/net.py:15-65 for those layers).
//
//     dw[co][t][c] += sum over pixels p of  gy[p][co] * x[p + tap(t)][c]
//
// wgrad16_kernel (igemm_bf16.hip) treats this as a plain GEMM with (t, c) as columns: every column tile gathers ITS tap's
// shifted copy of x and re-reads gy, so a 3x3 layer pulls x nine times and gy K / BJ times through L2 -> LDS (4.8 GB for one
// res2 layer of configs[2], whose tensors are 0.54 GB) at 32-64 FLOP per staged byte -- 2-3 x the time of the forward conv.
// Here a block owns BCO output channels x 64 input channels x all k*k taps and walks 8 x 16 pixel tiles of the images: per
// tile it stages the gradient tile [128 px][BCO] and the input HALO tile [10 x 18 px][64] ONCE; a tap is a shifted window of
// the halo image (an LDS address offset), the reduction index of the MFMA is the pixel.  One staged byte now feeds
// 2 * 9 * BCO * 64 * 128 / (128 * BCO * 2 + 180 * 128) FLOP = 250-335 (the MFMA, not the L2 -> LDS path, bounds the block),
// x crosses L2 -> LDS 1.4 times (the halo) per BCO-channel tile of the gradient instead of nine times.
//
// Operands are pixel-major in memory (channels contiguous) = transposed for the MFMA: tiles are staged as they lie, rows
// padded by 32 elements (192 / 320-byte strides: the rows a 16-lane group touches start 16 banks apart), fragments come
// from ds_read_b64_tr_b16 exactly as in wgrad16_kernel.  A wave holds one 32 x 32 tile per tap = 9 x 16 accumulator
// registers; per halo row it reads 3 input fragments (one per horizontal tap) and 1 gradient fragment for up to 9 MFMAs.
// Global loads of tile n + 1 are issued before the MFMAs of tile n and land in registers (the LDS image is single: a
// block's ~2300-4600 MFMA cycles per tile hide the two short LDS-write phases around its barriers).
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// development only (tools/wgrad_ablate.sh): -DLOANS_WGH_DBG=bits builds the kernel without its atomics (1), MFMAs (2),
// global loads after the first tile (4), fragment reads and MFMAs (8); 0 in the library
#ifndef LOANS_WGH_DBG
#define LOANS_WGH_DBG 0
#endif
constexpr int WGH_DBG = LOANS_WGH_DBG;

constexpr int TH = 8, TW = 16;                  // output pixels per tile
constexpr int HH = TH + 2, HW = TW + 2;         // halo image (k <= 3)
constexpr int BC = 64;                          // input channels per block
constexpr int SX = BC + 32;                     // padded LDS row strides (elements)

struct WgHaloArgs {
    const __bf16* x;
    const __bf16* gy;
    float* dw;
    int B, H, W, Cin, Cout;         // stride 1: input and output share H x W
    int dy0, dx0;                   // taps (dy0 + i, dx0 + j), i, j < 3, row-major = the weight's tap order
    int tiles_y, tiles_x, ntiles;   // pixel tiles per image column / row, in all
    int pairs_co, pairs_c;          // channel tile grid
    int tiles_per_block;
    unsigned x_bytes, gy_bytes;
    // partial slabs (loans_wgrad_bf16s_ws): block (pair, split) STORES its raw tile into slab `split` of `ws` ([splits][Cout][9 Cin],
    // the layout of dw) instead of adding it to dw with atomics; loans_fold_slabs_f32 then sums the slabs in a fixed order
    float* ws;
    int64_t slab;
};

__device__ __forceinline__ int xcd_remap(int id, int nblk) {
    // consecutive logical ids on one XCD (hardware deals blocks round-robin over the 8 XCDs): the channel-tile pairs that
    // share a pixel range then share an L2
    const int per = nblk >> 3;
    if (per == 0 || (nblk & 7)) return id;
    return (id & 7) * per + (id >> 3);
}

__device__ __forceinline__ u32x4 relu8(u32x4 v) {
    const i16x8 s = __builtin_bit_cast(i16x8, v);
    const i16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(u32x4, __builtin_elementwise_max(s, z));         // negative bf16 = negative int16
}

template <int BCO, int NWV, bool RELU>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void wgrad_halo16_kernel(const WgHaloArgs a) {
    constexpr int NT = 64 * NWV;
    constexpr int SY = BCO + 32;
    constexpr int YU = BCO / 8;                              // 16-byte units per gradient pixel
    constexpr int NGY = TH * TW * YU / NT;                   // gradient units per thread (4)
    constexpr int XUNITS = HH * HW * (BC / 8);               // 1440
    constexpr int NX = (XUNITS + NT - 1) / NT;               // input units per thread (6 / 3)
    static_assert(TH * TW * YU % NT == 0, "gradient tile divides over the threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Ys = reinterpret_cast<__bf16*>(smem);            // [TH * TW][SY]
    __bf16* Xs = Ys + TH * TW * SY;                          // [HH * HW][SX]

    const int tid = threadIdx.x;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int npairs = a.pairs_co * a.pairs_c;
    const int split = logical / npairs;
    const int pair = logical - split * npairs;
    const int tco = pair % a.pairs_co, tc = pair / a.pairs_co;
    const int t_begin = split * a.tiles_per_block;
    int t_end = t_begin + a.tiles_per_block;
    if (t_end > a.ntiles) t_end = a.ntiles;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.gy), 0, (int)a.gy_bytes, 0x00020000);

    // this thread's fixed places in the two tiles: unit u = tid + k * NT of the gradient tile is channel unit tid % YU of pixel
    // u / YU; of the halo image, channel unit tid % 8 of halo pixel u / 8.  Kept per unit: the byte offset relative to the
    // tile's first pixel is recomputed per tile (two multiplies: registers are what this kernel is short of); kept: for the
    // halo, the packed (row, column) of each unit.
    const int cuy = tid & (YU - 1), cux = tid & (BC / 8 - 1);
    const unsigned gch = (unsigned)(tco * BCO + cuy * 8) * 2u, xch = (unsigned)(tc * BC + cux * 8) * 2u;
    int xpk[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int u = tid + k * NT;
        const int hp = u < XUNITS ? u / (BC / 8) : 0;
        const int hy = hp / HW, hx = hp - hy * HW;
        xpk[k] = u < XUNITS ? ((hy << 8) | hx) : 0xFFFF;          // 0xFFFF: no such unit (row 255 never passes the bounds check)
    }

    u32x4 ry[NGY], rx[NX];
    const int tiles_img = a.tiles_y * a.tiles_x;
    auto load_tile = [&](int t) {
        const int b = t / tiles_img;
        const int rem = t - b * tiles_img;
        const int iy = rem / a.tiles_x;
        const int y0 = iy * TH, x0 = (rem - iy * a.tiles_x) * TW;
        const bool tv = t < t_end;
        const unsigned gbase = (unsigned)((b * a.H + y0) * a.W + x0) * (unsigned)a.Cout * 2u + gch;
        // the halo's first pixel may lie above / left of the image: its (wrapped) offset is only used where the bounds hold
        const unsigned xbase = (unsigned)((b * a.H + y0 + a.dy0) * a.W + x0 + a.dx0) * (unsigned)a.Cin * 2u + xch;
        const unsigned gpix = (unsigned)a.Cout * 2u, xpix = (unsigned)a.Cin * 2u;
#pragma unroll
        for (int k = 0; k < NGY; ++k) {
            const int p = (tid + k * NT) / YU;
            const bool ok = tv & (y0 + p / TW < a.H) & (x0 + p % TW < a.W);
            const unsigned rel = (unsigned)((p / TW) * a.W + (p % TW)) * gpix;
            ry[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)((gbase + rel) | ((unsigned)ok - 1u)), 0, 0);
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int y = y0 + a.dy0 + (xpk[k] >> 8), x = x0 + a.dx0 + (xpk[k] & 255);
            const bool ok = tv & ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W);
            const unsigned rel = (unsigned)((xpk[k] >> 8) * a.W + (xpk[k] & 255)) * xpix;
            rx[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)((xbase + rel) | ((unsigned)ok - 1u)), 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < NGY; ++k)
            *reinterpret_cast<u32x4*>(Ys + ((tid + k * NT) / YU) * SY + cuy * 8) = ry[k];
#pragma unroll
        for (int k = 0; k < NX; ++k)
            if (tid + k * NT < XUNITS)
                *reinterpret_cast<u32x4*>(Xs + ((tid + k * NT) / (BC / 8)) * SX + cux * 8) = RELU ? relu8(rx[k]) : rx[k];
    };

    // fragment addressing (see wgrad16_kernel): 16-lane group g = lane >> 4 takes pixels 8 * (g >> 1) + 0..3 (+4: second read)
    // and channels 16 * (g & 1) .. +15 of its 32-wide MFMA tile; lane 4q + p of the group addresses pixel row q, channels
    // 4p .. 4p + 3, and receives channel (lane & 15) of the four pixels
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                 // 32-channel tiles: gradient (co) x input (c)
    const int li = lane & 15, fq = li >> 2, fp = li & 3, cg = (lane >> 4) & 1;
    const int trY = (8 * h + fq) * SY + wm * 32 + 16 * cg + 4 * fp;
    const int trX = (8 * h + fq) * SX + wn * 32 + 16 * cg + 4 * fp;
    typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_t;
    auto frag = [&](const __bf16* base, int stride) {
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)base);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(base + 4 * stride));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (t_begin < t_end) load_tile(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();                    // the previous tile's fragments have been read
        store_tile();
        __syncthreads();
        if constexpr (!(WGH_DBG & 4)) load_tile(t + 1);       // in flight under this tile's MFMAs (nothing is fetched beyond t_end)
        if constexpr (WGH_DBG & 8) continue;
        // (Tried in round 5 and dropped: the fragments of halo row rr + 1 read while row rr multiplies -- two register sets, the
        // order pinned with sched_barrier -- and with it the next tile's global loads pinned in front of the MFMAs: res2 0.181 ->
        // 0.185 ms, res3 0.140 -> 0.148, res4 0.131 -> 0.138; the loads alone pinned there (the scheduler sinks them to the end of
        // the MFMAs): +1-2 %.  The scheduler's own order interleaves the eight transposing reads
        // with the previous row's MFMAs; a burst of reads in front of nine MFMAs is worse.)
        bf16x8_t ay[3];                     // gradient fragments of output rows rr, rr - 1, rr - 2 (slot = row % 3)
#pragma unroll
        for (int rr = 0; rr < HH; ++rr) {
            // halo row rr = input row y0 + dy0 + rr: it meets output row rr - i under vertical tap i
            if (rr < TH) ay[rr % 3] = frag(Ys + rr * TW * SY + trY, SY);
            bf16x8_t bx[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) bx[j] = frag(Xs + (rr * HW + j) * SX + trX, SX);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int orow = rr - i;
                if (orow >= 0 && orow < TH) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if constexpr (WGH_DBG & 2) { acc[i * 3 + j][0] += (float)ay[orow % 3][0] * (float)bx[j][0]; continue; }
                        acc[i * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay[orow % 3], bx[j], acc[i * 3 + j], 0, 0, 0);
                    }
                }
            }
        }
    }

    // dw[co][(i, j)][c] += acc: fp32 atomics into the gradient arena (one per element and block), or -- with a workspace --
    // plain stores of the raw tile into this block's slab (every block writes its whole tile, also an idle one its zeros)
    const int ktot = 9 * a.Cin;
    float* const slab = a.ws ? a.ws + (int64_t)split * a.slab : nullptr;
    if constexpr (WGH_DBG & 1) {        // every accumulator stays live
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[t][e];
        if (sum == 123.456f) a.dw[0] = sum;
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int col = (i * 3 + j) * a.Cin + tc * BC + wn * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = tco * BCO + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (slab) slab[(int64_t)co * ktot + col] = acc[i * 3 + j][e];
                else atomic_add_f32(a.dw + (int64_t)co * ktot + col, acc[i * 3 + j][e]);
            }
        }
}

// blocks per channel-tile pair the launcher runs for a request (0 = its default), or an error code (< 0); sets the tile grid of `a`
template <int BCO, int NWV>
int plan_splits(WgHaloArgs& a, int splits_req) {
    a.pairs_co = a.Cout / BCO;
    a.pairs_c = a.Cin / BC;
    const int npairs = a.pairs_co * a.pairs_c;
    int splits = splits_req;
    if (splits <= 0) {
        const int cus = loans_device_cus();
        if (cus <= 0) return LOANS_EINVAL;
        const int slots = cus * (NWV == 4 ? 2 : 1);
        splits = (2 * slots + npairs - 1) / npairs;                 // about two rounds of the machine's block slots
        const int max_splits = (a.ntiles + 3) / 4;                  // >= 4 pixel tiles per block: 9 * 32 * 32 partial sums each
        if (splits > max_splits) splits = max_splits;
    }
    if (splits > a.ntiles) splits = a.ntiles;
    if (splits < 1) splits = 1;
    a.tiles_per_block = (a.ntiles + splits - 1) / splits;
    return (a.ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
}

template <int BCO, int NWV, bool RELU>
int launch(WgHaloArgs& a, int splits_req, hipStream_t st, int* slabs) {
    static loans_device_once lds_limit_set;
    constexpr size_t lds = (size_t)(TH * TW * (BCO + 32) + HH * HW * SX) * 2;
    auto kern = wgrad_halo16_kernel<BCO, NWV, RELU>;
    const int splits = plan_splits<BCO, NWV>(a, splits_req);
    if (splits < 0) return splits;
    if (slabs) *slabs = splits;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), lds)) return rc_;
    hipLaunchKernelGGL(kern, dim3(a.pairs_co * a.pairs_c * splits), dim3(64 * NWV), lds, st, a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

void fill_args(WgHaloArgs& a, const loans_igemm_desc* d) {
    a.B = d->B; a.H = d->inH; a.W = d->inW; a.Cin = d->Cin; a.Cout = d->Cout;
    a.dy0 = d->dy[0]; a.dx0 = d->dx[0];
    a.tiles_y = (a.H + TH - 1) / TH; a.tiles_x = (a.W + TW - 1) / TW;
    a.ntiles = a.B * a.tiles_y * a.tiles_x;
    a.slab = (int64_t)d->Cout * 9 * d->Cin;
}

}  // namespace

// LOANS_TILE_WGHALO_* covers: the forward geometry of a stride-1 convolution with a 3 x 3 tap grid (row-major, any padding),
// Cin % 64 == 0, Cout % (64 | 128) == 0, not the dense RGB layout
int loans_wgrad_halo16_covers(const loans_igemm_desc* d, int tile) {
    if (tile != LOANS_TILE_WGHALO_64 && tile != LOANS_TILE_WGHALO_128) return 0;
    if (d->flags & ~LOANS_F_RELU_IN) return 0;
    if (d->isy != 1 || d->isx != 1 || d->osy != 1 || d->osx != 1 || d->oy0 || d->ox0) return 0;
    if (d->inH != d->outH || d->inW != d->outW || d->gridH != d->outH || d->gridW != d->outW) return 0;
    if ((d->Cin % BC) || (d->Cout % (tile == LOANS_TILE_WGHALO_64 ? 64 : 128))) return 0;
    if (d->ntaps != 9) return 0;
    for (int t = 0; t < 9; ++t)
        if (d->dy[t] != d->dy[0] + t / 3 || d->dx[t] != d->dx[0] + t % 3) return 0;
    if (d->dy[0] < -2 || d->dy[0] > 0 || d->dx[0] < -2 || d->dx[0] > 0) return 0;
    if ((int64_t)d->B * d->inH * d->inW * (d->Cin > d->Cout ? d->Cin : d->Cout) * 2 >= 0xFFFFFFF0ll) return 0;
    return 1;
}

// slabs the launch below would write for this request (>= 1), or an error code
int loans_wgrad_halo16_slabs(const loans_igemm_desc* d, int tile, int splits) {
    if (!loans_wgrad_halo16_covers(d, tile)) return LOANS_EINVAL;
    WgHaloArgs a;
    fill_args(a, d);
    return tile == LOANS_TILE_WGHALO_64 ? plan_splits<64, 4>(a, splits) : plan_splits<128, 8>(a, splits);
}

// ws = nullptr: dw += the gradient by fp32 atomics.  ws != nullptr: the blocks store raw partial tiles into `*slabs` slabs of
// Cout * 9 * Cin floats (the caller folds them, loans_fold_slabs_f32); dw is not touched
int loans_wgrad_halo16_launch(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, int tile, int splits,
                              unsigned x_bytes, unsigned gy_bytes, float* ws, int* slabs, hipStream_t st) {
    if (!loans_wgrad_halo16_covers(d, tile)) return LOANS_EINVAL;
    WgHaloArgs a;
    a.x = static_cast<const __bf16*>(x); a.gy = static_cast<const __bf16*>(gy); a.dw = dw;
    fill_args(a, d);
    a.x_bytes = x_bytes; a.gy_bytes = gy_bytes;
    a.ws = ws;
    const bool relu = d->flags & LOANS_F_RELU_IN;
    if (tile == LOANS_TILE_WGHALO_64)
        return relu ? launch<64, 4, true>(a, splits, st, slabs) : launch<64, 4, false>(a, splits, st, slabs);
    return relu ? launch<128, 8, true>(a, splits, st, slabs) : launch<128, 8, false>(a, splits, st, slabs);
}
