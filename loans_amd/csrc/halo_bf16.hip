// Stride-1 convolutions on bf16 STORAGE with the input tile staged ONCE per channel chunk (LOANS_TILE_HALO_*, reached
// through loans_igemm_bf16s): the forward 3x3 / 1 convolutions of the residual stages and of the assessor
// (sheep/resnet.py:128-132,151-153, common/net.py:15,37,58-59) and their data gradients (the same geometry with the
// transposed weights).
//
// Why: as an implicit GEMM (igemm_bf16.hip) every one of the nine taps gathers its own BM x 64 A tile from L2 -- nine times
// the input through the L2 -> LDS path (2.4 GB per res2 convolution of BASELINE configs[2], 8.8 TB/s of a path that tops
// out near 17 TB/s) and, per 1 KiB DMA piece, the tap-mask / address arithmetic that the stamps of DESIGN 7 found to bound
// the loop.  Here a block owns a TH x TW tile of output pixels of ONE image and BN output channels; per 64-channel chunk it
// stages the (TH + 2) x (TW + 2) halo image once (out-of-image pixels: out-of-range offsets, zeros), and the taps read
// SHIFTED WINDOWS of that image: a tap is an LDS address offset, not a gather.  Per (chunk, tap) step only the BN x 64
// weight tile streams in.  Same LDS-DMA + XOR-swizzle idiom, fragment reads, pipelined step loop, epilogue and flags as
// igemm16_kernel; K is walked chunk-major (64 channels x all taps), so sums agree with that kernel to fp32 rounding,
// bit for bit when Cin = 64.
//   tile 8 x 16 x BN 128 (LOANS_TILE_HALO_128): A 2 x 23 KiB + B 2 x 16 KiB = 78 KiB, two blocks per CU
//   tile 16 x 16 x BN 64 (LOANS_TILE_HALO_256x64, Cin = 64 only -- one chunk, one A buffer): A 41 KiB + B 2 x 8 KiB
//   tile 8 x 16 x BN 64, one chunk (LOANS_TILE_HALO_128x64S, Cin = 64): 39 KiB, FOUR blocks per CU -- a Cin = 64 tile lives for
//     nine steps only, so what hides its image load and its epilogue is other blocks, not its own pipeline
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BKH = 64;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

struct Halo16Args {
    const __bf16* in;
    const __bf16* w;
    __bf16* out;
    const float* bias;
    double* stats;
    const __bf16* ref;
    const __bf16* addend;
    loans_igemm_desc d;
    int Ktot, cchunks, tiles_y, tiles_x, tiles_n;
    int nx, ny, dymin, dxmin, sdy, sdx;     // the taps are an ny x nx grid: dy = dy0 + row * sdy, dx = dx0 + col * sdx, sd = +-1
    int HH, HW;                             // halo image of a tile: (TH + ny - 1) x (TW + nx - 1) pixels
    unsigned in_bytes, w_bytes, out_bytes;
    int nt_out;         // output stores non-temporal (loans_conv_nt)
    int dbg;                                // experiment bits (LOANS_EXPERIMENT builds only; 0 in the product library)
};

#ifdef LOANS_EXPERIMENT
#define HDBG(bit) (a.dbg & (bit))
#else
#define HDBG(bit) false
#endif

__device__ __forceinline__ int xcd_remap_h(int id, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}
__device__ __forceinline__ bf16x8_t relu8(bf16x8_t v) {
    const s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8_t, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, v), z));
}
__device__ __forceinline__ f32x4 lo4(bf16x8_t v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
__device__ __forceinline__ f32x4 hi4(bf16x8_t v) { return f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]}; }

template <int TH, int TW, int BN, bool ADB>
constexpr size_t halo_stage_bytes() {
    constexpr int apieces = ((TH + 2) * (TW + 2) + 7) / 8;
    return (size_t)(ADB ? 2 : 1) * apieces * 1024 + (size_t)2 * BN * BKH * 2;
}
// the fp32 staging tile of the epilogue goes through the LDS in EP passes of BM / EP rows (two for the 16 x 16 x 256 tile)
template <int TH, int TW, int BN>
constexpr int halo_epilogue_passes() { return (size_t)TH * TW * (BN + 4) * 4 > 140 * 1024 ? 2 : 1; }
template <int TH, int TW, int BN, bool ADB>
constexpr size_t halo_aux_bytes() {
    constexpr size_t cs = (size_t)TH * TW / halo_epilogue_passes<TH, TW, BN>() * (BN + 4) * 4;
    return halo_stage_bytes<TH, TW, BN, ADB>() > cs ? halo_stage_bytes<TH, TW, BN, ADB>() : cs;
}
template <int TH, int TW, int BN, bool ADB>
constexpr size_t halo_lds_bytes() { return halo_aux_bytes<TH, TW, BN, ADB>() + (size_t)TH * TW * 4; }

// ADB: the A image is double-buffered across channel chunks (false: a single chunk, Cin = 64)
template <int TH, int TW, int BN, int WM, int WN, bool ADB, bool RELU>
__global__ __launch_bounds__(64 * WM * WN) void halo16_kernel(const Halo16Args a) {
    constexpr int NW = WM * WN, NT = 64 * NW;             // 4 waves, or 8 for the 16 x 16 x 128 tile (one block per CU)
    constexpr int RPP = NT / 8;                             // weight-tile rows staged per pass
    constexpr int BM = TH * TW;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RB = BN / RPP;
    constexpr int NMMA = TM * TN;
    constexpr int APIECES = ((TH + 2) * (TW + 2) + 7) / 8;     // 1 KiB pieces (8 halo pixels x 128 B) of one A image
    constexpr int APW = (APIECES + NW - 1) / NW;               // per wave
    constexpr int ABUF = APIECES * 8 * BKH;                    // elements per A buffer
    static_assert((NW == 4 || NW == 8) && TM >= 1 && TN >= 1 && TW == 16 && RB >= 1, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);              // [ADB ? 2 : 1][APIECES * 8][64]
    __bf16* Bs = As + (ADB ? 2 : 1) * ABUF;                    // [2][BN][64]
    unsigned* opix = reinterpret_cast<unsigned*>(smem + halo_aux_bytes<TH, TW, BN, ADB>());     // [BM] output byte offset, ~0u = none

    const loans_igemm_desc& d = a.d;
    const int tid = threadIdx.x;
    int logical = xcd_remap_h(blockIdx.x, gridDim.x);
    const int tn = logical % a.tiles_n; logical /= a.tiles_n;
    const int tx = logical % a.tiles_x; logical /= a.tiles_x;
    const int ty = logical % a.tiles_y;
    const int b = logical / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    for (int m = tid; m < BM; m += NT) {
        const int y = y0 + m / TW, x = x0 + m % TW;
        opix[m] = (y < d.outH && x < d.outW) ? (unsigned)((b * d.outH + y) * d.outW + x) * (unsigned)d.Cout * 2u : 0xFFFFFFFFu;
    }

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // ---- A image: piece pi = wave + NW j holds halo pixels 8 pi .. 8 pi + 7; lane (pixel l >> 3, slot l & 7) fetches unit
    // slot ^ key(pixel) of its pixel.  The source offsets are fixed for the block (only the channel chunk moves).
    unsigned aoff[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int pi = wave + NW * j;
        const int q = pi * 8 + (lane >> 3);
        const int qy = q / a.HW, qx = q - qy * a.HW;
        const int iy = y0 + a.dymin + qy, ix = x0 + a.dxmin + qx;
        const bool ok = pi < APIECES && qy < a.HH && (unsigned)iy < (unsigned)d.inH && (unsigned)ix < (unsigned)d.inW;
        const int unit = (lane & 7) ^ ((qx >> 1) & 7);          // keyed by the image COLUMN: see read_frag
        aoff[j] = ok ? (unsigned)((b * d.inH + iy) * d.inW + ix) * (unsigned)d.Cin * 2u + (unsigned)unit * 16u : 0x80000000u;
    }
    auto dma_a = [&](int buf, int j, int cc) {
        if (wave_u + NW * j < APIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(As + buf * ABUF + (wave_u + NW * j) * 8 * BKH), 16,
                                                     (int)(aoff[j] + (unsigned)cc * 128u), 0, 0, 0);
    };
    // ---- B tile of a step: rows n = tn BN + RPP i + 8 wave + (lane >> 3), unit (lane & 7) ^ key(row); k = tap Cin + 64 cc
    const int lrow = tid >> 3;
    const int lu = (tid & 7) ^ ((tid >> 4) & 7);
    unsigned woff[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = tn * BN + lrow + RPP * i;
        woff[i] = n < d.Cout ? (unsigned)n * (unsigned)a.Ktot * 2u + (unsigned)lu * 16u : 0x80000000u;
    }
    auto dma_b = [&](int buf, int i, unsigned kbytes) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + (buf * BN + RPP * i + 8 * wave_u) * BKH), 16,
                                                 (int)(woff[i] + kbytes), 0, 0, 0);
    };

    // ---- fragments: A row = tile pixel m = (wm TM + i) 32 + r at halo position (m / TW, m % TW) + the tap's shift
    const int wm = wave / WN, wn = wave % WN;
    int q0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 32 + r;
        q0[i] = (m / TW) * a.HW + (m % TW);
    }
    const int fkey = (r >> 1) & 7;
    const int fragB = (wn * TN * 32 + r) * BKH + ((h ^ fkey) & 7) * 8;
    // The 16-byte unit of a pixel's 128-byte row is swizzled by the pixel's COLUMN in the halo image, (qx >> 1) & 7, not by its
    // linear index: the 16 lanes one LDS cycle of a ds_read_b128 serves (MI355X_MICROARCH.md, LDS) are tile columns 0-3 / 12-15 of
    // one image row and 4-11 of the next -- sixteen consecutive columns whatever the tap's shift, i.e. sixteen different
    // (column parity, key) slots of the 256-byte LDS row.  Keyed by the linear index the second row sits HW = 18 pixels on and two
    // of its eight lanes land on slots of the first (25-29 % of the LDS-array cycles of these kernels were bank conflicts,
    // profiles/r5_conv_kernels_sq.txt).  (HW is even for one and three taps per row; with two the odd rows keep a 2-way overlap.)
    const int xr = r & (TW - 1);
    auto read_frag = [&](int abuf, int bbuf, int tq, int s, bf16x8_t (&af)[TM], bf16x8_t (&bf)[TN]) {
        const __bf16* Ab = As + abuf * ABUF;
        const __bf16* Bb = Bs + bbuf * BN * BKH + (fragB ^ (s * 16));
        const int key = (h ^ ((xr + (tq & 0xFFFF)) >> 1)) & 7;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int q = q0[i] + (tq >> 16);
            af[i] = *reinterpret_cast<const bf16x8_t*>(Ab + ((q * BKH + key * 8) ^ (s * 16)));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8_t*>(Bb + j * 32 * BKH);
    };
    auto relu_frag = [&](bf16x8_t (&af)[TM]) {
        if constexpr (RELU) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = relu8(af[i]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto mma_one = [&](int q, const bf16x8_t (&af)[TM], const bf16x8_t (&bf)[TN]) {
        const int i = q / TN, j = q % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    auto mma = [&](const bf16x8_t (&af)[TM], const bf16x8_t (&bf)[TN]) {
#pragma unroll
        for (int q = 0; q < NMMA; ++q) mma_one(q, af, bf);
    };

    // ---- step state (wave-uniform): tap (row tr, column tc of the tap grid), channel chunk cc
    const int ntaps = d.ntaps;
    auto tap_shift = [&](int tr, int tc) {      // halo shift of tap (tr, tc) in pixels
        const int ry = a.sdy > 0 ? tr : a.ny - 1 - tr, rx = a.sdx > 0 ? tc : a.nx - 1 - tc;
        return ((ry * a.HW + rx) << 16) | rx;       // (pixel shift, column shift)
    };
    int t = 0, tr = 0, tc = 0, cc = 0;            // current step
    int tn_ = 0, trn = 0, tcn = 0, ccn = 0;       // next step
    auto next_of = [&](int& tt, int& r_, int& c_, int& ch) {
        ++tt; ++c_;
        if (c_ == a.nx) { c_ = 0; ++r_; }
        if (tt == ntaps) { tt = 0; r_ = 0; c_ = 0; ++ch; }
    };
    next_of(tn_, trn, tcn, ccn);
    const int nsteps = a.cchunks * ntaps;

    // the DMA pieces of a step: the RB weight pieces of the NEXT step, and (double-buffered A) this step's share of the
    // APW pieces of the next chunk's image
    constexpr int APS = ADB ? (APW + 8) / 9 : 0;            // A pieces per step so that 9 steps cover APW (fewer taps: see below)
    constexpr int NPIECE = RB + APS;
    constexpr int PPG = (NPIECE + 2 * NMMA - 1) / (2 * NMMA);
    auto dma_piece = [&](int p, int step) {
        if (p < RB) {
            dma_b((step + 1) & 1, p, (unsigned)((tn_ * d.Cin + ccn * BKH) * 2));
        } else if (ADB && p < NPIECE) {
            if (cc + 1 < a.cchunks) {
                // piece j of the next image at tap t: j = t * APS + (p - RB); with fewer than 9 taps the rest goes out at the
                // last tap
                const int j0 = t * APS + (p - RB);
                if (j0 < APW) dma_a((cc + 1) & 1, j0, cc + 1);
                if (t == ntaps - 1 && (p - RB) == APS - 1)
                    for (int j = ntaps * APS; j < APW; ++j) dma_a((cc + 1) & 1, j, cc + 1);
            }
        }
    };

    // ---- prologue: image of chunk 0 and weights of step 0
#pragma unroll
    for (int j = 0; j < APW; ++j) dma_a(0, j, 0);
#pragma unroll
    for (int i = 0; i < RB; ++i) dma_b(0, i, 0u);
    __syncthreads();
    bf16x8_t fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    int tq = tap_shift(0, 0);
    read_frag(0, 0, tq, 0, fa0, fb0);
    int step = 0;
    for (; step + 1 < nsteps; ++step) {
        const int abuf = ADB ? (cc & 1) : 0, bbuf = step & 1;
        read_frag(abuf, bbuf, tq, 1, fa1, fb1);
        relu_frag(fa0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NMMA; ++q) {
            mma_one(q, fa0, fb0);
#pragma unroll
            for (int p = q * PPG; p < (q + 1) * PPG; ++p) dma_piece(p, step);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frag(abuf, bbuf, tq, 2, fa0, fb0);
        relu_frag(fa1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NMMA; ++q) {
            mma_one(q, fa1, fb1);
#pragma unroll
            for (int p = (NMMA + q) * PPG; p < (NMMA + q + 1) * PPG; ++p) dma_piece(p, step);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frag(abuf, bbuf, tq, 3, fa1, fb1);
        relu_frag(fa0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        relu_frag(fa1);
        // (round 5: the same barrier WITHOUT its vmcnt wait -- wrong results, the bound of what a deeper weight ring could buy -- runs
        // the res3 layers in the same 0.205 ms: the step's DMA is not what the waves wait for, profiles/r5_halo_kernels_ab.txt)
        __syncthreads();                        // the next step's weights (and, at a chunk's end, the next image) have landed
        t = tn_; tr = trn; tc = tcn; cc = ccn;
        next_of(tn_, trn, tcn, ccn);
        tq = tap_shift(tr, tc);
        read_frag(ADB ? (cc & 1) : 0, (step + 1) & 1, tq, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    }
    {   // last step
        const int abuf = ADB ? (cc & 1) : 0, bbuf = step & 1;
        read_frag(abuf, bbuf, tq, 1, fa1, fb1);
        relu_frag(fa0);
        mma(fa0, fb0);
        read_frag(abuf, bbuf, tq, 2, fa0, fb0);
        relu_frag(fa1);
        mma(fa1, fb1);
        read_frag(abuf, bbuf, tq, 3, fa1, fb1);
        relu_frag(fa0);
        mma(fa0, fb0);
        relu_frag(fa1);
        mma(fa1, fb1);
    }

    // ---- epilogue (as igemm16_kernel): BN statistics from the fp32 accumulators, tile staged through LDS in fp32, every lane
    // converts and stores 8 contiguous channels
    const bool f_bias = d.flags & LOANS_F_BIAS, f_stats = d.flags & LOANS_F_STATS;
    const bool f_mask = d.flags & LOANS_F_MASK, f_add = d.flags & LOANS_F_ADDEND;
    const bool f_addmask = d.flags & LOANS_F_ADDEND_MASK;
    const bool f_bnsums = d.flags & LOANS_F_BNSUMS;            // see igemm16_kernel
    constexpr int LDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(smem);
    __syncthreads();
    if (f_stats) {
        int nvalid = TM * 16;           // a tile inside the image (block-uniform): every row counts, the row table is not read
        if (y0 + TH > d.outH || x0 + TW > d.outW) {
            nvalid = 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    nvalid += opix[wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] != 0xFFFFFFFFu;
        }
        const float cnt = (float)nvalid;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = tn * BN + wn * TN * 32 + j * 32 + r;
            const bool cok = col < d.Cout;
            const float bvv = (f_bias && cok) ? a.bias[col] : 0.f;
            float s = 0.f, q2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    // rows beyond the image hold sums of zeros only if their A rows were zero: they are (out-of-range pixels
                    // of the halo image), but a ragged tile's rows are real pixels of the NEXT tile -- mask them here
                    const bool live = opix[wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] != 0xFFFFFFFFu;
                    const float v = live ? acc[i][j][e] : 0.f;
                    s += v;
                    q2 += v * v;
                }
            q2 = q2 + 2.f * bvv * s + cnt * bvv * bvv;
            s = s + cnt * bvv;
            s += __shfl_xor(s, 32, 64);
            q2 += __shfl_xor(q2, 32, 64);
            if (h == 0 && cok) {
                double* st = a.stats + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * d.Cout;
                atomic_add_f64(st + col, (double)s);
                atomic_add_f64(st + d.Cout + col, (double)q2);
            }
        }
    }

    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ref = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.ref ? a.ref : a.out), 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.addend ? a.addend : a.out), 0, (int)a.out_bytes, 0x00020000);
    constexpr int CPR = BN / 8;
    constexpr int RSTEP = NT / CPR;
    const int oc8 = tid % CPR, r0 = tid / CPR;
    const int col0 = tn * BN + oc8 * 8;
    const unsigned cbad = (col0 + 7 < d.Cout) ? 0u : 0xFFFFFFFFu;
    const unsigned coff = (unsigned)col0 * 2u;
    f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = b_lo;
    if (f_bias && !cbad) {
        b_lo = *reinterpret_cast<const f32x4*>(a.bias + col0);
        b_hi = *reinterpret_cast<const f32x4*>(a.bias + col0 + 4);
    }
    auto keep_pos = [](f32x4 v, f32x4 m) {
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        return v;
    };
    f32x4 bn_mean[2], bn_scale[2], bn_shift[2], bn_s1[2], bn_s2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { bn_mean[q] = bn_scale[q] = bn_shift[q] = bn_s1[q] = bn_s2[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (f_bnsums && !cbad) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bn_mean[q] = *reinterpret_cast<const f32x4*>(a.bias + col0 + 4 * q);
            bn_scale[q] = *reinterpret_cast<const f32x4*>(a.bias + 2 * d.Cout + col0 + 4 * q);
            bn_shift[q] = *reinterpret_cast<const f32x4*>(a.bias + 3 * d.Cout + col0 + 4 * q);
        }
    }
    // The epilogue's operands (ReLU reference, addend, the BN input of LOANS_F_BNSUMS) are requested BEFORE the tile goes through the
    // staging area: inside the store loop every load waited behind the previous row's store (vector memory operations retire in
    // order) and paid its own latency, once per row.  (The two-pass tile, 16 x 16 x 256, holds 128 accumulators per wave up to its
    // pass: there the operands are requested row by row behind the staging, as in igemm16_kernel's 128 x 64 wave tiles.)
    constexpr int EP = halo_epilogue_passes<TH, TW, BN>();
    constexpr int PR = BM / EP;                 // rows per pass
    static_assert(PR % (TM * 32) == 0 || (TM * 32) % PR == 0, "a wave's rows fall into whole passes");
    constexpr int NIT = PR / RSTEP;
    constexpr bool EARLY = EP == 1;
    constexpr int NA = EARLY ? NIT : 1;
    unsigned eoff[NA];
    bf16x8_t e_ref[NA], e_add[NA];
#pragma unroll 1
    for (int ep = 0; ep < EP; ++ep) {
        if (EARLY) {
#pragma unroll
            for (int p = 0; p < NIT; ++p) {
                const unsigned po = opix[r0 + p * RSTEP];
                eoff[p] = (po + coff) | (po == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u) | cbad;
            }
            if (f_mask || f_addmask || f_bnsums) {
#pragma unroll
                for (int p = 0; p < NIT; ++p)
                    e_ref[p] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, (int)eoff[p], 0, 0));
            }
            if (f_add) {
#pragma unroll
                for (int p = 0; p < NIT; ++p)
                    e_add[p] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)eoff[p], 0, 0));
            }
        }
        if (ep) __syncthreads();            // the previous pass has been read
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int trow = wm * TM * 32 + i * 32;         // first row of this 32-row MFMA tile
            if (trow / PR == ep) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        Cs[(trow % PR + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + wn * TN * 32 + j * 32 + r] = acc[i][j][e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NIT; ++p) {
            const int row = r0 + p * RSTEP;
            if (!EARLY) {
                const unsigned po = opix[ep * PR + row];
                eoff[0] = (po + coff) | (po == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u) | cbad;
                if (f_mask || f_addmask || f_bnsums)
                    e_ref[0] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, (int)eoff[0], 0, 0));
                if (f_add)
                    e_add[0] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)eoff[0], 0, 0));
            }
            const unsigned off = eoff[p % NA];
            f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + row * LDC + oc8 * 8) + b_lo;
            f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + row * LDC + oc8 * 8 + 4) + b_hi;
            if (f_mask || f_addmask) {
                const f32x4 rl = lo4(e_ref[p % NA]), rh = hi4(e_ref[p % NA]);
                if (f_mask) { lo = keep_pos(lo, rl); hi = keep_pos(hi, rh); }
                if (f_add) {
                    f32x4 al = lo4(e_add[p % NA]), ah = hi4(e_add[p % NA]);
                    if (f_addmask) { al = keep_pos(al, rl); ah = keep_pos(ah, rh); }
                    lo += al; hi += ah;
                }
            } else if (f_add) {
                lo += lo4(e_add[p % NA]); hi += hi4(e_add[p % NA]);
            }
            bf16x8_t o;
            const bf16x4_t ol = __builtin_convertvector(lo, bf16x4_t), oh = __builtin_convertvector(hi, bf16x4_t);
            o[0] = ol[0]; o[1] = ol[1]; o[2] = ol[2]; o[3] = ol[3];
            o[4] = oh[0]; o[5] = oh[1]; o[6] = oh[2]; o[7] = oh[3];
            if (f_bnsums) {          // block-uniform; a row that does not exist loaded zeros and its gradient is zeroed below
                const f32x4 y2[2] = {lo4(e_ref[p % NA]), hi4(e_ref[p % NA])};
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                const bool live = off != 0xFFFFFFFFu;
                const f32x4 g2[2] = {live ? __builtin_convertvector(ol, f32x4) : zero4, live ? __builtin_convertvector(oh, f32x4) : zero4};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 gm = keep_pos(g2[q], y2[q] * bn_scale[q] + bn_shift[q]);
                    bn_s1[q] += gm;
                    bn_s2[q] += gm * (y2[q] - bn_mean[q]);
                }
            }
            LOANS_STORE_B128(__builtin_bit_cast(u32x4, o), rs_out, (int)off, a.nt_out);
        }
    }
    if (f_bnsums) {
        __syncthreads();
        float* Red = reinterpret_cast<float*>(smem);            // [NT / CPR][CPR][16]
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Red[(r0 * CPR + oc8) * 16 + q * 4 + e] = bn_s1[q][e];
                Red[(r0 * CPR + oc8) * 16 + 8 + q * 4 + e] = bn_s2[q][e];
            }
        __syncthreads();
        if (tid < CPR * 16) {
            const int u8 = tid >> 4, j = tid & 15;
            float acc_ = 0.f;
#pragma unroll 4
            for (int rr = 0; rr < NT / CPR; ++rr) acc_ += Red[(rr * CPR + u8) * 16 + j];
            const int col = tn * BN + u8 * 8 + (j & 7);
            if (col < d.Cout) {
                double* st = a.stats + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * d.Cout;
                atomic_add_f64(st + (j >> 3) * d.Cout + col, (double)acc_);
            }
        }
    }
}

// ---- weights stationary in LDS: 3x3 / 1, Cin = 64, Cout <= 64 (the res2 convolutions and their data gradients) ----------
// A Cin = 64 tile of the kernel above lives for nine steps of 8 MFMAs per wave, each behind a barrier and the L2 latency of its
// weight tile: latency-bound at 0.2 ms per res2 convolution of configs[2] against an HBM bound of 0.085.  Here ONE 512-thread
// block per CU keeps the whole weight matrix (9 taps x 64 x 64 bf16 = 72 KiB) in LDS for its lifetime and walks 16 x 16 pixel
// tiles: per tile the only traffic is the halo image (double-buffered: tile k + 1 streams in under tile k's MFMAs) and the
// output; no per-tap barrier, no weight re-read.  Wave w owns row block w (32 pixels) x all 64 output channels: 72 MFMAs per
// tile, 3 fragment reads per 2 MFMAs.  (The same idea with the weights in REGISTERS -- 144 VGPRs per wave -- spills at two waves
// per SIMD and its scratch reloads serialise the LDS-DMA: 0.46 ms.  Eight waves on one block need no weight registers.)
// The output tile is staged through the tile's own, by then idle, image buffer in two passes of 128 rows; BN statistics stay in
// registers (fp64) across the block's tiles and leave as one atomic pair per wave and column.
constexpr int W8_T = 16, W8_BM = 256;
constexpr int W8_APIECES = ((W8_T + 2) * (W8_T + 2) + 7) / 8, W8_APW = (W8_APIECES + 7) / 8, W8_ABUF = W8_APIECES * 8 * BKH;
constexpr int W8_BELEMS = 9 * 64 * BKH;
constexpr size_t ws8_lds_bytes() { return (size_t)2 * W8_ABUF * 2 + (size_t)W8_BELEMS * 2 + W8_BM * 4; }

__global__ __launch_bounds__(512, 1) void ws8_kernel(const Halo16Args a, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);                       // [2][APIECES * 8][64]
    __bf16* Bl = As + 2 * W8_ABUF;                                      // [9][64 n][64 k], rows swizzled like a B tile
    unsigned* opix = reinterpret_cast<unsigned*>(smem + (size_t)2 * W8_ABUF * 2 + (size_t)W8_BELEMS * 2);
    const loans_igemm_desc& d = a.d;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // ---- all weights into LDS, once: 72 pieces of 1 KiB = 8 rows (n) x 128 B of one tap; lane (row l >> 3, slot l & 7) fetches
    // unit slot ^ key(row), so that the fragment reads are conflict-free (the B-tile idiom of igemm16_kernel)
    for (int pc = wave_u; pc < 72; pc += 8) {
        const int t = pc >> 3, n = (pc & 7) * 8 + (lane >> 3);
        const int unit = (lane & 7) ^ ((n >> 1) & 7);
        const unsigned off = n < d.Cout ? (unsigned)n * (unsigned)a.Ktot * 2u + (unsigned)(t * 64) * 2u + (unsigned)unit * 16u : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bl + pc * 8 * BKH), 16, (int)off, 0, 0, 0);
    }
    // per piece of the image this lane stages: its halo pixel (qy, qx) and its swizzled 16-byte unit -- fixed for the block
    int pqy[W8_APW], pqx[W8_APW];
#pragma unroll
    for (int j = 0; j < W8_APW; ++j) {
        const int q = (wave + 8 * j) * 8 + (lane >> 3);
        pqy[j] = q / a.HW;
        pqx[j] = (q - pqy[j] * a.HW) | ((((lane & 7) ^ (((q - pqy[j] * a.HW) >> 1) & 7)) * 16) << 16);    // qx | unit bytes << 16 (keyed by the column: halo16_kernel)
        if (wave + 8 * j >= W8_APIECES || pqy[j] >= a.HH) pqy[j] = 1 << 20;                     // never inside an image
    }
    const int tiles_per_img = a.tiles_y * a.tiles_x;
    auto issue_image = [&](int tile, int buf) {                         // the halo image of `tile` into A[buf] (this wave's pieces)
        const int b = tile / tiles_per_img, tt = tile - b * tiles_per_img;
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        const int iy0 = ty * W8_T + a.dymin, ix0 = tx * W8_T + a.dxmin;
        const int base = b * d.inH;
        const bool tile_ok = tile < ntiles && !HDBG(1);                 // beyond the last tile: out-of-range offsets, no traffic
#pragma unroll
        for (int j = 0; j < W8_APW; ++j) {
            if (wave_u + 8 * j < W8_APIECES) {
                const int iy = iy0 + pqy[j], ix = ix0 + (pqx[j] & 0xFFFF);
                const bool ok = tile_ok && (unsigned)iy < (unsigned)d.inH && (unsigned)ix < (unsigned)d.inW;
                const unsigned off = ok ? (unsigned)((base + iy) * d.inW + ix) * 128u + ((unsigned)pqx[j] >> 16) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(As + buf * W8_ABUF + (wave_u + 8 * j) * 8 * BKH), 16, (int)off, 0, 0, 0);
            }
        }
    };
    const int m0 = wave * 32 + r;                                       // this lane's row of the tile
    const int q0 = (m0 / W8_T) * a.HW + (m0 % W8_T);
    const int fragB = r * BKH + ((h ^ ((r >> 1) & 7)) & 7) * 8;         // column r (and r + 32) of a tap's [64][64] block
    const bool f_bias = d.flags & LOANS_F_BIAS, f_stats = d.flags & LOANS_F_STATS;
    const bool f_mask = d.flags & LOANS_F_MASK, f_add = d.flags & LOANS_F_ADDEND, f_addmask = d.flags & LOANS_F_ADDEND_MASK;
    const float bv0 = (f_bias && r < d.Cout) ? a.bias[r] : 0.f, bv1 = (f_bias && r + 32 < d.Cout) ? a.bias[r + 32] : 0.f;
    double st_s0 = 0.0, st_q0 = 0.0, st_s1 = 0.0, st_q1 = 0.0;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ref = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.ref ? a.ref : a.out), 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.addend ? a.addend : a.out), 0, (int)a.out_bytes, 0x00020000);
    constexpr int LDC = 64 + 4;
    const int oc8 = tid & 7, er0 = tid >> 3;                            // epilogue: 8-channel unit, row within a 64-row sweep
    const unsigned cbad = (oc8 * 8 + 7 < d.Cout) ? 0u : 0xFFFFFFFFu;
    f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = b_lo;
    if (f_bias && !cbad) {
        b_lo = *reinterpret_cast<const f32x4*>(a.bias + oc8 * 8);
        b_hi = *reinterpret_cast<const f32x4*>(a.bias + oc8 * 8 + 4);
    }
    auto keep_pos = [](f32x4 v, f32x4 m_) {
        v.x = m_.x > 0.f ? v.x : 0.f; v.y = m_.y > 0.f ? v.y : 0.f;
        v.z = m_.z > 0.f ? v.z : 0.f; v.w = m_.w > 0.f ? v.w : 0.f;
        return v;
    };

    issue_image(blockIdx.x, 0);
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        __syncthreads();            // this tile's image (first tile: the weights too) has landed; the previous tile's stage is read out
        issue_image(tile + gridDim.x, buf ^ 1);
        if (tid < W8_BM) {
            const int b = tile / tiles_per_img, tt = tile - b * tiles_per_img;
            const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
            const int y = ty * W8_T + tid / W8_T, x = tx * W8_T + tid % W8_T;
            opix[tid] = (y < d.outH && x < d.outW) ? (unsigned)((b * d.outH + y) * d.outW + x) * (unsigned)d.Cout * 2u : 0xFFFFFFFFu;
        }
        // ---- 9 taps x 4 k-steps x 2 column blocks = 72 MFMAs; fragments of step s + 1 are read while step s multiplies
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        const __bf16* Ab = As + buf * W8_ABUF;
        auto frags = [&](int t, int s_, bf16x8_t& fa, bf16x8_t& f0, bf16x8_t& f1) {
            const int tr = t / 3, tc = t - tr * 3;
            const int cx = a.sdx > 0 ? tc : 2 - tc;
            const int q = q0 + (a.sdy > 0 ? tr : 2 - tr) * a.HW + cx;
            fa = *reinterpret_cast<const bf16x8_t*>(Ab + ((q * BKH + ((h ^ (((r & 15) + cx) >> 1)) & 7) * 8) ^ (s_ * 16)));
            const __bf16* bt = Bl + t * 64 * BKH + (fragB ^ (s_ * 16));
            f0 = *reinterpret_cast<const bf16x8_t*>(bt);
            f1 = *reinterpret_cast<const bf16x8_t*>(bt + 32 * BKH);
        };
        bf16x8_t fa, f0, f1, ga, g0, g1;
        frags(0, 0, fa, f0, f1);
        if (!HDBG(2))
#pragma unroll
        for (int st = 0; st < 36; st += 2) {
            frags((st + 1) >> 2, (st + 1) & 3, ga, g0, g1);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, f0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, f1, acc1, 0, 0, 0);
            if (st + 2 < 36) frags((st + 2) >> 2, (st + 2) & 3, fa, f0, f1);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, g0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, g1, acc1, 0, 0, 0);
        }
        __syncthreads();            // all fragment reads of this image are done (and opix is visible): the buffer becomes the stage
        if (f_stats) {
            float s0 = 0.f, q20 = 0.f, s1 = 0.f, q21 = 0.f;
            int nvalid = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool live = opix[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] != 0xFFFFFFFFu;
                const float v0 = live ? acc0[e] : 0.f, v1 = live ? acc1[e] : 0.f;
                s0 += v0; q20 += v0 * v0;
                s1 += v1; q21 += v1 * v1;
                nvalid += (int)live;
            }
            const float cnt = (float)nvalid;
            st_q0 += (double)(q20 + 2.f * bv0 * s0 + cnt * bv0 * bv0); st_s0 += (double)(s0 + cnt * bv0);
            st_q1 += (double)(q21 + 2.f * bv1 * s1 + cnt * bv1 * bv1); st_s1 += (double)(s1 + cnt * bv1);
        }
        float* Cs = reinterpret_cast<float*>(As + buf * W8_ABUF);      // [128][LDC]: the row blocks of waves 4 pass .. 4 pass + 3
        if (!HDBG(4))
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) __syncthreads();                                   // the first pass's readers are done
            if ((wave >> 2) == pass) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float* cp = Cs + ((wave & 3) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + r;
                    cp[0] = acc0[e];
                    cp[32] = acc1[e];
                }
            }
            __syncthreads();
#pragma unroll
            for (int sw = 0; sw < 2; ++sw) {
                const int lr = er0 + 64 * sw;                            // staged row = tile row 128 pass + lr
                const unsigned po = opix[pass * 128 + lr];
                const unsigned off = (po + (unsigned)oc8 * 16u) | (po == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u) | cbad;
                f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + lr * LDC + oc8 * 8) + b_lo;
                f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + lr * LDC + oc8 * 8 + 4) + b_hi;
                if (f_mask || f_addmask) {
                    const bf16x8_t rf = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, (int)off, 0, 0));
                    const f32x4 rl = lo4(rf), rh = hi4(rf);
                    if (f_mask) { lo = keep_pos(lo, rl); hi = keep_pos(hi, rh); }
                    if (f_add) {
                        const bf16x8_t ad = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)off, 0, 0));
                        f32x4 al = lo4(ad), ah = hi4(ad);
                        if (f_addmask) { al = keep_pos(al, rl); ah = keep_pos(ah, rh); }
                        lo += al; hi += ah;
                    }
                } else if (f_add) {
                    const bf16x8_t ad = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)off, 0, 0));
                    lo += lo4(ad); hi += hi4(ad);
                }
                bf16x8_t o;
                const bf16x4_t ol = __builtin_convertvector(lo, bf16x4_t), oh = __builtin_convertvector(hi, bf16x4_t);
                o[0] = ol[0]; o[1] = ol[1]; o[2] = ol[2]; o[3] = ol[3];
                o[4] = oh[0]; o[5] = oh[1]; o[6] = oh[2]; o[7] = oh[3];
                if (!HDBG(8)) LOANS_STORE_B128(__builtin_bit_cast(u32x4, o), rs_out, (int)off, a.nt_out);
            }
        }
    }
    if (f_stats) {
        st_s0 += __shfl_xor(st_s0, 32, 64); st_q0 += __shfl_xor(st_q0, 32, 64);
        st_s1 += __shfl_xor(st_s1, 32, 64); st_q1 += __shfl_xor(st_q1, 32, 64);
        if (h == 0) {
            double* st = a.stats + (size_t)((blockIdx.x * 8 + wave) % LOANS_STATS_REPLICAS) * 2 * d.Cout;
            if (r < d.Cout) { atomic_add_f64(st + r, st_s0); atomic_add_f64(st + d.Cout + r, st_q0); }
            if (r + 32 < d.Cout) { atomic_add_f64(st + r + 32, st_s1); atomic_add_f64(st + d.Cout + r + 32, st_q1); }
        }
    }
}

int launch_ws8(Halo16Args& a, hipStream_t st) {
    static loans_device_once lds_limit_set;
    constexpr size_t lds = ws8_lds_bytes();
    static_assert(lds <= 160 * 1024, "one block per CU");
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(ws8_kernel), lds)) return rc_;
    a.tiles_y = (a.d.outH + W8_T - 1) / W8_T;
    a.tiles_x = (a.d.outW + W8_T - 1) / W8_T;
    a.tiles_n = 1;
    a.HH = W8_T + 2;
    a.HW = W8_T + 2;
    const int64_t ntiles = (int64_t)a.d.B * a.tiles_y * a.tiles_x;
    if (ntiles >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    const int cus = loans_device_cus();
    if (cus <= 0) return LOANS_EINVAL;
    const int64_t grid = ntiles < cus ? ntiles : cus;
    hipLaunchKernelGGL(ws8_kernel, dim3((unsigned)grid), dim3(512), lds, st, a, (int)ntiles);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

// ---- weights stationary, waves autonomous: LOANS_TILE_WSW64 ------------------------------------------------------------------
// ws8_kernel's phases (image DMA, 72 MFMAs, statistics, staged epilogue) ADD: its eight waves work on one 16 x 16 tile and meet
// at five barriers per tile, so nothing runs under the epilogue or under the wait for the next image (0.061 ms of skeleton +
// 0.092 MFMA + 0.038 DMA + 0.042 epilogue per res2 convolution of configs[2]).  Here a WAVE owns its unit of work end to end --
// 2 rows x 16 pixels of one image x all 64 output channels: it stages its own 4 x 18 pixel halo image (nine 1 KiB pieces
// into its own 9 KiB of LDS; round 5: loaded into registers a whole unit ahead, see load_image), runs the same 72 MFMAs against
// the block's stationary weights,
// transposes the result through its own image buffer (idle by then) and stores it.  No block barrier after the weights have
// landed: the eight waves of a CU drift apart and one wave's DMA wait and epilogue lie under the other waves' MFMAs.  The
// price is halo: 72 staged pixels per 32 outputs instead of 324 per 256 (served by L2: the waves of a block take eight
// neighbouring units of a row pair, the blocks of an XCD neighbouring row pairs).  LDS: 72 KiB of weights + 8 x 9 KiB.
constexpr int WSW_HW = 18, WSW_PIECES = 9, WSW_ABUF = WSW_PIECES * 8 * BKH;          // 4 x 18 = 72 halo pixels
constexpr size_t wsw_lds_bytes() { return (size_t)W8_BELEMS * 2 + (size_t)8 * WSW_ABUF * 2; }

// EPI: the launch has epilogue operands (ReLU reference / addend of a data gradient, LOANS_F_BNSUMS) -- false compiles their
// prefetch registers and the BN coefficient state out (72 VGPRs the forward launches then have for the image in flight).
// RS: the next image travels through registers (load_image); false: LDS-DMA, requested once the buffer is free.
#ifndef LOANS_WSW_RS_EPI
#define LOANS_WSW_RS_EPI 0
#endif
// the data-gradient launches keep the LDS-DMA image: with their epilogue operands prefetched a unit ahead the 36 image registers
// spill (45 VGPRs) and configs[2] runs 18.84 ms against 18.11 (profiles/r5_wsw_regstage_ab.txt); -DLOANS_WSW_RS_EPI=1 builds that arm
constexpr bool WSW_RS_EPI = LOANS_WSW_RS_EPI;
template <bool EPI, bool RS>
__global__ __launch_bounds__(512, 1) void wsw_kernel(const Halo16Args a, int nunits, int units_y, int units_x) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Bl = reinterpret_cast<__bf16*>(smem);                       // [9][64 n][64 k], rows swizzled like a B tile
    const loans_igemm_desc& d = a.d;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    __bf16* Aw = Bl + W8_BELEMS + wave_u * WSW_ABUF;                    // this wave's image / staging slab
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    for (int pc = wave_u; pc < 72; pc += 8) {                           // all weights, once (as ws8_kernel)
        const int t = pc >> 3, n = (pc & 7) * 8 + (lane >> 3);
        const int unit = (lane & 7) ^ ((n >> 1) & 7);
        const unsigned off = n < d.Cout ? (unsigned)n * (unsigned)a.Ktot * 2u + (unsigned)(t * 64) * 2u + (unsigned)unit * 16u : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bl + pc * 8 * BKH), 16, (int)off, 0, 0, 0);
    }
    // the nine pieces of an image: piece j holds halo pixels 8 j .. 8 j + 7, lane (pixel l >> 3, slot l & 7) fetches unit
    // slot ^ key(pixel) of its pixel
    int pqy[WSW_PIECES], pqx[WSW_PIECES];
#pragma unroll
    for (int j = 0; j < WSW_PIECES; ++j) {
        const int q = j * 8 + (lane >> 3);
        pqy[j] = q / WSW_HW;
        pqx[j] = (q - pqy[j] * WSW_HW) | ((((lane & 7) ^ (((q - pqy[j] * WSW_HW) >> 1) & 7)) * 16) << 16);  // qx | unit bytes << 16 (keyed by the column: halo16_kernel)
    }
    const int q0 = (r >> 4) * WSW_HW + (r & 15);                        // this lane's output pixel inside the halo image
    const int fragB = r * BKH + ((h ^ ((r >> 1) & 7)) & 7) * 8;
    const bool f_bias = d.flags & LOANS_F_BIAS, f_stats = d.flags & LOANS_F_STATS;
    const bool f_mask = EPI && (d.flags & LOANS_F_MASK), f_add = EPI && (d.flags & LOANS_F_ADDEND);
    const bool f_addmask = EPI && (d.flags & LOANS_F_ADDEND_MASK);
    const bool f_bnsums = EPI && (d.flags & LOANS_F_BNSUMS);            // see igemm16_kernel
    const float bv0 = (f_bias && r < d.Cout) ? a.bias[r] : 0.f, bv1 = (f_bias && r + 32 < d.Cout) ? a.bias[r + 32] : 0.f;
    double st_s0 = 0.0, st_q0 = 0.0, st_s1 = 0.0, st_q1 = 0.0;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ref = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.ref ? a.ref : a.out), 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.addend ? a.addend : a.out), 0, (int)a.out_bytes, 0x00020000);
    constexpr int LDC = 64 + 4;
    const int oc8 = lane & 7;                                           // epilogue: this lane's 8-channel unit
    const unsigned cbad = (oc8 * 8 + 7 < d.Cout) ? 0u : 0xFFFFFFFFu;
    f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = b_lo;
    if (f_bias && !cbad) {
        b_lo = *reinterpret_cast<const f32x4*>(a.bias + oc8 * 8);
        b_hi = *reinterpret_cast<const f32x4*>(a.bias + oc8 * 8 + 4);
    }
    auto keep_pos = [](f32x4 v, f32x4 m_) {
        v.x = m_.x > 0.f ? v.x : 0.f; v.y = m_.y > 0.f ? v.y : 0.f;
        v.z = m_.z > 0.f ? v.z : 0.f; v.w = m_.w > 0.f ? v.w : 0.f;
        return v;
    };
    // LOANS_F_BNSUMS: this lane's eight channels of the BN's coefficient table and its sums over all units of this wave
    f32x4 bn_mean[2], bn_scale[2], bn_shift[2], bn_s1[2], bn_s2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { bn_mean[q] = bn_scale[q] = bn_shift[q] = bn_s1[q] = bn_s2[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (f_bnsums && !cbad) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bn_mean[q] = *reinterpret_cast<const f32x4*>(a.bias + oc8 * 8 + 4 * q);
            bn_scale[q] = *reinterpret_cast<const f32x4*>(a.bias + 2 * d.Cout + oc8 * 8 + 4 * q);
            bn_shift[q] = *reinterpret_cast<const f32x4*>(a.bias + 3 * d.Cout + oc8 * 8 + 4 * q);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                    // the weights are in LDS: the only block barrier of the kernel

    // units in row-pair strips: the eight waves of a block take eight neighbouring units, the blocks of an XCD the strips next
    // to each other (their halo rows overlap: L2 hits)
    int gw = xcd_remap_h(blockIdx.x, gridDim.x) * 8 + wave_u, nw = gridDim.x * 8;
    if (HDBG(16)) {         // experiment: four working waves per CU (one per SIMD)
        if (wave_u >= 4) return;
        gw = xcd_remap_h(blockIdx.x, gridDim.x) * 4 + wave_u;
        nw = gridDim.x * 4;
    }
    const int per_img = units_y * units_x;
    // The NEXT unit's halo image travels through registers: its nine 16-byte loads per lane are issued at the top of a unit,
    // fly under that unit's 72 MFMAs and its epilogue, and are written to the wave's image buffer (ds_write_b128, the layout an
    // LDS-DMA piece would have left) once the buffer -- image, then staging slab of the current unit -- has been read out.
    // As LDS-DMA the image could only be requested at that point, one epilogue ahead of its use: the waves were parked on
    // s_waitcnt for 55 % of their cycles (profiles/r5_conv_kernels_sq.txt) and there is no LDS for a second image per wave
    // (72 KiB of weights + 8 x 9 KiB).  36 VGPRs carry what LDS could not.
    u32x4 img[WSW_PIECES];
    auto load_image = [&](int u) {      // unit u's halo image into registers (out-of-image pixels: out-of-range offsets, zeros)
        const int b = u / per_img, uu = u - b * per_img;
        const int uy = uu / units_x, ux = uu - uy * units_x;
        const int iy0 = 2 * uy + a.dymin, ix0 = 16 * ux + a.dxmin, base = b * d.inH;
#pragma unroll
        for (int j = 0; j < WSW_PIECES; ++j) {
            const int iy = iy0 + pqy[j], ix = ix0 + (pqx[j] & 0xFFFF);
            const bool ok = (unsigned)iy < (unsigned)d.inH && (unsigned)ix < (unsigned)d.inW;
            const unsigned off = ok ? (unsigned)((base + iy) * d.inW + ix) * 128u + ((unsigned)pqx[j] >> 16) : 0x80000000u;
            img[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0));
        }
    };
    auto write_image = [&]() {          // ... and from there into this wave's buffer: piece j, lane l at byte 1024 j + 16 l
#pragma unroll
        for (int j = 0; j < WSW_PIECES; ++j)
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(Aw) + j * 1024 + lane * 16) = img[j];
    };
    // The next unit's image is requested as soon as this unit's staging slab (the same LDS) has been read out -- BEFORE this
    // unit's output stores: at the top of the loop the wave then waits for its nine DMA pieces only, with a counted vmcnt that
    // leaves the four younger stores in flight (vector memory operations retire in order).
    auto dma_image = [&](int u) {       // RS = false: unit u's halo image by LDS-DMA into this wave's buffer
        const int b = u / per_img, uu = u - b * per_img;
        const int uy = uu / units_x, ux = uu - uy * units_x;
        const int iy0 = 2 * uy + a.dymin, ix0 = 16 * ux + a.dxmin, base = b * d.inH;
#pragma unroll
        for (int j = 0; j < WSW_PIECES; ++j) {
            const int iy = iy0 + pqy[j], ix = ix0 + (pqx[j] & 0xFFFF);
            const bool ok = (unsigned)iy < (unsigned)d.inH && (unsigned)ix < (unsigned)d.inW;
            const unsigned off = ok ? (unsigned)((base + iy) * d.inW + ix) * 128u + ((unsigned)pqx[j] >> 16) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(Aw + j * 8 * BKH), 16, (int)off, 0, 0, 0);
        }
    };
    if (gw < nunits) {
        if constexpr (RS) { load_image(gw); write_image(); }
        else { dma_image(gw); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    for (int u = gw; u < nunits; u += nw) {
        const int b = u / per_img, uu = u - b * per_img;
        const int uy = uu / units_x, ux = uu - uy * units_x;
        const int y0 = 2 * uy, x0 = 16 * ux;
        if constexpr (RS) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the image written below (or above, the first one) is in LDS
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            // the wave waits for its nine DMA pieces only: a counted vmcnt leaves the four younger stores in flight
            if (u != gw) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        // The epilogue's operands (ReLU reference / addend of a data gradient, the BN input of LOANS_F_BNSUMS) are requested HERE,
        // a unit's worth of MFMAs ahead of their use and -- the point -- BEFORE the next unit's image: vector memory operations
        // retire in order, so a load issued behind that DMA made the epilogue wait for the whole image to land (+83 us on a
        // 164 us res2 data gradient with BN sums).
        unsigned eoff[4];
        bf16x8_t e_ref[4], e_add[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = it * 8 + (lane >> 3);
            const int y = y0 + (px >> 4), x = x0 + (px & 15);
            const bool pok = y < d.outH && x < d.outW;
            eoff[it] = pok ? ((unsigned)((b * d.outH + y) * d.outW + x) * (unsigned)d.Cout * 2u + (unsigned)oc8 * 16u) | cbad : 0xFFFFFFFFu;
        }
        if (EPI && (f_mask || f_addmask || f_bnsums)) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                e_ref[it] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, (int)eoff[it], 0, 0));
        }
        if (EPI && f_add) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                e_add[it] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)eoff[it], 0, 0));
        }
        const bool more = u + nw < nunits;
        if constexpr (RS) {
            if (more && !HDBG(1)) load_image(u + nw);   // behind the epilogue operands (in-order returns): they are needed first
        }
        asm volatile("" ::: "memory");
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        auto frags = [&](int t, int s_, bf16x8_t& fa, bf16x8_t& f0, bf16x8_t& f1) {
            const int tr = t / 3, tc = t - tr * 3;
            const int cx = a.sdx > 0 ? tc : 2 - tc;
            const int q = q0 + (a.sdy > 0 ? tr : 2 - tr) * WSW_HW + cx;
            fa = *reinterpret_cast<const bf16x8_t*>(Aw + ((q * BKH + ((h ^ (((r & 15) + cx) >> 1)) & 7) * 8) ^ (s_ * 16)));
            const __bf16* bt = Bl + t * 64 * BKH + (fragB ^ (s_ * 16));
            f0 = *reinterpret_cast<const bf16x8_t*>(bt);
            f1 = *reinterpret_cast<const bf16x8_t*>(bt + 32 * BKH);
        };
        // fragments run two steps ahead of the MFMAs (three register sets): with two waves per SIMD an LDS read issued one step
        // (two MFMAs = 64 pipe cycles) ahead is not back in time
        bf16x8_t fa[3], f0[3], f1[3];
        frags(0, 0, fa[0], f0[0], f1[0]);
        frags(0, 1, fa[1], f0[1], f1[1]);
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            if (st + 2 < 36 && !HDBG(4)) frags((st + 2) >> 2, (st + 2) & 3, fa[(st + 2) % 3], f0[(st + 2) % 3], f1[(st + 2) % 3]);
            // (pinned: left to itself the scheduler sinks every fragment read to one MFMA before its use -- s_waitcnt lgkmcnt(0 / 1)
            // in front of nearly every MFMA, the wave parked for the LDS latency 36 times per unit)
            __builtin_amdgcn_sched_barrier(0);
            if (!HDBG(2)) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st % 3], f0[st % 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st % 3], f1[st % 3], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (f_stats) {
            float s0 = 0.f, q20 = 0.f, s1 = 0.f, q21 = 0.f;
            float cnt = 16.f;
            if (y0 + 2 <= d.outH && x0 + 16 <= d.outW) {                // a whole unit (wave-uniform): every element counts
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s0 += acc0[e]; q20 += acc0[e] * acc0[e];
                    s1 += acc1[e]; q21 += acc1[e] * acc1[e];
                }
            } else {
                int nvalid = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = (e & 3) + 8 * (e >> 2) + 4 * h;       // tile row of this accumulator element
                    const bool live = (y0 + (m >> 4) < d.outH) && (x0 + (m & 15) < d.outW);
                    const float v0 = live ? acc0[e] : 0.f, v1 = live ? acc1[e] : 0.f;
                    s0 += v0; q20 += v0 * v0;
                    s1 += v1; q21 += v1 * v1;
                    nvalid += (int)live;
                }
                cnt = (float)nvalid;
            }
            st_q0 += (double)(q20 + 2.f * bv0 * s0 + cnt * bv0 * bv0); st_s0 += (double)(s0 + cnt * bv0);
            st_q1 += (double)(q21 + 2.f * bv1 * s1 + cnt * bv1 * bv1); st_s1 += (double)(s1 + cnt * bv1);
        }
        // the image has been read (this wave's LDS operations execute in order): it becomes the fp32 staging slab [32][LDC]
        float* Cs = reinterpret_cast<float*>(Aw);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float* cp = Cs + ((e & 3) + 8 * (e >> 2) + 4 * h) * LDC + r;
            cp[0] = acc0[e];
            cp[32] = acc1[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x4 lo[4], hi[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = it * 8 + (lane >> 3);
            lo[it] = *reinterpret_cast<const f32x4*>(Cs + px * LDC + oc8 * 8) + b_lo;
            hi[it] = *reinterpret_cast<const f32x4*>(Cs + px * LDC + oc8 * 8 + 4) + b_hi;
        }
        // the slab has been read out (this wave's LDS operations execute in order): the next unit's image goes into it
        if constexpr (RS) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (more) write_image();
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (more && !HDBG(1)) dma_image(u + nw);
            asm volatile("" ::: "memory");          // the stores below stay behind the DMA: the counted wait relies on it
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const unsigned off = eoff[it];
            if (f_mask || f_addmask) {
                const f32x4 rl = lo4(e_ref[it]), rh = hi4(e_ref[it]);
                if (f_mask) { lo[it] = keep_pos(lo[it], rl); hi[it] = keep_pos(hi[it], rh); }
                if (f_add) {
                    f32x4 al = lo4(e_add[it]), ah = hi4(e_add[it]);
                    if (f_addmask) { al = keep_pos(al, rl); ah = keep_pos(ah, rh); }
                    lo[it] += al; hi[it] += ah;
                }
            } else if (f_add) {
                lo[it] += lo4(e_add[it]); hi[it] += hi4(e_add[it]);
            }
            bf16x8_t o;
            const bf16x4_t ol = __builtin_convertvector(lo[it], bf16x4_t), oh = __builtin_convertvector(hi[it], bf16x4_t);
            o[0] = ol[0]; o[1] = ol[1]; o[2] = ol[2]; o[3] = ol[3];
            o[4] = oh[0]; o[5] = oh[1]; o[6] = oh[2]; o[7] = oh[3];
            if (f_bnsums) {          // block-uniform; a row that does not exist loaded zeros and its gradient is zeroed below
                const f32x4 y2[2] = {lo4(e_ref[it]), hi4(e_ref[it])};
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                const bool live = off != 0xFFFFFFFFu;
                const f32x4 g2[2] = {live ? __builtin_convertvector(ol, f32x4) : zero4, live ? __builtin_convertvector(oh, f32x4) : zero4};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 gm = keep_pos(g2[q], y2[q] * bn_scale[q] + bn_shift[q]);
                    bn_s1[q] += gm;
                    bn_s2[q] += gm * (y2[q] - bn_mean[q]);
                }
            }
            if (!HDBG(8)) LOANS_STORE_B128(__builtin_bit_cast(u32x4, o), rs_out, (int)off, a.nt_out);
        }
    }
    if (f_bnsums) {
        // the eight lanes that share a channel unit (lane & 7) differ in lane >> 3: butterfly over those bits, then lanes 0..7
        // hold the wave's sums of their eight channels
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v1 = bn_s1[q][e], v2 = bn_s2[q][e];
#pragma unroll
                for (int o_ = 8; o_ < 64; o_ <<= 1) { v1 += __shfl_xor(v1, o_, 64); v2 += __shfl_xor(v2, o_, 64); }
                if (lane < 8 && !cbad) {
                    double* st = a.stats + (size_t)((blockIdx.x * 8 + wave) % LOANS_STATS_REPLICAS) * 2 * d.Cout;
                    atomic_add_f64(st + oc8 * 8 + q * 4 + e, (double)v1);
                    atomic_add_f64(st + d.Cout + oc8 * 8 + q * 4 + e, (double)v2);
                }
            }
    }
    if (f_stats) {
        st_s0 += __shfl_xor(st_s0, 32, 64); st_q0 += __shfl_xor(st_q0, 32, 64);
        st_s1 += __shfl_xor(st_s1, 32, 64); st_q1 += __shfl_xor(st_q1, 32, 64);
        if (h == 0) {
            double* st = a.stats + (size_t)((blockIdx.x * 8 + wave) % LOANS_STATS_REPLICAS) * 2 * d.Cout;
            if (r < d.Cout) { atomic_add_f64(st + r, st_s0); atomic_add_f64(st + d.Cout + r, st_q0); }
            if (r + 32 < d.Cout) { atomic_add_f64(st + r + 32, st_s1); atomic_add_f64(st + d.Cout + r + 32, st_q1); }
        }
    }
}

int launch_wsw(Halo16Args& a, hipStream_t st) {
    static loans_device_once lds_limit_set;
    constexpr size_t lds = wsw_lds_bytes();
    static_assert(lds <= 160 * 1024, "one block per CU");
    static_assert((size_t)32 * (64 + 4) * 4 <= (size_t)WSW_ABUF * 2, "the staging slab fits the wave's image buffer");
    const bool epi = a.d.flags & (LOANS_F_MASK | LOANS_F_ADDEND | LOANS_F_ADDEND_MASK | LOANS_F_BNSUMS);
    const void* fn = epi ? reinterpret_cast<const void*>(wsw_kernel<true, WSW_RS_EPI>) : reinterpret_cast<const void*>(wsw_kernel<false, true>);
    static loans_device_once lds_limit_set2;
    if (int rc_ = loans_raise_lds_limit(epi ? lds_limit_set : lds_limit_set2, fn, lds)) return rc_;
    const int units_y = (a.d.outH + 1) / 2, units_x = (a.d.outW + 15) / 16;
    const int64_t nunits = (int64_t)a.d.B * units_y * units_x;
    if (nunits >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    const int cus = loans_device_cus();
    if (cus <= 0) return LOANS_EINVAL;
    const int64_t blocks_needed = (nunits + 7) / 8;
    const int64_t grid = blocks_needed < cus ? blocks_needed : cus;
    if (epi) hipLaunchKernelGGL((wsw_kernel<true, WSW_RS_EPI>), dim3((unsigned)grid), dim3(512), lds, st, a, (int)nunits, units_y, units_x);
    else hipLaunchKernelGGL((wsw_kernel<false, true>), dim3((unsigned)grid), dim3(512), lds, st, a, (int)nunits, units_y, units_x);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <int TH, int TW, int BN, int WM, int WN, bool ADB, bool RELU>
int launch_halo_r(Halo16Args& a, hipStream_t st) {
    static loans_device_once lds_limit_set;
    constexpr size_t lds = halo_lds_bytes<TH, TW, BN, ADB>();
    static_assert(WM * WN == 8 ? lds <= 156 * 1024 : lds <= 80 * 1024, "one 512-thread block or two 256-thread blocks per CU");
    auto kern = halo16_kernel<TH, TW, BN, WM, WN, ADB, RELU>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), lds)) return rc_;
    a.tiles_y = (a.d.outH + TH - 1) / TH;
    a.tiles_x = (a.d.outW + TW - 1) / TW;
    a.tiles_n = (a.d.Cout + BN - 1) / BN;
    a.HH = TH + a.ny - 1;
    a.HW = TW + a.nx - 1;
    const int64_t nblk = (int64_t)a.d.B * a.tiles_y * a.tiles_x * a.tiles_n;
    if (nblk >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * WM * WN), lds, st, a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <int TH, int TW, int BN, int WM, int WN, bool ADB>
int launch_halo(Halo16Args& a, hipStream_t st) {
    return (a.d.flags & LOANS_F_RELU_IN) ? launch_halo_r<TH, TW, BN, WM, WN, ADB, true>(a, st)
                                         : launch_halo_r<TH, TW, BN, WM, WN, ADB, false>(a, st);
}

}  // namespace

// 1 if the descriptor is a geometry the halo kernels cover (the caller -- loans_igemm_bf16s -- has validated everything else)
int loans_halo16_covers(const loans_igemm_desc* d, int tile) {
    if (d->flags & LOANS_F_DENSE) return 0;
    if ((d->flags & LOANS_F_BNSUMS) && tile == LOANS_TILE_WS64) return 0;        // ws8_kernel's epilogue does not take the BN sums
    if (d->isy != 1 || d->isx != 1 || d->osy != 1 || d->osx != 1 || d->oy0 || d->ox0) return 0;
    if (d->gridH != d->outH || d->gridW != d->outW) return 0;
    if (d->Cin % 64 || d->ntaps > 9) return 0;
    if ((tile == LOANS_TILE_HALO_256x64 || tile == LOANS_TILE_HALO_128x64S) && d->Cin != 64) return 0;
    if ((tile == LOANS_TILE_WS64 || tile == LOANS_TILE_WSW64) && (d->Cin != 64 || d->Cout > 64 || d->ntaps != 9 || (d->flags & LOANS_F_RELU_IN))) return 0;
    int nx = 1;
    while (nx < d->ntaps && d->dy[nx] == d->dy[0]) ++nx;
    if (d->ntaps % nx) return 0;
    const int ny = d->ntaps / nx;
    if (nx > 3 || ny > 3) return 0;
    const int sdx = nx > 1 ? d->dx[1] - d->dx[0] : 1, sdy = ny > 1 ? d->dy[nx] - d->dy[0] : 1;
    if ((sdx != 1 && sdx != -1) || (sdy != 1 && sdy != -1)) return 0;
    for (int t = 0; t < d->ntaps; ++t)
        if (d->dy[t] != d->dy[0] + (t / nx) * sdy || d->dx[t] != d->dx[0] + (t % nx) * sdx) return 0;
    return 1;
}

int loans_halo16_launch(const void* in, const void* w, void* out, const float* bias, double* stats, const void* ref,
                        const void* addend, const loans_igemm_desc* d, int tile, unsigned in_bytes, unsigned w_bytes,
                        unsigned out_bytes, hipStream_t st) {
    if (!loans_halo16_covers(d, tile)) return LOANS_EINVAL;
    Halo16Args a;
    a.in = static_cast<const __bf16*>(in); a.w = static_cast<const __bf16*>(w); a.out = static_cast<__bf16*>(out);
    a.bias = bias; a.stats = stats;
    a.ref = static_cast<const __bf16*>(ref); a.addend = static_cast<const __bf16*>(addend);
    a.d = *d;
    a.Ktot = d->ntaps * d->Cin;
    a.cchunks = d->Cin / BKH;
    a.in_bytes = in_bytes; a.w_bytes = w_bytes; a.out_bytes = out_bytes;
    a.nt_out = loans_conv_nt(out_bytes);
    a.dbg = 0;
#ifdef LOANS_EXPERIMENT
    if (const char* e = getenv("LOANS_HALO_DBG")) a.dbg = atoi(e);
#endif
    if (in_bytes >= 0x80000000u || w_bytes >= 0x80000000u) return LOANS_ERANGE;       // offsets >= 2^31 mean "no load" here
    int nx = 1;
    while (nx < d->ntaps && d->dy[nx] == d->dy[0]) ++nx;
    a.nx = nx; a.ny = d->ntaps / nx;
    a.sdx = nx > 1 ? d->dx[1] - d->dx[0] : 1;
    a.sdy = a.ny > 1 ? d->dy[nx] - d->dy[0] : 1;
    a.dymin = a.sdy > 0 ? d->dy[0] : d->dy[0] - (a.ny - 1);
    a.dxmin = a.sdx > 0 ? d->dx[0] : d->dx[0] - (a.nx - 1);
    switch (tile) {
        case LOANS_TILE_HALO_128: return launch_halo<8, 16, 128, 2, 2, true>(a, st);
        case LOANS_TILE_HALO_128x64: return launch_halo<8, 16, 64, 4, 1, true>(a, st);
        case LOANS_TILE_HALO_256x64: return launch_halo<16, 16, 64, 4, 1, false>(a, st);
        case LOANS_TILE_HALO_128x64S: return launch_halo<8, 16, 64, 4, 1, false>(a, st);
        case LOANS_TILE_HALO_256x128: return launch_halo<16, 16, 128, 4, 2, true>(a, st);      // 512 threads, one block per CU
        case LOANS_TILE_HALO_256x256: return launch_halo<16, 16, 256, 2, 4, true>(a, st);      // 512 threads: eight 128 x 64 wave tiles
        case LOANS_TILE_WS64: return a.nx == 3 && a.ny == 3 ? launch_ws8(a, st) : LOANS_EINVAL;
        case LOANS_TILE_WSW64: return a.nx == 3 && a.ny == 3 ? launch_wsw(a, st) : LOANS_EINVAL;
        default: return LOANS_EINVAL;
    }
}
