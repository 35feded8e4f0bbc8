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
#include <stdlib.h>

#ifdef LOANS_STAMPS
// Diagnostic build only (tools/stamp_build.sh): per-wave cycle sums of the K-loop phases of the
// first 64 blocks, written to a buffer of their own.  Never part of libloans_hip.so.
__device__ unsigned long long g_stamps[64 * 4 * 8];
#define STAMP(t)                                                                         \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define STAMP(t) do { } while (0)
#endif

#ifdef LOANS_EXPERIMENT
#define DBGSKIP(bit) (a.dbg & (bit))
#else
#define DBGSKIP(bit) false
#endif

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;
#ifndef PRIO_AUX
#define PRIO_AUX 3
#endif

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
    // pair launch (loans_igemm_pair_f32): a second convolution of the SAME input and geometry (other weights / Cout / output /
    // statistics) takes the blocks behind the first one's: one grid, one tail
    const float* w2;
    float* out2;
    double* stats2;
    int Cout2, tiles_n2;
    unsigned w2_bytes, out2_bytes;
    int splits, chunks_per_split;   // split-K (LOANS_TILE_SPLITK): block (tile, s) contracts chunks [s * cps, (s + 1) * cps) and ADDS its tile
    int n_full, tail_splits;        // LOANS_TILE_FINETAIL (n_full > 0): the first n_full tiles at full K with the normal epilogue, the
                                    // tiles behind them in tail_splits K-slices each (raw partial tiles, like split-K)
    int m_begin;        // first GEMM row of this launch (LOANS_TILE_SPLIT runs a row range per tile shape); rows end at M
    int tail_groups;    // 8-deep k groups of the last chunk that hold any real K (1..4)
    int bf16;           // 1: round the operands to bf16 and use the bf16 MFMA (fp32 accumulate)
    int dma;            // 1: LDS-DMA staging of the operand tiles (LOANS_TILE_DMA; fp32 arm)
    int nt_out;         // output stores non-temporal (loans_conv_nt)
    int dbg;            // experiment bits (LOANS_EXPERIMENT builds only; 0 in the product library)
    unsigned in_bytes, w_bytes, out_bytes;
    struct {            // nx > 0: taps are an ny x nx grid, dy = dy0 + row*sdy, dx = dx0 + col*sdx, sd* = +-1
        int nx, ny, dy0, sdy, dx0, sdx;
        unsigned long long rowpat;      // bit (row * nx) set for every row
    } ap;
    // class launch (loans_igemm_classes_f32): the stride-parity classes of ONE strided data gradient share a grid and a tail.
    // Everything that differs between classes lives here and the kernel reads it through `k` -- an ordinary launch is a class
    // launch with ncls = 1 (cls[0] mirrors d / ap / M / Ktot).
    int ncls;
    struct Cls {
        const float* w;
        unsigned w_bytes;
        int gridH, gridW, oy0, ox0, ntaps;
        int M, Ktot, nchunks, tail_groups, tiles_m, blk0, per_xcd;
        int nx, ny, dy0, sdy, dx0, sdx;
        unsigned long long rowpat;
        signed char dy[LOANS_MAX_CLS_TAPS], dx[LOANS_MAX_CLS_TAPS];
    } cls[LOANS_MAX_CLASSES];
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// XCD-aware, bijective block remap: blocks that share an XCD (id % 8) get a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int id, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// sched_group_barrier masks (LLVM SchedGroupMask)
#define SG_VALU 0x2
#define SG_MFMA 0x8
#define SG_VMEM_READ 0x20
#define SG_DS_READ 0x100
#define SG_DS_WRITE 0x200

typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int LDKH = BK + 8;      // bf16 LDS row (halves): 80 bytes, ds_read_b128 conflict-free

// BF16 = true: the fp32 operands are rounded to bf16 (RNE) while they are staged into LDS and contracted on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 16x the fp32 MFMA rate); prologue, epilogue and the C ABI are shared.
// DMA = true (fp32 arm): the operand tiles are staged by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging
// registers, no ds_write pass.  A wave-instruction writes 64 x 16 B contiguously, so LDS rows are unpadded
// [row][32 floats] and bank conflicts are avoided by an XOR swizzle instead: the 16-byte unit u of row r lives in
// slot u ^ ((r >> 1) & 7) -- applied to the per-lane SOURCE address by the loader and to the fragment reads
// (every 16-lane group of a ds_read_b128 then covers all 64 banks once).
template <int BM, int BN, bool DMA>
constexpr size_t igemm_aux_floats() {       // offset (floats) of the tap table / row table behind the tiles
    constexpr size_t stage = (size_t)2 * (BM + BN) * (DMA ? BK : LDK);
    constexpr size_t cs = (size_t)BM * (BN + 4);             // epilogue staging tile
    return stage > cs ? stage : cs;
}

template <int BM, int BN, int WM, int WN, bool RELU, bool BF16, bool DMA>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs a) {
    static_assert(!(DMA && BF16), "the bf16 arm converts while it stages through registers");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int NMMA = TM * TN * 4;          // MFMAs per 8-deep k group
    constexpr int LDR = DMA ? BK : LDK;        // LDS row stride (floats)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);        // [2][BM][LDR]
    float* Bs = As + 2 * BM * LDR;                     // [2][BN][LDR]
    int* taps = reinterpret_cast<int*>(As + igemm_aux_floats<BM, BN, DMA>());
    unsigned* opix = reinterpret_cast<unsigned*>(taps + LOANS_MAX_TAPS);   // [BM] output row byte offset, ~0u = no row

    const loans_igemm_desc& d = a.d;
    const int tid = threadIdx.x;
#ifdef LOANS_STAMPS
    unsigned long long t_start = 0;
    STAMP(t_start);
#endif
    // prologue and epilogue run at raised wave priority: their scalar / vector bookkeeping then is not queued behind the
    // MFMA streams of the co-resident blocks (+1.5 % on the short-K stem / res2 tiles, neutral elsewhere)
    __builtin_amdgcn_s_setprio(3);
    // mixed grid (LOANS_TILE_FINETAIL): blocks [0, n_full) are whole tiles in XCD order, the blocks behind them K-slices of
    // the remaining tiles, again in XCD order among themselves -- dispatched last, they even out the CUs' finish times
    const bool fine = a.n_full > 0;
    const bool fine_tail = fine && (int)blockIdx.x >= a.n_full;
    const int logical = !fine ? xcd_remap(blockIdx.x, gridDim.x)
                              : (fine_tail ? xcd_remap(blockIdx.x - a.n_full, gridDim.x - a.n_full) : xcd_remap(blockIdx.x, a.n_full));
    const int my_splits = fine ? (fine_tail ? a.tail_splits : 1) : a.splits;
    // class launch: block-uniform choice of the stride-parity class this block works for (class 0 in every other launch)
    // Each XCD (blockIdx % 8) gets a contiguous eighth of EVERY class, longest K first: classes differ in K (1, 2, 2 and 4 taps
    // for 3x3 / 2), so whole classes per XCD would leave some XCDs with 4x the work of others.  blk0 / per_xcd count blocks
    // per XCD; a class's tile count is padded to a multiple of 8 and the (< 8) surplus blocks leave here.
    int ci = 0, lcls = logical;
    if (a.ncls > 1) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        for (int c = 1; c < a.ncls; ++c) ci += j >= a.cls[c].blk0;
        lcls = x * a.cls[ci].per_xcd + (j - a.cls[ci].blk0);
        if (lcls >= a.cls[ci].tiles_m * a.tiles_n) return;
    }
    const IgemmArgs::Cls& k = a.cls[ci];
    const signed char* const tap_dy = a.ncls > 1 ? k.dy : reinterpret_cast<const signed char*>(d.dy);
    const signed char* const tap_dx = a.ncls > 1 ? k.dx : reinterpret_cast<const signed char*>(d.dx);
    // pair launch: the blocks behind the first convolution's tiles belong to the second one (block-uniform selection)
    const bool second = a.w2 != nullptr && logical >= k.tiles_m * a.tiles_n;
    const int Cout = second ? a.Cout2 : d.Cout;
    const int tiles_n = second ? a.tiles_n2 : a.tiles_n;
    const float* const w_sel = second ? a.w2 : k.w;
    float* const out_sel = second ? a.out2 : a.out;
    double* const stats_sel = second ? a.stats2 : a.stats;
    const unsigned w_bytes = second ? a.w2_bytes : k.w_bytes, out_bytes = second ? a.out2_bytes : a.out_bytes;
    const int lfirst = second ? logical - k.tiles_m * a.tiles_n : lcls;
    const int ntile = k.tiles_m * tiles_n;
    int split = a.w2 ? 0 : lfirst / ntile;              // 0 unless split-K
    int ltile = lfirst - split * ntile;
    if (fine) {
        const int n_tail = ntile - a.n_full;
        split = fine_tail ? logical / n_tail : 0;
        ltile = fine_tail ? a.n_full + (logical - split * n_tail) : logical;
    }
    const int tn = ltile % tiles_n;
    const int tm = ltile / tiles_n;
    const int cps = (fine && !fine_tail) ? k.nchunks : (a.ncls > 1 ? k.nchunks : a.chunks_per_split);
    const int c_begin = split * cps;                    // this block's K chunks
    const int c_end = min(c_begin + cps, k.nchunks);
    const int nch = c_end - c_begin;
    const int tail_groups = c_end == k.nchunks ? k.tail_groups : 4;
    const int lrow = tid >> 3;
    const int lu = DMA ? ((tid & 7) ^ ((tid >> 4) & 7)) : (tid & 7);   // K unit this thread stages (DMA: slot ^ row key)
    // tap table in LDS: byte offset of tap t relative to the row's base pixel
    // LOANS_F_DENSE: inW / isx / dx count floats (packed 3-channel rows), a "tap" is a run of Cin consecutive
    // floats of one input row, and the caller's zero padding makes every tap of every pixel readable
    const bool dense = d.flags & LOANS_F_DENSE;
    const int ubytes = dense ? 4 : d.Cin * 4;          // bytes per unit of inW / ix
    if (tid < LOANS_MAX_TAPS) {
        const int t = tid < k.ntaps ? tid : 0;
        taps[tid] = (int(tap_dy[t]) * d.inW + int(tap_dx[t])) * ubytes;
    }

    // per row (fixed for the whole K loop): byte offset of its base pixel and a bitmask with bit t SET
    // when tap t must read zero (outside the image, beyond ntaps, or the row does not exist).
    // Every conv on this path has taps on an ny x nx grid whose dy / dx run in unit steps, so the in-bounds
    // taps of a row are an index RANGE per axis and the mask is two shifts and a multiply -- no loops, no
    // table reads.  (b, y, x) of the first row comes from two divisions, the others advance by 32 pixels.
    unsigned rowoff[RA];
    unsigned long long badmask[RA];
    {
        const int gridH = k.gridH, gridW = k.gridW;
        const int gHW = gridH * gridW;
        const float inv_gw = 1.f / (float)gridW, inv_gh = 1.f / (float)gridH;
        const int m0 = a.m_begin + tm * BM + lrow;
        int b = m0 / gHW;
        int rem = m0 - b * gHW;
        int y = rem / gridW;
        int x = rem - y * gridW;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + 32 * i;
            unsigned pixoff = 0xFFFFFFFFu;
            unsigned long long mask = 0;
            rowoff[i] = 0;
            if (m < k.M) {
                const int iy0 = y * d.isy, ix0 = x * d.isx;
                rowoff[i] = (unsigned)((b * d.inH + iy0) * d.inW + ix0) * (unsigned)ubytes;
                pixoff = (unsigned)((b * d.outH + y * d.osy + k.oy0) * d.outW + x * d.osx + k.ox0) * (unsigned)Cout *
                         ((d.flags & LOANS_F_OUT_BF16) ? 2u : 4u);
                if (dense) {
                    mask = ~0ull;
                } else if (k.nx > 0) {
                    // column j valid <=> 0 <= ix0 + dx0 + j*sdx < inW  (sdx = +-1): a contiguous j range
                    const int cx = ix0 + k.dx0, cy = iy0 + k.dy0;
                    int jlo, jhi, rlo, rhi;
                    if (k.sdx > 0) { jlo = max(0, -cx); jhi = min(k.nx, d.inW - cx); }
                    else { jlo = max(0, cx - d.inW + 1); jhi = min(k.nx, cx + 1); }
                    if (k.sdy > 0) { rlo = max(0, -cy); rhi = min(k.ny, d.inH - cy); }
                    else { rlo = max(0, cy - d.inH + 1); rhi = min(k.ny, cy + 1); }
                    if (jhi > jlo && rhi > rlo) {
                        const unsigned long long colbits = ((1ull << jhi) - 1ull) & ~((1ull << jlo) - 1ull);
                        const int blo = rlo * k.nx, bhi = rhi * k.nx;     // bhi <= 64
                        const unsigned long long below_hi = bhi >= 64 ? ~0ull : ((1ull << bhi) - 1ull);
                        const unsigned long long rowsel = k.rowpat & below_hi & ~((1ull << blo) - 1ull);
                        mask = colbits * rowsel;       // colbits < 2^nx, rowsel bits nx apart: no carries
                    }
                } else {
                    for (int t = 0; t < k.ntaps; ++t) {
                        const int iy = iy0 + tap_dy[t], ix = ix0 + tap_dx[t];
                        if ((unsigned)iy < (unsigned)d.inH && (unsigned)ix < (unsigned)d.inW) mask |= 1ull << t;
                    }
                }
            }
            badmask[i] = ~mask;
#ifdef LOANS_EXPERIMENT
            if (a.dbg & 4) { rowoff[i] = (unsigned)((d.inW + 1) * ubytes) + (rowoff[i] & 0xFFFu); badmask[i] = 0; }   // cache-hot gathers
#endif
            if (lu == 0) opix[lrow + 32 * i] = pixoff;
            // advance 32 pixels: exact floor((v + .5) / n) for the small integers involved
            x += 32;
            const int qx = (int)(((float)x + 0.5f) * inv_gw);
            x -= qx * gridW;
            y += qx;
            const int qy = (int)(((float)y + 0.5f) * inv_gh);
            y -= qy * gridH;
            b += qy;
        }
    }
    __syncthreads();

    // bounds-checked buffer descriptors: an offset beyond num_records loads zeros / drops the store, so
    // padding, ragged tiles and K tails need no branches: invalid accesses get offset 0xFFFFFFFF
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(w_sel), 0, (int)w_bytes, 0x00020000);

    const int cpt = d.Cin >> 2;   // float4 units per tap
    const int Ktot = k.Ktot;
    const int q8 = 8 / cpt, r8 = 8 - q8 * cpt;
    int u = lu + 8 * c_begin;     // this thread's K unit in the chunk being loaded
    int tap = u / cpt, c4 = u - tap * cpt;
    unsigned woff[RB], wbad[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = tn * BN + lrow + 32 * i;
        wbad[i] = n < Cout ? 0u : 0xFFFFFFFFu;
        woff[i] = n < Cout ? (unsigned)n * (unsigned)Ktot * 4u : 0u;
    }
    unsigned toff = (unsigned)taps[min(tap, LOANS_MAX_TAPS - 1)] + (unsigned)c4 * 16u;   // prefetched one chunk ahead

    f32x4 ra[RA], rb[RB];
    // The loader is cut into RA + RB + 1 independent pieces (one buffer load each, then the advance) so
    // that the K loop can drop one piece behind each MFMA of group 0: no waits, no branches.
    auto load_a = [&](int i) {
        const int tc = min(tap, LOANS_MAX_TAPS - 1);
        const unsigned kbad = (unsigned)(u * 4 < Ktot) - 1u;
        const unsigned bad = 0u - ((unsigned)(badmask[i] >> tc) & 1u);
        const unsigned off = (rowoff[i] + toff) | bad | kbad;
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)off, 0, 0));
    };
    auto load_b = [&](int i) {
        const unsigned kbad = (unsigned)(u * 4 < Ktot) - 1u;
        const unsigned off = (woff[i] + (unsigned)u * 16u) | wbad[i] | kbad;
        rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)off, 0, 0));
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto dma_a = [&](int buf, int i) {      // one 1 KiB LDS-DMA piece: 8 rows x 128 B of the A tile
        const int tc = min(tap, LOANS_MAX_TAPS - 1);
        const unsigned kbad = (unsigned)(u * 4 < Ktot) - 1u;
        const unsigned bad = 0u - ((unsigned)(badmask[i] >> tc) & 1u);
        const unsigned off = (rowoff[i] + toff) | bad | kbad;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(As + (buf * BM + 32 * i + 8 * wave_u) * BK), 16, (int)off, 0, 0, 0);
    };
    auto dma_b = [&](int buf, int i) {
        const unsigned kbad = (unsigned)(u * 4 < Ktot) - 1u;
        const unsigned off = (woff[i] + (unsigned)u * 16u) | wbad[i] | kbad;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + (buf * BN + 32 * i + 8 * wave_u) * BK), 16, (int)off, 0, 0, 0);
    };
    auto advance = [&]() {      // to the following chunk (8 units further along K); prefetch its tap offset
        u += 8;                 // 8 = q8 * cpt + r8: whole taps, then at most one wrap -- no branch, no division
        tap += q8;
        c4 += r8;
        const int wrap = c4 >= cpt;
        c4 -= wrap ? cpt : 0;
        tap += wrap;
        toff = (unsigned)taps[min(tap, LOANS_MAX_TAPS - 1)] + (unsigned)c4 * 16u;
    };
    auto load_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < RA; ++i) load_a(i);
#pragma unroll
        for (int i = 0; i < RB; ++i) load_b(i);
        advance();
    };
    auto store_a = [&](int buf, int i) {
        if (RELU) {
            ra[i].x = fmaxf(ra[i].x, 0.f); ra[i].y = fmaxf(ra[i].y, 0.f);
            ra[i].z = fmaxf(ra[i].z, 0.f); ra[i].w = fmaxf(ra[i].w, 0.f);
        }
        if constexpr (BF16) {
            __bf16* Ah = reinterpret_cast<__bf16*>(smem);
            *reinterpret_cast<bf16x4_t*>(Ah + buf * BM * LDKH + (lrow + 32 * i) * LDKH + lu * 4) = __builtin_convertvector(ra[i], bf16x4_t);
        } else {
            *reinterpret_cast<f32x4*>(As + buf * BM * LDK + (lrow + 32 * i) * LDK + lu * 4) = ra[i];
        }
    };
    auto store_b = [&](int buf, int i) {
        if constexpr (BF16) {
            __bf16* Bh = reinterpret_cast<__bf16*>(smem) + 2 * BM * LDKH;
            *reinterpret_cast<bf16x4_t*>(Bh + buf * BN * LDKH + (lrow + 32 * i) * LDKH + lu * 4) = __builtin_convertvector(rb[i], bf16x4_t);
        } else {
            *reinterpret_cast<f32x4*>(Bs + buf * BN * LDK + (lrow + 32 * i) * LDK + lu * 4) = rb[i];
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) store_a(buf, i);
#pragma unroll
        for (int i = 0; i < RB; ++i) store_b(buf, i);
    };

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // DMA: unit 2g + h of row r sits in slot (2g + h) ^ key, key = (r >> 1) & 7: the address of group g is the
    // address of group 0 with g XOR-ed into the slot's upper two bits
    const int fkey = (r >> 1) & 7;
    const int fragA = DMA ? (wm * TM * 32 + r) * BK + ((h ^ fkey) & 7) * 4 : (wm * TM * 32 + r) * LDK + h * 4;
    const int fragB = DMA ? (wn * TN * 32 + r) * BK + ((h ^ fkey) & 7) * 4 : (wn * TN * 32 + r) * LDK + h * 4;
    auto read_frag = [&](int buf, int g, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
        const float* Ab = DMA ? As + buf * BM * BK + (fragA ^ (g * 8)) : As + buf * BM * LDK + fragA + g * 8;
        const float* Bb = DMA ? Bs + buf * BN * BK + (fragB ^ (g * 8)) : Bs + buf * BN * LDK + fragB + g * 8;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDR);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDR);
    };
    auto relu_frag = [&](f32x4 (&af)[TM]) {      // DMA arm: relu(in) is applied to the fragments (no staging registers)
        if constexpr (DMA && RELU) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                af[i].x = fmaxf(af[i].x, 0.f); af[i].y = fmaxf(af[i].y, 0.f);
                af[i].z = fmaxf(af[i].z, 0.f); af[i].w = fmaxf(af[i].w, 0.f);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#ifdef LOANS_STAMPS
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, s_load = 0, s_mfma = 0, s_store = 0, s_bar = 0, t_begin = 0;
#endif
    if constexpr (BF16) {
        // ---- bf16 MFMA K loop: a chunk (32 of K) is two 32x32x16 steps per tile; the loop is bound by the
        // operand traffic, not the matrix pipe, so it stays simple: next chunk's loads in flight across the
        // MFMAs of the current one, conversion + LDS write behind them, one barrier per chunk.
        const __bf16* Ah = reinterpret_cast<const __bf16*>(smem);
        const __bf16* Bh = Ah + 2 * BM * LDKH;
        const int hA = (wm * TM * 32 + r) * LDKH + h * 8;
        const int hB = (wn * TN * 32 + r) * LDKH + h * 8;
        load_chunk();
        store_chunk(0);
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);
        for (int c = 0; c < nch; ++c) {
            const int buf = c & 1;
            const bool more = (c + 1) < nch;
            if (more) load_chunk();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8_t af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8_t*>(Ah + buf * BM * LDKH + hA + i * 32 * LDKH + s * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8_t*>(Bh + buf * BN * LDKH + hB + j * 32 * LDKH + s * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            if (more) store_chunk(buf ^ 1);
            __syncthreads();
        }
    } else if constexpr (DMA) {
    // ---- LDS-DMA K loop: the pieces of chunk c+1 are issued one behind each of the first MFMAs of chunk c,
    // straight into the other LDS stage (free since the barrier that ended chunk c-1); they land while groups
    // 0..2 compute, __syncthreads() drains them (hipcc puts vmcnt(0) in front of the barrier while an LDS-DMA is
    // outstanding) and group 3's MFMAs run behind the barrier as in the register-staged loop.
    auto mma_one = [&](int s, const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
        const int kk = s / (TM * TN), i = (s / TN) % TM, j = s % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
    };
    auto mma = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
#pragma unroll
        for (int s = 0; s < NMMA; ++s) mma_one(s, af, bf);
    };
    constexpr int NPIECE = RA + RB + 1;
    static_assert(NPIECE <= 2 * NMMA, "one piece per MFMA gap of groups 0 and 1");
    auto dma_piece = [&](int buf, int p) {
        if (p < RA) dma_a(buf, p);
        else if (p < RA + RB) dma_b(buf, p - RA);
        else if (p == RA + RB) advance();
    };
    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) dma_piece(0, p);
    __syncthreads();
    read_frag(0, 0, fa0, fb0);
    __builtin_amdgcn_s_setprio(0);
    int c = 0;
    for (; c + 1 < nch; ++c) {
        const int buf = c & 1;
        read_frag(buf, 1, fa1, fb1);
        relu_frag(fa0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NMMA; ++s) {
            mma_one(s, fa0, fb0);
            dma_piece(buf ^ 1, s);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frag(buf, 2, fa0, fb0);
        relu_frag(fa1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NMMA; ++s) {
            mma_one(s, fa1, fb1);
            dma_piece(buf ^ 1, NMMA + s);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frag(buf, 3, fa1, fb1);
        relu_frag(fa0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        relu_frag(fa1);
        __syncthreads();
        read_frag(buf ^ 1, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
    }
    {   // last chunk (see the register-staged loop)
        const int buf = c & 1;
        const int tg = tail_groups;
        if (tg > 1) read_frag(buf, 1, fa1, fb1);
        relu_frag(fa0);
        mma(fa0, fb0);
        if (tg > 1) {
            if (tg > 2) read_frag(buf, 2, fa0, fb0);
            relu_frag(fa1);
            mma(fa1, fb1);
            if (tg > 2) {
                if (tg > 3) read_frag(buf, 3, fa1, fb1);
                relu_frag(fa0);
                mma(fa0, fb0);
                if (tg > 3) { relu_frag(fa1); mma(fa1, fb1); }
            }
        }
    }
    } else {
    // MFMA number s (0 .. NMMA-1) of a k group
    auto mma_one = [&](int s, const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
        const int kk = s / (TM * TN), i = (s / TN) % TM, j = s % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
    };
    auto mma = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
#pragma unroll
        for (int s = 0; s < NMMA; ++s) mma_one(s, af, bf);
    };

    // ---- software-pipelined K loop ---------------------------------------------------------------
    // A chunk (32 of K) = 4 groups of 8; fragments of group g+1 are read from LDS while the MFMAs of
    // group g run; the loader pieces of chunk c+1 ride one behind each MFMA of group 0, their LDS
    // writes one behind each MFMA of group 2 (order pinned with sched_barrier: a wave issues in order,
    // and anything queued between two of its own MFMAs is free), and group 3's MFMAs (operands already
    // in registers) run after the barrier so that they cover its skew and the next chunk's first reads.
    constexpr int NPIECE = RA + RB + 1;                       // loader pieces (last = advance)
    constexpr int PPG = (NPIECE + NMMA - 1) / NMMA;           // pieces per MFMA gap
    auto load_piece = [&](int p) {
        if (p < RA) { if (!DBGSKIP(16)) load_a(p); }
        else if (p < RA + RB) { if (!DBGSKIP(16)) load_b(p - RA); }
        else if (p == RA + RB) advance();
    };
    auto store_piece = [&](int buf, int p) {
        if (DBGSKIP(32)) return;
        if (p < RA) store_a(buf, p);
        else if (p < RA + RB) store_b(buf, p - RA);
    };
    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    load_chunk();
    store_chunk(0);
    __syncthreads();
    read_frag(0, 0, fa0, fb0);
    __builtin_amdgcn_s_setprio(0);

    STAMP(t_begin);
    int c = 0;
    for (; c + 1 < nch; ++c) {
        const int buf = c & 1;
        STAMP(t0);
        // group 0 (+ loader)
        if (!DBGSKIP(128)) read_frag(buf, 1, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NMMA; ++s) {
            mma_one(s, fa0, fb0);
#pragma unroll
            for (int p = s * PPG; p < (s + 1) * PPG; ++p) load_piece(p);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(t1);
        // group 1
        if (!DBGSKIP(128)) read_frag(buf, 2, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        // group 2 (+ LDS writes of the staged chunk into the other buffer)
        if (!DBGSKIP(128)) read_frag(buf, 3, fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NMMA; ++s) {
            mma_one(s, fa0, fb0);
#pragma unroll
            for (int p = s * PPG; p < (s + 1) * PPG; ++p) store_piece(buf ^ 1, p);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(t2);
        // group 3 runs behind the barrier with operands that are already in registers
        if (!DBGSKIP(64)) __syncthreads();
        STAMP(t3);
        if (!DBGSKIP(128)) read_frag(buf ^ 1, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(t4);
#ifdef LOANS_STAMPS
        s_load += t1 - t0; s_mfma += t2 - t1; s_store += t3 - t2; s_bar += t4 - t3;
#endif
    }
    {   // last chunk: nothing left to stage; 8-deep groups that lie wholly beyond Ktot hold zeros on both sides
        // and are skipped (block-uniform branch): exact, and a quarter of the stem's K is such padding
        const int buf = c & 1;
        const int tg = tail_groups;
        if (tg > 1) read_frag(buf, 1, fa1, fb1);
        mma(fa0, fb0);
        if (tg > 1) {
            if (tg > 2) read_frag(buf, 2, fa0, fb0);
            mma(fa1, fb1);
            if (tg > 2) {
                if (tg > 3) read_frag(buf, 3, fa1, fb1);
                mma(fa0, fb0);
                if (tg > 3) mma(fa1, fb1);
            }
        }
    }
    }   // fp32 / bf16 K loop
#ifdef LOANS_STAMPS
    unsigned long long t_loop_end = 0;
    STAMP(t_loop_end);
#endif

    // ---- epilogue -----------------------------------------------------------------------------------
    // BN statistics come straight from the accumulators; the tile itself is staged through LDS (the K-loop
    // buffers are free now) so that every lane stores 16 contiguous bytes: BN/4 lanes cover one output row.
    const bool f_bias = d.flags & LOANS_F_BIAS, f_stats = d.flags & LOANS_F_STATS;
    const bool f_mask = d.flags & LOANS_F_MASK, f_add = d.flags & LOANS_F_ADDEND;
    const bool f_addmask = d.flags & LOANS_F_ADDEND_MASK;
    const bool f_bnsums = d.flags & LOANS_F_BNSUMS;      // the sums of the BN below a data gradient, from the tile (loans_hip.h)
    constexpr int LDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(smem);          // [BM][LDC]
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();                                     // every wave is done with the fragment buffers
    if (f_stats && my_splits == 1) {
        int nvalid = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                nvalid += opix[wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] != 0xFFFFFFFFu;
        const float cnt = (float)nvalid;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = tn * BN + wn * TN * 32 + j * 32 + r;
            const bool cok = col < Cout;
            const float bv = (f_bias && cok) ? a.bias[col] : 0.f;
            float s = 0.f, q2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {      // rows that do not exist gathered zeros: acc == 0 there
                    s += acc[i][j][e];
                    q2 += acc[i][j][e] * acc[i][j][e];
                }
            // statistics of (acc + bias) over the rows that exist, from the raw sums
            q2 = q2 + 2.f * bv * s + cnt * bv * bv;
            s = s + cnt * bv;
            s += __shfl_xor(s, 32, 64);
            q2 += __shfl_xor(q2, 32, 64);
            if (h == 0 && cok) {       // LOANS_STATS_REPLICAS accumulators, picked by block, against contention
                double* st = stats_sel + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * Cout;
                atomic_add_f64(st + col, (double)s);
                atomic_add_f64(st + Cout + col, (double)q2);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                Cs[(wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + wn * TN * 32 + j * 32 + r] = acc[i][j][e];
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out_sel, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ref = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.ref ? a.ref : out_sel), 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.addend ? a.addend : out_sel), 0, (int)out_bytes, 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int CPR = BN / 4;                 // float4 columns per row
    constexpr int RSTEP = 256 / CPR;            // rows covered by the block per pass
    const int oc4 = tid % CPR, r0 = tid / CPR;
    const int col0 = tn * BN + oc4 * 4;
    const unsigned cbad = (col0 + 3 < Cout) ? 0u : 0xFFFFFFFFu;       // Cout % 4 == 0 on every layer here
    const bool f_out16 = d.flags & LOANS_F_OUT_BF16;      // bf16 output tensor (no ref / addend in this mode: checked)
    const unsigned coff = (unsigned)col0 * (f_out16 ? 2u : 4u);
    f32x4 bv4 = {0.f, 0.f, 0.f, 0.f};
    if (f_bias && !cbad) bv4 = *reinterpret_cast<const f32x4*>(a.bias + col0);
    // LOANS_F_BNSUMS: a.bias = the BN's coefficient table [mean | rstd | scale | shift][Cout]; this thread's four channels
    f32x4 bn_mean = {0.f, 0.f, 0.f, 0.f}, bn_scale = bn_mean, bn_shift = bn_mean, bn_s1 = bn_mean, bn_s2 = bn_mean;
    if (f_bnsums && !cbad) {
        bn_mean = *reinterpret_cast<const f32x4*>(a.bias + col0);
        bn_scale = *reinterpret_cast<const f32x4*>(a.bias + 2 * Cout + col0);
        bn_shift = *reinterpret_cast<const f32x4*>(a.bias + 3 * Cout + col0);
        bv4 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (my_splits > 1) {
        // split-K: the raw partial tile is added to `out` (zeroed, or holding the addend, by the caller); bias, statistics,
        // mask and addend are applied to the finished sums by loans_igemm_finalize_f32
#pragma unroll
        for (int p = 0; p < BM / RSTEP; ++p) {
            const int row = r0 + p * RSTEP;
            const unsigned po = opix[row];
            if (po == 0xFFFFFFFFu || cbad) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * LDC + oc4 * 4);
            float* dst = out_sel + ((po + coff) >> 2);
            atomic_add_f32(dst + 0, v.x); atomic_add_f32(dst + 1, v.y);
            atomic_add_f32(dst + 2, v.z); atomic_add_f32(dst + 3, v.w);
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < BM / RSTEP; ++p) {
        const int row = r0 + p * RSTEP;
        const unsigned po = opix[row];
        const unsigned off = (po + coff) | (po == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u) | cbad;
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * LDC + oc4 * 4) + bv4;
        if (f_mask || f_addmask) {
            const f32x4 rf = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, (int)off, 0, 0));
            if (f_mask) {
                v.x = rf.x > 0.f ? v.x : 0.f; v.y = rf.y > 0.f ? v.y : 0.f;
                v.z = rf.z > 0.f ? v.z : 0.f; v.w = rf.w > 0.f ? v.w : 0.f;
            }
            if (f_add) {
                f32x4 ad = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)off, 0, 0));
                if (f_addmask) {
                    ad.x = rf.x > 0.f ? ad.x : 0.f; ad.y = rf.y > 0.f ? ad.y : 0.f;
                    ad.z = rf.z > 0.f ? ad.z : 0.f; ad.w = rf.w > 0.f ? ad.w : 0.f;
                }
                v += ad;
            }
        } else if (f_add) {
            v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)off, 0, 0));
        }
        if (f_bnsums) {
            // (block-uniform test only: the load of a row that does not exist goes out of range and returns zeros, its gradient is
            // forced to zero -- a per-lane test around the load kept the unrolled rows' loads from being issued together)
            const f32x4 y = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, (int)off, 0, 0));
            const f32x4 z = y * bn_scale + bn_shift;
            const bool live = off != 0xFFFFFFFFu;
            f32x4 gm;
            gm.x = (live && z.x > 0.f) ? v.x : 0.f; gm.y = (live && z.y > 0.f) ? v.y : 0.f;
            gm.z = (live && z.z > 0.f) ? v.z : 0.f; gm.w = (live && z.w > 0.f) ? v.w : 0.f;
            bn_s1 += gm;
            bn_s2 += gm * (y - bn_mean);
        }
#ifdef LOANS_EXPERIMENT
        if ((a.dbg & 8) && v.x != 12345.f) continue;      // no output stores
#endif
        if (f_out16) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            {
                const u32x2 ov = __builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4_t));
                if (a.nt_out) __builtin_amdgcn_raw_buffer_store_b64(ov, rs_out, (int)off, 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b64(ov, rs_out, (int)off, 0, 0);
            }
            continue;
        }
        LOANS_STORE_B128(__builtin_bit_cast(u32x4, v), rs_out, (int)off, a.nt_out);
    }
    if (f_bnsums) {
        // the threads that share a channel quad (same oc4, RSTEP rows apart) are summed through LDS, then one fp64 atomic per
        // channel and sum into this block's replica
        __syncthreads();
        float* Red = reinterpret_cast<float*>(smem);            // [256 / CPR][CPR][8]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Red[(r0 * CPR + oc4) * 8 + e] = bn_s1[e];
            Red[(r0 * CPR + oc4) * 8 + 4 + e] = bn_s2[e];
        }
        __syncthreads();
        if (tid < CPR * 8) {
            const int u4 = tid >> 3, j = tid & 7;               // channel quad, (sum, channel of the quad)
            float acc_ = 0.f;
#pragma unroll 4
            for (int rr = 0; rr < 256 / CPR; ++rr) acc_ += Red[(rr * CPR + u4) * 8 + j];
            const int col = tn * BN + u4 * 4 + (j & 3);
            if (col < Cout) {
                double* st = stats_sel + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * Cout;
                atomic_add_f64(st + (j >> 2) * Cout + col, (double)acc_);
            }
        }
    }
#ifdef LOANS_STAMPS
    unsigned long long t_end = 0;
    STAMP(t_end);
    if (logical < 64 && lane == 0) {
        unsigned long long* o = g_stamps + (logical * 4 + wave) * 8;
        o[0] = s_load; o[1] = s_mfma; o[2] = s_store; o[3] = s_bar;
        o[4] = t_loop_end - t_begin; o[5] = t_end - t_loop_end; o[6] = a.nchunks; o[7] = t_begin - t_start;
    }
#endif
}

template <int BM, int BN, bool DMA>
constexpr size_t igemm_lds_bytes() {
    return igemm_aux_floats<BM, BN, DMA>() * 4 + LOANS_MAX_TAPS * 4 + BM * 4;
}

// an ordinary launch is a class launch with one class: the kernel reads the per-class view only
void fill_class0(IgemmArgs& a) {
    IgemmArgs::Cls& k = a.cls[0];
    a.ncls = 1;
    k.w = a.w; k.w_bytes = a.w_bytes;
    k.gridH = a.d.gridH; k.gridW = a.d.gridW; k.oy0 = a.d.oy0; k.ox0 = a.d.ox0; k.ntaps = a.d.ntaps;
    k.M = a.M; k.Ktot = a.Ktot; k.nchunks = a.nchunks; k.tail_groups = a.tail_groups; k.tiles_m = a.tiles_m; k.blk0 = 0; k.per_xcd = 0;
    k.nx = a.ap.nx; k.ny = a.ap.ny; k.dy0 = a.ap.dy0; k.sdy = a.ap.sdy; k.dx0 = a.ap.dx0; k.sdx = a.ap.sdx;
    k.rowpat = a.ap.rowpat;
}

template <int BM, int BN, int WM, int WN, bool RELU, bool BF16, bool DMA>
int launch_igemm_r(IgemmArgs& a, hipStream_t st) {
    static loans_device_once lds_limit_set;       // per template instance = per kernel, one bit per device
#ifdef LOANS_STAMPS
    // diagnostic: LOANS_DBG_LDS=<bytes> pads the LDS request to force fewer blocks per CU
    const char* dbg_lds = getenv("LOANS_DBG_LDS");
    const size_t lds = dbg_lds ? (size_t)atol(dbg_lds) : igemm_lds_bytes<BM, BN, DMA>();
#else
    constexpr size_t lds = igemm_lds_bytes<BM, BN, DMA>();
#endif
    auto kern = igemm_kernel<BM, BN, WM, WN, RELU, BF16, DMA>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), lds)) return rc_;
    a.tiles_m = (a.M - a.m_begin + BM - 1) / BM;
    a.tiles_n = (a.d.Cout + BN - 1) / BN;
    a.tiles_n2 = a.w2 ? (a.Cout2 + BN - 1) / BN : 0;
    if (a.ncls > 1) {           // class launch: every class at full K, its tiles behind the previous class's
        if (a.w2 || a.splits != 1 || a.n_full > 0 || a.m_begin) return LOANS_EINVAL;
        int per_xcd = 0;
        for (int c = 0; c < a.ncls; ++c) {
            a.cls[c].tiles_m = (a.cls[c].M + BM - 1) / BM;
            a.cls[c].per_xcd = (a.cls[c].tiles_m * a.tiles_n + 7) / 8;
            a.cls[c].blk0 = per_xcd;
            per_xcd += a.cls[c].per_xcd;
        }
        a.chunks_per_split = a.cls[0].nchunks;
        hipLaunchKernelGGL(kern, dim3(8 * per_xcd), dim3(256), lds, st, a);
        LOANS_LAUNCH_CHECK();
        return LOANS_OK;
    }
    fill_class0(a);
    if (a.splits > a.nchunks) a.splits = a.nchunks;
    a.chunks_per_split = (a.nchunks + a.splits - 1) / a.splits;
    a.splits = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    int nblk = a.w2 ? a.tiles_m * (a.tiles_n + a.tiles_n2) : a.tiles_m * a.tiles_n * a.splits;
    if (a.n_full > 0) {         // LOANS_TILE_FINETAIL: whole tiles, then K-slices of the rest (set up by igemm_impl)
        if (a.w2 || a.splits != 1 || a.n_full >= a.tiles_m * a.tiles_n || a.tail_splits < 2) return LOANS_EINVAL;
        a.chunks_per_split = (a.nchunks + a.tail_splits - 1) / a.tail_splits;
        a.tail_splits = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
        nblk = a.n_full + (a.tiles_m * a.tiles_n - a.n_full) * a.tail_splits;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

// recognise a row-major ny x nx tap grid with unit-step dy / dx (every conv of this path); nx = 0 otherwise
void detect_tap_grid(const loans_igemm_desc* d, IgemmArgs& a) {
    a.ap.nx = 0; a.ap.ny = 0; a.ap.dy0 = a.ap.dx0 = 0; a.ap.sdy = a.ap.sdx = 1; a.ap.rowpat = 0;
    int nx = 1;
    while (nx < d->ntaps && d->dy[nx] == d->dy[0]) ++nx;
    if (d->ntaps % nx) return;
    const int ny = d->ntaps / nx;
    const int sdx = nx > 1 ? d->dx[1] - d->dx[0] : 1;
    const int sdy = ny > 1 ? d->dy[nx] - d->dy[0] : 1;
    if ((sdx != 1 && sdx != -1) || (sdy != 1 && sdy != -1)) return;
    for (int t = 0; t < d->ntaps; ++t)
        if (d->dy[t] != d->dy[0] + (t / nx) * sdy || d->dx[t] != d->dx[0] + (t % nx) * sdx) return;
    a.ap.nx = nx; a.ap.ny = ny; a.ap.dy0 = d->dy[0]; a.ap.sdy = sdy; a.ap.dx0 = d->dx[0]; a.ap.sdx = sdx;
    for (int r = 0; r < ny; ++r) a.ap.rowpat |= 1ull << (r * nx);
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(IgemmArgs& a, hipStream_t st) {
    const bool relu = a.d.flags & LOANS_F_RELU_IN;
    if (a.bf16) return relu ? launch_igemm_r<BM, BN, WM, WN, true, true, false>(a, st) : launch_igemm_r<BM, BN, WM, WN, false, true, false>(a, st);
    if (a.dma) return relu ? launch_igemm_r<BM, BN, WM, WN, true, false, true>(a, st) : launch_igemm_r<BM, BN, WM, WN, false, false, true>(a, st);
    return relu ? launch_igemm_r<BM, BN, WM, WN, true, false, false>(a, st) : launch_igemm_r<BM, BN, WM, WN, false, false, false>(a, st);
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
    if ((int64_t)d->B * d->inH * d->inW * ((d->flags & LOANS_F_DENSE) ? 1 : d->Cin) >= lim) return LOANS_ERANGE;
    if ((int64_t)d->B * d->outH * d->outW * d->Cout >= lim) return LOANS_ERANGE;
    if ((int64_t)d->B * d->gridH * d->gridW >= lim) return LOANS_ERANGE;
    if ((int64_t)d->ntaps * d->Cin * d->Cout >= lim) return LOANS_ERANGE;
    if (d->flags & LOANS_F_DENSE) {
        // no bounds masks in this mode: every K-row of every grid pixel has to lie inside its input row
        for (int t = 0; t < d->ntaps; ++t) {
            if (d->dy[t] < 0 || d->dx[t] < 0) return LOANS_EINVAL;
            if ((d->gridH - 1) * d->isy + d->dy[t] >= d->inH) return LOANS_EINVAL;
            if ((d->gridW - 1) * d->isx + d->dx[t] + d->Cin > d->inW) return LOANS_EINVAL;
        }
    }
    return LOANS_OK;
}

}  // namespace

struct IgemmPair {       // second convolution of a pair launch
    const float* w;
    float* out;
    double* stats;
    int Cout;
};

struct IgemmClasses {    // class launch: descs[0] is `d`, the weights of class c at w[c]
    int n;
    const loans_igemm_desc* descs;
    const float* const* w;
};

static int igemm_impl(const float* in, const float* w, float* out, const float* bias, double* stats,
                      const float* ref, const float* addend, const loans_igemm_desc* d, void* stream, int bf16,
                      const IgemmPair* pair = nullptr, const IgemmClasses* mc = nullptr) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!in || !w || !out || (d->Cout & 3)) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_BIAS) && !bias) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_STATS) && !stats) return LOANS_EINVAL;
    if ((d->flags & (LOANS_F_MASK | LOANS_F_ADDEND_MASK)) && !ref) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_ADDEND_MASK) && !(d->flags & LOANS_F_ADDEND)) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_ADDEND) && !addend) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_OUT_BF16) && (d->flags & (LOANS_F_MASK | LOANS_F_ADDEND | LOANS_F_ADDEND_MASK))) return LOANS_EINVAL;
    if (d->flags & LOANS_F_BNSUMS) {        // a data gradient's epilogue takes the sums of the BN below it: nothing else rides along
        if (!ref || !bias || !stats || pair || mc) return LOANS_EINVAL;
        if (d->flags & (LOANS_F_BIAS | LOANS_F_STATS | LOANS_F_MASK | LOANS_F_ADDEND | LOANS_F_ADDEND_MASK | LOANS_F_DENSE | LOANS_F_OUT_BF16))
            return LOANS_EINVAL;
        if ((d->tile & 0xEF) == LOANS_TILE_FINETAIL) return LOANS_EINVAL;
    }
    IgemmArgs a;
    a.in = in; a.w = w; a.out = out; a.bias = bias; a.stats = stats; a.ref = ref; a.addend = addend;
    a.d = *d;
    a.bf16 = bf16;
    a.w2 = nullptr; a.out2 = nullptr; a.stats2 = nullptr; a.Cout2 = 0; a.tiles_n2 = 0; a.w2_bytes = a.out2_bytes = 0;
    a.dbg = 0;
#ifdef LOANS_EXPERIMENT
    if (const char* e = getenv("LOANS_DBG")) a.dbg = atoi(e);
#endif
    a.M = d->B * d->gridH * d->gridW;
    a.m_begin = 0;
    a.n_full = 0; a.tail_splits = 1;
    a.Ktot = d->ntaps * d->Cin;
    a.nchunks = (a.Ktot + BK - 1) / BK;
    a.tail_groups = (a.Ktot - (a.nchunks - 1) * BK + 7) / 8;
    {
        const int64_t ib = (int64_t)d->B * d->inH * d->inW * ((d->flags & LOANS_F_DENSE) ? 1 : d->Cin) * 4;
        const int64_t wb = (int64_t)d->Cout * a.Ktot * 4;
        const int64_t ob = (int64_t)d->B * d->outH * d->outW * d->Cout * ((d->flags & LOANS_F_OUT_BF16) ? 2 : 4);
        if (ib >= 0xFFFFFFF0ll || wb >= 0xFFFFFFF0ll || ob >= 0xFFFFFFF0ll) return LOANS_ERANGE;   // 32-bit buffer offsets
        a.in_bytes = (unsigned)ib;
        a.w_bytes = (unsigned)wb;
        a.out_bytes = (unsigned)ob;
        a.nt_out = loans_conv_nt((size_t)ob);
    }
    detect_tap_grid(d, a);
    if (pair) {
        const int64_t wb2 = (int64_t)pair->Cout * a.Ktot * 4;
        const int64_t ob2 = (int64_t)d->B * d->outH * d->outW * pair->Cout * 4;
        if (wb2 >= 0xFFFFFFF0ll || ob2 >= 0xFFFFFFF0ll) return LOANS_ERANGE;
        a.w2 = pair->w; a.out2 = pair->out; a.stats2 = pair->stats; a.Cout2 = pair->Cout;
        a.w2_bytes = (unsigned)wb2; a.out2_bytes = (unsigned)ob2;
    }
    a.ncls = 1;
    if (mc) {
        // the classes differ in their grid, their output phase and their taps; image, strides, channels and flags are shared
        if (pair || bf16 || mc->n < 2 || mc->n > LOANS_MAX_CLASSES) return LOANS_EINVAL;
        if (d->flags & (LOANS_F_DENSE | LOANS_F_STATS | LOANS_F_BIAS)) return LOANS_EINVAL;
        a.ncls = mc->n;
        for (int c = 0; c < mc->n; ++c) {
            const loans_igemm_desc* e = mc->descs + c;
            if ((rc = check_desc(e))) return rc;
            if (!mc->w[c] || e->ntaps > LOANS_MAX_CLS_TAPS) return LOANS_EINVAL;
            if (e->B != d->B || e->inH != d->inH || e->inW != d->inW || e->Cin != d->Cin || e->outH != d->outH ||
                e->outW != d->outW || e->Cout != d->Cout || e->osy != d->osy || e->osx != d->osx || e->isy != d->isy ||
                e->isx != d->isx || e->flags != d->flags)
                return LOANS_EINVAL;
            IgemmArgs t;                    // tap grid of this class
            detect_tap_grid(e, t);
            IgemmArgs::Cls& k = a.cls[c];
            k.w = mc->w[c];
            k.gridH = e->gridH; k.gridW = e->gridW; k.oy0 = e->oy0; k.ox0 = e->ox0; k.ntaps = e->ntaps;
            k.M = e->B * e->gridH * e->gridW;
            k.Ktot = e->ntaps * e->Cin;
            k.w_bytes = (unsigned)((int64_t)e->Cout * k.Ktot * 4);
            k.nchunks = (k.Ktot + BK - 1) / BK;
            k.tail_groups = (k.Ktot - (k.nchunks - 1) * BK + 7) / 8;
            k.tiles_m = 0; k.blk0 = 0; k.per_xcd = 0;      // per tile shape: launch_igemm_r
            k.nx = t.ap.nx; k.ny = t.ap.ny; k.dy0 = t.ap.dy0; k.sdy = t.ap.sdy; k.dx0 = t.ap.dx0; k.sdx = t.ap.sdx;
            k.rowpat = t.ap.rowpat;
            for (int i = 0; i < LOANS_MAX_CLS_TAPS; ++i) { k.dy[i] = e->dy[i < e->ntaps ? i : 0]; k.dx[i] = e->dx[i < e->ntaps ? i : 0]; }
        }
    }
    hipStream_t st = as_stream(stream);
    int tile = d->tile;
    if (mc) {
        const int t = tile & ~LOANS_TILE_DMA;
        if (t != LOANS_TILE_128x128 && t != LOANS_TILE_128x64 && t != LOANS_TILE_64x64 && t != LOANS_TILE_256x64) return LOANS_EINVAL;
    }
    if (pair && ((tile >> 8) || (tile & 0xFF) == LOANS_TILE_SPLIT)) return LOANS_EINVAL;
    a.splits = (tile >> 8) & 0xFF;          // LOANS_TILE_SPLITK(s)
    if (a.splits < 1) a.splits = 1;
    tile &= 0xFF;
    if (a.splits > 1 && (bf16 || (d->flags & ~(LOANS_F_DENSE | LOANS_F_RELU_IN)) || tile == LOANS_TILE_SPLIT))
        return LOANS_EINVAL;                // raw partial sums only: the epilogue flags belong to loans_igemm_finalize_f32
    a.dma = (tile & LOANS_TILE_DMA) ? 1 : 0;
    if (a.dma && bf16) return LOANS_EINVAL;
    tile &= ~LOANS_TILE_DMA;
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
    if (tile == LOANS_TILE_SPLIT) {
        // 128x128 tiles for as many rows as fill the machine in whole rounds (2 blocks per CU), 64x64 tiles for the
        // remaining rows: the big tile's better MFMA rate without its last, mostly empty round
        const int slots = 2 * loans_device_cus();           // of the current device
        if (slots <= 0) return LOANS_EINVAL;
        const int tiles_n = (d->Cout + 127) / 128;
        const int64_t full = ((int64_t)(a.M / 128) * tiles_n / slots) * slots;      // big tiles in whole rounds
        const int rows_big = (int)(full / tiles_n) * 128;
        if (rows_big > 0) {
            IgemmArgs b = a;
            b.M = rows_big;
            rc = launch_igemm<128, 128, 2, 2>(b, st);
            if (rc) return rc;
        }
        if (rows_big == a.M) return LOANS_OK;
        a.m_begin = rows_big;
        return launch_igemm<64, 64, 2, 2>(a, st);
    }
    if (tile == LOANS_TILE_STEM) {          // the dense RGB stem as a direct convolution (stem.hip)
        if (pair || a.splits > 1 || a.dma || mc) return LOANS_EINVAL;
        if (bf16) return loans_stem7_bf16_launch(in, w, out, bias, stats, d, st);       // needs LOANS_F_OUT_BF16 (checked there)
        return loans_stem7_launch(in, w, out, bias, stats, d, st);
    }
    if (tile == LOANS_TILE_FINETAIL) {
        // 64x64 tiles; as many whole tiles as share out evenly over the CUs run at full K, the remaining ones (< one per CU)
        // are cut into K-slices behind them in the SAME launch, so that every CU ends with a small unit instead of some CUs
        // with a whole extra tile (res5 at B = 256: 1568 tiles on 256 CUs = 6.125 each -> 6 + 32 tiles x 8 slices).  The
        // sliced tiles' rows are zeroed here, receive raw partial sums and get bias / statistics from the finalize pass.
        if (pair || bf16 || a.splits > 1) return LOANS_EINVAL;
        if (d->flags & (LOANS_F_MASK | LOANS_F_ADDEND | LOANS_F_ADDEND_MASK | LOANS_F_OUT_BF16)) return LOANS_EINVAL;
        if (d->osy != 1 || d->osx != 1 || d->oy0 || d->ox0 || d->outH != d->gridH || d->outW != d->gridW) return LOANS_EINVAL;
        const int c4 = d->Cout / 4;
        if ((d->Cout & 3) || !(c4 <= 256 ? (256 % c4 == 0) : (c4 % 256 == 0))) return LOANS_EINVAL;   // finalize's thread map
        const int cus = loans_device_cus();                 // of the current device
        if (cus <= 0) return LOANS_EINVAL;
        const int tiles_n = (d->Cout + 63) / 64, tiles_m = (a.M + 63) / 64;
        const int ntile = tiles_m * tiles_n;
        int n_full = ntile / cus * cus;
        n_full -= n_full % tiles_n;                         // the sliced tiles are whole tile rows: contiguous output rows
        const int n_tail = ntile - n_full;
        int sl = n_tail > 0 ? cus / n_tail : 0;
        if (sl > 16) sl = 16;
        while (sl > 1 && a.nchunks / sl < 4) --sl;
        if (n_full <= 0 || n_tail <= 0 || sl < 2) return launch_igemm<64, 64, 2, 2>(a, st);      // nothing to even out
        const int rows_head = n_full / tiles_n * 64;
        const int tail_rows = a.M - rows_head;
        float* tail_out = out + (int64_t)rows_head * d->Cout;
        if (hipMemsetAsync(tail_out, 0, (size_t)tail_rows * d->Cout * sizeof(float), st) != hipSuccess) return LOANS_EINVAL;
        a.n_full = n_full;
        a.tail_splits = sl;
        rc = launch_igemm<64, 64, 2, 2>(a, st);
        if (rc) return rc;
        const int fin = d->flags & (LOANS_F_BIAS | LOANS_F_STATS);
        return fin ? loans_igemm_finalize_f32(tail_out, bias, stats, nullptr, nullptr, fin, tail_rows, d->Cout, stream) : LOANS_OK;
    }
    switch (tile) {
        case LOANS_TILE_128x128: return launch_igemm<128, 128, 2, 2>(a, st);
        case LOANS_TILE_128x64: return launch_igemm<128, 64, 2, 2>(a, st);
        case LOANS_TILE_64x64: return launch_igemm<64, 64, 2, 2>(a, st);
        case LOANS_TILE_256x64: return launch_igemm<256, 64, 4, 1>(a, st);
        default: return LOANS_EINVAL;
    }
}

extern "C" int loans_igemm_f32(const float* in, const float* w, float* out, const float* bias, double* stats,
                               const float* ref, const float* addend, const loans_igemm_desc* d, void* stream) {
    return igemm_impl(in, w, out, bias, stats, ref, addend, d, stream, 0);
}

extern "C" int loans_igemm_classes_f32(const float* in, const float* const* w, float* out, const float* ref, const float* addend,
                                       const loans_igemm_desc* descs, int32_t n, void* stream) {
    if (!descs || !w || n < 1) return LOANS_EINVAL;
    if (n == 1) return igemm_impl(in, w[0], out, nullptr, nullptr, ref, addend, descs, stream, 0);
    const IgemmClasses mc = {n, descs, w};
    return igemm_impl(in, w[0], out, nullptr, nullptr, ref, addend, descs, stream, 0, nullptr, &mc);
}

extern "C" int loans_igemm_pair_f32(const float* in, const float* w_a, float* out_a, double* stats_a, const float* w_b,
                                    float* out_b, double* stats_b, int32_t Cout_b, const loans_igemm_desc* d, void* stream) {
    if (!d || !w_b || !out_b || Cout_b <= 0 || (Cout_b & 3)) return LOANS_EINVAL;
    if (d->flags & ~(LOANS_F_STATS | LOANS_F_RELU_IN)) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_STATS) && !stats_b) return LOANS_EINVAL;
    const IgemmPair p = {w_b, out_b, stats_b, Cout_b};
    return igemm_impl(in, w_a, out_a, nullptr, stats_a, nullptr, nullptr, d, stream, 0, &p);
}

extern "C" int loans_igemm_bf16_f32(const float* in, const float* w, float* out, const float* bias, double* stats,
                                    const float* ref, const float* addend, const loans_igemm_desc* d, void* stream) {
    return igemm_impl(in, w, out, bias, stats, ref, addend, d, stream, 1);
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
    int bf16;
    unsigned x_bytes, gy_bytes;
};

// BF16 = true: fragments are packed to bf16 after the (fp32, conflict-free) LDS reads and contracted on the
// 32x32x16 bf16 MFMA; staging and accumulation stay fp32.
// GY16 = true (LOANS_F_GY_BF16, the stem of the bf16-storage arm): gy is a bf16 tensor; a thread fetches its four
// channels as 8 bytes and widens them while staging, the LDS image and everything behind it are unchanged.
template <int BCO, int BJ, bool RELU, bool BF16, bool GY16 = false>   // BCO (output channels) x BJ (tap-channel columns) block tile, 4 waves as 2 x 2
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    static_assert(BCO <= BJ, "the loader's thread map follows the wider (X) tile");
    constexpr int TM = BCO / 2 / 32, TN = BJ / 2 / 32;   // MFMA tiles per wave
    constexpr int UPR = BJ / 4;             // float4 units per X row (thread map); Y rows use the first BCO/4
    constexpr int RPP = 256 / UPR;          // rows per loader pass
    constexpr int NP = 32 / RPP;            // passes (rows per thread) per 32-row chunk
    constexpr int NMMA = TM * TN;           // MFMAs per k step (2 reduction rows)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ys = reinterpret_cast<float*>(smem);     // [2][32][BCO]
    float* Xs = Ys + 2 * 32 * BCO;                  // [2][32][BJ]

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
    const int ucin = (d.flags & LOANS_F_DENSE) ? 1 : d.Cin;     // floats per unit of inW / ix (see igemm_kernel)
    const int cpt = d.Cin >> 2;
    const int ug = tj * UPR + unit;
    const int xtap = ug / cpt;
    const int xc4 = ug - xtap * cpt;
    const bool xtv = xtap < d.ntaps;
    const int dy = xtv ? (int)d.dy[xtap] : 0;
    const int dx = xtv ? (int)d.dx[xtap] : 0;
    const int yco = tco * BCO + unit * 4;
    const bool ythread = unit < BCO / 4;                 // this thread also stages a piece of the Y tile
    const bool yv = ythread && yco < d.Cout;             // Cout is a multiple of 4 for every layer on this path

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy), 0, (int)a.gy_bytes, 0x00020000);

    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    const int total_chunks = (a.M + 31) / 32;
    if (c_end > total_chunks) c_end = total_chunks;

    // (b, y, x) of this thread's NP rows, advanced by 32 pixels per chunk without divisions
    int pb[NP], py[NP], px[NP];
    {
        const int gHW = d.gridH * d.gridW;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int m = c_begin * 32 + prow + RPP * p;
            pb[p] = m / gHW;
            const int rem = m - pb[p] * gHW;
            py[p] = rem / d.gridW;
            px[p] = rem - py[p] * d.gridW;
        }
    }
    const float inv_gw = 1.f / (float)d.gridW, inv_gh = 1.f / (float)d.gridH;

    f32x4 ry[NP], rx[NP];
    // one loader piece = one chunk row of this thread: two bounds-checked loads + the coordinate advance
    auto load_row = [&](int p) {
        const int b = pb[p], y = py[p], x = px[p];
        const bool rv = b < d.B;
        const int pix = (b * d.outH + y * d.osy + d.oy0) * d.outW + x * d.osx + d.ox0;
        const unsigned goff = ((unsigned)(pix * d.Cout + yco) * (GY16 ? 2u : 4u)) | ((unsigned)(rv & yv) - 1u);
        if constexpr (GY16) {
            ry[p] = __builtin_convertvector(__builtin_bit_cast(bf16x4_t, __builtin_amdgcn_raw_buffer_load_b64(rs_g, (int)goff, 0, 0)), f32x4);
        } else {
            ry[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)goff, 0, 0));
        }
        const int iy = y * d.isy + dy, ix = x * d.isx + dx;
        const unsigned ok = (unsigned)(rv & xtv) & (unsigned)((unsigned)iy < (unsigned)d.inH) &
                            (unsigned)((unsigned)ix < (unsigned)d.inW);
        const unsigned xoff = ((unsigned)(((b * d.inH + iy) * d.inW + ix) * ucin + xc4 * 4) * 4u) | (ok - 1u);
        rx[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xoff, 0, 0));
        // advance 32 pixels: exact floor((v + .5) / n) for the small integers involved
        int nx = x + 32;
        const int qx = (int)(((float)nx + 0.5f) * inv_gw);
        nx -= qx * d.gridW;
        int ny = y + qx;
        const int qy = (int)(((float)ny + 0.5f) * inv_gh);
        ny -= qy * d.gridH;
        px[p] = nx; py[p] = ny; pb[p] = b + qy;
    };
    auto store_y = [&](int buf, int p) {
        if (ythread) *reinterpret_cast<f32x4*>(Ys + (buf * 32 + prow + RPP * p) * BCO + unit * 4) = ry[p];
    };
    auto store_x = [&](int buf, int p) {
        if (RELU) {
            rx[p].x = fmaxf(rx[p].x, 0.f); rx[p].y = fmaxf(rx[p].y, 0.f);
            rx[p].z = fmaxf(rx[p].z, 0.f); rx[p].w = fmaxf(rx[p].w, 0.f);
        }
        *reinterpret_cast<f32x4*>(Xs + (buf * 32 + prow + RPP * p) * BJ + unit * 4) = rx[p];
    };

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int fragY = h * BCO + wm * TM * 32 + r;
    const int fragX = h * BJ + wn * TN * 32 + r;
    auto read_k = [&](int buf, int s, float (&af)[TM], float (&bf)[TN]) {
        const float* Yb = Ys + (buf * 32 + 2 * s) * BCO + fragY;
        const float* Xb = Xs + (buf * 32 + 2 * s) * BJ + fragX;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = Yb[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = Xb[j * 32];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto mma_one = [&](int q, const float (&af)[TM], const float (&bf)[TN]) {
        const int i = q / TN, j = q % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    };

    // ---- software-pipelined reduction loop (same scheme as igemm_kernel) -------------------------
    // 16 k steps per chunk; fragments of step s+1 are read while the MFMAs of step s run; loader rows of
    // the next chunk ride in steps 0 .. NP-1, their LDS writes in steps 7 .. 7+2NP-1 (<= 14), and step 15's MFMAs
    // run behind the barrier.
    float fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    if (c_begin < c_end) {
#pragma unroll
        for (int p = 0; p < NP; ++p) load_row(p);
#pragma unroll
        for (int p = 0; p < NP; ++p) { store_y(0, p); store_x(0, p); }
    }
    __syncthreads();
    if constexpr (BF16) {
        // lane (r, h) of a 32x32x16 step needs reduction rows 16s + 8h .. + 7 of its column: eight b32 reads
        // (32 consecutive lanes = 32 consecutive banks), packed to bf16 in registers
        for (int c = c_begin; c < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
            const bool more = (c + 1) < c_end;
            if (more) {
#pragma unroll
                for (int p = 0; p < NP; ++p) load_row(p);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float* Yb = Ys + (buf * 32 + 16 * s + 8 * h) * BCO + wm * TM * 32 + r;
                const float* Xb = Xs + (buf * 32 + 16 * s + 8 * h) * BJ + wn * TN * 32 + r;
                bf16x8_t af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) af[i][e] = (__bf16)Yb[e * BCO + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bf[j][e] = (__bf16)Xb[e * BJ + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            if (more) {
#pragma unroll
                for (int p = 0; p < NP; ++p) { store_y(buf ^ 1, p); store_x(buf ^ 1, p); }
            }
            __syncthreads();
        }
    } else if (c_begin < c_end) {
        read_k(0, 0, fa0, fb0);
        int c = c_begin;
        for (; c + 1 < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s == 15) __syncthreads();
                if (s & 1) {
                    if (s < 15) read_k(buf, s + 1, fa0, fb0); else read_k(buf ^ 1, 0, fa0, fb0);
                } else {
                    read_k(buf, s + 1, fa1, fb1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NMMA; ++q) {
                    if (s & 1) mma_one(q, fa1, fb1); else mma_one(q, fa0, fb0);
                    if (q == 0) {
                        if (s < NP) load_row(s);
                        if (s >= 7 && s < 7 + 2 * NP) {        // all LDS writes land before step 15's barrier
                            const int w = s - 7;
                            if (w & 1) store_x(buf ^ 1, w >> 1); else store_y(buf ^ 1, w >> 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        {   // last chunk of this split
            const int buf = (c - c_begin) & 1;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s < 15) {
                    if (s & 1) read_k(buf, s + 1, fa0, fb0); else read_k(buf, s + 1, fa1, fb1);
                }
#pragma unroll
                for (int q = 0; q < NMMA; ++q) {
                    if (s & 1) mma_one(q, fa1, fb1); else mma_one(q, fa0, fb0);
                }
            }
        }
    }

    const __amdgpu_buffer_rsrc_t rs_dw = __builtin_amdgcn_make_buffer_rsrc(a.dw, 0, (int)((unsigned)d.Cout * (unsigned)a.Ktot * 4u), 0x00020000);
    (void)rs_dw;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int jc = tj * BJ + wn * TN * 32 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = tco * BCO + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < d.Cout && jc < a.Ktot) atomic_add_f32(a.dw + (int64_t)co * a.Ktot + jc, acc[i][j][e]);
            }
        }
}

template <int BCO, int BJ, bool RELU, bool BF16, bool GY16 = false>
int launch_wgrad_r(WgradArgs& a, int splits_req, hipStream_t st) {
    static loans_device_once lds_limit_set;       // per template instance = per kernel, one bit per device
    constexpr size_t lds = (size_t)2 * 32 * (BCO + BJ) * 4;
    auto kern = wgrad_kernel<BCO, BJ, RELU, BF16, GY16>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), lds)) return rc_;
    a.tiles_co = (a.d.Cout + BCO - 1) / BCO;
    a.tiles_j = (a.Ktot + BJ - 1) / BJ;
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

template <int BCO, int BJ>
int launch_wgrad(WgradArgs& a, int splits_req, hipStream_t st) {
    const bool relu = a.d.flags & LOANS_F_RELU_IN;
    if (a.d.flags & LOANS_F_GY_BF16) return (relu || !a.bf16) ? LOANS_EINVAL : launch_wgrad_r<BCO, BJ, false, true, true>(a, splits_req, st);
    if (a.bf16) return relu ? launch_wgrad_r<BCO, BJ, true, true>(a, splits_req, st) : launch_wgrad_r<BCO, BJ, false, true>(a, splits_req, st);
    return relu ? launch_wgrad_r<BCO, BJ, true, false>(a, splits_req, st) : launch_wgrad_r<BCO, BJ, false, false>(a, splits_req, st);
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

static int wgrad_impl(const float* x, const float* gy, float* dw, const loans_igemm_desc* d, int32_t splits,
                      void* stream, int bf16) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !gy || !dw || (d->Cout & 3)) return LOANS_EINVAL;
    WgradArgs a;
    a.x = x; a.gy = gy; a.dw = dw; a.d = *d;
    a.bf16 = bf16;
    a.M = d->B * d->gridH * d->gridW;
    a.Ktot = d->ntaps * d->Cin;
    {
        const int64_t xb = (int64_t)d->B * d->inH * d->inW * ((d->flags & LOANS_F_DENSE) ? 1 : d->Cin) * 4;
        const int64_t gb = (int64_t)d->B * d->outH * d->outW * d->Cout * ((d->flags & LOANS_F_GY_BF16) ? 2 : 4);
        if (xb >= 0xFFFFFFF0ll || gb >= 0xFFFFFFF0ll) return LOANS_ERANGE;
        a.x_bytes = (unsigned)xb;
        a.gy_bytes = (unsigned)gb;
    }
    hipStream_t st = as_stream(stream);
    const bool small = (d->Cout <= 64) || (a.Ktot <= 64);
    int tile = d->tile;
    if (tile == 0) tile = small ? LOANS_TILE_64x64 : LOANS_TILE_128x128;
    if (tile == LOANS_TILE_STEM) return bf16 ? LOANS_EINVAL : loans_stem7_wgrad_launch(x, gy, dw, d, st);     // direct (stem.hip)
    if (tile == LOANS_TILE_64x64) return launch_wgrad<64, 64>(a, splits, st);
    if (tile == LOANS_TILE_128x128) return launch_wgrad<128, 128>(a, splits, st);
    if (tile == LOANS_TILE_64x128) return launch_wgrad<64, 128>(a, splits, st);
    return LOANS_EINVAL;
}

extern "C" int loans_wgrad_f32(const float* x, const float* gy, float* dw, const loans_igemm_desc* d,
                               int32_t splits, void* stream) {
    return wgrad_impl(x, gy, dw, d, splits, stream, 0);
}

extern "C" int loans_wgrad_bf16_f32(const float* x, const float* gy, float* dw, const loans_igemm_desc* d,
                                    int32_t splits, void* stream) {
    return wgrad_impl(x, gy, dw, d, splits, stream, 1);
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

#ifdef LOANS_STAMPS
extern "C" int loans_debug_read_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#endif
