// Implicit-GEMM convolution on bf16 STORAGE (BASELINE configs 3 / 5): activations, gradients and weights are bf16
// in HBM, the contraction runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, BN statistics come from the
// fp32 accumulators, the result is rounded to bf16 (RNE) once, in the epilogue.
//
//   out[m][n] = sum_k A[m][k] * Wp[n][k],   k = (tap, channel),   NHWC with C % 8 == 0
//
// A K chunk is 64 elements = eight 16-byte units (8 channels each) = one 128-byte LDS row per tile row.  The
// operand tiles are staged by LDS-DMA (`buffer_load_dwordx4 ... lds`: 64 lanes x 16 B written contiguously, no
// staging registers, no ds_write pass) into unpadded rows; bank conflicts are avoided by an XOR swizzle on the
// SOURCE side (lane (row, slot) fetches unit slot ^ key(row), key = (row >> 1) & 7) that the fragment reads undo:
// lane (r, h) of MFMA step s reads unit (2s + h) ^ key(r) as ONE ds_read_b128 = its eight k values.
// Same problem descriptor, tap masks, XCD-aware tile order and epilogue flags as igemm.hip; forward convolution and
// data gradient are the same kernel (dgrad: one launch per stride-parity class on re-packed weights).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#ifdef LOANS_STAMPS
// Diagnostic build only (tools/stamp_run16.py): per-wave cycle sums of the K-loop phases of the first 64 blocks.
__device__ unsigned long long g_stamps16[64 * 4 * 8];
__device__ unsigned long long g_stamps16b[64 * 4 * 4];     // per wave: kernel entry, K loop begin, K loop end, kernel exit
__device__ unsigned long long g_stamps16c[64 * 4 * 4];     // per wave: last chunk done, statistics done, staging pass 0 stored, pass 1 stored
#define STAMP16(t)                                                                       \
    do {                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
#else
#define STAMP16(t) do { } while (0)
#endif

namespace {

constexpr int BKH = 64;                 // K elements per chunk (8 units of 16 bytes)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Igemm16Args {
    const __bf16* in;
    const __bf16* w;
    __bf16* out;
    const float* bias;
    double* stats;
    const __bf16* ref;
    const __bf16* addend;
    loans_igemm_desc d;
    int M, Ktot, nchunks, tiles_m, tiles_n;
    int nt_out;         // output stores non-temporal (loans_conv_nt)
    int dbg;            // experiment bits (LOANS_EXPERIMENT builds only)
    unsigned in_bytes, w_bytes, out_bytes;
    struct {            // nx > 0: taps are an ny x nx grid, dy = dy0 + row*sdy, dx = dx0 + col*sdx, sd* = +-1
        int nx, ny, dy0, sdy, dx0, sdx;
        unsigned long long rowpat;
    } ap;
    // split-K (loans_igemm_bf16s_splitk): block (tile, s) contracts chunks [s * cps, (s + 1) * cps) and ADDS its raw fp32 tile
    // to `partial` [pixels][Cout] (zeroed by the caller); loans_igemm_finalize_bf16 makes the bf16 tensor of the finished sums
    int splits, chunks_per_split;
    float* partial;
    // pair launch (loans_igemm_pair_bf16s): the GEMM has 2 x out_c columns -- two convolutions of the same input, weights
    // stacked along N -- and columns >= csplit belong to the second output tensor (right behind the first one) and to stats2:
    // the input tile is staged ONCE for both.  csplit = 0, out_c = d.Cout otherwise.
    int csplit, out_c;
    unsigned tensor_bytes;
    double* stats2;
};

__device__ __forceinline__ int xcd_remap16(int id, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// relu on packed bf16: as signed 16-bit integers the negative floats (and -0) are negative, so max(v, 0) is relu
typedef short s16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t relu_bf16x8(bf16x8_t v) {
    const s16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8_t, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, v), z));
}

__device__ __forceinline__ f32x4 cvt_lo(bf16x8_t v) {
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ f32x4 cvt_hi(bf16x8_t v) {
    return f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
}

// LDS stages of the operand ring (ST): with two, the DMA of chunk c + 1 rides behind the MFMAs of chunk c -- enough where two
// or more blocks share a CU and cover each other's waits.  A block that owns its CU gets a longer ring: chunk c + ST - 1 is
// issued while chunk c is contracted and the chunk barrier waits with a COUNTED vmcnt for chunk c + 1 only (loads retire in
// order: the ST - 2 younger chunks stay in flight).  Three stages for the 512-thread 256 x 128 tile; the LOANS_TILE_DEEP forms
// of the small tiles (few blocks, long K: res6 / res7 at 512 px, where one block per CU waited a full memory latency per
// chunk) take as many stages as fit half the LDS.
template <int BM, int BN, int WM, int WN>
constexpr int igemm16_stages(bool deep) {
    if (deep) return BM * BN <= 64 * 64 ? 8 : (BM * BN <= 128 * 64 ? 5 : 4);
    return (WM * WN == 8 && (size_t)3 * (BM + BN) * BKH * 2 <= 150 * 1024) ? 3 : 2;
}
template <int BM, int BN>
constexpr int igemm16_epilogue_passes() { return (size_t)BM * (BN + 4) * 4 > 140 * 1024 ? 2 : 1; }
template <int BM, int BN, int ST>
constexpr size_t igemm16_aux_bytes() {      // offset of the tap table / row table behind the tiles
    constexpr size_t stage = (size_t)ST * (BM + BN) * BKH * 2;
    constexpr size_t cs = (size_t)BM / igemm16_epilogue_passes<BM, BN>() * (BN + 4) * 4;      // fp32 epilogue staging tile
    return stage > cs ? stage : cs;
}
template <int BM, int BN, int ST>
constexpr size_t igemm16_lds_bytes() { return igemm16_aux_bytes<BM, BN, ST>() + LOANS_MAX_TAPS * 4 + BM * 4; }

// RELU: gather relu(in) (LOANS_F_RELU_IN, the assessor's pre-activation convs): applied to the A fragments
template <int BM, int BN, int WM, int WN, int ST, bool RELU>
__global__ __launch_bounds__(64 * WM * WN) void igemm16_kernel(const Igemm16Args a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NT = 64 * WM * WN;           // 256 threads, or 512 for the 256-row tiles (two waves per SIMD from ONE block)
    constexpr int RPP = NT / 8;                // tile rows staged per pass: a thread owns one 16-byte K unit of one row
    constexpr int RA = BM / RPP, RB = BN / RPP;
    constexpr int NMMA = TM * TN;              // MFMAs per 16-deep k step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);      // [ST][BM][64]
    __bf16* Bs = As + ST * BM * BKH;                   // [ST][BN][64]
    int* taps = reinterpret_cast<int*>(smem + igemm16_aux_bytes<BM, BN, ST>());
    unsigned* opix = reinterpret_cast<unsigned*>(taps + LOANS_MAX_TAPS);   // [BM] output row byte offset, ~0u = no row

    const loans_igemm_desc& d = a.d;
    const int tid = threadIdx.x;
#ifdef LOANS_STAMPS
    unsigned long long t_entry = 0;
    STAMP16(t_entry);
#endif
    const int logical = xcd_remap16(blockIdx.x, gridDim.x);
    const int ntile = a.tiles_m * a.tiles_n;
    const int split = logical / ntile;                  // 0 unless split-K
    const int ltile = logical - split * ntile;
    const int tn = ltile % a.tiles_n;
    const int tm = ltile / a.tiles_n;
    const int c_begin = split * a.chunks_per_split;     // this block's K chunks
    const int c_end = min(c_begin + a.chunks_per_split, a.nchunks);
    const int nch = c_end - c_begin;
    const int lrow = tid >> 3;
    const int lu = (tid & 7) ^ ((tid >> 4) & 7);       // K unit this thread stages: slot ^ key(row)
    // LOANS_F_DENSE (the RGB stem, see igemm.hip): inW / isx / dx count ELEMENTS of packed 3-channel rows inside a zero
    // border, a "tap" is a run of Cin consecutive elements of one input row; no bounds masks; K units are 4-byte aligned
    const bool dense = d.flags & LOANS_F_DENSE;
    const int pbytes = dense ? 2 : d.Cin * 2;           // bytes per unit of inW / ix (a gathered pixel)
    if (tid < LOANS_MAX_TAPS) {
        const int t = tid < d.ntaps ? tid : 0;
        taps[tid] = (int(d.dy[t]) * d.inW + int(d.dx[t])) * pbytes;
    }

    // per row: byte offset of its base pixel and the bitmask of taps that must read zero (see igemm.hip)
    // (32 bits here: the launcher admits at most 32 taps without LOANS_F_DENSE, and a dense launch has no bad tap --
    // one v_bfe_i32 per piece instead of a 64-bit shift + extract)
    unsigned rowoff[RA];
    unsigned badmask[RA];
    {
        const int gHW = d.gridH * d.gridW;
        const float inv_gw = 1.f / (float)d.gridW, inv_gh = 1.f / (float)d.gridH;
        const int m0 = tm * BM + lrow;
        int b = m0 / gHW;
        int rem = m0 - b * gHW;
        int y = rem / d.gridW;
        int x = rem - y * d.gridW;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0 + RPP * i;
            unsigned pixoff = 0xFFFFFFFFu;
            unsigned long long mask = 0;
            rowoff[i] = 0;
            if (m < a.M) {
                const int iy0 = y * d.isy, ix0 = x * d.isx;
                rowoff[i] = (unsigned)((b * d.inH + iy0) * d.inW + ix0) * (unsigned)pbytes;
                pixoff = (unsigned)((b * d.outH + y * d.osy + d.oy0) * d.outW + x * d.osx + d.ox0) * (unsigned)a.out_c * 2u;
                if (dense) {
                    mask = ~0ull;
                } else if (a.ap.nx > 0) {
                    const int cx = ix0 + a.ap.dx0, cy = iy0 + a.ap.dy0;
                    int jlo, jhi, rlo, rhi;
                    if (a.ap.sdx > 0) { jlo = max(0, -cx); jhi = min(a.ap.nx, d.inW - cx); }
                    else { jlo = max(0, cx - d.inW + 1); jhi = min(a.ap.nx, cx + 1); }
                    if (a.ap.sdy > 0) { rlo = max(0, -cy); rhi = min(a.ap.ny, d.inH - cy); }
                    else { rlo = max(0, cy - d.inH + 1); rhi = min(a.ap.ny, cy + 1); }
                    if (jhi > jlo && rhi > rlo) {
                        const unsigned long long colbits = ((1ull << jhi) - 1ull) & ~((1ull << jlo) - 1ull);
                        const int blo = rlo * a.ap.nx, bhi = rhi * a.ap.nx;
                        const unsigned long long below_hi = bhi >= 64 ? ~0ull : ((1ull << bhi) - 1ull);
                        const unsigned long long rowsel = a.ap.rowpat & below_hi & ~((1ull << blo) - 1ull);
                        mask = colbits * rowsel;
                    }
                } else {
                    for (int t = 0; t < d.ntaps; ++t) {
                        const int iy = iy0 + d.dy[t], ix = ix0 + d.dx[t];
                        if ((unsigned)iy < (unsigned)d.inH && (unsigned)ix < (unsigned)d.inW) mask |= 1ull << t;
                    }
                }
            }
            badmask[i] = ~(unsigned)mask;
#ifdef LOANS_EXPERIMENT
            if (a.dbg & 4) { rowoff[i] = (unsigned)((d.inW + 1) * pbytes) + (rowoff[i] & 0x3FFu); badmask[i] = 0; }   // cache-hot gathers
#endif
            if (lu == 0) opix[lrow + RPP * i] = pixoff;
            x += RPP;
            const int qx = (int)(((float)x + 0.5f) * inv_gw);
            x -= qx * d.gridW;
            y += qx;
            const int qy = (int)(((float)y + 0.5f) * inv_gh);
            y -= qy * d.gridH;
            b += qy;
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.w), 0, (int)a.w_bytes, 0x00020000);

    const int cpt = d.Cin >> 3;   // 16-byte units per tap
    const int q8 = 8 / cpt, r8 = 8 - q8 * cpt;
    const int kunits = a.Ktot >> 3;
    int u = lu + 8 * c_begin;     // this thread's K unit in the chunk being loaded
    int tap = u / cpt, c8 = u - tap * cpt;
    unsigned woff[RB], wbad[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int n = tn * BN + lrow + RPP * i;
        wbad[i] = n < d.Cout ? 0u : 0xFFFFFFFFu;
        woff[i] = n < d.Cout ? (unsigned)n * (unsigned)a.Ktot * 2u : 0u;
#ifdef LOANS_EXPERIMENT
        if (a.dbg & 4) woff[i] = (unsigned)(lrow & 7) * (unsigned)a.Ktot * 2u;
#endif
    }
    unsigned toff = (unsigned)taps[min(tap, LOANS_MAX_TAPS - 1)] + (unsigned)c8 * 16u;
    // per-chunk values every piece shares: the tap's bit position, and all-ones once this thread's K unit lies beyond K
    unsigned tcs = (unsigned)min(tap, 31);
    unsigned kb = (unsigned)((kunits - 1 - u) >> 31);
    // the tap-table entry of the chunk AFTER the next one is fetched a whole chunk before it is needed: read at the point
    // of use it sat behind an lgkmcnt(0) in the middle of the MFMA stream (LDS returns in order, and the fragment reads
    // of the next step were already queued in front of it)
    auto step_tap = [&](int& t, int& c) {
        t += q8;
        c += r8;
        const int wrap = c >= cpt;
        c -= wrap ? cpt : 0;
        t += wrap;
    };
    int tap_n = tap, c8_n = c8;
    step_tap(tap_n, c8_n);
    int traw_n = taps[min(tap_n, LOANS_MAX_TAPS - 1)];

    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto dma_a = [&](int buf, int i) {      // one 1 KiB LDS-DMA piece: 8 rows x 128 B of the A tile
        const unsigned bad = (unsigned)__builtin_amdgcn_sbfe((int)badmask[i], tcs, 1u);     // 0 / ~0: this row's bit of the tap
        const unsigned off = (rowoff[i] + toff) | bad | kb;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(As + (buf * BM + RPP * i + 8 * wave_u) * BKH), 16, (int)off, 0, 0, 0);
    };
    auto dma_b = [&](int buf, int i) {
        const unsigned off = (woff[i] + (unsigned)u * 16u) | wbad[i] | kb;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Bs + (buf * BN + RPP * i + 8 * wave_u) * BKH), 16, (int)off, 0, 0, 0);
    };
    auto advance = [&]() {      // to the following chunk (8 units further along K)
        u += 8;
        tap = tap_n;
        c8 = c8_n;
        toff = (unsigned)traw_n + (unsigned)c8 * 16u;
        tcs = (unsigned)min(tap, 31);
        kb = (unsigned)((kunits - 1 - u) >> 31);
        step_tap(tap_n, c8_n);
        traw_n = taps[min(tap_n, LOANS_MAX_TAPS - 1)];
    };
    constexpr int NPIECE = RA + RB + 1;
    auto dma_piece = [&](int buf, int p) {
        if (p < RA) dma_a(buf, p);
        else if (p < RA + RB) dma_b(buf, p - RA);
        else if (p == RA + RB) advance();
    };

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int fkey = (r >> 1) & 7;
    const int fragA = (wm * TM * 32 + r) * BKH + ((h ^ fkey) & 7) * 8;
    const int fragB = (wn * TN * 32 + r) * BKH + ((h ^ fkey) & 7) * 8;
    auto read_frag = [&](int buf, int s, bf16x8_t (&af)[TM], bf16x8_t (&bf)[TN]) {
        const __bf16* Ab = As + buf * BM * BKH + (fragA ^ (s * 16));
        const __bf16* Bb = Bs + buf * BN * BKH + (fragB ^ (s * 16));
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(Ab + i * 32 * BKH);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8_t*>(Bb + j * 32 * BKH);
    };
    auto relu_frag = [&](bf16x8_t (&af)[TM]) {
        if constexpr (RELU) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = relu_bf16x8(af[i]);
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
    // (RELU) a fragment set is rectified once, right before its MFMAs
    auto mma = [&](const bf16x8_t (&af)[TM], const bf16x8_t (&bf)[TN]) {
#pragma unroll
        for (int q = 0; q < NMMA; ++q) mma_one(q, af, bf);
    };

    // ---- K loop: a chunk = four 16-deep steps; the DMA pieces of chunk c+1 ride behind the MFMAs of steps 0 and 1
    // of chunk c into the other LDS stage (free since the barrier that ended chunk c-1), fragments of step s+1 are read
    // while step s computes, __syncthreads() drains the DMA (vmcnt(0)) and step 3's MFMAs run behind the barrier.
    constexpr int PPG = (NPIECE + 2 * NMMA - 1) / (2 * NMMA);      // pieces per MFMA gap
    bf16x8_t fa0[TM], fb0[TN], fa1[TM], fb1[TN];
#pragma unroll
    for (int p = 0; p < NPIECE; ++p) dma_piece(0, p);
    // end of a chunk: the next one has landed and every wave is done reading the stage the coming DMA overwrites.  More than
    // two stages: the pieces of the ST - 2 chunks issued after it may stay in flight (loads retire in order); the LDS reads of
    // this wave must have returned (lgkmcnt) because another wave's DMA may write that stage right behind the barrier.
    auto chunk_barrier = [&]() {
        if constexpr (ST == 2) {
            __syncthreads();
        } else {
            static_assert((ST - 2) * (RA + RB) <= 63, "vmcnt is a 6-bit counter");
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((ST - 2) * (RA + RB)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
#pragma unroll
    for (int k = 1; k < ST - 1; ++k)
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) dma_piece(k, p);
    chunk_barrier();
    read_frag(0, 0, fa0, fb0);
    int c = 0;
    int s_cur = 0, s_nxt = 1, s_pre = ST - 1;     // stages of chunk c, c + 1 and of the chunk being fetched (c + ST - 1)
#ifdef LOANS_STAMPS
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, p_s0 = 0, p_s1 = 0, p_s2 = 0, p_bar = 0, p_s3 = 0, q_begin = 0;
    STAMP16(q_begin);
#endif
    for (; c + 1 < nch; ++c) {
        const int buf = s_cur;
        STAMP16(q0);
        read_frag(buf, 1, fa1, fb1);
        relu_frag(fa0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NMMA; ++q) {
            mma_one(q, fa0, fb0);
#pragma unroll
            for (int p = q * PPG; p < (q + 1) * PPG; ++p) dma_piece(s_pre, p);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP16(q1);
        read_frag(buf, 2, fa0, fb0);
        relu_frag(fa1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NMMA; ++q) {
            mma_one(q, fa1, fb1);
#pragma unroll
            for (int p = (NMMA + q) * PPG; p < (NMMA + q + 1) * PPG; ++p) dma_piece(s_pre, p);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP16(q2);
        read_frag(buf, 3, fa1, fb1);
        relu_frag(fa0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        relu_frag(fa1);
        STAMP16(q3);
        chunk_barrier();
        STAMP16(q4);
        read_frag(s_nxt, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        STAMP16(q5);
#ifdef LOANS_STAMPS
        p_s0 += q1 - q0; p_s1 += q2 - q1; p_s2 += q3 - q2; p_bar += q4 - q3; p_s3 += q5 - q4;
#endif
        s_cur = s_nxt;
        s_nxt = s_nxt + 1 == ST ? 0 : s_nxt + 1;
        s_pre = s_pre + 1 == ST ? 0 : s_pre + 1;
    }
#ifdef LOANS_STAMPS
    if (logical < 64 && (tid & 63) == 0 && tid < 256) {
        unsigned long long* o = g_stamps16 + (logical * 4 + (tid >> 6)) * 8;
        o[0] = p_s0; o[1] = p_s1; o[2] = p_s2; o[3] = p_bar; o[4] = p_s3; o[5] = q5 - q_begin; o[6] = nch - 1; o[7] = 0;
        unsigned long long* ob = g_stamps16b + (logical * 4 + (tid >> 6)) * 4;
        ob[0] = t_entry; ob[1] = q_begin; ob[2] = q5;
    }
#endif
    {   // last chunk: steps that lie wholly beyond Ktot hold zeros on both sides and are skipped
        const int buf = s_cur;
        const int ts = c_end == a.nchunks ? (a.Ktot - (a.nchunks - 1) * BKH + 15) / 16 : 4;      // 1..4
        if (ts > 1) read_frag(buf, 1, fa1, fb1);
        relu_frag(fa0);
        mma(fa0, fb0);
        if (ts > 1) {
            if (ts > 2) read_frag(buf, 2, fa0, fb0);
            relu_frag(fa1);
            mma(fa1, fb1);
            if (ts > 2) {
                if (ts > 3) read_frag(buf, 3, fa1, fb1);
                relu_frag(fa0);
                mma(fa0, fb0);
                if (ts > 3) { relu_frag(fa1); mma(fa1, fb1); }
            }
        }
    }

#ifdef LOANS_STAMPS
    unsigned long long t_e0 = 0, t_e1 = 0, t_e2 = 0;
    STAMP16(t_e0);
    if (logical < 64 && (tid & 63) == 0 && tid < 256) g_stamps16c[(logical * 4 + (tid >> 6)) * 4 + 0] = t_e0;
#endif
    // ---- epilogue: BN statistics from the fp32 accumulators, tile staged through LDS (fp32) so that every lane
    // converts and stores 8 contiguous channels (16 bytes of bf16)
    const bool f_bias = d.flags & LOANS_F_BIAS, f_stats = d.flags & LOANS_F_STATS;
    const bool f_mask = d.flags & LOANS_F_MASK, f_add = d.flags & LOANS_F_ADDEND;
    const bool f_addmask = d.flags & LOANS_F_ADDEND_MASK;
    const bool f_bnsums = d.flags & LOANS_F_BNSUMS;
    constexpr int LDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(smem);          // [BM / passes][LDC]
    __syncthreads();
    if (f_stats) {
        // rows of this lane that exist (the bias term of the sums counts them): all of them unless this is the last, ragged tile
        // (block-uniform test; the row table is only consulted there -- 16 TM LDS reads per lane otherwise)
        int nvalid = TM * 16;
        if ((tm + 1) * BM > a.M) {
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
            const float bv = (f_bias && cok) ? a.bias[col] : 0.f;
            float s = 0.f, q2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s += acc[i][j][e];
                    q2 += acc[i][j][e] * acc[i][j][e];
                }
            q2 = q2 + 2.f * bv * s + cnt * bv * bv;
            s = s + cnt * bv;
            s += __shfl_xor(s, 32, 64);
            q2 += __shfl_xor(q2, 32, 64);
            if (h == 0 && cok) {
                const bool sec = a.csplit && col >= a.csplit;           // the second convolution of a pair launch
                double* st = (sec ? a.stats2 : a.stats) + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * a.out_c;
                const int scol = sec ? col - a.csplit : col;
                atomic_add_f64(st + scol, (double)s);
                atomic_add_f64(st + a.out_c + scol, (double)q2);
            }
        }
    }
#ifdef LOANS_STAMPS
    STAMP16(t_e1);
    if (logical < 64 && (tid & 63) == 0 && tid < 256) g_stamps16c[(logical * 4 + (tid >> 6)) * 4 + 1] = t_e1;
#endif
    constexpr int CPR = BN / 8;                 // 8-channel units per row
    constexpr int RSTEP = NT / CPR;            // rows covered by the block per pass
    const int oc8 = tid % CPR, r0 = tid / CPR;
    const int col0 = tn * BN + oc8 * 8;
    const unsigned cbad = (col0 + 7 < d.Cout) ? 0u : 0xFFFFFFFFu;     // Cout % 8 == 0 (checked)
    // (descriptors, bias: once; unused by a partial launch)
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ref = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.ref ? a.ref : a.out), 0, (int)a.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.addend ? a.addend : a.out), 0, (int)a.out_bytes, 0x00020000);
    const unsigned coff = (a.csplit && col0 >= a.csplit) ? (unsigned)(col0 - a.csplit) * 2u + a.tensor_bytes : (unsigned)col0 * 2u;
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
    // LOANS_F_BNSUMS: this thread's eight channels of the BN's coefficient table (a.bias = [mean | rstd | scale | shift][C]) and
    // its partial sums over the rows it stores
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
    // the tile goes through the fp32 staging area in EP passes of PR rows (one pass for every tile whose [BM][BN + 4] floats
    // fit the operand stages; the 256 x 256 tile takes two): the waves that own the pass's rows write, everybody stores
    constexpr int EP = igemm16_epilogue_passes<BM, BN>();
    constexpr int PR = BM / EP;
    static_assert(PR % (TM * 32) == 0 || (TM * 32) % PR == 0, "a wave's rows fall into whole passes");
    constexpr int NIT = PR / RSTEP;         // rows a thread stores per pass
    constexpr bool EARLY_REF = TM * TN <= 4;    // the 128 x 64 wave tiles (128 live accumulators) take the reference after the staging too
    constexpr int LG = EARLY_REF ? NIT : 1;     // rows per load group behind the staging (those tiles: row by row, as before --
                                                // any group of loads beside their accumulators spills, and the spills cost more)
    static_assert(NIT % LG == 0, "whole load groups");
#pragma unroll 1
    for (int ep = 0; ep < EP; ++ep) {
        // The pass's epilogue operands (ReLU reference, addend, the BN input of LOANS_F_BNSUMS) are requested BEFORE the tile goes
        // through the staging area: inside the store loop every load waited behind the previous row's store (vector memory
        // operations retire in order) and paid its own latency, eight times per pass.
        constexpr int NA = EARLY_REF ? NIT : 1;     // (row by row: one slot, re-used)
        unsigned eoff[NA];
        bf16x8_t e_ref[NA], e_add[NA];
        if (EARLY_REF && !a.partial) {
#pragma unroll
            for (int p = 0; p < NIT; ++p) {
                const unsigned po = opix[ep * PR + r0 + p * RSTEP];
                eoff[p] = (po + coff) | (po == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u) | cbad;
            }
            if (f_mask || f_addmask || f_bnsums) {
#pragma unroll
                for (int p = 0; p < NIT; ++p)
                    e_ref[p] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, (int)eoff[p], 0, 0));
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
        if (a.partial) {
            // raw partial tile: fp32 atomic adds into the zeroed workspace, one wave-instruction = 256 CONTIGUOUS bytes of one row
            // (64 lanes x 4 B: the full-rate shape; lanes 32 B apart run an order of magnitude slower); bias / statistics / mask /
            // addend / the rounding to bf16 belong to loans_igemm_finalize_bf16
            constexpr int WPR = BN / 64;                        // wave-instructions per row
            const int wv = tid >> 6, ln = tid & 63;
#pragma unroll 4
            for (int q = wv; q < PR * WPR; q += NT / 64) {
                const int row = q / WPR, cc = (q - row * WPR) * 64 + ln;
                const unsigned po = opix[ep * PR + row];        // byte offset of the row in a bf16 tensor = 2 * element offset
                const int col = tn * BN + cc;
                if (po != 0xFFFFFFFFu && col < d.Cout) atomic_add_f32(a.partial + (size_t)(po >> 1) + col, Cs[row * LDC + cc]);
            }
            continue;
        }
        // the addends (and, for the 128-accumulator wave tiles, the reference) are requested per group of LG rows, all of a group
        // together, once the pass's accumulators have gone to LDS: beside the early reference and all accumulators they spilled
#pragma unroll
        for (int p = 0; p < NIT; ++p) {
            if (p % LG == 0) {
                if (!EARLY_REF) {
#pragma unroll
                    for (int q = p; q < p + LG; ++q) {
                        const unsigned po = opix[ep * PR + r0 + q * RSTEP];
                        eoff[q % NA] = (po + coff) | (po == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u) | cbad;
                    }
                }
                if (!EARLY_REF && (f_mask || f_addmask || f_bnsums)) {
#pragma unroll
                    for (int q = p; q < p + LG; ++q)
                        e_ref[q % NA] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, (int)eoff[q % NA], 0, 0));
                }
                if (f_add) {
#pragma unroll
                    for (int q = p; q < p + LG; ++q)
                        e_add[q % NA] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)eoff[q % NA], 0, 0));
                }
            }
            const int row = r0 + p * RSTEP;
            const unsigned off = eoff[p % NA];
            f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + row * LDC + oc8 * 8) + b_lo;
            f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + row * LDC + oc8 * 8 + 4) + b_hi;
            if (f_mask || f_addmask) {
                const f32x4 rl = cvt_lo(e_ref[p % NA]), rh = cvt_hi(e_ref[p % NA]);
                if (f_mask) { lo = keep_pos(lo, rl); hi = keep_pos(hi, rh); }
                if (f_add) {
                    f32x4 al = cvt_lo(e_add[p % NA]), ah = cvt_hi(e_add[p % NA]);
                    if (f_addmask) { al = keep_pos(al, rl); ah = keep_pos(ah, rh); }
                    lo += al; hi += ah;
                }
            } else if (f_add) {
                lo += cvt_lo(e_add[p % NA]); hi += cvt_hi(e_add[p % NA]);
            }
            bf16x8_t o;
            const bf16x4_t ol = __builtin_convertvector(lo, bf16x4_t), oh = __builtin_convertvector(hi, bf16x4_t);
            o[0] = ol[0]; o[1] = ol[1]; o[2] = ol[2]; o[3] = ol[3];
            o[4] = oh[0]; o[5] = oh[1]; o[6] = oh[2]; o[7] = oh[3];
            if (f_bnsums) {          // block-uniform; a row that does not exist loaded zeros and its gradient is zeroed below
                // the BN's input tile; the sums take the ROUNDED gradient (what the apply pass will read back)
                const f32x4 y2[2] = {cvt_lo(e_ref[p % NA]), cvt_hi(e_ref[p % NA])};
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
#ifdef LOANS_STAMPS
        STAMP16(t_e2);
        if (logical < 64 && (tid & 63) == 0 && tid < 256 && ep < 2) g_stamps16c[(logical * 4 + (tid >> 6)) * 4 + 2 + ep] = t_e2;
#endif
    }
    if (f_bnsums) {
        // per block: the threads that share a channel unit (same oc8, RSTEP rows apart) are summed through LDS, then one fp64
        // atomic per channel and sum into this block's replica
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
            const int u8 = tid >> 4, j = tid & 15;              // channel unit, (sum, channel of the unit)
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
#ifdef LOANS_STAMPS
    unsigned long long t_exit = 0;
    STAMP16(t_exit);
    if (logical < 64 && (tid & 63) == 0 && tid < 256) g_stamps16b[(logical * 4 + (tid >> 6)) * 4 + 3] = t_exit;
#endif
}

template <int BM, int BN, int WM, int WN, int ST, bool RELU>
int launch_igemm16_r(Igemm16Args& a, hipStream_t st) {
    static loans_device_once lds_limit_set;       // per template instance = per kernel, one bit per device
    constexpr size_t lds = igemm16_lds_bytes<BM, BN, ST>();
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = igemm16_kernel<BM, BN, WM, WN, ST, RELU>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), lds)) return rc_;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.d.Cout + BN - 1) / BN;
    if (a.splits < 1) a.splits = 1;
    if (a.splits > a.nchunks) a.splits = a.nchunks;
    a.chunks_per_split = (a.nchunks + a.splits - 1) / a.splits;
    a.splits = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;       // no empty slice
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n * a.splits), dim3(64 * WM * WN), lds, st, a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <int BM, int BN, int WM, int WN, bool DEEP = false>
int launch_igemm16(Igemm16Args& a, hipStream_t st) {
    constexpr int ST = igemm16_stages<BM, BN, WM, WN>(DEEP);
    return (a.d.flags & LOANS_F_RELU_IN) ? launch_igemm16_r<BM, BN, WM, WN, ST, true>(a, st) : launch_igemm16_r<BM, BN, WM, WN, ST, false>(a, st);
}

#include "igemm16_pp.h"      // LOANS_TILE_256x256PP: the same block tile with a ping-pong K loop

void detect_tap_grid16(const loans_igemm_desc* d, Igemm16Args& a) {
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

// fp32 -> bf16 (RNE), n a multiple of 4; dst[co][tapsel..] re-pack variant below
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* src, __bf16* dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        *reinterpret_cast<bf16x4_t*>(dst + i * 4) = __builtin_convertvector(*reinterpret_cast<const f32x4*>(src + i * 4), bf16x4_t);
}

struct Repack16Args {
    const float* src;
    __bf16* dst;
    int Cout, Cin, src_taps, ntaps;
    int tapsel[LOANS_MAX_TAPS];
};

// dst[ci][t][co] = bf16(src[co][tapsel[t]][ci]) through a 32x33 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void repack_dgrad16_kernel(const Repack16Args a) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int st = a.tapsel[t];
    for (int k = ty; k < 32; k += 8) {
        const int co = co0 + k, ci = ci0 + tx;
        tile[k][tx] = (co < a.Cout && ci < a.Cin) ? a.src[((int64_t)co * a.src_taps + st) * a.Cin + ci] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int ci = ci0 + k, co = co0 + tx;
        if (ci < a.Cin && co < a.Cout) a.dst[((int64_t)ci * a.ntaps + t) * a.Cout + co] = (__bf16)tile[tx][k];
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient on bf16 tensors:  dw[co][t][c] += sum_m gy[opix(m)][co] * x[pix(m,t)][c]   (fp32 atomics into dw)
// GEMM rows = co, columns = (t, c), reduction = pixels m, split over blocks.  Both operands are pixel-major in
// memory (channels contiguous), i.e. TRANSPOSED for the MFMA, whose lane wants 8 consecutive reduction indices of
// one channel: the tiles are staged as they lie ([pixel][channel] rows, 16-byte units) and the fragments are read
// with ds_read_b64_tr_b16, the hardware transpose read (per 16 lanes: a 4-pixel x 16-channel block, column-major
// to registers) -- two of them are one 32x32x16 operand.  Rows are padded by 32 elements: the 4 pixel rows a
// 32-lane half touches then start 16 banks apart (row strides of 192 / 320 / 576 bytes) and the reads are
// conflict-free.
// ------------------------------------------------------------------------------------------
struct Wgrad16Args {
    const __bf16* x;
    const __bf16* gy;
    float* dw;
    loans_igemm_desc d;
    int M, Ktot, tiles_co, tiles_j, splits, chunks_per_split;
    unsigned x_bytes, gy_bytes;
    // partial slabs (loans_wgrad_bf16s_ws): block (tile, split) STORES its raw tile into slab `split` of `ws` ([splits][Cout][Ktot],
    // the layout of dw) instead of adding it to dw with atomics; loans_fold_slabs_f32 sums the slabs in a fixed order
    float* ws;
    int64_t slab;
    // LOANS_F_AFFINE_IN (1 x 1 convolutions): x is the INPUT of the BatchNormalization in front of the convolution, affine = its
    // float[2][Cin] = scale, shift: rows go to LDS as relu(x * scale + shift) rounded to bf16 (loans_bn_apply_bf16's arithmetic)
    const float* affine;
};

constexpr int WPC = 32;     // pixels (reduction rows) per staged chunk
constexpr int WDEPTH = 4;   // register sets: global loads run WDEPTH - 1 chunks ahead of the chunk being computed

// BCO (output channels) x BJ (tap-channel columns) block tile on NWV waves: 4 as 2 x 2, or 8 as 2 x 4 (the 256 x 256 tile: one
// 512-thread block per CU, twice the MFMA work per staged byte of 128 x 128); RELU: relu(x)
template <int BCO, int BJ, int NWV, bool RELU>
__global__ __launch_bounds__(64 * NWV) void wgrad16_kernel(const Wgrad16Args a) {
    constexpr int WGM = 2, WGN = NWV / 2;           // wave grid
    static_assert(BCO <= BJ, "the loader's thread map follows the wider (X) tile");
    static_assert(WDEPTH % 2 == 0, "LDS stage = chunk parity = register-set parity");
    constexpr int TM = BCO / WGM / 32, TN = BJ / WGN / 32;   // MFMA tiles per wave
    constexpr int UPR = BJ / 8;             // 16-byte units per X row (thread map); Y rows use the first BCO/8
    constexpr int RPP = 64 * NWV / UPR;     // rows per loader pass
    constexpr int NP = WPC / RPP;           // passes (rows per thread) per chunk
    constexpr int SY = BCO + 32, SX = BJ + 32;      // padded LDS row strides (elements)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Ys = reinterpret_cast<__bf16*>(smem);   // [2][WPC][SY]
    __bf16* Xs = Ys + 2 * WPC * SY;                 // [2][WPC][SX]

    const loans_igemm_desc& d = a.d;
    const int tid = threadIdx.x;
    const int logical = xcd_remap16(blockIdx.x, gridDim.x);
    const int ntile = a.tiles_co * a.tiles_j;
    const int split = logical / ntile;
    const int tile = logical - split * ntile;
    const int tco = tile % a.tiles_co;
    const int tj = tile / a.tiles_co;
    const int unit = tid % UPR, prow = tid / UPR;

    // this thread's fixed column unit of the X tile: (tap, 8 channels)
    // LOANS_F_DENSE: inW / isx / dx count elements of packed 3-channel rows, a tap is Cin consecutive elements of a row
    const int ucin = (d.flags & LOANS_F_DENSE) ? 1 : d.Cin;
    const int cpt = d.Cin >> 3;
    const int ug = tj * UPR + unit;
    const int xtap = ug / cpt;
    const int xc8 = ug - xtap * cpt;
    const bool xtv = xtap < d.ntaps;
    const int dy = xtv ? (int)d.dy[xtap] : 0;
    const int dx = xtv ? (int)d.dx[xtap] : 0;
    const int yco = tco * BCO + unit * 8;
    const bool ythread = unit < BCO / 8;
    const bool yv = ythread && yco < d.Cout;             // Cout % 8 == 0 (checked)

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.gy), 0, (int)a.gy_bytes, 0x00020000);

    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    const int total_chunks = (a.M + WPC - 1) / WPC;
    if (c_end > total_chunks) c_end = total_chunks;

    // Per staged row: its pixel (b, y, x) for the bounds checks, and the two BYTE offsets the loads use, kept incrementally.
    // The gradient is read at grid pixel m itself (the launcher admits only the forward geometry: out row = grid pixel),
    // so its offset just advances by WPC pixels; the input offset advances by WPC pixel steps plus a row / image
    // correction per wrap.  (Recomputing both from (b, y, x) cost twelve 32-bit multiplies per row and chunk -- quarter
    // rate -- and made the loop VALU-bound: 294 vector instructions beside 32 MFMAs.)
    int pb[NP], py[NP], px[NP];
    unsigned glin[NP], xlin[NP];
    const int xconst = ((dy * d.inW + dx) * ucin + xc8 * 8) * 2;        // this thread's tap / channel offset (may be < 0)
    {
        const int gHW = d.gridH * d.gridW;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int m = c_begin * WPC + prow + RPP * p;
            pb[p] = m / gHW;
            const int rem = m - pb[p] * gHW;
            py[p] = rem / d.gridW;
            px[p] = rem - py[p] * d.gridW;
            glin[p] = (unsigned)(m * d.Cout + yco) * 2u;
            xlin[p] = (unsigned)(((pb[p] * d.inH + py[p] * d.isy) * d.inW + px[p] * d.isx) * ucin) * 2u + (unsigned)xconst;
        }
    }
    const float inv_gw = 1.f / (float)d.gridW, inv_gh = 1.f / (float)d.gridH;
    const unsigned g_step = (unsigned)(WPC * d.Cout) * 2u;
    const int x_step = WPC * d.isx * ucin * 2;
    const int x_row = (d.isy * d.inW - d.isx * d.gridW) * ucin * 2;          // per wrapped grid row   (|.| < 2^23: checked)
    const int x_img = (d.inH - d.isy * d.gridH) * d.inW * ucin * 2;          // per wrapped image

    // A chunk's compute (a few hundred MFMA cycles) is far shorter than a global load's latency, so the loads run
    // WDEPTH - 1 chunks ahead through a ring of register sets (chunk c in set c % WDEPTH); the wait in front of a
    // set's LDS write is a counted vmcnt that leaves the younger sets' loads in flight.
    // The loop is branch-free (hipcc keeps counted waits only along straight-line code): every block runs a multiple
    // of WDEPTH chunks, the ones beyond its slice load nothing (masked like rows beyond the last image) and add zeros.
    u32x4 ry[WDEPTH][NP], rx[WDEPTH][NP];
    // LOANS_F_AFFINE_IN: this thread's eight channels of [scale | shift] (its X column unit is fixed), and per register set the rows
    // that were really loaded (a row beyond the tensor or the block's slice must stay a zero operand, not become relu(shift))
    const bool aff = a.affine != nullptr;
    f32x4 as0 = {0.f, 0.f, 0.f, 0.f}, as1 = as0, at0 = as0, at1 = as0;
    if (aff && xtv) {
        as0 = *reinterpret_cast<const f32x4*>(a.affine + xc8 * 8); as1 = *reinterpret_cast<const f32x4*>(a.affine + xc8 * 8 + 4);
        at0 = *reinterpret_cast<const f32x4*>(a.affine + d.Cin + xc8 * 8); at1 = *reinterpret_cast<const f32x4*>(a.affine + d.Cin + xc8 * 8 + 4);
    }
    unsigned okm[WDEPTH];
#pragma unroll
    for (int k = 0; k < WDEPTH; ++k) okm[k] = 0u;
    int lc = c_begin;           // chunk the next load_chunk() fetches
    auto load_row = [&](int k, int p) {
        const int b = pb[p], y = py[p], x = px[p];
        const bool rv = (b < d.B) & (lc < c_end);
        const unsigned goff = glin[p] | ((unsigned)(rv & yv) - 1u);
        ry[k][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)goff, 0, 0);
        const int iy = __mul24(y, d.isy) + dy, ix = __mul24(x, d.isx) + dx;
        const unsigned ok = (unsigned)(rv & xtv) & (unsigned)((unsigned)iy < (unsigned)d.inH) &
                            (unsigned)((unsigned)ix < (unsigned)d.inW);
        const unsigned xoff = xlin[p] | (ok - 1u);
        rx[k][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xoff, 0, 0);
        okm[k] = (okm[k] & ~(1u << p)) | (ok << p);
        int nx = x + WPC;                       // advance one chunk: exact floor((v + .5) / n) for these small integers
        const int qx = (int)(((float)nx + 0.5f) * inv_gw);
        nx -= __mul24(qx, d.gridW);
        int ny = y + qx;
        const int qy = (int)(((float)ny + 0.5f) * inv_gh);
        ny -= __mul24(qy, d.gridH);
        px[p] = nx; py[p] = ny; pb[p] = b + qy;
        glin[p] += g_step;
        xlin[p] += (unsigned)(x_step + __mul24(qx, x_row) + __mul24(qy, x_img));
    };
    auto load_chunk = [&](int k) {
#pragma unroll
        for (int p = 0; p < NP; ++p) load_row(k, p);
        ++lc;
    };
    auto store_rows = [&](int buf, int k) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (ythread) *reinterpret_cast<u32x4*>(Ys + (buf * WPC + prow + RPP * p) * SY + unit * 8) = ry[k][p];
            if constexpr (RELU) rx[k][p] = __builtin_bit_cast(u32x4, relu_bf16x8(__builtin_bit_cast(bf16x8_t, rx[k][p])));
            if (aff) {
                const bf16x8_t v = __builtin_bit_cast(bf16x8_t, rx[k][p]);
                f32x4 lo = cvt_lo(v) * as0 + at0, hi = cvt_hi(v) * as1 + at1;
                lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
                hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
                const bf16x4_t ol = __builtin_convertvector(lo, bf16x4_t), oh = __builtin_convertvector(hi, bf16x4_t);
                const u32x4 t = __builtin_bit_cast(u32x4, __builtin_shufflevector(ol, oh, 0, 1, 2, 3, 4, 5, 6, 7));
                const u32x4 zero = {0u, 0u, 0u, 0u};
                rx[k][p] = ((okm[k] >> p) & 1u) ? t : zero;
            }
            *reinterpret_cast<u32x4*>(Xs + (buf * WPC + prow + RPP * p) * SX + unit * 8) = rx[k][p];
        }
    };

    // transposed fragment reads: 16-lane group g = lane >> 4 takes the block of pixels 8*(g>>1) + 4t .. +3 (MFMA half h =
    // g >> 1) and channels 16*(g&1) .. +15 of its 32-wide MFMA tile; lane 4q + p of the group addresses pixel row q,
    // channels 4p .. 4p+3, and receives channel (lane & 15) of the four pixels
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 15, fq = li >> 2, fp = li & 3, cg = (lane >> 4) & 1;
    const int trY = (8 * h + fq) * SY + wm * TM * 32 + 16 * cg + 4 * fp;
    const int trX = (8 * h + fq) * SX + wn * TN * 32 + 16 * cg + 4 * fp;
    typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_t;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
    for (int k = 0; k < WDEPTH - 1; ++k) load_chunk(k);
    store_rows(0, 0);
    __syncthreads();
    for (int c0 = c_begin; c0 < c_end; c0 += WDEPTH) {
#pragma unroll
        for (int k = 0; k < WDEPTH; ++k) {
            const int buf = k & 1;                      // WDEPTH is even: chunk parity = k parity
            load_chunk((k + WDEPTH - 1) % WDEPTH);      // that set went to LDS in the previous step
#pragma unroll
            for (int s = 0; s < WPC / 16; ++s) {
                const __bf16* Yb = Ys + (buf * WPC + 16 * s) * SY + trY;
                const __bf16* Xb = Xs + (buf * WPC + 16 * s) * SX + trX;
                bf16x8_t af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(Yb + i * 32));
                    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(Yb + i * 32 + 4 * SY));
                    af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(Xb + j * 32));
                    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(Xb + j * 32 + 4 * SX));
                    bf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            store_rows(buf ^ 1, (k + 1) % WDEPTH);
            __syncthreads();
        }
    }

    float* const slab = a.ws ? a.ws + (int64_t)split * a.slab : nullptr;     // (every block writes its whole tile: an idle one its zeros)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int jc = tj * BJ + wn * TN * 32 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = tco * BCO + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < d.Cout && jc < a.Ktot) {
                    if (slab) slab[(int64_t)co * a.Ktot + jc] = acc[i][j][e];
                    else atomic_add_f32(a.dw + (int64_t)co * a.Ktot + jc, acc[i][j][e]);
                }
            }
        }
}

// the pixel slices per output tile a request runs (0 = the default); sets the tile grid of `a`
template <int BCO, int BJ>
int plan_wgrad16(Wgrad16Args& a, int splits_req) {
    a.tiles_co = (a.d.Cout + BCO - 1) / BCO;
    a.tiles_j = (a.Ktot + BJ - 1) / BJ;
    const int total_chunks = (a.M + WPC - 1) / WPC;
    int splits = splits_req;
    if (splits <= 0) {
        const int ntile = a.tiles_co * a.tiles_j;
        splits = (1024 + ntile - 1) / ntile;            // ~4 blocks per CU in flight
        const int max_splits = (total_chunks + 7) / 8;   // >= 8 chunks (512 pixels) per block
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    if (splits > total_chunks) splits = total_chunks;
    a.chunks_per_split = (total_chunks + splits - 1) / splits;
    a.splits = (total_chunks + a.chunks_per_split - 1) / a.chunks_per_split;
    return a.splits;
}

template <int BCO, int BJ, int NWV, bool RELU>
int launch_wgrad16_r(Wgrad16Args& a, int splits_req, hipStream_t st) {
    static loans_device_once lds_limit_set;       // per template instance = per kernel, one bit per device
    constexpr size_t lds = (size_t)2 * WPC * ((BCO + 32) + (BJ + 32)) * 2;
    auto kern = wgrad16_kernel<BCO, BJ, NWV, RELU>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), lds)) return rc_;
    plan_wgrad16<BCO, BJ>(a, splits_req);
    hipLaunchKernelGGL(kern, dim3(a.tiles_co * a.tiles_j * a.splits), dim3(64 * NWV), lds, st, a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

template <int BCO, int BJ, int NWV = 4>
int launch_wgrad16(Wgrad16Args& a, int splits_req, hipStream_t st) {
    return (a.d.flags & LOANS_F_RELU_IN) ? launch_wgrad16_r<BCO, BJ, NWV, true>(a, splits_req, st)
                                         : launch_wgrad16_r<BCO, BJ, NWV, false>(a, splits_req, st);
}

}  // namespace

static int igemm_bf16s_impl(const void* in, const void* w, void* out, const float* bias, double* stats, const void* ref,
                           const void* addend, const loans_igemm_desc* d, float* partial, int splits, void* stream,
                           double* pair_stats = nullptr, bool pair = false) {
    if (!d || !in || !w || (!out && !partial)) return LOANS_EINVAL;
    if (partial && (d->flags & ~(LOANS_F_RELU_IN | LOANS_F_DENSE))) return LOANS_EINVAL;      // raw partial sums only
    if (d->B <= 0 || d->inH <= 0 || d->inW <= 0 || d->Cin <= 0 || (d->Cin & 7)) return LOANS_EINVAL;
    if (d->outH <= 0 || d->outW <= 0 || d->Cout <= 0 || (d->Cout & 7)) return LOANS_EINVAL;
    if (d->gridH <= 0 || d->gridW <= 0 || d->osy <= 0 || d->osx <= 0 || d->isy <= 0 || d->isx <= 0) return LOANS_EINVAL;
    if (d->oy0 < 0 || d->ox0 < 0) return LOANS_EINVAL;
    if ((d->gridH - 1) * d->osy + d->oy0 >= d->outH) return LOANS_EINVAL;
    if ((d->gridW - 1) * d->osx + d->ox0 >= d->outW) return LOANS_EINVAL;
    if (d->ntaps < 1 || d->ntaps > LOANS_MAX_TAPS) return LOANS_EINVAL;
    const bool dense = d->flags & LOANS_F_DENSE;
    if (!dense && d->ntaps > 32) return LOANS_EINVAL;       // the kernel keeps one 32-bit tap mask per tile row
    if (dense) {
        // no bounds masks in this mode: every K-row of every grid pixel has to lie inside its input row; rows and row
        // steps must keep the 16-byte loads 4-byte aligned (even element counts)
        if ((d->inW & 1) || (d->isx & 1)) return LOANS_EINVAL;
        for (int t = 0; t < d->ntaps; ++t) {
            if (d->dy[t] < 0 || d->dx[t] < 0 || (d->dx[t] & 1)) return LOANS_EINVAL;
            if ((d->gridH - 1) * d->isy + d->dy[t] >= d->inH) return LOANS_EINVAL;
            if ((d->gridW - 1) * d->isx + d->dx[t] + d->Cin > d->inW) return LOANS_EINVAL;
        }
    }
    if ((d->flags & LOANS_F_BIAS) && !bias) return LOANS_EINVAL;
    // the BN + ReLU in front of the convolution on load: the VGPR-fed 1 x 1 kernels only, `bias` = its [scale | shift]
    if ((d->flags & LOANS_F_AFFINE_IN) && (d->tile != LOANS_TILE_PW || !bias || partial || pair || (d->flags & ~(LOANS_F_AFFINE_IN | LOANS_F_STATS))))
        return LOANS_EINVAL;
    if ((d->flags & LOANS_F_STATS) && !stats) return LOANS_EINVAL;
    if (d->flags & LOANS_F_BNSUMS) {        // a data gradient's epilogue takes the sums of the BN below it: nothing else rides along
        if (!ref || !bias || !stats || partial || pair) return LOANS_EINVAL;
        if (d->flags & (LOANS_F_BIAS | LOANS_F_STATS | LOANS_F_MASK | LOANS_F_ADDEND | LOANS_F_ADDEND_MASK | LOANS_F_DENSE)) return LOANS_EINVAL;
    }
    if ((d->flags & (LOANS_F_MASK | LOANS_F_ADDEND_MASK)) && !ref) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_ADDEND_MASK) && !(d->flags & LOANS_F_ADDEND)) return LOANS_EINVAL;
    if ((d->flags & LOANS_F_ADDEND) && !addend) return LOANS_EINVAL;
    const int64_t lim = (int64_t)1 << 31;
    if ((int64_t)d->B * d->gridH * d->gridW >= lim) return LOANS_ERANGE;
    Igemm16Args a;
    a.in = static_cast<const __bf16*>(in); a.w = static_cast<const __bf16*>(w); a.out = static_cast<__bf16*>(out);
    a.bias = bias; a.stats = stats;
    a.ref = static_cast<const __bf16*>(ref); a.addend = static_cast<const __bf16*>(addend);
    a.d = *d;
    a.M = d->B * d->gridH * d->gridW;
    a.Ktot = d->ntaps * d->Cin;
    a.dbg = 0;
#ifdef LOANS_EXPERIMENT
    if (const char* e = getenv("LOANS_DBG")) a.dbg = atoi(e);
#endif
    a.partial = partial;
    a.csplit = 0; a.out_c = d->Cout; a.tensor_bytes = 0; a.stats2 = nullptr;
    if (pair) {                 // `d` describes the stacked GEMM: Cout = 2 x the channels of either convolution
        if (partial || (d->Cout & 63) || (d->flags & ~(LOANS_F_STATS | LOANS_F_RELU_IN))) return LOANS_EINVAL;
        if ((d->flags & LOANS_F_STATS) && !pair_stats) return LOANS_EINVAL;
        a.csplit = a.out_c = d->Cout / 2;
        a.stats2 = pair_stats;
    }
    a.splits = partial ? splits : 1;
    a.nchunks = (a.Ktot + BKH - 1) / BKH;
    {
        const int64_t ib = (int64_t)d->B * d->inH * d->inW * (dense ? 1 : d->Cin) * 2;
        const int64_t wb = (int64_t)d->Cout * a.Ktot * 2;
        const int64_t ob = (int64_t)d->B * d->outH * d->outW * d->Cout * 2;
        if (ib >= 0xFFFFFFF0ll || wb >= 0xFFFFFFF0ll || ob >= 0xFFFFFFF0ll) return LOANS_ERANGE;   // 32-bit buffer offsets
        a.in_bytes = (unsigned)ib; a.w_bytes = (unsigned)wb; a.out_bytes = (unsigned)ob;
        a.nt_out = loans_conv_nt((size_t)ob);
        if (pair) a.tensor_bytes = (unsigned)(ob / 2);
    }
    detect_tap_grid16(d, a);
    hipStream_t st = as_stream(stream);
    int tile = d->tile;
    if (tile == 0) {
        const int64_t big = (int64_t)((a.M + 127) / 128) * ((d->Cout + 127) / 128);
        tile = d->Cout <= 64 ? LOANS_TILE_128x64 : (big >= 512 ? LOANS_TILE_128x128 : LOANS_TILE_64x64);
    }
    const bool halo_tile = (tile >= LOANS_TILE_HALO_128 && tile <= LOANS_TILE_WS64) || tile == LOANS_TILE_HALO_256x128 || tile == LOANS_TILE_HALO_256x256 || tile == LOANS_TILE_WSW64;
    if (partial && halo_tile) return LOANS_EINVAL;          // the halo tiles have no split-K form
    if (pair && (tile == LOANS_TILE_STEM || halo_tile)) return LOANS_EINVAL;
    if (tile == LOANS_TILE_STEM) {          // the dense RGB stem as a direct convolution (stem.hip)
        if (partial || splits > 1) return LOANS_EINVAL;
        return loans_stem7_bf16s_launch(in, w, out, bias, stats, d, st);
    }
    if (tile == LOANS_TILE_PW) {            // short-K 1 x 1 convolutions, operands never in LDS (pw_bf16.hip); w in fragment order
        if (partial || splits > 1 || pair) return LOANS_EINVAL;
        return loans_pw16_launch(in, w, out, stats, (d->flags & LOANS_F_AFFINE_IN) ? bias : nullptr, d, st);
    }
    switch (tile) {
        case LOANS_TILE_128x128: return launch_igemm16<128, 128, 2, 2>(a, st);
        case LOANS_TILE_128x64: return launch_igemm16<128, 64, 2, 2>(a, st);
        case LOANS_TILE_64x64: return launch_igemm16<64, 64, 2, 2>(a, st);
        case LOANS_TILE_256x64: return launch_igemm16<256, 64, 4, 1>(a, st);
        case LOANS_TILE_128x128 | LOANS_TILE_DEEP: return launch_igemm16<128, 128, 2, 2, true>(a, st);
        case LOANS_TILE_128x64 | LOANS_TILE_DEEP: return launch_igemm16<128, 64, 2, 2, true>(a, st);
        case LOANS_TILE_64x64 | LOANS_TILE_DEEP: return launch_igemm16<64, 64, 2, 2, true>(a, st);
        case LOANS_TILE_256x128: return launch_igemm16<256, 128, 4, 2>(a, st);      // 512 threads: eight 64 x 64 wave tiles
        case LOANS_TILE_256x256: return launch_igemm16<256, 256, 2, 4>(a, st);      // 512 threads: eight 128 x 64 wave tiles
        case LOANS_TILE_256x256PP: return launch_igemm16pp<false>(a, st);           // the same tile, wave rows half a phase apart
        case LOANS_TILE_256x256PP16: return launch_igemm16pp<true>(a, st);          // ... on v_mfma_f32_16x16x32_bf16
        case LOANS_TILE_HALO_128:
        case LOANS_TILE_HALO_128x64:
        case LOANS_TILE_HALO_256x64:
        case LOANS_TILE_HALO_128x64S:
        case LOANS_TILE_HALO_256x128:
        case LOANS_TILE_HALO_256x256:
        case LOANS_TILE_WSW64:
        case LOANS_TILE_WS64:
            return loans_halo16_launch(in, w, out, bias, stats, ref, addend, d, tile, a.in_bytes, a.w_bytes, a.out_bytes, st);
        default: return LOANS_EINVAL;
    }
}

extern "C" int loans_igemm_bf16s(const void* in, const void* w, void* out, const float* bias, double* stats,
                                 const void* ref, const void* addend, const loans_igemm_desc* d, void* stream) {
    if (!out) return LOANS_EINVAL;
    return igemm_bf16s_impl(in, w, out, bias, stats, ref, addend, d, nullptr, 1, stream);
}

extern "C" int loans_igemm_pair_bf16s(const void* in, const void* w_ab, void* out_ab, double* stats_a, double* stats_b,
                                      const loans_igemm_desc* d, void* stream) {
    if (!d || !out_ab || d->Cout <= 0 || (d->Cout & 31)) return LOANS_EINVAL;
    loans_igemm_desc d2 = *d;
    d2.Cout = 2 * d->Cout;
    return igemm_bf16s_impl(in, w_ab, out_ab, nullptr, stats_a, nullptr, nullptr, &d2, nullptr, 1, stream, stats_b, true);
}

extern "C" int loans_igemm_bf16s_splitk(const void* in, const void* w, float* partial, const loans_igemm_desc* d, int32_t splits,
                                        void* stream) {
    if (!partial || splits < 1 || splits > 64) return LOANS_EINVAL;
    return igemm_bf16s_impl(in, w, nullptr, nullptr, nullptr, nullptr, nullptr, d, partial, splits, stream);
}

namespace {
// the epilogue of a split-K convolution on bf16 storage: out = bf16(act(partial + bias)), statistics of (partial + bias) in fp64.
// 8 channels per thread, C8 = Cout / 8 divides 256; one pass over the finished sums.
__global__ __launch_bounds__(256) void igemm16_finalize_kernel(const float* partial, __bf16* out, const float* bias, double* stats,
                                                               const __bf16* ref, const __bf16* addend, int flags, int64_t rows,
                                                               int C8, int rows_per_block) {
    __shared__ float red[2][256][8];
    const int tid = threadIdx.x;
    const int cl = tid % C8, rl = tid / C8, RL = 256 / C8;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    const bool f_bias = flags & LOANS_F_BIAS, f_stats = flags & LOANS_F_STATS, f_mask = flags & LOANS_F_MASK;
    const bool f_add = flags & LOANS_F_ADDEND, f_addmask = flags & LOANS_F_ADDEND_MASK;
    f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = b_lo;
    if (f_bias) {
        b_lo = *reinterpret_cast<const f32x4*>(bias + cl * 8);
        b_hi = *reinterpret_cast<const f32x4*>(bias + cl * 8 + 4);
    }
    auto keep_pos = [](f32x4 v, f32x4 m) {
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        return v;
    };
    f32x4 s1l = {0.f, 0.f, 0.f, 0.f}, s1h = s1l, s2l = s1l, s2h = s1l;
    for (int64_t r = r0 + rl; r < r1; r += RL) {
        const int64_t o = (r * C8 + cl) * 8;
        f32x4 lo = *reinterpret_cast<const f32x4*>(partial + o) + b_lo;
        f32x4 hi = *reinterpret_cast<const f32x4*>(partial + o + 4) + b_hi;
        s1l += lo; s1h += hi;
        s2l += lo * lo; s2h += hi * hi;
        if (f_mask || f_addmask) {
            const bf16x8_t rf = *reinterpret_cast<const bf16x8_t*>(ref + o);
            const f32x4 ml = cvt_lo(rf), mh = cvt_hi(rf);
            if (f_mask) { lo = keep_pos(lo, ml); hi = keep_pos(hi, mh); }
            if (f_add) {
                const bf16x8_t ad = *reinterpret_cast<const bf16x8_t*>(addend + o);
                f32x4 al = cvt_lo(ad), ah = cvt_hi(ad);
                if (f_addmask) { al = keep_pos(al, ml); ah = keep_pos(ah, mh); }
                lo += al; hi += ah;
            }
        } else if (f_add) {
            const bf16x8_t ad = *reinterpret_cast<const bf16x8_t*>(addend + o);
            lo += cvt_lo(ad); hi += cvt_hi(ad);
        }
        bf16x8_t v;
        const bf16x4_t ol = __builtin_convertvector(lo, bf16x4_t), oh = __builtin_convertvector(hi, bf16x4_t);
        v[0] = ol[0]; v[1] = ol[1]; v[2] = ol[2]; v[3] = ol[3];
        v[4] = oh[0]; v[5] = oh[1]; v[6] = oh[2]; v[7] = oh[3];
        *reinterpret_cast<bf16x8_t*>(out + o) = v;
    }
    if (!f_stats) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[0][tid][e] = s1l[e]; red[0][tid][4 + e] = s1h[e];
        red[1][tid][e] = s2l[e]; red[1][tid][4 + e] = s2h[e];
    }
    __syncthreads();
    for (int c = tid; c < C8 * 8; c += 256) {
        float t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < RL; ++k) {
            t1 += red[0][(c >> 3) + k * C8][c & 7];
            t2 += red[1][(c >> 3) + k * C8][c & 7];
        }
        const int C = C8 * 8;
        double* st = stats + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * C;
        atomic_add_f64(st + c, (double)t1);
        atomic_add_f64(st + C + c, (double)t2);
    }
}
}  // namespace

// Statistics are those of (sum + bias) BEFORE mask / addend -- what the fused epilogue of loans_igemm_bf16s takes from its
// accumulators (the BN that follows a convolution sees the convolution's output; mask / addend belong to data gradients, which
// carry no statistics).
extern "C" int loans_igemm_finalize_bf16(const float* partial, void* out, const float* bias, double* stats, const void* ref,
                                         const void* addend, int32_t flags, int64_t rows, int32_t Cout, void* stream) {
    if (!partial || !out || rows <= 0 || Cout <= 0 || (Cout & 7)) return LOANS_EINVAL;
    const int C8 = Cout / 8;
    if (C8 > 256 || 256 % C8) return LOANS_EINVAL;          // the thread map: Cout / 8 divides 256
    if ((flags & LOANS_F_BIAS) && !bias) return LOANS_EINVAL;
    if ((flags & LOANS_F_STATS) && !stats) return LOANS_EINVAL;
    if ((flags & (LOANS_F_MASK | LOANS_F_ADDEND_MASK)) && !ref) return LOANS_EINVAL;
    if ((flags & LOANS_F_ADDEND) && !addend) return LOANS_EINVAL;
    if (flags & ~(LOANS_F_BIAS | LOANS_F_STATS | LOANS_F_MASK | LOANS_F_ADDEND | LOANS_F_ADDEND_MASK)) return LOANS_EINVAL;
    const int RL = 256 / C8;
    int rows_per_block = RL * 8;
    const int64_t nblk = (rows + rows_per_block - 1) / rows_per_block;
    if (nblk >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    hipLaunchKernelGGL(igemm16_finalize_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), partial,
                       static_cast<__bf16*>(out), bias, stats, static_cast<const __bf16*>(ref),
                       static_cast<const __bf16*>(addend), flags, rows, C8, rows_per_block);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0 || (n & 3)) return LOANS_EINVAL;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, as_stream(stream), src,
                       static_cast<__bf16*>(dst), n / 4);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

extern "C" int loans_repack_dgrad_bf16(const float* src, void* dst, int32_t Cout, int32_t Cin, int32_t src_taps,
                                       const int32_t* tapsel_host, int32_t ntaps, void* stream) {
    if (!src || !dst || !tapsel_host || Cout <= 0 || Cin <= 0 || src_taps <= 0) return LOANS_EINVAL;
    if (ntaps < 1 || ntaps > LOANS_MAX_TAPS) return LOANS_EINVAL;
    Repack16Args a;
    a.src = src; a.dst = static_cast<__bf16*>(dst); a.Cout = Cout; a.Cin = Cin; a.src_taps = src_taps; a.ntaps = ntaps;
    for (int i = 0; i < ntaps; ++i) {
        if (tapsel_host[i] < 0 || tapsel_host[i] >= src_taps) return LOANS_EINVAL;
        a.tapsel[i] = tapsel_host[i];
    }
    dim3 grid((Cout + 31) / 32, (Cin + 31) / 32, ntaps);
    hipLaunchKernelGGL(repack_dgrad16_kernel, grid, dim3(256), 0, as_stream(stream), a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

// One implementation behind loans_wgrad_bf16s (atomics into dw), loans_wgrad_bf16s_ws (partial slabs + fold) and
// loans_wgrad_bf16s_ws_floats (plan only: *need = floats of workspace the request takes, nothing is launched)
static int wgrad_bf16s_impl(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, int32_t splits, float* ws,
                            int64_t ws_floats, int64_t* need, void* stream, const float* affine = nullptr) {
    const bool plan_only = need != nullptr;
    if (!d || (!plan_only && (!x || !gy || !dw))) return LOANS_EINVAL;
    if (d->B <= 0 || d->inH <= 0 || d->inW <= 0 || d->Cin <= 0 || (d->Cin & 7)) return LOANS_EINVAL;
    if (d->outH <= 0 || d->outW <= 0 || d->Cout <= 0 || (d->Cout & 7)) return LOANS_EINVAL;
    if (d->gridH <= 0 || d->gridW <= 0 || d->osy <= 0 || d->osx <= 0 || d->isy <= 0 || d->isx <= 0) return LOANS_EINVAL;
    if (d->oy0 < 0 || d->ox0 < 0) return LOANS_EINVAL;
    if ((d->gridH - 1) * d->osy + d->oy0 >= d->outH) return LOANS_EINVAL;
    if ((d->gridW - 1) * d->osx + d->ox0 >= d->outW) return LOANS_EINVAL;
    if (d->ntaps < 1 || d->ntaps > LOANS_MAX_TAPS) return LOANS_EINVAL;
    // the kernel reads the gradient at grid pixel m itself and keeps the input offset incrementally with 24-bit multiplies
    if (d->osy != 1 || d->osx != 1 || d->oy0 || d->ox0 || d->outH != d->gridH || d->outW != d->gridW) return LOANS_EINVAL;
    {
        const int64_t uc = (d->flags & LOANS_F_DENSE) ? 1 : d->Cin;
        const int64_t xr = ((int64_t)d->isy * d->inW - (int64_t)d->isx * d->gridW) * uc * 2;
        const int64_t xi = ((int64_t)d->inH - (int64_t)d->isy * d->gridH) * d->inW * uc * 2;
        const int64_t lim24 = (int64_t)1 << 23;
        if (xr <= -lim24 || xr >= lim24 || xi <= -lim24 || xi >= lim24) return LOANS_ERANGE;
        if (d->gridW >= lim24 || d->gridH >= lim24) return LOANS_ERANGE;
    }
    const bool dense = d->flags & LOANS_F_DENSE;
    if (dense) {            // as in loans_igemm_bf16s
        if ((d->inW & 1) || (d->isx & 1)) return LOANS_EINVAL;
        for (int t = 0; t < d->ntaps; ++t) {
            if (d->dy[t] < 0 || d->dx[t] < 0 || (d->dx[t] & 1)) return LOANS_EINVAL;
            if ((d->gridH - 1) * d->isy + d->dy[t] >= d->inH) return LOANS_EINVAL;
            if ((d->gridW - 1) * d->isx + d->dx[t] + d->Cin > d->inW) return LOANS_EINVAL;
        }
    }
    if ((int64_t)d->B * d->gridH * d->gridW >= ((int64_t)1 << 31)) return LOANS_ERANGE;
    Wgrad16Args a;
    a.x = static_cast<const __bf16*>(x); a.gy = static_cast<const __bf16*>(gy); a.dw = dw; a.d = *d;
    a.M = d->B * d->gridH * d->gridW;
    a.Ktot = d->ntaps * d->Cin;
    a.ws = nullptr;
    a.slab = (int64_t)d->Cout * a.Ktot;
    a.affine = nullptr;
    if (d->flags & LOANS_F_AFFINE_IN) {     // 1 x 1 / 1 convolutions on the GEMM tiles only; x = the BN's input, affine = [scale | shift][Cin]
        if (plan_only) { /* the slab count does not depend on it */ }
        else if (!affine) return LOANS_EINVAL;
        if (d->ntaps != 1 || d->dy[0] != 0 || d->dx[0] != 0 || d->isy != 1 || d->isx != 1 || (d->flags & ~LOANS_F_AFFINE_IN)) return LOANS_EINVAL;
        a.affine = affine;
    }
    {
        const int64_t xb = (int64_t)d->B * d->inH * d->inW * (dense ? 1 : d->Cin) * 2;
        const int64_t gb = (int64_t)d->B * d->outH * d->outW * d->Cout * 2;
        if (xb >= 0xFFFFFFF0ll || gb >= 0xFFFFFFF0ll) return LOANS_ERANGE;
        a.x_bytes = (unsigned)xb; a.gy_bytes = (unsigned)gb;
    }
    hipStream_t st = as_stream(stream);
    int tile = d->tile;
    if (tile == 0) tile = (d->Cout <= 64) ? (a.Ktot <= 64 ? LOANS_TILE_64x64 : LOANS_TILE_64x128) : LOANS_TILE_128x128;
    const bool halo = tile == LOANS_TILE_WGHALO_64 || tile == LOANS_TILE_WGHALO_128;
    if (halo && (d->flags & LOANS_F_AFFINE_IN)) return LOANS_EINVAL;
    if (tile == LOANS_TILE_STEM) {          // the dense RGB stem's weight gradient as a direct kernel (stem.hip): one slab per block
        const int slabs = loans_stem7_wgrad_bf16_slabs(d);
        if (slabs < 1) return LOANS_EINVAL;
        if (plan_only) { *need = (int64_t)slabs * a.slab; return LOANS_OK; }
        if (ws && ws_floats < (int64_t)slabs * a.slab) return LOANS_EINVAL;
        const int rc = loans_stem7_wgrad_bf16_launch(x, gy, dw, d, ws, st);
        if (rc != LOANS_OK || !ws) return rc;
        return loans_fold_slabs_f32(ws, dw, a.slab, slabs, stream);
    }
    // the slabs this request writes: one per pixel slice (the launchers' own arithmetic)
    int slabs;
    if (halo) slabs = loans_wgrad_halo16_slabs(d, tile, splits);
    else if (tile == LOANS_TILE_64x64) slabs = plan_wgrad16<64, 64>(a, splits);
    else if (tile == LOANS_TILE_128x128) slabs = plan_wgrad16<128, 128>(a, splits);
    else if (tile == LOANS_TILE_64x128) slabs = plan_wgrad16<64, 128>(a, splits);
    else if (tile == LOANS_TILE_256x256) slabs = plan_wgrad16<256, 256>(a, splits);
    else return LOANS_EINVAL;
    if (slabs < 1) return slabs < 0 ? slabs : LOANS_EINVAL;
    if (plan_only) { *need = (int64_t)slabs * a.slab; return LOANS_OK; }
    const bool use_ws = ws != nullptr;
    if (use_ws) {
        if (ws_floats < (int64_t)slabs * a.slab) return LOANS_EINVAL;
        a.ws = ws;
    }
    int rc;
    if (halo) rc = loans_wgrad_halo16_launch(x, gy, dw, d, tile, splits, a.x_bytes, a.gy_bytes, a.ws, nullptr, st);
    else if (tile == LOANS_TILE_64x64) rc = launch_wgrad16<64, 64>(a, splits, st);
    else if (tile == LOANS_TILE_128x128) rc = launch_wgrad16<128, 128>(a, splits, st);
    else if (tile == LOANS_TILE_64x128) rc = launch_wgrad16<64, 128>(a, splits, st);
    else rc = launch_wgrad16<256, 256, 8>(a, splits, st);
    if (rc != LOANS_OK || !use_ws) return rc;
    return loans_fold_slabs_f32(ws, dw, a.slab, slabs, stream);
}

extern "C" int loans_wgrad_bf16s(const void* x, const void* gy, float* dw, const loans_igemm_desc* d,
                                 int32_t splits, void* stream) {
    return wgrad_bf16s_impl(x, gy, dw, d, splits, nullptr, 0, nullptr, stream);
}

extern "C" int loans_wgrad_bf16s_ws(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, int32_t splits,
                                    float* ws, int64_t ws_floats, void* stream) {
    if (!ws) return LOANS_EINVAL;
    return wgrad_bf16s_impl(x, gy, dw, d, splits, ws, ws_floats, nullptr, stream);
}

extern "C" int loans_wgrad_bf16s_affine_ws(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, int32_t splits,
                                           float* ws, int64_t ws_floats, const float* affine, void* stream) {
    if (!ws || !affine || !d || !(d->flags & LOANS_F_AFFINE_IN)) return LOANS_EINVAL;
    return wgrad_bf16s_impl(x, gy, dw, d, splits, ws, ws_floats, nullptr, stream, affine);
}

extern "C" int64_t loans_wgrad_bf16s_ws_floats(const loans_igemm_desc* d, int32_t splits) {
    int64_t need = 0;
    const int rc = wgrad_bf16s_impl(nullptr, nullptr, nullptr, d, splits, nullptr, 0, &need, nullptr);
    return rc == LOANS_OK ? need : (int64_t)rc;
}

#ifdef LOANS_STAMPS
extern "C" int loans_debug_read_stamps16c(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps16c), sizeof(unsigned long long) * n);
}
extern "C" int loans_debug_read_stamps16b(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps16b), sizeof(unsigned long long) * n);
}
#endif

#ifdef LOANS_STAMPS
extern "C" int loans_debug_read_stamps16(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps16), sizeof(unsigned long long) * n);
}
#endif
