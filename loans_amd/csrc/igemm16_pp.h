// LOANS_TILE_256x256PP -- the 256 x 256 implicit-GEMM tile of igemm_bf16.hip with a PING-PONG K loop (round 6).
// Included inside igemm_bf16.hip's anonymous namespace (shares Igemm16Args, the tap / row tables and the epilogue conventions of
// igemm16_kernel; reached through loans_igemm_bf16s).
//
// Why: igemm16_kernel<256, 256, 2, 4> runs its eight waves in lock-step -- every wave reads fragments, issues its share of the
// next chunk's LDS-DMA and then its MFMAs, one barrier per 64-deep chunk -- so the two waves of a SIMD want the matrix pipe at
// the same time and the DMA / LDS pipes at the same time: MFMA pipe 51-57 % busy on the 256- and 512-channel layers
// (profiles/r5_conv_kernels_sq.txt), and neither a lighter staging path (LOANS_TILE_HALO_256x256) nor one wave per SIMD with
// 128 x 128 wave tiles changed that (profiles/r6_halo256x256_vs_igemm.txt, r6_w4_one_wave_per_simd.txt).  Here the two wave
// rows of the block (waves 0-3 and 4-7: the two waves of every SIMD) run HALF A PHASE APART: while one group issues its eight
// MFMAs (one quadrant of its 128 x 64 output x one 64-deep chunk = 256 pipe cycles), the other reads that quadrant's fragments
// and issues one half-tile of LDS-DMA; a raw s_barrier flips the roles.  The matrix pipe of a SIMD always has exactly one wave
// on it, and fragment reads, DMA issue and address arithmetic sit beside the partner's MFMAs instead of beside the wave's own.
// (The 8-phase structure of cdna_hip_programming.md, 5 "The 256^2 8-phase template", on the implicit-GEMM operand gather.)
//
// Geometry: operand tiles in HALVES of 128 rows x 64 k (16 KiB), two K chunks resident: LDS = 2 sets x {A0, A1, B0, B1}.
// Wave (wr, wc) = (wave >> 2, wave & 3) owns, of row half i and column half j, rows i 128 + wr 64 .. + 63 and columns
// j 128 + wc 32 .. + 31: four 64 x 32 quadrants Q(i, j), two 32 x 32 MFMA tiles each.  Per chunk t, four phases:
//     phase 0  read A0, B0 (12 ds_read_b128)   stage B0(t + 1)   MFMA Q(0,0)
//     phase 1  read B1     ( 4)                stage A0(t + 2)   MFMA Q(0,1)
//     phase 2  read A1     ( 8)                stage B1(t + 2)   MFMA Q(1,1)
//     phase 3  read B0     ( 4)                stage A1(t + 2)   MFMA Q(1,0)     s_waitcnt vmcnt(6): all of chunk t + 1 has landed
// A half-tile is re-staged in the phase after its last read; every phase retires its own LDS reads (lgkmcnt(0)) BEFORE its
// first barrier, so the partner group's DMA, issued behind that barrier, never meets a read in flight.  Landing: a wave's
// counted vmcnt in phase 3 covers everything it issued up to phase 0 of this chunk (three half-tiles = 6 pieces stay in flight);
// the reads of chunk t + 1 start two barriers later, after the OTHER group's wait as well.  K is walked in igemm16_kernel's
// order (tap-major, 64 channels at a time, four 16-deep MFMA steps per chunk), so results are bit-identical to its tiles.

constexpr int PP_HT = 128 * BKH;                                   // elements of a half-tile
constexpr size_t pp_aux_bytes() {
    constexpr size_t stage = (size_t)8 * PP_HT * 2, cs = (size_t)128 * (256 + 4) * 4;
    return stage > cs ? stage : cs;
}
constexpr size_t pp_lds_bytes() { return pp_aux_bytes() + LOANS_MAX_TAPS * 4 + 256 * 4; }

// M16: the contraction on v_mfma_f32_16x16x32_bf16 (sixteen per quadrant and chunk) instead of v_mfma_f32_32x32x16_bf16 (eight): the
// same fragment bytes, LDS reads and cycles per FLOP; the chip holds a higher clock on the 16 x 16 shape under load
// (MI355X_MICROARCH.md, DVFS give-back item 7).  A product of 32 k values per instruction instead of 16: another rounding order
// than igemm16_kernel's, results agree to the position of rare bf16 roundings (as the halo tiles' do).
template <bool RELU, bool M16>
__global__ __launch_bounds__(512) void igemm16pp_kernel(const Igemm16Args a) {
    constexpr int BM = 256, BN = 256, NT = 512, RPP = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* T = reinterpret_cast<__bf16*>(smem);           // [set][A0 A1 B0 B1][128][64]
    int* taps = reinterpret_cast<int*>(smem + pp_aux_bytes());
    unsigned* opix = reinterpret_cast<unsigned*>(taps + LOANS_MAX_TAPS);   // [BM] output row byte offset, ~0u = no row

    const loans_igemm_desc& d = a.d;
    const int tid = threadIdx.x;
    const int logical = xcd_remap16(blockIdx.x, gridDim.x);
    const int tn = logical % a.tiles_n;
    const int tm = logical / a.tiles_n;
    const int nch = a.nchunks;
    const int lrow = tid >> 3;
    const int lu = (tid & 7) ^ ((tid >> 4) & 7);            // K unit this thread stages: slot ^ key(row)
    const int pbytes = d.Cin * 2;
    if (tid < LOANS_MAX_TAPS) {
        const int t = tid < d.ntaps ? tid : 0;
        taps[tid] = (int(d.dy[t]) * d.inW + int(d.dx[t])) * pbytes;
    }
    // per staged row (tile row lrow + 64 q, q = 0..3: rows 64 q' of half q >> 1): base pixel offset, bitmask of taps that read zero
    unsigned rowoff[4], badmask[4];
    {
        const int gHW = d.gridH * d.gridW;
        const float inv_gw = 1.f / (float)d.gridW, inv_gh = 1.f / (float)d.gridH;
        const int m0 = tm * BM + lrow;
        int b = m0 / gHW;
        int rem = m0 - b * gHW;
        int y = rem / d.gridW;
        int x = rem - y * d.gridW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + RPP * i;
            unsigned pixoff = 0xFFFFFFFFu;
            unsigned long long mask = 0;
            rowoff[i] = 0;
            if (m < a.M) {
                const int iy0 = y * d.isy, ix0 = x * d.isx;
                rowoff[i] = (unsigned)((b * d.inH + iy0) * d.inW + ix0) * (unsigned)pbytes;
                pixoff = (unsigned)((b * d.outH + y * d.osy + d.oy0) * d.outW + x * d.osx + d.ox0) * (unsigned)a.out_c * 2u;
                if (a.ap.nx > 0) {
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
    unsigned woff[4], wbad[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = tn * BN + lrow + RPP * i;
        wbad[i] = n < d.Cout ? 0u : 0xFFFFFFFFu;
        woff[i] = n < d.Cout ? (unsigned)n * (unsigned)a.Ktot * 2u + (unsigned)lu * 16u : 0u;
    }
    // ---- the A side walks K chunk by chunk (igemm16_kernel's bookkeeping): this thread's K unit of the chunk being staged, its tap
    // and channel unit, the tap's gather offset and mask position; the tap-table entry of the following chunk is read a chunk ahead
    int ua = lu;
    int tap = ua / cpt, c8 = ua - tap * cpt;
    auto step_tap = [&](int& t, int& c) {
        t += q8;
        c += r8;
        const int wrap = c >= cpt;
        c -= wrap ? cpt : 0;
        t += wrap;
    };
    unsigned toff = (unsigned)taps[min(tap, LOANS_MAX_TAPS - 1)] + (unsigned)c8 * 16u;
    unsigned tcs = (unsigned)min(tap, 31);
    unsigned kba = (unsigned)((kunits - 1 - ua) >> 31);
    int tap_n = tap, c8_n = c8;
    step_tap(tap_n, c8_n);
    int traw_n = taps[min(tap_n, LOANS_MAX_TAPS - 1)];
    auto advance_a = [&]() {
        ua += 8;
        tap = tap_n;
        c8 = c8_n;
        toff = (unsigned)traw_n + (unsigned)c8 * 16u;
        tcs = (unsigned)min(tap, 31);
        kba = (unsigned)((kunits - 1 - ua) >> 31);
        step_tap(tap_n, c8_n);
        traw_n = taps[min(tap_n, LOANS_MAX_TAPS - 1)];
    };

#ifdef LOANS_EXPERIMENT
    // ablations (wrong results): 8 = no LDS-DMA inside the K loop, 16 = no fragment reads inside it, 32 = no MFMAs, 64 = the A gather
    // from a few cache-hot rows
    const bool x_nodma = a.dbg & 8, x_noread = a.dbg & 16, x_nomma = a.dbg & 32;
    if (a.dbg & 64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { rowoff[i] = (unsigned)((d.inW + 1) * pbytes) + (rowoff[i] & 0x3FFu); badmask[i] = 0; }
    }
    bool x_inloop = false;
#else
    constexpr bool x_nodma = false, x_noread = false, x_nomma = false, x_inloop = false;
#endif
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // one half-tile = 128 rows = two 1 KiB pieces per thread (rows 8 wave .. + 7 of either 64-row block)
    auto stage_a = [&](int set, int half) {        // A half `half` of the chunk the A state stands at
        if (x_nodma && x_inloop) return;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = 2 * half + q;
            const unsigned bad = (unsigned)__builtin_amdgcn_sbfe((int)badmask[i], tcs, 1u);
            const unsigned off = (rowoff[i] + toff) | bad | kba;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(T + (set * 4 + half) * PP_HT + (64 * q + 8 * wave_u) * BKH), 16, (int)off, 0, 0, 0);
        }
    };
    auto stage_b = [&](int set, int half, int kt) { // B half `half` of chunk kt
        if (x_nodma && x_inloop) return;
        const int u = lu + 8 * kt;
        const unsigned kb = (unsigned)((kunits - 1 - u) >> 31);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = 2 * half + q;
            const unsigned off = (woff[i] + (unsigned)kt * 128u) | wbad[i] | kb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(T + (set * 4 + 2 + half) * PP_HT + (64 * q + 8 * wave_u) * BKH), 16, (int)off, 0, 0, 0);
        }
    };

    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int wr_u = __builtin_amdgcn_readfirstlane(wr);
    // fragments.  32 x 32 x 16: lane (r, h) = row r of a 32-row tile, k = 16 s + 8 h .. + 7 of step s (4 steps per chunk);
    // 16 x 16 x 32: lane (fr, fq) = row fr of a 16-row tile, k = 32 s + 8 fq .. + 7 of step s (2 steps).  Either way one ds_read_b128
    // of unit (k / 8) ^ key(row), key = (row >> 1) & 7 -- sixteen different 16-byte slots per 16-lane read group for both maps.
    const int fr = lane & 15, fq = lane >> 4;
    const int frow = M16 ? fr : r, fun = M16 ? fq : h;
    const int fkey = (frow >> 1) & 7;
    const int fragA = (wr * 64 + frow) * BKH + ((fun ^ fkey) & 7) * 8;
    const int fragB = (wc * 32 + frow) * BKH + ((fun ^ fkey) & 7) * 8;
    constexpr int KS = M16 ? 2 : 4, KX = M16 ? 32 : 16;      // MFMA steps per chunk, elements the step index moves the unit by
    constexpr int TR = M16 ? 16 : 32;                         // rows (and columns) of an MFMA tile
    constexpr int MT = 64 / TR, NTL = 32 / TR;                // tiles per quadrant: rows, columns
    bf16x8_t af[MT * KS], bfr[NTL * KS];                      // 8 + 4 fragments either way
    auto read_a = [&](int set, int half) {
        if (x_noread && x_inloop) return;
        const __bf16* Ab = T + (set * 4 + half) * PP_HT;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m * KS + s] = *reinterpret_cast<const bf16x8_t*>(Ab + ((fragA + m * TR * BKH) ^ (s * KX)));
    };
    auto read_b = [&](int set, int half) {
        if (x_noread && x_inloop) return;
        const __bf16* Bb = T + (set * 4 + 2 + half) * PP_HT;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int n = 0; n < NTL; ++n) bfr[n * KS + s] = *reinterpret_cast<const bf16x8_t*>(Bb + ((fragB + n * TR * BKH) ^ (s * KX)));
    };
    auto relu_a = [&]() {
        if constexpr (RELU) {
#pragma unroll
            for (int q = 0; q < MT * KS; ++q) af[q] = relu_bf16x8(af[q]);
        }
    };

    // accumulators of quadrant (i, j): 2 tiles of 32 x 32 (16 floats per lane) or 4 x 2 tiles of 16 x 16 (4 floats per lane)
    typedef typename std::conditional<M16, f32x4, f32x16>::type acc_t;
    constexpr int NACC = MT * NTL, NE = M16 ? 4 : 16;
    acc_t acc[2][2][NACC];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < NACC; ++q)
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[i][j][q][e] = 0.f;
    // element e of accumulator q: row within the wave's 64 rows of a half, column within its 32 columns
    auto acc_row = [&](int q, int e) { return M16 ? (q / NTL) * 16 + fq * 4 + e : q * 32 + (e & 3) + 8 * (e >> 2) + 4 * h; };
    auto acc_col = [&](int q) { return M16 ? (q % NTL) * 16 + fr : r; };

    // the first barrier of a phase: this wave's fragment reads have returned (the partner group re-stages the buffer they came
    // from right behind it); the second ends the MFMA half.  Raw barriers: __syncthreads() would drain the DMA in flight.
#define PP_BARRIER_READS_DONE()                                           \
    do {                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                \
        __builtin_amdgcn_s_barrier();                                     \
        asm volatile("" ::: "memory");                                    \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#define PP_BARRIER()                                                      \
    do {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                \
        asm volatile("" ::: "memory");                                    \
        __builtin_amdgcn_s_barrier();                                     \
        asm volatile("" ::: "memory");                                    \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
    // (the MFMAs touch no memory: nothing but their operands orders them against the barriers, and hipcc sinks them out of
    // their phase, keeping fragment sets alive and spilling -- the empty asm on the accumulators pins the cluster on both sides)
#define PP_PIN(I, J)                                                                                               \
    do {                                                                                                           \
        if constexpr (M16)                                                                                         \
            asm volatile("" : "+v"(acc[I][J][0]), "+v"(acc[I][J][1]), "+v"(acc[I][J][2]), "+v"(acc[I][J][3]),      \
                              "+v"(acc[I][J][4]), "+v"(acc[I][J][5]), "+v"(acc[I][J][6]), "+v"(acc[I][J][7]));     \
        else                                                                                                       \
            asm volatile("" : "+v"(acc[I][J][0]), "+v"(acc[I][J][1]));                                             \
    } while (0)
#define PP_MMA(I, J)                                                                                               \
    do {                                                                                                           \
        PP_PIN(I, J);                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                             \
        if (!x_nomma) {                                                                                            \
            _Pragma("unroll") for (int s_ = 0; s_ < KS; ++s_)                                                      \
            _Pragma("unroll") for (int m_ = 0; m_ < MT; ++m_)                                                      \
            _Pragma("unroll") for (int n_ = 0; n_ < NTL; ++n_) {                                                   \
                if constexpr (M16)                                                                                 \
                    acc[I][J][m_ * NTL + n_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                            \
                        af[m_ * KS + s_], bfr[n_ * KS + s_], acc[I][J][m_ * NTL + n_], 0, 0, 0);                   \
                else                                                                                               \
                    acc[I][J][m_ * NTL + n_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                            \
                        af[m_ * KS + s_], bfr[n_ * KS + s_], acc[I][J][m_ * NTL + n_], 0, 0, 0);                   \
            }                                                                                                      \
        }                                                                                                          \
        __builtin_amdgcn_s_setprio(0);                                                                             \
        PP_PIN(I, J);                                                                                              \
    } while (0)

    // ---- prologue: all of chunk 0 and A0, B1, A1 of chunk 1 (its B0 goes out in phase 0 of chunk 0, as in the steady state)
    stage_a(0, 0);
    stage_b(0, 1, 0);
    stage_a(0, 1);
    stage_b(0, 0, 0);
    advance_a();
    stage_a(1, 0);
    stage_b(1, 1, 1);
    stage_a(1, 1);
    advance_a();
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");       // chunk 0 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                     // ... and everybody else's
    __builtin_amdgcn_sched_barrier(0);
    if (wr_u == 1) __builtin_amdgcn_s_barrier();                      // the stagger: waves 4-7 run one barrier behind
    __builtin_amdgcn_sched_barrier(0);

#ifdef LOANS_EXPERIMENT
    if (x_noread) { read_b(0, 0); read_a(0, 0); }
    x_inloop = true;
    if (x_nodma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll 1
    for (int t = 0; t < nch; ++t) {
        const int set = t & 1;
        // phase 0
        read_b(set, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(set, 0);
        stage_b(set ^ 1, 0, t + 1);
        relu_a();
        PP_BARRIER_READS_DONE();
        PP_MMA(0, 0);
        PP_BARRIER();
        // phase 1
        read_b(set, 1);
        stage_a(set, 0);
        PP_BARRIER_READS_DONE();
        PP_MMA(0, 1);
        PP_BARRIER();
        // phase 2
        read_a(set, 1);
        stage_b(set, 1, t + 2);
        relu_a();
        PP_BARRIER_READS_DONE();
        PP_MMA(1, 1);
        PP_BARRIER();
        // phase 3
        read_b(set, 0);
        stage_a(set, 1);
        advance_a();
        if (!x_nodma) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");              // everything up to phase 0's pieces: chunk t + 1 is complete
        PP_BARRIER_READS_DONE();
        PP_MMA(1, 0);
        PP_BARRIER();
    }
    if (wr_u == 0) __builtin_amdgcn_s_barrier();                      // waves 0-3 make up the barrier of the stagger
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the (masked) pieces staged beyond K: nothing may land in the staging tile
#undef PP_BARRIER_READS_DONE
#undef PP_BARRIER
#undef PP_MMA
#undef PP_PIN

    // ---- epilogue (igemm16_kernel's, for this wave layout): BN statistics from the fp32 accumulators, the tile through LDS (fp32)
    // in two passes of 128 rows = row half i -- every wave holds 64 rows of either half --, 16-byte bf16 stores
    const bool f_bias = d.flags & LOANS_F_BIAS, f_stats = d.flags & LOANS_F_STATS;
    const bool f_mask = d.flags & LOANS_F_MASK, f_add = d.flags & LOANS_F_ADDEND;
    const bool f_addmask = d.flags & LOANS_F_ADDEND_MASK;
    const bool f_bnsums = d.flags & LOANS_F_BNSUMS;
    constexpr int LDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(smem);          // [128][LDC]
    __syncthreads();
    if (f_stats) {
        // per lane: the raw sums of its elements of a column and the rows among them that exist (all of them unless this is the
        // last, ragged tile: the row table is only consulted there); lanes that hold other rows of the column are added by
        // shuffles, then the bias term (it counts rows) and one fp64 atomic pair per column
        const bool ragged = (tm + 1) * BM > a.M;
        constexpr int NCL = M16 ? NTL : 1;          // columns a lane holds per column half
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < NCL; ++c) {
                float s1 = 0.f, q2 = 0.f;
                int nvalid = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < NACC; ++q) {
                        if (M16 && q % NTL != c) continue;
#pragma unroll
                        for (int e = 0; e < NE; ++e) {
                            s1 += acc[i][j][q][e];
                            q2 += acc[i][j][q][e] * acc[i][j][q][e];
                            nvalid += ragged ? (opix[i * 128 + wr * 64 + acc_row(q, e)] != 0xFFFFFFFFu) : 1;
                        }
                    }
                float cnt = (float)nvalid;
                if constexpr (M16) {
                    s1 += __shfl_xor(s1, 16, 64); q2 += __shfl_xor(q2, 16, 64); cnt += __shfl_xor(cnt, 16, 64);
                }
                s1 += __shfl_xor(s1, 32, 64); q2 += __shfl_xor(q2, 32, 64); cnt += __shfl_xor(cnt, 32, 64);
                const int col = tn * BN + j * 128 + wc * 32 + (M16 ? c * 16 + fr : r);
                const bool cok = col < d.Cout;
                const float bv = (f_bias && cok) ? a.bias[col] : 0.f;
                q2 = q2 + 2.f * bv * s1 + cnt * bv * bv;
                s1 = s1 + cnt * bv;
                if ((M16 ? fq : h) == 0 && cok) {
                    const bool sec = a.csplit && col >= a.csplit;           // the second convolution of a pair launch
                    double* st = (sec ? a.stats2 : a.stats) + (size_t)(blockIdx.x % LOANS_STATS_REPLICAS) * 2 * a.out_c;
                    const int scol = sec ? col - a.csplit : col;
                    atomic_add_f64(st + scol, (double)s1);
                    atomic_add_f64(st + a.out_c + scol, (double)q2);
                }
            }
    }
    constexpr int CPR = BN / 8;                 // 8-channel units per row
    constexpr int RSTEP = NT / CPR;             // rows covered by the block per sweep
    constexpr int NIT = 128 / RSTEP;            // rows a thread stores per pass
    const int oc8 = tid % CPR, r0 = tid / CPR;
    const int col0 = tn * BN + oc8 * 8;
    const unsigned cbad = (col0 + 7 < d.Cout) ? 0u : 0xFFFFFFFFu;     // Cout % 8 == 0 (checked)
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
#pragma unroll
    for (int ep = 0; ep < 2; ++ep) {
        if (ep) __syncthreads();            // the previous pass has been read
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < NACC; ++q)
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    Cs[(wr * 64 + acc_row(q, e)) * LDC + j * 128 + wc * 32 + acc_col(q)] = acc[ep][j][q][e];
        __syncthreads();
        // the epilogue's operands are requested row by row behind the staging (the wave still holds the other half's accumulators)
#pragma unroll
        for (int p = 0; p < NIT; ++p) {
            const int row = r0 + p * RSTEP;
            const unsigned po = opix[ep * 128 + row];
            const unsigned off = (po + coff) | (po == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u) | cbad;
            bf16x8_t e_ref = {}, e_add = {};
            if (f_mask || f_addmask || f_bnsums) e_ref = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_ref, (int)off, 0, 0));
            if (f_add) e_add = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_add, (int)off, 0, 0));
            f32x4 lo = *reinterpret_cast<const f32x4*>(Cs + row * LDC + oc8 * 8) + b_lo;
            f32x4 hi = *reinterpret_cast<const f32x4*>(Cs + row * LDC + oc8 * 8 + 4) + b_hi;
            if (f_mask || f_addmask) {
                const f32x4 rl = cvt_lo(e_ref), rh = cvt_hi(e_ref);
                if (f_mask) { lo = keep_pos(lo, rl); hi = keep_pos(hi, rh); }
                if (f_add) {
                    f32x4 al = cvt_lo(e_add), ah = cvt_hi(e_add);
                    if (f_addmask) { al = keep_pos(al, rl); ah = keep_pos(ah, rh); }
                    lo += al; hi += ah;
                }
            } else if (f_add) {
                lo += cvt_lo(e_add); hi += cvt_hi(e_add);
            }
            bf16x8_t o;
            const bf16x4_t ol = __builtin_convertvector(lo, bf16x4_t), oh = __builtin_convertvector(hi, bf16x4_t);
            o[0] = ol[0]; o[1] = ol[1]; o[2] = ol[2]; o[3] = ol[3];
            o[4] = oh[0]; o[5] = oh[1]; o[6] = oh[2]; o[7] = oh[3];
            if (f_bnsums) {          // block-uniform; a row that does not exist loaded zeros and its gradient is zeroed below
                const f32x4 y2[2] = {cvt_lo(e_ref), cvt_hi(e_ref)};
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
}

template <bool RELU, bool M16>
int launch_igemm16pp_r(Igemm16Args& a, hipStream_t st) {
    static loans_device_once lds_limit_set;
    constexpr size_t lds = pp_lds_bytes();
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = igemm16pp_kernel<RELU, M16>;
    if (int rc_ = loans_raise_lds_limit(lds_limit_set, reinterpret_cast<const void*>(kern), lds)) return rc_;
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.d.Cout + 255) / 256;
    a.splits = 1;
    a.chunks_per_split = a.nchunks;
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
    LOANS_LAUNCH_CHECK();
    return LOANS_OK;
}

// not for LOANS_F_DENSE (the packed RGB stem: 4-byte aligned units, no tap masks) nor for raw partial tiles (split-K)
template <bool M16>
int launch_igemm16pp(Igemm16Args& a, hipStream_t st) {
    if ((a.d.flags & LOANS_F_DENSE) || a.partial) return LOANS_EINVAL;
    return (a.d.flags & LOANS_F_RELU_IN) ? launch_igemm16pp_r<true, M16>(a, st) : launch_igemm16pp_r<false, M16>(a, st);
}
