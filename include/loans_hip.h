/*
 * loans_hip.h -- C ABI of the MI355X (gfx950) kernels behind the LoANs
 * localizer -> STN crop -> assessor training hot path.
 *
 * The reference (Bartzi/loans) has NO FFI of its own: every FLOP of this path
 * is executed inside Chainer 4.1 / CuPy / cuDNN, reached through Chainer's
 * Link / Function interface.  Each entry point below therefore replaces the
 * third-party kernel behind a reference *call site* (cited per function,
 * paths relative to /root/reference).  The Python host in loans_amd/ binds
 * this header with ctypes (loans_amd/_lib.py); INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Contract for every function:
 *   - plain device pointers + explicit sizes; no torch / framework types;
 *   - returns 0 on success, a negative LOANS_E* code for rejected arguments,
 *     or a positive hipError_t from the launch;
 *   - never allocates, frees or synchronises; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream);
 *   - caller owns every buffer (workspaces included) until the stream passes;
 *   - device = current HIP device of the calling thread; re-entrant and
 *     thread-safe given distinct streams, for any number of devices per
 *     process: the only state the launchers keep is per kernel AND per device
 *     ordinal (the raised dynamic-LDS limit of a kernel, the device's CU
 *     count), held in lock-free tables (csrc/common.h).  The only runtime
 *     calls besides the launch are hipGetDevice (thread-local read) and, once
 *     per (kernel, device), hipFuncSetAttribute / hipDeviceGetAttribute --
 *     none is a stream operation and none blocks, so the first call of an
 *     entry point may sit inside a hipGraph stream capture.
 *
 * Layouts: activations NHWC float32 with C a multiple of 4 (3-channel images
 * are carried as 4 channels, the 4th zero); conv weights "OHWI"
 * [Cout][kh][kw][Cin]; Linear W [out][in].
 */
#ifndef LOANS_HIP_H
#define LOANS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOANS_OK 0
#define LOANS_EINVAL (-1)   /* bad shape / null pointer / misaligned size */
#define LOANS_ERANGE (-2)   /* size beyond what the kernel's 32-bit indexing supports */

#define LOANS_MAX_TAPS 64
#define LOANS_STATS_REPLICAS 32   /* BN-statistics accumulators are replicated to spread fp64 atomics */

/* epilogue / loader flags of the implicit-GEMM kernels */
#define LOANS_F_RELU_IN   1   /* gather relu(in) instead of in (pre-activation blocks, common/net.py:22,43,44,64,65) */
#define LOANS_F_BIAS      2   /* out += bias[n]                         (sheep/resnet.py:43 conv1) */
#define LOANS_F_STATS     4   /* stats[r][0][n] += sum_m out, stats[r][1][n] += sum_m out^2 (double), r = block % LOANS_STATS_REPLICAS */
#define LOANS_F_MASK      8   /* out = (ref[m][n] > 0) ? out : 0        (ReLU backward folded into dgrad) */
#define LOANS_F_ADDEND   16   /* out += addend[m][n]                    (residual sums; may alias out) */
#define LOANS_F_ADDEND_MASK 32 /* with ADDEND: out += (ref[m][n] > 0) ? addend[m][n] : 0  (identity shortcut through a ReLU) */
#define LOANS_F_DENSE     64   /* dense K rows for packed 3-channel images (conv1 7x7/2, sheep/resnet.py:43): `in` is a zero-padded
                                 [B][inH][inW floats] buffer (loans_prep_images_dense_f32), inW / isx / dx[] count FLOATS,
                                 tap t is the run of Cin consecutive floats at row y*isy + dy[t], float x*isx + dx[t]; w is
                                 [Cout][ntaps][Cin].  No bounds masks: every run must lie inside its row (checked). */

#define LOANS_F_OUT_BF16 128   /* loans_igemm_bf16_f32 only: `out` is a bf16 tensor (fp32 in, bf16 out: the stem conv of the bf16
                                  storage arm); not with MASK / ADDEND */
#define LOANS_F_BNSUMS  512   /* data gradients only (loans_igemm_f32 / loans_igemm_bf16s; no other epilogue flag): the launch's output g is the
                                  gradient that reaches a BatchNormalization whose own ReLU follows it (sheep/resnet.py:137,157).  `ref` is that
                                  BN's INPUT y (same shape as out), `bias` its coefficient table float[4][C] = mean, rstd, scale, shift (C = out
                                  channels) and `stats` fp64 accumulators [LOANS_STATS_REPLICAS][2][C]: the epilogue adds sum_m g m and
                                  sum_m g m (y - mean) with m = (y scale + shift > 0) -- the two sums of that BN's backward, taken from the tile
                                  while it is in registers instead of by a pass over the stored tensor (loans_bn_bwd_reduce_xmask_*); g is
                                  stored unmasked, on bf16 tensors the sums use the ROUNDED g the later passes will read.  Not with split-K /
                                  fine-tail tiles, LOANS_TILE_WS64, class launches. */
#define LOANS_F_AFFINE_IN 1024  /* (round 5) the convolution's input operand is relu(x * scale + shift) rounded to bf16: the BatchNormalization + ReLU in
                                  front of it (bn2 -> conv3 of a bottleneck, sheep/resnet.py:163-216) applied on load, with loans_bn_apply_bf16's
                                  arithmetic bit for bit, so that the activation tensor between them is never written.  Forward LOANS_TILE_PW
                                  launches of loans_igemm_bf16s only (`bias` = float[2][Cin] = scale, shift; with or without LOANS_F_STATS) and
                                  loans_wgrad_bf16s_affine_ws */
#define LOANS_F_GY_BF16  256   /* loans_wgrad_bf16_f32 only: `gy` is a bf16 tensor, x stays fp32 (the stem's weight gradient) */

/*
 * One implicit-GEMM problem:  out[m][n] = sum_{t<ntaps} sum_{c<Cin} in[pix(m,t)][c] * w[n][t][c]
 * m enumerates (b, y, x) over B x gridH x gridW;
 *   output pixel  = (y*osy + oy0, x*osx + ox0)   in an outH x outW x Cout tensor,
 *   gathered pixel = (y*isy + dy[t], x*isx + dx[t]) in an inH x inW x Cin tensor (zero outside).
 * fprop:  isy = stride, dy[t] = r - pad, os = 1.          (F.convolution_2d forward)
 * dgrad:  `in` is the output gradient, one launch per stride-parity class.
 */
typedef struct loans_igemm_desc {
    int32_t B, inH, inW, Cin;
    int32_t outH, outW, Cout;
    int32_t gridH, gridW;
    int32_t osy, osx, oy0, ox0;
    int32_t isy, isx;
    int32_t ntaps;
    int32_t flags;
    int32_t tile;               /* 0 = auto, else LOANS_TILE_* */
    int8_t dy[LOANS_MAX_TAPS];
    int8_t dx[LOANS_MAX_TAPS];
} loans_igemm_desc;

#define LOANS_TILE_128x128 1
#define LOANS_TILE_128x64  2
#define LOANS_TILE_64x64   3
#define LOANS_TILE_256x64  4
#define LOANS_TILE_64x128  5   /* wgrad only: 64 output channels x 128 tap-channel columns */
#define LOANS_TILE_SPLIT   6   /* igemm only: 128x128 tiles over the rows that fill whole machine rounds, 64x64 over the rest (two launches) */
#define LOANS_TILE_256x128 7   /* loans_igemm_bf16s only: one 512-thread block per CU, three LDS stages (chunk c + 2 in flight) */
#define LOANS_TILE_DEEP    32  /* loans_igemm_bf16s, OR-ed onto 128x128 / 128x64 / 64x64: a 4 / 5 / 8-stage LDS ring (3 / 4 / 7 chunks of K in
                                  flight per block) for grids of about one block per CU with a long K (res6 / res7 at 512 px) */
#define LOANS_TILE_256x256PP 43 /* loans_igemm_bf16s only (not LOANS_F_DENSE, no split-K): LOANS_TILE_256x256 with a ping-pong K loop -- waves 0-3 and
                                   4-7 (the two waves of every SIMD) run half a phase apart, one group on the matrix pipe while the other reads
                                   fragments and issues LDS-DMA; operand HALF-tiles re-staged the phase after their last read, counted vmcnt,
                                   raw barriers (csrc/igemm16_pp.h).  Results bit-identical to LOANS_TILE_256x256. */
#define LOANS_TILE_256x256PP16 44 /* LOANS_TILE_256x256PP on v_mfma_f32_16x16x32_bf16 (the shape the chip holds a higher clock on); 32 k values per
                                     MFMA: results agree with the other tiles to the position of rare bf16 roundings, not bit for bit */
#define LOANS_TILE_256x256 9   /* loans_igemm_bf16s only: 512 threads, eight 128 x 64 wave tiles, two LDS stages; 128 FLOP per staged byte */
#define LOANS_TILE_FINETAIL 8  /* loans_igemm_f32, forward geometry (out row = grid pixel), flags BIAS / STATS / RELU_IN / DENSE only:
                                  64x64 tiles; the tiles that share out evenly over the CUs at full K, the remaining ones (fewer than
                                  one per CU) as K-slices behind them in the same launch (raw partial tiles added with atomics to rows
                                  zeroed here, then loans_igemm_finalize_f32 over those rows).  Plain 64x64 when nothing is left over. */
#define LOANS_TILE_STEM    10   /* loans_igemm_f32 with LOANS_F_DENSE, the 7x7 / 2, Cout = 64 forward geometry, flags BIAS / STATS: direct
                                  convolution -- a block stages the input rows of R output rows and the whole weight matrix in LDS
                                  once and feeds the fp32 MFMA from that image (stem.hip); LOANS_EINVAL for frame sizes it does not
                                  cover (R * Wo must be a multiple of 64 for an R in {4, 2, 1} dividing Ho, <= 448 pixels, <= 80 KB).
                                  loans_igemm_bf16_f32 with LOANS_F_OUT_BF16 (the stem of the bf16 arm): the same direct scheme on the
                                  bf16 MFMA in a persistent kernel -- weights rounded once per block into registers, the image staged
                                  as bf16, any frame size whose 2R + 5 rows fit 78 KB next to the output slabs.
                                  loans_wgrad_f32 with LOANS_F_DENSE (flags DENSE only; `splits` ignored): conv1's weight gradient as a
                                  persistent direct kernel -- per output row the 7 input rows and the row's gradient pixels arrive by
                                  LDS-DMA (double-buffered), rows = 64 channels, columns = the 147 real window positions, every wave
                                  keeps the whole 64 x 160 tile in registers over all its units; the three window-padding columns of
                                  dw [64][7][24] are not written.  LOANS_EINVAL when two unit buffers exceed 156 KB.
                                  loans_wgrad_bf16s / _ws with LOANS_F_DENSE (bf16 frame buffer, bf16 gradient; `splits` ignored): the same
                                  scheme on the bf16 MFMA, 16 pixels per step -- both operands read with transposing LDS reads (the
                                  input rows staged as 12-byte cells, one per 16 bytes: a kernel row is one 32-column tile), the next
                                  unit staged through registers; one slab of the workspace per block (loans_wgrad_bf16s_ws_floats
                                  says how many) or, without one, atomics.  LOANS_EINVAL unless Wo % 16 == 0, Wo <= 256 and
                                  inW == 6 (Wo + 3) (even frame widths) */
#define LOANS_TILE_HALO_128    11  /* loans_igemm_bf16s, stride-1 geometries (forward k x k / 1 and its data gradient, k <= 3, Cin % 64 == 0):
                                      a block owns an 8 x 16 pixel tile x 128 output channels and stages the input halo image once
                                      per 64-channel chunk -- a tap is an LDS window shift, not a gather (csrc/halo_bf16.hip) */
#define LOANS_TILE_HALO_128x64 13  /* the same with 64 output channels per block */
#define LOANS_TILE_HALO_128x64S 14 /* 8 x 16 pixels x 64 output channels, Cin = 64, single-buffered image: four blocks per CU */
#define LOANS_TILE_WS64       15  /* 3x3 / 1, Cin = 64, Cout <= 64 (res2 and its data gradient): one persistent 512-thread block per CU, all
                                      nine taps' weights stationary in LDS, halo images double-buffered across 16 x 16 pixel tiles */
#define LOANS_TILE_HALO_256x128 36 /* 16 x 16 pixels x 128 output channels in one 512-thread block per CU (eight 64 x 64 wave tiles): twice the
                                      MFMA work per staged byte of LOANS_TILE_HALO_128 -- the N = 128 layers, too narrow for a 256-column tile */
#define LOANS_TILE_HALO_256x256 42 /* 16 x 16 pixels x 256 output channels in one 512-thread block per CU (eight 128 x 64 wave tiles, the wave layout
                                      of LOANS_TILE_256x256): the 256- and 512-channel 3 x 3 layers (res4 / res5) with their input staged once per
                                      64-channel chunk instead of once per tap -- 329 KB through the L2 -> LDS path per chunk where the implicit GEMM
                                      moves 576 KB; the fp32 staging tile of the epilogue takes two passes */
#define LOANS_TILE_WSW64      37  /* the same layers with the weights stationary and every WAVE on its own unit (2 rows x 16 pixels x 64 channels:
                                      own halo image, own vmcnt, own staging slab): no block barrier after the weights have landed */
#define LOANS_TILE_WGHALO_64   38  /* loans_wgrad_bf16s, stride-1 3 x 3 forward geometries with Cin % 64 == 0, Cout % 64 == 0: a block owns 64 output x 64
                                      input channels x ALL nine taps and walks 8 x 16 pixel tiles; gradient tile and input halo tile staged once per
                                      tile, a tap is a window shift in LDS (csrc/wgrad_halo_bf16.hip).  splits = blocks per channel-tile pair */
#define LOANS_TILE_WGHALO_128  39  /* the same with 128 output channels per block on eight waves (Cout % 128 == 0) */
#define LOANS_TILE_PW          40  /* loans_igemm_bf16s, 1 x 1 / 1 forward geometries with Cin in {64, 128}, Cout % 64 == 0, Cout <= 512, or Cin = 256,
                                      Cout % 128 == 0, Cout <= 1024; flags STATS or none (ResNet-50's res2 / res3 / res4 bottleneck expansions): a wave owns 32-pixel strips, operands go global -> VGPR
                                      in MFMA fragment layout (never through LDS), output rows leave through a per-wave LDS slab; `w` must be
                                      the weights in FRAGMENT ORDER (loans_pw_pack_bf16).  Outputs bit-identical to the other tiles
                                      (csrc/pw_bf16.hip) */
#define LOANS_TILE_HALO_256x64 12  /* 16 x 16 pixels x 64 output channels, Cin = 64 (one chunk): the res2 convolutions */
#define LOANS_TILE_SPLITK(s) ((s) << 8) /* loans_igemm_f32, OR-ed onto a tile shape, s = 2..255: split-K for small grids (few tiles, long K:
                                  the deep layers at small batch, single-image inference).  Block (tile, i) contracts every s-th
                                  part of K and ADDS its raw partial tile to `out` with fp32 atomics: the caller zero-fills `out`
                                  first (or leaves the addend in it), passes no epilogue flag here, and applies bias / statistics /
                                  mask / addend to the finished sums with loans_igemm_finalize_f32 */
#define LOANS_TILE_DMA    16   /* igemm, fp32 arm, OR-ed onto a tile shape: operand tiles staged by LDS-DMA (buffer_load ... lds)
                                  into XOR-swizzled unpadded LDS rows instead of through registers; same results bit for bit */

/* ---- convolution (replaces cuDNN ConvolutionForward / BackwardData / BackwardFilter behind
 *      L.Convolution2D: sheep/resnet.py:43,128-133,151-153 ; common/net.py:15-17,37-39,59-60) ---- */

/* fprop and dgrad. `w` is [Cout][ntaps][Cin] in the launch's tap order. Optional pointers may be NULL
 * when their flag is clear. */
int loans_igemm_f32(const float* in, const float* w, float* out,
                    const float* bias, double* stats, const float* ref, const float* addend,
                    const loans_igemm_desc* d, void* stream);

/* The stride-parity classes of ONE strided data gradient in ONE launch (conv_transpose of a k x k / s convolution is s * s
 * stride-1 correlations over interleaved output pixels -- in the reference a single F.deconvolution_2d inside
 * Convolution2DFunction.backward, reached from every strided convolution of sheep/resnet.py:115-141).  descs[0..n) are the
 * per-class descriptors loans_igemm_f32 would get one launch each; they may differ in gridH / gridW, oy0 / ox0 and the taps
 * (<= LOANS_MAX_CLS_TAPS each) and must agree in everything else; w[c] is class c's [Cout][ntaps_c][Cin] matrix (host array of
 * n device pointers, read before the call returns).  The classes' tiles share one grid -- the longest-K class should come
 * first -- and so one tail.  Flags MASK / ADDEND / ADDEND_MASK / RELU_IN; fp32 arm; descs[0].tile one of 128x128, 128x64,
 * 64x64, 256x64 (| LOANS_TILE_DMA), used for every class. */
#define LOANS_MAX_CLASSES 4
#define LOANS_MAX_CLS_TAPS 16
int loans_igemm_classes_f32(const float* in, const float* const* w, float* out, const float* ref, const float* addend,
                            const loans_igemm_desc* descs, int32_t n, void* stream);

/* Two forward convolutions of the SAME input with the same geometry (kernel, stride, padding, Cin) in ONE launch: BasicA's
 * conv1 and its strided conv shortcut (sheep/resnet.py:128-133), a bottleneck's conv1 and conv4.  `d` describes
 * convolution a (Cout = Cout_a); b differs in weights, Cout_b, output and statistics.  The second one's tiles take the
 * blocks behind the first one's, so the two share one grid and one tail.  Flags STATS / RELU_IN; fp32 arm; tile shapes
 * without LOANS_TILE_SPLIT / LOANS_TILE_SPLITK. */
int loans_igemm_pair_f32(const float* in, const float* w_a, float* out_a, double* stats_a, const float* w_b,
                         float* out_b, double* stats_b, int32_t Cout_b, const loans_igemm_desc* d, void* stream);

/* epilogue of a split-K convolution (LOANS_TILE_SPLITK) over the finished sums, in place on the whole tensor
 * out [rows][C]: (+ bias) (* (ref > 0)) (+ addend [masked by ref > 0]) and the BN statistics of the result; flags as above */
int loans_igemm_finalize_f32(float* out, const float* bias, double* stats, const float* ref, const float* addend,
                             int32_t flags, int64_t rows, int32_t C, void* stream);

/* Same problem with the operands rounded to bf16 (round-to-nearest-even, done while the fp32 tensors are staged
 * into LDS) and contracted on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; every tensor stays fp32 in memory.
 * The "bf16" arm of BASELINE configs 3 / 5 (compute dtype bf16, fp32 accumulate). */
int loans_igemm_bf16_f32(const float* in, const float* w, float* out,
                         const float* bias, double* stats, const float* ref, const float* addend,
                         const loans_igemm_desc* d, void* stream);

/* wgrad: dw[n][t][c] += sum_m gy[opix(m)][n] * x[pix(m,t)][c]   (atomic accumulation into dw).
 * The descriptor is the FORWARD conv's (Cin = x channels, Cout = gy channels; LOANS_F_RELU_IN gathers
 * relu(x)). `splits` = number of m-slices (0 = auto). */
int loans_wgrad_f32(const float* x, const float* gy, float* dw,
                    const loans_igemm_desc* d, int32_t splits, void* stream);

/* wgrad with both operands rounded to bf16 (RNE) and the bf16 MFMA; fp32 accumulation into dw. */
int loans_wgrad_bf16_f32(const float* x, const float* gy, float* dw,
                         const loans_igemm_desc* d, int32_t splits, void* stream);

/* ---- bf16 STORAGE arm (BASELINE configs 3 / 5: "bf16", "bf16 with fp32 grad accumulate").  Activations and
 *      gradients of the localizer's residual stages live in HBM as bf16 NHWC with C % 8 == 0; parameters, their
 *      gradients, BN statistics / coefficients and every accumulation stay fp32.  `void*` = bf16 tensor. ---- */

/* fprop / dgrad on bf16 tensors: in, w ([Cout][ntaps][Cin], see loans_cast_bf16 / loans_repack_dgrad_bf16), out, ref and
 * addend are bf16; bias and stats as in loans_igemm_f32 (statistics are taken from the fp32 accumulators, the output
 * is rounded to bf16 once).  Flags RELU_IN / BIAS / STATS / MASK / ADDEND / ADDEND_MASK; tiles 128x128, 128x64, 64x64,
 * 256x64, 256x128 (0 = auto).  Operand tiles are staged by LDS-DMA, contraction on v_mfma_f32_32x32x16_bf16.
 * LOANS_F_DENSE (here and in loans_wgrad_bf16s): `in` / `x` is the bf16 buffer of loans_prep_images_dense_bf16, inW / isx /
 * dx[] count bf16 ELEMENTS and must be even (the 16-byte K units are then 4-byte aligned), Cin % 8 == 0. */
int loans_igemm_bf16s(const void* in, const void* w, void* out, const float* bias, double* stats,
                      const void* ref, const void* addend, const loans_igemm_desc* d, void* stream);
/* The weights of a 1 x 1 convolution, [Cout][Cin] bf16, in the order LOANS_TILE_PW reads them -- one 1 KiB block per (32 output
 * channels, 16 input channels) MFMA B fragment: packed[((nt * Cin / 16 + ks) * 64 + lane) * 8 + j] =
 * w[nt * 32 + lane % 32][ks * 16 + lane / 32 * 8 + j].  Cout % 32 == 0, Cin % 16 == 0; `packed` holds Cout x Cin bf16. */
int loans_pw_pack_bf16(const void* w, void* packed, int32_t Cout, int32_t Cin, void* stream);
/* The same for every LOANS_TILE_PW layer of a step in ONE launch, from the fp32 master weights (rounded to nearest even): job j packs
 * src [Cout][Cin] fp32 into dst (Cout x Cin bf16, fragment order); first_unit = the sum of Cout x Cin / 8 over the jobs before it,
 * total_units that sum over all jobs.  `jobs_dev` is a table in DEVICE memory, caller-owned, alive until the launch has run. */
typedef struct loans_pw_pack_job {
    const void* src;
    void* dst;
    int32_t Cout, Cin;
    int32_t first_unit, reserved;
} loans_pw_pack_job;
int loans_pw_pack_batch_f32(const loans_pw_pack_job* jobs_dev, int32_t njobs, int32_t total_units, void* stream);
/* Two forward convolutions of the SAME bf16 input with the same geometry AND the same channel count (BasicA's conv1 and its
 * strided conv shortcut, sheep/resnet.py:128-133) as ONE GEMM with 2 x Cout columns: w_ab = [2][Cout][ntaps][Cin] (a's
 * matrix, then b's), out_ab = [2][B][outH][outW][Cout] (two ordinary tensors back to back).  Unlike loans_igemm_pair_f32
 * the two share the staged input tile, not just the grid: at N = 2 x 128 the 256-column tiles apply.  `d` describes
 * convolution a; flags STATS / RELU_IN; Cout % 32 == 0; the implicit-GEMM tiles only. */
int loans_igemm_pair_bf16s(const void* in, const void* w_ab, void* out_ab, double* stats_a, double* stats_b,
                           const loans_igemm_desc* d, void* stream);
/* wgrad with bf16 x and gy, fp32 atomic accumulation into dw ("fp32 grad accumulate"); the pixel-major tiles are staged as
 * they lie and transposed by the fragment reads (ds_read_b64_tr_b16).  Tiles 128x128, 64x64, 64x128, and 256x256 (512 threads, one
 * block per CU) for layers with >= 256 output channels. */
int loans_wgrad_bf16s(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, int32_t splits, void* stream);
/* The same gradient WITHOUT atomics (round 5; replaces the backward-filter of F.convolution_2d at sheep/resnet.py:121-160 like the
 * call above): every block STORES its raw partial tile into slab s = its pixel slice of `ws` ([slabs][Cout][ntaps * Cin] floats, the
 * layout of dw; caller-owned, at least loans_wgrad_bf16s_ws_floats(d, splits) floats, no need to clear it), then
 * loans_fold_slabs_f32 adds the slabs to dw in a FIXED order on the same stream: two runs give bit-identical gradients, and a
 * launch closes with plain 128-byte-row stores plus one streaming pass instead of one fp32 atomic per partial sum (the halo form
 * ends with 18.9 M of them whatever the layer).  dw must not be written by another stream between the two launches. */
int loans_wgrad_bf16s_ws(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, int32_t splits,
                         float* ws, int64_t ws_floats, void* stream);
/* (round 5) the same with LOANS_F_AFFINE_IN in d->flags (1 x 1 / 1 convolutions, the GEMM tiles): `x` is the INPUT of the
 * BatchNormalization in front of the convolution and `affine` its float[2][Cin] = scale, shift; the kernel contracts gy with
 * relu(x * scale + shift) rounded to bf16 -- the tensor loans_bn_apply_bf16 would have written, bit for bit */
int loans_wgrad_bf16s_affine_ws(const void* x, const void* gy, float* dw, const loans_igemm_desc* d, int32_t splits,
                                float* ws, int64_t ws_floats, const float* affine, void* stream);
/* floats of workspace that call takes for (d, splits) -- host arithmetic only, nothing is launched; < 0: a LOANS_E* code */
int64_t loans_wgrad_bf16s_ws_floats(const loans_igemm_desc* d, int32_t splits);
/* dst[i] += ws[0][i] + ws[1][i] + ... + ws[slabs - 1][i] for i < n (n % 4 == 0, 16-byte aligned pointers): every output is summed
 * by ONE thread group in an order that depends on (n, slabs) only -- deterministic, no atomics. */
int loans_fold_slabs_f32(const float* ws, float* dst, int64_t n, int32_t slabs, void* stream);
/* Split-K on bf16 storage for grids that cannot fill the machine (res6 / res7 at 512 px: 2048 - 8192 rows, K = 4608): block
 * (tile, s) contracts every `splits`-th slice of K and ADDS its raw fp32 tile to `partial` [B * outH * outW][Cout] (zeroed by the
 * caller; the descriptor's flags may only carry RELU_IN / DENSE; the implicit-GEMM tiles only); the parity classes of a strided
 * data gradient add into the same buffer.  loans_igemm_finalize_bf16 then writes the bf16 tensor of the finished sums with the
 * epilogue flags BIAS / STATS / MASK / ADDEND / ADDEND_MASK of loans_igemm_bf16s (Cout / 8 must divide 256). */
int loans_igemm_bf16s_splitk(const void* in, const void* w, float* partial, const loans_igemm_desc* d, int32_t splits, void* stream);
int loans_igemm_finalize_bf16(const float* partial, void* out, const float* bias, double* stats, const void* ref,
                              const void* addend, int32_t flags, int64_t rows, int32_t Cout, void* stream);
/* fp32 master weights -> bf16 operand copies: plain cast (n % 4 == 0), and the dgrad re-pack with the cast folded in */
int loans_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
int loans_repack_dgrad_bf16(const float* src, void* dst, int32_t Cout, int32_t Cin, int32_t src_taps,
                            const int32_t* tapsel_host, int32_t ntaps, void* stream);

/* All data-gradient weight re-packs of a step in ONE launch (csrc/weightprep.hip).  `jobs_dev`: a table of `njobs` jobs in DEVICE
 * memory, caller-owned and kept alive while launches that read it are in flight (built once: sources are parameter views of a
 * fixed arena, destinations persistent buffers).  Job i covers the blocks [first_tile, first_tile + tiles_co * tiles_ci * ntaps)
 * with tiles_co = ceil(Cout / 64), tiles_ci = ceil(Cin / 64); first_tile ascending from 0, `total_tiles` = their sum.  Each job
 * is one loans_repack_dgrad_f32 (dst_bf16 = 0) or loans_repack_dgrad_bf16 (dst_bf16 = 1) call with at most
 * LOANS_REPACK_JOB_TAPS selected taps.  Replaces the per-class launches in front of every F.convolution_2d backward-data
 * (sheep/resnet.py:121-160, common/net.py:15-65). */
#define LOANS_REPACK_JOB_TAPS 16
typedef struct loans_repack_job {
    const void* src;            /* fp32 [Cout][src_taps][Cin] */
    void* dst;                  /* [Cin][ntaps][Cout], fp32 or bf16 */
    int32_t Cout, Cin, src_taps, ntaps;
    int32_t tapsel[LOANS_REPACK_JOB_TAPS];
    int32_t first_tile, tiles_co, tiles_ci, dst_bf16;
} loans_repack_job;
int loans_repack_dgrad_batch(const loans_repack_job* jobs_dev, int32_t njobs, int32_t total_tiles, void* stream);

/* dgrad for convolutions whose input has 4 physical channels (the RGB crops, common/net.py:15,17):
 * out[opix(m)][0..3] = sum_t sum_co gy[pix(m,t)][co] * w_ohwi[co][tapsel[t]][0..3]; same descriptor as
 * loans_igemm_f32 (Cin = gy channels, Cout must be 4), flags MASK / ADDEND only. */
int loans_dgrad_c4_f32(const float* gy, const float* w_ohwi, float* out, const float* ref, const float* addend,
                       const loans_igemm_desc* d, const int32_t* tapsel_host, int32_t src_taps, void* stream);

/* same with a bf16 gradient tensor (the assessor's first block in the bf16-storage arm); w, out, ref, addend fp32 */
int loans_dgrad_c4_bf16_f32(const void* gy, const float* w_ohwi, float* out, const float* ref, const float* addend,
                            const loans_igemm_desc* d, const int32_t* tapsel_host, int32_t src_taps, void* stream);

/* The crop gradient as ONE launch (csrc/cropgrad.hip): gx[B][H][W][4] = dgrad_a(gy_a, w_a) + dgrad_b(gy_b, w_b) (+ addend),
 * the two convolutions of DownResBlock1 that read the 4-channel crops (common/net.py:15,17: c0 3x3 / 1, cs 4x4 / 2; the
 * reference's backward runs cuDNN's dgrad twice and adds).  Each gradient tensor [B][outH][outW][C] is read once (taps in
 * the GEMM's N, col2im inside the block); w_* are the FORWARD weights, OHWI [C][k][k][4]; channel 3 of gx is written 0.
 * gy_b / w_b / cb may be NULL (one convolution).  k <= 4, stride <= 2, pad < k, C == 128 (the assessor's width); outH / outW
 * must be the forward output size of an H x W input; each gradient tensor below 2 GiB.  `addend` may alias `out`.
 * `wpack`: caller-owned workspace of LOANS_CROP_WPACK_FLOATS floats (the weights in MFMA fragment order, written by a
 * pre-pass of the same call on the same stream). */
#define LOANS_CROP_WPACK_FLOATS 16384
typedef struct loans_small_conv {
    int32_t k, stride, pad, outH, outW;
} loans_small_conv;
int loans_crop_dgrad_f32(const float* gy_a, const float* w_a, const loans_small_conv* ca, const float* gy_b, const float* w_b,
                         const loans_small_conv* cb, float* out, const float* addend, float* wpack, int32_t B, int32_t H,
                         int32_t W, int32_t C, void* stream);
/* same with bf16 gradient tensors (bf16-storage arm); weights, addend and gx fp32.  The contraction runs on bf16 MFMAs with fp32
 * accumulation: the call's packing pre-pass rounds the weights to bf16 (RNE) into `wpack` (round 6; on fp32 MFMAs the kernel was
 * bound by the matrix pipe: 56 GFLOP at B = 256). */
int loans_crop_dgrad_bf16_f32(const void* gy_a, const float* w_a, const loans_small_conv* ca, const void* gy_b, const float* w_b,
                              const loans_small_conv* cb, float* out, const float* addend, float* wpack, int32_t B, int32_t H,
                              int32_t W, int32_t C, void* stream);

/* weight repack for dgrad: dst[ci][t][co] = src[co][tapsel[t]][ci]  (src is OHWI with `src_taps` taps) */
int loans_repack_dgrad_f32(const float* src, float* dst, int32_t Cout, int32_t Cin, int32_t src_taps,
                           const int32_t* tapsel_host, int32_t ntaps, void* stream);

/* ---- preprocessing (replaces the per-image PIL round trip of resnet.prepare,
 *      sheep/sheep_localizer.py:45,72-82): NCHW RGB [0,1] -> trunc_u8(x*255) -> BGR - mean -> NHWC4 ---- */
int loans_prep_images_f32(const float* images_nchw, float* out_nhwc4, int32_t B, int32_t H, int32_t W, void* stream);
/* same arithmetic, written as packed 3-channel rows into a zero-padded [B][Hp][Wp][3] buffer with the image at
 * (pad, pad): the LOANS_F_DENSE input of conv1.  Every element of the buffer is written. */
int loans_prep_images_dense_f32(const float* images_nchw, float* out_padded, int32_t B, int32_t H, int32_t W,
                                int32_t pad, int32_t Hp, int32_t Wp, void* stream);
/* the same buffer in bf16 (rounded to nearest even, exactly what the bf16 arm's operand staging does to the fp32 one):
 * the LOANS_F_DENSE input of loans_igemm_bf16s / loans_wgrad_bf16s (bf16 storage arm, BASELINE configs 3 / 5) */
int loans_prep_images_dense_bf16(const float* images_nchw, void* out_padded, int32_t B, int32_t H, int32_t W,
                                 int32_t pad, int32_t Hp, int32_t Wp, void* stream);
/* NCHW (C=3) -> NHWC4 without arithmetic (the assessor's `real` batch, sheep_updater.py:32-35) */
int loans_nchw3_to_nhwc4_f32(const float* in, float* out, int32_t B, int32_t H, int32_t W, void* stream);

/* ---- batch normalisation (replaces cuDNN BatchNormalizationForwardTraining/Backward behind
 *      L.BatchNormalization, sheep/resnet.py:44,129-134,152-154; eps 2e-5, decay 0.9) ---- */

/* stats (double [LOANS_STATS_REPLICAS][2][C]: sum, sum of squares over `count` rows) -> mean, rstd, scale=gamma*rstd,
 * shift=beta-mean*scale; updates running stats in place (unbiased var, + eps if eps_in_running_var). */
int loans_bn_finalize_f32(const double* stats, int32_t C, int64_t count, float eps, float decay,
                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                          int32_t eps_in_running_var,
                          float* mean, float* rstd, float* scale, float* shift, void* stream);
/* test-mode coefficients from the running statistics */
int loans_bn_eval_coeffs_f32(int32_t C, float eps, const float* gamma, const float* beta,
                             const float* running_mean, const float* running_var,
                             float* mean, float* rstd, float* scale, float* shift, void* stream);

/* y = act(x*scale+shift [+ r | + x2*scale2+shift2]);  mode 0: no second term, 1: + r, 2: + bn(x2). relu!=0 -> ReLU */
int loans_bn_apply_f32(const float* x, const float* scale, const float* shift,
                       const float* x2, const float* scale2, const float* shift2,
                       float* y, int64_t rows, int32_t C, int32_t mode, int32_t relu, void* stream);

/* stem: y = max_pool_3x3_s2_cover_all(relu(x*scale+shift)); idx = argmax position 0..8 (first maximum)
 * (sheep/resnet.py:72-73). */
int loans_bn_relu_maxpool_f32(const float* x, const float* scale, const float* shift, float* y, uint8_t* idx,
                              int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
/* (round 5) the same, and xsel[pooled element] = the RAW x at its argmax ([B][OH][OW][C], the type of x): the sums of the stem's
 * BN backward are then loans_bn_bwd_reduce_rep_* (mask kind 2) over (gy, xsel) -- two contiguous pooled-size tensors -- instead of
 * loans_pool_bn_bwd_reduce_*'s gather out of the four times larger x through idx (BackwardData of F.max_pooling_2d +
 * F.batch_normalization, sheep/resnet.py:72-73).  LOANS_EINVAL for channel counts the 16-byte-unit kernel does not tile. */
int loans_bn_relu_maxpool_sel_f32(const float* x, const float* scale, const float* shift, float* y, uint8_t* idx, float* xsel,
                                  int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
int loans_bn_relu_maxpool_sel_bf16(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx, void* xsel,
                                   int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
/* gx = (sum over windows whose argmax is this pixel of gy) * (x*scale+shift > 0) */
int loans_maxpool_relu_bwd_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale,
                               const float* shift, float* gx,
                               int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);

/* the same three passes on bf16 tensors (x, x2, y / gy, gx, idx as above; coefficients fp32) */
int loans_bn_apply_bf16(const void* x, const float* scale, const float* shift,
                        const void* x2, const float* scale2, const float* shift2,
                        void* y, int64_t rows, int32_t C, int32_t mode, int32_t relu, void* stream);
int loans_bn_relu_maxpool_bf16(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx,
                               int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
int loans_maxpool_relu_bwd_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale,
                                const float* shift, void* gx,
                                int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);

/* backward reductions: sums[0][c] += sum g, sums[1][c] += sum g*xhat  with g = gy*(mask>0) (mask may be
 * NULL), xhat = (x-mean)*rstd. If x2 != NULL also sums[2], sums[3] for (x2, mean2, rstd2). */
int loans_bn_bwd_reduce_f32(const float* gy, const float* mask, const float* x, const float* mean,
                            const float* rstd, const float* x2, const float* mean2, const float* rstd2,
                            double* sums, int64_t rows, int32_t C, void* stream);
/* from the sums: ggamma += , gbeta += , and the coefficients of gx = k1*g + k2*x + k3 */
int loans_bn_bwd_coeffs_f32(const double* sums, int32_t C, int64_t count, const float* gamma,
                            const float* mean, const float* rstd, float* ggamma, float* gbeta,
                            float* k1, float* k2, float* k3, void* stream);
/* BN-backward reduction into REPLICATED accumulators (round 3; replaces loans_bn_bwd_reduce_* / _bits_* / _xmask_* where the
 * channel count allows it: C / V a divisor of 256, V = 4 fp32 / 8 bf16 channels = one 16-byte unit): one entry per storage type,
 * mask_kind 0 = none, 1 = g (mask > 0), 2 = g (x scale + shift > 0), 3 = sign bits.  `sums` fp64 [replicas][2][C], dual (x2 given)
 * [replicas][4][C] = [sum g | sum g xhat | sum g | sum g xhat2], zeroed by the caller; blocks add into replica block % replicas
 * (thousands of blocks on 2 C addresses cost more than the pass).  F.batch_normalization's backward, sheep/resnet.py:129-134. */
int loans_bn_bwd_reduce_rep_f32(const float* gy, const void* mask, int32_t mask_kind, const float* x, const float* mean,
                                const float* rstd, const float* x2, const float* mean2, const float* rstd2, const float* scale,
                                const float* shift, double* sums, int32_t replicas, int64_t rows, int32_t C, void* stream);
int loans_bn_bwd_reduce_rep_bf16(const void* gy, const void* mask, int32_t mask_kind, const void* x, const float* mean,
                                 const float* rstd, const void* x2, const float* mean2, const float* rstd2, const float* scale,
                                 const float* shift, double* sums, int32_t replicas, int64_t rows, int32_t C, void* stream);
/* the stem tail's reduction (loans_pool_bn_bwd_reduce_*) into replicated accumulators [replicas][2][C] (round 3) */
int loans_pool_bn_bwd_reduce_rep_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale, const float* shift,
                                     const float* mean, const float* rstd, double* sums, int32_t replicas, int32_t B, int32_t H,
                                     int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
int loans_pool_bn_bwd_reduce_rep_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale, const float* shift,
                                      const float* mean, const float* rstd, double* sums, int32_t replicas, int32_t B, int32_t H,
                                      int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
/* loans_pool_bn_bwd_apply_* with the bias-gradient sums (gxsum) going into `replicas` accumulators gxsum_rep[replicas][C], zeroed by
 * the caller and folded into the bias gradient by loans_fold_replicas_f32 (dst[c] += sum_r src[r][c]): every block closes with C float
 * atomics, and on ONE set of addresses they made the pass slower the more blocks it had.  C / 4 must divide 256.  (round 3) */
int loans_pool_bn_bwd_apply_rep_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale, const float* shift,
                                    const float* k1, const float* k2, const float* k3, float* gx, float* gxsum_rep, int32_t replicas,
                                    int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
int loans_pool_bn_bwd_apply_rep_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale, const float* shift,
                                     const float* k1, const float* k2, const float* k3, void* gx, float* gxsum_rep, int32_t replicas,
                                     int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
int loans_fold_replicas_f32(const float* src, float* dst, int32_t replicas, int32_t C, void* stream);
/* loans_bn_bwd_coeffs_f32 from `replicas` (<= 32) accumulators `rep_stride` doubles apart, each [sum g | second sum][C]; centred = 1:
 * the second sum is sum g (y - mean) as a data gradient's epilogue took it (LOANS_F_BNSUMS), centred = 0: it is sum g xhat */
int loans_bn_bwd_coeffs_rep_f32(const double* sums, int32_t replicas, int32_t rep_stride, int32_t centred, int32_t C, int64_t count,
                                const float* gamma, const float* mean, const float* rstd, float* ggamma, float* gbeta, float* k1,
                                float* k2, float* k3, void* stream);
/* gx = k1*g + k2*x + k3, g = gy*(mask>0); optional second output for (x2, k1b, k2b, k3b) */
int loans_bn_bwd_apply_f32(const float* gy, const float* mask, const float* x,
                           const float* k1, const float* k2, const float* k3, float* gx,
                           const float* x2, const float* k1b, const float* k2b, const float* k3b, float* gx2,
                           int64_t rows, int32_t C, void* stream);

/* ReLU mask as sign bits: loans_bn_apply_bits_* is loans_bn_apply_* that also writes signbits[rows * C / 4], one byte per four
 * channels, bit e = (y[4i + e] > 0); loans_bn_bwd_{reduce,apply}_bits_* are the backward passes taking those bits where the
 * plain ones take the mask tensor -- the final BN of a residual unit, whose mask is the unit's output relu(bn(x) + shortcut)
 * (sheep/resnet.py:141,160): 1/16 (fp32) or 1/8 (bf16) of the activation's bytes in both passes. */
int loans_bn_apply_bits_f32(const float* x, const float* scale, const float* shift,
                            const float* x2, const float* scale2, const float* shift2,
                            float* y, uint8_t* signbits, int64_t rows, int32_t C, int32_t mode, int32_t relu, void* stream);
int loans_bn_apply_bits_bf16(const void* x, const float* scale, const float* shift,
                             const void* x2, const float* scale2, const float* shift2,
                             void* y, uint8_t* signbits, int64_t rows, int32_t C, int32_t mode, int32_t relu, void* stream);
int loans_bn_bwd_reduce_bits_f32(const float* gy, const uint8_t* signbits, const float* x, const float* mean,
                                 const float* rstd, const float* x2, const float* mean2, const float* rstd2,
                                 double* sums, int64_t rows, int32_t C, void* stream);
int loans_bn_bwd_reduce_bits_bf16(const void* gy, const uint8_t* signbits, const void* x, const float* mean,
                                  const float* rstd, const void* x2, const float* mean2, const float* rstd2,
                                  double* sums, int64_t rows, int32_t C, void* stream);
int loans_bn_bwd_apply_bits_f32(const float* gy, const uint8_t* signbits, const float* x,
                                const float* k1, const float* k2, const float* k3, float* gx,
                                const float* x2, const float* k1b, const float* k2b, const float* k3b, float* gx2,
                                int64_t rows, int32_t C, void* stream);
int loans_bn_bwd_apply_bits_bf16(const void* gy, const uint8_t* signbits, const void* x,
                                 const float* k1, const float* k2, const float* k3, void* gx,
                                 const void* x2, const float* k1b, const float* k2b, const float* k3b, void* gx2,
                                 int64_t rows, int32_t C, void* stream);

/* The same two passes for a BN whose own ReLU supplies the mask (relu(bn(x)), the inner BNs of a residual unit,
 * sheep/resnet.py:137,157): g = gy * (x*scale+shift > 0) with the pre-activation recomputed from x -- no mask tensor. */
int loans_bn_bwd_reduce_xmask_f32(const float* gy, const float* x, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, double* sums, int64_t rows, int32_t C, void* stream);
int loans_bn_bwd_apply_xmask_f32(const float* gy, const float* x, const float* scale, const float* shift,
                                 const float* k1, const float* k2, const float* k3, float* gx, int64_t rows, int32_t C,
                                 void* stream);
int loans_bn_bwd_reduce_xmask_bf16(const void* gy, const void* x, const float* scale, const float* shift,
                                   const float* mean, const float* rstd, double* sums, int64_t rows, int32_t C, void* stream);
int loans_bn_bwd_apply_xmask_bf16(const void* gy, const void* x, const float* scale, const float* shift,
                                  const float* k1, const float* k2, const float* k3, void* gx, int64_t rows, int32_t C,
                                  void* stream);

/* The stem's tail fused (sheep/resnet.py:72-73 backwards: max_pooling_2d -> relu -> bn1): with
 * g = (sum over windows whose argmax is this pixel of gy) * (x*scale+shift > 0) never written to memory,
 * reduce: sums[0][c] += sum g, sums[1][c] += sum g*xhat (one gather of x per pooled element);
 * apply:  gx = k1*g + k2*x + k3 (k from loans_bn_bwd_coeffs_f32); gxsum (may be NULL; needs C/4 to divide 256):
 *         gxsum[c] += sum of gx over pixels = the bias gradient of the convolution in front of the BN.
 * gy, idx: [B][OH][OW][C]; x, gx: [B][H][W][C]. */
int loans_pool_bn_bwd_reduce_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale,
                                 const float* shift, const float* mean, const float* rstd, double* sums,
                                 int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
int loans_pool_bn_bwd_apply_f32(const float* gy, const uint8_t* idx, const float* x, const float* scale,
                                const float* shift, const float* k1, const float* k2, const float* k3, float* gx,
                                float* gxsum, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
int loans_pool_bn_bwd_reduce_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale,
                                  const float* shift, const float* mean, const float* rstd, double* sums,
                                  int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);
int loans_pool_bn_bwd_apply_bf16(const void* gy, const uint8_t* idx, const void* x, const float* scale,
                                 const float* shift, const float* k1, const float* k2, const float* k3, void* gx,
                                 float* gxsum, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, void* stream);

/* bf16 tensors (gy, mask, x, x2, gx, gx2), fp32 sums and coefficients */
int loans_bn_bwd_reduce_bf16(const void* gy, const void* mask, const void* x, const float* mean,
                             const float* rstd, const void* x2, const float* mean2, const float* rstd2,
                             double* sums, int64_t rows, int32_t C, void* stream);
int loans_bn_bwd_apply_bf16(const void* gy, const void* mask, const void* x,
                            const float* k1, const float* k2, const float* k3, void* gx,
                            const void* x2, const float* k1b, const float* k2b, const float* k3b, void* gx2,
                            int64_t rows, int32_t C, void* stream);

/* ---- small dense ops ---- */
/* out[c] += sum_rows x[row][c]   (conv bias gradient) */
int loans_colsum_f32(const float* x, float* out, int64_t rows, int32_t C, void* stream);
int loans_colsum_bf16(const void* x, float* out, int64_t rows, int32_t C, void* stream);
/* _global_average_pooling_2d (sheep_localizer.py:58): x [B][HW][C] -> y [B][C] ; backward broadcast */
int loans_gap_fwd_f32(const float* x, float* y, int32_t B, int32_t HW, int32_t C, void* stream);
int loans_gap_bwd_f32(const float* gy, float* gx, int32_t B, int32_t HW, int32_t C, void* stream);
/* the boundary of the bf16 region: pooled features leave it as fp32, their gradient enters it as bf16 */
int loans_gap_fwd_bf16_f32(const void* x, float* y, int32_t B, int32_t HW, int32_t C, void* stream);
int loans_gap_bwd_f32_bf16(const float* gy, void* gx, int32_t B, int32_t HW, int32_t C, void* stream);
/* L.Linear (sheep_localizer.py:60 ; common/net.py:81,90). act_in: 1 = relu(x) on load; act_out: 1 = sigmoid */
int loans_linear_fwd_f32(const float* x, const float* W, const float* b, float* y,
                         int32_t B, int32_t K, int32_t N, int32_t act_in, int32_t act_out, void* stream);
/* gz = gy (act_out 0) or gy*y*(1-y) (act_out 1). gx[b][k] = sum_n gz W (times x>0 if act_in);
 * gW[n][k] += sum_b gz relu?(x); gb[n] += sum_b gz. gx / gW / gb may be NULL. */
int loans_linear_bwd_f32(const float* x, const float* W, const float* y, const float* gy,
                         float* gx, float* gW, float* gb,
                         int32_t B, int32_t K, int32_t N, int32_t act_in, int32_t act_out, void* stream);
/* the same layer with x (and gx) as bf16 tensors; W, y, gy, gW, gb fp32 */
int loans_linear_fwd_bf16(const void* x, const float* W, const float* b, float* y,
                          int32_t B, int32_t K, int32_t N, int32_t act_in, int32_t act_out, void* stream);
int loans_linear_bwd_bf16(const void* x, const float* W, const float* y, const float* gy,
                          void* gx, float* gW, float* gb,
                          int32_t B, int32_t K, int32_t N, int32_t act_in, int32_t act_out, void* stream);
/* y = x * mask (elementwise, n floats) -- RotationDropout forward/backward (functions/rotation_droput.py:26-48) */
int loans_mul_f32(const float* x, const float* mask, float* y, int64_t n, void* stream);
/* y = a*x + b*y */
int loans_axpby_f32(float a, const float* x, float b, float* y, int64_t n, void* stream);

/* ---- spatial transformer (replaces cuDNN SpatialTf*; sheep_localizer.py:62-63) ---- */
/* grid [B][2][th][tw] (channel 0 = x) from theta [B][2][3] */
int loans_st_grid_fwd_f32(const float* theta, float* grid, int32_t B, int32_t th, int32_t tw, void* stream);
int loans_st_grid_bwd_f32(const float* ggrid, float* gtheta, int32_t B, int32_t th, int32_t tw, void* stream);
/* bilinear, align-corners, one-pixel zero border. images NCHW (C=3) ; rois NHWC4 [B][th][tw][4] */
int loans_st_sampler_fwd_f32(const float* images_nchw, const float* grid, float* rois_nhwc4,
                             int32_t B, int32_t H, int32_t W, int32_t th, int32_t tw, void* stream);
/* ggrid [B][2][th][tw] (+= if accumulate) from grois NHWC4 */
int loans_st_sampler_bwd_grid_f32(const float* images_nchw, const float* grid, const float* grois_nhwc4,
                                  float* ggrid, int32_t accumulate,
                                  int32_t B, int32_t H, int32_t W, int32_t th, int32_t tw, void* stream);

/* ---- losses (sheep_updater.py:42-46,60 ; common/utils.py:142-178,301-316) ---- */
/* loss[0] = mean((y-t)^2); t = target[i] or the constant `tconst` when target == NULL */
int loans_mse_fwd_f32(const float* y, const float* target, float tconst, float* loss, int32_t n, void* stream);
/* gy[i] = gloss[0] * 2/n * (y-t) */
int loans_mse_bwd_f32(const float* y, const float* target, float tconst, const float* gloss, float* gy,
                      int32_t n, void* stream);
/* kind 0 = DirectionLoss (needs imgH,imgW), 1 = OutOfImageLoss * oob_scale. grid [B][2][th][tw]. */
int loans_grid_loss_fwd_f32(const float* grid, float* loss, int32_t kind, float imgH, float imgW, float oob_scale,
                            int32_t B, int32_t th, int32_t tw, void* stream);
/* ggrid += gloss[0] * dloss/dgrid (touches only the three corner points) */
int loans_grid_loss_bwd_f32(const float* grid, const float* gloss, float* ggrid, int32_t kind,
                            float imgH, float imgW, float oob_scale,
                            int32_t B, int32_t th, int32_t tw, void* stream);

/* ---- VisualBackprop and grayscale rois (insights/visual_backprop.py:16-53; sheep/sheep_localizer.py:65-68,105-108) ---- */
/* out[row] = (1 / cdiv) * sum_c a(x[row][c]); a = identity, or relu(x * scale[c] + shift[c]) when scale / shift are given
 * (F.average(input, axis=1) of a convolution / pooling node's input, visual_backprop.py:39); C % 4 == 0; cdiv = the logical
 * channel count (3 for the 4-channel padded frames) */
int loans_channel_mean_f32(const float* x, const float* scale, const float* shift, float* out, int64_t rows, int32_t C,
                           int32_t cdiv, void* stream);
int loans_channel_mean_bf16(const void* x, const float* scale, const float* shift, float* out, int64_t rows, int32_t C,
                            int32_t cdiv, void* stream);
/* out = deconvolution_2d(feat [B][fh][fw], ones(kh, kw), stride, pad, outsize (H, W)) * avg [B][H][W]   (visual_backprop.py:31-40) */
int loans_vbp_scale_f32(const float* feat, const float* avg, float* out, int32_t B, int32_t fh, int32_t fw, int32_t H, int32_t W,
                        int32_t kh, int32_t kw, int32_t sy, int32_t sx, int32_t ph, int32_t pw, void* stream);
/* per image of n values, in place: (x - min) / (max - min)   (visual_backprop.py:48-52) */
int loans_minmax_normalize_f32(float* x, int32_t B, int32_t n, void* stream);
/* rois NHWC4 -> 0.299 * ch2 + 0.587 * ch1 + 0.114 * ch0 per pixel, and its gradient (sheep_localizer.py:65-68) */
int loans_gray_fwd_f32(const float* rois_nhwc4, float* out, int64_t npix, void* stream);
int loans_gray_bwd_f32(const float* g, float* grois_nhwc4, int64_t npix, void* stream);

/* ---- the imgaug branch of the input pipeline (common/datasets/image_dataset.py:57-70,80-83), one position of the sampled
 * operation order per launch: params_dev [B][8] int32, [0] = op (0 copy, 1 Fliplr, 2 AddToHueAndSaturation: [1] dh [2] ds,
 * 3 CropAndPad: [1..4] top, right, bottom, left pixels (< 0 crop, > 0 pad), [5] fill 0 constant / 1 edge, resized back to H x W);
 * uint8 HWC RGB frames of one size, in != out.  Integer arithmetic throughout (loans_amd/common/datasets/augment.py is the
 * NumPy form of the same operations). */
int loans_augment_stage_u8(const void* in, void* out, int32_t B, int32_t H, int32_t W, const int32_t* params_dev, void* stream);

/* ---- optimiser: chainer.optimizers.Adam(alpha, amsgrad=True) over one flat buffer
 *      (train_sheep_localizer.py:130-134; sheep_updater.py:52,66). lr_t = alpha*sqrt(1-b2^t)/(1-b1^t)
 *      is computed by the caller; eps sits outside the bias correction. grad_scale multiplies g first
 *      (1/world_size after a sum all-reduce). ---- */
int loans_adam_amsgrad_f32(float* p, const float* g, float* m, float* v, float* vhat, int64_t n,
                           double lr_t, double beta1, double beta2, double eps, double eta, double weight_decay_rate,
                           double grad_scale, void* stream);
/* same update with lr_t read from device memory at execution time (one float): the launch is then independent of the
 * step count and can be captured once in a hipGraph; the caller refreshes *lr_t_dev before each replay */
int loans_adam_amsgrad_devlr_f32(float* p, const float* g, float* m, float* v, float* vhat, int64_t n,
                                 const float* lr_t_dev, double beta1, double beta2, double eps, double eta,
                                 double weight_decay_rate, double grad_scale, void* stream);

/* plain Adam -- chainer.optimizers.Adam's default amsgrad=False (train_sheep_localizer.py:130 passes amsgrad=True; the class
 * surface takes both): the same update with sqrt(v) in the denominator, no vhat.  lr_t_dev != NULL: the rate is read from
 * device memory (hipGraph), lr_t is ignored */
int loans_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr_t, const float* lr_t_dev,
                   double beta1, double beta2, double eps, double eta, double weight_decay_rate, double grad_scale,
                   void* stream);

/* ---- input contract on the GPU (common/datasets/image_dataset.py:16-28 `resize_image` -> Pillow
 *      Image.resize(LANCZOS); :98 `image / 255`).  Bit-exact restatement of Pillow's 8-bit two-pass resampler
 *      (libImaging/Resample.c): per output coordinate a window bounds[2*i] = first input index, bounds[2*i+1] = taps, and
 *      ks int32 coefficients in 22-bit fixed point (tables from loans_amd/common/datasets/resample.py); horizontal pass
 *      into tmp [B][inH][outW][3], vertical pass into dst.  src is [B][inH][inW][3] uint8 RGB.
 *      _u8: dst [B][outH][outW][3] uint8;  _u8_f32: dst [B][3][outH][outW] float32 = resized / 255 (the hot path's frames). ---- */
/* A batch of frames of DIFFERENT sizes (the naive crop branch of ImageDataset, image_dataset.py:86-90, gives every frame a
 * size of its own) resized to one outH x outW batch in ONE launch pair.  One job per frame: byte offsets of the frame in
 * `src` ([inH][inW][3] uint8) and of its intermediate in `tmp` ([inH][outW][3]), and the int32-word offsets of ITS coefficient
 * tables in `tables` (per axis: bounds [out][2], coefficients [out][ks]); frame j writes dst[j] ([3][outH][outW] float32 =
 * resized / 255), i.e. the batch is produced in job order.  jobs / tables are device memory; max_inH = the tallest frame.
 * A frame may be stored mirrored (`flip`): the host then stages contiguous rows instead of reversing 3-byte pixels. */
typedef struct loans_resample_job {
    int64_t src_off, tmp_off;
    int32_t inH, inW;
    int32_t hb_off, hk_off, hks;
    int32_t vb_off, vk_off, vks;
    int32_t flip;               /* != 0: the frame is the horizontal mirror of the buffer (random_flip, image_dataset.py:40-44) */
} loans_resample_job;
int loans_resize_ragged_u8_f32(const uint8_t* src, uint8_t* tmp, float* dst, const loans_resample_job* jobs, int32_t njobs,
                               const int32_t* tables, int32_t max_inH, int32_t outH, int32_t outW, void* stream);
int loans_resize_lanczos_u8(const uint8_t* src, uint8_t* tmp, uint8_t* dst, int32_t B, int32_t inH, int32_t inW,
                            int32_t outH, int32_t outW, const int32_t* hbounds, const int32_t* hk, int32_t hks,
                            const int32_t* vbounds, const int32_t* vk, int32_t vks, void* stream);
int loans_resize_lanczos_u8_f32(const uint8_t* src, uint8_t* tmp, float* dst, int32_t B, int32_t inH, int32_t inW,
                                int32_t outH, int32_t outW, const int32_t* hbounds, const int32_t* hk, int32_t hks,
                                const int32_t* vbounds, const int32_t* vk, int32_t vks, void* stream);
/* [B][H][W][3] uint8 -> [B][3][H][W] float32 / 255 (frames that need no resize) */
int loans_u8hwc3_to_f32chw(const uint8_t* src, float* dst, int32_t B, int32_t H, int32_t W, void* stream);

/* library / build identification */
const char* loans_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LOANS_HIP_H */
