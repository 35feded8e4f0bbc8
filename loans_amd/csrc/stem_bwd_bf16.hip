// The stem's backward on bf16 storage in ONE kernel behind the two BN sums (round 5):
//     max_pooling_2d(3, 2, cover_all) <- relu <- bn1 <- conv1 (7x7 / 2, 3 -> 64, bias)        (sheep/resnet.py:43-44,72-73)
// Rounds 1-4 ran  loans_pool_bn_bwd_apply_rep_bf16  (gx = k1 g m + k2 y + k3 for every conv1 output: the largest tensor of the
// network, 1.07 GB at configs[2], written once)  ->  loans_wgrad_bf16s with LOANS_F_DENSE  (reads it back once per 64-column
// tile of K = 168: three times)  ->  loans_mul_f32 (window-padding columns): 0.53 + 0.43 ms at configs[2], the second one ALONE at
// the end of the step.  Here conv1's weight gradient is contracted from gx tiles that are REBUILT IN REGISTERS and never stored:
//
//     dw[co][ky][6 kx + c] += sum_p gx[p][co] * frame[b][2 oy + ky][6 ox + 6 kx + c]            (dense K rows, see igemm.hip)
//     gx[p][co] = k1 * (g[p][co] * (y scale + shift > 0)) + k2 * y[p][co] + k3,   g = the pooled gradient routed by the argmax
//     gbias[co] += sum_p gx[p][co]                                                                (conv1's bias gradient)
//
// A chunk = 32 consecutive output pixels of one conv1 output row.  Thread (pixel, 8-channel unit) -- the unit, and with it the five
// coefficient vectors, fixed for the thread's life -- loads y (16 B) and the (gradient, argmax) units of the at most four pooling
// windows that cover its pixel, forms gx in fp32 (summed into the bias gradient as it is), rounds it to bf16 -- the value the
// stored tensor held -- and writes it into the [pixel][channel] LDS tile; the frame tile [32 px][7 rows x 24 elements] goes
// global -> registers -> LDS as in wgrad16_kernel's dense mode; fragments by ds_read_b64_tr_b16; wave (wm, wn) of a 2 x 2 grid
// owns 32 output channels x 96 of the 192 (168 real) columns: 6 MFMAs per chunk.  Chunk c + 1's loads are in flight under chunk c's
// MFMAs (register prefetch, two LDS stages, one barrier per chunk).  Every block ends with plain stores of its [64][168] partial
// into its slab of a workspace (loans_fold_slabs_f32 adds them to dw in a fixed order; the three window-padding columns of each
// K row are written as zeros: no mask pass).  Algorithmic bytes: y + pooled gradient + argmax + frames once = 1.68 GB at configs[2].
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int CP = 32;                  // pixels per chunk
constexpr int NCOL = 192;               // staged columns: 7 x 24 = 168 real + 24 of padding
constexpr int SY = 64 + 32, SX = NCOL + 32;     // padded LDS row strides (elements), as in wgrad16_kernel
constexpr int XUNITS = CP * 24;         // 16-byte units of a frame tile: 3 per thread

struct StemBwdArgs {
    const __bf16* x;        // zero-padded packed-RGB frames [B][Hp][Wp3] (loans_prep_images_dense_bf16)
    const __bf16* y;        // conv1 output [B][Ho][Wo][64]
    const __bf16* gyp;      // pooled gradient [B][OH][OW][64]
    const uint8_t* idx;     // argmax positions [B][OH][OW][64]
    const float* scale;     // bn1 forward coefficients
    const float* shift;
    const float* k1;        // bn1 backward coefficients
    const float* k2;
    const float* k3;
    float* ws;              // [gridDim.x][64 * 168] partial weight gradients
    float* gbias;           // [64], += (fp32 atomics)
    int B, Hp, Wp3, Ho, Wo, OH, OW;
    int chunks_row, nchunks, chunks_per_block;
    unsigned x_bytes, y_bytes, p_bytes, i_bytes;
};

__global__ __launch_bounds__(256, 2) void stem_bwd16_kernel(const StemBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Ys = reinterpret_cast<__bf16*>(smem);       // [2][CP][SY]
    __bf16* Xs = Ys + 2 * CP * SY;                      // [2][CP][SX]
    const int tid = threadIdx.x;
    const int px = tid >> 3, cu = tid & 7;              // this thread's pixel of a chunk and its 8-channel unit
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.y), 0, (int)a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.gyp), 0, (int)a.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.idx), 0, (int)a.i_bytes, 0x00020000);

    f32x4 sc[2], sh[2], c1[2], c2[2], c3[2], bsum[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = cu * 8 + 4 * q;
        sc[q] = *reinterpret_cast<const f32x4*>(a.scale + c); sh[q] = *reinterpret_cast<const f32x4*>(a.shift + c);
        c1[q] = *reinterpret_cast<const f32x4*>(a.k1 + c); c2[q] = *reinterpret_cast<const f32x4*>(a.k2 + c);
        c3[q] = *reinterpret_cast<const f32x4*>(a.k3 + c);
        bsum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the frame tile's units of this thread: n = tid + 256 i -> (pixel n / 24, unit n % 24 = (row ky, 8 elements c8)); units 21..23 of a
    // pixel are padding (zeros).  Byte offset relative to the chunk's first K row, kept per unit.
    unsigned xrel[3];
    int xdst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int n = tid + 256 * i, p = n / 24, q = n - p * 24;
        const int ky = q / 3, c8 = q - ky * 3;
        xrel[i] = q < 21 ? (unsigned)((ky * a.Wp3 + 6 * p + 8 * c8) * 2) : 0xFFFFFFFFu;
        xdst[i] = p * SX + q * 8;
    }

    const int c_begin = blockIdx.x * a.chunks_per_block;
    int c_end = c_begin + a.chunks_per_block;
    if (c_end > a.nchunks) c_end = a.nchunks;

    // ---- the loads of one chunk, into registers
    u32x4 ry, rg[4], rx[3];
    u32x2 ri[4];
    unsigned wmeta;          // per window w (8 bits each): bit 7 = the window exists and covers this pixel, bits 0-3 = its argmax code here
    bool pvalid = false;     // this thread's pixel of the loaded chunk exists
    auto load_chunk = [&](int c) {
        const bool cv = c < c_end;
        const int row = c / a.chunks_row;
        const int ox = (c - row * a.chunks_row) * CP + px;
        const int b = row / a.Ho, oy = row - b * a.Ho;
        const bool pv = cv && ox < a.Wo;
        pvalid = pv;
        const unsigned yoff = (unsigned)(((b * a.Ho + oy) * a.Wo + ox) * 64 + cu * 8) * 2u;
        ry = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)(yoff | ((unsigned)pv - 1u)), 0, 0);
        // windows (oh1 - dh, ow1 - dw), dh, dw in {0, 1}: oh1 = oy / 2 holds the pixel at window row oy & 1, the window above it at
        // row (oy & 1) + 2 -- which exists only for even oy
        const int oh1 = oy >> 1, ow1 = ox >> 1;
        wmeta = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int dh = w >> 1, dw = w & 1;
            const int oh = oh1 - dh, ow = ow1 - dw;
            const int kr = (oy & 1) + 2 * dh, kq = (ox & 1) + 2 * dw;
            const bool ok = pv && kr <= 2 && kq <= 2 && oh >= 0 && ow >= 0 && oh < a.OH && ow < a.OW;
            const unsigned e = (unsigned)(((b * a.OH + oh) * a.OW + ow) * 64 + cu * 8);
            const unsigned bad = (unsigned)ok - 1u;
            rg[w] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, (int)((e * 2u) | bad), 0, 0);
            ri[w] = __builtin_amdgcn_raw_buffer_load_b64(rs_i, (int)(e | bad), 0, 0);
            wmeta |= (ok ? (0x80u | (unsigned)(kr * 3 + kq)) : 0u) << (8 * w);
        }
        const unsigned xbase = (unsigned)(((b * a.Hp + 2 * oy) * a.Wp3 + 6 * (ox - px)) * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned off = (cv && xrel[i] != 0xFFFFFFFFu) ? xbase + xrel[i] : 0xFFFFFFFFu;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0);
        }
    };
    // ---- gx of this thread's (pixel, unit) from the registers, into the LDS stage; the frame units beside it
    auto stage_chunk = [&](int buf) {
        const bf16x8_t yb = __builtin_bit_cast(bf16x8_t, ry);
        f32x4 yv[2] = {f32x4{(float)yb[0], (float)yb[1], (float)yb[2], (float)yb[3]}, f32x4{(float)yb[4], (float)yb[5], (float)yb[6], (float)yb[7]}};
        f32x4 g[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned m = (wmeta >> (8 * w)) & 0xFFu;
            if (!(m & 0x80u)) continue;
            const unsigned k = m & 0xFu;
            const bf16x8_t gb = __builtin_bit_cast(bf16x8_t, rg[w]);
            const unsigned lo = ri[w].x, hi = ri[w].y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (((lo >> (8 * e)) & 0xFFu) == k) g[0][e] += (float)gb[e];
                if (((hi >> (8 * e)) & 0xFFu) == k) g[1][e] += (float)gb[4 + e];
            }
        }
        bf16x8_t o;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 act = yv[q] * sc[q] + sh[q];
            f32x4 gm = g[q];
            gm.x = act.x > 0.f ? gm.x : 0.f; gm.y = act.y > 0.f ? gm.y : 0.f;
            gm.z = act.z > 0.f ? gm.z : 0.f; gm.w = act.w > 0.f ? gm.w : 0.f;
            f32x4 v = c1[q] * gm + c2[q] * yv[q] + c3[q];
            if (!pvalid) v = f32x4{0.f, 0.f, 0.f, 0.f};          // a pixel beyond the row (or the block's range): no gradient, not k3
            o[4 * q + 0] = (__bf16)v.x; o[4 * q + 1] = (__bf16)v.y; o[4 * q + 2] = (__bf16)v.z; o[4 * q + 3] = (__bf16)v.w;
            bsum[q] += v;
        }
        *reinterpret_cast<bf16x8_t*>(Ys + (buf * CP + px) * SY + cu * 8) = o;
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(Xs + buf * CP * SX + xdst[i]) = rx[i];
    };

    // fragment addressing: wgrad16_kernel's (16-lane group g = lane >> 4: pixels 8 (g >> 1) + 0..3 (+4: second read), channels
    // 16 (g & 1) .. +15 of the wave's 32-wide tile; lane 4 q + p of the group addresses pixel row q, channels 4 p .. 4 p + 3)
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, fq = li >> 2, fp = li & 3, cg = (lane >> 4) & 1;
    const int trY = (8 * h + fq) * SY + wm * 32 + 16 * cg + 4 * fp;
    const int trX = (8 * h + fq) * SX + wn * 96 + 16 * cg + 4 * fp;
    typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_t;
    auto frag = [&](const __bf16* base, int stride) {
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)base);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_t)(base + 4 * stride));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    if (c_begin < c_end) load_chunk(c_begin);
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
        stage_chunk(buf);                   // (waits for chunk c's loads)
        load_chunk(c + 1);                  // in flight under this chunk's MFMAs; nothing is fetched beyond c_end
        __syncthreads();                    // stage `buf` is complete; every wave has left the MFMAs that read the other stage
#pragma unroll
        for (int s = 0; s < CP / 16; ++s) {
            const bf16x8_t af = frag(Ys + (buf * CP + 16 * s) * SY + trY, SY);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bf16x8_t bf = frag(Xs + (buf * CP + 16 * s) * SX + trX + j * 32, SX);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[j], 0, 0, 0);
            }
        }
        buf ^= 1;
    }

    // ---- this block's partial [64][168] into its slab; the bias-gradient sums of the block through LDS, one atomic per channel
    float* slab = a.ws + (size_t)blockIdx.x * (64 * 168);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int col = wn * 96 + j * 32 + r;
        if (col >= 168) continue;
        const bool real = (col % 24) < 21;          // the 8th pixel of a K row met real frame data: not a weight
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            slab[co * 168 + col] = real ? acc[j][e] : 0.f;
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);        // [32 pixels][64]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[px * 64 + cu * 8 + 4 * q + e] = bsum[q][e];
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
#pragma unroll 8
        for (int p = 0; p < CP; ++p) s += red[p * 64 + tid];
        atomic_add_f32(a.gbias + tid, s);
    }
}

}  // namespace

extern "C" int64_t loans_stem_bwd_bf16_ws_floats(int32_t B, int32_t Ho, int32_t Wo) {
    if (B <= 0 || Ho <= 0 || Wo <= 0) return LOANS_EINVAL;
    const int cus = loans_device_cus();
    if (cus <= 0) return LOANS_EINVAL;
    const int64_t nchunks = (int64_t)B * Ho * ((Wo + CP - 1) / CP);
    int64_t blocks = 2 * cus;
    if (blocks > nchunks) blocks = nchunks;
    const int64_t per = (nchunks + blocks - 1) / blocks;
    blocks = (nchunks + per - 1) / per;
    return blocks * 64 * 168;
}

extern "C" int loans_stem_bwd_bf16(const void* frames, const void* y, const void* gy_pooled, const uint8_t* idx, const float* scale,
                                   const float* shift, const float* k1, const float* k2, const float* k3, float* dw, float* gbias,
                                   float* ws, int64_t ws_floats, int32_t B, int32_t Hp, int32_t Wp3, int32_t Ho, int32_t Wo,
                                   int32_t OH, int32_t OW, void* stream) {
    if (!frames || !y || !gy_pooled || !idx || !scale || !shift || !k1 || !k2 || !k3 || !dw || !gbias || !ws) return LOANS_EINVAL;
    if (B <= 0 || Ho < 3 || Wo < 3 || (Wp3 & 1)) return LOANS_EINVAL;
    if (OH != (Ho - 2) / 2 + 1 || OW != (Wo - 2) / 2 + 1) return LOANS_EINVAL;           // max_pooling_2d(3, 2, cover_all)
    if (2 * (Ho - 1) + 7 > Hp || 6 * (Wo - 1) + 24 > Wp3) return LOANS_EINVAL;           // every K row lies inside its frame row
    const int64_t xb = (int64_t)B * Hp * Wp3 * 2, yb = (int64_t)B * Ho * Wo * 64 * 2, pb = (int64_t)B * OH * OW * 64 * 2;
    if (xb >= 0x7FFFFFF0ll || yb >= 0x7FFFFFF0ll) return LOANS_ERANGE;                  // 31-bit byte offsets (all-ones = no load)
    const int64_t need = loans_stem_bwd_bf16_ws_floats(B, Ho, Wo);
    if (need <= 0 || ws_floats < need) return LOANS_EINVAL;
    StemBwdArgs a;
    a.x = static_cast<const __bf16*>(frames); a.y = static_cast<const __bf16*>(y); a.gyp = static_cast<const __bf16*>(gy_pooled);
    a.idx = idx; a.scale = scale; a.shift = shift; a.k1 = k1; a.k2 = k2; a.k3 = k3; a.ws = ws; a.gbias = gbias;
    a.B = B; a.Hp = Hp; a.Wp3 = Wp3; a.Ho = Ho; a.Wo = Wo; a.OH = OH; a.OW = OW;
    a.chunks_row = (Wo + CP - 1) / CP;
    a.nchunks = B * Ho * a.chunks_row;
    const int blocks = (int)(need / (64 * 168));
    a.chunks_per_block = (a.nchunks + blocks - 1) / blocks;
    a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb; a.p_bytes = (unsigned)pb; a.i_bytes = (unsigned)(pb / 2);
    constexpr size_t lds = (size_t)2 * CP * (SY + SX) * 2;
    static_assert(lds >= (size_t)CP * 64 * 4, "the bias-gradient fold fits the operand stages");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(stem_bwd16_kernel, dim3(blocks), dim3(256), lds, st, a);
    LOANS_LAUNCH_CHECK();
    return loans_fold_slabs_f32(ws, dw, 64 * 168, blocks, stream);
}
