// PAN (Efficient Image Super-Resolution Using Pixel Attention) forward on gfx950.
// Replaces architectures/PAN_arch.py:11-222 (defaults utils/defaults.py:78-89: nf 40, unf 24,
// nb 16, self attention on, nearest up-blocks) and SelfAttentionBlock (block.py:398-473).
//
//   every conv of the network (3..40 channels, 3x3 and 1x1) runs on the SR path's halo-tile MFMA kernel (conv3x3.hip,
//   conv3x3_pc): channels padded to whole 32-channel groups of blocked-NHWC fp16 slabs whose pad channels are kept zero
//   (zero weights and bias), a 1x1 conv = a 3x3 panel with a centre tap only (the kernel is bound by its loads and
//   stores at these widths; its epilogue writes the fp16 slab directly); epilogues used: bias, LeakyReLU(0.2), + residual,
//   nearest-2x input (`up`), res1 * sigmoid(conv) (pixel attention, with / without LeakyReLU), planar fp32 (conv_last)
//   conv1_a | conv1_b                          ONE 40->52 conv: a -> channel group 0 (0..19), b -> group 1 (32..51)
//   PACnv: k3(x) * sigmoid(k2(x))              k3 -> fp16 slab, then k2 with the gate epilogue multiplying it
//   torch.cat([a, b])                          channel groups 0 / 1 of one slab; conv3 reads both
//   FSA: MaxPool2d(4) -> f,g,h 1x1 -> softmax(f^T g) -> h att^T -> bicubic up -> gamma*out + in
//                                              pan_maxpool, one 40->50 gather GEMM (gather_gemm.h, fp32 rows), pan_attention (a query
//                                              per lane, keys split over 4 waves and tiled through LDS, two-pass
//                                              softmax, no N x N matrix in memory), pan_fsa_combine (ATen's
//                                              bicubic, A = -0.75, align_corners=False)
//   nearest-2x Upsample in the up-blocks       folded into the conv's input addressing (ConvLaunch.up)
//   + bilinear(x, align_corners=True)          pan_final, NCHW output
// Reference quirk kept: B.sequential() flattens with children(), which yields the shared LeakyReLU
// of pa_upconv_block once, so NO activation follows HRconv (golden G8 confirms).
#include "common.h"
#include "gather_gemm.h"

#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

using namespace innfer;

namespace {

__device__ __forceinline__ float slab_get(const f16* s, long g, long pix, int ch) {
    return (float)s[(ch >> 5) * g + pix * 32 + (ch & 31)];
}

__global__ void pan_pre(const void* in, int in_f32, int C, long HW, int N, f16* slab) {        // NCHW -> one zero-padded 32-channel group
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * HW) return;
    const long n = i / HW, px = i % HW;
    f16 v[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        f16 t = (f16)0.f;
        if (c < C) { const long o = (n * C + c) * HW + px; t = in_f32 ? (f16)((const float*)in)[o] : ((const f16*)in)[o]; }
        v[c] = t;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f16x8*)(slab + i * 32 + 8 * q) = *(const f16x8*)(v + 8 * q);
}

// MaxPool2d(4): one thread per (pooled pixel, 8 channels), 16-byte loads (C % 8 == 0); the pad channels up to the next multiple of 32 are
// written as zeros (the projections behind it read whole 32-channel groups)
// fp32 mode: NCHW fp32 -> a (hi, lo) pair of fp16 slabs (lo = fp16((x - hi) * 2^11), `lo` elements behind hi; conv3x3_pc's split operands), pad channels zero.
// A thread: 8 channels of one pixel.
__global__ void pan_nchw_to_slab_pair(const float* x, int C, long HW, int N, int groups, f16* slab, long lo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x, npx = (long)N * HW;
    if (i >= npx * groups * 4) return;
    const long pix = i / (groups * 4);
    const int o = (int)(i - pix * groups * 4), g = o >> 2, oct = o & 3;
    const long n = pix / HW, q = pix - n * HW;
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = g * 32 + oct * 8 + e;
        const float f = c < C ? x[(n * C + c) * HW + q] : 0.f;
        const f16 hh = (f16)f;
        h[e] = hh; l[e] = (f16)((f - (float)hh) * 2048.0f);
    }
    f16* d = slab + (long)g * npx * 32 + pix * 32 + oct * 8;
    *(f16x8*)d = h;
    *(f16x8*)(d + lo) = l;
}

__global__ void pan_maxpool(const f16* in, long in_g, int C, int N, int H, int W, int hp, int wp, f16* out, long out_g) {
    const int c8 = ((C + 31) / 32 * 32) >> 3;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * hp * wp * c8) return;
    const long i = t / c8;
    const int c = (int)(t - i * c8) * 8;
    const int x = (int)(i % wp), y = (int)((i / wp) % hp);
    const long n = i / ((long)wp * hp);
    const f16* base = in + (c >> 5) * in_g + (c & 31);
    f16x8 m;
    if (c >= C) {
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (f16)0.f;
        *(f16x8*)(out + (c >> 5) * out_g + i * 32 + (c & 31)) = m;
        return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = (f16)-INFINITY;
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const f16x8 v = *(const f16x8*)(base + ((n * H + 4 * y + dy) * W + 4 * x + dx) * 32);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
        }
    *(f16x8*)(out + (c >> 5) * out_g + i * 32 + (c & 31)) = m;
}

// fgh: fp32 [N*Np][64] = [f(5) | g(5) | h(40)] without bias.  att_j = softmax_j(f_i . g_j);
// out_i[c] = sum_j h_j[c] att_j.  A workgroup owns 64 queries (a query per lane, 40 accumulators in
// registers) and its 4 waves split the keys: wave w takes the key tiles t = w, w+4, ...; keys / values
// go through a wave-private LDS tile of 64 and are read as broadcasts, so they are fetched once per 64
// queries instead of once per query.  Two passes (max, then exp / accumulate); the four partial sums
// are added in wave order: the result does not depend on the batch.
__global__ __launch_bounds__(256) void pan_attention(const float* fgh, const float* bf, const float* bg, const float* bh,
                                                     int Np, float* out) {
    constexpr int CQ = 5, C = 40, KT = 64, PITCH = 48;       // LDS row: g at 0..4, h at 8..47
    __shared__ __attribute__((aligned(16))) float lds[4 * KT * PITCH];
    __shared__ float redm[4][64], reds[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = blockIdx.y;
    float* kt = lds + wave * KT * PITCH;
    const int i = blockIdx.x * 64 + lane;
    const float* base = fgh + (long)n * Np * 64;
    float fi[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) fi[c] = i < Np ? base[(long)i * 64 + c] + bf[c] : 0.f;
    const int ntiles = (Np + KT - 1) / KT, rounds = (ntiles + 3) / 4;
    auto load_tile = [&](int t, bool with_h) {
        const int j = t * KT + lane;                          // this lane stages key j
        const float* row = base + (long)j * 64;
#pragma unroll
        for (int c = 0; c < CQ; ++c) kt[lane * PITCH + c] = j < Np ? row[CQ + c] + bg[c] : 0.f;
        if (with_h) {
#pragma unroll
            for (int c = 0; c < C; ++c) kt[lane * PITCH + 8 + c] = j < Np ? row[2 * CQ + c] + bh[c] : 0.f;
        }
    };
    auto score = [&](int jj) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CQ; ++c) s += fi[c] * kt[jj * PITCH + c];
        return s;
    };
    float m = -INFINITY;
    for (int r = 0; r < rounds; ++r) {
        const int t = 4 * r + wave;
        __syncthreads();
        load_tile(t, false);
        __syncthreads();
        const int cnt = min(KT, Np - t * KT);
        for (int jj = 0; jj < cnt; ++jj) m = fmaxf(m, score(jj));
    }
    redm[wave][lane] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redm[0][lane], redm[1][lane]), fmaxf(redm[2][lane], redm[3][lane]));
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    float sum = 0.f;
    for (int r = 0; r < rounds; ++r) {
        const int t = 4 * r + wave;
        __syncthreads();
        load_tile(t, true);
        __syncthreads();
        const int cnt = min(KT, Np - t * KT);
        for (int jj = 0; jj < cnt; ++jj) {
            const float e = expf(score(jj) - m);
            sum += e;
            const f32x4* hv = (const f32x4*)(kt + jj * PITCH + 8);
#pragma unroll
            for (int q = 0; q < C / 4; ++q) {
                const f32x4 h = hv[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[4 * q + k] += h[k] * e;
            }
        }
    }
    // combine the four key partitions in wave order: partials go through the (now free) tile buffers
    __syncthreads();
    reds[wave][lane] = sum;
#pragma unroll
    for (int c = 0; c < C; ++c) kt[c * 64 + lane] = acc[c];       // [wave][c][query]
    __syncthreads();
    if (i < Np) {
        const float inv = 1.0f / (((reds[0][lane] + reds[1][lane]) + reds[2][lane]) + reds[3][lane]);
        float* o = out + ((long)n * Np + i) * C;
        for (int c = wave; c < C; c += 4) {
            const float v = ((lds[(0 * KT * PITCH) + c * 64 + lane] + lds[(1 * KT * PITCH) + c * 64 + lane]) +
                             lds[(2 * KT * PITCH) + c * 64 + lane]) + lds[(3 * KT * PITCH) + c * 64 + lane];
            o[c] = v * inv;
        }
    }
}

// ---- the same attention on the matrix cores (round 4) -----------------------------------------------------------------------------------------
// pan_attention above is a VALU kernel: 33 TFLOP/s, 3.2 ms of a 6.0 ms forward at 540 x 960 (32 400 pooled pixels: 1.05 G query-key pairs) once the SCPA
// trunk had shrunk.  Here, per workgroup of 8 waves and 128 queries (16 per wave: one MFMA column set):
//   scores   s[key][query] = g_key . f_query as v_mfma_f32_16x16x32_f16 with keys as rows: both operands are fp16 (hi, lo * 2^11) pairs packed into the
//            32-deep k dimension -- one MFMA for gh . fh, one for (gh . fl + gl . fh), s = s_hh + 2^-11 s_x: fp32-accurate scores (the 2^-22 term is dropped)
//   softmax  one pass with a running row maximum (accumulators and sums rescaled when it moves): p = exp(s - m) <= 1 and the sum in fp32 on the lanes
//   out      o[c][query] += h[key][c] p[key][query]: the score MFMAs' rows are assigned to keys so that a lane ends with 8 CONSECUTIVE keys of its query
//            (key of row rho of score tile t: 8 (rho >> 2) + 4 t + (rho & 3)) -- its fp16 p values ARE the B fragment of the P V product; h as fp16 A
//            fragments [channel][key] prepared once per forward (pan_attn_prep)
// Keys go through a double-buffered LDS stage of two 32-key blocks (per block 1 KB of g pairs + 3 KB of h fragments: one 16-byte load per thread), one
// barrier per 64 keys; 8 waves x 16 queries per workgroup.  Never materialises the Np x Np matrix either.  att rows as before: fp32 [N * Np][C].
constexpr int ATT_C = 40, ATT_CQ = 5, ATT_KB = 32;
constexpr int ATT_SP = 4;          // block pairs (64 keys each) per LDS stage of pan_attention_mfma

// per pooled pixel: QK[point] = {fh, fl', gh, gl'} (four 16-byte octets: 5 values + 3 zeros each; x = xh + xl' * 2^-11, bias added);
// per image and 32-key block: Vt[blk][t 0..2][lane][8] = h[key blk * 32 + 8 lg + e][channel 16 t + li] + bias as fp16 (0 beyond Np / C)
// Vl != nullptr (fp32 mode): the h fragments as (hi, lo) pairs, lo = fp16((h - hi) * 2^11) in a second array of Vt's layout (the row of ones: hi 1, lo 0)
__global__ void pan_attn_prep(const float* fgh, const float* bf, const float* bg, const float* bh, int Np, int nblk, f16* QK, f16* Vt, f16* Vl) {
    const int n = blockIdx.y;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const float* base = fgh + (long)n * Np * 64;
    if (i < Np) {
        f16x8 o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[q][e] = (f16)0.f;
#pragma unroll
        for (int c = 0; c < ATT_CQ; ++c) {
            const float f = (base[i * 64 + c] + bf[c]) * 1.44269504088896340736f, g = base[i * 64 + ATT_CQ + c] + bg[c];      // f in log2 units: the kernel's exponential is 2^x
            const f16 fh = (f16)f, gh = (f16)g;
            o[0][c] = fh; o[1][c] = (f16)((f - (float)fh) * 2048.0f);
            o[2][c] = gh; o[3][c] = (f16)((g - (float)gh) * 2048.0f);
        }
        f16* dst = QK + ((long)n * nblk * ATT_KB + i) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f16x8*)(dst + 8 * q) = o[q];
    } else if (i < (long)nblk * ATT_KB) {                       // padding keys of the last block: zeros (masked to -inf in the kernel)
        f16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (f16)0.f;
        f16* dst = QK + ((long)n * nblk * ATT_KB + i) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f16x8*)(dst + 8 * q) = z;
    }
    if (i < (long)nblk * 3 * 64) {                              // one h fragment (8 keys of one channel) per thread
        const int blk = (int)(i / 192), r = (int)(i - (long)blk * 192), t = r >> 6, lane = r & 63, li = lane & 15, lg = lane >> 4, c = 16 * t + li;
        f16x8 v, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long key = (long)blk * ATT_KB + 8 * lg + e;
            const float f = (key < Np && c < ATT_C) ? base[key * 64 + 2 * ATT_CQ + c] + bh[c] : ((key < Np && c == 47) ? 1.0f : 0.f);      // channel 47: ones -- its row of P V is the row sum
            v[e] = (f16)f;
            vl[e] = (f16)((f - (float)v[e]) * 2048.0f);
        }
        *(f16x8*)(Vt + ((long)n * nblk * 192 + i) * 8) = v;
        if (Vl) *(f16x8*)(Vl + ((long)n * nblk * 192 + i) * 8) = vl;
    }
}

// SPLIT (the fp32 mode's attention, round 6): p and h as (hi, lo * 2^11) fp16 pairs as well -- p V = ph vh + 2^-11 (ph vl + pl vh), three MFMAs per fragment into two accumulator
// sets (the row of ones gives sum(ph) and sum(pl): numerator and denominator use the same 22-bit p); a stage block grows by the 3 KB of lo fragments.
template <bool SPLIT>
__global__ __launch_bounds__(512) void pan_attention_mfma(const f16* QK, const f16* Vt, const f16* Vl, int Np, int nblk, float* out) {
    // 8 waves x 16 queries (two waves per SIMD: one wave's exps and LDS reads run under the other's MFMAs); a stage holds TWO 32-key blocks (8 KB: one
    // 16-byte piece per thread), so a barrier is paid once per 64 keys
    // (round 6: a stage holds ATT_SP block pairs = 256 keys -- the barrier per 64 keys was a quarter of the kernel: 506 of them per workgroup at 32 400 keys)
    constexpr int SP = SPLIT ? 2 : ATT_SP;
    constexpr int BLK = SPLIT ? 7168 : 4096, PER = BLK / 16, NPC = (2 * PER + 511) / 512;      // bytes / 16-byte pieces of a block in a stage; pieces per thread and pair
    __shared__ __attribute__((aligned(16))) char st[2][SP * 2 * BLK];  // per stage, pair and block: g pairs of 32 keys (32 x 32 B) | three h fragments (3 x 1 KB) [| their lo parts]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4, n = blockIdx.y;
    const f16* qk = QK + (long)n * nblk * ATT_KB * 32;
    const f16* vt = Vt + (long)n * nblk * 192 * 8;
    [[maybe_unused]] const f16* vl = SPLIT ? Vl + (long)n * nblk * 192 * 8 : nullptr;
    const f16x8 z8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // B operands of the score MFMAs for this wave's 16 queries (a query beyond Np reads a padding key's zero row: its results are never stored)
    const int q = blockIdx.x * 128 + wave * 16 + li;
    f16x8 bhh, bx;
    {
        const int qq = q < nblk * ATT_KB ? q : 0;
        const f16x8 fh = *(const f16x8*)(qk + (long)qq * 32), fl = *(const f16x8*)(qk + (long)qq * 32 + 8);
        bhh = lg == 0 ? fh : z8;
        bx = lg == 0 ? fl : (lg == 1 ? fh : z8);
        // (round 6) the cross operand carries the 2^-11 of s = s_hh + 2^-11 s_x itself, so the cross MFMA accumulates ON TOP of the main one and the sixteen joining
        // multiply-adds per key pair are gone (the kernel is bound by its vector instructions): fl' 2^-11 and fh 2^-11 are fp16 values below ~5e-4 |f| -- where they
        // fall under the normal range their rounding is <= 3e-8 absolute per term, against scores that carry 1e-6
#pragma unroll
        for (int e = 0; e < 8; ++e) bx[e] = bx[e] * (f16)(1.0f / 2048.0f);
    }
    // staging: thread tid moves the 16-byte pieces tid, tid + 512, .. (< 2 PER) of block pair bp (blocks 2 bp, 2 bp + 1)
    auto stage_src = [&](int bp, int c) -> const f16* {
        const int pc = tid + 512 * c, half = pc >= PER ? 1 : 0, t8 = pc - half * PER;
        const int b = 2 * bp + half < nblk ? 2 * bp + half : nblk - 1;                            // (an odd tail: the last block twice, its second copy masked)
        if (t8 < 64) return qk + ((long)b * ATT_KB + (t8 >> 1)) * 32 + 16 + (t8 & 1) * 8;        // {gh, gl'} of key t8 / 2
        if (!SPLIT || t8 < 256) return vt + ((long)b * 192 + (t8 - 64)) * 8;
        return vl + ((long)b * 192 + (t8 - 256)) * 8;
    };
    auto piece_ok = [&](int c) { return tid + 512 * c < 2 * PER; };
    auto dst_off = [&](int c) { const int pc = tid + 512 * c, half = pc >= PER ? 1 : 0, t8 = pc - half * PER; return half * BLK + t8 * 16; };
    // the key of row rho (= li) of score tile t, and this lane's A-operand octet: gh for k-octet 0, gl' for k-octet 1
    const int koff0 = (8 * (li >> 2) + (li & 3)) * 32 + (lg == 1 ? 16 : 0), koff1 = koff0 + 4 * 32;
    auto scores = [&](const char* sp, const f32x4& c0, f32x4 (&sc)[2]) __attribute__((always_inline)) {      // sc[t][j]: key 8 lg + 4 t + j of the block, query li, MINUS the offset in c0
        const f16x8 g0 = *(const f16x8*)(sp + koff0), g1 = *(const f16x8*)(sp + koff1);
        // (no masking of the A operand: the B operands are zero in every k-octet that must not contribute -- bhh beyond octet 0, bx beyond octet 1 -- so whatever finite
        //  g data a lane of those octets holds is multiplied by zero; round 6: the four selects per key block were 16 of ~35 vector instructions)
        const f32x4 h0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0, bhh, c0, 0, 0, 0), h1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(g1, bhh, c0, 0, 0, 0);
        sc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0, bx, h0, 0, 0, 0);
        sc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g1, bx, h1, 0, 0, 0);
    };
    const int npair = (nblk + 1) / 2;
    // ONE pass over the keys with a running row maximum m.  Round 5 (<= 0.3 ms at 32 400 keys; the kernel is bound by the VALU work per score, 1.05 G scores):
    //  * the scores arrive in log2 units (pan_attn_prep scales f by log2 e) and ALREADY MINUS m: -m is the score MFMAs' C operand, so p = 2^(s - m) is one v_exp_f32 per score
    //    with no subtraction and no multiplication in front of it; when the maximum moves (wave-uniform test; after the first blocks it rarely does) the pair takes the
    //    slow path: m += delta, accumulators rescaled by 2^-delta, p = 2^(s - m_old - delta);
    //  * the row sum is a row of P V: channel 47 of the h fragments is 1 (pan_attn_prep), so lane (li, lg = 3) carries sum(p) in acc[2][3] -- the fp16 p the numerator uses --
    //    and the sixteen additions per lane and pair are gone;
    //  * p -> fp16 two at a time (v_cvt_pkrtz); the maximum over a lane's sixteen scores as a tree (v_max3).
    // Masked keys (>= Np, the padding of the last block) score -inf: p = 0.  The first pair starts from m = 0 and takes its own maximum as delta.
    float m = 0.f;
    f32x4 c0 = z4;                                                   // -m: the C operand of the score MFMAs
    f32x4 acc[3] = {z4, z4, z4};
    [[maybe_unused]] f32x4 accx[3] = {z4, z4, z4};
#pragma unroll
    for (int u = 0; u < SP; ++u)
#pragma unroll
        for (int c = 0; c < NPC; ++c)
            if (u < npair && piece_ok(c)) *(f16x8*)(st[0] + u * 2 * BLK + dst_off(c)) = *(const f16x8*)stage_src(u, c);
    __syncthreads();
    for (int bp0 = 0; bp0 < npair; bp0 += SP) {
        f16x8 nxt[SP][NPC];
#pragma unroll
        for (int u = 0; u < SP; ++u)
#pragma unroll
            for (int c = 0; c < NPC; ++c) nxt[u][c] = (bp0 + SP + u < npair && piece_ok(c)) ? *(const f16x8*)stage_src(bp0 + SP + u, c) : z8;
        const char* const stg = st[(bp0 / SP) & 1];
        for (int bp = bp0; bp < bp0 + SP && bp < npair; ++bp) {
        const char* const sb = stg + (bp - bp0) * 2 * BLK;
        f32x4 sc[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b = 2 * bp + h;
            scores(sb + h * BLK, c0, sc[h]);
            if ((b + 1) * ATT_KB > Np) {                             // (wave-uniform: only the last block of an image is ragged, an odd tail's second copy wholly masked)
                const int kb = b < nblk ? b * ATT_KB + 8 * lg : Np;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (kb + 4 * t + j >= Np) sc[h][t][j] = -INFINITY;
            }
        }
        // the lane's own maximum decides whether the pair takes the slow path (some score of some query of the wave above the running maximum, or the first pair); only
        // there do the four lanes of a query need their common maximum (two cross-lane exchanges, off the fast path since round 6)
        float bm;
        {
            const float m0 = fmaxf(fmaxf(sc[0][0][0], sc[0][0][1]), sc[0][0][2]), m1 = fmaxf(fmaxf(sc[0][0][3], sc[0][1][0]), sc[0][1][1]);
            const float m2 = fmaxf(fmaxf(sc[0][1][2], sc[0][1][3]), sc[1][0][0]), m3 = fmaxf(fmaxf(sc[1][0][1], sc[1][0][2]), sc[1][0][3]);
            const float m4 = fmaxf(fmaxf(sc[1][1][0], sc[1][1][1]), sc[1][1][2]);
            bm = fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), fmaxf(m4, sc[1][1][3]));
        }
        if (bp == 0 || __any(bm > 0.f)) {                            // the maximum moved for some query of this wave
            bm = fmaxf(bm, __shfl_xor(bm, 16));
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            const float delta = bp == 0 ? bm : fmaxf(bm, 0.f);      // (finite on the first pair: every image has a key in its first block)
            const float rescale = __builtin_amdgcn_exp2f(-delta);    // (acc is zero on the first pair)
            m += delta;
            c0 = f32x4{-m, -m, -m, -m};
#pragma unroll
            for (int t = 0; t < 3; ++t) { acc[t] = acc[t] * rescale; if (SPLIT) accx[t] = accx[t] * rescale; }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sc[h][t][j] -= delta;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const char* sp = sb + h * BLK;
            f16x8 v[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) v[t] = *(const f16x8*)(sp + 1024 + t * 1024 + lane * 16);
            typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
            union { f16x8 v8; f16x2_t v2[4]; } pk;
            if constexpr (SPLIT) {
                f16x8 w[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) w[t] = *(const f16x8*)(sp + 4096 + t * 1024 + lane * 16);
                union { f16x8 v8; f16x2_t v2[4]; } pl;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const float e0 = __builtin_amdgcn_exp2f(sc[h][t][2 * jj]), e1 = __builtin_amdgcn_exp2f(sc[h][t][2 * jj + 1]);
                        const f16x2_t ph = __builtin_bit_cast(f16x2_t, __builtin_amdgcn_cvt_pkrtz(e0, e1));
                        pk.v2[2 * t + jj] = ph;
                        pl.v2[2 * t + jj] = __builtin_bit_cast(f16x2_t, __builtin_amdgcn_cvt_pkrtz((e0 - (float)ph[0]) * 2048.0f, (e1 - (float)ph[1]) * 2048.0f));
                    }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[t], pk.v8, acc[t], 0, 0, 0);
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t], pk.v8, accx[t], 0, 0, 0);
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[t], pl.v8, accx[t], 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    pk.v2[2 * t + jj] = __builtin_bit_cast(f16x2_t, __builtin_amdgcn_cvt_pkrtz(__builtin_amdgcn_exp2f(sc[h][t][2 * jj]), __builtin_amdgcn_exp2f(sc[h][t][2 * jj + 1])));
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v[t], pk.v8, acc[t], 0, 0, 0);
            }
        }
        }
        if (bp0 + SP < npair) {
            char* const nst = st[((bp0 / SP) + 1) & 1];
#pragma unroll
            for (int u = 0; u < SP; ++u)
#pragma unroll
                for (int c = 0; c < NPC; ++c)
                    if (bp0 + SP + u < npair && piece_ok(c)) *(f16x8*)(nst + u * 2 * BLK + dst_off(c)) = nxt[u][c];
        }
        __syncthreads();
    }
    if constexpr (SPLIT) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_fmaf(accx[t][j], 1.0f / 2048.0f, acc[t][j]);
    }
    const float sum = __shfl(acc[2][3], 48 + li);                    // channel 47 = row 15 of tile 2: lane (li, lg = 3), element 3
    if (q < Np) {
        const float inv = 1.0f / sum;
        float* o = out + ((long)n * Np + q) * ATT_C;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int c = 16 * t + 4 * lg;
            if (c < ATT_C) *(f32x4*)(o + c) = acc[t] * inv;
        }
    }
}

__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// t = gamma * bicubic(att, size=(H,W), align_corners=False) + inp      (ATen upsample_bicubic2d)
__global__ void pan_fsa_combine(const float* att, int hp, int wp, int C, const f16* inp, long g, int N, int H, int W,
                                const float* gamma, f16* dst) {
    const int c8 = ((C + 31) / 32 * 32) >> 3;        // one thread per (pixel, 8 channels); per channel the same sums in the same order
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * H * W * c8) return;
    const long i = t / c8;
    const int c0 = (int)(t - i * c8) * 8;
    if (c0 >= C) {                                   // pad channels of the last group: zeros (the up conv reads whole groups)
        f16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (f16)0.f;
        *(f16x8*)(dst + (c0 >> 5) * g + i * 32 + (c0 & 31)) = z;
        return;
    }
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const long n = i / ((long)W * H);
    const float A = -0.75f;
    const float sy = (float)hp / (float)H, sx = (float)wp / (float)W;
    const float ry = sy * ((float)Y + 0.5f) - 0.5f, rx = sx * ((float)X + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    const float ty = ry - (float)iy, tx = rx - (float)ix;
    const float wy[4] = {cc2(ty + 1.f, A), cc1(ty, A), cc1(1.f - ty, A), cc2(2.f - ty, A)};
    const float wx[4] = {cc2(tx + 1.f, A), cc1(tx, A), cc1(1.f - tx, A), cc2(2.f - tx, A)};
    const float gm = gamma[0];
    const float* an = att + n * (long)hp * wp * C + c0;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), hp - 1);
        float row[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(ix - 1 + b, 0), wp - 1);
            const float* q = an + ((long)yy * wp + xx) * C;
            const f32x4 q0 = *(const f32x4*)q, q1 = *(const f32x4*)(q + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { row[e] += q0[e] * wx[b]; row[4 + e] += q1[e] * wx[b]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += row[e] * wy[a];
    }
    const long so = (c0 >> 5) * g + i * 32 + (c0 & 31);
    const f16x8 in8 = *(const f16x8*)(inp + so);
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (f16)(gm * v[e] + (float)in8[e]);
    *(f16x8*)(dst + so) = h;
}

// The same for W == 4 wp (the 4 x 4 max-pool of a width that is a multiple of 4): the four pixels X = 4 k + 2 .. 4 k + 5 read the SAME four att columns k - 1 .. k + 2
// (rx = X / 4 - 0.375: floor k for all four), so one thread takes the strip -- 32 cached 16-byte loads per four pixels instead of per pixel (the one-pixel form
// moved 2 GB through L1 / L2 for a 540 x 960 map: 77 us).  Per value the same products and sums in the same order as pan_fsa_combine.
__global__ __launch_bounds__(256) void pan_fsa_combine_x4(const float* att, int hp, int wp, int C, const f16* inp, long g, int N, int H, int W, const float* gamma, f16* dst) {
    const int c8 = ((C + 31) / 32 * 32) >> 3, ns = wp + 1;          // strips k = -1 .. wp - 1 of a row
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * H * ns * c8) return;
    const long r = t / c8;
    const int c0 = (int)(t - r * c8) * 8;
    const int k = (int)(r % ns) - 1, Y = (int)((r / ns) % H);
    const long n = r / ((long)ns * H);
    const int X0 = 4 * k + 2;
    const long row_i = (n * H + Y) * (long)W;
    if (c0 >= C) {                                   // pad channels of the last group: zeros (the up conv reads whole groups)
        f16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (f16)0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (X0 + j >= 0 && X0 + j < W) *(f16x8*)(dst + (c0 >> 5) * g + (row_i + X0 + j) * 32 + (c0 & 31)) = z;
        return;
    }
    const float A = -0.75f;
    const float sy = (float)hp / (float)H, sx = (float)wp / (float)W;
    const float ry = sy * ((float)Y + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry);
    const float ty = ry - (float)iy;
    const float wy[4] = {cc2(ty + 1.f, A), cc1(ty, A), cc1(1.f - ty, A), cc2(2.f - ty, A)};
    float wx[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float rx = sx * ((float)(X0 + j) + 0.5f) - 0.5f;
        const float tx = rx - (float)k;              // floor(rx) == k for the strip's four pixels
        wx[j][0] = cc2(tx + 1.f, A); wx[j][1] = cc1(tx, A); wx[j][2] = cc1(1.f - tx, A); wx[j][3] = cc2(2.f - tx, A);
    }
    const float gm = gamma[0];
    const float* an = att + n * (long)hp * wp * C + c0;
    float v[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), hp - 1);
        f32x4 q0[4], q1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(k - 1 + b, 0), wp - 1);
            const float* q = an + ((long)yy * wp + xx) * C;
            q0[b] = *(const f32x4*)q; q1[b] = *(const f32x4*)(q + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float row[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) { row[e] += q0[b][e] * wx[j][b]; row[4 + e] += q1[b][e] * wx[j][b]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] += row[e] * wy[a];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (X0 + j < 0 || X0 + j >= W) continue;
        const long so = (c0 >> 5) * g + (row_i + X0 + j) * 32 + (c0 & 31);
        const f16x8 in8 = *(const f16x8*)(inp + so);
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)(gm * v[j][e] + (float)in8[e]);
        *(f16x8*)(dst + so) = h;
    }
}

// F.interpolate(t, scale_factor=f, mode) on a blocked slab, f = 2 | 3.  bilinear (align_corners=False, ATen upsample_bilinear2d): source index
// max(0, (dst + 0.5) / f - 0.5), the second tap clamped to the last pixel, fp32 arithmetic in ATen's association; nearest: source dst / f.
// One thread per (output pixel, 8 channels) of `groups` 32-channel groups.
__global__ void pan_upsample(const f16* src, long sg, f16* dst, long dg, int groups, int N, int h, int w, int f, int bilinear) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H2 = f * h, W2 = f * w;
    const long total = (long)N * H2 * W2 * groups * 4;
    if (i >= total) return;
    const int q = (int)(i & 3);
    long r = i >> 2;
    const int grp = (int)(r % groups); r /= groups;
    const int x = (int)(r % W2), y = (int)((r / W2) % H2);
    const long n = r / ((long)W2 * H2);
    const f16* b = src + grp * sg + n * (long)h * w * 32 + q * 8;
    f16x8 o;
    if (!bilinear) {
        o = *(const f16x8*)(b + ((long)(y / f) * w + x / f) * 32);
    } else {
        const float inv = 1.0f / (float)f;
        const float sy = fmaxf(inv * ((float)y + 0.5f) - 0.5f, 0.f), sx = fmaxf(inv * ((float)x + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const f16x8 v00 = *(const f16x8*)(b + ((long)y0 * w + x0) * 32), v01 = *(const f16x8*)(b + ((long)y0 * w + x1) * 32);
        const f16x8 v10 = *(const f16x8*)(b + ((long)y1 * w + x0) * 32), v11 = *(const f16x8*)(b + ((long)y1 * w + x1) * 32);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = (f16)(hy * (hx * (float)v00[e] + lx * (float)v01[e]) + ly * (hx * (float)v10[e] + lx * (float)v11[e]));
    }
    *(f16x8*)(dst + grp * dg + ((n * H2 + y) * (long)W2 + x) * 32 + q * 8) = o;
}

// out = conv_last + bias + bilinear(x, align_corners=True) -> NCHW
// (rs == 0: raw is the planar fp32 [N][C][FH][FW] output of the halo-tile conv, bias already added)
__global__ void pan_final(const float* raw, int rs, const float* bias, int C, const void* x, int x_f32, int N, int H, int W,
                          int scale, void* out, int out_f32) {
    const int FH = H * scale, FW = W * scale;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * FH * FW) return;
    const int X = (int)(i % FW), Y = (int)((i / FW) % FH);
    const long n = i / ((long)FW * FH);
    const float sy = FH > 1 ? (float)(H - 1) / (float)(FH - 1) : 0.f, sx = FW > 1 ? (float)(W - 1) / (float)(FW - 1) : 0.f;
    const float fy = sy * (float)Y, fx = sx * (float)X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    for (int c = 0; c < C; ++c) {
        const long pl = (n * C + c) * (long)H * W;
        auto at = [&](int yy, int xx) {
            const long o = pl + (long)yy * W + xx;
            return x_f32 ? ((const float*)x)[o] : (float)((const f16*)x)[o];
        };
        float il;
        if (scale > 1)
            il = (1.f - ly) * ((1.f - lx) * at(y0, x0) + lx * at(y0, x1)) + ly * ((1.f - lx) * at(y1, x0) + lx * at(y1, x1));
        else
            il = at(Y, X);
        const long o = ((n * C + c) * FH + Y) * (long)FW + X;
        const float v = (rs ? raw[i * rs + c] + bias[c] : raw[o]) + il;
        if (out_f32) ((float*)out)[o] = v; else ((f16*)out)[o] = (f16)v;
    }
}

// The same for the planar raw buffer (rs == 0) when FW % 4 == 0: four consecutive pixels of one channel per thread -- one 16-byte read of the conv's result, the
// four bilinear samples of the input in pan_final's own arithmetic, one 8- / 16-byte store (the one-pixel-per-thread form moved 1.3 TB/s)
__global__ void pan_final_x4(const float* raw, int C, const void* x, int x_f32, int N, int H, int W, int scale, void* out, int out_f32) {
    const int FH = H * scale, FW = W * scale, FW4 = FW >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * C * FH * FW4) return;
    const int X0 = (int)(i % FW4) * 4, Y = (int)((i / FW4) % FH);
    const long nc = i / ((long)FW4 * FH);
    const float sy = FH > 1 ? (float)(H - 1) / (float)(FH - 1) : 0.f, sx = FW > 1 ? (float)(W - 1) / (float)(FW - 1) : 0.f;
    const float fy = sy * (float)Y;
    const int y0 = (int)fy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
    const float ly = fy - (float)y0;
    const long pl = nc * (long)H * W;
    auto at = [&](int yy, int xx) {
        const long o = pl + (long)yy * W + xx;
        return x_f32 ? ((const float*)x)[o] : (float)((const f16*)x)[o];
    };
    const long o = (nc * FH + Y) * (long)FW + X0;
    const f32x4 r = *(const f32x4*)(raw + o);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float il;
        if (scale > 1) {
            const float fx = sx * (float)(X0 + e);
            const int x0 = (int)fx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float lx = fx - (float)x0;
            il = (1.f - ly) * ((1.f - lx) * at(y0, x0) + lx * at(y0, x1)) + ly * ((1.f - lx) * at(y1, x0) + lx * at(y1, x1));
        } else {
            il = at(Y, X0 + e);
        }
        v[e] = r[e] + il;
    }
    if (out_f32) *(f32x4*)((float*)out + o) = f32x4{v[0], v[1], v[2], v[3]};
    else *(f16x4*)((f16*)out + o) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
}

struct Param { std::string key; std::vector<int> shape; std::vector<float> host; bool set = false; };

struct Gemm {                       // one packed GEMM
    int cin_pad = 32, cout = 0, ntaps = 1;
    f16* d_w = nullptr;
    std::function<float(int, int, int)> weight;        // (co, ci over the padded slab channels, tap) -> value
    int cin_used = 0;                                   // channels of the slab that may carry weights
    // plain 3x3 convs (bias, optional LeakyReLU / residual) run on the SR path's halo-tile kernel (conv3x3.hip) instead:
    bool tile3 = false; std::string bias_key;           // its packed panels [K3][cin_pad][3][3] and the bias padded with zeros
    void* d_w3 = nullptr; float* d_b3 = nullptr; int K3 = 0;
    std::function<int(int)> bias_row;                   // optional: output row -> index into the bias parameter (-1: none); default: row == index
    bool one_tap = false;                               // a 1x1 conv: single-tap panel (conv_pack_1x1), the kernel's one-tap instantiation
    void* d_fuse = nullptr;                             // conv_last behind a 32-channel HRconv: its panel for that conv's fused epilogue (conv_pack_fuse_last(.., cin = 32)); round 5
    void* d_gate_s = nullptr;                           // fp32 mode: the PA block's 1x1 conv as hi | lo self-gate fragments (4 KB) for the split up-conv's epilogue
    void* d_w3s = nullptr;                              // fp32 mode: the (wl | wh | wh) panels of conv3x3_pc's split-operand form (conv_pack_split / conv_pack_1x1_split), the HR side's convs
    bool pa_gate = false; void* d_gate = nullptr;       // the PA block's 1x1 conv (upsample.<i>.conv): also packed as the self-gate fragments of the conv in front of it (conv_pack_selfgate)
};

}  // namespace

struct innfer_pan {
    int in_nc = 3, out_nc = 3, nf = 40, unf = 24, nb = 16, scale = 4, n_up = 2;
    bool bilinear_up = false;              // ups_inter_mode 'bilinear': the up-blocks' Upsample(2) as its own pass (nearest rides in the conv's loader)
    bool self_attention = true;      // FSA after fea + trunk (PAN_arch.py:200-203)
    bool double_scpa = false;        // a second SCPA trunk + trunk_conv2 behind the first (PAN_arch.py:139-141,195-196)
    std::vector<Param> params;
    std::vector<Gemm> gemms;
    std::vector<float*> d_vecs;      // device copies of bias vectors / gamma, by param index (nullptr if unused)
    bool fp32 = false;               // innfer_pan_set_precision(1): the fp32 forward on NCHW fp32 tensors (f32ops.hip), the reference's -no_fp16 mode
    std::vector<float*> f32_w;       //   f32conv panels in forward order (pan_forward_f32 walks them)
    std::vector<void*> d_scpa;       // one weight blob per SCPA block (pan_scpa.hip), trunk by trunk
    std::vector<void*> d_scpa32;     // ... as (hi, lo) blob pairs for the fp32 mode's fused block (pan_scpa_split.hip): packed by innfer_pan_set_precision(1)
    int scpa_c8 = 1;                 // compact channel plane between the fused SCPA blocks (pan_scpa_launch in_c8 / out_c8); 0: A/B (fused_scpa 3)
    int mfma_attention = 1;          // the FSA block's attention on the matrix cores (pan_attention_mfma); 0: the VALU kernel of rounds 1-3 (set with fused_scpa: one A/B switch)
    int fused_last = 1;              // the last stage's HRconv with conv_last in its epilogue (conv3x3_pc FUSE on 32 channels, round 5); 0: two launches (fused_scpa 4)
    int scpa_duo = -1;               // fp16: an SCPA block on two 4-wave workgroups per CU, 8 x 32 tiles (pan_scpa_duo; innfer_pan_set_fused_scpa(pan, 6)); 0: one 8-wave workgroup, 16 x 32 (7); -1: the default = the two-workgroup form
    bool pa_two_launches = false;    // fp32 mode A/B (innfer_pan_set_fused_scpa(pan, 5)): the PA block as its own split 1x1 launch instead of the up-conv's epilogue
    int fused_scpa = 1;              // an SCPA block as ONE launch (innfer_pan_set_fused_scpa); 0: the five halo-tile launches of rounds 1-3
    bool uploaded = false;
};

static int P(innfer_pan* p, const std::string& key, std::vector<int> shape) {
    Param q; q.key = key; q.shape = shape;
    p->params.push_back(q);
    return (int)p->params.size() - 1;
}

extern "C" int innfer_pan_create(innfer_pan** out, int in_nc, int out_nc, int nf, int unf, int nb, int scale) {
    return innfer_pan_create_ex(out, in_nc, out_nc, nf, unf, nb, scale, 1, 0, 0);
}

extern "C" int innfer_pan_create_ex(innfer_pan** out, int in_nc, int out_nc, int nf, int unf, int nb, int scale, int self_attention, int double_scpa, int bilinear_up) {
    if (!out) return set_error(INNFER_ERR_INVALID, "pan_create: null out");
    if (nf != 40 || unf != 24 || in_nc < 1 || in_nc > 8 || out_nc < 1 || out_nc > 8 || nb < 1 ||
        (scale != 1 && scale != 2 && scale != 3 && scale != 4))
        return set_error(INNFER_ERR_UNSUPPORTED, "pan_create: nf=%d unf=%d scale=%d (built: nf 40, unf 24, scale 1/2/3/4)", nf, unf, scale);
    innfer_pan* p = new innfer_pan();
    p->in_nc = in_nc; p->out_nc = out_nc; p->nf = nf; p->unf = scale == 1 ? nf : unf; p->nb = nb; p->scale = scale;
    p->n_up = scale == 4 ? 2 : (scale == 1 ? 0 : 1);          // scale 3: ONE Upsample(scale_factor=3) stage (PAN_arch.py:112-114,164-166)
    p->self_attention = self_attention != 0; p->double_scpa = double_scpa != 0; p->bilinear_up = bilinear_up != 0;
    const int gw = nf / 2, UF = p->unf;
    P(p, "conv_first.weight", {nf, in_nc, 3, 3}); P(p, "conv_first.bias", {nf});
    for (int k = 0; k < (p->double_scpa ? 2 : 1); ++k) {          // registration order of PAN.__init__ (PAN_arch.py:134-141)
        const std::string sfx = k ? "2" : "";
        for (int b = 0; b < nb; ++b) {
            const std::string s = "SCPA_trunk" + sfx + "." + std::to_string(b) + ".";
            P(p, s + "conv1_a.weight", {gw, nf, 1, 1}); P(p, s + "conv1_b.weight", {gw, nf, 1, 1});
            P(p, s + "k1.0.weight", {gw, gw, 3, 3});
            P(p, s + "PACnv.k2.weight", {gw, gw, 1, 1}); P(p, s + "PACnv.k2.bias", {gw});
            P(p, s + "PACnv.k3.weight", {gw, gw, 3, 3}); P(p, s + "PACnv.k4.weight", {gw, gw, 3, 3});
            P(p, s + "conv3.weight", {nf, nf, 1, 1});
        }
        P(p, "trunk_conv" + sfx + ".weight", {nf, nf, 3, 3}); P(p, "trunk_conv" + sfx + ".bias", {nf});
    }
    if (p->self_attention) {
        P(p, "FSA.gamma", {1});
        P(p, "FSA.conv_f.weight", {nf / 8, nf, 1}); P(p, "FSA.conv_f.bias", {nf / 8});
        P(p, "FSA.conv_g.weight", {nf / 8, nf, 1}); P(p, "FSA.conv_g.bias", {nf / 8});
        P(p, "FSA.conv_h.weight", {nf, nf, 1}); P(p, "FSA.conv_h.bias", {nf});
    }
    for (int u = 0; u < p->n_up; ++u) {
        const int i = 5 * u, ci = u == 0 ? nf : UF;
        const std::string s = "upsample.";
        P(p, s + std::to_string(i + 1) + ".weight", {UF, ci, 3, 3}); P(p, s + std::to_string(i + 1) + ".bias", {UF});
        P(p, s + std::to_string(i + 2) + ".conv.weight", {UF, UF, 1, 1}); P(p, s + std::to_string(i + 2) + ".conv.bias", {UF});
        P(p, s + std::to_string(i + 4) + ".weight", {UF, UF, 3, 3}); P(p, s + std::to_string(i + 4) + ".bias", {UF});
    }
    P(p, "conv_last.weight", {out_nc, UF, 3, 3}); P(p, "conv_last.bias", {out_nc});
    *out = p;
    return INNFER_OK;
}

extern "C" void innfer_pan_destroy(innfer_pan* p) {
    if (!p) return;
    for (auto& g : p->gemms) { if (g.d_w) (void)hipFree(g.d_w); if (g.d_w3) (void)hipFree(g.d_w3); if (g.d_b3) (void)hipFree(g.d_b3); if (g.d_gate) (void)hipFree(g.d_gate); if (g.d_fuse) (void)hipFree(g.d_fuse); if (g.d_w3s) (void)hipFree(g.d_w3s); if (g.d_gate_s) (void)hipFree(g.d_gate_s); }
    for (auto v : p->d_vecs) if (v) (void)hipFree(v);
    for (auto v : p->d_scpa) if (v) (void)hipFree(v);
    for (auto v : p->d_scpa32) if (v) (void)hipFree(v);
    for (auto v : p->f32_w) if (v) (void)hipFree(v);
    delete p;
}

extern "C" int innfer_pan_set_fused_scpa(innfer_pan* p, int on) {
    if (!p) return set_error(INNFER_ERR_INVALID, "pan_set_fused_scpa: null network");
    p->fused_scpa = on ? 1 : 0;
    p->mfma_attention = on == 2 ? 0 : 1;          // (2: the fused trunk with the VALU attention -- A/B of the attention alone)
    p->scpa_c8 = on == 3 ? 0 : 1;                 // (3: the fused blocks on two-group slabs throughout -- A/B of the compact channel plane alone, same bits)
    p->fused_last = on == 4 ? 0 : 1;              // (4: HRconv and conv_last of the last stage as two launches -- A/B of the fused tail alone)
    if (!on) p->mfma_attention = 0;
    p->scpa_duo = on == 6 ? 1 : (on == 7 ? 0 : -1);      // (6 / 7: A/B of the block kernel's two forms -- two 4-wave workgroups per CU / one 8-wave workgroup; default: the two-workgroup form)
    p->pa_two_launches = on == 5;                 // (5, fp32 mode: the PA block of the HR side as its own launch -- A/B of the split self gate alone)
    return INNFER_OK;
}

extern "C" int innfer_pan_num_params(innfer_pan* p) { return p ? (int)p->params.size() : INNFER_ERR_INVALID; }

extern "C" int innfer_pan_param_info(innfer_pan* p, int idx, char* key, size_t key_cap, int* ndim, int* shape4) {
    if (!p || idx < 0 || idx >= (int)p->params.size()) return set_error(INNFER_ERR_INVALID, "pan_param_info: bad index");
    const Param& q = p->params[idx];
    if (key && key_cap) { strncpy(key, q.key.c_str(), key_cap - 1); key[key_cap - 1] = 0; }
    if (ndim) *ndim = (int)q.shape.size();
    if (shape4) for (size_t i = 0; i < 4; ++i) shape4[i] = i < q.shape.size() ? q.shape[i] : 1;
    return INNFER_OK;
}

extern "C" int innfer_pan_set_param(innfer_pan* p, int idx, const float* h_data) {
    if (!p || idx < 0 || idx >= (int)p->params.size() || !h_data) return set_error(INNFER_ERR_INVALID, "pan_set_param: bad arguments");
    Param& q = p->params[idx];
    size_t n = 1;
    for (int s : q.shape) n *= (size_t)s;
    q.host.assign(h_data, h_data + n);
    q.set = true;
    p->uploaded = false;
    return INNFER_OK;
}

namespace {

int find(const innfer_pan* p, const std::string& key) {
    for (size_t i = 0; i < p->params.size(); ++i) if (p->params[i].key == key) return (int)i;
    return -1;
}

// GEMM list in forward order; weights are looked up in the host copies when packing
int build_gemms(innfer_pan* p) {
    p->gemms.clear();
    const int nf = p->nf, gw = nf / 2, UF = p->unf;
    auto W = [p](const std::string& key) -> const std::vector<float>& { return p->params[find(p, key)].host; };
    auto add = [&](int cin_pad, int cout, int ntaps, std::function<float(int, int, int)> f, const char* tile3_bias = nullptr) {
        Gemm g; g.cin_pad = cin_pad; g.cout = cout; g.ntaps = ntaps; g.weight = f;
        if (tile3_bias) { g.tile3 = true; g.bias_key = tile3_bias; }           // "" = a halo-tile conv without bias
        p->gemms.push_back(g);
    };
    {   const auto& w = W("conv_first.weight"); const int ci_n = p->in_nc;
        add(32, nf, 9, [&w, ci_n](int co, int ci, int t) { return ci < ci_n ? w[((size_t)co * ci_n + ci) * 9 + t] : 0.f; }, "conv_first.bias"); }
    for (int k = 0; k < (p->double_scpa ? 2 : 1); ++k) {
    const std::string sfx = k ? "2" : "";
    for (int b = 0; b < p->nb; ++b) {
        const std::string s = "SCPA_trunk" + sfx + "." + std::to_string(b) + ".";
        const auto &wa = W(s + "conv1_a.weight"), &wb = W(s + "conv1_b.weight"), &k1 = W(s + "k1.0.weight"),
                   &k2 = W(s + "PACnv.k2.weight"), &k3 = W(s + "PACnv.k3.weight"), &k4 = W(s + "PACnv.k4.weight"),
                   &c3 = W(s + "conv3.weight");
        // branch a lives in channel group 0 (channels 0..gw-1), branch b in group 1 (32..32+gw-1): group-aligned halves let the
        // 3x3 convs of either branch run on the halo-tile kernel, which writes whole 32-channel groups
        // (the two 40-channel 1x1 convs of a block also run on the halo-tile kernel, as 3x3 convs with a centre tap only: the kernel is
        //  bound by its loads and stores at these widths, and its epilogue writes the fp16 slab that a GEMM + post pair needs two passes for)
        add(64, 32 + gw, 9, [&wa, &wb, nf, gw](int co, int ci, int t) {
            if (ci >= nf || t != 4) return 0.f;
            if (co < gw) return wa[(size_t)co * nf + ci];
            return co >= 32 ? wb[(size_t)(co - 32) * nf + ci] : 0.f; }, "");
        add(32, gw, 9, [&k1, gw](int co, int ci, int t) { return ci < gw ? k1[((size_t)co * gw + ci) * 9 + t] : 0.f; }, "");
        // PAConv's k3(b) * sigmoid(k2(b) + bias) as ONE 64-row conv with the pair-gate epilogue (ConvLaunch.act 7): a lane of the MFMA result holds rows
        // 16 lg + 4 t + j, its tiles t = 2, 3 gate its tiles t = 0, 1 -- so row 16 lg + r is k3's channel 8 lg + r for r < 8 and k2's channel
        // 8 lg + r - 8 (a 1x1 conv: centre tap) for r >= 8; one read of b instead of two launches and a round trip of k3(b)
        const std::string kb = s + "PACnv.k2.bias";
        add(32, 64, 9, [&k3, &k2, gw](int co, int ci, int t) {
            const int lg = co >> 4, r = co & 15, c = 8 * lg + (r & 7);
            if (ci >= gw || c >= gw) return 0.f;
            if (r < 8) return k3[((size_t)c * gw + ci) * 9 + t];
            return t == 4 ? k2[(size_t)c * gw + ci] : 0.f; }, kb.c_str());
        p->gemms.back().bias_row = [gw](int co) { const int r = co & 15, c = 8 * (co >> 4) + (r & 7); return (r >= 8 && c < gw) ? c : -1; };
        add(32, gw, 9, [&k4, gw](int co, int ci, int t) { return ci < gw ? k4[((size_t)co * gw + ci) * 9 + t] : 0.f; }, "");
        add(64, nf, 9, [&c3, nf, gw](int co, int ci, int t) {           // cat[a, b] = channels 0..gw-1 and 32..32+gw-1
            if (t != 4) return 0.f;
            if (ci < gw) return c3[(size_t)co * nf + ci];
            return ci >= 32 && ci < 32 + gw ? c3[(size_t)co * nf + gw + (ci - 32)] : 0.f; }, "");
    }
    {   const auto& w = W("trunk_conv" + sfx + ".weight");
        const std::string bk = "trunk_conv" + sfx + ".bias";
        add(64, nf, 9, [&w, nf](int co, int ci, int t) { return ci < nf ? w[((size_t)co * nf + ci) * 9 + t] : 0.f; }, bk.c_str()); }
    }
    if (p->self_attention) {
        const auto &wf = W("FSA.conv_f.weight"), &wg = W("FSA.conv_g.weight"), &wh = W("FSA.conv_h.weight");
        const int cq = nf / 8;
        add(64, 2 * cq + nf, 1, [&wf, &wg, &wh, nf, cq](int co, int ci, int) {
            if (ci >= nf) return 0.f;
            if (co < cq) return wf[(size_t)co * nf + ci];
            if (co < 2 * cq) return wg[(size_t)(co - cq) * nf + ci];
            return wh[(size_t)(co - 2 * cq) * nf + ci]; }); }
    for (int u = 0; u < p->n_up; ++u) {
        const int i = 5 * u, cin = u == 0 ? nf : UF;
        const auto &w1 = W("upsample." + std::to_string(i + 1) + ".weight"), &wp = W("upsample." + std::to_string(i + 2) + ".conv.weight"),
                   &w4 = W("upsample." + std::to_string(i + 4) + ".weight");
        const std::string b1 = "upsample." + std::to_string(i + 1) + ".bias", b4 = "upsample." + std::to_string(i + 4) + ".bias";
        add(u == 0 ? 64 : 32, UF, 9, [&w1, cin](int co, int ci, int t) { return ci < cin ? w1[((size_t)co * cin + ci) * 9 + t] : 0.f; }, b1.c_str());
        const std::string bp = "upsample." + std::to_string(i + 2) + ".conv.bias";
        add(32, UF, 9, [&wp, UF](int co, int ci, int t) { return ci < UF && t == 4 ? wp[(size_t)co * UF + ci] : 0.f; }, bp.c_str());
        p->gemms.back().pa_gate = true;
        add(32, UF, 9, [&w4, UF](int co, int ci, int t) { return ci < UF ? w4[((size_t)co * UF + ci) * 9 + t] : 0.f; }, b4.c_str());
    }
    {   const auto& w = W("conv_last.weight");
        add(p->n_up ? 32 : 64, p->out_nc, 9, [&w, UF](int co, int ci, int t) { return ci < UF ? w[((size_t)co * UF + ci) * 9 + t] : 0.f; }, "conv_last.bias"); }
    return INNFER_OK;
}

int upload(innfer_pan* p) {
    for (auto& q : p->params) if (!q.set) return set_error(INNFER_ERR_INVALID, "pan: parameter '%s' was never set", q.key.c_str());
    for (auto& g : p->gemms) {
        if (g.d_w) { (void)hipFree(g.d_w); g.d_w = nullptr; }
        if (g.d_w3) { (void)hipFree(g.d_w3); g.d_w3 = nullptr; }
        if (g.d_b3) { (void)hipFree(g.d_b3); g.d_b3 = nullptr; }
        if (g.d_gate) { (void)hipFree(g.d_gate); g.d_gate = nullptr; }
        if (g.d_fuse) { (void)hipFree(g.d_fuse); g.d_fuse = nullptr; }
        if (g.d_w3s) { (void)hipFree(g.d_w3s); g.d_w3s = nullptr; }
        if (g.d_gate_s) { (void)hipFree(g.d_gate_s); g.d_gate_s = nullptr; }
    }
    build_gemms(p);
    std::vector<f16> panel;
    for (auto& g : p->gemms) {
        if (g.tile3) {
            // slab outputs carry whole 32-channel groups (pad channels: zero weights and bias, so they stay zero); the planar last conv
            // keeps its own channel count
            g.K3 = g.cout <= 16 ? g.cout : (g.cout + 31) / 32 * 32;
            std::vector<float> w3((size_t)g.K3 * g.cin_pad * 9, 0.f), b3((size_t)(g.K3 + 63) / 64 * 64, 0.f);
            for (int co = 0; co < g.cout; ++co)
                for (int ci = 0; ci < g.cin_pad; ++ci)
                    for (int t = 0; t < 9; ++t) w3[((size_t)co * g.cin_pad + ci) * 9 + t] = g.weight(co, ci, t);
            if (!g.bias_key.empty()) {
                const std::vector<float>& hb = p->params[find(p, g.bias_key)].host;
                for (int co = 0; co < g.cout; ++co) {
                    const int bi = g.bias_row ? g.bias_row(co) : co;
                    if (bi >= 0) b3[co] = hb[bi];
                }
            }
            g.one_tap = g.K3 >= 32;                     // slab outputs only (the planar last conv keeps the 3x3 kernel)
            for (size_t i = 0; i < w3.size() && g.one_tap; ++i) if (i % 9 != 4 && w3[i] != 0.f) g.one_tap = false;
            std::vector<char> packed(g.one_tap ? conv_packed_bytes_taps(g.K3, g.cin_pad, 0x10) : conv_packed_bytes(g.K3, g.cin_pad));
            if (g.one_tap) {
                std::vector<float> w1((size_t)g.K3 * g.cin_pad);
                for (size_t i = 0; i < w1.size(); ++i) w1[i] = w3[i * 9 + 4];
                conv_pack_1x1(w1.data(), g.K3, g.cin_pad, packed.data());
                if (g.pa_gate && g.K3 == 32 && g.cin_pad == 32) {
                    std::vector<char> gp(2048);
                    conv_pack_selfgate(w1.data(), gp.data());
                    INNFER_HIP(hipMalloc(&g.d_gate, gp.size()));
                    INNFER_HIP(hipMemcpy(g.d_gate, gp.data(), gp.size(), hipMemcpyHostToDevice));
                }
            } else {
                conv_pack(w3.data(), g.K3, g.cin_pad, packed.data());
                if (g.K3 <= 3 && g.cin_pad == 32 && &g == &p->gemms.back()) {      // conv_last behind the last stage's HRconv (unf <= 32 channels): also as that conv's fused epilogue
                    std::vector<char> fp(4096);
                    conv_pack_fuse_last(w3.data(), g.K3, fp.data(), 0, 32);
                    INNFER_HIP(hipMalloc(&g.d_fuse, fp.size()));
                    INNFER_HIP(hipMemcpy(g.d_fuse, fp.data(), fp.size(), hipMemcpyHostToDevice));
                }
            }
            INNFER_HIP(hipMalloc(&g.d_w3, packed.size()));
            INNFER_HIP(hipMemcpy(g.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
            INNFER_HIP(hipMalloc((void**)&g.d_b3, b3.size() * sizeof(float)));
            INNFER_HIP(hipMemcpy(g.d_b3, b3.data(), b3.size() * sizeof(float), hipMemcpyHostToDevice));
            continue;
        }
        gg::pack_panels(panel, g.cout, g.cin_pad, g.cin_pad, g.ntaps, g.weight);
        INNFER_HIP(hipMalloc((void**)&g.d_w, panel.size() * sizeof(f16)));
        INNFER_HIP(hipMemcpy(g.d_w, panel.data(), panel.size() * sizeof(f16), hipMemcpyHostToDevice));
    }
    for (auto v : p->f32_w) if (v) (void)hipFree(v);                 // the fp32 panels follow the parameters: rebuilt by innfer_pan_set_precision
    p->f32_w.clear();
    for (auto v : p->d_scpa) if (v) (void)hipFree(v);
    p->d_scpa.clear();
    for (auto v : p->d_scpa32) if (v) (void)hipFree(v);
    p->d_scpa32.clear();
    {   // one blob per SCPA block for the fused launch (pan_scpa.hip): the block's eight tensors as MFMA fragments
        std::vector<char> blob(pan_scpa_blob_bytes());
        auto Wk = [p](const std::string& key) -> const float* { return p->params[find(p, key)].host.data(); };
        for (int k = 0; k < (p->double_scpa ? 2 : 1); ++k)
            for (int b = 0; b < p->nb; ++b) {
                const std::string s = "SCPA_trunk" + std::string(k ? "2" : "") + "." + std::to_string(b) + ".";
                pan_scpa_pack(Wk(s + "conv1_a.weight"), Wk(s + "conv1_b.weight"), Wk(s + "k1.0.weight"), Wk(s + "PACnv.k2.weight"), Wk(s + "PACnv.k2.bias"),
                              Wk(s + "PACnv.k3.weight"), Wk(s + "PACnv.k4.weight"), Wk(s + "conv3.weight"), blob.data());
                void* d = nullptr;
                INNFER_HIP(hipMalloc(&d, blob.size()));
                p->d_scpa.push_back(d);
                INNFER_HIP(hipMemcpy(d, blob.data(), blob.size(), hipMemcpyHostToDevice));
            }
    }
    for (auto v : p->d_vecs) if (v) (void)hipFree(v);
    p->d_vecs.assign(p->params.size(), nullptr);
    for (size_t i = 0; i < p->params.size(); ++i) {
        const Param& q = p->params[i];
        if (q.shape.size() != 1) continue;
        INNFER_HIP(hipMalloc((void**)&p->d_vecs[i], q.host.size() * sizeof(float)));
        INNFER_HIP(hipMemcpy(p->d_vecs[i], q.host.data(), q.host.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    p->uploaded = true;
    return INNFER_OK;
}

struct PCarve { size_t x0, fea, xa, xb, ab, ab2, k3y, inp, t, pool, fgh, att, aqk, avt, raw, hr[2][3], ups, total, slab_end; };

PCarve pcarve(const innfer_pan* p, int N, int H, int W) {
    PCarve c{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t px = (size_t)N * H * W, hp = H / 4, wp = W / 4, np = (size_t)N * hp * wp;
    size_t off = 0;
    auto slab = [&](size_t pixels, int groups) { size_t o = off; off += al(pixels * 32 * 2 * groups); return o; };
    c.x0 = slab(px, 1); c.fea = slab(px, 2); c.xa = slab(px, 2); c.xb = slab(px, 2); c.ab = slab(px, 2);
    c.ab2 = slab(px, 2); c.k3y = slab(px, 1); c.inp = slab(px, 2); c.t = slab(px, 2); c.pool = slab(np ? np : 1, 2);
    size_t m = 1;
    for (int u = 0; u < p->n_up; ++u) { m *= p->scale == 3 ? 9 : 4; for (int k = 0; k < 3; ++k) c.hr[u][k] = slab(px * m, 1); }
    // ups_inter_mode 'bilinear': the upsampled input of a stage (stage 0: 4 px of 2 groups; stage 1: 16 px of 1 group -- the larger of the two)
    // (nearest 3x has no place in the conv's loader either: it is materialised the same way)
    c.ups = ((p->bilinear_up || p->scale == 3) && p->n_up) ? (p->n_up == 2 ? slab(px * 16, 1) : slab(px * (p->scale == 3 ? 9 : 4), 2)) : 0;
    c.slab_end = off;
    c.fgh = off; off += al((np ? np : 1) * 64 * 4);
    c.att = off; off += al((np ? np : 1) * p->nf * 4);
    {   // MFMA attention: per image nblk 32-key blocks of {fh, fl', gh, gl'} rows (64 B per key) and of h fragments (3 KB per block)
        const size_t nblk = ((size_t)hp * wp + 31) / 32;
        c.aqk = off; off += al((size_t)N * nblk * 32 * 64 + 256);
        c.avt = off; off += al((size_t)N * nblk * 3072 + 256);
    }
    c.raw = off; off += al(px * m * (size_t)p->out_nc * 4);          // planar fp32 output of conv_last (every other conv writes fp16 slabs)
    c.total = off;
    return c;
}

}  // namespace

namespace {
// ---- the fp32 mode: PAN.forward on NCHW fp32 tensors with the generic fp32 ops (f32ops.hip); graph = oracle/nets.py pan_forward = PAN_arch.py:178-222 ----
struct PCarve32 { size_t fea, xa, xb, ab, cat, k3y, yv, inp, t, pool, fgh, att, aqk, avt, avl, hr[2][3], ups, raw, total; };
PCarve32 pcarve32(const innfer_pan* p, int N, int H, int W) {
    PCarve32 c{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t px = (size_t)N * H * W, hp = H / 4, wp = W / 4, np = (size_t)N * hp * wp, nf = p->nf, gw = nf / 2, UF = p->unf;
    size_t off = 0;
    auto buf = [&](size_t floats) { size_t o = off; off += al(floats * 4); return o; };
    c.fea = buf(px * nf); c.xa = buf(px * nf); c.xb = buf(px * nf); c.ab = buf(px * nf); c.cat = buf(px * nf); c.k3y = buf(px * gw); c.yv = buf(px * gw);
    c.inp = buf(px * nf); c.t = buf(px * nf); c.pool = buf((np ? np : 1) * nf); c.fgh = buf((np ? np : 1) * 64); c.att = buf((np ? np : 1) * nf);
    {   // the attention on the matrix cores (pan_attention_mfma<true>): per image nblk 32-key blocks of {fh, fl', gh, gl'} rows (64 B per key) and of h fragments (3 KB hi + 3 KB lo)
        const size_t nblk = (hp * wp + 31) / 32;
        c.aqk = buf((size_t)N * nblk * 32 * 16 + 64); c.avt = buf((size_t)N * nblk * 768 + 64); c.avl = buf((size_t)N * nblk * 768 + 64);
    }
    size_t m = 1, ups = 0;
    for (int u = 0; u < p->n_up; ++u) {
        const size_t cin = u == 0 ? nf : UF;
        m *= p->scale == 3 ? 9 : 4;
        ups = std::max(ups, px * m * cin);
        for (int k = 0; k < 3; ++k) c.hr[u][k] = buf(px * m * 32);          // (32 floats = 128 B per pixel: a one-group (hi, lo) slab pair of the split-operand HR side; the generic path uses UF floats of it)
    }
    c.ups = (p->bilinear_up || p->scale == 3) ? buf(ups) : 0;
    c.raw = buf(px * m * p->out_nc);
    c.total = off;
    return c;
}

int pan_forward_f32(innfer_pan* p, const float* x, float* y, int N, int H, int W, char* ws, hipStream_t s) {
    const PCarve32 cv = pcarve32(p, N, H, W);
    const int nf = p->nf, gw = nf / 2, UF = p->unf;
    const long hw = (long)H * W;
    auto B = [&](size_t o) { return (float*)(ws + o); };
    float *FEA = B(cv.fea), *XA = B(cv.xa), *XB = B(cv.xb), *AB = B(cv.ab), *CAT = B(cv.cat), *K3Y = B(cv.k3y), *YV = B(cv.yv), *INP = B(cv.inp), *T = B(cv.t);
    auto vec = [&](const std::string& key) -> const float* { const int i = find(p, key); return i >= 0 ? p->d_vecs[i] : nullptr; };
    int wi = 0;
    // conv of `ksz` x `ksz` taps (zero padding ksz / 2) over C input channels of a tensor with `ctot` channels at h x w (read through nearest-2x when `up`) -> K channels of a
    // tensor with `ktot` channels, starting at the pointer given
    auto conv = [&](const float* in, int ctot, int C, int h, int w, int up, int ksz, const float* bias, int K, float* out, int ktot, int act,
                    const float* res = nullptr, int rtot = 0, const float* mul = nullptr, int mtot = 0) -> int {
        F32Conv c{};
        const int ho = up ? 2 * h : h, wo = up ? 2 * w : w;
        c.in = in; c.in_nstride = (long)ctot * h * w; c.in_cstride = (long)h * w; c.C = C; c.Hin = h; c.Win = w;
        c.wp = p->f32_w[wi++]; c.bias = bias; c.K = K;
        c.out = out; c.out_nstride = (long)ktot * ho * wo; c.out_cstride = (long)ho * wo; c.out_pstride = 1; c.Wout = wo;
        c.Ho = ho; c.Wo = wo; c.osy = c.osx = 1; c.isy = c.isx = 1; c.up = up;
        c.ntap = ksz * ksz;
        for (int t = 0; t < c.ntap; ++t) { c.dy[t] = t / ksz - ksz / 2; c.dx[t] = t % ksz - ksz / 2; }
        c.act = act; c.N = N;
        c.res = res; c.res_nstride = (long)rtot * ho * wo; c.res_cstride = (long)ho * wo;
        c.mul = mul; c.mul_nstride = (long)mtot * ho * wo; c.mul_cstride = (long)ho * wo;
        return f32conv_launch(c, s);
    };
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    CK(conv(x, p->in_nc, p->in_nc, H, W, 0, 3, vec("conv_first.bias"), nf, FEA, nf, 0));
    const float* xc = FEA;
    for (int k = 0; k < (p->double_scpa ? 2 : 1); ++k) {
        const std::string sfx = k ? "2" : "";
        // the trunk's blocks as one launch each on (hi, lo) fp16 operand pairs (pan_scpa_split.hip; innfer_pan_set_fused_scpa(pan, 0): the six generic fp32 launches per block)
        if (p->fused_scpa && (int)p->d_scpa32.size() == (p->double_scpa ? 2 : 1) * p->nb && pan_scpa_split_ok(N, H, W)) {
            void *sa = XA, *sb = XB;
            const double px = (double)N * H * W;
            { GtScope gt(s, "pan split planes <-> NCHW fp32", 0.0, 320.0 * px); CK(pan_split_from_nchw(xc, sa, N, H, W, s)); }
            for (int b = 0; b < p->nb; ++b) {
                GtScope gt(s, "pan_scpa_split (one SCPA block, fp32 mode)", 2.0 * (2 * 20 * 40 + 3 * 9 * 20 * 20 + 20 * 20 + 40 * 40) * px, 320.0 * px + 57600.0);
                CK(pan_scpa_split_launch(sa, sb, p->d_scpa32[(size_t)k * p->nb + b], N, H, W, s));
                std::swap(sa, sb);
            }
            { GtScope gt(s, "pan split planes <-> NCHW fp32", 0.0, 320.0 * px); CK(pan_split_to_nchw(sa, AB, N, H, W, s)); }
            xc = AB;
            wi += 6 * p->nb;
        } else
        for (int b = 0; b < p->nb; ++b) {
            const std::string sb = "SCPA_trunk" + sfx + "." + std::to_string(b) + ".";
            float* xn = (b & 1) ? XB : XA;
            CK(conv(xc, nf, nf, H, W, 0, 1, nullptr, nf, AB, nf, 1));                                    // lrelu(conv1_a | conv1_b)
            CK(conv(AB, nf, gw, H, W, 0, 3, nullptr, gw, CAT, nf, 1));                                   // a = lrelu(k1(a))
            CK(conv(AB + gw * hw, nf, gw, H, W, 0, 3, nullptr, gw, K3Y, gw, 0));                         // k3(b)
            CK(conv(AB + gw * hw, nf, gw, H, W, 0, 1, vec(sb + "PACnv.k2.bias"), gw, YV, gw, 0, nullptr, 0, K3Y, gw));   // k3(b) * sigmoid(k2(b) + bias)
            CK(conv(YV, gw, gw, H, W, 0, 3, nullptr, gw, CAT + gw * hw, nf, 1));                          // b = lrelu(k4(.))
            CK(conv(CAT, nf, nf, H, W, 0, 1, nullptr, nf, xn, nf, 0, xc, nf));                           // conv3(cat[a, b]) + x
            xc = xn;
        }
        const std::string bk = "trunk_conv" + sfx + ".bias";
        if (p->double_scpa && k == 0) { CK(conv(xc, nf, nf, H, W, 0, 3, vec(bk), nf, INP, nf, 0)); xc = INP; }
        else CK(conv(xc, nf, nf, H, W, 0, 3, vec(bk), nf, p->double_scpa ? T : INP, nf, 0, FEA, nf));       // + fea
    }
    if (p->double_scpa) std::swap(INP, T);
    const float* cur = INP;
    if (p->self_attention) {
        const int hp = H / 4, wp = W / 4, Np = hp * wp;
        float *POOL = B(cv.pool), *FGH = B(cv.fgh), *ATT = B(cv.att);
        { GtScope gt(s, "f32 maxpool 4x4", 0.0, (double)N * nf * hw * 4.0 * (1.0 + 1.0 / 16)); CK(f32_maxpool4_launch(INP, POOL, (long)N * nf, H, W, s)); }
        {   // [f | g | h] = 1x1 convs of the pooled pixels, written as 64-float rows per pixel (biases are added by the attention kernel)
            F32Conv c{};
            c.in = POOL; c.in_nstride = (long)nf * Np; c.in_cstride = Np; c.C = nf; c.Hin = hp; c.Win = wp;
            c.wp = p->f32_w[wi++]; c.K = 2 * (nf / 8) + nf;
            c.out = FGH; c.out_nstride = (long)Np * 64; c.out_cstride = 1; c.out_pstride = 64; c.Wout = wp;
            c.Ho = hp; c.Wo = wp; c.osy = c.osx = 1; c.isy = c.isx = 1; c.ntap = 1; c.N = N;
            CK(f32conv_launch(c, s));
        }
        if (p->mfma_attention && nf == ATT_C) {      // softmax(f^T g) h on the matrix cores with (hi, lo) fp16 pairs of f, g, p and h (pan_attention_mfma<true>)
            const int nblk = (Np + ATT_KB - 1) / ATT_KB;
            GtScope gt(s, "pan_attention_mfma, fp32 mode (+ prep)", 2.0 * N * (double)Np * Np * (2 * 5 + 40), (double)N * Np * (64.0 + nf) * 4.0);
            const long work = (long)nblk * 192 > (long)nblk * ATT_KB ? (long)nblk * 192 : (long)nblk * ATT_KB;
            hipLaunchKernelGGL(pan_attn_prep, dim3((unsigned)((work + 255) / 256), N), dim3(256), 0, s, (const float*)FGH, vec("FSA.conv_f.bias"), vec("FSA.conv_g.bias"),
                               vec("FSA.conv_h.bias"), Np, nblk, (f16*)(ws + cv.aqk), (f16*)(ws + cv.avt), (f16*)(ws + cv.avl));
            hipLaunchKernelGGL(pan_attention_mfma<true>, dim3((Np + 127) / 128, N), dim3(512), 0, s, (const f16*)(ws + cv.aqk), (const f16*)(ws + cv.avt), (const f16*)(ws + cv.avl),
                               Np, nblk, ATT);
        } else {
            GtScope gt(s, "pan_attention (fp32 VALU)", 2.0 * N * (double)Np * Np * (2 * 5 + 40), (double)N * Np * (64.0 + nf) * 4.0);
            hipLaunchKernelGGL(pan_attention, dim3((Np + 63) / 64, N), dim3(256), 0, s, (const float*)FGH, vec("FSA.conv_f.bias"), vec("FSA.conv_g.bias"), vec("FSA.conv_h.bias"), Np, ATT);
        }
        INNFER_HIP(hipGetLastError());
        { GtScope gt(s, "f32 fsa_combine (bicubic + gamma * out + in)", 0.0, (double)N * nf * hw * 8.0); CK(f32_fsa_combine_launch(ATT, hp, wp, nf, INP, T, N, H, W, vec("FSA.gamma"), s, FGH)); }      // (FGH: dead behind the attention, 64 >= nf floats per pooled pixel)
        cur = T;
    }
    int h = H, w = W, cc = nf;
    float* RAW = B(cv.raw);
    // The HR side on the halo-tile kernel's split-operand form (conv3x3_pc SPLIT: (hi, lo) fp16 slab pairs, three MFMAs per product, fp32 accumulators): per stage
    // conv(nearest2x(.)) -> 1x1 conv with the PA gate as its epilogue -> HRconv, then conv_last -> planar fp32.  innfer_pan_set_fused_scpa(pan, 0), bilinear / 3x stages:
    // the generic fp32 convs below.
    const size_t g_first = p->gemms.size() - 1 - 3 * (size_t)p->n_up;
    bool split_hr = p->fused_scpa && p->n_up > 0 && !p->bilinear_up && p->scale != 3 && (long)N * H * W * (p->scale * p->scale) * 64 < 0x7fffffffL;
    for (size_t i = g_first; i < p->gemms.size() && split_hr; ++i) if (!p->gemms[i].d_w3s) split_hr = false;
    if (split_hr) {
        const long px = (long)N * H * W;
        f16* TS = (f16*)XA;                                     // (XA | XB are contiguous and free here: 320 B per pixel for the 256 of a two-group pair)
        {   GtScope gt(s, "pan NCHW fp32 -> (hi, lo) slab pair", 0.0, (double)px * (nf * 4.0 + 256.0));
            hipLaunchKernelGGL(pan_nchw_to_slab_pair, dim3((unsigned)((px * 8 + 255) / 256)), dim3(256), 0, s, cur, nf, hw, N, 2, TS, 2 * px * 32);
            INNFER_HIP(hipGetLastError()); }
        auto convs = [&](const Gemm& g, const f16* in, long in_g, long in_lo, int Ho, int Wo, int up, int act, const f16* res, f16* dst, float* planar, const Gemm* gate = nullptr) -> int {
            ConvLaunch L{};
            const long og = (long)N * Ho * Wo * 32;
            L.in = in; L.in_gstride = in_g; L.C = g.cin_pad;
            L.wpk = (const f16*)g.d_w3s; L.bias = g.d_b3;
            L.out = planar ? (void*)planar : (void*)dst; L.out_gstride = og; L.K = g.K3;
            L.N = N; L.H = Ho; L.W = Wo; L.act = act; L.up = up;
            L.res1 = res; L.res1_gstride = og; L.s1 = 1.f; L.s2 = 1.f;
            L.y0 = 0; L.y1 = Ho;
            L.out_mode = planar ? OUT_NCHW : OUT_SLAB; L.out_f32 = planar ? 1 : 0;
            L.conv1x1 = g.one_tap ? 1 : 0;
            L.split = 1; L.in_lo = in_lo; L.out_lo = og; L.res1_lo = og;
            if (gate) { L.gate_w = (const f16*)gate->d_gate_s; L.gate_bias = gate->d_b3; }      // out = act(v * sigmoid(W v + b)): the PA block as this conv's epilogue
            return conv_launch(L, s);
        };
        const f16* c16 = TS;
        long c_g = px * 32, c_lo = 2 * px * 32;
        size_t gi = g_first;
        for (int u = 0; u < p->n_up; ++u) {
            const int hh = 2 * h, ww = 2 * w;
            const long HG = (long)N * hh * ww * 32;
            f16 *V = (f16*)B(cv.hr[u][0]), *PAo = (f16*)B(cv.hr[u][1]), *HRC = (f16*)B(cv.hr[u][2]);
            if (p->gemms[gi + 1].d_gate_s && p->fused_scpa != 0 && !p->pa_two_launches)
                CK(convs(p->gemms[gi], c16, c_g, c_lo, hh, ww, 1, 1, nullptr, PAo, nullptr, &p->gemms[gi + 1]));  // conv(nearest2x(t)) with the PA block as its epilogue: V is never written
            else {
            CK(convs(p->gemms[gi], c16, c_g, c_lo, hh, ww, 1, 0, nullptr, V, nullptr));                          // conv(nearest2x(t))
            CK(convs(p->gemms[gi + 1], V, HG, HG, hh, ww, 0, 4, V, PAo, nullptr));                              // lrelu(v * sigmoid(conv1x1(v)))
            }
            CK(convs(p->gemms[gi + 2], PAo, HG, HG, hh, ww, 0, p->n_up == 1 ? 1 : 0, nullptr, HRC, nullptr));  // HRconv
            gi += 3;
            c16 = HRC; c_g = HG; c_lo = HG; h = hh; w = ww;
        }
        CK(convs(p->gemms[gi], c16, c_g, c_lo, h, w, 0, 0, nullptr, nullptr, RAW));                               // conv_last -> planar fp32 (+ bias)
    } else {
    for (int u = 0; u < p->n_up; ++u) {
        const int uf = p->scale == 3 ? 3 : 2, hh = uf * h, ww = uf * w, i = 5 * u;
        float *V = B(cv.hr[u][0]), *PAo = B(cv.hr[u][1]), *HRC = B(cv.hr[u][2]);
        const std::string s1 = "upsample." + std::to_string(i + 1) + ".bias", s2 = "upsample." + std::to_string(i + 2) + ".conv.bias", s4 = "upsample." + std::to_string(i + 4) + ".bias";
        if (p->bilinear_up || uf == 3) {
            float* UPS = B(cv.ups);
            CK(f32_upsample_launch(cur, UPS, (long)N * cc, h, w, uf, p->bilinear_up ? 1 : 0, s));
            CK(conv(UPS, cc, cc, hh, ww, 0, 3, vec(s1), UF, V, UF, 0));
        } else {
            CK(conv(cur, cc, cc, h, w, 1, 3, vec(s1), UF, V, UF, 0));                                   // conv(nearest2x(t))
        }
        CK(conv(V, UF, UF, hh, ww, 0, 1, vec(s2), UF, PAo, UF, 1, nullptr, 0, V, UF));                   // lrelu(v * sigmoid(conv1x1(v)))
        CK(conv(PAo, UF, UF, hh, ww, 0, 3, vec(s4), UF, HRC, UF, p->n_up == 1 ? 1 : 0));                 // HRconv (+ the LeakyReLU only in one-stage nets: see the fp16 path)
        cur = HRC; h = hh; w = ww; cc = UF;
    }
    CK(conv(cur, cc, cc, h, w, 0, 3, vec("conv_last.bias"), p->out_nc, RAW, p->out_nc, 0));
    }
    {
        const long fpx = (long)N * h * w;
        GtScope gt(s, "pan_final (+ bilinear skip, NCHW)", 0.0, (double)fpx * p->out_nc * 8.0);
        hipLaunchKernelGGL(pan_final, dim3((unsigned)((fpx + 255) / 256)), dim3(256), 0, s, (const float*)RAW, 0, (const float*)nullptr, p->out_nc, (const void*)x, 1, N, H, W, p->scale, (void*)y, 1);
        INNFER_HIP(hipGetLastError());
    }
#undef CK
    return INNFER_OK;
}
}  // namespace

// The reference's fp16 switch for this generator (run.py:345,421-422): fp32 = 1 runs PAN.forward in fp32 on NCHW fp32 tensors (csrc/f32ops.hip; the attention on the fp32
// VALU kernel) -- <= 1e-4 of the fp32 reference; input / output fp32.  A load-time call: it packs the fp32 panels of the parameters set so far.
extern "C" int innfer_pan_set_precision(innfer_pan* p, int fp32) {
    if (!p || (fp32 != 0 && fp32 != 1)) return set_error(INNFER_ERR_INVALID, "pan_set_precision: 0 (fp16 arithmetic) or 1 (fp32)");
    p->fp32 = fp32 != 0;
    if (!p->fp32) return INNFER_OK;
    if (!p->uploaded) { int rc = upload(p); if (rc) return rc; }
    if (!p->f32_w.empty()) return INNFER_OK;
    const int nf = p->nf, gw = nf / 2, UF = p->unf, cq = nf / 8;
    auto Wk = [p](const std::string& key) -> const std::vector<float>& { return p->params[find(p, key)].host; };
    std::vector<float> host;
    auto put = [&](int K, int C, int ntap, const std::function<float(int, int, int)>& w) -> int {
        host.resize(f32conv_packed_floats(K, C, ntap));
        f32conv_pack(K, C, ntap, w, host.data());
        float* d = nullptr;
        INNFER_HIP(hipMalloc((void**)&d, host.size() * sizeof(float)));
        p->f32_w.push_back(d);
        INNFER_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
        return INNFER_OK;
    };
    auto plain = [&](const std::string& key, int K, int C, int ksz) -> int {      // torch layout [K][C][ksz][ksz]
        const std::vector<float>& w = Wk(key);
        const int T = ksz * ksz;
        return put(K, C, T, [&w, C, T](int k, int c, int t) { return w[((size_t)k * C + c) * T + t]; });
    };
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    CK(plain("conv_first.weight", nf, p->in_nc, 3));
    for (int k = 0; k < (p->double_scpa ? 2 : 1); ++k) {
        const std::string sfx = k ? "2" : "";
        for (int b = 0; b < p->nb; ++b) {
            const std::string sb = "SCPA_trunk" + sfx + "." + std::to_string(b) + ".";
            const std::vector<float>&wa = Wk(sb + "conv1_a.weight"), &wb = Wk(sb + "conv1_b.weight");
            CK(put(nf, nf, 1, [&wa, &wb, nf, gw](int k2, int c, int) { return k2 < gw ? wa[(size_t)k2 * nf + c] : wb[(size_t)(k2 - gw) * nf + c]; }));
            CK(plain(sb + "k1.0.weight", gw, gw, 3));
            CK(plain(sb + "PACnv.k3.weight", gw, gw, 3));
            CK(plain(sb + "PACnv.k2.weight", gw, gw, 1));
            CK(plain(sb + "PACnv.k4.weight", gw, gw, 3));
            CK(plain(sb + "conv3.weight", nf, nf, 1));
        }
        CK(plain("trunk_conv" + sfx + ".weight", nf, nf, 3));
    }
    if (p->self_attention) {
        const std::vector<float>&wf = Wk("FSA.conv_f.weight"), &wg = Wk("FSA.conv_g.weight"), &wh = Wk("FSA.conv_h.weight");
        CK(put(2 * cq + nf, nf, 1, [&wf, &wg, &wh, nf, cq](int k2, int c, int) {
            return k2 < cq ? wf[(size_t)k2 * nf + c] : (k2 < 2 * cq ? wg[(size_t)(k2 - cq) * nf + c] : wh[(size_t)(k2 - 2 * cq) * nf + c]); }));
    }
    for (int u = 0; u < p->n_up; ++u) {
        const int i = 5 * u, cin = u == 0 ? nf : UF;
        CK(plain("upsample." + std::to_string(i + 1) + ".weight", UF, cin, 3));
        CK(plain("upsample." + std::to_string(i + 2) + ".conv.weight", UF, UF, 1));
        CK(plain("upsample." + std::to_string(i + 4) + ".weight", UF, UF, 3));
    }
    CK(plain("conv_last.weight", p->out_nc, UF, 3));
#undef CK
    {   // the HR side's convs (per stage: up-conv, the PA block's 1x1 conv, HRconv; conv_last) as split panels of the halo-tile kernel (conv3x3_pc SPLIT)
        const size_t first = p->gemms.size() - 1 - 3 * (size_t)p->n_up;
        for (size_t i = first; i < p->gemms.size(); ++i) {
            Gemm& g = p->gemms[i];
            if (!g.tile3 || g.d_w3s) continue;
            std::vector<float> w3((size_t)g.K3 * g.cin_pad * 9, 0.f);
            for (int co = 0; co < g.cout; ++co)
                for (int ci = 0; ci < g.cin_pad; ++ci)
                    for (int t = 0; t < 9; ++t) w3[((size_t)co * g.cin_pad + ci) * 9 + t] = g.weight(co, ci, t);
            std::vector<char> packed(3 * (g.one_tap ? conv_packed_bytes_taps(g.K3, g.cin_pad, 0x10) : conv_packed_bytes(g.K3, g.cin_pad)));
            if (g.one_tap) {
                std::vector<float> w1((size_t)g.K3 * g.cin_pad);
                for (size_t j = 0; j < w1.size(); ++j) w1[j] = w3[j * 9 + 4];
                conv_pack_1x1_split(w1.data(), g.K3, g.cin_pad, packed.data());
                if (g.pa_gate && g.K3 == 32 && g.cin_pad == 32) {
                    std::vector<float> wh(w1.size()), wl(w1.size());
                    for (size_t j = 0; j < w1.size(); ++j) { const f16 h = (f16)w1[j]; wh[j] = (float)h; wl[j] = (float)(f16)((w1[j] - (float)h) * 2048.0f); }
                    std::vector<char> gp(4096);
                    conv_pack_selfgate(wh.data(), gp.data());
                    conv_pack_selfgate(wl.data(), gp.data() + 2048);
                    INNFER_HIP(hipMalloc(&g.d_gate_s, gp.size()));
                    INNFER_HIP(hipMemcpy(g.d_gate_s, gp.data(), gp.size(), hipMemcpyHostToDevice));
                }
            } else conv_pack_split(w3.data(), g.K3, g.cin_pad, packed.data());
            INNFER_HIP(hipMalloc(&g.d_w3s, packed.size()));
            INNFER_HIP(hipMemcpy(g.d_w3s, packed.data(), packed.size(), hipMemcpyHostToDevice));
        }
    }
    {   // the SCPA blocks as (hi, lo) blob pairs for the fused split-operand launch (pan_scpa_split.hip)
        std::vector<char> blob(pan_scpa_split_blob_bytes());
        auto Wp = [p](const std::string& key) -> const float* { return p->params[find(p, key)].host.data(); };
        for (int k = 0; k < (p->double_scpa ? 2 : 1); ++k)
            for (int b = 0; b < p->nb; ++b) {
                const std::string s = "SCPA_trunk" + std::string(k ? "2" : "") + "." + std::to_string(b) + ".";
                pan_scpa_split_pack(Wp(s + "conv1_a.weight"), Wp(s + "conv1_b.weight"), Wp(s + "k1.0.weight"), Wp(s + "PACnv.k2.weight"), Wp(s + "PACnv.k2.bias"),
                                    Wp(s + "PACnv.k3.weight"), Wp(s + "PACnv.k4.weight"), Wp(s + "conv3.weight"), blob.data());
                void* d = nullptr;
                INNFER_HIP(hipMalloc(&d, blob.size()));
                p->d_scpa32.push_back(d);
                INNFER_HIP(hipMemcpy(d, blob.data(), blob.size(), hipMemcpyHostToDevice));
            }
    }
    return INNFER_OK;
}

extern "C" size_t innfer_pan_workspace_bytes(innfer_pan* p, int N, int H, int W) {
    if (!p || N <= 0 || H <= 0 || W <= 0) return 0;
    return p->fp32 ? pcarve32(p, N, H, W).total : pcarve(p, N, H, W).total;
}

extern "C" int innfer_pan_forward(innfer_pan* p, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                                  int N, int H, int W, void* d_ws, size_t ws_bytes, void* stream) {
    if (!p || !d_in || !d_out || !d_ws) return set_error(INNFER_ERR_INVALID, "pan_forward: null argument");
    if (N <= 0 || H <= 0 || W <= 0 || (p->self_attention && (H < 4 || W < 4)))
        return set_error(INNFER_ERR_INVALID, "pan_forward: input must be at least 4x4 (MaxPool2d(4))");
    if (!p->uploaded) { int rc = upload(p); if (rc) return rc; if (p->fp32) { rc = innfer_pan_set_precision(p, 1); if (rc) return rc; } }
    if (p->fp32) {
        if (in_dtype != INNFER_F32 || out_dtype != INNFER_F32) return set_error(INNFER_ERR_INVALID, "pan_forward: the fp32 mode takes and returns fp32 tensors");
        if (p->f32_w.empty()) return set_error(INNFER_ERR_INVALID, "pan_forward: call innfer_pan_set_precision(p, 1) after the last innfer_pan_set_param");
        const PCarve32 c32 = pcarve32(p, N, H, W);
        if (ws_bytes < c32.total) return set_error(INNFER_ERR_WORKSPACE, "pan_forward: workspace %zu < %zu bytes", ws_bytes, c32.total);
        return pan_forward_f32(p, (const float*)d_in, (float*)d_out, N, H, W, (char*)d_ws, (hipStream_t)stream);
    }
    const PCarve cv = pcarve(p, N, H, W);
    if (ws_bytes < cv.total) return set_error(INNFER_ERR_WORKSPACE, "pan_forward: workspace %zu < %zu bytes", ws_bytes, cv.total);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d_ws;
    // pad channels of every slab read as zero: every producer (pre, convs with zero pad weights / bias, MaxPool, FSA combine) writes whole
    // 32-channel groups, so the workspace needs no clearing
    const long px = (long)N * H * W, G = px * 32;
    const int nf = p->nf;
    float* raw = (float*)(ws + cv.raw);
    int dy9[9], dx9[9], d0[1] = {0};
    for (int t = 0; t < 9; ++t) { dy9[t] = t / 3 - 1; dx9[t] = t % 3 - 1; }
    int gi = 0;
    auto vec = [&](const std::string& key) { return p->d_vecs[find(p, key)]; };
    // raw GEMM rows hold only the layer's channels (rounded to 4), not the whole 64-channel tile
    int raw_rs = 64;
    auto gemm = [&](const f16* in, long in_g, int Hin, int Win, int Ho, int Wo, int up, bool full_rows = false) -> int {
        const Gemm& g = p->gemms[gi++];
        raw_rs = full_rows ? 64 : (g.cout + 3) / 4 * 4;
        return gg::launch(g.d_w, g.cin_pad, 64, in, in_g, N, Hin, Win, raw, Ho, Wo, 1, g.ntaps,
                          g.ntaps == 9 ? dy9 : d0, g.ntaps == 9 ? dx9 : d0, Ho, Wo, 1, 0, 0, up, s, nullptr, 0, raw_rs);
    };
    // plain 3x3 conv on conv3x3.hip: dst = act(conv(in) + bias) [+ res]; planar != nullptr: fp32 NCHW output instead of a slab
    auto conv3 = [&](const f16* in, long in_g, int Ho, int Wo, int up, int act, const f16* res, long res_g, f16* dst, long dst_g,
                     float* planar = nullptr, const Gemm* gate = nullptr, const Gemm* last = nullptr, float* last_out = nullptr) -> int {
        const Gemm& g = p->gemms[gi++];
        if (!g.tile3) return set_error(INNFER_ERR_INVALID, "pan: conv %d is not a halo-tile conv", gi - 1);
        ConvLaunch L{};
        L.in = in; L.in_gstride = in_g; L.C = g.cin_pad;
        L.wpk = (const f16*)g.d_w3; L.bias = g.d_b3;
        L.out = planar ? (void*)planar : (void*)dst; L.out_gstride = dst_g; L.K = g.K3;
        L.N = N; L.H = Ho; L.W = Wo; L.act = act; L.up = up;
        L.res1 = res; L.res1_gstride = res_g; L.s1 = 1.f; L.s2 = 1.f;
        L.y0 = 0; L.y1 = Ho;
        L.out_mode = planar ? OUT_NCHW : OUT_SLAB; L.out_f32 = planar ? 1 : 0;
        L.conv1x1 = g.one_tap ? 1 : 0;
        if (gate) { L.gate_w = (const f16*)gate->d_gate; L.gate_bias = gate->d_b3; }      // out = act(v * sigmoid(W v + b)): the PA block as this conv's epilogue
        if (last) {          // HRconv -> conv_last in one launch (conv3x3_pc FUSE on 32 channels): `dst` is never written -- its slab holds the rim buffer instead
            L.fuse_w = (const f16*)last->d_fuse; L.fuse_bias = last->d_b3; L.fuse_side = (float*)dst; L.fuse_out = last_out; L.fuse_oc = last->K3; L.fuse_out_mode = 1;
        }
        return conv_launch(L, s);
    };
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    f16 *X0 = (f16*)(ws + cv.x0), *FEA = (f16*)(ws + cv.fea), *XA = (f16*)(ws + cv.xa), *XB = (f16*)(ws + cv.xb),
        *AB = (f16*)(ws + cv.ab), *AB2 = (f16*)(ws + cv.ab2), *K3Y = (f16*)(ws + cv.k3y), *INP = (f16*)(ws + cv.inp),
        *T = (f16*)(ws + cv.t), *POOL = (f16*)(ws + cv.pool);

    {   GtScope gt(s, "pan_pre (NCHW -> slab)", 0.0, (double)px * (p->in_nc * (in_dtype == INNFER_F32 ? 4.0 : 2.0) + 64.0));
        hipLaunchKernelGGL(pan_pre, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, s, d_in, in_dtype == INNFER_F32, p->in_nc, (long)H * W, N, X0);
        INNFER_HIP(hipGetLastError()); }
    CK(conv3(X0, G, H, W, 0, 0, nullptr, 0, FEA, G));                                  // conv_first
    const f16* x = FEA;
    for (int k = 0; k < (p->double_scpa ? 2 : 1); ++k) {
    for (int b = 0; b < p->nb; ++b) {
        f16* xn = (b & 1) ? XB : XA;
        if (p->fused_scpa && (long)N * H * W * 64 + 2 * G < 0x7fffffffL) {          // the whole block in one launch (pan_scpa.hip)
            GtScope gt(s, "pan_scpa_fused (one SCPA block)", 2.0 * (2 * 20 * 40 + 3 * 9 * 20 * 20 + 20 * 20 + 40 * 40) * (double)px, 4.0 * 40 * (double)px + 28800.0);
            // between two blocks of a trunk the tensor's channels 32..39 travel as a compact 16-byte plane (80 instead of 128 bytes per pixel each way; the counters
            // showed 138 MB per launch against 83 MB algorithmic: profiles/r5/traffic_other.txt); the trunk's first block reads and its last writes the two-group slab
            CK(pan_scpa_launch(x, xn, G, p->d_scpa[(size_t)k * p->nb + b], N, H, W, s, p->scpa_c8 && b > 0, p->scpa_c8 && b + 1 < p->nb, p->scpa_duo));
            gi += 5;
            x = xn;
            continue;
        }
        CK(conv3(x, G, H, W, 0, 1, nullptr, 0, AB, G));                               // lrelu([conv1_a | . | conv1_b]): a -> group 0, b -> group 1
        CK(conv3(AB, G, H, W, 0, 1, nullptr, 0, AB2, G));                             // lrelu(k1(a)) -> cat group 0
        CK(conv3(AB + G, G, H, W, 0, 7, nullptr, 0, K3Y, G));                         // y = k3(b) * sigmoid(k2(b) + bias): one conv, pair-gate epilogue
        CK(conv3(K3Y, G, H, W, 0, 1, nullptr, 0, AB2 + G, G));                        // lrelu(k4(.)) -> cat group 1
        CK(conv3(AB2, G, H, W, 0, 0, x, G, xn, G));                                   // conv3(cat[a,b]) + x
        x = xn;
    }
    if (p->double_scpa && k == 0) {
        CK(conv3(x, G, H, W, 0, 0, nullptr, 0, INP, G));                               // trunk_conv; the second trunk starts from it
        x = INP;
    } else {
        CK(conv3(x, G, H, W, 0, 0, FEA, G, p->double_scpa ? T : INP, G));              // last trunk conv; + fea (INP still feeds the second trunk's first block)
    }
    }
    if (p->double_scpa) { f16* sw = INP; INP = T; T = sw; }                             // INP = fea + trunk from here on
    if (p->self_attention) {   // FSA
        const int hp = H / 4, wp = W / 4;
        const long np = (long)N * hp * wp, Gp = np * 32;
        {   GtScope gt(s, "pan_maxpool", 0.0, (double)px * nf * 2.0 + (double)np * 128.0);
            hipLaunchKernelGGL(pan_maxpool, dim3((unsigned)((np * ((nf + 31) / 32 * 4) + 255) / 256)), dim3(256), 0, s, INP, G, nf, N, H, W, hp, wp, POOL, Gp);
            INNFER_HIP(hipGetLastError()); }
        float* save = raw;
        raw = (float*)(ws + cv.fgh);
        CK(gemm(POOL, Gp, hp, wp, hp, wp, 0, true));                                  // [f | g | h], 64-float rows for pan_attention
        raw = save;
        if (p->mfma_attention && nf == ATT_C) {   // softmax(f^T g) h on the matrix cores: Np^2 (5 + 40) multiply-adds per image (the scores twice: max pass, accumulate pass)
            const int Np = hp * wp, nblk = (Np + ATT_KB - 1) / ATT_KB;
            GtScope gt(s, "pan_attention_mfma (+ prep)", 2.0 * N * (double)Np * Np * (2 * 5 + 40), (double)np * (64.0 + nf) * 4.0);
            const long work = (long)nblk * 192 > (long)nblk * ATT_KB ? (long)nblk * 192 : (long)nblk * ATT_KB;
            hipLaunchKernelGGL(pan_attn_prep, dim3((unsigned)((work + 255) / 256), N), dim3(256), 0, s, (const float*)(ws + cv.fgh), vec("FSA.conv_f.bias"),
                               vec("FSA.conv_g.bias"), vec("FSA.conv_h.bias"), Np, nblk, (f16*)(ws + cv.aqk), (f16*)(ws + cv.avt), (f16*)nullptr);
            hipLaunchKernelGGL(pan_attention_mfma<false>, dim3((Np + 127) / 128, N), dim3(512), 0, s, (const f16*)(ws + cv.aqk), (const f16*)(ws + cv.avt), (const f16*)nullptr, Np, nblk,
                               (float*)(ws + cv.att));
            INNFER_HIP(hipGetLastError());
        } else {
            GtScope gt(s, "pan_attention", 2.0 * N * (double)(hp * wp) * (hp * wp) * (2 * 5 + 40), (double)np * (64.0 + nf) * 4.0);
            hipLaunchKernelGGL(pan_attention, dim3((hp * wp + 63) / 64, N), dim3(256), 0, s, (const float*)(ws + cv.fgh), vec("FSA.conv_f.bias"),
                               vec("FSA.conv_g.bias"), vec("FSA.conv_h.bias"), hp * wp, (float*)(ws + cv.att));
            INNFER_HIP(hipGetLastError()); }
        {   GtScope gt(s, "pan_fsa_combine (bicubic + gamma * out + in)", 0.0, (double)px * nf * 4.0 + (double)np * nf * 4.0);
            if (W == 4 * wp)
                hipLaunchKernelGGL(pan_fsa_combine_x4, dim3((unsigned)(((long)N * H * (wp + 1) * ((nf + 31) / 32 * 4) + 255) / 256)), dim3(256), 0, s, (const float*)(ws + cv.att), hp, wp, nf,
                                   INP, G, N, H, W, vec("FSA.gamma"), T);
            else
            hipLaunchKernelGGL(pan_fsa_combine, dim3((unsigned)((px * ((nf + 31) / 32 * 4) + 255) / 256)), dim3(256), 0, s, (const float*)(ws + cv.att), hp, wp, nf,
                               INP, G, N, H, W, vec("FSA.gamma"), T);
            INNFER_HIP(hipGetLastError()); }
    }
    const f16* cur = p->self_attention ? T : INP;
    long cur_g = G;
    int h = H, w = W;
    bool fused_last = false;
    for (int u = 0; u < p->n_up; ++u) {
        const int uf = p->scale == 3 ? 3 : 2;
        const int hh = uf * h, ww = uf * w;
        const long hpx = (long)N * hh * ww, HG = hpx * 32;
        f16 *V = (f16*)(ws + cv.hr[u][0]), *PA = (f16*)(ws + cv.hr[u][1]), *HRC = (f16*)(ws + cv.hr[u][2]);
        // PA (x * sigmoid(conv1x1(x)) -> LeakyReLU) as the epilogue of the conv in front of it: V is never written (fused_scpa != 0; same values to the last fp16 rounding)
        const Gemm* gate = (p->fused_scpa && p->gemms[gi + 1].d_gate) ? &p->gemms[gi + 1] : nullptr;
        f16* Vd = gate ? PA : V;
        const int va = gate ? 1 : 0;
        if (p->bilinear_up || uf == 3) {                                               // conv(upsampled(t)): the upsampling as its own pass
            f16* UPS = (f16*)(ws + cv.ups);
            const int groups = u == 0 ? 2 : 1;                                         // nf = 40 / unf = 24 channels (pad channels stay zero: 0 interpolates to 0)
            const long tot = hpx * groups * 4;
            hipLaunchKernelGGL(pan_upsample, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, cur, cur_g, UPS, HG, groups, N, h, w, uf, p->bilinear_up ? 1 : 0);
            INNFER_HIP(hipGetLastError());
            CK(conv3(UPS, HG, hh, ww, 0, va, nullptr, 0, Vd, HG, nullptr, gate));
        } else
        CK(conv3(cur, cur_g, hh, ww, 1, va, nullptr, 0, Vd, HG, nullptr, gate));       // conv(nearest2x(t)) [+ PA]
        if (gate) ++gi;                                                                // (the 1x1 conv ran inside that launch)
        else CK(conv3(V, HG, hh, ww, 0, 4, V, HG, PA, HG));                            // PA: lrelu(v * sigmoid(conv1x1(v)))
        // HRconv.  Two stages (4x): PAN's outer B.sequential flattens the stages with children(), which yields the shared LeakyReLU once per
        // stage -- nothing follows HRconv.  One stage (2x): the stage's own nn.Sequential is used as it is and holds the LeakyReLU in two slots,
        // so HRconv IS followed by it (PAN_arch.py:11-19, block.py:197-210; golden G18)
        // (round 5) the last stage's HRconv carries conv_last in its epilogue where the grid is whole 16 x 32 tiles: the 24-channel tensor at full resolution (531 MB of slab
        // at 2160 x 3840) is neither written nor read back; the rim sums live in its place
        const Gemm& gl = p->gemms.back();
        fused_last = u == p->n_up - 1 && gl.d_fuse && p->fused_scpa && p->fused_last && hh % 16 == 0 && ww % 32 == 0 && conv_fuse_side_bytes(N, hh, ww) <= (size_t)HG * 2 &&
                     (long)N * (hh / 16) * (ww / 32) * 92 < 0x7fffffffL;
        if (fused_last) CK(conv3(PA, HG, hh, ww, 0, p->n_up == 1 ? 1 : 0, nullptr, 0, HRC, HG, nullptr, nullptr, &gl, raw));
        else CK(conv3(PA, HG, hh, ww, 0, p->n_up == 1 ? 1 : 0, nullptr, 0, HRC, HG));
        cur = HRC; cur_g = HG; h = hh; w = ww;
    }
    // (round 5: the bilinear skip as conv_last's own store was built and measured -- 168 + 72 us -> 585 us at 540 x 960: the 16 lanes that hold the three planar channels
    //  gather 72 values per tile on the consumer waves' critical path; reverted, profiles/r5/pan_tail.txt)
    if (fused_last) ++gi;                                                              // (conv_last ran inside the HRconv launch)
    else
    CK(conv3(cur, cur_g, h, w, 0, 0, nullptr, 0, nullptr, 0, raw));                    // conv_last -> planar fp32 (+ bias)
    {
        const long fpx = (long)N * h * w;
        GtScope gt(s, "pan_final (+ bilinear skip, NCHW)", 0.0, (double)fpx * p->out_nc * (4.0 + (out_dtype == INNFER_F32 ? 4.0 : 2.0)));
        if (w % 4 == 0) {
            const long nthr = fpx * p->out_nc / 4;
            hipLaunchKernelGGL(pan_final_x4, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (const float*)raw, p->out_nc, d_in, in_dtype == INNFER_F32, N, H, W, p->scale,
                               d_out, out_dtype == INNFER_F32);
        } else
        hipLaunchKernelGGL(pan_final, dim3((unsigned)((fpx + 255) / 256)), dim3(256), 0, s, raw, 0, vec("conv_last.bias"), p->out_nc,
                           d_in, in_dtype == INNFER_F32, N, H, W, p->scale, d_out, out_dtype == INNFER_F32);
        INNFER_HIP(hipGetLastError());
    }
#undef CK
    return INNFER_OK;
}
