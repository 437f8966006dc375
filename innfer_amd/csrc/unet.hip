// pix2pix UNet generator (UnetGenerator, norm=batch, deconv) on gfx950 -- BASELINE config 5.
//
// Replaces UNet_arch.py:11-161 as run.py runs it (meval=False: BatchNorm normalises with the
// statistics of the CURRENT image, run.py:299-303,98-99; SURVEY.md D6: a batch is N independent
// batch-1 forwards, so statistics are per image).
//
//   4x4 stride-2 pad-1 conv           -> gather GEMM on MFMA (gather_gemm.h): 16 taps x Cin/32 chunks; the outermost one (3 input
//                                        channels) reads a 64-channel patch slab at half resolution instead (unet_pre_patch, 1 tap)
//   4x4 stride-2 pad-1 ConvTranspose  -> four output phases (oy&1, ox&1) of 2x2 taps each, ONE grouped launch (GP.g_phase) unless the
//                                        layer is split over K; the outermost one (128 -> 3, bias, tanh) is ONE 3x3 conv with 4*3 phase
//                                        channels on the SR path's halo-tile kernel (conv3x3.hip, planar epilogue with the phase scatter)
//   BatchNorm2d (training mode)       -> norm_stats.h: per-(image, channel) mean / biased variance over H*W in fp32 from one read of
//                                        the fp32 conv output, eps 1e-5
//   LeakyReLU(0.2, inplace) / ReLU / torch.cat / Tanh -> the "post" kernel writes every consumer's
//       view of a tensor directly: the in-place LeakyReLU at the head of each block also rewrites the
//       skip branch (UNet_arch.py:109,160), and the parent's in-place ReLU acts on the concatenation,
//       so a down-path tensor t is consumed as lrelu(t) by the next down conv and as
//       relu(lrelu(t)) = relu(t) by the up conv: two fp16 slabs, no activation on load.
//
// Activations are blocked-NHWC fp16 like the SR path; GEMM outputs stay fp32 until normalised.  Deep layers (<= 64 output
// pixels per image) are split over K; whether a layer is split never depends on the batch, so a batch equals its batch-1 forwards.
#include "common.h"
#include "gather_gemm.h"
#include "norm_stats.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace innfer;

namespace {

struct PostDst { f16* p; long g; int coff; int act; };      // act: 1 lrelu(0.2), 2 relu

// raw fp32 -> (BatchNorm) -> activation -> up to two fp16 blocked-NHWC destinations; 8 channels per thread
// alpha / shift: [N][C] (train mode: statistics of each image, nstride = C) or [C] (eval mode: running statistics, nstride = 0)
__global__ void unet_post(const float* raw, int cpad, int C, long HW, int N, const float* alpha, const float* shift, int nstride,
                          PostDst d0, PostDst d1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C / 8;
    if (i >= (long)N * HW * c8) return;
    const int c = (int)(i % c8) * 8;
    const long pix = i / c8;                                  // n*HW + px
    const long n = pix / HW;
    const float* rp = raw + pix * cpad + c;
    float v[8];
    const f32x4 x0 = *(const f32x4*)rp, x1 = *(const f32x4*)(rp + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = x0[e]; v[4 + e] = x1[e]; }
    if (alpha) {
        const float* ap = alpha + n * nstride + c;
        const float* sp = shift + n * nstride + c;
        const f32x4 a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 4), s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v[e] * a0[e] + s0[e]; v[4 + e] = v[4 + e] * a1[e] + s1[e]; }
    }
    const PostDst ds[2] = {d0, d1};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!ds[k].p) continue;
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)(ds[k].act == 1 ? fmaxf(v[e], 0.2f * v[e]) : fmaxf(v[e], 0.f));
        const int ch = ds[k].coff + c;
        *(f16x8*)(ds[k].p + (ch >> 5) * ds[k].g + pix * 32 + (ch & 31)) = h;
    }
}

// The same behind a halo-tile conv (upsample_mode 'upconv'): fp16 conv result (slab) -> norm -> activation -> ONE fp16 destination at a channel offset
__global__ void unet_post_slab(const f16* src, long sg, int C, long HW, int N, const float* alpha, const float* shift, int nstride, PostDst d, PostDst d1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C / 8;
    if (i >= (long)N * HW * c8) return;
    const int c = (int)(i % c8) * 8;
    const long pix = i / c8;
    const long n = pix / HW;
    const f16x8 x = *(const f16x8*)(src + (c >> 5) * sg + pix * 32 + (c & 31));
    const float* ap = alpha + n * nstride + c;
    const float* sp = shift + n * nstride + c;
    f16x8 h, h1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = (float)x[e] * ap[e] + sp[e];
        h[e] = (f16)(d.act == 1 ? fmaxf(v, 0.2f * v) : fmaxf(v, 0.f));
        h1[e] = (f16)(d1.act == 1 ? fmaxf(v, 0.2f * v) : fmaxf(v, 0.f));
    }
    const int ch = d.coff + c;
    *(f16x8*)(d.p + (ch >> 5) * d.g + pix * 32 + (ch & 31)) = h;
    if (d1.p) {
        const int c1 = d1.coff + c;
        *(f16x8*)(d1.p + (c1 >> 5) * d1.g + pix * 32 + (c1 & 31)) = h1;
    }
}

// The same with the statistics merge inside (as resnet.hip's rn_post_slab_parts): a workgroup = pxb pixels x 32 channels of one image; it merges that
// image's per-tile records of its 32 channels itself (8 lanes per channel, Chan's update, lanes merged in order: every workgroup of the group
// computes the same alpha / shift) and applies them -- the combine launch between conv and post is gone.
__global__ __launch_bounds__(256) void unet_post_slab_parts(const f16* src, long sg, int C, long HW, const float* part, int nper, float eps,
                                                            const float* gamma, const float* beta, PostDst d, PostDst d1, int pxb) {
    __shared__ float sn[8][32], sm[8][32], sq[8][32], sal[32], ssh[32];
    const int n = blockIdx.z, cb = blockIdx.y * 32, t = threadIdx.x;
    {
        const int cl = t & 31, lg = t >> 5, c = cb + cl;
        float cnt = 0.f, mu = 0.f, m2 = 0.f;
        // (four records per trip, their loads issued together: one record per trip was one exposed load latency per record -- 32 trips for the 256 records of the
        //  128 x 128 level, repeated by each of its 16 pixel blocks: that launch ran at 3.2 TB/s against 4.6 for its siblings.  Same merge order: same bits.)
        for (int r = lg; r < nper; r += 32) {
            float rec[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rr = r + 8 * j;
                const float* q = part + (((long)n * nper + min(rr, nper - 1)) * C + c) * 3;
                rec[j][0] = rr < nper ? q[0] : 0.f; rec[j][1] = q[1]; rec[j][2] = q[2];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) norm::chan_merge(cnt, mu, m2, rec[j][0], rec[j][1], rec[j][2]);      // (count 0: no-op)
        }
        sn[lg][cl] = cnt; sm[lg][cl] = mu; sq[lg][cl] = m2;
        __syncthreads();
        if (lg == 0) {
            for (int i = 1; i < 8; ++i) norm::chan_merge(cnt, mu, m2, sn[i][cl], sm[i][cl], sq[i][cl]);
            const float a = (1.0f / sqrtf(m2 / (float)HW + eps)) * (gamma ? gamma[c] : 1.0f);
            sal[cl] = a;
            ssh[cl] = (beta ? beta[c] : 0.f) - mu * a;
        }
        __syncthreads();
    }
    const int q8 = (t & 3) * 8, pl = t >> 2;
    float al[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { al[e] = sal[q8 + e]; sh[e] = ssh[q8 + e]; }
    const int c0 = d.coff + cb + q8, c1 = d1.coff + cb + q8;
    const f16* sp = src + (cb >> 5) * sg + q8;
    f16* o0 = d.p + (c0 >> 5) * d.g + (c0 & 31);
    f16* o1 = d1.p ? d1.p + (c1 >> 5) * d1.g + (c1 & 31) : nullptr;
    auto apply = [&](const f16x8 x, long pix) __attribute__((always_inline)) {
        f16x8 h, h1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (float)x[e] * al[e] + sh[e];
            h[e] = (f16)(d.act == 1 ? fmaxf(v, 0.2f * v) : fmaxf(v, 0.f));
            h1[e] = (f16)(d1.act == 1 ? fmaxf(v, 0.2f * v) : fmaxf(v, 0.f));
        }
        *(f16x8*)(o0 + pix * 32) = h;
        if (o1) *(f16x8*)(o1 + pix * 32) = h1;
    };
    const long px0 = (long)blockIdx.x * pxb + pl, pixb = (long)n * HW + px0;
    if ((long)(blockIdx.x + 1) * pxb <= HW) {
        // whole block: four loads in flight per lane before the first is touched (a loop with a bounds exit keeps ONE in flight -- the compiler cannot hoist a load
        // above the exit test -- and the pass then ran at 2.5 TB/s on the 64^2 level)
        for (int i = 0; i < pxb / 64; i += 4) {
            f16x8 x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = *(const f16x8*)(sp + (pixb + 64 * (i + j)) * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) apply(x[j], pixb + 64 * (i + j));
        }
    } else {
        for (int i = 0; i < pxb / 64; ++i) {
            if (px0 + 64 * i >= HW) break;
            apply(*(const f16x8*)(sp + (pixb + 64 * i) * 32), pixb + 64 * i);
        }
    }
}
static int launch_post_slab_parts(const f16* src, long sg, int C, long HW, int N, const float* part, int nper, const float* gamma, const float* beta,
                                  PostDst d, PostDst d1, hipStream_t s) {
    // (4096 pixels per workgroup where that still leaves >= 4 workgroups per CU: the statistics merge in front is repeated by every pixel block of an image)
    const int pxb = ((HW + 4095) / 4096) * (C / 32) * N >= 1024 ? 4096 : ((HW + 1023) / 1024) * (C / 32) * N >= 256 ? 1024 : 256;
    GtScope gt(s, "unet_post_slab_parts (BatchNorm + views from the fp16 slab)", 0.0, (double)N * HW * C * 2.0 * (1 + (d.p ? 1 : 0) + (d1.p ? 1 : 0)));
    hipLaunchKernelGGL(unet_post_slab_parts, dim3((unsigned)((HW + pxb - 1) / pxb), C / 32, N), dim3(256), 0, s, src, sg, C, HW, part, nper, 1e-5f,
                       gamma, beta, d, d1, pxb);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

// Deep levels (at most DEEP_PX output pixels per image): split-K reduction + BatchNorm statistics + normalisation / activation in ONE
// launch.  A workgroup owns 32 channels of one image: 32 channel lanes x 32 pixel lanes, every thread keeps its <= DEEP_PX/32 pixels in
// registers; the partial results are added in segment order, the statistics are the two-pass form on registers (as norm_stats.h), the
// result goes to up to two fp16 destinations like unet_post.  part: ks segments of split_elems floats (ks == 1: the GEMM result itself).
// (round 3: up to 256 pixels -- eight per thread -- so that the 8x8 -> 16x16 ConvTranspose level of a 256 x 256 input no longer takes a reduce, a statistics
//  and a post launch over its fp32 rows: unet_deep_post<8>; the <= 64-pixel levels keep the two-pixel form)
constexpr int DEEP_PX = 256;
// ev_alpha / ev_shift != nullptr: eval-mode BatchNorm (the per-channel transform of the running statistics) instead of this image's.
// (round 5: PL pixel lanes -- 32, or 16 / 4 / 1 for the levels of at most 16 / 4 / 1 pixels: a 1024-thread workgroup per 32 channels of a 1 x 1 level kept 31 of its 32
//  pixel lanes idle and still paid its two 32-way reductions: 12.7 us a launch whatever the level.  One pixel per lane in the same lane order: the same sums, the same bits)
template <int NPX, int PL = 32>
__global__ __launch_bounds__(32 * PL) void unet_deep_post(const float* part, long split_elems, int ks, int cpad, int C, int HW, float eps,
                                                       const float* gamma, const float* beta, const float* ev_alpha, const float* ev_shift,
                                                       PostDst d0, PostDst d1) {
    __shared__ float red[32 * PL];
    const int n = blockIdx.y, cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    auto reduce32 = [&](float v) {
        if constexpr (PL == 1) return 0.f + v;                      // (t = 0 + red[cl], as the loop below would)
        red[threadIdx.x] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < PL; ++i) t += red[cl + 32 * i];        // (lanes beyond PL held pixels >= HW: zeros)
        __syncthreads();
        return t;
    };
    float v[NPX];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int px = pl + PL * i;
        float a = 0.f;
        if (px < HW && c < C) {
            const float* q = part + ((long)n * HW + px) * cpad + c;
            a = q[0];
            int z = 1;
            for (; z + 8 <= ks; z += 8) {           // eight loads in flight, added in segment order (the order never depends on how they were fetched)
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = q[(long)(z + j) * split_elems];
#pragma unroll
                for (int j = 0; j < 8; ++j) a += t[j];
            }
            for (; z < ks; ++z) a += q[z * split_elems];
        }
        v[i] = a;
        sum += a;
    }
    float al = 1.f, sh = 0.f;
    if (ev_alpha) {
        if (c < C) { al = ev_alpha[c]; sh = ev_shift[c]; }
    } else if (gamma) {                              // train-mode BatchNorm of this image (uniform branch)
        const float mu = reduce32(sum) / (float)HW;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < NPX; ++i) { const float d = pl + PL * i < HW ? v[i] - mu : 0.f; m2 += d * d; }
        const float var = reduce32(m2) / (float)HW;
        if (c < C) { al = (1.0f / sqrtf(var + eps)) * gamma[c]; sh = beta[c] - mu * al; }
    }
    if (c >= C) return;
    const PostDst ds[2] = {d0, d1};
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int px = pl + PL * i;
        if (px >= HW) continue;
        const float y = (gamma || ev_alpha) ? v[i] * al + sh : v[i];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!ds[k].p) continue;
            const int ch = ds[k].coff + c;
            ds[k].p[(ch >> 5) * ds[k].g + ((long)n * HW + px) * 32 + (ch & 31)] = (f16)(ds[k].act == 1 ? fmaxf(y, 0.2f * y) : fmaxf(y, 0.f));
        }
    }
}

// outermost: raw[.., 0:C] + bias -> tanh -> NCHW
__global__ void unet_final(const float* raw, int cpad, int C, long HW, int N, const float* bias, void* out, int out_f32) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * HW) return;
    const long n = i / HW, px = i % HW;
    for (int c = 0; c < C; ++c) {
        const float y = fast_tanh(raw[i * cpad + c] + bias[c]);
        const long o = (n * C + c) * HW + px;
        if (out_f32) ((float*)out)[o] = y; else ((f16*)out)[o] = (f16)y;
    }
}

__global__ void unet_pre(const void* in, int in_f32, int C, long HW, int N, f16* slab) {    // NCHW -> one zero-padded group
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * HW) return;
    const long n = i / HW, px = i % HW;
    f16 h[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) h[c] = (f16)0.f;
    for (int c = 0; c < C; ++c) {
        const long o = (n * C + c) * HW + px;
        h[c] = in_f32 ? (f16)((const float*)in)[o] : ((const f16*)in)[o];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f16x8*)(slab + i * 32 + 8 * q) = *(const f16x8*)(h + 8 * q);
}

// Outermost down conv with C <= 4 input channels: the 4x4 stride-2 window of an output pixel is only 16*C <= 64 values, so the
// NCHW input is rewritten as a 64-channel "patch" slab at HALF resolution (channel j = (ky*4+kx)*C + c, zero outside the image and
// beyond 16*C) and the conv becomes a 1-tap GEMM over 64 channels instead of 16 taps over a 32-channel group holding C values.
// One thread per (output pixel, 32-channel group).
__global__ void unet_pre_patch(const void* in, int in_f32, int C, int H, int W, int N, f16* slab, long g) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ho = H >> 1, wo = W >> 1;
    const long M = (long)N * ho * wo;
    if (i >= 2 * M) return;
    const int grp = (int)(i / M);
    const long m = i - grp * M;
    const int ox = (int)(m % wo), oy = (int)((m / wo) % ho);
    const long n = m / ((long)wo * ho);
    f16 hbuf[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const int j = grp * 32 + e, t = j / C, c = j - t * C;
        const int iy = 2 * oy - 1 + (t >> 2), ix = 2 * ox - 1 + (t & 3);
        f16 v = (f16)0.f;
        if (t < 16 && iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const long o = ((n * C + c) * H + iy) * W + ix;
            v = in_f32 ? (f16)((const float*)in)[o] : ((const f16*)in)[o];
        }
        hbuf[e] = v;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f16x8*)(slab + grp * g + m * 32 + 8 * q) = *(const f16x8*)(hbuf + 8 * q);
}

// The same conv straight from the NCHW input, for 64 output channels: the 16 * C window values are the k dimension of v_mfma_f32_16x16x32_f16
// (k = c * 16 + ky * 4 + kx, two k-steps), a wave takes 16 consecutive output pixels of a row, gathers their windows into the B fragments
// (the input is 1/5 of the bytes this kernel writes and every value is re-read from L1 / L2), multiplies by the 64 x 64 weight panel it keeps in
// registers and writes BOTH views of the result (no norm layer follows this conv): LeakyReLU(0.2) for the next down conv, ReLU for the
// concatenation.  Replaces unet_pre_patch + two one-tap conv launches (178 -> 98 us at 64 x 256^2): one pass.
// wpk: [k-step][16-channel tile t][row rho][k-block lg][8], row rho of tile t = output channel 32 (t >> 1) + 8 (rho >> 2) + 4 (t & 1) + (rho & 3) (the plane
// row order of conv3x3.hip's 64-channel kernels): lane group lg ends up with channels 8 lg .. 8 lg + 7 of EACH 32-channel slab plane of its pixel, the four
// groups cover the pixel's whole 64 bytes of a plane, and a store instruction touches one plane (round 4; 16 lg .. 16 lg + 15 before: two planes per store; neutral
// for this kernel in a same-box A/B, profiles/r4/unet_first_rowp_ab.txt).
template <typename TI>
__global__ __launch_bounds__(256) void unet_first_mfma(const TI* in, int C, int H, int W, int N, const f16* wpk, const float* bias,
                                                       f16* d0, f16* d1, long g, int abl) {
    (void)abl;
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const int ho = H >> 1, wo = W >> 1, spr = wo >> 4;
    f16x8 a[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) a[ks][t] = *(const f16x8*)(wpk + ((ks * 4 + t) * 16 + li) * 32 + lg * 8);
    f32x4 b4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[t][j] = bias ? bias[32 * (t >> 1) + 8 * lg + 4 * (t & 1) + j] : 0.f;
    const int ky0 = (lg & 1) * 2;
    // element offsets of this lane's 16 window values from the first value of its row pair in channel 0 (loop constants; one image < 2^31 elements)
    int off[2][8];
    bool cok[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 2 + (lg >> 1);
        cok[ks] = c < C;
#pragma unroll
        for (int e = 0; e < 8; ++e) off[ks][e] = c * H * W + (e >> 2) * W + (e & 3) + 2 * li;
    }
    // a wave walks whole output rows (one division per row, none per segment)
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < N * ho; row += gridDim.x * 4) {
        const int n = row / ho, oy = row - n * ho;
        const int iy0 = 2 * oy - 1 + ky0;
        const bool row_ok[2] = {iy0 >= 0, iy0 + 1 < H};                // (the other bounds hold by construction: oy < H / 2, ky0 <= 2)
        const TI* rb = in + ((long)n * C * H + iy0) * W - 1;            // window origin of output pixel 0 of this row (wave-uniform)
        const int safe = iy0 >= 0 ? 1 : W + 1;                           // offset of a real element (column 0 of the window's first real row)
        const long m0 = (long)row * wo + li;
        // The window of segment sx + 1 is requested before segment sx is multiplied and stored, and first touched at the top of the next iteration
        // (two register sets used alternately: a single loop-carried set is copied at the loop edge, and the copy touches the data).
        // Measured on the diagnostic build (scripts/unet_first_abl.py, 64 x 256^2): 98 us; without the stores 37, with every load from one address 61,
        // with neither 26 -- what is left is the L1 request rate: the 2-byte gathers of a segment touch 64 cache lines, as many as its stores.
        unsigned rawA[2][8], rawB[2][8];   // one full register per value (a 16-bit load leaves its upper half zero): packing pairs here would touch the data
        auto request = [&](int sx, unsigned (&raw)[2][8]) __attribute__((always_inline)) {
            const int ox = sx * 16 + li;
            const bool col_ok[4] = {ox > 0, true, true, 2 * ox + 2 < W};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {      // unconditional loads from a clamped offset: independent, all in flight together
                    const bool ok = cok[ks] && row_ok[e >> 2] && col_ok[e & 3];
                    int o = ok ? off[ks][e] + sx * 32 : safe;
#ifdef INNFER_ABLATE
                    if (abl & 1) o = safe;          // diagnostic build: every load from one address
#endif
                    if constexpr (sizeof(TI) == 2) raw[ks][e] = ((const unsigned short*)rb)[o];
                    else raw[ks][e] = ((const unsigned*)rb)[o];
                }
        };
        auto segment = [&](int sx, unsigned (&cur)[2][8], unsigned (&nxt)[2][8]) __attribute__((always_inline)) {
            f16x8 bf[2];
            {
                const int ox = sx * 16 + li;
                const bool col_ok[4] = {ox > 0, true, true, 2 * ox + 2 < W};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f16 v;
                        if constexpr (sizeof(TI) == 2) v = __builtin_bit_cast(f16, (unsigned short)cur[ks][e]);
                        else v = (f16)__builtin_bit_cast(float, cur[ks][e]);
                        bf[ks][e] = (cok[ks] && row_ok[e >> 2] && col_ok[e & 3]) ? v : (f16)0.f;
                    }
            }
            if (sx + 1 < spr) request(sx + 1, nxt);
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = b4[t];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][t], bf[ks], acc[t], 0, 0, 0);
            f16x8 h0[2];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[t][j];
                    h0[t >> 1][(t & 1) * 4 + j] = (f16)fmaxf(v, 0.2f * v);
                }
            const long o = (m0 + sx * 16) * 32 + lg * 8;              // h[0]: plane 0 (tiles 0, 1), h[1]: plane 1 (tiles 2, 3)
#ifdef INNFER_ABLATE
            if ((abl & 2) && h0[0][0] != (f16)12345.f) return;       // diagnostic build: no stores
#endif
            *(f16x8*)(d0 + o) = h0[0]; *(f16x8*)(d0 + g + o) = h0[1];
            if (d1) {                                                 // (wave-uniform; nullptr: one stored form, the up conv applies the ReLU as it reads -- and the ReLU view is not even computed)
                f16x8 h1[2];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) h1[t >> 1][(t & 1) * 4 + j] = (f16)fmaxf(acc[t][j], 0.f);
                *(f16x8*)(d1 + o) = h1[0]; *(f16x8*)(d1 + g + o) = h1[1];
            }
        };
        request(0, rawA);
        int sx = 0;
        for (; sx + 1 < spr; sx += 2) {
            segment(sx, rawA, rawB);
            segment(sx + 1, rawB, rawA);
        }
        if (sx < spr) segment(sx, rawA, rawB);
    }
}

struct Param { std::string key; std::vector<int> shape; std::vector<float> host; bool set = false; };

struct Layer {            // one conv / conv-transpose
    bool transposed = false;
    bool patch = false;                                    // outermost down conv as a 1-tap GEMM on the patch slab (unet_pre_patch)
    bool phases = false;                                   // outermost up conv: the four output phases as 4*cout channels of ONE 3x3-tap GEMM
    int cin = 0, cout = 0, cin_pad = 0, cout_pad = 0;
    int w = -1, bias = -1, gamma = -1, beta = -1, rmean = -1, rvar = -1;   // indices into params
    f16* d_w[4] = {nullptr, nullptr, nullptr, nullptr};    // conv: [0]; convT: one panel set per phase, [1..3] point into [0]'s allocation
    long phase_elems = 0;
    void* d_w3 = nullptr; float* d_b3 = nullptr;           // `phases` layer: conv3x3.hip panels [4*cout][cin][3][3] and the bias repeated per phase
    float *d_bias = nullptr, *d_gamma = nullptr, *d_beta = nullptr;
    float *d_ev_alpha = nullptr, *d_ev_shift = nullptr;    // eval-mode BatchNorm: weight / sqrt(running_var + eps), bias - running_mean * that
    float* d_ones = nullptr;                               // instance-norm nets: the unit scale beside d_bias of a conv that no norm layer follows
    bool normed = false;                                   // a norm layer follows this conv
    f16* d_wf = nullptr;                                   // patch layer with 64 outputs: the panel of unet_first_mfma
    bool tile4 = false;                                    // ConvTranspose2d also packed as four 2x2-tap phase panels for conv3x3.hip's halo-tile kernel (d_w3)
    bool upconv = false;                                   // upsample_mode 'upconv': Upsample(nearest 2x) + Conv2d(3x3) instead of ConvTranspose2d(4, 2, 1)
};

}  // namespace

struct innfer_unet {
    int in_nc = 3, out_nc = 3, num_downs = 8, ngf = 64;
    std::vector<Param> params;
    std::vector<Layer> down, up;       // down[k], up[k] for level k = 0 .. num_downs-1
    std::vector<int> dc;               // down-path channels per level
    bool uploaded = false;
    bool eval_mode = false;            // BatchNorm on running statistics (nn.Module.eval()) instead of the current image's
    bool upconv = false;               // upsample_mode 'upconv' (UNet_arch.py:114-118,127-131,142-146; block.py:348-361)
    bool fp32 = false;                 // innfer_unet_set_precision(1): the fp32 forward on NCHW fp32 tensors (f32ops.hip), the reference's -no_fp16 mode
    std::vector<float*> f32_down, f32_up[4];      //   f32conv panels per level (up: one per output phase; upconv: [0] only)
    bool instance_norm = false;        // norm_type 'instance' (UNet_arch.py:38-41): nn.InstanceNorm2d -- no parameters, no running statistics, always the
                                       // statistics of the image; every conv then has a bias (use_bias, :101-104), which only matters where no norm follows
};

static int add_param(innfer_unet* u, const std::string& key, std::vector<int> shape) {
    Param p; p.key = key; p.shape = shape;
    u->params.push_back(p);
    return (int)u->params.size() - 1;
}

extern "C" int innfer_unet_create(innfer_unet** out, int in_nc, int out_nc, int num_downs, int ngf) {
    return innfer_unet_create_ex(out, in_nc, out_nc, num_downs, ngf, 0, 0);
}

extern "C" int innfer_unet_create_ex(innfer_unet** out, int in_nc, int out_nc, int num_downs, int ngf, int instance_norm, int upconv) {
    if (!out) return set_error(INNFER_ERR_INVALID, "unet_create: null out");
    if (num_downs < 5 || num_downs > 9 || ngf % 32 || ngf <= 0 || in_nc < 1 || in_nc > 32 || out_nc < 1 || out_nc > 32)
        return set_error(INNFER_ERR_UNSUPPORTED, "unet_create: in_nc=%d out_nc=%d num_downs=%d ngf=%d", in_nc, out_nc, num_downs, ngf);
    innfer_unet* u = new innfer_unet();
    u->in_nc = in_nc; u->out_nc = out_nc; u->num_downs = num_downs; u->ngf = ngf;
    u->instance_norm = instance_norm != 0;
    u->upconv = upconv != 0;
    const bool in_ = u->instance_norm;
    const int L = num_downs;
    u->dc.resize(L);
    for (int k = 0; k < L; ++k) u->dc[k] = ngf * (k < 3 ? (1 << k) : 8);      // 64,128,256,512,512,...
    u->down.resize(L); u->up.resize(L);
    std::string blk = "model.model.";                       // children of the block at level k
    for (int k = 0; k < L; ++k) {
        Layer& d = u->down[k]; Layer& p = u->up[k];
        const bool outer = k == 0, inner = k == L - 1;
        d.cin = outer ? in_nc : u->dc[k - 1]; d.cout = u->dc[k];
        p.transposed = true;
        p.cin = inner ? u->dc[k] : 2 * u->dc[k]; p.cout = outer ? out_nc : u->dc[k - 1];
        d.normed = !outer && !inner;
        if (outer) {
            d.w = add_param(u, blk + "0.weight", {d.cout, d.cin, 4, 4});
            if (in_) d.bias = add_param(u, blk + "0.bias", {d.cout});
        } else {
            d.w = add_param(u, blk + "1.weight", {d.cout, d.cin, 4, 4});
            if (in_) d.bias = add_param(u, blk + "1.bias", {d.cout});
            if (!inner && !in_) {
                d.gamma = add_param(u, blk + "2.weight", {d.cout});
                d.beta = add_param(u, blk + "2.bias", {d.cout});
                d.rmean = add_param(u, blk + "2.running_mean", {d.cout});
                d.rvar = add_param(u, blk + "2.running_var", {d.cout});
                add_param(u, blk + "2.num_batches_tracked", {});
            }
        }
        const std::string next = blk + (outer ? "1.model." : "3.model.");
        // (the up-path parameters of level k follow its submodule in module order)
        if (k + 1 < L) {
            // recurse textually: deeper levels are appended first, so defer the up params
        }
        d.cin_pad = (d.cin + 31) / 32 * 32; d.cout_pad = (d.cout + 63) / 64 * 64;
        p.cin_pad = (p.cin + 31) / 32 * 32; p.cout_pad = (p.cout + 63) / 64 * 64;
        if (outer && d.cin <= 4) { d.patch = true; d.cin_pad = 64; }
        if (outer && 4 * p.cout <= 16 && !u->upconv) p.phases = true;          // conv3x3.hip's planar epilogue: one 16-channel tile
        if (u->upconv) { p.upconv = true; p.transposed = false; }
        // levels whose input grid fills the halo-tile kernel's 16 x 32 tiles at the usual 256 x 256 (and larger) inputs; deeper ones stay on the
        // gather GEMM (a 16 x 16 grid would pad every tile to twice its pixels)
        if (p.transposed && !p.phases && k >= 1 && k <= 3 && p.cout % 64 == 0) p.tile4 = true;
        if (!d.patch && k >= 1 && k <= 3 && k < L - 1 && d.cout % 64 == 0 && d.cin % 32 == 0) d.tile4 = true;       // the 4x4 stride-2 convs of the same levels
        blk = next;
    }
    // up params, innermost first (= module order after the submodule)
    for (int k = L - 1; k >= 0; --k) {
        std::string b = "model.model.";
        for (int j = 0; j < k; ++j) b += (j == 0 ? "1.model." : "3.model.");
        Layer& p = u->up[k];
        const bool outer = k == 0, inner = k == L - 1;
        // upconv_block is a Sequential(Upsample, Conv2d) in the block's Sequential: its conv is `<i>.1`
        const std::string wi = std::string(outer ? "3" : (inner ? "3" : "5")) + (u->upconv ? ".1" : ""), ni = inner ? "4" : "6";
        p.w = u->upconv ? add_param(u, b + wi + ".weight", {p.cout, p.cin, 3, 3}) : add_param(u, b + wi + ".weight", {p.cin, p.cout, 4, 4});
        p.normed = !outer;
        if (outer) {
            p.bias = add_param(u, b + wi + ".bias", {p.cout});
        } else if (in_) {
            p.bias = add_param(u, b + wi + ".bias", {p.cout});          // in front of an InstanceNorm2d: cancels in the mean subtraction, kept for the state dict
        } else {
            p.gamma = add_param(u, b + ni + ".weight", {p.cout});
            p.beta = add_param(u, b + ni + ".bias", {p.cout});
            p.rmean = add_param(u, b + ni + ".running_mean", {p.cout});
            p.rvar = add_param(u, b + ni + ".running_var", {p.cout});
            add_param(u, b + ni + ".num_batches_tracked", {});
        }
    }
    *out = u;
    return INNFER_OK;
}

extern "C" void innfer_unet_destroy(innfer_unet* u) {
    if (!u) return;
    for (auto* v : {&u->down, &u->up})
        for (auto& l : *v) {
            if (l.d_w[0]) (void)hipFree(l.d_w[0]);
            if (l.d_w3) (void)hipFree(l.d_w3);
            if (l.d_wf) (void)hipFree(l.d_wf);
            if (l.d_b3) (void)hipFree(l.d_b3);
            if (l.d_bias) (void)hipFree(l.d_bias);
            if (l.d_gamma) (void)hipFree(l.d_gamma);
            if (l.d_beta) (void)hipFree(l.d_beta);
            if (l.d_ev_alpha) (void)hipFree(l.d_ev_alpha);
            if (l.d_ev_shift) (void)hipFree(l.d_ev_shift);
            if (l.d_ones) (void)hipFree(l.d_ones);
        }
    for (auto v : u->f32_down) if (v) (void)hipFree(v);
    for (auto& vv : u->f32_up) for (auto v : vv) if (v) (void)hipFree(v);
    delete u;
}

extern "C" int innfer_unet_num_params(innfer_unet* u) { return u ? (int)u->params.size() : INNFER_ERR_INVALID; }

extern "C" int innfer_unet_param_info(innfer_unet* u, int idx, char* key, size_t key_cap, int* ndim, int* shape4) {
    if (!u || idx < 0 || idx >= (int)u->params.size()) return set_error(INNFER_ERR_INVALID, "unet_param_info: bad index");
    const Param& p = u->params[idx];
    if (key && key_cap) { strncpy(key, p.key.c_str(), key_cap - 1); key[key_cap - 1] = 0; }
    if (ndim) *ndim = (int)p.shape.size();
    if (shape4) for (size_t i = 0; i < 4; ++i) shape4[i] = i < p.shape.size() ? p.shape[i] : 1;
    return INNFER_OK;
}

extern "C" int innfer_unet_set_eval(innfer_unet* u, int eval_mode) {
    if (!u) return set_error(INNFER_ERR_INVALID, "unet_set_eval: null handle");
    u->eval_mode = eval_mode != 0;
    return INNFER_OK;
}

extern "C" int innfer_unet_set_param(innfer_unet* u, int idx, const float* h_data) {
    if (!u || idx < 0 || idx >= (int)u->params.size() || !h_data) return set_error(INNFER_ERR_INVALID, "unet_set_param: bad arguments");
    Param& p = u->params[idx];
    size_t n = 1;
    for (int s : p.shape) n *= (size_t)s;
    p.host.assign(h_data, h_data + n);
    p.set = true;
    u->uploaded = false;
    return INNFER_OK;
}

static int upload_f32(float** dst, const std::vector<float>& v) {
    if (!*dst) INNFER_HIP(hipMalloc((void**)dst, v.size() * sizeof(float)));
    INNFER_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return INNFER_OK;
}

static int upload_f16(f16** dst, const std::vector<f16>& v) {
    if (!*dst) INNFER_HIP(hipMalloc((void**)dst, v.size() * sizeof(f16)));
    INNFER_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(f16), hipMemcpyHostToDevice));
    return INNFER_OK;
}

// taps of output phase (a, b) of ConvTranspose2d(k=4, s=2, p=1): oy = 2*iy - 1 + ky
static void phase_taps(int a, int b, int ky[4], int kx[4], int dy[4], int dx[4]) {
    const int kys[2] = {a == 0 ? 1 : 0, a == 0 ? 3 : 2}, dys[2] = {a == 0 ? 0 : 1, a == 0 ? -1 : 0};
    const int kxs[2] = {b == 0 ? 1 : 0, b == 0 ? 3 : 2}, dxs[2] = {b == 0 ? 0 : 1, b == 0 ? -1 : 0};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) { ky[i * 2 + j] = kys[i]; dy[i * 2 + j] = dys[i]; kx[i * 2 + j] = kxs[j]; dx[i * 2 + j] = dxs[j]; }
}

static int upload_all(innfer_unet* u) {
    for (auto v : u->f32_down) if (v) (void)hipFree(v);               // the fp32 panels follow the parameters: rebuilt by innfer_unet_set_precision
    u->f32_down.clear();
    for (auto& vv : u->f32_up) { for (auto v : vv) if (v) (void)hipFree(v); vv.clear(); }
    for (auto& p : u->params)
        if (!p.set && p.key.find("running_") == std::string::npos && p.key.find("num_batches") == std::string::npos)
            return set_error(INNFER_ERR_INVALID, "unet: parameter '%s' was never set", p.key.c_str());
    std::vector<f16> panel;
    for (auto* v : {&u->down, &u->up})
        for (auto& l : *v) {
            const std::vector<float>& w = u->params[l.w].host;
            if (l.patch) {
                gg::pack_panels(panel, l.cout, 16 * l.cin, 64, 1,
                            [&](int co, int j, int) { const int t = j / l.cin, ci = j - t * l.cin;
                                                      return w[(((size_t)co * l.cin + ci) * 4 + (t >> 2)) * 4 + (t & 3)]; });
                int rc = upload_f16(&l.d_w[0], panel); if (rc) return rc;
                if (l.cout % 64 == 0) {        // ... and as a one-tap panel for the halo-tile kernel, whose epilogue writes the fp16 slabs directly
                    std::vector<float> w1((size_t)l.cout * 64, 0.f), b3((size_t)l.cout, 0.f);
                    if (l.bias >= 0) b3 = u->params[l.bias].host;          // instance-norm nets: the outermost down conv has a bias and no norm
                    for (int co = 0; co < l.cout; ++co)
                        for (int j = 0; j < 16 * l.cin; ++j) {
                            const int t = j / l.cin, ci = j - t * l.cin;
                            w1[(size_t)co * 64 + j] = w[(((size_t)co * l.cin + ci) * 4 + (t >> 2)) * 4 + (t & 3)];
                        }
                    std::vector<char> packed(conv_packed_bytes_taps(l.cout, 64, 0x10));
                    conv_pack_1x1(w1.data(), l.cout, 64, packed.data());
                    if (!l.d_w3) INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
                    INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
                    rc = upload_f32(&l.d_b3, b3); if (rc) return rc;
                }
                if (l.cout == 64) {            // unet_first_mfma's panel: k = c * 16 + ky * 4 + kx
                    std::vector<f16> wf(2 * 4 * 16 * 32);
                    for (int ks = 0; ks < 2; ++ks)
                        for (int t = 0; t < 4; ++t)
                            for (int rho = 0; rho < 16; ++rho)
                                for (int q = 0; q < 32; ++q) {
                                    const int co = 32 * (t >> 1) + 8 * (rho >> 2) + 4 * (t & 1) + (rho & 3), kk = ks * 32 + q, c = kk >> 4, tap = kk & 15;      // (plane row order: see unet_first_mfma)
                                    wf[((ks * 4 + t) * 16 + rho) * 32 + q] = (f16)(c < l.cin ? w[(((size_t)co * l.cin + c) * 4 + (tap >> 2)) * 4 + (tap & 3)] : 0.f);
                                }
                    rc = upload_f16(&l.d_wf, wf); if (rc) return rc;
                }
            } else if (l.phases) {
                // The four output phases of ConvTranspose2d(k4, s2, p1) as ONE 3x3 convolution with 4*cout output channels over the
                // un-upsampled input (conv3x3.hip's halo-tile kernel reads each input pixel once instead of once per tap): output channel
                // (2a+b)*cout + c at tap (dy,dx) in {-1,0,1}^2 is w[ci][c][ky][kx] with ky = 1 (dy 0), 3 (dy -1) for a == 0 and
                // ky = 0 (dy +1), 2 (dy 0) for a == 1 (oy = 2*iy - 1 + ky); the other (phase, tap) pairs are structural zeros.
                auto kof = [](int a, int d) { return a == 0 ? (d == 0 ? 1 : (d == -1 ? 3 : -1)) : (d == 1 ? 0 : (d == 0 ? 2 : -1)); };
                const int K3 = 4 * l.cout;
                std::vector<float> w3((size_t)K3 * l.cin_pad * 9, 0.f), b3(16, 0.f);
                for (int co = 0; co < K3; ++co) {
                    const int ph = co / l.cout, c = co - ph * l.cout;
                    for (int ci = 0; ci < l.cin; ++ci)
                        for (int t = 0; t < 9; ++t) {
                            const int ky = kof(ph >> 1, t / 3 - 1), kx = kof(ph & 1, t % 3 - 1);
                            if (ky >= 0 && kx >= 0) w3[((size_t)co * l.cin_pad + ci) * 9 + t] = w[(((size_t)ci * l.cout + c) * 4 + ky) * 4 + kx];
                        }
                    b3[co] = u->params[l.bias].host[c];
                }
                std::vector<char> packed(conv_packed_bytes(K3, l.cin_pad));
                conv_pack(w3.data(), K3, l.cin_pad, packed.data());
                if (!l.d_w3) INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
                INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
                int rc = upload_f32(&l.d_b3, b3); if (rc) return rc;
            } else if (l.upconv) {
                // Upsample(nearest 2x) + Conv2d(3x3, zero padding): conv3x3.hip's halo-tile kernel with the upsampled read (`up`);
                // the bias (outermost / instance-norm nets) rides in the conv epilogue
                std::vector<char> packed(conv_packed_bytes(l.cout, l.cin));
                conv_pack(w.data(), l.cout, l.cin, packed.data());
                if (!l.d_w3) INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
                INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
                std::vector<float> b3((size_t)std::max(16, l.cout), 0.f);
                if (l.bias >= 0) std::copy(u->params[l.bias].host.begin(), u->params[l.bias].host.end(), b3.begin());
                int rc = upload_f32(&l.d_b3, b3); if (rc) return rc;
            } else if (!l.transposed) {
                gg::pack_panels(panel, l.cout, l.cin, l.cin_pad, 16,
                            [&](int co, int ci, int t) { return w[(((size_t)co * l.cin + ci) * 4 + (t >> 2)) * 4 + (t & 3)]; });
                int rc = upload_f16(&l.d_w[0], panel); if (rc) return rc;
                if (l.tile4) {
                    // Conv2d(4, 2, 1) = the 2x2-tap conv of the space-to-depth source (conv3x3_pc's stride-2 loader): virtual channel (2 pa + pb) * cin + ci,
                    // tap (1 + dy, 1 + dx) of the 3x3 lattice carries w[co][ci][2 dy + pa][2 dx + pb]
                    std::vector<float> b3((size_t)l.cout, 0.f);
                    if (l.bias >= 0) b3 = u->params[l.bias].host;
                    std::vector<char> packed(conv_packed_bytes_s2k4(l.cout, l.cin));
                    conv_pack_s2k4(w.data(), l.cout, l.cin, packed.data());
                    if (!l.d_w3) INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
                    INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
                    rc = upload_f32(&l.d_b3, b3); if (rc) return rc;
                }
            } else {
                if (l.tile4) {
                    // phase (a, b) of ConvTranspose2d(4, 2, 1) = taps (dy, dx) in {-1, 0}^2 at the virtual pixel (y + a, x + b) (conv3x3_pc<.., TM = 0x1B>):
                    // tap (r, s) of the 3x3 lattice (r, s in {0, 1}) carries w[ci][c][3 - 2r - a][3 - 2s - b]  (oy = 2 iy - 1 + ky)
                    std::vector<float> b3((size_t)4 * l.cout, 0.f);
                    if (l.bias >= 0) for (int co = 0; co < 4 * l.cout; ++co) b3[co] = u->params[l.bias].host[co % l.cout];
                    std::vector<char> packed(conv_packed_bytes_deconv2x(l.cout, l.cin));
                    conv_pack_deconv2x(w.data(), l.cout, l.cin, 4, packed.data());
                    if (!l.d_w3) INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
                    INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
                    int rc = upload_f32(&l.d_b3, b3); if (rc) return rc;
                }
                std::vector<f16> all;                               // the four phase panels back to back: one grouped launch reads them
                for (int ph = 0; ph < 4; ++ph) {
                    int ky[4], kx[4], dy[4], dx[4];
                    phase_taps(ph >> 1, ph & 1, ky, kx, dy, dx);
                    gg::pack_panels(panel, l.cout, l.cin, l.cin_pad, 4,
                                [&](int co, int ci, int t) { return w[(((size_t)ci * l.cout + co) * 4 + ky[t]) * 4 + kx[t]]; });
                    all.insert(all.end(), panel.begin(), panel.end());
                }
                int rc = upload_f16(&l.d_w[0], all); if (rc) return rc;
                l.phase_elems = (long)panel.size();
                for (int ph = 1; ph < 4; ++ph) l.d_w[ph] = l.d_w[0] + ph * l.phase_elems;      // aliases, never freed
            }
            if (l.bias >= 0) { int rc = upload_f32(&l.d_bias, u->params[l.bias].host); if (rc) return rc; }
            if (l.gamma >= 0) { int rc = upload_f32(&l.d_gamma, u->params[l.gamma].host); if (rc) return rc; }
            if (l.beta >= 0) { int rc = upload_f32(&l.d_beta, u->params[l.beta].host); if (rc) return rc; }
            if (u->instance_norm) {            // InstanceNorm2d(affine=False): unit scale, zero shift on the statistics of the image
                const std::vector<float> ones((size_t)l.cout, 1.f), zeros((size_t)l.cout, 0.f);
                int rc = upload_f32(&l.d_ones, ones); if (rc) return rc;
                if (l.normed) { rc = upload_f32(&l.d_gamma, ones); if (rc) return rc; rc = upload_f32(&l.d_beta, zeros); if (rc) return rc; }
            }
            if (l.gamma >= 0) {
                // eval-mode transform as ATen forms it (batch_norm_cpu_transform_input: alpha = weight / sqrt(running_var + eps),
                // beta = bias - running_mean * alpha); a checkpoint without running statistics means a fresh BatchNorm's (0, 1)
                const int C = l.cout;
                const std::vector<float>&g = u->params[l.gamma].host, &b = u->params[l.beta].host;
                const Param &pm = u->params[l.rmean], &pv = u->params[l.rvar];
                std::vector<float> al(C), sh(C);
                for (int c = 0; c < C; ++c) {
                    const float mean = pm.set ? pm.host[c] : 0.f, var = pv.set ? pv.host[c] : 1.f;
                    al[c] = g[c] * (1.0f / std::sqrt(var + 1e-5f));
                    sh[c] = b[c] - mean * al[c];
                }
                int rc = upload_f32(&l.d_ev_alpha, al); if (rc) return rc;
                rc = upload_f32(&l.d_ev_shift, sh); if (rc) return rc;
            }
        }
    u->uploaded = true;
    return INNFER_OK;
}

namespace {
struct UCarve { size_t x0, raw, mean, rstd, bnpart, r_inner, splitk, total; std::vector<size_t> D, CAT; };
// partial results of the split-K deep layers: 8 segments x (<= GG_SPLIT_MAX_PX pixels x <= 1024 channels) fp32 per image,
// so that whether a layer is split never depends on the batch size
inline size_t splitk_bytes(int N) { return (size_t)N * 8 * gg::SPLIT_MAX_PX * 1024 * sizeof(float); }

UCarve ucarve(const innfer_unet* u, int N, int H, int W) {
    UCarve c;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int L = u->num_downs;
    size_t off = 0;
    c.x0 = off; off += al((size_t)N * H * W * 32 * 2);
    size_t raw = (size_t)N * H * W * 64 * 4;                                       // outermost up conv (64-padded)
    for (int k = 0; k < L; ++k) raw = std::max(raw, (size_t)N * (H >> (k + 1)) * (W >> (k + 1)) * (size_t)((u->dc[k] + 63) / 64 * 64) * 4);
    c.raw = off; off += al(raw);
    c.mean = off; off += al((size_t)N * 1024 * 4);
    c.rstd = off; off += al((size_t)N * 1024 * 4);
    size_t part = 0;                                                               // (mean, M2) per BN segment of the widest layer
    for (int k = 0; k < L; ++k) {
        const size_t hw = (size_t)(H >> (k + 1)) * (W >> (k + 1)), hw2 = (size_t)(H >> k) * (W >> k);
        part = std::max(part, norm::part_floats(u->dc[k], (long)hw));
        if (k > 0) part = std::max(part, norm::part_floats(u->dc[k - 1], (long)hw2));
        // ... or the partial statistics out of the halo-tile convs' epilogues (3 floats per 16 x 32 tile, consumer wave and channel)
        const int hk = H >> (k + 1), wk = W >> (k + 1);
        part = std::max(part, norm::parts_floats(u->dc[k], conv_stats_nper(hk, wk, 1)));
        if (k > 0) part = std::max(part, norm::parts_floats(u->dc[k - 1], std::max(conv_stats_nper(hk, wk, 4), conv_stats_nper(2 * hk, 2 * wk, 1))));
    }
    c.bnpart = off; off += al((size_t)N * part * 4);
    c.D.resize(L); c.CAT.resize(L);
    for (int k = 0; k < L - 1; ++k) {
        const size_t px = (size_t)N * (H >> (k + 1)) * (W >> (k + 1));
        c.D[k] = off; off += al(px * u->dc[k] * 2);
        c.CAT[k] = off; off += al(px * 2 * u->dc[k] * 2);
    }
    c.r_inner = off; off += al((size_t)N * (H >> L) * (W >> L) * u->dc[L - 1] * 2);
    c.splitk = off; off += splitk_bytes(N);
    c.total = off;
    return c;
}

int run_gemm(const Layer& l, const f16* wpk, const f16* in, long in_g, int N, int Hin, int Win, float* raw,
             int Ho, int Wo, int stride, int ntaps, const int* dy, const int* dx, int Hfull, int Wfull,
             int os, int ooy, int oox, hipStream_t s, float* scratch, int raw_stride = 0, int* ks_out = nullptr) {
    return gg::launch(wpk, l.cin_pad, l.cout_pad, in, in_g, N, Hin, Win, raw, Ho, Wo, stride, ntaps, dy, dx,
                      Hfull, Wfull, os, ooy, oox, 0, s, scratch, splitk_bytes(N), raw_stride, 0, 1, 0, 0, 0, 0, 0, ks_out);
}
}  // namespace

// >= 70 % of the halo-tile kernel's pixels are real ones: 16-row tiles of 32 columns; grids at most 16 wide share a tile row between two images (conv3x3_pc's image pairs)
static bool fills_tiles(int h, int w) {
    const long tw = w <= 16 ? 16 : (w + 31) / 32 * 32;
    return (long)h * w * 10 >= (long)((h + 15) / 16 * 16) * tw * 7;
}

namespace {
// fp32 mode: CAT[k] (k = 1 .. L-1) = [t_k | u_k], 2 dc[k-1] channels at H >> k; RAW: the largest conv output in front of a norm; INNER: the innermost conv's output
struct UCarve32 { std::vector<size_t> CAT; size_t raw, inner, total; };
UCarve32 ucarve32(const innfer_unet* u, int N, int H, int W) {
    UCarve32 c;
    const int L = u->num_downs;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0, rawmax = 0;
    c.CAT.assign(L, 0);
    for (int k = 1; k < L; ++k) {
        c.CAT[k] = off; off += al((size_t)N * 2 * u->dc[k - 1] * (H >> k) * (W >> k) * 4);
        rawmax = std::max(rawmax, (size_t)N * u->dc[k - 1] * (H >> k) * (W >> k) * 4);        // up[k]'s output (and down[k-1]'s) before its norm
    }
    c.raw = off; off += al(rawmax);
    c.inner = off; off += al((size_t)N * u->dc[L - 1] * (H >> L) * (W >> L) * 4);
    c.total = off;
    return c;
}
}  // namespace

extern "C" size_t innfer_unet_workspace_bytes(innfer_unet* u, int N, int H, int W) {
    if (!u || N <= 0 || H <= 0 || W <= 0) return 0;
    return u->fp32 ? ucarve32(u, N, H, W).total : ucarve(u, N, H, W).total;
}

// The reference's fp16 switch for this generator (run.py:345,421-422): fp32 = 1 runs every conv, norm and activation of UnetGenerator.forward in fp32 on NCHW fp32
// tensors (csrc/f32ops.hip: the fp32 matrix instruction, fp32 statistics) -- <= 1e-4 of the fp32 reference; input / output fp32.  A load-time call: it
// packs the fp32 panels of the parameters set so far (hipMalloc + synchronous copies); the workspace differs (ask innfer_unet_workspace_bytes after it).
extern "C" int innfer_unet_set_precision(innfer_unet* u, int fp32) {
    if (!u || (fp32 != 0 && fp32 != 1)) return set_error(INNFER_ERR_INVALID, "unet_set_precision: 0 (fp16 arithmetic) or 1 (fp32)");
    u->fp32 = fp32 != 0;
    if (!u->fp32) return INNFER_OK;
    if (!u->uploaded) { int rc = upload_all(u); if (rc) return rc; }
    if (!u->f32_down.empty()) return INNFER_OK;                    // (upload_all clears them when a parameter changes)
    const int L = u->num_downs;
    u->f32_down.assign(L, nullptr);
    for (auto& v : u->f32_up) v.assign(L, nullptr);
    std::vector<float> host;
    auto put = [&](float** dst, int K, int C, int ntap, const std::function<float(int, int, int)>& w) -> int {
        host.resize(f32conv_packed_floats(K, C, ntap));
        f32conv_pack(K, C, ntap, w, host.data());
        INNFER_HIP(hipMalloc((void**)dst, host.size() * sizeof(float)));
        INNFER_HIP(hipMemcpy(*dst, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
        return INNFER_OK;
    };
    for (int k = 0; k < L; ++k) {
        const Layer &d = u->down[k], &p = u->up[k];
        const std::vector<float>& wd = u->params[d.w].host;           // Conv2d(4, 2, 1): [cout][cin][4][4]
        int rc = put(&u->f32_down[k], d.cout, d.cin, 16, [&](int co, int ci, int t) { return wd[((size_t)co * d.cin + ci) * 16 + t]; });
        if (rc) return rc;
        const std::vector<float>& wu = u->params[p.w].host;
        if (p.upconv) {                                               // Upsample(nearest 2x) + Conv2d(3x3): [cout][cin][3][3]
            rc = put(&u->f32_up[0][k], p.cout, p.cin, 9, [&](int co, int ci, int t) { return wu[((size_t)co * p.cin + ci) * 9 + t]; });
            if (rc) return rc;
        } else if (4 * p.cout <= 16) {
            // ConvTranspose2d(4, 2, 1) with so few outputs that all four phases fit the 16-channel tile one phase would occupy (the outermost layer: 3): ONE conv of
            // 9 taps on the input grid and 4 cout channels, channel ph * cout + co = phase ph of output co; a phase's panel rows are zero at the taps that are not
            // its own (F32Conv.phase_k).  The 128-channel input is read once instead of four times -- the four launches were HBM-bound on it (profiles/r5/fp32_modes.txt)
            rc = put(&u->f32_up[0][k], 4 * p.cout, p.cin, 9, [&](int kq, int ci, int t) {
                const int ph = kq / p.cout, co = kq - ph * p.cout, ty = t / 3 - 1, tx = t % 3 - 1;
                int ky[4], kx[4], dy[4], dx[4];
                phase_taps(ph >> 1, ph & 1, ky, kx, dy, dx);
                for (int q = 0; q < 4; ++q)
                    if (dy[q] == ty && dx[q] == tx) return wu[(((size_t)ci * p.cout + co) * 4 + ky[q]) * 4 + kx[q]];
                return 0.f;
            });
            if (rc) return rc;
        } else {                                                      // ConvTranspose2d(4, 2, 1): [cin][cout][4][4], one panel per output phase
            for (int ph = 0; ph < 4; ++ph) {
                int ky[4], kx[4], dy[4], dx[4];
                phase_taps(ph >> 1, ph & 1, ky, kx, dy, dx);
                rc = put(&u->f32_up[ph][k], p.cout, p.cin, 4, [&](int co, int ci, int t) { return wu[(((size_t)ci * p.cout + co) * 4 + ky[t]) * 4 + kx[t]]; });
                if (rc) return rc;
            }
        }
    }
    return INNFER_OK;
}

namespace {
// UnetGenerator.forward in fp32 (UNet_arch.py:70-161; the in-place LeakyReLU at the head of every block rewrites the tensor the skip connection carries, the parent's
// in-place ReLU acts on the concatenation: oracle/nets.py unet_forward states the same graph)
int unet_forward_f32(innfer_unet* u, const float* x, float* y, int N, int H, int W, char* ws, hipStream_t s) {
    const int L = u->num_downs;
    const UCarve32 cv = ucarve32(u, N, H, W);
    const bool ev = u->eval_mode && !u->instance_norm;
    auto CAT = [&](int k) { return (float*)(ws + cv.CAT[k]); };
    float* RAW = (float*)(ws + cv.raw);
    float* INNER = (float*)(ws + cv.inner);
    auto norm = [&](const Layer& l, const float* in, int C, int h, int w, float* out, long out_ns, int act) -> int {
        const long hw = (long)h * w;
        if (u->instance_norm) return f32_norm_launch(in, C * hw, hw, out, out_ns, hw, N, C, hw, 2, 1e-5f, nullptr, nullptr, nullptr, nullptr, act, s);
        if (ev) return f32_norm_launch(in, C * hw, hw, out, out_ns, hw, N, C, hw, 3, 1e-5f, l.d_ev_alpha, l.d_ev_shift, nullptr, nullptr, act, s);
        return f32_norm_launch(in, C * hw, hw, out, out_ns, hw, N, C, hw, 0, 1e-5f, l.d_gamma, l.d_beta, nullptr, nullptr, act, s);
    };
    // Conv2d(4, 2, 1) of level k: input view (C channels of a tensor with `ctot` channels) at h x w -> h/2 x w/2
    auto down = [&](int k, const float* in, int ctot, int h, int w, int in_act, float* out, long out_ns, int act) -> int {
        const Layer& d = u->down[k];
        F32Conv c{};
        c.in = in; c.in_nstride = (long)ctot * h * w; c.in_cstride = (long)h * w; c.C = d.cin; c.Hin = h; c.Win = w;
        c.wp = u->f32_down[k]; c.bias = d.d_bias; c.K = d.cout;
        c.out = out; c.out_nstride = out_ns; c.out_cstride = (long)(h / 2) * (w / 2); c.out_pstride = 1; c.Wout = w / 2;
        c.Ho = h / 2; c.Wo = w / 2; c.osy = c.osx = 1; c.isy = c.isx = 2;
        c.ntap = 16;
        for (int t = 0; t < 16; ++t) { c.dy[t] = t / 4 - 1; c.dx[t] = t % 4 - 1; }
        c.in_act = in_act; c.act = act; c.N = N;
        return f32conv_launch(c, s);
    };
    // the up conv of level k: relu(input of `cin` channels at h x w) -> cout channels at 2h x 2w
    auto up = [&](int k, const float* in, int h, int w, float* out, long out_ns, int act) -> int {
        const Layer& p = u->up[k];
        F32Conv c{};
        c.in = in; c.in_nstride = (long)p.cin * h * w; c.in_cstride = (long)h * w; c.C = p.cin; c.Hin = h; c.Win = w;
        c.bias = p.d_bias; c.K = p.cout;
        c.out = out; c.out_nstride = out_ns; c.out_cstride = (long)4 * h * w; c.out_pstride = 1; c.Wout = 2 * w;
        c.in_act = 2; c.act = act; c.N = N;
        if (p.upconv) {
            c.wp = u->f32_up[0][k]; c.up = 1; c.Ho = 2 * h; c.Wo = 2 * w; c.osy = c.osx = 1; c.isy = c.isx = 1; c.ntap = 9;
            for (int t = 0; t < 9; ++t) { c.dy[t] = t / 3 - 1; c.dx[t] = t % 3 - 1; }
            return f32conv_launch(c, s);
        }
        if (4 * p.cout <= 16) {       // all four phases in one launch (see innfer_unet_set_precision)
            c.wp = u->f32_up[0][k]; c.K = 4 * p.cout; c.phase_k = p.cout; c.Ho = h; c.Wo = w; c.osy = c.osx = 1; c.isy = c.isx = 1; c.ntap = 9;
            for (int t = 0; t < 9; ++t) { c.dy[t] = t / 3 - 1; c.dx[t] = t % 3 - 1; }
            return f32conv_launch(c, s);
        }
        for (int ph = 0; ph < 4; ++ph) {
            int ky[4], kx[4];
            phase_taps(ph >> 1, ph & 1, ky, kx, c.dy, c.dx);
            c.wp = u->f32_up[ph][k]; c.Ho = h; c.Wo = w; c.osy = c.osx = 2; c.ooy = ph >> 1; c.oox = ph & 1; c.isy = c.isx = 1; c.ntap = 4;
            int rc = f32conv_launch(c, s);
            if (rc) return rc;
        }
        return INNFER_OK;
    };
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    // level 0: no norm behind the outermost conv; the next block's in-place LeakyReLU is this conv's epilogue (t_1 = the first half of CAT[1])
    CK(down(0, x, u->in_nc, H, W, 0, CAT(1), (long)2 * u->dc[0] * (H / 2) * (W / 2), 1));
    for (int k = 1; k <= L - 2; ++k) {
        const int h = H >> k, w = W >> k;
        CK(down(k, CAT(k), 2 * u->dc[k - 1], h, w, 0, RAW, (long)u->dc[k] * (h / 2) * (w / 2), 0));
        CK(norm(u->down[k], RAW, u->dc[k], h / 2, w / 2, CAT(k + 1), (long)2 * u->dc[k] * (h / 2) * (w / 2), 1));     // norm, then the next block's LeakyReLU
    }
    {   // innermost block: conv (no norm) -> relu -> up conv -> norm -> second half of CAT[L-1]
        const int k = L - 1, h = H >> k, w = W >> k;
        CK(down(k, CAT(k), 2 * u->dc[k - 1], h, w, 0, INNER, (long)u->dc[k] * (h / 2) * (w / 2), 0));
        CK(up(k, INNER, h / 2, w / 2, RAW, (long)u->dc[k - 1] * h * w, 0));
        CK(norm(u->up[k], RAW, u->dc[k - 1], h, w, CAT(k) + (long)u->dc[k - 1] * h * w, (long)2 * u->dc[k - 1] * h * w, 0));
    }
    for (int k = L - 2; k >= 1; --k) {
        const int h = H >> k, w = W >> k;                          // up[k]: relu(CAT[k+1]) at h/2 -> dc[k-1] channels at h
        CK(up(k, CAT(k + 1), h / 2, w / 2, RAW, (long)u->dc[k - 1] * h * w, 0));
        CK(norm(u->up[k], RAW, u->dc[k - 1], h, w, CAT(k) + (long)u->dc[k - 1] * h * w, (long)2 * u->dc[k - 1] * h * w, 0));
    }
    CK(up(0, CAT(1), H / 2, W / 2, y, (long)u->out_nc * H * W, 3));               // outermost: bias + tanh
#undef CK
    return INNFER_OK;
}
}  // namespace

extern "C" double innfer_unet_flops(innfer_unet* u, int N, int H, int W) {
    if (!u) return 0.0;
    double f = 0.0;
    for (int k = 0; k < u->num_downs; ++k) {
        const double px = (double)N * (H >> (k + 1)) * (W >> (k + 1));           // down output = up input grid
        f += 2.0 * 16.0 * u->down[k].cin * u->down[k].cout * px;
        f += 2.0 * (u->upconv ? 36.0 : 16.0) * u->up[k].cin * u->up[k].cout * px;          // upconv: 9 taps on the 2x grid
    }
    return f;
}

extern "C" int innfer_unet_forward(innfer_unet* u, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                                   int N, int H, int W, void* d_ws, size_t ws_bytes, void* stream) {
    if (!u || !d_in || !d_out || !d_ws) return set_error(INNFER_ERR_INVALID, "unet_forward: null argument");
    const int L = u->num_downs;
    if (N <= 0 || H <= 0 || W <= 0 || (H & ((1 << L) - 1)) || (W & ((1 << L) - 1)))
        return set_error(INNFER_ERR_INVALID, "unet_forward: %dx%d must be a multiple of %d", H, W, 1 << L);
    const bool ev = u->eval_mode && !u->instance_norm;            // InstanceNorm2d keeps no running statistics: eval() changes nothing
    if (!ev && (H >> (L - 1)) * (W >> (L - 1)) < 2)
        return set_error(INNFER_ERR_INVALID, "unet_forward: the norm layers need more than one value per channel");
    if (!u->uploaded) { int rc = upload_all(u); if (rc) return rc; if (u->fp32) { rc = innfer_unet_set_precision(u, 1); if (rc) return rc; } }
    if (u->fp32) {
        if (in_dtype != INNFER_F32 || out_dtype != INNFER_F32) return set_error(INNFER_ERR_INVALID, "unet_forward: the fp32 mode takes and returns fp32 tensors");
        if (u->f32_down.empty()) return set_error(INNFER_ERR_INVALID, "unet_forward: call innfer_unet_set_precision(u, 1) after the last innfer_unet_set_param");
        const UCarve32 c32 = ucarve32(u, N, H, W);
        if (ws_bytes < c32.total) return set_error(INNFER_ERR_WORKSPACE, "unet_forward: workspace %zu < %zu bytes", ws_bytes, c32.total);
        return unet_forward_f32(u, (const float*)d_in, (float*)d_out, N, H, W, (char*)d_ws, (hipStream_t)stream);
    }
    const UCarve cv = ucarve(u, N, H, W);
    if (ws_bytes < cv.total) return set_error(INNFER_ERR_WORKSPACE, "unet_forward: workspace %zu < %zu bytes", ws_bytes, cv.total);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d_ws;
    float* raw = (float*)(ws + cv.raw);
    float* splitk = (float*)(ws + cv.splitk);
    float* mean = (float*)(ws + cv.mean);
    float* rstd = (float*)(ws + cv.rstd);
    float* bnpart = (float*)(ws + cv.bnpart);
    int dy16[16], dx16[16];
    for (int t = 0; t < 16; ++t) { dy16[t] = (t >> 2) - 1; dx16[t] = (t & 3) - 1; }

    int ks_last = 1;                  // segments the last GEMM left in `splitk` (1: its result is in raw)
    auto post = [&](const Layer& l, long HW, bool bn, PostDst d0, PostDst d1) -> int {
        // a conv that no norm layer follows but that has a bias (instance-norm nets: outermost / innermost down conv): y = 1 * x + bias
        const bool nb = !bn && !l.transposed && l.d_bias && l.d_ones;
        if (HW <= DEEP_PX) {          // deep level: reduce + statistics + post in one launch
            GtScope gt(s, "unet_deep_post (split-K reduce + BatchNorm + views)", 0.0, (double)N * HW * l.cout_pad * 4.0 * ks_last + (double)N * HW * l.cout * 2.0 * ((d0.p ? 1 : 0) + (d1.p ? 1 : 0)));
#define INNFER_DEEP_POST(NPX_, PL_) hipLaunchKernelGGL((unet_deep_post<NPX_, PL_>), dim3((l.cout + 31) / 32, N), dim3(32 * PL_), 0, s, ks_last > 1 ? (const float*)splitk : (const float*)raw, \
                                   (long)N * HW * l.cout_pad, ks_last, l.cout_pad, l.cout, (int)HW, 1e-5f, bn && !ev ? l.d_gamma : nullptr, bn && !ev ? l.d_beta : nullptr, \
                                   bn && ev ? l.d_ev_alpha : (nb ? l.d_ones : nullptr), bn && ev ? l.d_ev_shift : (nb ? l.d_bias : nullptr), d0, d1)
            if (HW == 1) INNFER_DEEP_POST(1, 1);
            else if (HW <= 4) INNFER_DEEP_POST(1, 4);
            else if (HW <= 16) INNFER_DEEP_POST(1, 16);
            else if (HW <= 64)
                hipLaunchKernelGGL(unet_deep_post<2>, dim3((l.cout + 31) / 32, N), dim3(1024), 0, s, ks_last > 1 ? (const float*)splitk : (const float*)raw,
                                   (long)N * HW * l.cout_pad, ks_last, l.cout_pad, l.cout, (int)HW, 1e-5f, bn && !ev ? l.d_gamma : nullptr, bn && !ev ? l.d_beta : nullptr,
                                   bn && ev ? l.d_ev_alpha : (nb ? l.d_ones : nullptr), bn && ev ? l.d_ev_shift : (nb ? l.d_bias : nullptr), d0, d1);
            else
                hipLaunchKernelGGL(unet_deep_post<8>, dim3((l.cout + 31) / 32, N), dim3(1024), 0, s, ks_last > 1 ? (const float*)splitk : (const float*)raw,
                                   (long)N * HW * l.cout_pad, ks_last, l.cout_pad, l.cout, (int)HW, 1e-5f, bn && !ev ? l.d_gamma : nullptr, bn && !ev ? l.d_beta : nullptr,
                                   bn && ev ? l.d_ev_alpha : (nb ? l.d_ones : nullptr), bn && ev ? l.d_ev_shift : (nb ? l.d_bias : nullptr), d0, d1);
            INNFER_HIP(hipGetLastError());
            return INNFER_OK;
        }
        if (bn && !ev) {
            GtScope gt(s, "norm statistics (fp32 rows)", 0.0, (double)N * HW * l.cout_pad * 4.0);
            int rc = norm::launch_stats(raw, l.cout_pad, HW, 1e-5f, l.d_gamma, l.d_beta, mean, rstd, l.cout, N, bnpart, s);
            if (rc) return rc;
        }
        const long total = (long)N * HW * (l.cout / 8);
        GtScope gt(s, "unet_post (BatchNorm + views from fp32 rows)", 0.0, (double)N * HW * l.cout_pad * 4.0 + (double)N * HW * l.cout * 2.0 * ((d0.p ? 1 : 0) + (d1.p ? 1 : 0)));
        hipLaunchKernelGGL(unet_post, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, raw, l.cout_pad, l.cout, HW, N,
                           bn ? (ev ? l.d_ev_alpha : mean) : (nb ? l.d_ones : nullptr), bn ? (ev ? l.d_ev_shift : rstd) : (nb ? l.d_bias : nullptr),
                           (bn && !ev) ? l.cout : 0, d0, d1);
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    };

    const bool first_mfma = u->down[0].patch && u->down[0].d_wf && L > 1;
    // One stored form of the outermost skip tensor.  The first conv's result t feeds the next down conv as LeakyReLU(t) and the outermost up conv's concatenation as
    // ReLU(t).  That up conv (HBM-bound: 128 -> 3 channels) takes max(stored, 0) as it reads its fragments from LDS (ConvLaunch.in_relu; relu(fp16(lrelu(t))) ==
    // fp16(relu(t)) bit for bit), so only lrelu(t) is stored -- in the concatenation buffer, where the next down conv reads it too: 134 MB of the 64 x 256^2
    // forward's writes less (first conv 94 -> 63 us, the up conv +2 us).  The inner levels keep two stored views: their up convs are matrix-bound and the operand
    // ReLU shares the MFMAs' issue port (measured: +7.7 % on those launches against 6 .. 12 us less in the post pass; scripts/r4/unet_one_view.sh).
    const int two_views = INNFER_KNOB("INNFER_UNET_TWO_VIEWS", 0);
    auto one_view = [&](int k) { return !two_views && k == 0 && L > 1 && !u->up[0].upconv && u->up[0].phases; };
    if (first_mfma) {          // the outermost down conv reads the NCHW input itself
    } else if (u->down[0].patch) {   // NCHW input -> 64-channel patch slab at half resolution
        const long M = (long)N * (H / 2) * (W / 2);
        hipLaunchKernelGGL(unet_pre_patch, dim3((unsigned)((2 * M + 255) / 256)), dim3(256), 0, s, d_in, in_dtype == INNFER_F32, u->in_nc, H, W, N,
                           (f16*)(ws + cv.x0), M * 32);
        INNFER_HIP(hipGetLastError());
    } else {                  // NCHW input -> one zero-padded 32-channel group
        const long HW = (long)H * W;
        hipLaunchKernelGGL(unet_pre, dim3((unsigned)((N * HW + 255) / 256)), dim3(256), 0, s, d_in, in_dtype == INNFER_F32, u->in_nc, HW, N, (f16*)(ws + cv.x0));
        INNFER_HIP(hipGetLastError());
    }
    // ---- down path ----
    const f16* cur = (const f16*)(ws + cv.x0);
    long cur_g = (long)N * H * W * 32;
    int h = H, w = W;
    for (int k = 0; k < L; ++k) {
        const Layer& l = u->down[k];
        const int ho = h / 2, wo = w / 2;
        int rc;
        if (k == 0 && first_mfma) {
            const long HWo = (long)ho * wo, Go = (long)N * HWo * 32;
            if ((long)u->in_nc * H * W >= 0x7fffffffL || (long)N * ho >= 0x7fffffffL) return set_error(INNFER_ERR_UNSUPPORTED, "unet_forward: image too large");
            const unsigned grid = (unsigned)std::min<long>(((long)N * ho + 3) / 4, 256L * 8);
            const int abl = INNFER_KNOB("INNFER_FIRST_ABL", 0);
            const bool ov = one_view(0);
            GtScope gt(s, "unet_first_mfma (3 -> 64, 4x4 s2)", 2.0 * 16 * u->in_nc * l.cout * (double)N * HWo,
                       (double)N * H * W * u->in_nc * (in_dtype == INNFER_F32 ? 4.0 : 2.0) + (ov ? 1.0 : 2.0) * N * HWo * l.cout * 2.0);
            f16* dl = (f16*)(ws + (ov ? cv.CAT[0] : cv.D[0]));
            f16* dr = ov ? nullptr : (f16*)(ws + cv.CAT[0]);
            if (in_dtype == INNFER_F32)
                hipLaunchKernelGGL(unet_first_mfma<float>, dim3(grid), dim3(256), 0, s, (const float*)d_in, u->in_nc, H, W, N, (const f16*)l.d_wf,
                                   (const float*)(l.bias >= 0 ? l.d_bias : nullptr), dl, dr, Go, abl);
            else
                hipLaunchKernelGGL(unet_first_mfma<f16>, dim3(grid), dim3(256), 0, s, (const f16*)d_in, u->in_nc, H, W, N, (const f16*)l.d_wf,
                                   (const float*)(l.bias >= 0 ? l.d_bias : nullptr), dl, dr, Go, abl);
            INNFER_HIP(hipGetLastError());
            cur = dl; cur_g = Go;
            h = ho; w = wo;
            continue;
        }
        if (l.patch && l.d_w3 && k < L - 1) {
            // no BatchNorm behind the outermost conv: the two views of its output (lrelu for the next conv, relu for the concatenation) come
            // straight out of the conv epilogue, one launch each (HBM-bound launches; the GEMM + post pair makes three passes)
            const long HWo = (long)ho * wo, Go = (long)N * HWo * 32;
            const bool ov = one_view(k);
            f16* dsts[2] = {(f16*)(ws + (ov ? cv.CAT[k] : cv.D[k])), (f16*)(ws + cv.CAT[k])};
            for (int v = 0; v < (ov ? 1 : 2); ++v) {
                ConvLaunch Lc{};
                Lc.in = cur; Lc.in_gstride = Go; Lc.C = 64;
                Lc.wpk = (const f16*)l.d_w3; Lc.bias = l.d_b3;
                Lc.out = dsts[v]; Lc.out_gstride = Go; Lc.K = l.cout; Lc.N = N; Lc.H = ho; Lc.W = wo; Lc.act = v == 0 ? 1 : 2;
                Lc.s1 = Lc.s2 = 1.f; Lc.y0 = 0; Lc.y1 = ho; Lc.out_mode = OUT_SLAB; Lc.conv1x1 = 1;
                rc = conv_launch(Lc, s);
                if (rc) return rc;
            }
            cur = dsts[0]; cur_g = Go;
            h = ho; w = wo;
            continue;
        }
        if (l.tile4 && k < L - 1 && fills_tiles(ho, wo)) {
            // output grid fills the 16 x 32 tiles: the halo-tile kernel with the stride-2 gather loader, fp16 slab out, statistics and the two views from it
            const long HWo = (long)ho * wo, Go = (long)N * HWo * 32;
            f16* Y = (f16*)raw;
            ConvLaunch Lc{};
            Lc.in = cur; Lc.in_gstride = cur_g; Lc.C = l.cin;
            Lc.wpk = (const f16*)l.d_w3; Lc.bias = l.d_b3;
            Lc.out = Y; Lc.out_gstride = Go; Lc.K = l.cout; Lc.N = N; Lc.H = ho; Lc.W = wo; Lc.act = 0;
            Lc.s1 = Lc.s2 = 1.f; Lc.y0 = 0; Lc.y1 = ho; Lc.out_mode = OUT_SLAB; Lc.stride2 = 1;
            if (!ev) Lc.stats_part = bnpart;              // statistics as per-tile partials out of the conv epilogue
            rc = conv_launch(Lc, s);
            if (rc) return rc;
            const long total = (long)N * HWo * (l.cout / 8);
            PostDst dl{(f16*)(ws + cv.D[k]), Go, 0, 1}, dr{(f16*)(ws + cv.CAT[k]), Go, 0, 2};
            if (one_view(k)) { dl.p = dr.p; dr = PostDst{nullptr, 0, 0, 0}; }
            if (!ev) {           // the records are merged inside the post pass
                rc = launch_post_slab_parts((const f16*)Y, Go, l.cout, HWo, N, bnpart, conv_stats_nper(ho, wo, 1), l.d_gamma, l.d_beta, dl, dr, s);
                if (rc) return rc;
            } else {
                GtScope gt(s, "unet_post_slab (eval-mode BatchNorm + views)", 0.0, (double)N * HWo * l.cout * 2.0 * 3);
                hipLaunchKernelGGL(unet_post_slab, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const f16*)Y, Go, l.cout, HWo, N,
                                   (const float*)l.d_ev_alpha, (const float*)l.d_ev_shift, 0, dl, dr);
                INNFER_HIP(hipGetLastError());
            }
            cur = dl.p; cur_g = Go;
            h = ho; w = wo;
            continue;
        }
        if (l.patch) {
            const int zero = 0;
            rc = run_gemm(l, l.d_w[0], cur, (long)N * ho * wo * 32, N, ho, wo, raw, ho, wo, 1, 1, &zero, &zero, ho, wo, 1, 0, 0, s, splitk);
        } else {
            ks_last = 1;
            rc = run_gemm(l, l.d_w[0], cur, cur_g, N, h, w, raw, ho, wo, 2, 16, dy16, dx16, ho, wo, 1, 0, 0, s, splitk, 0,
                          (long)ho * wo <= DEEP_PX ? &ks_last : nullptr);
        }
        if (rc) return rc;
        const long HW = (long)ho * wo, G = (long)N * HW * 32;
        if (k < L - 1) {
            PostDst dl{(f16*)(ws + cv.D[k]), G, 0, 1};              // lrelu(t): next down conv (and the skip's stored form)
            PostDst dr{(f16*)(ws + cv.CAT[k]), G, 0, 2};            // relu(t): first half of the up conv's concatenation
            if (one_view(k)) { dl.p = dr.p; dr = PostDst{nullptr, 0, 0, 0}; }
            rc = post(l, HW, k > 0, dl, dr);
            cur = dl.p; cur_g = G;
        } else {
            PostDst dr{(f16*)(ws + cv.r_inner), G, 0, 2};           // innermost: only relu(conv) is consumed
            rc = post(l, HW, false, dr, PostDst{nullptr, 0, 0, 0});
        }
        if (rc) return rc;
        h = ho; w = wo;
    }
    // ---- up path ----
    for (int k = L - 1; k >= 0; --k) {
        const Layer& l = u->up[k];
        const f16* in = k == L - 1 ? (const f16*)(ws + cv.r_inner) : (const f16*)(ws + cv.CAT[k]);
        const long in_g = (long)N * h * w * 32;
        const int hf = 2 * h, wf = 2 * w;
        // a ConvTranspose whose input grid fills the 16 x 32 tiles (>= 70 % real pixels) runs as four 2x2-tap phase convs on the halo-tile kernel
        const bool tile4 = l.tile4 && fills_tiles(h, w);
        if (l.upconv || tile4) {   // upconv: nearest 2x + 3x3 conv on the halo-tile kernel; outermost: bias + tanh -> NCHW in its planar epilogue
            ConvLaunch Lc{};
            Lc.in = in; Lc.in_gstride = in_g; Lc.C = l.cin;
            Lc.wpk = (const f16*)l.d_w3; Lc.bias = l.d_b3;
            Lc.K = l.cout; Lc.N = N; Lc.H = hf; Lc.W = wf; Lc.up = 1;
            Lc.s1 = Lc.s2 = 1.f; Lc.y0 = 0; Lc.y1 = hf;
            if (tile4) { Lc.K = 4 * l.cout; Lc.phase_c = l.cout; Lc.deconv_phases = 1; Lc.H = h; Lc.W = w; Lc.up = 0; Lc.y1 = h; }
            const long HW = (long)hf * wf, G = (long)N * HW * 32;
            if (k == 0) {
                Lc.out = d_out; Lc.act = 3; Lc.out_mode = OUT_NCHW; Lc.out_f32 = out_dtype == INNFER_F32;
                int rc = conv_launch(Lc, s);
                if (rc) return rc;
            } else {
                f16* Y = (f16*)raw;       // fp16 conv result; the norm statistics are taken from it (as behind resnet.hip's upconv layers)
                Lc.out = Y; Lc.out_gstride = G; Lc.act = 0; Lc.out_mode = OUT_SLAB;
                const bool epi_stats = !ev && l.cout % 64 == 0;          // statistics as per-tile partials out of the conv epilogue
                if (epi_stats) Lc.stats_part = bnpart;
                int rc = conv_launch(Lc, s);
                if (rc) return rc;
                PostDst dr{(f16*)(ws + cv.CAT[k - 1]), G, u->dc[k - 1], 2};
                if (epi_stats) {           // the records are merged inside the post pass
                    rc = launch_post_slab_parts((const f16*)Y, G, l.cout, HW, N, bnpart, tile4 ? conv_stats_nper(h, w, 4) : conv_stats_nper(hf, wf, 1),
                                                l.d_gamma, l.d_beta, dr, PostDst{nullptr, 0, 0, 0}, s);
                    if (rc) return rc;
                } else {
                    if (!ev) {
                        GtScope gt(s, "norm statistics (fp16 slab)", 0.0, (double)N * HW * l.cout * 2.0);
                        rc = norm::launch_stats_slab(Y, G, HW, 1e-5f, l.d_gamma, l.d_beta, mean, rstd, l.cout, N, bnpart, s);
                        if (rc) return rc;
                    }
                    const long total = (long)N * HW * (l.cout / 8);
                    GtScope gt(s, "unet_post_slab (BatchNorm + views)", 0.0, (double)N * HW * l.cout * 2.0 * 2);
                    hipLaunchKernelGGL(unet_post_slab, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const f16*)Y, G, l.cout, HW, N,
                                       (const float*)(ev ? l.d_ev_alpha : mean), (const float*)(ev ? l.d_ev_shift : rstd), ev ? 0 : l.cout, dr, PostDst{nullptr, 0, 0, 0});
                    INNFER_HIP(hipGetLastError());
                }
            }
            h = hf; w = wf;
            continue;
        }
        if (l.phases) {       // outermost: 3x3 conv with 4*cout phase channels, bias, tanh and the phase scatter in the epilogue
            ConvLaunch Lc{};
            Lc.in = in; Lc.in_gstride = in_g; Lc.C = l.cin_pad;
            Lc.wpk = (const f16*)l.d_w3; Lc.bias = l.d_b3;
            Lc.out = d_out; Lc.K = 4 * l.cout; Lc.N = N; Lc.H = h; Lc.W = w;
            Lc.act = 3; Lc.s1 = Lc.s2 = 1.f; Lc.y0 = 0; Lc.y1 = h;
            Lc.out_mode = OUT_NCHW; Lc.out_f32 = out_dtype == INNFER_F32; Lc.phase_c = l.cout;
            Lc.in_relu = k < L - 1 && one_view(k);
            int rc = conv_launch(Lc, s);
            if (rc) return rc;
            h = hf; w = wf;
            continue;
        }
        // the outermost layer has out_nc (3) channels: its raw rows are 4..8 floats, not a 64-channel tile
        const int rs = k == 0 ? (l.cout + 3) / 4 * 4 : 0;
        {   // the four output phases as ONE grouped launch (4x the workgroups, wide tiles where the layer is big enough); deep layers are
            // also split over K inside that launch and reduced over the full-resolution grid in one pass
            int dy[16], dx[16];
            for (int ph = 0; ph < 4; ++ph) { int ky[4], kx[4]; phase_taps(ph >> 1, ph & 1, ky, kx, dy + 4 * ph, dx + 4 * ph); }
            ks_last = 1;
            int rc = gg::launch(l.d_w[0], l.cin_pad, l.cout_pad, in, in_g, N, h, w, raw, h, w, 1, 4, dy, dx, hf, wf, 2, 0, 0, 0, s,
                                splitk, splitk_bytes(N), rs, 0, 4, l.phase_elems * (long)sizeof(f16), 0, 0, 0, 1,
                                (long)hf * wf <= DEEP_PX ? &ks_last : nullptr);
            if (rc) return rc;
        }
        const long HW = (long)hf * wf;
        if (k > 0) {
            const long G = (long)N * HW * 32;
            PostDst dr{(f16*)(ws + cv.CAT[k - 1]), G, u->dc[k - 1], 2};     // second half of the parent's concatenation
            int rc = post(l, HW, true, dr, PostDst{nullptr, 0, 0, 0});
            if (rc) return rc;
        } else {
            GtScope gt(s, "unet_final (bias + tanh -> NCHW)", 0.0, (double)N * HW * ((l.cout + 3) / 4 * 4) * 4.0 + (double)N * HW * l.cout * (out_dtype == INNFER_F32 ? 4.0 : 2.0));
            hipLaunchKernelGGL(unet_final, dim3((unsigned)((N * HW + 255) / 256)), dim3(256), 0, s, raw, (l.cout + 3) / 4 * 4, l.cout, HW, N,
                               l.d_bias, d_out, out_dtype == INNFER_F32);
            INNFER_HIP(hipGetLastError());
        }
        h = hf; w = wf;
    }
    return INNFER_OK;
}
