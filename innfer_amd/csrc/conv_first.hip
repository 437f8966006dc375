// First convolution of a generator: few input channels (1..4), NCHW planar input
// straight from the caller's tensor, fp16 blocked-NHWC slab output.
// Replaces `fea_conv = conv_block(in_nc, nf, 3)` (RRDBNet_arch.py:25, SRResNet_arch.py:24).
//
// 1728 MAC per pixel for 3->64: 0.01 % of an RRDBNet-23 forward, bound by its 128 B/pixel store.  On the matrix cores with split fp32 operands (first_conv_mfma
// below); the VALU kernel of round 1 it replaced (K/8 lanes per pixel) was removed in round 6: every engine's first conv has 32 or 64 outputs.
#include "common.h"
#include <type_traits>

namespace innfer {
namespace {

struct FP;
__device__ __forceinline__ float first_conv_input(const FP& p, long n, int ci, int Y, int X, long hw);

struct FP {
    const void* in; int in_f32; int Cin;
    int in_u8, in_norm, in_round16;
    const float* w; const float* bias;
    f16* out; long out_gstride; f16* out2; long out2_gstride;
    int K; long npix; int H, W; int act;
    long out_lo, out2_lo;          // != 0: fp32-accurate mode -- the lo part fp16((f - hi) * 2^11) of every value goes this many elements behind its hi part
};

// One input value of the first conv: planar fp16 / fp32, or np2tensor of a uint8 HWC image (float32(u8) / 255, BGR -> RGB flip as in
// colors.py:5-21, optional ((x - 0.5) * 2).clamp(-1, 1), optional rounding to fp16) -- bit for bit what the separate pass produces.
__device__ __forceinline__ float first_conv_input(const FP& p, long n, int ci, int Y, int X, long hw) {
    if (p.in_u8) {
        int sc = ci;
        if (p.Cin % 3 == 0) sc = p.Cin - 1 - ci; else if (p.Cin == 4 && ci < 3) sc = 2 - ci;
        float v = __fdiv_rn((float)((const uint8_t*)p.in)[(n * hw + (long)Y * p.W + X) * p.Cin + sc], 255.0f);
        if (p.in_norm) v = fminf(fmaxf(__fmul_rn(__fsub_rn(v, 0.5f), 2.0f), -1.0f), 1.0f);
        if (p.in_round16) v = (float)(f16)v;
        return v;
    }
    const long o = (n * p.Cin + ci) * hw + (long)Y * p.W + X;
    return p.in_f32 ? ((const float*)p.in)[o] : (float)((const f16*)p.in)[o];
}

// The same conv on the matrix cores (K = 32 or 64 outputs): the 9 * Cin taps of a pixel are the k dimension of a 16 x 16 x 32 MFMA
// (27 of 32 for RGB; ceil(9 Cin / 32) steps), 16 consecutive pixels are its columns, 16 * NT output channels its rows.  A lane gathers
// the 8 patch values of its (pixel, k-octet) straight from the planar input -- 8 loads per lane and 16 pixels instead of the VALU
// kernel's 27 loads per lane and 8 pixels, no LDS -- and ends up with 4 * NT consecutive output channels of its pixel (rows permuted as in
// conv_pack): one or two 16-byte stores per output slab.  fp32 weights and fp32 input are split into an fp16 head and an fp16 remainder
// (w = wh + wl, x = xh + xl; wh*xh + wl*xh + wh*xl with fp32 accumulation), so the result keeps the fp32 VALU kernel's accuracy (product
// terms below 2^-22 relative dropped) although the MFMA operands are fp16.  Bound by its two 128 B/pixel stores, as it should be.
// (round 5: the walk.  The first form strode over 16-pixel groups of the flattened pixel index: two 64-bit divisions and eight bounds-tested, index-rebuilt
//  loads per lane and group -- ~500 VALU instructions for 8 MFMAs: 0.26 ms for a 1080p frame whose stores need 0.07-0.14.  Now a workgroup owns 64 columns x
//  FIRST_ROWS rows of one image, wave w its 16-column strip: a lane's eight patch offsets and column validity are fixed for the strip, a row costs eight loads
//  at base + offset, and the weight fragments are built once per FIRST_ROWS groups.  Same operands, same MFMA order: the same bits.)
constexpr int FIRST_ROWS = 16;
template <int NT, int STEPS, bool FAST16>                           // STEPS = ceil(9 Cin / 32): 1 for gray / RGB, 2 for 4..7 channels, 3 for 8; FAST16: planar fp16 input
__global__ __launch_bounds__(256) void first_conv_mfma(const FP p) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    const int nk = p.Cin * 9;
    const int x = blockIdx.x * 64 + (threadIdx.x >> 6) * 16 + li;    // this lane's column
    const int y0 = blockIdx.y * FIRST_ROWS, y1 = min(y0 + FIRST_ROWS, p.H);
    const long n = blockIdx.z;
    if (blockIdx.x * 64 + (int)(threadIdx.x >> 6) * 16 >= p.W) return;      // (a strip beyond the image: whole waves)
    // weight fragments, once per wave: row li of sub-tile t is output channel 32 (t >> 1) + 8 (li >> 2) + 4 (t & 1) + (li & 3) (the plane row order of conv3x3.hip's
    // 64-channel kernels): lane group lg ends with channels 8 lg .. 8 lg + 7 of EACH 32-channel slab plane of its pixel, so the four groups of a pixel column write its whole
    // 64-byte line of a plane and a store instruction covers 16 pixels x 64 bytes of ONE plane (round 5, with the row walk: the stores are what is left of this kernel)
    f16x8 wh[STEPS][NT], wl[STEPS][NT];
#pragma unroll
    for (int st = 0; st < STEPS; ++st)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int oc = 32 * (t >> 1) + 8 * (li >> 2) + 4 * (t & 1) + (li & 3);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = st * 32 + lg * 8 + e;
                const float wv = p.w[(long)min(kk, nk - 1) * p.K + oc];          // (unconditional load, then the select: a predicated load is a branch each -- 64 of them per wave)
                const float w = kk < nk ? wv : 0.f;
                const f16 h = (f16)w;
                wh[st][t][e] = h;
                wl[st][t][e] = (f16)(w - (float)h);
            }
        }
    f32x4 bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias[t] = *(const f32x4*)(p.bias + 32 * (t >> 1) + 8 * lg + 4 * (t & 1));
    const long hw = (long)p.H * p.W;
    const bool live = x < p.W;
    // the lane's patch elements: (channel, row offset, column) of k = 32 st + 8 lg + e; column validity never changes along the strip, row validity only on the
    // image's first and last row (bit masks, wave-uniform tests)
    int eoff[STEPS][8];                                              // FAST16: element offset from (row y, channel 0, column 0); else (channel << 20 | column + 1) -- W < 2^20 (launch)
    unsigned xok = 0, top = 0, bot = 0;
#pragma unroll
    for (int st = 0; st < STEPS; ++st)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = st * 32 + lg * 8 + e;
            const int ci = kk / 9, tap = kk - ci * 9, r = tap / 3, sx = tap - r * 3, X = x + sx - 1;
            eoff[st][e] = FAST16 ? (int)(ci * hw) + (r - 1) * p.W + X : (ci << 20) | (X + 1);
            if (live && kk < nk && X >= 0 && X < p.W) xok |= 1u << (st * 8 + e);
            if (r == 0) top |= 1u << (st * 8 + e);
            if (r == 2) bot |= 1u << (st * 8 + e);
        }
    const bool any_lo = !FAST16 && (p.in_f32 != 0 || (p.in_u8 && !p.in_round16));
    [[maybe_unused]] const f16* in16 = (const f16*)p.in + n * p.Cin * hw;
    // One row of the strip: (FAST16) `raw` holds the row's 8 STEPS patch values as loaded -- requested one row ahead, first touched here.
    auto process = [&](int y, const unsigned (&raw)[STEPS][8]) __attribute__((always_inline)) {
        const unsigned ok = xok & (y == 0 ? ~top : ~0u) & (y == p.H - 1 ? ~bot : ~0u);
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = bias[t];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {             // compile-time index into the fragment arrays (a runtime one would send them to scratch)
            f16x8 xh, xl;
            if constexpr (FAST16) {                      // the values are the hi operands as they lie in memory, no lo part
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool o = (ok >> (st * 8 + e)) & 1;
                    xh[e] = o ? __builtin_bit_cast(f16, (unsigned short)raw[st][e]) : (f16)0.f;
                    xl[e] = (f16)0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int kk = st * 32 + lg * 8 + e;
                    const int r = (kk - (kk / 9) * 9) / 3;
                    float v = 0.f;
                    if ((ok >> (st * 8 + e)) & 1) v = first_conv_input(p, n, eoff[st][e] >> 20, y + r - 1, (eoff[st][e] & 0xfffff) - 1, hw);
                    const f16 h = (f16)v;
                    xh[e] = h;
                    xl[e] = (f16)(v - (float)h);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[st][t], xh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[st][t], xh, acc[t], 0, 0, 0);
                if (any_lo) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[st][t], xl, acc[t], 0, 0, 0);
            }
        }
        if (!live) return;
        const long pix = n * hw + (long)y * p.W + x;
        f16 h[4 * NT], l[4 * NT];
        // (the activation chosen ONCE per row: a uniform test per value is a branch per value in this unrolled code -- 16 of them, with the accumulators copied around each)
        auto finish = [&](auto act_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float f = acc[t][j];
                    if (ACT == 1) f = __builtin_amdgcn_fmed3f(f, 0.2f * f, 3.0e38f);      // (one instruction behind the multiply; a finite top: conv3x3_epilogue_slab.h ACT_TOP)
                    else if (ACT == 2) f = __builtin_amdgcn_fmed3f(f, 0.f, 3.0e38f);
                    h[4 * t + j] = (f16)f;
                    l[4 * t + j] = (f16)((f - (float)h[4 * t + j]) * 2048.0f);
                }
        };
        if (p.act == 1) finish(std::integral_constant<int, 1>{}); else if (p.act == 2) finish(std::integral_constant<int, 2>{}); else finish(std::integral_constant<int, 0>{});
        const long o = pix * 32 + 8 * lg;
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {                            // plane q: tiles 2 q, 2 q + 1
            f16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = h[8 * q + e];
            *(f16x8*)(p.out + q * p.out_gstride + o) = v;
            if (p.out2) *(f16x8*)(p.out2 + q * p.out2_gstride + o) = v;
            if (p.out_lo) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = l[8 * q + e];
                *(f16x8*)(p.out + p.out_lo + q * p.out_gstride + o) = v;
                if (p.out2) *(f16x8*)(p.out2 + p.out2_lo + q * p.out2_gstride + o) = v;
            }
        }
    };
    if constexpr (FAST16) {
        // unconditional loads from a clamped offset (an element outside the image reads the row's own first value, zeroed when it is used): independent, all in flight
        // together, and the NEXT row's are requested before this row is multiplied and stored (two register sets used alternately, as in unet_first_mfma)
        auto request = [&](int y, unsigned (&raw)[STEPS][8]) __attribute__((always_inline)) {
            const unsigned ok = xok & (y == 0 ? ~top : ~0u) & (y == p.H - 1 ? ~bot : ~0u);
            const unsigned short* row = (const unsigned short*)in16 + (long)y * p.W;
#pragma unroll
            for (int st = 0; st < STEPS; ++st)
#pragma unroll
                for (int e = 0; e < 8; ++e) raw[st][e] = row[((ok >> (st * 8 + e)) & 1) ? eoff[st][e] : 0];
        };
        unsigned rawA[STEPS][8], rawB[STEPS][8];
        request(y0, rawA);
        int y = y0;
        for (; y + 1 < y1; y += 2) {
            request(y + 1, rawB);
            process(y, rawA);
            if (y + 2 < y1) request(y + 2, rawA);
            process(y + 1, rawB);
        }
        if (y < y1) process(y, rawA);
    } else {
        unsigned none[STEPS][8] = {};
        for (int y = y0; y < y1; ++y) process(y, none);
    }
}

}  // namespace

int first_conv_launch(const FirstConvLaunch& L, hipStream_t s) {
    if (L.K % 8 || L.K > 256 || L.K <= 0)
        return set_error(INNFER_ERR_UNSUPPORTED, "first conv: nf=%d must be a multiple of 8, <= 256", L.K);
    if (L.Cin < 1 || L.Cin > 8) return set_error(INNFER_ERR_UNSUPPORTED, "first conv: in_nc=%d unsupported", L.Cin);
    FP p{L.in, L.in_f32, L.Cin, L.in_u8, L.in_norm, L.in_round16, L.w, L.bias, L.out, L.out_gstride, L.out2, L.out2_gstride,
         L.K, (long)L.N * L.H * L.W, L.H, L.W, L.act, L.out_lo, L.out2_lo};
    if (L.K == 32 || L.K == 64) {
        if (L.N > 65535 || (L.H + FIRST_ROWS - 1) / FIRST_ROWS > 65535) return set_error(INNFER_ERR_UNSUPPORTED, "first conv: %d images of %d rows exceed the launch grid", L.N, L.H);
        const dim3 grid((unsigned)((L.W + 63) / 64), (unsigned)((L.H + FIRST_ROWS - 1) / FIRST_ROWS), (unsigned)L.N);
        const int steps = (L.Cin * 9 + 31) / 32;
        const bool fast16 = !L.in_u8 && !L.in_f32 && (long)L.Cin * L.H * L.W < 0x7fffffffL;
        if (L.W >= (1 << 20) - 1) return set_error(INNFER_ERR_UNSUPPORTED, "first conv: %d columns", L.W);
#define FC(NT_, ST_) do { if (fast16) hipLaunchKernelGGL((first_conv_mfma<NT_, ST_, true>), grid, dim3(256), 0, s, p); \
                          else hipLaunchKernelGGL((first_conv_mfma<NT_, ST_, false>), grid, dim3(256), 0, s, p); } while (0)
        if (L.K == 64) { if (steps == 1) FC(4, 1); else if (steps == 2) FC(4, 2); else FC(4, 3); }
        else { if (steps == 1) FC(2, 1); else if (steps == 2) FC(2, 2); else FC(2, 3); }
#undef FC
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    }
    return set_error(INNFER_ERR_UNSUPPORTED, "first conv: K = %d (built: 32 or 64 outputs on the matrix cores)", L.K);
}

}  // namespace innfer
