// First convolution of a generator: few input channels (1..4), NCHW planar input
// straight from the caller's tensor, fp16 blocked-NHWC slab output.
// Replaces `fea_conv = conv_block(in_nc, nf, 3)` (RRDBNet_arch.py:25, SRResNet_arch.py:24).
//
// 1728 MAC per pixel for 3->64: 0.01 % of an RRDBNet-23 forward, so this is a plain
// VALU kernel bound by its 128 B/pixel store: K/8 lanes per pixel, each lane owns
// 8 consecutive output channels and writes one 16-byte piece, so a wave writes
// whole contiguous pixels.  Weights live in LDS as fp32 [Cin*9][K].
#include "common.h"

namespace innfer {
namespace {

struct FP {
    const void* in; int in_f32; int Cin;
    const float* w; const float* bias;
    f16* out; long out_gstride; f16* out2; long out2_gstride;
    int K; long npix; int H, W; int act;
};

__global__ __launch_bounds__(256) void first_conv_kernel(const FP p) {
    extern __shared__ __attribute__((aligned(16))) float sw[];
    const int nw = p.Cin * 9 * p.K;
    for (int i = threadIdx.x; i < nw + p.K; i += 256) sw[i] = i < nw ? p.w[i] : p.bias[i - nw];
    __syncthreads();
    const int tpp = p.K >> 3;                         // lanes per pixel
    const int ppb = 256 / tpp;
    const int sub = threadIdx.x / tpp;
    if (sub >= ppb) return;
    const long pix = (long)blockIdx.x * ppb + sub;
    if (pix >= p.npix) return;
    const int cg = (threadIdx.x % tpp) * 8;
    const int x = (int)(pix % p.W);
    const int y = (int)((pix / p.W) % p.H);
    const long n = pix / ((long)p.W * p.H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = sw[nw + cg + e];
    for (int ci = 0; ci < p.Cin; ++ci) {
        const long plane = (n * p.Cin + ci) * (long)p.H * p.W;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int Y = y + r - 1;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int X = x + s - 1;
                float v = 0.f;
                if (Y >= 0 && Y < p.H && X >= 0 && X < p.W) {
                    const long o = plane + (long)Y * p.W + X;
                    v = p.in_f32 ? ((const float*)p.in)[o] : (float)((const f16*)p.in)[o];
                }
                const float* wk = sw + (ci * 9 + r * 3 + s) * p.K + cg;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(v, wk[e], acc[e]);
            }
        }
    }
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = acc[e];
        if (p.act == 1) f = f > 0.f ? f : 0.2f * f;
        else if (p.act == 2) f = f > 0.f ? f : 0.f;
        h[e] = (f16)f;
    }
    *(f16x8*)(p.out + (cg >> 5) * p.out_gstride + pix * 32 + (cg & 31)) = h;
    if (p.out2) *(f16x8*)(p.out2 + (cg >> 5) * p.out2_gstride + pix * 32 + (cg & 31)) = h;
}

}  // namespace

int first_conv_launch(const FirstConvLaunch& L, hipStream_t s) {
    if (L.K % 8 || L.K > 256 || L.K <= 0)
        return set_error(INNFER_ERR_UNSUPPORTED, "first conv: nf=%d must be a multiple of 8, <= 256", L.K);
    if (L.Cin < 1 || L.Cin > 8) return set_error(INNFER_ERR_UNSUPPORTED, "first conv: in_nc=%d unsupported", L.Cin);
    FP p{L.in, L.in_f32, L.Cin, L.w, L.bias, L.out, L.out_gstride, L.out2, L.out2_gstride,
         L.K, (long)L.N * L.H * L.W, L.H, L.W, L.act};
    const int ppb = 256 / (L.K / 8);
    const long grid = (p.npix + ppb - 1) / ppb;
    const size_t lds = (size_t)(L.Cin * 9 * L.K + L.K) * sizeof(float);
    if (lds > 64 * 1024) return set_error(INNFER_ERR_UNSUPPORTED, "first conv: %d x %d weights exceed LDS", L.Cin, L.K);
    hipLaunchKernelGGL(first_conv_kernel, dim3((unsigned)grid), dim3(256), lds, s, p);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

}  // namespace innfer
