// Gather-GEMM on MFMA shared by the UNet and PAN engines: out[pixel][co] = sum over taps and input
// channels of in[pixel displaced by the tap][ci] * W[tap][ci][co].  One 64-pixel x 64-channel tile per
// workgroup, operands staged through LDS with plain loads (correctness-first; the SR hot path uses
// conv3x3.hip).  Input: blocked-NHWC fp16 slab; output: fp32 [pixel][cout_pad].
#pragma once
#include "common.h"

#include <vector>

namespace innfer {
namespace gg {

struct GP {
    const f16* in; long in_g; int nchunks; int N, Hin, Win;
    const f16* wpk;                       // [cot][tap][chunk][64 rows][64 B], the LDS image of the A operand
    float* out; int cout_pad;             // raw fp32 [N*Hfull*Wfull][cout_pad]
    int Ho, Wo, stride;                   // this launch's output grid; in = out*stride + d
    int ntaps; int dy[16], dx[16];
    int Hfull, Wfull, os, ooy, oox;       // out pixel = (oy*os + ooy, ox*os + oox)
    int up;                               // input is read through nearest-2x upsampling (Hin, Win = source size)
};

static __global__ __launch_bounds__(256) void gemm_gather(const GP p) {
    __shared__ __attribute__((aligned(16))) char lds[8192];
    char* lds_b = lds;                    // 64 pixels x 64 B
    char* lds_a = lds + 4096;             // 64 out channels x 64 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const long M = (long)p.N * p.Ho * p.Wo;
    const long m0 = (long)blockIdx.x * 64;
    const int cot = blockIdx.y;

    // staging role of this thread: pixel row (tid>>2), 16-byte slot (tid&3)
    const int srow = tid >> 2, sslot = tid & 3;
    const long sm = m0 + srow;
    int sn = 0, soy = 0, sox = 0;
    const bool sm_ok = sm < M;
    if (sm_ok) {
        sox = (int)(sm % p.Wo);
        soy = (int)((sm / p.Wo) % p.Ho);
        sn = (int)(sm / ((long)p.Wo * p.Ho));
    }
    const int cslot = sslot ^ (((srow >> 2) & 1) << 1);           // channel slot stored at LDS slot sslot

    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* wbase = (const char*)p.wpk + (long)cot * p.ntaps * p.nchunks * 4096 + tid * 16;
    const int brow = wave * 16 + li;
    const int boff = brow * 64 + ((lg ^ (((brow >> 2) & 1) << 1)) << 4);
    const int aoff = li * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4);

    for (int t = 0; t < p.ntaps; ++t) {
        int iy = soy * p.stride + p.dy[t], ix = sox * p.stride + p.dx[t];
        bool ok = sm_ok && iy >= 0 && ix >= 0;
        if (p.up) { ok = ok && iy < 2 * p.Hin && ix < 2 * p.Win; iy >>= 1; ix >>= 1; }
        else ok = ok && iy < p.Hin && ix < p.Win;
        const f16* src = p.in + (((long)sn * p.Hin + iy) * p.Win + ix) * 32 + cslot * 8;
        for (int c = 0; c < p.nchunks; ++c) {
            u32x4 vb = u32x4{0u, 0u, 0u, 0u};
            if (ok) vb = *(const u32x4*)(src + (long)c * p.in_g);
            const u32x4 va = *(const u32x4*)(wbase + ((long)t * p.nchunks + c) * 4096);
            __syncthreads();
            *(u32x4*)(lds_b + tid * 16) = vb;
            *(u32x4*)(lds_a + tid * 16) = va;
            __syncthreads();
            const f16x8 b = *(const f16x8*)(lds_b + boff);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f16x8 a = *(const f16x8*)(lds_a + q * 1024 + aoff);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
            }
        }
    }
    // D rows = out channels (16*lg + 4*q + j after the panel permutation), cols = pixels
    const long m = m0 + wave * 16 + li;
    if (m < M) {
        const int ox = (int)(m % p.Wo);
        const int oy = (int)((m / p.Wo) % p.Ho);
        const long n = m / ((long)p.Wo * p.Ho);
        const long opix = (n * p.Hfull + (long)oy * p.os + p.ooy) * p.Wfull + (long)ox * p.os + p.oox;
        float* op = p.out + opix * p.cout_pad + cot * 64 + 16 * lg;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)(op + 4 * q) = acc[q];
    }
}


// ---- host: weight panels -----------------------------------------------------------------------
// row R = q*16 + rho of a 64-channel tile holds out channel cot*64 + 16*(rho>>2) + 4*q + (rho&3);
// LDS slot s of that row holds input channels chunk*32 + 8*(s ^ 2*bit2(R)) .. +7
template <typename W>
inline void pack_panels(std::vector<f16>& dst, int cout, int cin, int cin_pad, int ntaps, W weight_of) {
    const int cots = (cout + 63) / 64, nch = cin_pad / 32;
    dst.assign((size_t)cots * ntaps * nch * 64 * 32, (f16)0.f);
    size_t o = 0;
    for (int cot = 0; cot < cots; ++cot)
        for (int t = 0; t < ntaps; ++t)
            for (int c = 0; c < nch; ++c)
                for (int R = 0; R < 64; ++R) {
                    const int q = R >> 4, rho = R & 15;
                    const int co = cot * 64 + 16 * (rho >> 2) + 4 * q + (rho & 3);
                    for (int s = 0; s < 4; ++s) {
                        const int cg = s ^ (((R >> 2) & 1) << 1);
                        for (int e = 0; e < 8; ++e, ++o) {
                            const int ci = c * 32 + cg * 8 + e;
                            if (co < cout && ci < cin) dst[o] = (f16)weight_of(co, ci, t);
                        }
                    }
                }
}


// One launch.  dy/dx: ntaps displacements; stride: input step per output pixel (2 for the 4x4 s2 conv).
inline int launch(const f16* wpk, int cin_pad, int cout_pad, const f16* in, long in_g, int N, int Hin, int Win,
                  float* raw, int Ho, int Wo, int stride, int ntaps, const int* dy, const int* dx,
                  int Hfull, int Wfull, int os, int ooy, int oox, int up, hipStream_t s) {
    GP g{};
    g.in = in; g.in_g = in_g; g.nchunks = cin_pad / 32; g.N = N; g.Hin = Hin; g.Win = Win;
    g.wpk = wpk; g.out = raw; g.cout_pad = cout_pad;
    g.Ho = Ho; g.Wo = Wo; g.stride = stride; g.ntaps = ntaps;
    for (int t = 0; t < ntaps; ++t) { g.dy[t] = dy[t]; g.dx[t] = dx[t]; }
    g.Hfull = Hfull; g.Wfull = Wfull; g.os = os; g.ooy = ooy; g.oox = oox; g.up = up;
    const long M = (long)N * Ho * Wo;
    if (M <= 0) return INNFER_OK;
    dim3 grid((unsigned)((M + 63) / 64), (unsigned)(cout_pad / 64));
    hipLaunchKernelGGL(gemm_gather, grid, dim3(256), 0, s, g);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

}  // namespace gg
}  // namespace innfer
