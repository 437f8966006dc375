// Per-(image, channel) normalisation statistics shared by the pix2pix UNet (train-mode BatchNorm2d, unet.hip) and the CycleGAN
// ResNet (InstanceNorm2d, resnet.hip): mean and biased variance over H*W of the fp32 conv output, emitted as the affine
// transform  out = x * alpha + shift.
#pragma once
#include "common.h"

namespace innfer {
namespace norm {

// (alpha, shift) from the mean and biased variance of raw[N][HW][cpad], fp32, ONE read of raw:
// a workgroup holds a segment of SEG pixels x 32 channels in registers (32 pixel lanes x 32 values), takes the segment mean and
// the sum of squared deviations from it (the two-pass form, on registers); segments are combined in index order with the
// parallel-variance formula  M2 = sum M2_s + sum n_s (mean_s - mean)^2  -- deterministic, no atomics.
constexpr int SEG = 1024;

__device__ __forceinline__ void bn_write(float mu, float var, float eps, const float* gamma, const float* beta, float* alpha, float* shift,
                                         long n, int C, int c) {
    // the transform ATen applies (batch_norm_cpu_transform_input): out = x * alpha + shift,
    // alpha = invstd * weight, shift = bias - mean * alpha; without affine parameters (gamma == nullptr) weight 1, bias 0
    const float a = (1.0f / sqrtf(var + eps)) * (gamma ? gamma[c] : 1.0f);
    alpha[n * C + c] = a;
    shift[n * C + c] = (beta ? beta[c] : 0.f) - mu * a;
}

// SLAB: the source is an fp16 blocked-NHWC slab (raw = slab base, cpad = group stride in elements taken from `gs`)
template <bool SLAB>
static __global__ __launch_bounds__(1024) void stats_kernel(const void* src, long gs, int cpad, long HW, float eps, const float* gamma, const float* beta,
                                                 float* alpha, float* shift, int C, float* part, int nseg) {
    __shared__ float red[1024];
    const int n = blockIdx.y, cb = blockIdx.x * 32, sg = blockIdx.z;
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = cb + cl;
    const long p0 = (long)sg * SEG;
    const int cnt = (int)min((long)SEG, HW - p0);
    const float* base = SLAB ? nullptr : (const float*)src + ((long)n * HW + p0) * cpad + c;
    const f16* sbase = SLAB ? (const f16*)src + (c >> 5) * gs + ((long)n * HW + p0) * 32 + (c & 31) : nullptr;
    auto reduce32 = [&](float v) {                 // sum over the 32 pixel lanes of one channel, result in every lane
        red[threadIdx.x] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[cl + 32 * i];
        __syncthreads();
        return t;
    };
    float v[SEG / 32];
    const int iters = (cnt + 31) >> 5;             // uniform: deep layers have a handful of pixels per image
#pragma unroll
    for (int i = 0; i < SEG / 32; ++i) {
        v[i] = 0.f;
        if (i < iters) {
            const int px = pl + 32 * i;
            if (px < cnt) v[i] = SLAB ? (float)sbase[(long)px * 32] : base[(long)px * cpad];
        }
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < SEG / 32; i += 4) { s0 += v[i]; s1 += v[i + 1]; s2 += v[i + 2]; s3 += v[i + 3]; }
    const float mu = reduce32((s0 + s1) + (s2 + s3)) / (float)cnt;
    s0 = s1 = s2 = s3 = 0.f;
#pragma unroll
    for (int i = 0; i < SEG / 32; i += 4) {
        const float d0 = pl + 32 * i < cnt ? v[i] - mu : 0.f, d1 = pl + 32 * (i + 1) < cnt ? v[i + 1] - mu : 0.f,
                    d2 = pl + 32 * (i + 2) < cnt ? v[i + 2] - mu : 0.f, d3 = pl + 32 * (i + 3) < cnt ? v[i + 3] - mu : 0.f;
        s0 += d0 * d0; s1 += d1 * d1; s2 += d2 * d2; s3 += d3 * d3;
    }
    const float m2 = reduce32((s0 + s1) + (s2 + s3));
    if (pl == 0 && c < C) {
        if (nseg == 1) bn_write(mu, m2 / (float)HW, eps, gamma, beta, alpha, shift, n, C, c);
        else { float* q = part + (((long)n * C + c) * nseg + sg) * 2; q[0] = mu; q[1] = m2; }
    }
}

// The slab form with 16-byte loads: a thread holds 8 channels of 4 pixels (lane = pixel lane * 4 + channel octet), sums over the wave's 16 pixel
// lanes by a fixed xor butterfly, over the 16 waves in index order through LDS -- deterministic, same two-pass scheme as stats_kernel.
// (stats_kernel<true> reads 2 bytes per lane: 1.8 TB/s on a 134 MB slab; this one reads the slab at the streaming rate.)
static __global__ __launch_bounds__(1024) void stats_slab8_kernel(const f16* slab, long gs, long HW, float eps, const float* gamma, const float* beta,
                                                                  float* alpha, float* shift, int C, float* part, int nseg) {
    __shared__ float red[16][32];
    __shared__ float tot[32];
    const int n = blockIdx.y, cb = blockIdx.x * 32, sg = blockIdx.z;
    const int t = threadIdx.x, q = t & 3, pl = t >> 2, wv = t >> 6, lane = t & 63;
    const long p0 = (long)sg * SEG;
    const int cnt = (int)min((long)SEG, HW - p0);
    const f16* base = slab + (cb >> 5) * gs + ((long)n * HW + p0) * 32 + q * 8;
    f16x8 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = pl + 256 * i;
        if (px < cnt) v[i] = *(const f16x8*)(base + (long)px * 32);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = (f16)0.f;
    }
    auto reduce = [&](float (&s)[8]) {           // per-channel total over the workgroup, returned in s for this thread's 8 channels
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int m = 4; m < 64; m <<= 1) s[e] += __shfl_xor(s[e], m);
        }
        if (lane < 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wv][q * 8 + e] = s[e];
        }
        __syncthreads();
        if (t < 32) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) a += red[w][t];
            tot[t] = a;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = tot[q * 8 + e];
        __syncthreads();
    };
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ((float)v[0][e] + (float)v[1][e]) + ((float)v[2][e] + (float)v[3][e]);
    reduce(s);
    float mu[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) mu[e] = s[e] / (float)cnt;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d = pl + 256 * i < cnt ? (float)v[i][e] - mu[e] : 0.f;
            a += d * d;
        }
        s[e] = a;
    }
    reduce(s);
    if (pl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cb + q * 8 + e;
            if (c >= C) continue;
            if (nseg == 1) bn_write(mu[e], s[e] / (float)HW, eps, gamma, beta, alpha, shift, n, C, c);
            else { float* o = part + (((long)n * C + c) * nseg + sg) * 2; o[0] = mu[e]; o[1] = s[e]; }
        }
    }
}

// segments -> (alpha, shift): one thread per (image, channel), segments in index order
static __global__ void combine_kernel(const float* part, int nseg, long HW, float eps, const float* gamma, const float* beta,
                           float* alpha, float* shift, int C, int N) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * C) return;
    const float* q = part + i * nseg * 2;
    float sum = 0.f;
    for (int g = 0; g < nseg; ++g) sum += q[2 * g] * (float)min((long)SEG, HW - (long)g * SEG);
    const float mu = sum / (float)HW;
    float m2 = 0.f;
    for (int g = 0; g < nseg; ++g) {
        const float d = q[2 * g] - mu;
        m2 += q[2 * g + 1] + (float)min((long)SEG, HW - (long)g * SEG) * d * d;
    }
    bn_write(mu, m2 / (float)HW, eps, gamma, beta, alpha, shift, i / C, C, (int)(i % C));
}

// Partials written by the conv epilogue (conv3x3.hip epilogue_stats: nper records of (count, mean, M2) per image and channel,
// part[((n * nper + r) * C + c) * 3]) -> (alpha, shift).  A workgroup owns (image, 32 channels): lane pl of a channel merges records pl, pl + 32, ..
// with Chan's update, then the 32 lanes' results are merged in lane order -- deterministic.
__device__ __forceinline__ void chan_merge(float& n, float& mu, float& m2, float nb, float mub, float m2b) {
    if (nb <= 0.f) return;
    const float tot = n + nb, d = mub - mu;
    mu += d * (nb / tot);
    m2 += m2b + d * d * (n * nb / tot);
    n = tot;
}
static __global__ __launch_bounds__(1024) void combine_parts_kernel(const float* part, int nper, long HW, float eps, const float* gamma, const float* beta,
                                                                    float* alpha, float* shift, int C) {
    __shared__ float sn[32][33], sm[32][33], sq[32][33];
    const int n = blockIdx.y, cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float cnt = 0.f, mu = 0.f, m2 = 0.f;
    if (c < C)
        for (int r = pl; r < nper; r += 32) {
            const float* q = part + (((long)n * nper + r) * C + c) * 3;
            chan_merge(cnt, mu, m2, q[0], q[1], q[2]);
        }
    sn[pl][cl] = cnt; sm[pl][cl] = mu; sq[pl][cl] = m2;
    __syncthreads();
    if (pl == 0 && c < C) {
        for (int i = 1; i < 32; ++i) chan_merge(cnt, mu, m2, sn[i][cl], sm[i][cl], sq[i][cl]);
        bn_write(mu, m2 / (float)HW, eps, gamma, beta, alpha, shift, n, C, c);
    }
}
inline int launch_combine_parts(const float* part, int nper, long HW, float eps, const float* gamma, const float* beta, float* alpha, float* shift,
                                int C, int N, hipStream_t s) {
    hipLaunchKernelGGL(combine_parts_kernel, dim3((C + 31) / 32, N), dim3(1024), 0, s, part, nper, HW, eps, gamma, beta, alpha, shift, C);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}
inline size_t parts_floats(int C, int nper) { return (size_t)C * nper * 3; }      // per image

// floats of `part` scratch for a layer of C channels and HW pixels per image (per image)
inline size_t part_floats(int C, long HW) { return (size_t)C * (size_t)((HW + SEG - 1) / SEG) * 2; }

inline int launch_stats(const float* raw, int cpad, long HW, float eps, const float* gamma, const float* beta, float* alpha, float* shift,
                        int C, int N, float* part, hipStream_t s) {
    const int nseg = (int)((HW + SEG - 1) / SEG);
    hipLaunchKernelGGL(stats_kernel<false>, dim3((C + 31) / 32, N, nseg), dim3(1024), 0, s, (const void*)raw, 0L, cpad, HW, eps, gamma, beta, alpha, shift, C, part, nseg);
    INNFER_HIP(hipGetLastError());
    if (nseg > 1) {
        hipLaunchKernelGGL(combine_kernel, dim3((unsigned)(((long)N * C + 255) / 256)), dim3(256), 0, s, (const float*)part, nseg, HW, eps,
                           gamma, beta, alpha, shift, C, N);
        INNFER_HIP(hipGetLastError());
    }
    return INNFER_OK;
}

// the same statistics over an fp16 slab of C channels (group stride gs elements)
inline int launch_stats_slab(const f16* slab, long gs, long HW, float eps, const float* gamma, const float* beta, float* alpha, float* shift,
                             int C, int N, float* part, hipStream_t s) {
    const int nseg = (int)((HW + SEG - 1) / SEG);
    hipLaunchKernelGGL(stats_slab8_kernel, dim3((C + 31) / 32, N, nseg), dim3(1024), 0, s, slab, gs, HW, eps, gamma, beta, alpha, shift, C, part, nseg);
    INNFER_HIP(hipGetLastError());
    if (nseg > 1) {
        hipLaunchKernelGGL(combine_kernel, dim3((unsigned)(((long)N * C + 255) / 256)), dim3(256), 0, s, (const float*)part, nseg, HW, eps,
                           gamma, beta, alpha, shift, C, N);
        INNFER_HIP(hipGetLastError());
    }
    return INNFER_OK;
}

}  // namespace norm
}  // namespace innfer
