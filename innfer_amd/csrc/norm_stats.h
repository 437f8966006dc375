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
    hipLaunchKernelGGL(stats_kernel<true>, dim3((C + 31) / 32, N, nseg), dim3(1024), 0, s, (const void*)slab, gs, 0, HW, eps, gamma, beta, alpha, shift, C, part, nseg);
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
