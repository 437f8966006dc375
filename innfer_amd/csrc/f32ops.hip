// fp32 building blocks for the generators whose fast engines compute in fp16 only (PAN, pix2pix UNet): the reference runs EVERY architecture in fp32 on the GPU
// under `-no_fp16` (run.py:345,421-422: the tensors' dtype is the arithmetic), and SURVEY 8c asks <= 1e-4 of it.  RRDBNet / SRResNet get there on the fp16
// matrix cores with (hi, lo) operand pairs (conv3x3.hip SPLIT); these two networks' graphs use convolution forms that engine has no split form of (4x4 stride-2,
// transposed, gates, 20 / 24 / 40-channel tensors), so their fp32 mode runs on plain fp32 NCHW tensors -- the reference's own layout -- with ONE generic
// convolution kernel on the fp32 matrix instruction (v_mfma_f32_16x16x4_f32: fp32 products, fp32 accumulation; 157 TFLOP/s peak) and a handful of
// pointwise kernels.  An accuracy mode: correctness and the reference's semantics first (every op is the textbook definition, cited below), speed second --
// it still runs one to two orders of magnitude above the reference's CPU path.
//
// f32conv: out[n][k][Y][X] = epilogue( bias[k] + sum_{tap, c} w[tap][c][k] * in_act(in[n][c][oy * isy + dy[tap]][ox * isx + dx[tap]]) ),  (Y, X) = (oy * osy + ooy, ox * osx + oox)
//   * nn.Conv2d(k, stride s, padding p, dilation 1): taps (ky, kx) with dy = ky - p, isy = s, osy = 1            (block.py:213-254, UNet_arch.py:107-118)
//   * nn.ConvTranspose2d(4, 2, 1): four launches, one per output phase (a, b): the two taps per axis that land on that parity, isy = 1, osy = 2, ooy = a   (UNet_arch.py:119-146)
//   * nn.Upsample(nearest 2x) in front of a conv (`up`): the conv walks the virtual 2H x 2W image, source pixel = virtual >> 1       (block.py:286-331,348-361)
//   * zero padding = the validity test of a tap; input views with a channel offset / stride: torch.cat is an offset, never a copy
//   epilogue: v = acc + bias; v = mul * sigmoid(v) (pixel attention, PAN_arch.py:21-55); activation; + residual; stored through an output view
// GEMM view: rows = 16 output channels (A = weights), columns = 16 consecutive output pixels (B = gathered input), reduction = (tap, 4 channels) per MFMA.
#include "common.h"

namespace innfer {

namespace {

__device__ __forceinline__ float f32_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.2f * v;
    if (act == 2) return v > 0.f ? v : 0.f;
    if (act == 3) return tanhf(v);
    if (act == 4) return 1.0f / (1.0f + expf(-v));
    return v;
}

// One wave: 32 output channels x 64 output pixels (2 x 4 MFMA tiles of 16 x 16); a block of 4 waves covers `kt_per_block` 32-channel tiles x (4 / kt_per_block)
// 64-pixel tiles.  Weights wp: [tap][C4][Kp][4] fp32 (Kp = K rounded up to 32, C4 = ceil(C / 4); zeros beyond K / C).
__global__ __launch_bounds__(256) void f32conv_kernel(const F32Conv p, int kt_per_block, int C4, int Kp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
    const int kt = blockIdx.y * kt_per_block + wave % kt_per_block;
    const int ptile = blockIdx.x * (4 / kt_per_block) + wave / kt_per_block;
    const int n = blockIdx.z;
    const int k0 = kt * 32;
    const long npx = (long)p.Ho * p.Wo;
    if (k0 >= Kp || (long)ptile * 64 >= npx) return;
    int oy[4], ox[4];
    bool pv[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const long q = (long)ptile * 64 + pt * 16 + li;
        pv[pt] = q < npx;
        const long qq = pv[pt] ? q : 0;
        oy[pt] = (int)(qq / p.Wo); ox[pt] = (int)(qq - (long)oy[pt] * p.Wo);
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[t][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* inb = p.in + (long)n * p.in_nstride;
    const int Hv = p.up ? 2 * p.Hin : p.Hin, Wv = p.up ? 2 * p.Win : p.Win;       // the (virtual) image the taps walk
    for (int tap = 0; tap < p.ntap; ++tap) {
        long off[4];
        bool ok[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            int vy = oy[pt] * p.isy + p.dy[tap], vx = ox[pt] * p.isx + p.dx[tap];
            if (p.pad_mode == 1) {                                   // reflection (pad < size: one fold is enough)
                vy = vy < 0 ? -vy : (vy >= Hv ? 2 * Hv - 2 - vy : vy);
                vx = vx < 0 ? -vx : (vx >= Wv ? 2 * Wv - 2 - vx : vx);
            } else if (p.pad_mode == 2) {                            // replication
                vy = min(max(vy, 0), Hv - 1); vx = min(max(vx, 0), Wv - 1);
            }
            ok[pt] = pv[pt] && vy >= 0 && vy < Hv && vx >= 0 && vx < Wv;
            const int iy = p.up ? vy >> 1 : vy, ix = p.up ? vx >> 1 : vx;
            off[pt] = ok[pt] ? (long)iy * p.Win + ix : 0;
        }
        const float* wt = p.wp + ((long)tap * C4 * Kp + k0 + li) * 4 + lg;
        for (int c4 = 0; c4 < C4; ++c4) {
            const int c = c4 * 4 + lg;
            const float a0 = wt[(long)c4 * Kp * 4], a1 = wt[(long)c4 * Kp * 4 + 64];
            const float* ic = inb + (long)(c < p.C ? c : 0) * p.in_cstride;
            float b[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                float v = (ok[pt] && c < p.C) ? ic[off[pt]] : 0.f;
                if (p.in_act == 1) v = v > 0.f ? v : 0.2f * v;
                else if (p.in_act == 2) v = v > 0.f ? v : 0.f;
                b[pt] = v;
            }
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                acc[0][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[pt], acc[0][pt], 0, 0, 0);
                acc[1][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[pt], acc[1][pt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        if (!pv[pt]) continue;
        const long opix = ((long)(oy[pt] * p.osy + p.ooy) * p.Wout + ox[pt] * p.osx + p.oox);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + 16 * t + 4 * lg + j;
                if (k >= p.K) continue;
                float v = acc[t][pt][j] + (p.bias ? p.bias[k] : 0.f);
                if (p.mul) v = p.mul[(long)n * p.mul_nstride + (long)k * p.mul_cstride + opix] * (1.0f / (1.0f + expf(-v)));
                v = f32_act(v, p.act);
                if (p.oscale != 0.f) v *= p.oscale;
                if (p.res) v += p.res[(long)n * p.res_nstride + (long)k * p.res_cstride + opix];
                p.out[(long)n * p.out_nstride + (long)k * p.out_cstride + opix * p.out_pstride] = v;
            }
    }
}

// Normalisation of one (image, channel) plane per block, fp32: mode 0 nn.BatchNorm2d in TRAINING mode on the statistics of the image (run.py runs pix2pix with
// meval=False, one image at a time: biased variance, eps, affine), 1 eval mode on running statistics, 2 nn.InstanceNorm2d (no affine), 3 a given per-channel
// transform y = x * weight[c] + bias[c] (eval mode with ATen's precomputed alpha / shift).  Then the activation, then
// the store through the output view (a channel offset into a concatenation).  Two passes over the plane for the statistics (mean, then squared deviations).
__global__ __launch_bounds__(256) void f32_norm_kernel(const float* in, long in_ns, long in_cs, float* out, long out_ns, long out_cs, int C, long HW, int mode, float eps,
                                                       const float* weight, const float* bias, const float* rmean, const float* rvar, int act,
                                                       const float* res, long res_ns, long res_cs) {
    __shared__ float red[256];
    const int c = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const float* x = in + (long)n * in_ns + (long)c * in_cs;
    float mean, var;
    if (mode == 3) {
        float* y3 = out + (long)n * out_ns + (long)c * out_cs;
        const float a3 = weight[c], s3 = bias[c];
        const float* r3 = res ? res + (long)n * res_ns + (long)c * res_cs : nullptr;
        for (long i = tid; i < HW; i += 256) y3[i] = f32_act(x[i] * a3 + s3, act) + (r3 ? r3[i] : 0.f);
        return;
    }
    if (mode == 1) {
        mean = rmean[c]; var = rvar[c];
    } else {
        float s = 0.f;
        for (long i = tid; i < HW; i += 256) s += x[i];
        red[tid] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
        mean = red[0] / (float)HW;
        __syncthreads();
        float q = 0.f;
        for (long i = tid; i < HW; i += 256) { const float d = x[i] - mean; q += d * d; }
        red[tid] = q;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
        var = red[0] / (float)HW;
    }
    const float inv = 1.0f / sqrtf(var + eps);
    const float al = mode == 2 ? inv : inv * weight[c], sh = mode == 2 ? -mean * inv : bias[c] - mean * inv * weight[c];
    float* y = out + (long)n * out_ns + (long)c * out_cs;
    const float* rr = res ? res + (long)n * res_ns + (long)c * res_cs : nullptr;
    for (long i = tid; i < HW; i += 256) y[i] = f32_act(x[i] * al + sh, act) + (rr ? rr[i] : 0.f);
}

__global__ void f32_act_copy_kernel(const float* in, long in_ns, float* out, long out_ns, long per_image, int N, int act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_image * N) return;
    const long n = i / per_image, r = i - n * per_image;
    out[n * out_ns + r] = f32_act(in[n * in_ns + r], act);
}

// nn.MaxPool2d(4) (block.py:414): NCHW planes
__global__ void f32_maxpool4_kernel(const float* in, float* out, long planes, int H, int W, int hp, int wp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * hp * wp) return;
    const int x = (int)(i % wp), y = (int)((i / wp) % hp);
    const long pl = i / ((long)wp * hp);
    const float* b = in + pl * (long)H * W + (long)(4 * y) * W + 4 * x;
    float m = -INFINITY;
    for (int dy = 0; dy < 4; ++dy)
        for (int dx = 0; dx < 4; ++dx) m = fmaxf(m, b[(long)dy * W + dx]);
    out[i] = m;
}

__device__ __forceinline__ float cub1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cub2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// out = gamma * bicubic(att, size = (H, W), align_corners = False) + inp   (block.py:463-471, ATen upsample_bicubic2d: A = -0.75, clamped taps);
// att: fp32 rows [N][hp * wp][C]; inp / out: NCHW fp32
__global__ void f32_fsa_combine_kernel(const float* att, int hp, int wp, int C, const float* inp, float* out, int N, int H, int W, const float* gamma) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * C * H * W) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H), c = (int)((i / ((long)W * H)) % C);
    const long n = i / ((long)W * H * C);
    const float A = -0.75f;
    const float sy = (float)hp / (float)H, sx = (float)wp / (float)W;
    const float ry = sy * ((float)Y + 0.5f) - 0.5f, rx = sx * ((float)X + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    const float ty = ry - (float)iy, tx = rx - (float)ix;
    const float wy[4] = {cub2(ty + 1.f, A), cub1(ty, A), cub1(1.f - ty, A), cub2(2.f - ty, A)};
    const float wx[4] = {cub2(tx + 1.f, A), cub1(tx, A), cub1(1.f - tx, A), cub2(2.f - tx, A)};
    const float* an = att + n * (long)hp * wp * C + c;
    float v = 0.f;
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), hp - 1);
        float row = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = min(max(ix - 1 + b, 0), wp - 1);
            row += an[((long)yy * wp + xx) * C] * wx[b];
        }
        v += row * wy[a];
    }
    out[i] = gamma[0] * v + inp[i];
}

// F.interpolate(scale_factor = f, mode = 'bilinear', align_corners = False) on NCHW fp32 planes (PAN ups_inter_mode 'bilinear', block.py:286-323)
__global__ void f32_bilinear_up_kernel(const float* in, float* out, long planes, int h, int w, int f) {
    const int H2 = f * h, W2 = f * w;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H2 * W2) return;
    const int x = (int)(i % W2), y = (int)((i / W2) % H2);
    const long pl = i / ((long)W2 * H2);
    const float inv = 1.0f / (float)f;
    const float sy = fmaxf(inv * ((float)y + 0.5f) - 0.5f, 0.f), sx = fmaxf(inv * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* b = in + pl * (long)h * w;
    out[i] = hy * (hx * b[(long)y0 * w + x0] + lx * b[(long)y0 * w + x1]) + ly * (hx * b[(long)y1 * w + x0] + lx * b[(long)y1 * w + x1]);
}

__global__ void f32_nearest_up_kernel(const float* in, float* out, long planes, int h, int w, int f) {
    const int H2 = f * h, W2 = f * w;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H2 * W2) return;
    const int x = (int)(i % W2), y = (int)((i / W2) % H2);
    const long pl = i / ((long)W2 * H2);
    out[i] = in[pl * (long)h * w + (long)(y / f) * w + x / f];
}

// out = up2x(in) + skip on NCHW fp32 planes.  pt: F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) (ATen: source index max(0, (dst + 0.5) / 2 - 0.5), second tap
// clamped); tf: tf_2xupsample_bilinear (WBCNet_arch.py:126-137): even positions copy, odd ones the mean with the next pixel (replicated at the border)
__global__ void f32_upadd_kernel(const float* in, const float* skip, float* out, long planes, int h, int w, int tf_mode) {
    const int H2 = 2 * h, W2 = 2 * w;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H2 * W2) return;
    const int x = (int)(i % W2), y = (int)((i / W2) % H2);
    const long pl = i / ((long)W2 * H2);
    const float* b = in + pl * (long)h * w;
    float v;
    if (tf_mode) {
        const int y0 = y >> 1, x0 = x >> 1, y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float a = b[(long)y0 * w + x0];
        if (!(y & 1) && !(x & 1)) v = a;
        else if ((y & 1) && !(x & 1)) v = (a + b[(long)y1 * w + x0]) / 2.f;
        else if (!(y & 1) && (x & 1)) v = (a + b[(long)y0 * w + x1]) / 2.f;
        else v = (a + b[(long)y1 * w + x1]) / 2.f;
    } else {
        const float sy = fmaxf(0.5f * ((float)y + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * ((float)x + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        v = hy * (hx * b[(long)y0 * w + x0] + lx * b[(long)y0 * w + x1]) + ly * (hx * b[(long)y1 * w + x0] + lx * b[(long)y1 * w + x1]);
    }
    out[i] = v + skip[i];
}

__global__ void f32_axpy_kernel(const float* x, const float* y, float* out, float a, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + y[i];
}

// t: [N][groups * gc][hw]; channel c of group k becomes LeakyReLU(0.2)(sum of channel c of groups 0 .. k), the sums formed in group order (PPON's
// cat(d1, d1 + d2, .., d1 + .. + d8) -> act: PPON_arch.py:104-114)
__global__ void f32_prefix_lrelu_kernel(float* t, int N, int groups, int gc, long hw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * gc * hw) return;
    const long px = i % hw, c = (i / hw) % gc, n = i / (hw * gc);
    float* b = t + (n * groups * gc + c) * hw + px;
    float run = 0.f;
    for (int k = 0; k < groups; ++k) {
        run += b[(long)k * gc * hw];
        b[(long)k * gc * hw] = run > 0.f ? run : 0.2f * run;
    }
}

}  // namespace

size_t f32conv_packed_floats(int K, int C, int ntap) { return (size_t)ntap * ((C + 3) / 4) * ((K + 31) / 32 * 32) * 4; }

// w(k, c, tap) -> [tap][C4][Kp][4]
void f32conv_pack(int K, int C, int ntap, const std::function<float(int, int, int)>& w, float* packed) {
    const int C4 = (C + 3) / 4, Kp = (K + 31) / 32 * 32;
    for (int t = 0; t < ntap; ++t)
        for (int c4 = 0; c4 < C4; ++c4)
            for (int k = 0; k < Kp; ++k)
                for (int e = 0; e < 4; ++e) {
                    const int c = c4 * 4 + e;
                    packed[(((size_t)t * C4 + c4) * Kp + k) * 4 + e] = (k < K && c < C) ? w(k, c, t) : 0.f;
                }
}

int f32conv_launch(const F32Conv& L, hipStream_t s) {
    if (L.ntap < 1 || L.ntap > 49 || L.C < 1 || L.K < 1 || L.N < 1 || L.Ho < 1 || L.Wo < 1) return set_error(INNFER_ERR_INVALID, "f32conv: bad arguments");
    const int C4 = (L.C + 3) / 4, Kp = (L.K + 31) / 32 * 32, nkt = Kp / 32;
    const int ktb = nkt >= 4 ? 4 : (nkt >= 2 ? 2 : 1);
    const long npx = (long)L.Ho * L.Wo, ptiles = (npx + 63) / 64;
    const int ptb = 4 / ktb;
    F32Conv k = L;
    if (k.out_pstride == 0) k.out_pstride = 1;
    GtScope gt(s, "f32conv (fp32 MFMA, -no_fp16 mode)", 2.0 * L.N * (double)npx * L.K * L.C * L.ntap, (double)L.N * npx * (L.C * L.ntap / (double)(L.isy * L.isx) + L.K) * 4.0);
    hipLaunchKernelGGL(f32conv_kernel, dim3((unsigned)((ptiles + ptb - 1) / ptb), (unsigned)((nkt + ktb - 1) / ktb), (unsigned)L.N), dim3(256), 0, s, k, ktb, C4, Kp);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int f32_norm_launch(const float* in, long in_ns, long in_cs, float* out, long out_ns, long out_cs, int N, int C, long HW, int mode, float eps,
                    const float* weight, const float* bias, const float* rmean, const float* rvar, int act, hipStream_t s,
                    const float* res, long res_ns, long res_cs) {
    hipLaunchKernelGGL(f32_norm_kernel, dim3(C, N), dim3(256), 0, s, in, in_ns, in_cs, out, out_ns, out_cs, C, HW, mode, eps, weight, bias, rmean, rvar, act, res, res_ns, res_cs);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int f32_act_copy_launch(const float* in, long in_ns, float* out, long out_ns, long per_image, int N, int act, hipStream_t s) {
    const long tot = per_image * N;
    hipLaunchKernelGGL(f32_act_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, in, in_ns, out, out_ns, per_image, N, act);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int f32_maxpool4_launch(const float* in, float* out, long planes, int H, int W, hipStream_t s) {
    const int hp = H / 4, wp = W / 4;
    const long tot = planes * hp * wp;
    hipLaunchKernelGGL(f32_maxpool4_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, in, out, planes, H, W, hp, wp);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int f32_fsa_combine_launch(const float* att, int hp, int wp, int C, const float* inp, float* out, int N, int H, int W, const float* gamma, hipStream_t s) {
    const long tot = (long)N * C * H * W;
    hipLaunchKernelGGL(f32_fsa_combine_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, att, hp, wp, C, inp, out, N, H, W, gamma);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int f32_upsample_launch(const float* in, float* out, long planes, int h, int w, int f, int bilinear, hipStream_t s) {
    const long tot = planes * (long)h * w * f * f;
    if (bilinear) hipLaunchKernelGGL(f32_bilinear_up_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, in, out, planes, h, w, f);
    else hipLaunchKernelGGL(f32_nearest_up_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, in, out, planes, h, w, f);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int f32_upadd_launch(const float* in, const float* skip, float* out, long planes, int h, int w, int tf_mode, hipStream_t s) {
    const long tot = planes * 4 * h * w;
    hipLaunchKernelGGL(f32_upadd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, in, skip, out, planes, h, w, tf_mode);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int f32_axpy_launch(const float* x, const float* y, float* out, float a, long n, hipStream_t s) {
    hipLaunchKernelGGL(f32_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, out, a, n);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

int f32_prefix_lrelu_launch(float* t, int N, int groups, int gc, long hw, hipStream_t s) {
    const long tot = (long)N * gc * hw;
    hipLaunchKernelGGL(f32_prefix_lrelu_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, t, N, groups, gc, hw);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

}  // namespace innfer
