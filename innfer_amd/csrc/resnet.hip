// CycleGAN ResnetGenerator forward on gfx950 -- SURVEY.md section 8f row n4.
// Replaces ResnetGenerator(norm=instance, padding=reflect, upsample=deconv).forward with ResnetBlock
// (architectures/ResNet_arch.py:19-151; defaults utils/defaults.py:124-140: ngf 64, 9 (or 6) blocks).
//
//   ReflectionPad2d(1) + 3x3 conv (the residual blocks)               the SR path's halo-tile kernel (conv3x3.hip, ConvLaunch.reflect):
//                                                                     fp16 slab out, statistics / normalisation on that slab
//   ReflectionPad2d(3) + 7x7 conv                                     gg::gemm_gather, out-of-image taps read the mirrored pixel
//                                                                     (GP.reflect); the first one (3 input channels) reads a
//                                                                     row-patch slab: 7 vertical taps (rn_pre); the last one (64 -> 3,
//                                                                     tanh) is nine displaced 3x3 convs on the halo-tile kernel
//   3x3 stride-2 zero-pad-1 convs                                     gg::gemm_gather (stride 2)
//   ConvTranspose2d(3, stride 2, padding 1, output_padding 1)         four output phases (1, 2, 2, 4 taps) of the same GEMM
//   InstanceNorm2d (no affine, statistics of the instance also under eval)   norm_stats.h: per-(image, channel) mean / biased variance in
//                                                                     fp32, one read; the conv bias cancels in the norm
//   ReLU / residual add / Tanh                                        rn_post, rn_final
// Activations are blocked-NHWC fp16 slabs, GEMM results fp32 rows.  A batch is N independent images.
#include "common.h"
#include "gather_gemm.h"
#include "norm_stats.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace innfer;

namespace {

// InstanceNorm statistics: norm_stats.h without affine parameters, i.e. alpha = 1/sqrt(var + eps), shift = (bias - mean_of(x + bias)) * alpha =
// -mean(x) * alpha: the conv bias cancels inside an instance norm, which is why the GEMM result is normalised without it

// raw fp32 -> instance norm -> [ReLU] -> [+ residual] -> fp16 slab; one thread per (pixel, 8 channels)
__global__ void rn_post(const float* raw, int cpad, int C, long HW, int N, const float* alpha, const float* shift, int relu,
                        const f16* res, f16* dst, long g) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C / 8;
    if (i >= (long)N * HW * c8) return;
    const int c = (int)(i % c8) * 8;
    const long pix = i / c8;
    const long n = pix / HW;
    const float* rp = raw + pix * cpad + c;
    const float* ap = alpha + n * C + c;
    const float* sp = shift + n * C + c;
    const long o = (c >> 5) * g + pix * 32 + (c & 31);
    const f32x4 x0 = *(const f32x4*)rp, x1 = *(const f32x4*)(rp + 4), a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 4),
                s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
    f16x8 r8;
    if (res) r8 = *(const f16x8*)(res + o);
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = (e < 4 ? x0[e & 3] : x1[e & 3]) * (e < 4 ? a0[e & 3] : a1[e & 3]) + (e < 4 ? s0[e & 3] : s1[e & 3]);
        if (relu) v = fmaxf(v, 0.f);
        if (res) v += (float)r8[e];
        h[e] = (f16)v;
    }
    *(f16x8*)(dst + o) = h;
}

// fp16 conv result (slab, bias included) -> instance norm -> [ReLU] -> [+ residual] -> fp16 slab (may be the source slab itself);
// one thread per (pixel, 8 channels).  Used behind the halo-tile convs of the residual blocks.
__global__ void rn_post_slab(const f16* src, int C, long HW, int N, const float* alpha, const float* shift, int relu,
                             const f16* res, f16* dst, long g) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C / 8;
    if (i >= (long)N * HW * c8) return;
    const int c = (int)(i % c8) * 8;
    const long pix = i / c8;
    const long n = pix / HW;
    const float* ap = alpha + n * C + c;
    const float* sp = shift + n * C + c;
    const long o = (c >> 5) * g + pix * 32 + (c & 31);
    const f16x8 x = *(const f16x8*)(src + o);
    f16x8 r8;
    if (res) r8 = *(const f16x8*)(res + o);
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = (float)x[e] * ap[e] + sp[e];
        if (relu) v = fmaxf(v, 0.f);
        if (res) v += (float)r8[e];
        h[e] = (f16)v;
    }
    *(f16x8*)(dst + o) = h;
}

// The last 7x7 conv (64 -> out_nc <= 3, ReflectionPad2d(3)) as ONE plain 3x3 conv with 9 * out_nc output channels + a 9-term gather: the 7x7
// kernel, zero-padded to 9x9, is nine 3x3 sub-blocks (sr, sc); channel (co, sr, sc) of the 3x3 conv is
//   Q[co, sr, sc][y'][x'] = sum_{r, s, ci} w9[co][ci][3 sr + r][3 sc + s] * pad(in)[ci][y' + r - 1][x' + s - 1]
// and  out[co][y][x] = tanh(bias + sum_{sr, sc} Q[co, sr, sc][y + 3 (sr - 1)][x + 3 (sc - 1)]).
// The sub-blocks sit in the MFMA's output rows (27 of 32 used) instead of being nine displaced passes over 3 of 16 rows (conv3x3_pc<.., S9>: 81 tap
// positions, 312 us at 16 x 256^2).  Q is needed 3 pixels beyond the image, from the reflection-padded input: the producer of the conv's input
// writes it as a (H + 8) x (W + 8) slab -- image at (4, 4), three mirrored rings, one outer ring of zeros that only meets the 9x9's zero border.
// rn_post_slab with `padW > 0`: destination is that padded slab (g = its group stride); a border pixel is written up to four times.
__device__ __forceinline__ int rn_mirror_targets(int y, int H, int* t) {       // padded rows that hold source row y (reflection of 3, offset 4)
    int n = 0;
    t[n++] = y + 4;
    if (y >= 1 && y <= 3) t[n++] = 4 - y;
    if (y >= H - 4 && y <= H - 2) t[n++] = 2 * (H - 1) - y + 4;
    return n;
}

__global__ void rn_post_slab_pad(const f16* src, int C, int H, int W, int N, const float* alpha, const float* shift, int relu, f16* dst, long gsrc, long gdst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C / 8;
    const long HW = (long)H * W;
    if (i >= (long)N * HW * c8) return;
    const int c = (int)(i % c8) * 8;
    const long pix = i / c8;
    const long n = pix / HW;
    const int y = (int)((pix - n * HW) / W), x = (int)(pix - n * HW - (long)y * W);
    const float* ap = alpha + n * C + c;
    const float* sp = shift + n * C + c;
    const f16x8 v = *(const f16x8*)(src + (c >> 5) * gsrc + pix * 32 + (c & 31));
    f16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = (float)v[e] * ap[e] + sp[e];
        if (relu) f = fmaxf(f, 0.f);
        h[e] = (f16)f;
    }
    int ty[3], tx[3];
    const int ny = rn_mirror_targets(y, H, ty), nx = rn_mirror_targets(x, W, tx);
    const int Wp = W + 8;
    f16* db = dst + (c >> 5) * gdst + n * (long)(H + 8) * Wp * 32 + (c & 31);
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) *(f16x8*)(db + ((long)ty[a] * Wp + tx[b]) * 32) = h;
}

// the outer ring of the padded slab (row / column 0 and H + 7 / W + 7): zeros
__global__ void rn_zero_ring(f16* dst, int G, int H, int W, int N, long g) {
    const int Hp = H + 8, Wp = W + 8, ring = 2 * Wp + 2 * (Hp - 2);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * G * ring * 4) return;
    const int q = (int)(i & 3);
    long r = i >> 2;
    const int k = (int)(r % ring); r /= ring;
    const int grp = (int)(r % G);
    const long n = r / G;
    int y, x;
    if (k < Wp) { y = 0; x = k; }
    else if (k < 2 * Wp) { y = Hp - 1; x = k - Wp; }
    else { const int j = k - 2 * Wp; y = 1 + (j >> 1); x = (j & 1) ? Wp - 1 : 0; }
    f16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (f16)0.f;
    *(f16x8*)(dst + grp * g + ((n * Hp + y) * (long)Wp + x) * 32 + q * 8) = z;
}

// out[n][co][y][x] = tanh(bias[co] + sum over the nine sub-blocks of Q[n][co * 9 + sr * 3 + sc][y + 3 sr + 1][x + 3 sc + 1]); Q: planar fp32 over the padded grid
__global__ void rn_sum9_tanh(const float* Q, const float* bias, void* out, int out_f32, int K, int H, int W, int N) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long HW = (long)H * W;
    if (i >= (long)N * K * HW) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const int co = (int)((i / HW) % K);
    const long n = i / (HW * K);
    const int Hp = H + 8, Wp = W + 8;
    const float* q = Q + ((n * 9 * K + co * 9) * Hp + (y + 1)) * (long)Wp + (x + 1);
    float a = bias[co];
#pragma unroll
    for (int sr = 0; sr < 3; ++sr)
#pragma unroll
        for (int sc = 0; sc < 3; ++sc) a += q[((long)(sr * 3 + sc) * Hp + 3 * sr) * Wp + 3 * sc];
    a = fast_tanh(a);
    if (out_f32) ((float*)out)[i] = a; else ((f16*)out)[i] = (f16)a;
}

// eval-mode BatchNorm: the per-channel transform of the running statistics, once per image (the post kernels index alpha / shift by image)
__global__ void rn_fill_ev(const float* ev_alpha, const float* ev_shift, float* alpha, float* shift, int C, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    alpha[i] = ev_alpha[i % C]; shift[i] = ev_shift[i % C];
}

// rn_post_slab behind a conv that left its statistics as partials (conv3x3.hip epilogue_stats): a workgroup owns PXB pixels of one (image, 32-channel
// group), merges that group's records itself -- 8 lanes per channel, Chan's update, lanes merged in order: every workgroup of the group computes the
// same (alpha, shift) -- and applies them: the combine launch between conv and post is gone.  pxb = 1024 pixels per workgroup where that still gives
// the chip a workgroup per CU, 256 below (one image: 0.887 -> 0.858 ms with 256, sixteen: 2.50 -> 2.46 ms with 1024).
__global__ __launch_bounds__(256) void rn_post_slab_parts(const f16* src, int C, long HW, const float* part, int nper, float eps, const float* gamma,
                                                          const float* beta, int relu, const f16* res, f16* dst, long g, int pxb) {
    __shared__ float sn[8][32], sm[8][32], sq[8][32], sal[32], ssh[32];
    const int n = blockIdx.z, cb = blockIdx.y * 32, t = threadIdx.x;
    {
        const int cl = t & 31, lg = t >> 5, c = cb + cl;
        float cnt = 0.f, mu = 0.f, m2 = 0.f;
        if (c < C)
            for (int r = lg; r < nper; r += 32) {          // four records per trip, their loads issued together (as unet_post_slab_parts, round 5): same merge order, same bits
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
            const float a = (1.0f / sqrtf(m2 / (float)HW + eps)) * ((gamma && c < C) ? gamma[c] : 1.0f);
            sal[cl] = a;
            ssh[cl] = ((beta && c < C) ? beta[c] : 0.f) - mu * a;
        }
        __syncthreads();
    }
    const int q8 = (t & 3) * 8, pl = t >> 2;
    float al[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { al[e] = sal[q8 + e]; sh[e] = ssh[q8 + e]; }
    const long base = (cb >> 5) * g + (long)n * HW * 32 + q8;
#pragma unroll 4
    for (int i = 0; i < pxb / 64; ++i) {
        const long px = (long)blockIdx.x * pxb + pl + 64 * i;
        if (px >= HW) break;
        const long o = base + px * 32;
        const f16x8 x = *(const f16x8*)(src + o);
        f16x8 r8;
        if (res) r8 = *(const f16x8*)(res + o);
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (float)x[e] * al[e] + sh[e];
            if (relu) v = fmaxf(v, 0.f);
            if (res) v += (float)r8[e];
            h[e] = (f16)v;
        }
        *(f16x8*)(dst + o) = h;
    }
}

// NCHW input -> "row patch" slab of the reflection-padded first 7x7 conv: channel kx*C + c of pixel (y, x) holds in[c][y][reflect(x + kx - 3)]
// (zero beyond 7*C <= 32 channels), so the 49-tap conv becomes 7 vertical taps (reflected by the GEMM's gather) over ONE 32-channel group.
// C > 4: plain copy into a zero-padded group (49 taps).  One thread per pixel, 16-byte stores.
__global__ void rn_pre(const void* in, int in_f32, int C, int H, int W, int N, f16* slab, int patch) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long HW = (long)H * W;
    if (i >= (long)N * HW) return;
    const long n = i / HW, px = i % HW;
    const int x = (int)(px % W);
    f16 v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        f16 t = (f16)0.f;
        if (patch) {
            const int kx = j / C, c = j - kx * C;
            int X = x + kx - 3;
            X = X < 0 ? -X : (X >= W ? 2 * W - 2 - X : X);
            if (kx < 7) {
                const long o = (n * C + c) * HW + px + (X - x);
                t = in_f32 ? (f16)((const float*)in)[o] : ((const f16*)in)[o];
            }
        } else if (j < C) {
            const long o = (n * C + j) * HW + px;
            t = in_f32 ? (f16)((const float*)in)[o] : ((const f16*)in)[o];
        }
        v[j] = t;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f16x8*)(slab + i * 32 + 8 * q) = *(const f16x8*)(v + 8 * q);
}

__global__ void rn_final(const float* raw, int rs, int C, long HW, int N, const float* bias, void* out, int out_f32) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * HW) return;
    const long n = i / HW, px = i % HW;
    for (int c = 0; c < C; ++c) {
        const float y = fast_tanh(raw[i * rs + c] + bias[c]);
        const long o = (n * C + c) * HW + px;
        if (out_f32) ((float*)out)[o] = y; else ((f16*)out)[o] = (f16)y;
    }
}

struct Param { std::string key; std::vector<int> shape; std::vector<float> host; bool set = false; };
struct Layer { int w = -1, b = -1, cin = 0, cout = 0, k = 3; bool transposed = false; std::vector<f16*> d_w; float* d_b = nullptr;
               void* d_w3 = nullptr;
               float* d_b4 = nullptr;      // ConvTranspose layers on the halo-tile kernel: the bias once per output phase
               int g = -1, be = -1, rm = -1, rv = -1;          // norm_type 'batch': the BatchNorm2d behind this conv (weight, bias, running_mean, running_var)
               float *d_gamma = nullptr, *d_beta = nullptr, *d_ev_alpha = nullptr, *d_ev_shift = nullptr;
               void* d_w27 = nullptr; float* d_z32 = nullptr;   // last 7x7 conv as a 3x3 conv over nine sub-blocks (rn_sum9_tanh): panels [9 * cout][cin][3][3], zero bias
               bool up2 = false; };        // upsample_mode 'upconv': Upsample(nearest 2x) + 3x3 conv on the halo-tile kernel (nearest-2x in the loader)    // residual-block convs: conv3x3.hip panels (reflection padding in the halo-tile loader)

}  // namespace

struct innfer_resnet {
    int in_nc = 3, out_nc = 3, ngf = 64, n_blocks = 9;
    int block_pad = 1;               // padding of the residual blocks' convs as a ConvLaunch.reflect code: 1 reflect, 2 replicate, 0 zero
    std::vector<Param> params;
    std::vector<Layer> layers;       // first, down1, down2, 2*n_blocks block convs, up1, up2, last
    bool fp32 = false;               // innfer_resnet_set_precision(1): the fp32 forward on NCHW fp32 tensors (f32ops.hip), the reference's -no_fp16 mode
    std::vector<std::vector<float*>> f32_w;   //   f32conv panels per layer (a ConvTranspose2d: one per output phase)
    bool uploaded = false;
    bool batch_norm = false;         // norm_type 'batch' (the constructor's default, ResNet_arch.py:19,39-49): nn.BatchNorm2d behind every conv but the last, and
                                     // no bias on those convs (use_bias is True for InstanceNorm2d only)
    bool eval_mode = false;          // BatchNorm on its running statistics (nn.Module.eval()); train(): the statistics of the image, like pix2pix's UNet
};

static int RP(innfer_resnet* r, const std::string& key, std::vector<int> shape) {
    Param q; q.key = key; q.shape = shape;
    r->params.push_back(q);
    return (int)r->params.size() - 1;
}

// norm: state-dict prefix of the norm layer that follows the conv ("" behind the last conv); with norm_type 'batch' the conv then has no bias and the
// BatchNorm2d's parameters and buffers follow it in the state dict
static void add_layer(innfer_resnet* r, const std::string& key, int cin, int cout, int k, bool transposed, const std::string& norm = "") {
    Layer l; l.cin = cin; l.cout = cout; l.k = k; l.transposed = transposed;
    l.w = RP(r, key + ".weight", transposed ? std::vector<int>{cin, cout, k, k} : std::vector<int>{cout, cin, k, k});
    const bool bn = r->batch_norm && !norm.empty();
    if (!bn) l.b = RP(r, key + ".bias", {cout});
    if (bn) {
        l.g = RP(r, norm + ".weight", {cout}); l.be = RP(r, norm + ".bias", {cout});
        l.rm = RP(r, norm + ".running_mean", {cout}); l.rv = RP(r, norm + ".running_var", {cout});
        RP(r, norm + ".num_batches_tracked", {});
    }
    r->layers.push_back(l);
}

extern "C" int innfer_resnet_create(innfer_resnet** out, int in_nc, int out_nc, int ngf, int n_blocks) {
    return innfer_resnet_create_ex(out, in_nc, out_nc, ngf, n_blocks, 0, 0, 0, 0);
}

extern "C" int innfer_resnet_set_eval(innfer_resnet* r, int eval_mode) {
    if (!r) return set_error(INNFER_ERR_INVALID, "resnet_set_eval: null handle");
    r->eval_mode = eval_mode != 0;
    return INNFER_OK;
}

extern "C" int innfer_resnet_create_ex(innfer_resnet** out, int in_nc, int out_nc, int ngf, int n_blocks, int padding, int use_dropout, int upconv, int batch_norm) {
    if (!out) return set_error(INNFER_ERR_INVALID, "resnet_create: null out");
    if (padding < 0 || padding > 2) return set_error(INNFER_ERR_INVALID, "resnet_create: padding %d (0 reflect, 1 replicate, 2 zero)", padding);
    if (ngf < 32 || ngf > 128 || ngf % 32 || in_nc < 1 || in_nc > 8 || out_nc < 1 || out_nc > 8 || n_blocks < 0 || n_blocks > 64)
        return set_error(INNFER_ERR_UNSUPPORTED, "resnet_create: ngf=%d n_blocks=%d (built: ngf 32, 64, 96, 128)", ngf, n_blocks);
    innfer_resnet* r = new innfer_resnet();
    r->in_nc = in_nc; r->out_nc = out_nc; r->ngf = ngf; r->n_blocks = n_blocks;
    r->batch_norm = batch_norm != 0;
    r->block_pad = padding == 0 ? 1 : (padding == 1 ? 2 : 0);        // ConvLaunch.reflect code of the residual blocks' convs
    // ResnetBlock.conv_block (ResNet_arch.py:118-146): [pad] conv norm relu [dropout] [pad] conv norm -- a pad layer for reflect / replicate only,
    // nn.Dropout(0.5) (identity in eval mode, which is how run.py runs CycleGAN generators: cyglegan_extras) when use_dropout
    const int c1 = padding == 2 ? 0 : 1, c2 = c1 + 3 + (use_dropout ? 1 : 0) + (padding == 2 ? 0 : 1);
    add_layer(r, "model.1", in_nc, ngf, 7, false, "model.2");
    add_layer(r, "model.4", ngf, 2 * ngf, 3, false, "model.5");
    add_layer(r, "model.7", 2 * ngf, 4 * ngf, 3, false, "model.8");
    for (int i = 0; i < n_blocks; ++i) {
        const std::string b = "model." + std::to_string(10 + i) + ".conv_block.";
        add_layer(r, b + std::to_string(c1), 4 * ngf, 4 * ngf, 3, false, b + std::to_string(c1 + 1));
        add_layer(r, b + std::to_string(c2), 4 * ngf, 4 * ngf, 3, false, b + std::to_string(c2 + 1));
    }
    const int i = 10 + n_blocks;
    if (upconv) {        // upconv_block (block.py:348-361): sequential(Upsample, Conv2d) inside the model's Sequential -> `model.<i>.1`
        add_layer(r, "model." + std::to_string(i) + ".1", 4 * ngf, 2 * ngf, 3, false, "model." + std::to_string(i + 1)); r->layers.back().up2 = true;
        add_layer(r, "model." + std::to_string(i + 3) + ".1", 2 * ngf, ngf, 3, false, "model." + std::to_string(i + 4)); r->layers.back().up2 = true;
    } else {
        add_layer(r, "model." + std::to_string(i), 4 * ngf, 2 * ngf, 3, true, "model." + std::to_string(i + 1));
        add_layer(r, "model." + std::to_string(i + 3), 2 * ngf, ngf, 3, true, "model." + std::to_string(i + 4));
    }
    add_layer(r, "model." + std::to_string(i + 7), ngf, out_nc, 7, false);
    *out = r;
    return INNFER_OK;
}

static void rn_free(innfer_resnet* r) {
    for (auto& vv : r->f32_w) for (auto v : vv) if (v) (void)hipFree(v);          // (the fp32 panels follow the parameters: rebuilt by innfer_resnet_set_precision)
    r->f32_w.clear();
    for (auto& l : r->layers) {
        for (auto w : l.d_w) if (w) (void)hipFree(w);
        l.d_w.clear();
        if (l.d_b) (void)hipFree(l.d_b);
        l.d_b = nullptr;
        if (l.d_w3) (void)hipFree(l.d_w3);
        l.d_w3 = nullptr;
        if (l.d_b4) (void)hipFree(l.d_b4);
        l.d_b4 = nullptr;
        for (float** q : {&l.d_gamma, &l.d_beta, &l.d_ev_alpha, &l.d_ev_shift}) { if (*q) (void)hipFree(*q); *q = nullptr; }
        if (l.d_w27) (void)hipFree(l.d_w27);
        l.d_w27 = nullptr;
        if (l.d_z32) (void)hipFree(l.d_z32);
        l.d_z32 = nullptr;
    }
}

extern "C" void innfer_resnet_destroy(innfer_resnet* r) {
    if (!r) return;
    rn_free(r);
    delete r;
}

extern "C" int innfer_resnet_num_params(innfer_resnet* r) { return r ? (int)r->params.size() : INNFER_ERR_INVALID; }

extern "C" int innfer_resnet_param_info(innfer_resnet* r, int idx, char* key, size_t key_cap, int* ndim, int* shape4) {
    if (!r || idx < 0 || idx >= (int)r->params.size()) return set_error(INNFER_ERR_INVALID, "resnet_param_info: bad index");
    const Param& q = r->params[idx];
    if (key && key_cap) { strncpy(key, q.key.c_str(), key_cap - 1); key[key_cap - 1] = 0; }
    if (ndim) *ndim = (int)q.shape.size();
    if (shape4) for (size_t i = 0; i < 4; ++i) shape4[i] = i < q.shape.size() ? q.shape[i] : 1;
    return INNFER_OK;
}

extern "C" int innfer_resnet_set_param(innfer_resnet* r, int idx, const float* h_data) {
    if (!r || idx < 0 || idx >= (int)r->params.size() || !h_data) return set_error(INNFER_ERR_INVALID, "resnet_set_param: bad arguments");
    Param& q = r->params[idx];
    size_t n = 1;
    for (int s : q.shape) n *= (size_t)s;
    q.host.assign(h_data, h_data + n);
    q.set = true;
    r->uploaded = false;
    return INNFER_OK;
}

namespace {

// ConvTranspose2d(3, s2, p1, op1): out[2i + a] takes in[i + d] * W[ky] for (a = 0: ky 1, d 0), (a = 1: ky 0, d +1; ky 2, d 0)
int phase_taps1d(int a, int ky[2], int d[2]) {
    if (a == 0) { ky[0] = 1; d[0] = 0; return 1; }
    ky[0] = 0; d[0] = 1; ky[1] = 2; d[1] = 0;
    return 2;
}

int rn_upload(innfer_resnet* r) {
    for (auto& q : r->params)
        if (!q.set && q.key.find("running_") == std::string::npos && q.key.find("num_batches") == std::string::npos)
            return set_error(INNFER_ERR_INVALID, "resnet: parameter '%s' was never set", q.key.c_str());
    rn_free(r);
    std::vector<f16> panel;
    for (auto& l : r->layers) {
        const std::vector<float>& w = r->params[l.w].host;
        const std::vector<float> bias = l.b >= 0 ? r->params[l.b].host : std::vector<float>((size_t)l.cout, 0.f);     // no conv bias in front of a BatchNorm2d
        if (l.g >= 0) {
            // train(): gamma / beta beside the statistics of the image; eval(): ATen's transform of the running statistics (alpha = weight /
            // sqrt(running_var + eps), shift = bias - running_mean * alpha; a checkpoint without them means a fresh BatchNorm's (0, 1))
            const std::vector<float>&g = r->params[l.g].host, &b = r->params[l.be].host;
            const Param &pm = r->params[l.rm], &pv = r->params[l.rv];
            std::vector<float> al((size_t)l.cout), sh((size_t)l.cout);
            for (int c = 0; c < l.cout; ++c) {
                const float mean = pm.set ? pm.host[c] : 0.f, var = pv.set ? pv.host[c] : 1.f;
                al[c] = g[c] * (1.0f / std::sqrt(var + 1e-5f));
                sh[c] = b[c] - mean * al[c];
            }
            for (auto pr : {std::make_pair(&l.d_gamma, &g), std::make_pair(&l.d_beta, &b), std::make_pair(&l.d_ev_alpha, (const std::vector<float>*)&al),
                            std::make_pair(&l.d_ev_shift, (const std::vector<float>*)&sh)}) {
                INNFER_HIP(hipMalloc((void**)pr.first, l.cout * sizeof(float)));
                INNFER_HIP(hipMemcpy(*pr.first, pr.second->data(), l.cout * sizeof(float), hipMemcpyHostToDevice));
            }
        }
        const int cin_pad = (l.cin + 31) / 32 * 32, k = l.k;
        auto put = [&](int ntaps, auto weight_of) -> int {
            gg::pack_panels(panel, l.cout, l.cin, cin_pad, ntaps, weight_of);
            f16* d = nullptr;
            INNFER_HIP(hipMalloc((void**)&d, panel.size() * sizeof(f16)));
            INNFER_HIP(hipMemcpy(d, panel.data(), panel.size() * sizeof(f16), hipMemcpyHostToDevice));
            l.d_w.push_back(d);
            return INNFER_OK;
        };
        if (!l.transposed && k == 7 && 7 * l.cin <= 32) {       // first conv on the row-patch slab (rn_pre): tap = ky, channel kx*cin + c
            gg::pack_panels(panel, l.cout, 7 * l.cin, 32, 7, [&](int co, int j, int ky) {
                const int kx = j / l.cin, c = j - kx * l.cin;
                return w[((size_t)co * l.cin + c) * 49 + ky * 7 + kx]; });
            f16* d = nullptr;
            INNFER_HIP(hipMalloc((void**)&d, panel.size() * sizeof(f16)));
            INNFER_HIP(hipMemcpy(d, panel.data(), panel.size() * sizeof(f16), hipMemcpyHostToDevice));
            l.d_w.push_back(d);
        } else if (!l.transposed) {
            int rc = put(k * k, [&](int co, int ci, int t) { return w[((size_t)co * l.cin + ci) * k * k + t]; });
            if (rc) return rc;
        } else {
            for (int ph = 0; ph < 4; ++ph) {
                int kyv[2], dyv[2], kxv[2], dxv[2];
                const int ny = phase_taps1d(ph >> 1, kyv, dyv), nx = phase_taps1d(ph & 1, kxv, dxv);
                int rc = put(ny * nx, [&](int co, int ci, int t) {
                    const int ky = kyv[t / nx], kx = kxv[t % nx];
                    return w[(((size_t)ci * l.cout + co) * 3 + ky) * 3 + kx];          // ConvTranspose2d weight is [in, out, kH, kW]
                });
                if (rc) return rc;
            }
        }
        INNFER_HIP(hipMalloc((void**)&l.d_b, l.cout * sizeof(float)));
        INNFER_HIP(hipMemcpy(l.d_b, bias.data(), l.cout * sizeof(float), hipMemcpyHostToDevice));
        if (!l.transposed && k == 7 && l.cin % 32 == 0 && l.cout <= 16) {          // c7s1-out: nine displaced 3x3 convs, ReflectionPad2d(3)
            std::vector<char> packed(conv_packed_bytes7x7(l.cout, l.cin));
            conv_pack7x7(w.data(), l.cout, l.cin, packed.data());
            INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
            INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
            std::vector<float> b3(64, 0.f);
            for (int c = 0; c < l.cout; ++c) b3[c] = bias[c];
            (void)hipFree(l.d_b); l.d_b = nullptr;
            INNFER_HIP(hipMalloc((void**)&l.d_b, b3.size() * sizeof(float)));
            INNFER_HIP(hipMemcpy(l.d_b, b3.data(), b3.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (!l.transposed && k == 7 && 7 * l.cin <= 32 && l.cout % 32 == 0) {
            // first conv over the row-patch slab (rn_pre) as a 7 x 1 column conv on the halo-tile kernel (three vertically displaced 3-tap blocks,
            // rows reflected by the loader): fp16 slab out, norm on the slab
            std::vector<float> wv((size_t)l.cout * 32 * 7, 0.f);
            for (int co = 0; co < l.cout; ++co)
                for (int kx = 0; kx < 7; ++kx)
                    for (int c = 0; c < l.cin; ++c)
                        for (int ky = 0; ky < 7; ++ky) wv[((size_t)co * 32 + kx * l.cin + c) * 7 + ky] = w[((size_t)co * l.cin + c) * 49 + ky * 7 + kx];
            std::vector<char> packed(conv_packed_bytes7v(l.cout, 32));
            conv_pack7v(wv.data(), l.cout, 32, packed.data());
            INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
            INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
        }
        if (!l.transposed && k == 7 && l.cin % 32 == 0 && 9 * l.cout <= 32) {
            const int K9 = 9 * l.cout;
            std::vector<float> w27((size_t)K9 * l.cin * 9, 0.f), z32(32, 0.f);
            for (int co = 0; co < l.cout; ++co)
                for (int sb = 0; sb < 9; ++sb)
                    for (int t = 0; t < 9; ++t) {
                        const int ky = 3 * (sb / 3) + t / 3 - 1, kx = 3 * (sb % 3) + t % 3 - 1;          // index into the 7x7 (the 9x9 has a zero border)
                        if (ky < 0 || ky > 6 || kx < 0 || kx > 6) continue;
                        for (int ci = 0; ci < l.cin; ++ci)
                            w27[((size_t)(co * 9 + sb) * l.cin + ci) * 9 + t] = w[(((size_t)co * l.cin + ci) * 7 + ky) * 7 + kx];
                    }
            std::vector<char> packed(conv_packed_bytes(K9, l.cin));
            conv_pack(w27.data(), K9, l.cin, packed.data());
            INNFER_HIP(hipMalloc(&l.d_w27, packed.size()));
            INNFER_HIP(hipMemcpy(l.d_w27, packed.data(), packed.size(), hipMemcpyHostToDevice));
            INNFER_HIP(hipMalloc((void**)&l.d_z32, z32.size() * sizeof(float)));
            INNFER_HIP(hipMemcpy(l.d_z32, z32.data(), z32.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (l.transposed && k == 3 && l.cout % 64 == 0 && l.cin % 32 == 0) {
            // ConvTranspose2d(3, stride 2, padding 1, output_padding 1) on the halo-tile kernel (conv3x3_pc<.., TM = 0x1B>, see unet.hip): output phase
            // (a, b) at the virtual pixel (y + a, x + b) reads taps (dy, dx) in {-1, 0}^2 with ky = 1 - a - 2 dy (oy = 2 iy - 1 + ky); ky = 3 does not
            // exist in a 3-tap kernel: a structural zero (9 of the 16 phase taps are real)
            std::vector<float> b4((size_t)4 * l.cout);
            for (int co = 0; co < 4 * l.cout; ++co) b4[co] = bias[co % l.cout];
            std::vector<char> packed(conv_packed_bytes_deconv2x(l.cout, l.cin));
            conv_pack_deconv2x(w.data(), l.cout, l.cin, 3, packed.data());
            INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
            INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
            INNFER_HIP(hipMalloc((void**)&l.d_b4, b4.size() * sizeof(float)));
            INNFER_HIP(hipMemcpy(l.d_b4, b4.data(), b4.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (l.up2 || (!l.transposed && k == 3 && l.cin == l.cout && l.cin % 64 == 0)) {       // ResnetBlock convs (stride 1, pad 1) and the upconv convs
            std::vector<char> packed(conv_packed_bytes(l.cout, l.cin));
            conv_pack(w.data(), l.cout, l.cin, packed.data());
            INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
            INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
        }
    }
    r->uploaded = true;
    return INNFER_OK;
}

struct RCarve { size_t x0, s1, s2, a, b, c, u1, u2, raw, alpha, shift, part, total; };

RCarve rcarve(const innfer_resnet* r, int N, int H, int W) {
    RCarve c{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t px = (size_t)N * H * W;
    size_t off = 0;
    auto slab = [&](size_t pixels, int ch) { size_t o = off; off += al(pixels * ch * 2); return o; };
    const int g1 = r->ngf, g2 = 2 * g1, g4 = 4 * g1, g1p = (g1 + 63) / 64 * 64;        // widths at full, half and quarter resolution; fp32 GEMM rows are 64-padded
    c.x0 = slab(px, 32); c.s1 = slab(px, g1p); c.s2 = slab(px / 4, g2);
    c.a = slab(px / 16, g4); c.b = slab(px / 16, g4); c.c = slab(px / 16, g4);
    c.u1 = slab(px / 4, g2); c.u2 = slab((size_t)N * (H + 8) * (W + 8), g1p);      // (room for the reflection-padded form the last conv may read)
    c.raw = off; off += al(px * g1p * 4);              // the largest fp32 GEMM result: the first / last level's channels at full resolution
    c.alpha = off; off += al((size_t)N * g4 * 4);
    c.shift = off; off += al((size_t)N * g4 * 4);
    // (mean, M2) per statistics segment: the widest case is 64 channels at full resolution or 256 at 1/16
    // ... or the conv epilogues' partial statistics (3 floats per 16 x 32 tile, consumer wave and channel; the phase-lattice convs count their input grid x 4)
    size_t pf = std::max(norm::part_floats(g1p, (long)H * W), norm::part_floats(g4, (long)H * W / 16));
    pf = std::max(pf, norm::part_floats(g2, (long)H * W / 4));
    pf = std::max(pf, norm::parts_floats(g1p, conv_stats_nper(H, W, 1)));
    pf = std::max(pf, norm::parts_floats(g1p, conv_stats_nper(H / 2, W / 2, 4)));
    pf = std::max(pf, norm::parts_floats(g2, conv_stats_nper(H / 4, W / 4, 4)));
    pf = std::max(pf, norm::parts_floats(g4, conv_stats_nper(H / 4, W / 4, 1)));
    c.part = off; off += al((size_t)N * pf * 4);
    c.total = off;
    (void)r;
    return c;
}

}  // namespace

namespace {
// ---- the fp32 mode: ResnetGenerator.forward (ResNet_arch.py:19-151) on NCHW fp32 tensors with the generic fp32 ops of f32ops.hip; graph = oracle/nets.py resnet_forward ----
struct RCarve32 { size_t raw, a, b, t0, t1, r1, u1, u2, total; };
RCarve32 rcarve32(const innfer_resnet* r, int N, int H, int W) {
    RCarve32 c{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t px = (size_t)N * H * W, g = r->ngf;
    size_t off = 0;
    auto buf = [&](size_t floats) { size_t o = off; off += al(floats * 4); return o; };
    c.raw = buf(px * g); c.a = buf(px * g); c.b = buf(px / 4 * 2 * g); c.t0 = buf(px / 16 * 4 * g); c.t1 = buf(px / 16 * 4 * g); c.r1 = buf(px / 16 * 4 * g);
    c.u1 = buf(px / 4 * 2 * g); c.u2 = buf(px * g);
    c.total = off;
    return c;
}

int resnet_forward_f32(innfer_resnet* r, const float* x, float* y, int N, int H, int W, char* ws, hipStream_t s) {
    const RCarve32 cv = rcarve32(r, N, H, W);
    const int nb = r->n_blocks;
    auto B = [&](size_t o) { return (float*)(ws + o); };
    float *RAW = B(cv.raw), *A = B(cv.a), *Bf = B(cv.b), *T0 = B(cv.t0), *T1 = B(cv.t1), *R1 = B(cv.r1), *U1 = B(cv.u1), *U2 = B(cv.u2);
    // layer li over an h x w input: Conv2d(k, stride, padding k / 2 in `pad_mode`) / ConvTranspose2d(3, 2, 1, output_padding 1) as four phase launches / upconv (nearest 2x + 3x3)
    auto conv = [&](int li, const float* in, int h, int w, int stride, int pad_mode, int act, float* out) -> int {
        const Layer& l = r->layers[li];
        F32Conv c{};
        c.in = in; c.in_nstride = (long)l.cin * h * w; c.in_cstride = (long)h * w; c.C = l.cin; c.Hin = h; c.Win = w;
        c.bias = l.d_b; c.K = l.cout; c.act = act; c.N = N; c.out = out; c.out_pstride = 1;
        if (l.transposed) {
            const int ho = 2 * h, wo = 2 * w;
            c.out_nstride = (long)l.cout * ho * wo; c.out_cstride = (long)ho * wo; c.Wout = wo;
            c.Ho = h; c.Wo = w; c.osy = c.osx = 2; c.isy = c.isx = 1;
            for (int ph = 0; ph < 4; ++ph) {
                int kyv[2], dyv[2], kxv[2], dxv[2];
                const int ny = phase_taps1d(ph >> 1, kyv, dyv), nx = phase_taps1d(ph & 1, kxv, dxv);
                c.ntap = ny * nx;
                for (int t = 0; t < c.ntap; ++t) { c.dy[t] = dyv[t / nx]; c.dx[t] = dxv[t % nx]; }
                c.ooy = ph >> 1; c.oox = ph & 1; c.wp = r->f32_w[li][ph];
                int rc = f32conv_launch(c, s);
                if (rc) return rc;
            }
            return INNFER_OK;
        }
        const int up = l.up2 ? 1 : 0, ho = up ? 2 * h : h / stride, wo = up ? 2 * w : w / stride;
        c.out_nstride = (long)l.cout * ho * wo; c.out_cstride = (long)ho * wo; c.Wout = wo;
        c.Ho = ho; c.Wo = wo; c.osy = c.osx = 1; c.isy = c.isx = stride; c.up = up; c.pad_mode = pad_mode;
        c.ntap = l.k * l.k;
        for (int t = 0; t < c.ntap; ++t) { c.dy[t] = t / l.k - l.k / 2; c.dx[t] = t % l.k - l.k / 2; }
        c.wp = r->f32_w[li][0];
        return f32conv_launch(c, s);
    };
    // the norm layer behind layer li (instance norm, or BatchNorm2d on the image's / the running statistics), activation, optional skip
    auto norm = [&](int li, const float* in, int h, int w, int act, float* out, const float* res = nullptr) -> int {
        const Layer& l = r->layers[li];
        const long hw = (long)h * w;
        const int C = l.cout;
        if (!r->batch_norm) return f32_norm_launch(in, C * hw, hw, out, C * hw, hw, N, C, hw, 2, 1e-5f, nullptr, nullptr, nullptr, nullptr, act, s, res, C * hw, hw);
        if (r->eval_mode) return f32_norm_launch(in, C * hw, hw, out, C * hw, hw, N, C, hw, 3, 1e-5f, l.d_ev_alpha, l.d_ev_shift, nullptr, nullptr, act, s, res, C * hw, hw);
        return f32_norm_launch(in, C * hw, hw, out, C * hw, hw, N, C, hw, 0, 1e-5f, l.d_gamma, l.d_beta, nullptr, nullptr, act, s, res, C * hw, hw);
    };
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    const int bp = r->block_pad;                                     // 1 reflect, 2 replicate, 0 zero: the same codes as F32Conv.pad_mode
    CK(conv(0, x, H, W, 1, 1, 0, RAW)); CK(norm(0, RAW, H, W, 2, A));                  // c7s1: ReflectionPad2d(3), conv, norm, ReLU
    CK(conv(1, A, H, W, 2, 0, 0, RAW)); CK(norm(1, RAW, H2, W2, 2, Bf));               // stride-2 3x3, zero padding
    CK(conv(2, Bf, H2, W2, 2, 0, 0, RAW)); CK(norm(2, RAW, H4, W4, 2, T0));
    float *cur = T0, *alt = T1;
    for (int b = 0; b < nb; ++b) {                                   // ResnetBlock: x + norm(conv(relu(norm(conv(x)))))  (dropout: identity under eval)
        CK(conv(3 + 2 * b, cur, H4, W4, 1, bp, 0, RAW)); CK(norm(3 + 2 * b, RAW, H4, W4, 2, R1));
        CK(conv(4 + 2 * b, R1, H4, W4, 1, bp, 0, RAW)); CK(norm(4 + 2 * b, RAW, H4, W4, 0, alt, cur));
        std::swap(cur, alt);
    }
    const int lu = 3 + 2 * nb;
    CK(conv(lu, cur, H4, W4, 1, 0, 0, RAW)); CK(norm(lu, RAW, H2, W2, 2, U1));
    CK(conv(lu + 1, U1, H2, W2, 1, 0, 0, RAW)); CK(norm(lu + 1, RAW, H, W, 2, U2));
    CK(conv(lu + 2, U2, H, W, 1, 1, 3, y));                          // c7s1-out: ReflectionPad2d(3), conv + bias, tanh
#undef CK
    return INNFER_OK;
}
}  // namespace

// The reference's fp16 switch for this generator (run.py:345,421-422), as innfer_unet_set_precision: fp32 = 1 runs ResnetGenerator.forward in fp32 on NCHW fp32 tensors.
extern "C" int innfer_resnet_set_precision(innfer_resnet* r, int fp32) {
    if (!r || (fp32 != 0 && fp32 != 1)) return set_error(INNFER_ERR_INVALID, "resnet_set_precision: 0 (fp16 arithmetic) or 1 (fp32)");
    r->fp32 = fp32 != 0;
    if (!r->fp32) return INNFER_OK;
    if (!r->uploaded) { int rc = rn_upload(r); if (rc) return rc; r->uploaded = true; }
    if (!r->f32_w.empty()) return INNFER_OK;
    std::vector<float> host;
    auto put = [&](std::vector<float*>& dst, int K, int C, int ntap, const std::function<float(int, int, int)>& w) -> int {
        host.resize(f32conv_packed_floats(K, C, ntap));
        f32conv_pack(K, C, ntap, w, host.data());
        float* d = nullptr;
        INNFER_HIP(hipMalloc((void**)&d, host.size() * sizeof(float)));
        dst.push_back(d);
        INNFER_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
        return INNFER_OK;
    };
    r->f32_w.assign(r->layers.size(), {});
    for (size_t li = 0; li < r->layers.size(); ++li) {
        const Layer& l = r->layers[li];
        const std::vector<float>& w = r->params[l.w].host;
        if (l.transposed) {                                           // [cin][cout][3][3], one panel per output phase
            for (int ph = 0; ph < 4; ++ph) {
                int kyv[2], dyv[2], kxv[2], dxv[2];
                const int ny = phase_taps1d(ph >> 1, kyv, dyv), nx = phase_taps1d(ph & 1, kxv, dxv);
                int rc = put(r->f32_w[li], l.cout, l.cin, ny * nx, [&](int co, int ci, int t) { return w[(((size_t)ci * l.cout + co) * 3 + kyv[t / nx]) * 3 + kxv[t % nx]]; });
                if (rc) return rc;
            }
        } else {
            const int T = l.k * l.k, C = l.cin;
            int rc = put(r->f32_w[li], l.cout, C, T, [&w, C, T](int k, int c, int t) { return w[((size_t)k * C + c) * T + t]; });
            if (rc) return rc;
        }
    }
    return INNFER_OK;
}

extern "C" size_t innfer_resnet_workspace_bytes(innfer_resnet* r, int N, int H, int W) {
    if (!r || N <= 0 || H <= 0 || W <= 0) return 0;
    return r->fp32 ? rcarve32(r, N, H, W).total : rcarve(r, N, H, W).total;
}

extern "C" int innfer_resnet_forward(innfer_resnet* r, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                                     int N, int H, int W, void* d_ws, size_t ws_bytes, void* stream) {
    if (!r || !d_in || !d_out || !d_ws) return set_error(INNFER_ERR_INVALID, "resnet_forward: null argument");
    if (N <= 0 || H < 16 || W < 16 || (H & 3) || (W & 3))
        return set_error(INNFER_ERR_INVALID, "resnet_forward: H and W must be multiples of 4 and at least 16 (two stride-2 stages, reflection pads)");
    if (!r->uploaded) { int rc = rn_upload(r); if (rc) return rc; if (r->fp32) { rc = innfer_resnet_set_precision(r, 1); if (rc) return rc; } }
    if (r->fp32) {
        if (in_dtype != INNFER_F32 || out_dtype != INNFER_F32) return set_error(INNFER_ERR_INVALID, "resnet_forward: the fp32 mode takes and returns fp32 tensors");
        if (r->f32_w.empty()) return set_error(INNFER_ERR_INVALID, "resnet_forward: call innfer_resnet_set_precision(r, 1) after the last innfer_resnet_set_param");
        const RCarve32 c32 = rcarve32(r, N, H, W);
        if (ws_bytes < c32.total) return set_error(INNFER_ERR_WORKSPACE, "resnet_forward: workspace %zu < %zu bytes", ws_bytes, c32.total);
        return resnet_forward_f32(r, (const float*)d_in, (float*)d_out, N, H, W, (char*)d_ws, (hipStream_t)stream);
    }
    const RCarve cv = rcarve(r, N, H, W);
    if (ws_bytes < cv.total) return set_error(INNFER_ERR_WORKSPACE, "resnet_forward: workspace %zu < %zu bytes", ws_bytes, cv.total);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d_ws;
    float* raw = (float*)(ws + cv.raw);
    float* alpha = (float*)(ws + cv.alpha);
    float* shift = (float*)(ws + cv.shift);
    float* part = (float*)(ws + cv.part);
    int dy49[49], dx49[49], dy9[9], dx9[9];
    for (int t = 0; t < 49; ++t) { dy49[t] = t / 7 - 3; dx49[t] = t % 7 - 3; }
    for (int t = 0; t < 9; ++t) { dy9[t] = t / 3 - 1; dx9[t] = t % 3 - 1; }
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    // GEMM of layer l over (Hi, Wi) -> (Ho, Wo) + instance norm + [relu] [+ res] -> dst slab
    // BatchNorm under eval(): no statistics of the image -- (alpha, shift) are the layer's transform of its running statistics
    const bool bn_eval = r->batch_norm && r->eval_mode;
    auto fill_ev = [&](const Layer& l) -> int {
        hipLaunchKernelGGL(rn_fill_ev, dim3((unsigned)((N * l.cout + 255) / 256)), dim3(256), 0, s, (const float*)l.d_ev_alpha, (const float*)l.d_ev_shift,
                           alpha, shift, l.cout, N);
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    };
    auto norm_post = [&](const Layer& l, int Ho, int Wo, int relu, const f16* res, f16* dst) -> int {
        const long HW = (long)Ho * Wo;
        const int cpad = (l.cout + 63) / 64 * 64;
        if (bn_eval) { int rc = fill_ev(l); if (rc) return rc; }
        else { int rc = norm::launch_stats(raw, cpad, HW, 1e-5f, l.d_gamma, l.d_beta, alpha, shift, l.cout, N, part, s); if (rc) return rc; }
        const long total = (long)N * HW * (l.cout / 8);
        hipLaunchKernelGGL(rn_post, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)raw, cpad, l.cout, HW, N,
                           (const float*)alpha, (const float*)shift, relu, res, dst, (long)N * HW * 32);
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    };
    auto conv = [&](const Layer& l, const f16* in, int Hi, int Wi, int Ho, int Wo, int stride, int reflect) -> int {
        const int cin_pad = (l.cin + 31) / 32 * 32, cout_pad = (l.cout + 63) / 64 * 64;
        return gg::launch(l.d_w[0], cin_pad, cout_pad, in, (long)N * Hi * Wi * 32, N, Hi, Wi, raw, Ho, Wo, stride, l.k * l.k,
                          l.k == 7 ? dy49 : dy9, l.k == 7 ? dx49 : dx9, Ho, Wo, 1, 0, 0, 0, s, nullptr, 0, 0, reflect);
    };
    auto deconv = [&](const Layer& l, const f16* in, int Hi, int Wi) -> int {
        const int cin_pad = (l.cin + 31) / 32 * 32, cout_pad = (l.cout + 63) / 64 * 64;
        for (int ph = 0; ph < 4; ++ph) {
            int kyv[2], dyv[2], kxv[2], dxv[2], dy[4], dx[4];
            const int ny = phase_taps1d(ph >> 1, kyv, dyv), nx = phase_taps1d(ph & 1, kxv, dxv);
            for (int t = 0; t < ny * nx; ++t) { dy[t] = dyv[t / nx]; dx[t] = dxv[t % nx]; }
            CK(gg::launch(l.d_w[ph], cin_pad, cout_pad, in, (long)N * Hi * Wi * 32, N, Hi, Wi, raw, Hi, Wi, 1, ny * nx, dy, dx,
                          2 * Hi, 2 * Wi, 2, ph >> 1, ph & 1, 0, s));
        }
        return INNFER_OK;
    };
    // ResnetBlock conv: reflection-padded 3x3 on the halo-tile kernel (fp16 slab out, bias included), statistics and normalisation on
    // that slab -- the conv output is rounded to fp16 before the instance norm, as in the reference's own fp16 mode
    auto block_conv = [&](const Layer& l, const f16* in, int Hc, int Wc, int relu, const f16* res, f16* dst) -> int {
        if (!l.d_w3) {
            if (r->block_pad == 2) return set_error(INNFER_ERR_UNSUPPORTED, "resnet: replication padding needs the halo-tile conv (channels %% 64 == 0)");
            CK(conv(l, in, Hc, Wc, Hc, Wc, 1, r->block_pad));
            return norm_post(l, Hc, Wc, relu, res, dst);
        }
        const long HW = (long)Hc * Wc, G = (long)N * HW * 32;
        f16* Y = (f16*)raw;                                           // the fp32 GEMM buffer is free here
        ConvLaunch L{};
        L.in = in; L.in_gstride = G; L.C = l.cin;
        L.wpk = (const f16*)l.d_w3; L.bias = l.d_b;
        L.out = Y; L.out_gstride = G; L.K = l.cout;
        L.N = N; L.H = Hc; L.W = Wc; L.act = 0; L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = Hc;
        L.out_mode = OUT_SLAB; L.reflect = r->block_pad;
        if (bn_eval) {
            CK(conv_launch(L, s));
            CK(fill_ev(l));
        } else if (l.cout % 64 == 0 && l.cout <= 256) {      // statistics as per-tile partials out of the conv epilogue: the slab is not read again for them
            L.stats_part = part;
            CK(conv_launch(L, s));
            const int pxb = ((HW + 1023) / 1024) * (l.cout / 32) * N >= 256 ? 1024 : 256;
            hipLaunchKernelGGL(rn_post_slab_parts, dim3((unsigned)((HW + pxb - 1) / pxb), l.cout / 32, N), dim3(256), 0, s, (const f16*)Y, l.cout, HW,
                               (const float*)part, conv_stats_nper(Hc, Wc, 1), 1e-5f, (const float*)l.d_gamma, (const float*)l.d_beta, relu, res, dst, G, pxb);
            INNFER_HIP(hipGetLastError());
            return INNFER_OK;
        } else {
        CK(conv_launch(L, s));
        CK(norm::launch_stats_slab(Y, G, HW, 1e-5f, l.d_gamma, l.d_beta, alpha, shift, l.cout, N, part, s));
        }
        const long total = (long)N * HW * (l.cout / 8);
        hipLaunchKernelGGL(rn_post_slab, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const f16*)Y, l.cout, HW, N,
                           (const float*)alpha, (const float*)shift, relu, res, dst, G);
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    };
    f16 *X0 = (f16*)(ws + cv.x0), *S1 = (f16*)(ws + cv.s1), *S2 = (f16*)(ws + cv.s2), *A = (f16*)(ws + cv.a), *B = (f16*)(ws + cv.b),
        *Cc = (f16*)(ws + cv.c), *U1 = (f16*)(ws + cv.u1), *U2 = (f16*)(ws + cv.u2);
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    size_t li = 0;
    const int patch = 7 * r->in_nc <= 32;
    hipLaunchKernelGGL(rn_pre, dim3((unsigned)(((long)N * H * W + 255) / 256)), dim3(256), 0, s, d_in, in_dtype == INNFER_F32, r->in_nc, H, W, N, X0, patch);
    INNFER_HIP(hipGetLastError());
    if (patch && r->layers[li].d_w3 && H >= 4 && (long)H * W * 64 < 0x7fffffffL) {
        // c7s1-64 as a 7 x 1 column conv over the row-patch slab on the halo-tile kernel (rows reflected by its loader), fp16 slab out, norm on the slab
        const Layer& l = r->layers[li];
        const long HW = (long)H * W, G = (long)N * HW * 32;
        f16* Y = (f16*)raw;
        ConvLaunch L{};
        L.in = X0; L.in_gstride = G; L.C = 32;
        L.wpk = (const f16*)l.d_w3; L.bias = l.d_b;
        L.out = Y; L.out_gstride = G; L.K = l.cout;
        L.N = N; L.H = H; L.W = W; L.act = 0; L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = H;
        L.out_mode = OUT_SLAB; L.conv7v = 1; L.reflect = 1;
        if (bn_eval) {
            CK(conv_launch(L, s));
            CK(fill_ev(l));
        } else if (l.cout == 64) {
            L.stats_part = part;
            CK(conv_launch(L, s));
            CK(norm::launch_combine_parts(part, conv_stats_nper(H, W, 1), HW, 1e-5f, l.d_gamma, l.d_beta, alpha, shift, l.cout, N, s));
        } else {
        CK(conv_launch(L, s));
        CK(norm::launch_stats_slab(Y, G, HW, 1e-5f, l.d_gamma, l.d_beta, alpha, shift, l.cout, N, part, s));
        }
        const long total = (long)N * HW * (l.cout / 8);
        hipLaunchKernelGGL(rn_post_slab, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const f16*)Y, l.cout, HW, N,
                           (const float*)alpha, (const float*)shift, 1, (const f16*)nullptr, S1, G);
        INNFER_HIP(hipGetLastError());
        ++li;
    } else {
    if (patch) {
        int dy7[7], dx7[7];
        for (int t = 0; t < 7; ++t) { dy7[t] = t - 3; dx7[t] = 0; }
        CK(gg::launch(r->layers[li].d_w[0], 32, (r->ngf + 63) / 64 * 64, X0, (long)N * H * W * 32, N, H, W, raw, H, W, 1, 7, dy7, dx7, H, W, 1, 0, 0, 0, s, nullptr, 0, 0, 1));
    } else
    CK(conv(r->layers[li], X0, H, W, H, W, 1, 1));
    CK(norm_post(r->layers[li], H, W, 1, nullptr, S1)); ++li;          // c7s1-64
    }
    CK(conv(r->layers[li], S1, H, W, H2, W2, 2, 0)); CK(norm_post(r->layers[li], H2, W2, 1, nullptr, S2)); ++li;     // d128
    CK(conv(r->layers[li], S2, H2, W2, H4, W4, 2, 0)); CK(norm_post(r->layers[li], H4, W4, 1, nullptr, A)); ++li;    // d256
    f16* t = A;
    f16* spare = B;
    for (int b = 0; b < r->n_blocks; ++b) {                                                                             // R256 x n
        CK(block_conv(r->layers[li], t, H4, W4, 1, nullptr, Cc)); ++li;
        CK(block_conv(r->layers[li], Cc, H4, W4, 0, t, spare)); ++li;
        f16* tmp = t; t = spare; spare = tmp;
    }
    // upsample_mode 'upconv': nearest-2x + 3x3 conv (zero padding) on the halo-tile kernel, instance norm + ReLU on its fp16 slab
    // norm + ReLU of an fp16 conv result; pad: into the reflection-padded (Ho + 8) x (Wo + 8) slab the sub-block form of the last conv reads
    auto post_relu = [&](const Layer& l, const f16* Y, int Ho, int Wo, f16* dst, bool pad) -> int {
        const long HW = (long)Ho * Wo, G = (long)N * HW * 32;
        const long total = (long)N * HW * (l.cout / 8);
        if (pad) {
            const long Gp = (long)N * (Ho + 8) * (Wo + 8) * 32;
            const long ring = (long)N * (l.cout / 32) * (2 * (Wo + 8) + 2 * (Ho + 6)) * 4;
            hipLaunchKernelGGL(rn_zero_ring, dim3((unsigned)((ring + 255) / 256)), dim3(256), 0, s, dst, l.cout / 32, Ho, Wo, N, Gp);
            hipLaunchKernelGGL(rn_post_slab_pad, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Y, l.cout, Ho, Wo, N,
                               (const float*)alpha, (const float*)shift, 1, dst, G, Gp);
        } else {
            hipLaunchKernelGGL(rn_post_slab, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Y, l.cout, HW, N,
                               (const float*)alpha, (const float*)shift, 1, (const f16*)nullptr, dst, G);
        }
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    };
    auto up_conv = [&](const Layer& l, const f16* in, int Hi, int Wi, f16* dst, bool pad) -> int {
        const int Ho = 2 * Hi, Wo = 2 * Wi;
        const long HW = (long)Ho * Wo, G = (long)N * HW * 32;
        f16* Y = (f16*)raw;
        ConvLaunch L{};
        L.in = in; L.in_gstride = (long)N * Hi * Wi * 32; L.C = l.cin;
        L.wpk = (const f16*)l.d_w3; L.bias = l.d_b;
        L.out = Y; L.out_gstride = G; L.K = l.cout;
        L.N = N; L.H = Ho; L.W = Wo; L.up = 1; L.act = 0; L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = Ho;
        L.out_mode = OUT_SLAB;
        if (bn_eval) {
            CK(conv_launch(L, s));
            CK(fill_ev(l));
        } else if (l.cout % 64 == 0 && l.cout <= 128) {
            L.stats_part = part;
            CK(conv_launch(L, s));
            CK(norm::launch_combine_parts(part, conv_stats_nper(Ho, Wo, 1), HW, 1e-5f, l.d_gamma, l.d_beta, alpha, shift, l.cout, N, s));
        } else {
        CK(conv_launch(L, s));
        CK(norm::launch_stats_slab(Y, G, HW, 1e-5f, l.d_gamma, l.d_beta, alpha, shift, l.cout, N, part, s));
        }
        return post_relu(l, Y, Ho, Wo, dst, pad);
    };
    // ConvTranspose whose input grid fills the 16 x 32 tiles: four phase convs in one launch of the halo-tile kernel, fp16 slab out, norm on the slab
    auto deconv_tile = [&](const Layer& l, const f16* in, int Hi, int Wi, f16* dst, bool pad) -> int {
        const int Ho = 2 * Hi, Wo = 2 * Wi;
        const long HW = (long)Ho * Wo, G = (long)N * HW * 32;
        f16* Y = (f16*)raw;
        ConvLaunch L{};
        L.in = in; L.in_gstride = (long)N * Hi * Wi * 32; L.C = l.cin;
        L.wpk = (const f16*)l.d_w3; L.bias = l.d_b4;
        L.out = Y; L.out_gstride = G; L.K = 4 * l.cout; L.phase_c = l.cout; L.deconv_phases = 1;
        L.N = N; L.H = Hi; L.W = Wi; L.act = 0; L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = Hi;
        L.out_mode = OUT_SLAB;
        if (bn_eval) {
            CK(conv_launch(L, s));
            CK(fill_ev(l));
        } else {
        L.stats_part = part;                      // (cout % 64 == 0 is this path's condition; 64 / 128 channels)
        CK(conv_launch(L, s));
        CK(norm::launch_combine_parts(part, conv_stats_nper(Hi, Wi, 4), HW, 1e-5f, l.d_gamma, l.d_beta, alpha, shift, l.cout, N, s));
        }
        return post_relu(l, Y, Ho, Wo, dst, pad);
    };
    auto fills_tiles = [](int h, int w) { return (long)h * w * 10 >= (long)((h + 15) / 16 * 16) * (w <= 16 ? 16 : (w + 31) / 32 * 32) * 5; };
    // the last conv's sub-block form needs its input reflection-padded (written so by the slab post pass) and Q in the fp32 buffer
    const Layer& last = r->layers.back();
    bool pad_last = last.d_w27 && H >= 8 && W >= 8 && (size_t)N * 9 * last.cout * (H + 8) * (W + 8) <= (size_t)N * H * W * 64 &&
                    (long)(H + 8) * (W + 8) * 64 < 0x7fffffffL;
    if (r->layers[li].up2) {
        CK(up_conv(r->layers[li], t, H4, W4, U1, false)); ++li;                                                         // u128
        CK(up_conv(r->layers[li], U1, H2, W2, U2, pad_last)); ++li;                                                     // u64
    } else if (r->layers[li].transposed && r->layers[li].d_b4 && r->layers[li + 1].d_b4 && fills_tiles(H4, W4)) {
        CK(deconv_tile(r->layers[li], t, H4, W4, U1, false)); ++li;                                                     // u128
        CK(deconv_tile(r->layers[li], U1, H2, W2, U2, pad_last)); ++li;                                                 // u64
    } else {
    pad_last = false;
    CK(deconv(r->layers[li], t, H4, W4)); CK(norm_post(r->layers[li], H2, W2, 1, nullptr, U1)); ++li;                 // u128
    CK(deconv(r->layers[li], U1, H2, W2)); CK(norm_post(r->layers[li], H, W, 1, nullptr, U2)); ++li;                  // u64
    }
    {   // c7s1-out + tanh
        const Layer& l = r->layers[li];
        if (pad_last) {     // 3x3 conv with nine sub-block channels per output over the padded grid (fp32, planar), then the 9-term gather + bias + tanh
            const int Hp = H + 8, Wp = W + 8;
            ConvLaunch L{};
            L.in = U2; L.in_gstride = (long)N * Hp * Wp * 32; L.C = l.cin;
            L.wpk = (const f16*)l.d_w27; L.bias = l.d_z32;
            L.out = raw; L.K = 9 * l.cout; L.N = N; L.H = Hp; L.W = Wp; L.act = 0;
            L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = Hp;
            L.out_mode = OUT_NCHW; L.out_f32 = 1;
            CK(conv_launch(L, s));
            const long total = (long)N * l.cout * H * W;
            hipLaunchKernelGGL(rn_sum9_tanh, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)raw, (const float*)l.d_b, d_out,
                               out_dtype == INNFER_F32, l.cout, H, W, N);
            INNFER_HIP(hipGetLastError());
            return INNFER_OK;
        }
        if (l.d_w3) {       // tanh(conv7x7(reflect3(x)) + bias) -> NCHW in the conv's planar epilogue
            ConvLaunch L{};
            L.in = U2; L.in_gstride = (long)N * H * W * 32; L.C = l.cin;
            L.wpk = (const f16*)l.d_w3; L.bias = l.d_b;
            L.out = d_out; L.K = l.cout; L.N = N; L.H = H; L.W = W; L.act = 3;
            L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = H;
            L.out_mode = OUT_NCHW; L.out_f32 = out_dtype == INNFER_F32; L.conv7 = 1; L.reflect = 1;
            CK(conv_launch(L, s));
            return INNFER_OK;
        }
        const int rs = (l.cout + 3) / 4 * 4;
        CK(gg::launch(l.d_w[0], (l.cin + 31) / 32 * 32, 64, U2, (long)N * H * W * 32, N, H, W, raw, H, W, 1, 49, dy49, dx49, H, W, 1, 0, 0, 0, s, nullptr, 0, rs, 1));
        hipLaunchKernelGGL(rn_final, dim3((unsigned)(((long)N * H * W + 255) / 256)), dim3(256), 0, s, (const float*)raw, rs, l.cout, (long)H * W, N,
                           (const float*)l.d_b, d_out, out_dtype == INNFER_F32);
        INNFER_HIP(hipGetLastError());
    }
#undef CK
    return INNFER_OK;
}
