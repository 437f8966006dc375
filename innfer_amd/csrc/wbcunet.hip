// White-box-Cartoonization UNet + guided filter on gfx950 -- SURVEY.md section 8f row n4 (second half).
// Replaces UnetGeneratorWBC(mode='pt' | 'tf').forward with ResBlock (architectures/WBCNet_arch.py:8-99; `-a wbcunet`,
// utils/defaults.py:90-97: nf 32) and guided_filter(x, y, r=1, eps) (utils/utils.py:548-626, run.py:427-429).
//
//   stride-1 3x3 convs, zero padding                      the SR path's halo-tile kernel (conv3x3.hip), epilogue bias / LeakyReLU / skip
//   first 7x7 conv (3 input channels)                      gg::gemm_gather over a row-patch slab (wb_pre): 7 vertical taps
//   last 7x7 conv (32 -> 3)                                nine displaced 3x3 convs on the halo-tile kernel (conv3x3_pc<..,S9>), planar output
//   stride-2 3x3 convs                                     gg::gemm_gather (9 taps, zero fill) + wb_post (bias, LeakyReLU)
//   bilinear 2x (align_corners=False) + skip addition      wb_upadd (ATen's source index / lambda arithmetic)
//   guided filter, 3x3 box means, reflect padding          gf_ab (means, covariance, A, b) + gf_out (mean_A * x + mean_b)
#include "common.h"
#include "gather_gemm.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace innfer;

namespace {

// raw fp32 + bias -> [LeakyReLU] -> [+ residual] -> fp16 slab (or NCHW when nchw != nullptr); one thread per (pixel, 4 channels)
__global__ void wb_post(const float* raw, int rs, int C, long npix, const float* bias, int act, const f16* res, f16* dst, long g,
                        void* nchw, int nchw_f32, long HW) {
    const int c4 = (C + 3) / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * c4) return;
    const long pix = i / c4;
    const int c = (int)(i - pix * c4) * 4;
    const f32x4 v = *(const f32x4*)(raw + pix * rs + c);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        y[e] = c + e < C ? v[e] + bias[c + e] : 0.f;
        if (act) y[e] = fmaxf(y[e], 0.2f * y[e]);
    }
    if (nchw) {
        const long n = pix / HW, px = pix % HW;
        for (int e = 0; e < 4 && c + e < C; ++e) {
            const long o = (n * C + c + e) * HW + px;
            if (nchw_f32) ((float*)nchw)[o] = y[e]; else ((f16*)nchw)[o] = (f16)y[e];
        }
        return;
    }
    const long o = (c >> 5) * g + pix * 32 + (c & 31);
    f16x4 h;
    if (res) {
        const f16x4 r = *(const f16x4*)(res + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] += (float)r[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (f16)y[e];
    *(f16x4*)(dst + o) = h;
}

// dst[2H x 2W] = bilinear2x(src[H x W], align_corners=False) + skip; slabs of C channels; thread per (out pixel, 8 channels)
__global__ void wb_upadd(const f16* src, long sg, const f16* skip, f16* dst, long dg, int C, int N, int H, int W, int tf) {
    const int c8 = C / 8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ho = 2 * H, Wo = 2 * W;
    if (i >= (long)N * Ho * Wo * c8) return;
    const int c = (int)(i % c8) * 8;
    const long opix = i / c8;
    const int X = (int)(opix % Wo), Y = (int)((opix / Wo) % Ho);
    const long n = opix / ((long)Wo * Ho);
    // ATen area_pixel_compute_source_index (align_corners=False): src = max(0, (dst + 0.5) * 0.5 - 0.5)
    const float fy = fmaxf(((float)Y + 0.5f) * 0.5f - 0.5f, 0.f), fx = fmaxf(((float)X + 0.5f) * 0.5f - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const long so = (c >> 5) * sg + (c & 31);
    auto at = [&](int yy, int xx) { return *(const f16x8*)(src + so + ((n * H + yy) * (long)W + xx) * 32); };
    const long o = (c >> 5) * dg + opix * 32 + (c & 31);
    const f16x8 sk = *(const f16x8*)(skip + o);
    f16x8 h;
    if (tf) {
        // tf_2xupsample_bilinear (WBCNet_arch.py:126-137): even/even copies, the others average x[i,j] with its
        // lower / right / lower-right (diagonal) neighbour, replicated at the border
        const int i = Y >> 1, j = X >> 1;
        const int i2 = (Y & 1) ? min(i + 1, H - 1) : i, j2 = (X & 1) ? min(j + 1, W - 1) : j;
        const f16x8 a = at(i, j), b = at(i2, j2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float up = ((Y | X) & 1) ? ((float)a[e] + (float)b[e]) / 2.f : (float)a[e];
            h[e] = (f16)((float)(f16)up + (float)sk[e]);
        }
        *(f16x8*)(dst + o) = h;
        return;
    }
    const f16x8 v00 = at(y0, x0), v01 = at(y0, x1), v10 = at(y1, x0), v11 = at(y1, x1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float up = hy * (hx * (float)v00[e] + lx * (float)v01[e]) + ly * (hx * (float)v10[e] + lx * (float)v11[e]);
        h[e] = (f16)((float)(f16)up + (float)sk[e]);          // the reference rounds the upsampled tensor to the storage type before the add
    }
    *(f16x8*)(dst + o) = h;
}

// NCHW input -> "row patch" slab for the first 7x7 conv: channel kx*C + c of pixel (y, x) holds in[c][y][x + kx - 3] (zero outside the
// image and beyond 7*C <= 32 channels), so the 49-tap conv becomes 7 vertical taps over ONE 32-channel group: 7x less gather traffic than
// 49 taps over a group holding C values.  One thread per pixel, 16-byte stores.
__global__ void wb_pre(const void* in, int in_f32, int C, int H, int W, int N, f16* slab) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long HW = (long)H * W;
    if (i >= (long)N * HW) return;
    const long n = i / HW, px = i % HW;
    const int x = (int)(px % W);
    f16 v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int kx = j / C, c = j - kx * C, X = x + kx - 3;
        f16 t = (f16)0.f;
        if (kx < 7 && X >= 0 && X < W) {
            const long o = (n * C + c) * HW + px + (kx - 3);
            t = in_f32 ? (f16)((const float*)in)[o] : ((const f16*)in)[o];
        }
        v[j] = t;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f16x8*)(slab + i * 32 + 8 * q) = *(const f16x8*)(v + 8 * q);
}

// ---- guided filter (r = 1) on NCHW planes --------------------------------------------------------
__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ float ld(const void* p, long o, int f32) { return f32 ? ((const float*)p)[o] : (float)((const f16*)p)[o]; }

// A = cov_xy / (var_x + eps), b = mean_y - A * mean_x  with 3x3 reflect box means (filter2D(., ones/9) / N, N = filter2D(ones))
__global__ void gf_ab(const void* x, const void* y, int f32, long planes, int H, int W, float eps, float* A, float* B) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const long base = i - (long)Y * W - X;
    const float k = 1.0f / 9.0f;
    float sx = 0.f, sy = 0.f, sxy = 0.f, sxx = 0.f, sn = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const long o = base + (long)refl(Y + dy, H) * W + refl(X + dx, W);
            const float xv = ld(x, o, f32), yv = ld(y, o, f32);
            sx += xv * k; sy += yv * k; sxy += (xv * yv) * k; sxx += (xv * xv) * k; sn += k;
        }
    const float mx = sx / sn, my = sy / sn;
    const float cov = sxy / sn - mx * my, var = sxx / sn - mx * mx;
    const float a = cov / (var + eps);
    A[i] = a;
    B[i] = my - a * mx;
}

__global__ void gf_out(const float* A, const float* B, const void* x, int f32, long planes, int H, int W, void* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const long base = i - (long)Y * W - X;
    const float k = 1.0f / 9.0f;
    float sa = 0.f, sb = 0.f, sn = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const long o = base + (long)refl(Y + dy, H) * W + refl(X + dx, W);
            sa += A[o] * k; sb += B[o] * k; sn += k;
        }
    const float v = (sa / sn) * ld(x, i, f32) + sb / sn;
    if (f32) ((float*)out)[i] = v; else ((f16*)out)[i] = (f16)v;
}

// The same with a (2r+1)^2 window (ks = 2r + 1 of guided_filter, utils.py:584-590; box value float32(1 / ks^2), get_box_kernel :536-545)
__global__ void gf_ab_r(const void* x, const void* y, int f32, long planes, int H, int W, int r, float k, float eps, float* A, float* B) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const long base = i - (long)Y * W - X;
    float sx = 0.f, sy = 0.f, sxy = 0.f, sxx = 0.f, sn = 0.f;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const long o = base + (long)refl(Y + dy, H) * W + refl(X + dx, W);
            const float xv = ld(x, o, f32), yv = ld(y, o, f32);
            sx += xv * k; sy += yv * k; sxy += (xv * yv) * k; sxx += (xv * xv) * k; sn += k;
        }
    const float mx = sx / sn, my = sy / sn;
    const float cov = sxy / sn - mx * my, var = sxx / sn - mx * mx;
    const float a = cov / (var + eps);
    A[i] = a;
    B[i] = my - a * mx;
}

__global__ void gf_out_r(const float* A, const float* B, const void* x, int f32, long planes, int H, int W, int r, float k, void* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const long base = i - (long)Y * W - X;
    float sa = 0.f, sb = 0.f, sn = 0.f;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const long o = base + (long)refl(Y + dy, H) * W + refl(X + dx, W);
            sa += A[o] * k; sb += B[o] * k; sn += k;
        }
    const float v = (sa / sn) * ld(x, i, f32) + sb / sn;
    if (f32) ((float*)out)[i] = v; else ((f16*)out)[i] = (f16)v;
}

// 'fast' mode (utils.py:611-619): A and b of the low-resolution pair are enlarged to x_HR's size (bilinear, align_corners=True) and applied there
__global__ void gf_out_fast(const float* A, const float* B, const void* xhr, int f32, long planes, int H, int W, int Hh, int Wh, void* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Hh * Wh) return;
    const int X = (int)(i % Wh), Y = (int)((i / Wh) % Hh);
    const long pl = i / ((long)Hh * Wh);
    const float sy = Hh > 1 ? (float)(H - 1) / (float)(Hh - 1) : 0.f, sx = Wh > 1 ? (float)(W - 1) / (float)(Wh - 1) : 0.f;
    const float fy = sy * (float)Y, fx = sx * (float)X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* a = A + pl * H * W;
    const float* b = B + pl * H * W;
    const float ma = hy * (hx * a[(long)y0 * W + x0] + lx * a[(long)y0 * W + x1]) + ly * (hx * a[(long)y1 * W + x0] + lx * a[(long)y1 * W + x1]);
    const float mb = hy * (hx * b[(long)y0 * W + x0] + lx * b[(long)y0 * W + x1]) + ly * (hx * b[(long)y1 * W + x0] + lx * b[(long)y1 * W + x1]);
    const float v = ma * ld(xhr, i, f32) + mb;
    if (f32) ((float*)out)[i] = v; else ((f16*)out)[i] = (f16)v;
}

struct Param { std::string key; std::vector<int> shape; std::vector<float> host; bool set = false; };
struct Layer { int w = -1, b = -1, cin = 0, cout = 0, k = 3; f16* d_w = nullptr; float* d_b = nullptr;
               void* d_w3 = nullptr; float* d_b3 = nullptr;
               void* d_ws2 = nullptr; };    // the two stride-2 3x3 convs: panels of the stride-2 gather loader (conv_pack_s2k4 over the kernel embedded in 4x4)     // 3x3 layers: conv3x3.hip panels + bias (the stride-1 ones run there)

}  // namespace

struct innfer_wbc {
    int nf = 32; int tf = 0;
    std::vector<Param> params;
    std::vector<Layer> layers;   // conv, conv_1..conv_4, block_0..3 (conv1, conv2), conv_5..conv_9
    bool fp32 = false;           // innfer_wbc_set_precision(1): the fp32 forward on NCHW fp32 tensors (f32ops.hip), the reference's -no_fp16 mode
    std::vector<float*> f32_w;   //   one f32conv panel per layer
    bool uploaded = false;
};

static void wb_add(innfer_wbc* u, const std::string& key, int cin, int cout, int k) {
    Layer l; l.cin = cin; l.cout = cout; l.k = k;
    Param pw; pw.key = key + ".weight"; pw.shape = {cout, cin, k, k}; u->params.push_back(pw); l.w = (int)u->params.size() - 1;
    Param pb; pb.key = key + ".bias"; pb.shape = {cout}; u->params.push_back(pb); l.b = (int)u->params.size() - 1;
    u->layers.push_back(l);
}

extern "C" int innfer_wbc_create(innfer_wbc** out, int nf, int tf_mode) {
    if (!out) return set_error(INNFER_ERR_INVALID, "wbc_create: null out");
    if (nf != 32) return set_error(INNFER_ERR_UNSUPPORTED, "wbc_create: nf=%d (built: 32)", nf);
    innfer_wbc* u = new innfer_wbc();
    u->nf = nf; u->tf = tf_mode ? 1 : 0;
    wb_add(u, "conv", 3, nf, 7);
    wb_add(u, "conv_1", nf, nf, 3); wb_add(u, "conv_2", nf, 2 * nf, 3);
    wb_add(u, "conv_3", 2 * nf, 2 * nf, 3); wb_add(u, "conv_4", 2 * nf, 4 * nf, 3);
    for (int b = 0; b < 4; ++b) {
        wb_add(u, "block_" + std::to_string(b) + ".conv1", 4 * nf, 4 * nf, 3);
        wb_add(u, "block_" + std::to_string(b) + ".conv2", 4 * nf, 4 * nf, 3);
    }
    wb_add(u, "conv_5", 4 * nf, 2 * nf, 3); wb_add(u, "conv_6", 2 * nf, 2 * nf, 3);
    wb_add(u, "conv_7", 2 * nf, nf, 3); wb_add(u, "conv_8", nf, nf, 3); wb_add(u, "conv_9", nf, 3, 7);
    *out = u;
    return INNFER_OK;
}

static void wb_free(innfer_wbc* u) {
    for (auto& l : u->layers) {
        if (l.d_w) (void)hipFree(l.d_w); if (l.d_b) (void)hipFree(l.d_b); if (l.d_w3) (void)hipFree(l.d_w3); if (l.d_b3) (void)hipFree(l.d_b3);
        if (l.d_ws2) (void)hipFree(l.d_ws2);
        l.d_w = nullptr; l.d_b = nullptr; l.d_w3 = nullptr; l.d_b3 = nullptr; l.d_ws2 = nullptr;
    }
}

extern "C" void innfer_wbc_destroy(innfer_wbc* u) { if (u) { wb_free(u); for (auto v : u->f32_w) if (v) (void)hipFree(v); delete u; } }
extern "C" int innfer_wbc_num_params(innfer_wbc* u) { return u ? (int)u->params.size() : INNFER_ERR_INVALID; }

extern "C" int innfer_wbc_param_info(innfer_wbc* u, int idx, char* key, size_t key_cap, int* ndim, int* shape4) {
    if (!u || idx < 0 || idx >= (int)u->params.size()) return set_error(INNFER_ERR_INVALID, "wbc_param_info: bad index");
    const Param& q = u->params[idx];
    if (key && key_cap) { strncpy(key, q.key.c_str(), key_cap - 1); key[key_cap - 1] = 0; }
    if (ndim) *ndim = (int)q.shape.size();
    if (shape4) for (size_t i = 0; i < 4; ++i) shape4[i] = i < q.shape.size() ? q.shape[i] : 1;
    return INNFER_OK;
}

extern "C" int innfer_wbc_set_param(innfer_wbc* u, int idx, const float* h_data) {
    if (!u || idx < 0 || idx >= (int)u->params.size() || !h_data) return set_error(INNFER_ERR_INVALID, "wbc_set_param: bad arguments");
    Param& q = u->params[idx];
    size_t n = 1;
    for (int s : q.shape) n *= (size_t)s;
    q.host.assign(h_data, h_data + n);
    q.set = true;
    u->uploaded = false;
    return INNFER_OK;
}

namespace {

int wb_upload(innfer_wbc* u) {
    for (auto& q : u->params) if (!q.set) return set_error(INNFER_ERR_INVALID, "wbc: parameter '%s' was never set", q.key.c_str());
    wb_free(u);
    for (auto v : u->f32_w) if (v) (void)hipFree(v);               // the fp32 panels follow the parameters: rebuilt by innfer_wbc_set_precision
    u->f32_w.clear();
    std::vector<f16> panel;
    for (auto& l : u->layers) {
        const std::vector<float>& w = u->params[l.w].host;
        const int cin_pad = (l.cin + 31) / 32 * 32, kk = l.k * l.k;
        if (l.k == 7 && 7 * l.cin <= 32)      // first conv on the row-patch slab (wb_pre): tap t = ky, channel kx*cin + c
            gg::pack_panels(panel, l.cout, 7 * l.cin, 32, 7, [&](int co, int j, int ky) {
                const int kx = j / l.cin, c = j - kx * l.cin;
                return w[((size_t)co * l.cin + c) * 49 + ky * 7 + kx]; });
        else
        gg::pack_panels(panel, l.cout, l.cin, cin_pad, kk, [&](int co, int ci, int t) { return w[((size_t)co * l.cin + ci) * kk + t]; });
        INNFER_HIP(hipMalloc((void**)&l.d_w, panel.size() * sizeof(f16)));
        INNFER_HIP(hipMemcpy(l.d_w, panel.data(), panel.size() * sizeof(f16), hipMemcpyHostToDevice));
        INNFER_HIP(hipMalloc((void**)&l.d_b, l.cout * sizeof(float)));
        INNFER_HIP(hipMemcpy(l.d_b, u->params[l.b].host.data(), l.cout * sizeof(float), hipMemcpyHostToDevice));
        if (l.k == 7 && 7 * l.cin <= 32 && l.cout % 32 == 0) {
            // first conv over the row-patch slab as a 7 x 1 column conv on the halo-tile kernel (three vertically displaced 3-tap blocks, conv_pack7v):
            // bias + LeakyReLU + the fp16 slab come out of its epilogue (gather GEMM -> 64-wide fp32 rows -> wb_post took 282 us at 1080p)
            std::vector<float> wv((size_t)l.cout * 32 * 7, 0.f);
            for (int co = 0; co < l.cout; ++co)
                for (int kx = 0; kx < 7; ++kx)
                    for (int c = 0; c < l.cin; ++c)
                        for (int ky = 0; ky < 7; ++ky) wv[((size_t)co * 32 + kx * l.cin + c) * 7 + ky] = w[((size_t)co * l.cin + c) * 49 + ky * 7 + kx];
            std::vector<char> packed(conv_packed_bytes7v(l.cout, 32));
            conv_pack7v(wv.data(), l.cout, 32, packed.data());
            std::vector<float> b3((size_t)(l.cout + 63) / 64 * 64, 0.f);
            for (int c = 0; c < l.cout; ++c) b3[c] = u->params[l.b].host[c];
            INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
            INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
            INNFER_HIP(hipMalloc((void**)&l.d_b3, b3.size() * sizeof(float)));
            INNFER_HIP(hipMemcpy(l.d_b3, b3.data(), b3.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (l.k == 7 && l.cin % 32 == 0 && l.cout <= 16) {          // last conv (32 -> 3): nine displaced 3x3 convs on the halo-tile kernel
            std::vector<char> packed(conv_packed_bytes7x7(l.cout, l.cin));
            conv_pack7x7(w.data(), l.cout, l.cin, packed.data());
            std::vector<float> b3(64, 0.f);
            for (int c = 0; c < l.cout; ++c) b3[c] = u->params[l.b].host[c];
            INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
            INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
            INNFER_HIP(hipMalloc((void**)&l.d_b3, b3.size() * sizeof(float)));
            INNFER_HIP(hipMemcpy(l.d_b3, b3.data(), b3.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        const size_t li_ = (size_t)(&l - &u->layers[0]);
        if ((li_ == 1 || li_ == 3) && l.k == 3 && l.cin % 32 == 0 && l.cout % 32 == 0) {
            // conv_1 / conv_3 (stride 2): Conv2d(3, 2, padding 1) reads rows 2y - 1 + ky -- the 4x4 / stride 2 / padding 1 lattice of the stride-2 gather
            // loader with a zero fourth tap; tf_same_padding (pad (0, 1, 0, 1), rows 2y + ky) is the same lattice one tap later
            const int o = u->tf ? 1 : 0;
            std::vector<float> w4((size_t)l.cout * l.cin * 16, 0.f);
            for (int co = 0; co < l.cout; ++co)
                for (int ci = 0; ci < l.cin; ++ci)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx)
                            w4[(((size_t)co * l.cin + ci) * 4 + ky + o) * 4 + kx + o] = w[((size_t)co * l.cin + ci) * 9 + ky * 3 + kx];
            std::vector<char> packed(conv_packed_bytes_s2k4(l.cout, l.cin));
            conv_pack_s2k4(w4.data(), l.cout, l.cin, packed.data());
            INNFER_HIP(hipMalloc(&l.d_ws2, packed.size()));
            INNFER_HIP(hipMemcpy(l.d_ws2, packed.data(), packed.size(), hipMemcpyHostToDevice));
        }
        if (l.k == 3 && l.cin % 32 == 0 && l.cout % 32 == 0) {       // halo-tile form for the stride-1 launches
            std::vector<char> packed(conv_packed_bytes(l.cout, l.cin));
            conv_pack(w.data(), l.cout, l.cin, packed.data());
            std::vector<float> b3((size_t)(l.cout + 63) / 64 * 64, 0.f);
            for (int c = 0; c < l.cout; ++c) b3[c] = u->params[l.b].host[c];
            INNFER_HIP(hipMalloc(&l.d_w3, packed.size()));
            INNFER_HIP(hipMemcpy(l.d_w3, packed.data(), packed.size(), hipMemcpyHostToDevice));
            INNFER_HIP(hipMalloc((void**)&l.d_b3, b3.size() * sizeof(float)));
            INNFER_HIP(hipMemcpy(l.d_b3, b3.data(), b3.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    u->uploaded = true;
    return INNFER_OK;
}

struct WCarve { size_t xin, x0, t1, x1, t2, a, b, c, u1, v1, u0, raw, total; };

WCarve wcarve(int N, int H, int W) {
    WCarve c{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t px = (size_t)N * H * W;
    size_t off = 0;
    auto slab = [&](size_t pixels, int ch) { size_t o = off; off += al(pixels * ch * 2); return o; };
    c.xin = slab(px, 32); c.x0 = slab(px, 32);
    c.t1 = slab(px / 4, 32); c.x1 = slab(px / 4, 64);
    c.t2 = slab(px / 16, 64); c.a = slab(px / 16, 128); c.b = slab(px / 16, 128); c.c = slab(px / 16, 128);
    c.u1 = slab(px / 4, 64); c.v1 = slab(px / 4, 64);
    c.u0 = slab(px, 32);
    c.raw = off; off += al(px * 64 * 4);
    c.total = off;
    return c;
}

}  // namespace

namespace {
// ---- the fp32 mode: UnetGeneratorWBC.forward (WBCNet_arch.py:22-99) on NCHW fp32 tensors with the generic fp32 ops of f32ops.hip; graph = oracle/nets.py wbcunet_forward ----
struct WCarve32 { size_t x0, t0, x1, t1, x2, r, u1, t2, u0, t3, total; };
WCarve32 wcarve32(const innfer_wbc* u, int N, int H, int W) {
    WCarve32 c{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t px = (size_t)N * H * W, nf = u->nf;
    size_t off = 0;
    auto buf = [&](size_t floats) { size_t o = off; off += al(floats * 4); return o; };
    c.x0 = buf(px * nf); c.t0 = buf(px / 4 * nf); c.x1 = buf(px / 4 * 2 * nf); c.t1 = buf(px / 16 * 2 * nf); c.x2 = buf(px / 16 * 4 * nf); c.r = buf(px / 16 * 4 * nf);
    c.u1 = buf(px / 4 * 2 * nf); c.t2 = buf(px / 4 * 2 * nf); c.u0 = buf(px * nf); c.t3 = buf(px * nf);
    c.total = off;
    return c;
}

int wbc_forward_f32(innfer_wbc* u, const float* x, float* y, int N, int H, int W, char* ws, hipStream_t s) {
    const WCarve32 cv = wcarve32(u, N, H, W);
    const int nf = u->nf;
    auto B = [&](size_t o) { return (float*)(ws + o); };
    // layer li: k x k conv (zero padding k / 2; stride 2: `pt` pads 1 on every side, `tf` pads (0, 1) = tf_same_padding) over the whole input tensor
    auto conv = [&](int li, const float* in, int h, int w, int stride, int act, float* out, const float* res = nullptr) -> int {
        const Layer& l = u->layers[li];
        F32Conv c{};
        const int ho = h / stride, wo = w / stride;
        c.in = in; c.in_nstride = (long)l.cin * h * w; c.in_cstride = (long)h * w; c.C = l.cin; c.Hin = h; c.Win = w;
        c.wp = u->f32_w[li]; c.bias = l.d_b; c.K = l.cout;
        c.out = out; c.out_nstride = (long)l.cout * ho * wo; c.out_cstride = (long)ho * wo; c.out_pstride = 1; c.Wout = wo;
        c.Ho = ho; c.Wo = wo; c.osy = c.osx = 1; c.isy = c.isx = stride;
        c.ntap = l.k * l.k;
        const int p0 = stride == 2 && u->tf ? 0 : l.k / 2;
        for (int t = 0; t < c.ntap; ++t) { c.dy[t] = t / l.k - p0; c.dx[t] = t % l.k - p0; }
        c.act = act; c.N = N;
        c.res = res; c.res_nstride = c.out_nstride; c.res_cstride = c.out_cstride;
        return f32conv_launch(c, s);
    };
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    float *X0 = B(cv.x0), *T0 = B(cv.t0), *X1 = B(cv.x1), *T1 = B(cv.t1), *X2 = B(cv.x2), *R = B(cv.r), *U1 = B(cv.u1), *T2 = B(cv.t2), *U0 = B(cv.u0), *T3 = B(cv.t3);
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    CK(conv(0, x, H, W, 1, 1, X0));                                  // x0 = lrelu(conv 7x7)
    CK(conv(1, X0, H, W, 2, 1, T0));                                 // lrelu(conv_1, stride 2)
    CK(conv(2, T0, H2, W2, 1, 1, X1));                               // x1 = lrelu(conv_2)
    CK(conv(3, X1, H2, W2, 2, 1, T1));
    CK(conv(4, T1, H4, W4, 1, 1, X2));                               // x2 = lrelu(conv_4)
    float *cur = X2, *alt = B(cv.u1);                                // the blocks ping-pong between X2 and U1 (2 nf at H/2 >= 4 nf at H/4; free until the first up-add)
    for (int b = 0; b < 4; ++b) {
        CK(conv(5 + 2 * b, cur, H4, W4, 1, 1, R));                   // lrelu(conv1)
        CK(conv(6 + 2 * b, R, H4, W4, 1, 0, alt, cur));              // conv2 + x
        std::swap(cur, alt);
    }
    CK(conv(13, cur, H4, W4, 1, 1, T1));                             // lrelu(conv_5): 2 nf at H/4
    CK(f32_upadd_launch(T1, X1, U1, (long)N * 2 * nf, H4, W4, u->tf, s));          // up(x2) + x1
    CK(conv(14, U1, H2, W2, 1, 1, T2));                              // lrelu(conv_6)
    CK(conv(15, T2, H2, W2, 1, 1, T0));                              // x3 = lrelu(conv_7): nf at H/2
    CK(f32_upadd_launch(T0, X0, U0, (long)N * nf, H2, W2, u->tf, s));              // up(x3) + x0
    CK(conv(16, U0, H, W, 1, 1, T3));                                // x4 = lrelu(conv_8)
    CK(conv(17, T3, H, W, 1, 0, y));                                 // conv_9 (7x7)
#undef CK
    return INNFER_OK;
}
}  // namespace

// The reference's fp16 switch for this generator (run.py:345,421-422), as innfer_unet_set_precision: fp32 = 1 runs UnetGeneratorWBC.forward in fp32 on NCHW fp32 tensors.
extern "C" int innfer_wbc_set_precision(innfer_wbc* u, int fp32) {
    if (!u || (fp32 != 0 && fp32 != 1)) return set_error(INNFER_ERR_INVALID, "wbc_set_precision: 0 (fp16 arithmetic) or 1 (fp32)");
    u->fp32 = fp32 != 0;
    if (!u->fp32) return INNFER_OK;
    if (!u->uploaded) { int rc = wb_upload(u); if (rc) return rc; u->uploaded = true; }
    if (!u->f32_w.empty()) return INNFER_OK;
    std::vector<float> host;
    for (auto& l : u->layers) {
        const std::vector<float>& w = u->params[l.w].host;
        const int T = l.k * l.k, C = l.cin;
        host.resize(f32conv_packed_floats(l.cout, C, T));
        f32conv_pack(l.cout, C, T, [&w, C, T](int k, int c, int t) { return w[((size_t)k * C + c) * T + t]; }, host.data());
        float* d = nullptr;
        INNFER_HIP(hipMalloc((void**)&d, host.size() * sizeof(float)));
        u->f32_w.push_back(d);
        INNFER_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return INNFER_OK;
}

extern "C" size_t innfer_wbc_workspace_bytes(innfer_wbc* u, int N, int H, int W) {
    if (!u || N <= 0 || H <= 0 || W <= 0) return 0;
    return u->fp32 ? wcarve32(u, N, H, W).total : wcarve(N, H, W).total;
}

extern "C" int innfer_wbc_forward(innfer_wbc* u, const void* d_in, int in_dtype, void* d_out, int out_dtype,
                                  int N, int H, int W, void* d_ws, size_t ws_bytes, void* stream) {
    if (!u || !d_in || !d_out || !d_ws) return set_error(INNFER_ERR_INVALID, "wbc_forward: null argument");
    if (N <= 0 || H < 4 || W < 4 || (H & 3) || (W & 3)) return set_error(INNFER_ERR_INVALID, "wbc_forward: H and W must be multiples of 4 (run.py applies modcrop(img, 4))");
    if (!u->uploaded) { int rc = wb_upload(u); if (rc) return rc; if (u->fp32) { rc = innfer_wbc_set_precision(u, 1); if (rc) return rc; } }
    if (u->fp32) {
        if (in_dtype != INNFER_F32 || out_dtype != INNFER_F32) return set_error(INNFER_ERR_INVALID, "wbc_forward: the fp32 mode takes and returns fp32 tensors");
        if (u->f32_w.empty()) return set_error(INNFER_ERR_INVALID, "wbc_forward: call innfer_wbc_set_precision(u, 1) after the last innfer_wbc_set_param");
        const WCarve32 c32 = wcarve32(u, N, H, W);
        if (ws_bytes < c32.total) return set_error(INNFER_ERR_WORKSPACE, "wbc_forward: workspace %zu < %zu bytes", ws_bytes, c32.total);
        return wbc_forward_f32(u, (const float*)d_in, (float*)d_out, N, H, W, (char*)d_ws, (hipStream_t)stream);
    }
    const WCarve cv = wcarve(N, H, W);
    if (ws_bytes < cv.total) return set_error(INNFER_ERR_WORKSPACE, "wbc_forward: workspace %zu < %zu bytes", ws_bytes, cv.total);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d_ws;
    float* raw = (float*)(ws + cv.raw);
    int dy49[49], dx49[49], dy9[9], dx9[9];
    for (int t = 0; t < 49; ++t) { dy49[t] = t / 7 - 3; dx49[t] = t % 7 - 3; }
    int dy7[7], dx7[7];                                  // first conv on the row-patch slab: vertical taps only
    for (int t = 0; t < 7; ++t) { dy7[t] = t - 3; dx7[t] = 0; }
    int dy9tf[9], dx9tf[9];                              // tf_same_padding: pad (0,1,0,1) in front of the stride-2 convs
    for (int t = 0; t < 9; ++t) { dy9[t] = t / 3 - 1; dx9[t] = t % 3 - 1; dy9tf[t] = t / 3; dx9tf[t] = t % 3; }
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    size_t li = 0;
    // conv of the next layer: in (Hi x Wi) -> (Ho x Wo), then bias / act / residual into dst (or NCHW output)
    auto layer = [&](const f16* in, int Hi, int Wi, int stride, int act, const f16* res, f16* dst, void* nchw) -> int {
        const Layer& l = u->layers[li++];
        const int Ho = Hi / stride, Wo = Wi / stride;
        if (l.d_w3 && l.k == 7 && stride == 1 && nchw) {     // 7x7, zero padding 3, planar output
            ConvLaunch L{};
            L.in = in; L.in_gstride = (long)N * Hi * Wi * 32; L.C = l.cin;
            L.wpk = (const f16*)l.d_w3; L.bias = l.d_b3;
            L.out = nchw; L.K = l.cout; L.N = N; L.H = Ho; L.W = Wo; L.act = act ? 1 : 0;
            L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = Ho;
            L.out_mode = OUT_NCHW; L.out_f32 = out_dtype == INNFER_F32; L.conv7 = 1;
            return conv_launch(L, s);
        }
        if (l.d_w3 && l.k == 7 && 7 * l.cin <= 32 && stride == 1 && !nchw && !res) {     // first conv: 7 x 1 column conv over the row-patch slab
            ConvLaunch L{};
            L.in = in; L.in_gstride = (long)N * Hi * Wi * 32; L.C = 32;
            L.wpk = (const f16*)l.d_w3; L.bias = l.d_b3;
            L.out = dst; L.out_gstride = (long)N * Ho * Wo * 32; L.K = l.cout;
            L.N = N; L.H = Ho; L.W = Wo; L.act = act ? 1 : 0; L.s1 = L.s2 = 1.f;
            L.y0 = 0; L.y1 = Ho; L.out_mode = OUT_SLAB; L.conv7v = 1;
            return conv_launch(L, s);
        }
        if (l.d_ws2 && l.d_b3 && stride == 2 && !nchw && !res && (Hi % 2 == 0) && (Wi % 2 == 0)) {     // stride-2 3x3 conv on the stride-2 gather loader: bias + LeakyReLU + slab out
            ConvLaunch L{};
            L.in = in; L.in_gstride = (long)N * Hi * Wi * 32; L.C = l.cin;
            L.wpk = (const f16*)l.d_ws2; L.bias = l.d_b3;
            L.out = dst; L.out_gstride = (long)N * Ho * Wo * 32; L.K = l.cout;
            L.N = N; L.H = Ho; L.W = Wo; L.act = act ? 1 : 0; L.s1 = L.s2 = 1.f;
            L.y0 = 0; L.y1 = Ho; L.out_mode = OUT_SLAB; L.stride2 = 1;
            return conv_launch(L, s);
        }
        if (l.d_w3 && l.k == 3 && stride == 1 && !nchw) {     // zero-padded stride-1 3x3 conv: the SR path's halo-tile kernel, epilogue = bias / LeakyReLU / + residual
            ConvLaunch L{};
            L.in = in; L.in_gstride = (long)N * Hi * Wi * 32; L.C = l.cin;
            L.wpk = (const f16*)l.d_w3; L.bias = l.d_b3;
            L.out = dst; L.out_gstride = (long)N * Ho * Wo * 32; L.K = l.cout;
            L.N = N; L.H = Ho; L.W = Wo; L.act = act ? 1 : 0;
            L.res1 = res; L.res1_gstride = L.out_gstride; L.s1 = 1.f; L.s2 = 1.f;
            L.y0 = 0; L.y1 = Ho; L.out_mode = OUT_SLAB;
            return conv_launch(L, s);
        }
        const int cin_pad = (l.cin + 31) / 32 * 32, cout_pad = (l.cout + 63) / 64 * 64, rs = (l.cout + 3) / 4 * 4;
        if (l.k == 7 && 7 * l.cin <= 32) {
            CK(gg::launch(l.d_w, 32, cout_pad, in, (long)N * Hi * Wi * 32, N, Hi, Wi, raw, Ho, Wo, 1, 7, dy7, dx7, Ho, Wo, 1, 0, 0, 0, s, nullptr, 0, rs));
        } else {
            CK(gg::launch(l.d_w, cin_pad, cout_pad, in, (long)N * Hi * Wi * 32, N, Hi, Wi, raw, Ho, Wo, stride, l.k * l.k,
                          l.k == 7 ? dy49 : (stride == 2 && u->tf ? dy9tf : dy9), l.k == 7 ? dx49 : (stride == 2 && u->tf ? dx9tf : dx9),
                          Ho, Wo, 1, 0, 0, 0, s, nullptr, 0, rs));
        }
        const long npix = (long)N * Ho * Wo;
        const long nthr = npix * ((l.cout + 3) / 4);
        hipLaunchKernelGGL(wb_post, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (const float*)raw, rs, l.cout, npix, (const float*)l.d_b, act,
                           res, dst, npix * 32, nchw, out_dtype == INNFER_F32, (long)Ho * Wo);
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    };
    auto upadd = [&](const f16* src, const f16* skip, f16* dst, int C, int h, int w) -> int {
        const long nthr = (long)N * 4 * h * w * (C / 8);
        hipLaunchKernelGGL(wb_upadd, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, src, (long)N * h * w * 32, skip, dst,
                           (long)N * 4 * h * w * 32, C, N, h, w, u->tf);
        INNFER_HIP(hipGetLastError());
        return INNFER_OK;
    };
    f16 *XIN = (f16*)(ws + cv.xin), *X0 = (f16*)(ws + cv.x0), *T1 = (f16*)(ws + cv.t1), *X1 = (f16*)(ws + cv.x1), *T2 = (f16*)(ws + cv.t2),
        *A = (f16*)(ws + cv.a), *B = (f16*)(ws + cv.b), *Cc = (f16*)(ws + cv.c), *U1 = (f16*)(ws + cv.u1), *V1 = (f16*)(ws + cv.v1), *U0 = (f16*)(ws + cv.u0);
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    hipLaunchKernelGGL(wb_pre, dim3((unsigned)(((long)N * H * W + 255) / 256)), dim3(256), 0, s, d_in, in_dtype == INNFER_F32, 3, H, W, N, XIN);
    INNFER_HIP(hipGetLastError());
    CK(layer(XIN, H, W, 1, 1, nullptr, X0, nullptr));                      // conv
    CK(layer(X0, H, W, 2, 1, nullptr, T1, nullptr));                       // conv_1 (stride 2)
    CK(layer(T1, H2, W2, 1, 1, nullptr, X1, nullptr));                     // conv_2
    CK(layer(X1, H2, W2, 2, 1, nullptr, T2, nullptr));                     // conv_3 (stride 2)
    CK(layer(T2, H4, W4, 1, 1, nullptr, A, nullptr));                      // conv_4
    f16* t = A;
    f16* spare = B;
    for (int b = 0; b < 4; ++b) {                                           // ResBlocks
        CK(layer(t, H4, W4, 1, 1, nullptr, Cc, nullptr));
        CK(layer(Cc, H4, W4, 1, 0, t, spare, nullptr));
        f16* tmp = t; t = spare; spare = tmp;
    }
    CK(layer(t, H4, W4, 1, 1, nullptr, T2, nullptr));                      // conv_5 -> 64 ch (T2 is free)
    CK(upadd(T2, X1, U1, 64, H4, W4));                                      // up(x2) + x1
    CK(layer(U1, H2, W2, 1, 1, nullptr, V1, nullptr));                     // conv_6
    CK(layer(V1, H2, W2, 1, 1, nullptr, T1, nullptr));                     // conv_7 -> 32 ch (T1 is free)
    CK(upadd(T1, X0, U0, 32, H2, W2));                                      // up(x3) + x0
    CK(layer(U0, H, W, 1, 1, nullptr, XIN, nullptr));                      // conv_8 (XIN is free)
    CK(layer(XIN, H, W, 1, 0, nullptr, nullptr, d_out));                   // conv_9 -> NCHW
#undef CK
    return INNFER_OK;
}

extern "C" size_t innfer_guided_filter_workspace_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return 2 * (((size_t)N * C * H * W * 4 + 255) & ~(size_t)255);
}

extern "C" int innfer_guided_filter(const void* d_x, const void* d_y, int dtype, int N, int C, int H, int W, float eps, void* d_out,
                                    void* d_ws, size_t ws_bytes, void* stream) {
    if (!d_x || !d_y || !d_out || !d_ws) return set_error(INNFER_ERR_INVALID, "guided_filter: null argument");
    if (N <= 0 || C <= 0 || H < 2 || W < 2) return set_error(INNFER_ERR_INVALID, "guided_filter: bad shape (reflect padding needs >= 2 pixels)");
    if (ws_bytes < innfer_guided_filter_workspace_bytes(N, C, H, W)) return set_error(INNFER_ERR_WORKSPACE, "guided_filter: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)N * C * H * W;
    float* A = (float*)d_ws;
    float* B = (float*)((char*)d_ws + (((size_t)n * 4 + 255) & ~(size_t)255));
    const int f32 = dtype == INNFER_F32;
    hipLaunchKernelGGL(gf_ab, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_x, d_y, f32, (long)N * C, H, W, eps, A, B);
    hipLaunchKernelGGL(gf_out, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)A, (const float*)B, d_x, f32, (long)N * C, H, W, d_out);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

// filter2D (utils/utils.py:484-535): every plane of x [planes][H][W] cross-correlated with ONE kH x kW kernel after F.pad(x, (pl, pr, pt, pb), mode) --
// the output keeps the plane's size (pl + pr = kW - 1, pt + pb = kH - 1).  One thread per output value; the kernel sits in device memory (fp32).
// border: 0 constant (zeros), 1 reflect (no edge repeat), 2 replicate, 3 circular.
__device__ __forceinline__ int f2d_src(int i, int n, int border) {
    if (i >= 0 && i < n) return i;
    if (border == 1) { i = i < 0 ? -i : 2 * n - 2 - i; return i; }
    if (border == 2) return i < 0 ? 0 : n - 1;
    if (border == 3) { i %= n; return i < 0 ? i + n : i; }
    return -1;
}
__global__ void k_filter2d(const void* x, int f32, long planes, int H, int W, const float* k, int kH, int kW, int pl, int pt, int border, void* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H);
    const long base = i - (long)Y * W - X;
    float acc = 0.f;
    for (int a = 0; a < kH; ++a) {
        const int sy = f2d_src(Y + a - pt, H, border);
        for (int b = 0; b < kW; ++b) {
            const int sx = f2d_src(X + b - pl, W, border);
            const float v = (sy < 0 || sx < 0) ? 0.f : ld(x, base + (long)sy * W + sx, f32);
            acc = fmaf(k[a * kW + b], v, acc);
        }
    }
    if (f32) ((float*)out)[i] = acc; else ((f16*)out)[i] = (f16)acc;
}

extern "C" int innfer_filter2d(const void* d_x, int dtype, long planes, int H, int W, const float* d_kernel, int kH, int kW, int pad_left, int pad_top,
                               int border, void* d_out, void* stream) {
    if (!d_x || !d_kernel || !d_out || planes <= 0 || H <= 0 || W <= 0 || kH <= 0 || kW <= 0) return set_error(INNFER_ERR_INVALID, "filter2d: bad argument");
    if (border < 0 || border > 3 || pad_left < 0 || pad_left >= kW || pad_top < 0 || pad_top >= kH)
        return set_error(INNFER_ERR_INVALID, "filter2d: border %d (0 constant, 1 reflect, 2 replicate, 3 circular), pad (%d, %d) for a %d x %d kernel", border, pad_left, pad_top, kH, kW);
    // F.pad's own limits: a reflection needs pad < size, a circular pad <= size
    const int pmax_y = std::max(pad_top, kH - 1 - pad_top), pmax_x = std::max(pad_left, kW - 1 - pad_left);
    if ((border == 1 && (pmax_y >= H || pmax_x >= W)) || (border == 3 && (pmax_y > H || pmax_x > W)))
        return set_error(INNFER_ERR_INVALID, "filter2d: padding (%d, %d) exceeds what the %d x %d plane allows for this border mode", pmax_x, pmax_y, H, W);
    const long n = planes * H * W;
    hipLaunchKernelGGL(k_filter2d, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x, dtype == INNFER_F32, planes, H, W, d_kernel, kH, kW,
                       pad_left, pad_top, border, d_out);
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}

extern "C" int innfer_guided_filter_ex(const void* d_x, const void* d_y, int dtype, int N, int C, int H, int W, int ks, float eps,
                                       const void* d_x_hr, int Hh, int Wh, void* d_out, void* d_ws, size_t ws_bytes, void* stream) {
    if (!d_x || !d_y || !d_out || !d_ws) return set_error(INNFER_ERR_INVALID, "guided_filter: null argument");
    if (ks < 1 || !(ks & 1)) return set_error(INNFER_ERR_UNSUPPORTED, "guided_filter: window size %d (odd sizes are built: ks = 2 r + 1)", ks);
    const int r = ks / 2;
    if (N <= 0 || C <= 0 || H <= r || W <= r) return set_error(INNFER_ERR_INVALID, "guided_filter: %dx%d image, window radius %d (reflect padding needs radius < size)", H, W, r);
    if (d_x_hr && (Hh <= 0 || Wh <= 0)) return set_error(INNFER_ERR_INVALID, "guided_filter: bad high-resolution size");
    if (ws_bytes < innfer_guided_filter_workspace_bytes(N, C, H, W)) return set_error(INNFER_ERR_WORKSPACE, "guided_filter: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)N * C * H * W;
    float* A = (float*)d_ws;
    float* B = (float*)((char*)d_ws + (((size_t)n * 4 + 255) & ~(size_t)255));
    const int f32 = dtype == INNFER_F32;
    const float k = (float)(1.0 / ((double)ks * (double)ks));
    hipLaunchKernelGGL(gf_ab_r, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_x, d_y, f32, (long)N * C, H, W, r, k, eps, A, B);
    if (d_x_hr) {
        const long nh = (long)N * C * Hh * Wh;
        hipLaunchKernelGGL(gf_out_fast, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, s, (const float*)A, (const float*)B, d_x_hr, f32, (long)N * C, H, W, Hh, Wh, d_out);
    } else {
        hipLaunchKernelGGL(gf_out_r, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)A, (const float*)B, d_x, f32, (long)N * C, H, W, r, k, d_out);
    }
    INNFER_HIP(hipGetLastError());
    return INNFER_OK;
}
