// PPON (Progressive Perception-Oriented Network) forward on gfx950 -- SURVEY.md section 8f row n3.
// Replaces PPON.forward with RRBlock_32 / _ResBlock_32 (architectures/PPON_arch.py:12-129; defaults
// utils/defaults.py:68-77: nf 64, nb 24, alpha 1, leakyrelu).
//
//   3x3 convs 64->64 (c1, LR_conv, up-convs, HR_conv0), 64->3 (HR_conv1)     conv3x3.hip (conv_launch): fused bias /
//                                                                           LeakyReLU / nearest-2x / shortcut add
//   first conv 3->64                                                        conv_first.hip
//   eight dilated 3x3 convs 64->32, rates 1..8 (conv_layer(.., dilation))   ONE launch of the halo-tile kernel (conv3x3_pc<..,POLY>, dilation groups):
//                                                                           a rate-d conv = ordinary 3x3 convs on the d*d polyphase components
//                                                                           of the tile; fp16 results in the eight groups of one 256-ch slab
//                                                                           (INNFER_PPON_POLY=0: the first version, one grouped gather GEMM)
//   d1, d1+d2, ..., d1+..+d8 -> cat -> LeakyReLU                            ppon_comb_slab: running sums in fp32, in place on that slab
//   c2 1x1 256->64, *0.2 + input (and the RRBlock's out*0.2 + input)        the halo-tile kernel's one-tap instantiation, both residual stages in its epilogue
//   out_s = SRM(..) + out_c, out_p = alpha * PRM(..) + out_s                ppon_axpy on the planar outputs
// Activations are blocked-NHWC fp16 slabs like everywhere else; the three reconstruction heads reuse one
// set of HR buffers.  Returns (out_c, out_s, out_p) like the reference; run.py keeps out_p.
#include "common.h"
#include "gather_gemm.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace innfer;

namespace {

// raw[px][256] fp32: the eight 32-channel dilated conv results of a residual block
constexpr int RAW_ROW = 256;

// in place on the 256-channel slab of the eight dilated conv results d1..d8 (fp16, bias included): group r <- lrelu(d1 + .. + d(r+1)),
// running sums in fp32; one thread per (pixel, 8 channels)
__global__ void ppon_comb_slab(f16* comb, long g, long npix) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 4) return;
    const long pix = i >> 2;
    const int c = (int)(i & 3) * 8;
    f16* q = comb + pix * 32 + c;
    float run[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f16x8 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *(const f16x8*)(q + k * g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            run[e] += (float)v[k][e];
            h[e] = (f16)fmaxf(run[e], 0.2f * run[e]);
        }
        *(f16x8*)(q + k * g) = h;
    }
}

// comb[.., 32*r + c] = lrelu(sum_{i<=r} (raw[.., 32*i + c] + bias[32*i + c])): one thread per (pixel, 4 channels)
__global__ void ppon_comb(const float* raw, const float* bias, long npix, f16* comb, long g) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 8) return;
    const long pix = i >> 3;
    const int c = (int)(i & 7) * 4;
    const float* r = raw + pix * RAW_ROW + c;
    float run[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x4 v = *(const f32x4*)(r + 32 * k);
        const f32x4 b = *(const f32x4*)(bias + 32 * k + c);
        f16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            run[e] += v[e] + b[e];
            h[e] = (f16)fmaxf(run[e], 0.2f * run[e]);
        }
        *(f16x4*)(comb + k * g + pix * 32 + c) = h;
    }
}

// out = x + 0.2 * (raw2 + bias), optionally followed by the RRBlock's  out * 0.2 + rrb_in
__global__ void ppon_res(const float* raw2, const float* bias, long npix, const f16* x, const f16* rrb_in, f16* out, long g) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 16) return;
    const long pix = i >> 4;
    const int c = (int)(i & 15) * 4;
    const f32x4 v = *(const f32x4*)(raw2 + pix * 64 + c);
    const f32x4 b = *(const f32x4*)(bias + c);
    const long o = (c >> 5) * g + pix * 32 + (c & 31);
    const f16x4 xv = *(const f16x4*)(x + o);
    f16x4 h;
    if (rrb_in) {
        const f16x4 rv = *(const f16x4*)(rrb_in + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (f16)(((float)xv[e] + (v[e] + b[e]) * 0.2f) * 0.2f + (float)rv[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (f16)((float)xv[e] + (v[e] + b[e]) * 0.2f);
    }
    *(f16x4*)(out + o) = h;
}

// planar outputs: dst = a * x + y   (x may alias dst)
__global__ void ppon_axpy(const void* x, const void* y, void* dst, float a, long n, int f32) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (f32) ((float*)dst)[i] = a * ((const float*)x)[i] + ((const float*)y)[i];
    else ((f16*)dst)[i] = (f16)(a * (float)((const f16*)x)[i] + (float)((const f16*)y)[i]);
}

struct Param { std::string key; std::vector<int> shape; std::vector<float> host; bool set = false; };

struct Conv3 { int w = -1, b = -1, K = 0, C = 0; void* d_w = nullptr; float* d_b = nullptr;      // conv3x3.hip panels
               void* d_up4 = nullptr; float* d_b4 = nullptr;     // the conv of an upconv_block also as four 2x2-tap phases (conv_pack_up2x_phases, bias once per phase)
               void* d_up4p = nullptr;                           //   64 -> 64: the same panels in the plane row order, for the one-visit form (conv3x3_pc UP4 | ROWP, as net.hip)
               void* d_fuse = nullptr; };                        // a head's last conv (64 -> <= 3): its panel for the epilogue of HR_conv0 (conv_pack_fuse_last)
struct ResB { Conv3 c1; int d_w[8], d_b[8], c2_w, c2_b; void* d_dw3[8] = {};   // [0]: halo-tile panels of the eight dilated convs, back to back
              f16* d_dw = nullptr; long dw_bytes = 0; f16* d_c2 = nullptr; float* d_dbias = nullptr; float* d_c2b = nullptr;
              void* d_c2t = nullptr; };        // c2 as a centre-tap panel for the halo-tile kernel
struct Head { Conv3 up[3]; Conv3 hr0, hr1; };

}  // namespace

struct innfer_ppon {
    int in_nc = 3, out_nc = 3, nf = 64, nb = 24, scale = 4, n_up = 2;
    float alpha = 1.f;
    std::vector<Param> params;
    int fea_w = -1, fea_b = -1;
    float* d_fea_w = nullptr; float* d_fea_b = nullptr;
    std::vector<ResB> rbs;                   // (nb + 4) * 3 residual blocks: CFEM trunk, SFEM, PFEM
    Conv3 lr;                                // CFEM.1.sub.<nb>
    Head heads[3];                           // CRM, SRM, PRM
    bool fp32 = false;                       // innfer_ppon_set_precision(1): PPON.forward in fp32 on NCHW fp32 tensors (f32ops.hip), the reference's -no_fp16 mode
    std::vector<float*> f32_w;               //   f32conv panels in forward order
    bool uploaded = false;
};

static int PP(innfer_ppon* p, const std::string& key, std::vector<int> shape) {
    Param q; q.key = key; q.shape = shape;
    p->params.push_back(q);
    return (int)p->params.size() - 1;
}

static Conv3 add_conv3(innfer_ppon* p, const std::string& key, int K, int C) {
    Conv3 c; c.K = K; c.C = C;
    c.w = PP(p, key + ".weight", {K, C, 3, 3});
    c.b = PP(p, key + ".bias", {K});
    return c;
}

static void add_rrblock(innfer_ppon* p, const std::string& prefix) {
    const int nf = p->nf;
    for (int k = 1; k <= 3; ++k) {
        const std::string b = prefix + "RB" + std::to_string(k) + ".";
        ResB r;
        r.c1 = add_conv3(p, b + "c1", nf, nf);
        for (int d = 0; d < 8; ++d) {
            r.d_w[d] = PP(p, b + "d" + std::to_string(d + 1) + ".weight", {nf / 2, nf, 3, 3});
            r.d_b[d] = PP(p, b + "d" + std::to_string(d + 1) + ".bias", {nf / 2});
        }
        r.c2_w = PP(p, b + "c2.weight", {nf, nf * 4, 1, 1});
        r.c2_b = PP(p, b + "c2.bias", {nf});
        p->rbs.push_back(r);
    }
}

extern "C" int innfer_ppon_create(innfer_ppon** out, int in_nc, int out_nc, int nf, int nb, int scale, float alpha) {
    if (!out) return set_error(INNFER_ERR_INVALID, "ppon_create: null out");
    if (nf != 64 || in_nc < 1 || in_nc > 8 || out_nc < 1 || out_nc > 16 || nb < 1 || (scale != 1 && scale != 2 && scale != 3 && scale != 4 && scale != 8))
        return set_error(INNFER_ERR_UNSUPPORTED, "ppon_create: nf=%d scale=%d (built: nf 64, scale 1/2/3/4/8)", nf, scale);
    innfer_ppon* p = new innfer_ppon();
    p->in_nc = in_nc; p->out_nc = out_nc; p->nf = nf; p->nb = nb; p->scale = scale; p->alpha = alpha;
    p->n_up = scale == 8 ? 3 : (scale == 4 ? 2 : (scale == 2 || scale == 3 ? 1 : 0));          // upscale 3: ONE Upsample(3) stage (PPON_arch.py:20-22,33-36)
    p->fea_w = PP(p, "CFEM.0.weight", {nf, in_nc, 3, 3});
    p->fea_b = PP(p, "CFEM.0.bias", {nf});
    for (int b = 0; b < nb; ++b) add_rrblock(p, "CFEM.1.sub." + std::to_string(b) + ".");
    p->lr = add_conv3(p, "CFEM.1.sub." + std::to_string(nb), nf, nf);
    for (int b = 0; b < 2; ++b) add_rrblock(p, "SFEM." + std::to_string(b) + ".");
    for (int b = 0; b < 2; ++b) add_rrblock(p, "PFEM." + std::to_string(b) + ".");
    const char* hn[3] = {"CRM.", "SRM.", "PRM."};
    // the reference declares the heads in the order CRM, SRM, PRM; keys: <3u+1> up-convs, then <3n>, <3n+2>
    for (int h = 0; h < 3; ++h) {
        Head& H = p->heads[h];
        for (int u = 0; u < p->n_up; ++u) H.up[u] = add_conv3(p, hn[h] + std::to_string(3 * u + 1), nf, nf);
        H.hr0 = add_conv3(p, hn[h] + std::to_string(3 * p->n_up), nf, nf);
        H.hr1 = add_conv3(p, hn[h] + std::to_string(3 * p->n_up + 2), out_nc, nf);
    }
    *out = p;
    return INNFER_OK;
}

static void free_conv3(Conv3& c) {
    if (c.d_w) (void)hipFree(c.d_w);
    if (c.d_b) (void)hipFree(c.d_b);
    if (c.d_up4) (void)hipFree(c.d_up4);
    if (c.d_up4p) (void)hipFree(c.d_up4p);
    if (c.d_b4) (void)hipFree(c.d_b4);
    if (c.d_fuse) (void)hipFree(c.d_fuse);
    c.d_w = nullptr; c.d_b = nullptr; c.d_up4 = nullptr; c.d_up4p = nullptr; c.d_b4 = nullptr; c.d_fuse = nullptr;
}

static void free_device(innfer_ppon* p) {
    for (auto v : p->f32_w) if (v) (void)hipFree(v);
    p->f32_w.clear();
    if (p->d_fea_w) (void)hipFree(p->d_fea_w);
    if (p->d_fea_b) (void)hipFree(p->d_fea_b);
    p->d_fea_w = p->d_fea_b = nullptr;
    for (auto& r : p->rbs) {
        free_conv3(r.c1);
        if (r.d_dw) (void)hipFree(r.d_dw);
        r.d_dw = nullptr;
        for (auto& w3 : r.d_dw3) { if (w3) (void)hipFree(w3); w3 = nullptr; }
        if (r.d_c2) (void)hipFree(r.d_c2);
        if (r.d_c2t) (void)hipFree(r.d_c2t);
        r.d_c2t = nullptr;
        if (r.d_dbias) (void)hipFree(r.d_dbias);
        if (r.d_c2b) (void)hipFree(r.d_c2b);
        r.d_c2 = nullptr; r.d_dbias = r.d_c2b = nullptr;
    }
    free_conv3(p->lr);
    for (auto& H : p->heads) { for (auto& u : H.up) free_conv3(u); free_conv3(H.hr0); free_conv3(H.hr1); }
}

extern "C" void innfer_ppon_destroy(innfer_ppon* p) {
    if (!p) return;
    free_device(p);
    delete p;
}

extern "C" int innfer_ppon_num_params(innfer_ppon* p) { return p ? (int)p->params.size() : INNFER_ERR_INVALID; }

extern "C" int innfer_ppon_param_info(innfer_ppon* p, int idx, char* key, size_t key_cap, int* ndim, int* shape4) {
    if (!p || idx < 0 || idx >= (int)p->params.size()) return set_error(INNFER_ERR_INVALID, "ppon_param_info: bad index");
    const Param& q = p->params[idx];
    if (key && key_cap) { strncpy(key, q.key.c_str(), key_cap - 1); key[key_cap - 1] = 0; }
    if (ndim) *ndim = (int)q.shape.size();
    if (shape4) for (size_t i = 0; i < 4; ++i) shape4[i] = i < q.shape.size() ? q.shape[i] : 1;
    return INNFER_OK;
}

extern "C" int innfer_ppon_set_param(innfer_ppon* p, int idx, const float* h_data) {
    if (!p || idx < 0 || idx >= (int)p->params.size() || !h_data) return set_error(INNFER_ERR_INVALID, "ppon_set_param: bad arguments");
    Param& q = p->params[idx];
    size_t n = 1;
    for (int s : q.shape) n *= (size_t)s;
    q.host.assign(h_data, h_data + n);
    q.set = true;
    p->uploaded = false;
    return INNFER_OK;
}

namespace {

int upload_conv3(innfer_ppon* p, Conv3& c, int role = 0) {       // role 1: the conv of an upconv_block (factor 2), 2: a head's last conv
    const std::vector<float>& w = p->params[c.w].host;
    const std::vector<float>& b = p->params[c.b].host;
    std::vector<char> host(conv_packed_bytes(c.K, c.C));
    conv_pack(w.data(), c.K, c.C, host.data());
    const int per = 16 * conv_nt_for(c.K);
    const size_t bias_n = (size_t)((c.K + per - 1) / per) * per;
    std::vector<float> bias(bias_n, 0.f);
    for (int k = 0; k < c.K; ++k) bias[k] = b[k];
    INNFER_HIP(hipMalloc(&c.d_w, host.size()));
    INNFER_HIP(hipMalloc((void**)&c.d_b, bias_n * sizeof(float)));
    INNFER_HIP(hipMemcpy(c.d_w, host.data(), host.size(), hipMemcpyHostToDevice));
    INNFER_HIP(hipMemcpy(c.d_b, bias.data(), bias_n * sizeof(float), hipMemcpyHostToDevice));
    if (role == 1 && c.K % 64 == 0 && c.C % 32 == 0) {
        std::vector<char> pk(conv_packed_bytes_deconv2x(c.K, c.C));
        conv_pack_up2x_phases(w.data(), c.K, c.C, pk.data());
        std::vector<float> b4((size_t)4 * c.K);
        for (int ph = 0; ph < 4; ++ph) for (int k = 0; k < c.K; ++k) b4[(size_t)ph * c.K + k] = b[k];
        INNFER_HIP(hipMalloc(&c.d_up4, pk.size()));
        INNFER_HIP(hipMalloc((void**)&c.d_b4, b4.size() * sizeof(float)));
        INNFER_HIP(hipMemcpy(c.d_up4, pk.data(), pk.size(), hipMemcpyHostToDevice));
        INNFER_HIP(hipMemcpy(c.d_b4, b4.data(), b4.size() * sizeof(float), hipMemcpyHostToDevice));
        if (c.K == 64 && c.C == 64) {
            conv_pack_up2x_phases(w.data(), c.K, c.C, pk.data(), 1);
            INNFER_HIP(hipMalloc(&c.d_up4p, pk.size()));
            INNFER_HIP(hipMemcpy(c.d_up4p, pk.data(), pk.size(), hipMemcpyHostToDevice));
        }
    }
    if (role == 2 && c.C == 64 && c.K <= 3) {
        std::vector<char> fp(4096);
        conv_pack_fuse_last(w.data(), c.K, fp.data());
        INNFER_HIP(hipMalloc(&c.d_fuse, fp.size()));
        INNFER_HIP(hipMemcpy(c.d_fuse, fp.data(), fp.size(), hipMemcpyHostToDevice));
    }
    return INNFER_OK;
}

int upload(innfer_ppon* p) {
    for (auto& q : p->params) if (!q.set) return set_error(INNFER_ERR_INVALID, "ppon: parameter '%s' was never set", q.key.c_str());
    free_device(p);
    const int nf = p->nf;
    {   // first conv: [Cin*9][K] fp32, k-major (conv_first.hip)
        const std::vector<float>& w = p->params[p->fea_w].host;
        std::vector<float> t((size_t)p->in_nc * 9 * nf);
        for (int ci = 0; ci < p->in_nc; ++ci)
            for (int tp = 0; tp < 9; ++tp)
                for (int k = 0; k < nf; ++k) t[((size_t)ci * 9 + tp) * nf + k] = w[((size_t)k * p->in_nc + ci) * 9 + tp];
        INNFER_HIP(hipMalloc((void**)&p->d_fea_w, t.size() * sizeof(float)));
        INNFER_HIP(hipMemcpy(p->d_fea_w, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        const std::vector<float>& b = p->params[p->fea_b].host;
        INNFER_HIP(hipMalloc((void**)&p->d_fea_b, nf * sizeof(float)));
        INNFER_HIP(hipMemcpy(p->d_fea_b, b.data(), nf * sizeof(float), hipMemcpyHostToDevice));
    }
    std::vector<f16> panel;
    for (auto& r : p->rbs) {
        int rc = upload_conv3(p, r.c1);
        if (rc) return rc;
        std::vector<float> dbias(256);
        std::vector<f16> all;                                     // the eight dilated convs' panels, back to back
        for (int d = 0; d < 8; ++d) {
            const std::vector<float>& w = p->params[r.d_w[d]].host;
            gg::pack_panels(panel, nf / 2, nf, nf, 9, [&](int co, int ci, int t) { return w[((size_t)co * nf + ci) * 9 + t]; });
            r.dw_bytes = (long)(panel.size() * sizeof(f16));
            all.insert(all.end(), panel.begin(), panel.end());
            const std::vector<float>& b = p->params[r.d_b[d]].host;
            for (int k = 0; k < nf / 2; ++k) dbias[32 * d + k] = b[k];
        }
        if (nf == 64) {          // the halo-tile kernel's polyphase form of the dilated convs: eight 32-output panels back to back (rate = channel group)
            const size_t pb = conv_packed_bytes(nf / 2, nf);
            std::vector<char> packed(8 * pb);
            for (int d = 0; d < 8; ++d) conv_pack(p->params[r.d_w[d]].host.data(), nf / 2, nf, packed.data() + d * pb);
            INNFER_HIP(hipMalloc(&r.d_dw3[0], packed.size()));
            INNFER_HIP(hipMemcpy(r.d_dw3[0], packed.data(), packed.size(), hipMemcpyHostToDevice));
        }
        INNFER_HIP(hipMalloc((void**)&r.d_dw, all.size() * sizeof(f16)));
        INNFER_HIP(hipMemcpy(r.d_dw, all.data(), all.size() * sizeof(f16), hipMemcpyHostToDevice));
        INNFER_HIP(hipMalloc((void**)&r.d_dbias, 256 * sizeof(float)));
        INNFER_HIP(hipMemcpy(r.d_dbias, dbias.data(), 256 * sizeof(float), hipMemcpyHostToDevice));
        const std::vector<float>& w2 = p->params[r.c2_w].host;
        gg::pack_panels(panel, nf, 4 * nf, 4 * nf, 1, [&](int co, int ci, int) { return w2[(size_t)co * 4 * nf + ci]; });
        INNFER_HIP(hipMalloc((void**)&r.d_c2, panel.size() * sizeof(f16)));
        INNFER_HIP(hipMemcpy(r.d_c2, panel.data(), panel.size() * sizeof(f16), hipMemcpyHostToDevice));
        if (nf == 64) {
            std::vector<char> pk(conv_packed_bytes_taps(nf, 4 * nf, 0x10));
            conv_pack_1x1(w2.data(), nf, 4 * nf, pk.data());
            INNFER_HIP(hipMalloc(&r.d_c2t, pk.size()));
            INNFER_HIP(hipMemcpy(r.d_c2t, pk.data(), pk.size(), hipMemcpyHostToDevice));
        }
        INNFER_HIP(hipMalloc((void**)&r.d_c2b, nf * sizeof(float)));
        INNFER_HIP(hipMemcpy(r.d_c2b, p->params[r.c2_b].host.data(), nf * sizeof(float), hipMemcpyHostToDevice));
    }
    int rc = upload_conv3(p, p->lr);
    if (rc) return rc;
    for (auto& H : p->heads) {
        for (int u = 0; u < p->n_up; ++u) { rc = upload_conv3(p, H.up[u], p->scale == 3 ? 0 : 1); if (rc) return rc; }
        rc = upload_conv3(p, H.hr0); if (rc) return rc;
        rc = upload_conv3(p, H.hr1, 2); if (rc) return rc;
    }
    p->uploaded = true;
    return INNFER_OK;
}

// nearest 3x of a 64-channel slab (two groups), 8 channels per thread
__global__ void ppon_upsample3(const f16* src, long src_g, f16* dst, long dst_g, int N, int H, int W) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int WO = 3 * W, HO = 3 * H;
    const long per = (long)N * HO * WO * 4;
    if (i >= per * 2) return;
    const int g = (int)(i / per);
    const long r = i % per;
    const int q = (int)(r & 3);
    const long m = r >> 2;
    const int X = (int)(m % WO), Y = (int)((m / WO) % HO);
    const long n = m / ((long)WO * HO);
    const long sp = (n * H + Y / 3) * W + X / 3;
    *(f16x8*)(dst + g * dst_g + m * 32 + q * 8) = *(const f16x8*)(src + g * src_g + sp * 32 + q * 8);
}

struct QCarve { size_t fea, t[4], o1, comb, raw, raw2, cfem, sfem, up[3], hr, tmp_c, tmp_s, total; };

QCarve qcarve(const innfer_ppon* p, int N, int H, int W, int out_elt) {
    QCarve c{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t px = (size_t)N * H * W;
    size_t off = 0;
    auto slab = [&](size_t pixels, int ch) { size_t o = off; off += al(pixels * ch * 2); return o; };
    c.fea = slab(px, 64);
    for (int i = 0; i < 4; ++i) c.t[i] = slab(px, 64);
    c.o1 = slab(px, 64);
    c.comb = slab(px, 256);
    c.raw = off; off += al(px * RAW_ROW * 4);
    c.raw2 = off; off += al(px * 64 * 4);
    c.cfem = slab(px, 64);
    c.sfem = slab(px, 64);
    size_t m = 1;
    for (int u = 0; u < p->n_up; ++u) { m *= p->scale == 3 ? 9 : 4; c.up[u] = slab(px * m, 64); }
    c.hr = slab(px * m, 64);
    c.tmp_c = off; off += al(px * m * p->out_nc * out_elt);
    c.tmp_s = off; off += al(px * m * p->out_nc * out_elt);
    c.total = off;
    return c;
}

}  // namespace

namespace {
// ---- the fp32 mode: PPON.forward (PPON_arch.py:65-129) on NCHW fp32 tensors with the generic fp32 ops of f32ops.hip; graph = oracle/nets.py ppon_forward ----
struct QCarve32 { size_t fea, t[4], o1, d, cfem, sfem, pfem, up[3], hr, ups, tmp_c, tmp_s, total; };
QCarve32 qcarve32(const innfer_ppon* p, int N, int H, int W) {
    QCarve32 c{};
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t px = (size_t)N * H * W, nf = p->nf;
    size_t off = 0;
    auto buf = [&](size_t floats) { size_t o = off; off += al(floats * 4); return o; };
    c.fea = buf(px * nf);
    for (int i = 0; i < 4; ++i) c.t[i] = buf(px * nf);
    c.o1 = buf(px * nf); c.d = buf(px * nf * 4); c.cfem = buf(px * nf); c.sfem = buf(px * nf); c.pfem = buf(px * nf);
    size_t m = 1;
    for (int u = 0; u < p->n_up; ++u) { m *= p->scale == 3 ? 9 : 4; c.up[u] = buf(px * m * nf); }
    c.hr = buf(px * m * nf);
    c.ups = p->scale == 3 ? buf(px * 9 * nf) : 0;
    c.tmp_c = buf(px * m * p->out_nc); c.tmp_s = buf(px * m * p->out_nc);
    c.total = off;
    return c;
}

int ppon_forward_f32(innfer_ppon* p, const float* x, float* out_c, float* out_s, float* out_p, int N, int H, int W, char* ws, hipStream_t s) {
    const QCarve32 cv = qcarve32(p, N, H, W);
    const int nf = p->nf;
    const long hw = (long)H * W;
    auto B = [&](size_t o) { return (float*)(ws + o); };
    if (!out_c) out_c = B(cv.tmp_c);
    if (!out_s) out_s = B(cv.tmp_s);
    int wi = 0;
    // 3x3 conv with dilation `dil` (zero padding = dil; 1x1 when ksz == 1) over C channels at h x w (read through nearest-2x when `up`) -> K channels of a tensor with ktot channels
    auto conv = [&](const float* in, int C, int h, int w, int up, int ksz, int dil, const float* bias, int K, float* out, int ktot, int act,
                    float oscale = 0.f, const float* res = nullptr, int rtot = 0) -> int {
        F32Conv c{};
        const int ho = up ? 2 * h : h, wo = up ? 2 * w : w;
        c.in = in; c.in_nstride = (long)C * h * w; c.in_cstride = (long)h * w; c.C = C; c.Hin = h; c.Win = w;
        c.wp = p->f32_w[wi++]; c.bias = bias; c.K = K;
        c.out = out; c.out_nstride = (long)ktot * ho * wo; c.out_cstride = (long)ho * wo; c.out_pstride = 1; c.Wout = wo;
        c.Ho = ho; c.Wo = wo; c.osy = c.osx = 1; c.isy = c.isx = 1; c.up = up;
        c.ntap = ksz * ksz;
        for (int t = 0; t < c.ntap; ++t) { c.dy[t] = (t / ksz - ksz / 2) * dil; c.dx[t] = (t % ksz - ksz / 2) * dil; }
        c.act = act; c.oscale = oscale; c.N = N;
        c.res = res; c.res_nstride = (long)rtot * ho * wo; c.res_cstride = (long)ho * wo;
        return f32conv_launch(c, s);
    };
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    float *FEA = B(cv.fea), *O1 = B(cv.o1), *D = B(cv.d), *CFEM = B(cv.cfem), *SFEM = B(cv.sfem), *PFEM = B(cv.pfem);
    float* T[4] = {B(cv.t[0]), B(cv.t[1]), B(cv.t[2]), B(cv.t[3])};
    CK(conv(x, p->in_nc, H, W, 0, 3, 1, p->d_fea_b, nf, FEA, nf, 0));
    size_t rbi = 0;
    // RRBlock_32 (PPON_arch.py:116-129): x -> RB1 -> RB2 -> RB3 -> * 0.2 + x; _ResBlock_32 (:79-114): c1, LeakyReLU, d1..d8 (dilation 1..8), running sums, LeakyReLU, c2 (1x1), * 0.2 + input
    auto rrblock = [&](const float* xin, float* dst, float* sa, float* sb) -> int {
        const float* cur = xin;
        for (int k = 0; k < 3; ++k) {
            const ResB& r = p->rbs[rbi++];
            float* o = k == 1 ? sb : sa;
            CK(conv(cur, nf, H, W, 0, 3, 1, r.c1.d_b, nf, O1, nf, 1));
            for (int d = 0; d < 8; ++d) CK(conv(O1, nf, H, W, 0, 3, d + 1, r.d_dbias + 32 * d, nf / 2, D + (long)(nf / 2) * d * hw, 4 * nf, 0));
            CK(f32_prefix_lrelu_launch(D, N, 8, nf / 2, hw, s));
            CK(conv(D, 4 * nf, H, W, 0, 1, 1, r.d_c2b, nf, o, nf, 0, 0.2f, cur, nf));
            cur = o;
        }
        return f32_axpy_launch(cur, xin, dst, 0.2f, (long)N * nf * hw, s);
    };
    // nblk RRBlocks from `from` (a buffer outside T); the last one writes `last` (outside T too).  T[0..3] rotate: a block's output and its two scratch
    // buffers are never the buffer it reads
    auto chain = [&](const float* from, int nblk, float* last) -> int {
        const float* c = from;
        int curi = -1;
        for (int b = 0; b < nblk; ++b) {
            int pick[3], n = 0;
            for (int i = 0; i < 4 && n < 3; ++i) if (i != curi) pick[n++] = i;
            float* dst = b == nblk - 1 ? last : T[pick[0]];
            CK(rrblock(c, dst, T[pick[1]], T[pick[2]]));
            c = dst;
            curi = b == nblk - 1 ? -1 : pick[0];
        }
        return INNFER_OK;
    };
    CK(chain(FEA, p->nb, PFEM));                                                  // (the trunk's output borrows the PFEM buffer: dead once CFEM is formed)
    CK(conv(PFEM, nf, H, W, 0, 3, 1, p->lr.d_b, nf, CFEM, nf, 0, 0.f, FEA, nf));  // CFEM = fea + LR_conv(trunk)
    auto recon = [&](const float* in, const Head& Hd, float* out, float oscale, const float* res) -> int {
        const float* c = in;
        int h = H, w = W;
        for (int u = 0; u < p->n_up; ++u) {
            float* dst = B(cv.up[u]);
            if (p->scale == 3) {
                CK(f32_upsample_launch(c, B(cv.ups), (long)N * nf, h, w, 3, 0, s));
                CK(conv(B(cv.ups), nf, 3 * h, 3 * w, 0, 3, 1, Hd.up[u].d_b, nf, dst, nf, 1));
                h *= 3; w *= 3;
            } else {
                CK(conv(c, nf, h, w, 1, 3, 1, Hd.up[u].d_b, nf, dst, nf, 1));
                h *= 2; w *= 2;
            }
            c = dst;
        }
        CK(conv(c, nf, h, w, 0, 3, 1, Hd.hr0.d_b, nf, B(cv.hr), nf, 1));
        return conv(B(cv.hr), nf, h, w, 0, 3, 1, Hd.hr1.d_b, p->out_nc, out, p->out_nc, 0, oscale, res, p->out_nc);
    };
    CK(recon(CFEM, p->heads[0], out_c, 0.f, nullptr));
    CK(chain(CFEM, 2, SFEM));
    CK(recon(SFEM, p->heads[1], out_s, 0.f, out_c));                              // out_s = SRM(sfem) + out_c
    CK(chain(SFEM, 2, PFEM));
    CK(recon(PFEM, p->heads[2], out_p, p->alpha, out_s));                         // out_p = alpha * PRM(pfem) + out_s
#undef CK
    return INNFER_OK;
}
}  // namespace

// The reference's fp16 switch for this generator (run.py:345,421-422), as innfer_unet_set_precision: fp32 = 1 runs PPON.forward in fp32 on NCHW fp32 tensors.
extern "C" int innfer_ppon_set_precision(innfer_ppon* p, int fp32) {
    if (!p || (fp32 != 0 && fp32 != 1)) return set_error(INNFER_ERR_INVALID, "ppon_set_precision: 0 (fp16 arithmetic) or 1 (fp32)");
    p->fp32 = fp32 != 0;
    if (!p->fp32) return INNFER_OK;
    if (!p->uploaded) { int rc = upload(p); if (rc) return rc; }
    if (!p->f32_w.empty()) return INNFER_OK;
    std::vector<float> host;
    auto plain = [&](int widx, int K, int C, int ksz) -> int {                    // torch layout [K][C][ksz][ksz]
        const std::vector<float>& w = p->params[widx].host;
        const int T = ksz * ksz;
        host.resize(f32conv_packed_floats(K, C, T));
        f32conv_pack(K, C, T, [&w, C, T](int k, int c, int t) { return w[((size_t)k * C + c) * T + t]; }, host.data());
        float* d = nullptr;
        INNFER_HIP(hipMalloc((void**)&d, host.size() * sizeof(float)));
        p->f32_w.push_back(d);
        INNFER_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
        return INNFER_OK;
    };
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    const int nf = p->nf;
    CK(plain(p->fea_w, nf, p->in_nc, 3));
    auto blocks = [&](size_t first, size_t count) -> int {
        for (size_t i = first; i < first + count; ++i) {
            const ResB& r = p->rbs[i];
            CK(plain(r.c1.w, nf, nf, 3));
            for (int d = 0; d < 8; ++d) CK(plain(r.d_w[d], nf / 2, nf, 3));
            CK(plain(r.c2_w, nf, 4 * nf, 1));
        }
        return INNFER_OK;
    };
    auto head = [&](const Head& Hd) -> int {
        for (int u = 0; u < p->n_up; ++u) CK(plain(Hd.up[u].w, nf, nf, 3));
        CK(plain(Hd.hr0.w, nf, nf, 3));
        CK(plain(Hd.hr1.w, p->out_nc, nf, 3));
        return INNFER_OK;
    };
    // forward order: CFEM trunk, LR conv, CRM head, SFEM, SRM head, PFEM, PRM head
    CK(blocks(0, (size_t)p->nb * 3));
    CK(plain(p->lr.w, nf, nf, 3));
    CK(head(p->heads[0]));
    CK(blocks((size_t)p->nb * 3, 6));
    CK(head(p->heads[1]));
    CK(blocks((size_t)p->nb * 3 + 6, 6));
    CK(head(p->heads[2]));
#undef CK
    return INNFER_OK;
}

extern "C" size_t innfer_ppon_workspace_bytes(innfer_ppon* p, int N, int H, int W) {
    if (!p || N <= 0 || H <= 0 || W <= 0) return 0;
    return p->fp32 ? qcarve32(p, N, H, W).total : qcarve(p, N, H, W, 4).total;
}

extern "C" int innfer_ppon_forward(innfer_ppon* p, const void* d_in, int in_dtype, void* d_out_c, void* d_out_s, void* d_out_p,
                                   int out_dtype, int N, int H, int W, void* d_ws, size_t ws_bytes, void* stream) {
    if (!p || !d_in || !d_out_p || !d_ws) return set_error(INNFER_ERR_INVALID, "ppon_forward: null argument");
    if (N <= 0 || H <= 0 || W <= 0) return set_error(INNFER_ERR_INVALID, "ppon_forward: bad shape");
    if (!p->uploaded) { int rc = upload(p); if (rc) return rc; if (p->fp32) { rc = innfer_ppon_set_precision(p, 1); if (rc) return rc; } }
    if (p->fp32) {
        if (in_dtype != INNFER_F32 || out_dtype != INNFER_F32) return set_error(INNFER_ERR_INVALID, "ppon_forward: the fp32 mode takes and returns fp32 tensors");
        if (p->f32_w.empty()) return set_error(INNFER_ERR_INVALID, "ppon_forward: call innfer_ppon_set_precision(p, 1) after the last innfer_ppon_set_param");
        const QCarve32 c32 = qcarve32(p, N, H, W);
        if (ws_bytes < c32.total) return set_error(INNFER_ERR_WORKSPACE, "ppon_forward: workspace %zu < %zu bytes", ws_bytes, c32.total);
        return ppon_forward_f32(p, (const float*)d_in, (float*)d_out_c, (float*)d_out_s, (float*)d_out_p, N, H, W, (char*)d_ws, (hipStream_t)stream);
    }
    const QCarve cv = qcarve(p, N, H, W, 4);
    if (ws_bytes < cv.total) return set_error(INNFER_ERR_WORKSPACE, "ppon_forward: workspace %zu < %zu bytes", ws_bytes, cv.total);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)d_ws;
    const long px = (long)N * H * W, G = px * 32;
    const int f32o = out_dtype == INNFER_F32;
    if (!d_out_c) d_out_c = ws + cv.tmp_c;
    if (!d_out_s) d_out_s = ws + cv.tmp_s;
#define CK(e) do { int _rc = (e); if (_rc) return _rc; } while (0)
    auto conv = [&](const Conv3& c, const f16* in, long in_g, void* out, long out_g, int h, int w, int act, int up,
                    const f16* res, long res_g, int mode) -> int {
        ConvLaunch L{};
        L.in = in; L.in_gstride = in_g; L.C = c.C;
        L.wpk = (const f16*)c.d_w; L.bias = c.d_b;
        L.out = out; L.out_gstride = out_g; L.out_coff = 0; L.K = c.K;
        L.N = N; L.H = h; L.W = w; L.act = act; L.up = up;
        L.res1 = res; L.res1_gstride = res_g; L.s1 = 1.f; L.s2 = 1.f;
        L.y0 = 0; L.y1 = h;
        L.out_mode = mode; L.out_f32 = f32o;
        return conv_launch(L, s);
    };
    f16 *FEA = (f16*)(ws + cv.fea), *O1 = (f16*)(ws + cv.o1), *COMB = (f16*)(ws + cv.comb), *CFEM = (f16*)(ws + cv.cfem),
        *SFEM = (f16*)(ws + cv.sfem);
    f16* T[4] = {(f16*)(ws + cv.t[0]), (f16*)(ws + cv.t[1]), (f16*)(ws + cv.t[2]), (f16*)(ws + cv.t[3])};
    float* raw = (float*)(ws + cv.raw);
    float* raw2 = (float*)(ws + cv.raw2);

    {   // CFEM.0
        FirstConvLaunch F{};
        F.in = d_in; F.in_f32 = in_dtype == INNFER_F32; F.Cin = p->in_nc; F.w = p->d_fea_w; F.bias = p->d_fea_b;
        F.out = FEA; F.out_gstride = G; F.out2 = nullptr; F.out2_gstride = 0;
        F.K = p->nf; F.N = N; F.H = H; F.W = W; F.act = 0;
        CK(first_conv_launch(F, s));
    }
    int dy[9], dx[9], d0[1] = {0};
    size_t rbi = 0;
    const bool poly = INNFER_KNOB("INNFER_PPON_POLY", 1) != 0;   // 0: grouped gather GEMM (A/B)
    // one RRBlock: x -> RB1 -> RB2 -> RB3 -> *0.2 + x, written to `dst` (any slab but x and the two scratch slabs)
    auto rrblock = [&](const f16* x, f16* dst, f16* sa, f16* sb) -> int {
        const f16* cur = x;
        for (int k = 0; k < 3; ++k) {
            const ResB& r = p->rbs[rbi++];
            f16* out = k == 2 ? dst : (k == 0 ? sa : sb);
            CK(conv(r.c1, cur, G, O1, G, H, W, 1, 0, nullptr, 0, OUT_SLAB));
            // the eight dilated convs of the block in ONE launch: group g = rate g+1 (taps scaled by the rate),
            // its own weight panel, its own 32-float column of the raw row
            if (r.d_dw3[0] && poly) {
                // the eight dilated convs as ONE launch of the halo-tile kernel (conv3x3_pc<..,POLY>, dilation groups): rate g+1 = output channel
                // group g; each is ordinary 3x3 convs on the rate^2 polyphase components of the tile, so every input pixel is staged once per
                // rate instead of once per tap and rate; fp16 results d_r (bias included) land in group r of COMB
                ConvLaunch Ld{};
                Ld.in = O1; Ld.in_gstride = G; Ld.C = 64;
                Ld.wpk = (const f16*)r.d_dw3[0]; Ld.bias = r.d_dbias;
                Ld.out = COMB; Ld.out_gstride = G; Ld.K = 256; Ld.N = N; Ld.H = H; Ld.W = W; Ld.act = 0;
                Ld.s1 = Ld.s2 = 1.f; Ld.y0 = 0; Ld.y1 = H; Ld.out_mode = OUT_SLAB; Ld.dilation_groups = 8;
                CK(conv_launch(Ld, s));
                // (d1, d1 + d2, .., d1 + .. + d8 -> LeakyReLU is the operand transform of c2 below; without the one-tap c2 the pass that writes it)
                if (!r.d_c2t) hipLaunchKernelGGL(ppon_comb_slab, dim3((unsigned)((px * 4 + 255) / 256)), dim3(256), 0, s, COMB, G, px);
            } else {
            for (int t = 0; t < 9; ++t) { dy[t] = t / 3 - 1; dx[t] = t % 3 - 1; }
            CK(gg::launch(r.d_dw, 64, 64, O1, G, N, H, W, raw, H, W, 1, 9, dy, dx, H, W, 1, 0, 0, 0, s, nullptr, 0, RAW_ROW, 0,
                          8, r.dw_bytes, 32, 1, 32));
            hipLaunchKernelGGL(ppon_comb, dim3((unsigned)((px * 8 + 255) / 256)), dim3(256), 0, s, (const float*)raw, (const float*)r.d_dbias, px, COMB, G);
            }
            if (r.d_c2t && poly) {
                // c2 (1x1, 256 -> 64) on the halo-tile kernel's one-tap instantiation; `(c2 + b) * 0.2 + input` and, at the end of the block of
                // three, `* 0.2 + block input` are its two residual epilogue stages (the RRDB epilogue of the SR path)
                ConvLaunch Lc{};
                Lc.in = COMB; Lc.in_gstride = G; Lc.C = 256;
                Lc.wpk = (const f16*)r.d_c2t; Lc.bias = r.d_c2b;
                Lc.out = out; Lc.out_gstride = G; Lc.K = 64; Lc.N = N; Lc.H = H; Lc.W = W; Lc.act = 0;
                Lc.res1 = cur; Lc.res1_gstride = G; Lc.s1 = 0.2f;
                if (k == 2) { Lc.res2 = x; Lc.res2_gstride = G; }
                Lc.s2 = 0.2f; Lc.y0 = 0; Lc.y1 = H; Lc.out_mode = OUT_SLAB; Lc.conv1x1 = 1;
                Lc.prefix_lrelu = r.d_dw3[0] ? 1 : 0;          // COMB holds d1 .. d8 as the dilated convs wrote them
                CK(conv_launch(Lc, s));
            } else {
            CK(gg::launch(r.d_c2, 256, 64, COMB, G, N, H, W, raw2, H, W, 1, 1, d0, d0, H, W, 1, 0, 0, 0, s));
            hipLaunchKernelGGL(ppon_res, dim3((unsigned)((px * 16 + 255) / 256)), dim3(256), 0, s, (const float*)raw2, (const float*)r.d_c2b, px,
                               cur, k == 2 ? x : (const f16*)nullptr, out, G);
            INNFER_HIP(hipGetLastError());
            }
            cur = out;
        }
        return INNFER_OK;
    };
    // CFEM trunk: nb RRBlocks rotating over T[0..3] (input, output and two scratch slabs are distinct), then LR_conv + shortcut
    const f16* x = FEA;
    for (int b = 0; b < p->nb; ++b) {
        f16* dst = T[b % 4];
        CK(rrblock(x, dst, T[(b + 1) % 4], T[(b + 2) % 4]));
        x = dst;
    }
    CK(conv(p->lr, x, G, CFEM, G, H, W, 0, 0, FEA, G, OUT_SLAB));
    auto recon = [&](const Head& Hd, const f16* feat, void* out) -> int {
        const f16* t = feat;
        long tg = G;
        int h = H, w = W;
        for (int u = 0; u < p->n_up; ++u) {
            f16* dst = (f16*)(ws + cv.up[u]);
            if (p->scale == 3) {          // Upsample(nearest 3x) materialised in the (still unused) HR slab, then conv -> LeakyReLU
                f16* U = (f16*)(ws + cv.hr);
                const long g3 = tg * 9, nthr = (long)N * 9 * h * w * 4 * 2;
                hipLaunchKernelGGL(ppon_upsample3, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, t, tg, U, g3, N, h, w);
                INNFER_HIP(hipGetLastError());
                CK(conv(Hd.up[u], U, g3, dst, g3, 3 * h, 3 * w, 1, 0, nullptr, 0, OUT_SLAB));
                t = dst; tg = g3; h *= 3; w *= 3;
                continue;
            }
            const long go = tg * 4;
            if (Hd.up[u].d_up4) {       // upconv_block as the four 2x2-tap phases of the equivalent transposed conv on the LR grid
                ConvLaunch L{};
                L.in = t; L.in_gstride = tg; L.C = Hd.up[u].C;
                const bool one_visit = Hd.up[u].d_up4p && w > 16;          // all four phases in one visit of a tile, plane-order panels (grids <= 16 wide: image pairs, one phase per visit)
                L.wpk = (const f16*)(one_visit ? Hd.up[u].d_up4p : Hd.up[u].d_up4); L.bias = Hd.up[u].d_b4;
                L.out = dst; L.out_gstride = go; L.K = 4 * Hd.up[u].K; L.phase_c = Hd.up[u].K; L.deconv_phases = one_visit ? 1 : 2; L.rowp = one_visit ? 1 : 0;
                L.N = N; L.H = h; L.W = w; L.act = 1; L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = h; L.out_mode = OUT_SLAB;
                CK(conv_launch(L, s));
            } else {
                CK(conv(Hd.up[u], t, tg, dst, go, 2 * h, 2 * w, 1, 1, nullptr, 0, OUT_SLAB));
            }
            t = dst; tg = go; h *= 2; w *= 2;
        }
        if (Hd.hr1.d_fuse && conv_fuse_side_bytes(N, h, w) <= (size_t)tg * 2 * (p->nf / 32)) {
            // HR_conv0 -> conv_last of the head as one kernel; the rim buffer lives where the HR slab would have been
            ConvLaunch L{};
            L.in = t; L.in_gstride = tg; L.C = Hd.hr0.C;
            L.wpk = (const f16*)Hd.hr0.d_w; L.bias = Hd.hr0.d_b;
            L.out = ws + cv.hr; L.out_gstride = tg; L.K = Hd.hr0.K;
            L.N = N; L.H = h; L.W = w; L.act = 1; L.s1 = L.s2 = 1.f; L.y0 = 0; L.y1 = h; L.out_mode = OUT_SLAB;
            L.fuse_w = (const f16*)Hd.hr1.d_fuse; L.fuse_bias = Hd.hr1.d_b; L.fuse_side = (float*)(ws + cv.hr); L.fuse_out = out; L.fuse_oc = Hd.hr1.K;
            L.fuse_out_mode = f32o ? 1 : 0;
            if (conv_fuse_last_ok(L)) { CK(conv_launch(L, s)); return INNFER_OK; }
        }
        CK(conv(Hd.hr0, t, tg, ws + cv.hr, tg, h, w, 1, 0, nullptr, 0, OUT_SLAB));
        CK(conv(Hd.hr1, (const f16*)(ws + cv.hr), tg, out, 0, h, w, 0, 0, nullptr, 0, OUT_NCHW));
        return INNFER_OK;
    };
    const long nout = (long)N * p->out_nc * H * W * p->scale * p->scale;
    CK(recon(p->heads[0], CFEM, d_out_c));                                           // out_c
    CK(rrblock(CFEM, T[0], T[1], T[2]));                                             // SFEM
    CK(rrblock(T[0], SFEM, T[1], T[2]));
    CK(recon(p->heads[1], SFEM, d_out_s));                                           // out_s = SRM(sfem) + out_c
    hipLaunchKernelGGL(ppon_axpy, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, (const void*)d_out_s, (const void*)d_out_c, d_out_s, 1.0f, nout, f32o);
    CK(rrblock(SFEM, T[0], T[1], T[2]));                                             // PFEM
    CK(rrblock(T[0], CFEM, T[1], T[2]));                                             // (CFEM's slab is free now)
    CK(recon(p->heads[2], CFEM, d_out_p));                                           // out_p = alpha * PRM(pfem) + out_s
    hipLaunchKernelGGL(ppon_axpy, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, (const void*)d_out_p, (const void*)d_out_s, d_out_p, p->alpha, nout, f32o);
    INNFER_HIP(hipGetLastError());
#undef CK
    return INNFER_OK;
}
